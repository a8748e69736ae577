set -x
export PPREC=fp16
python bench.py > gpurun_out/r3_head_bench.log 2>&1; grep '^{' gpurun_out/r3_head_bench.log > gpurun_out/r3_final_bench.json
python bench.py --precision bf16 --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_bf16_bench.json
python bench.py --mode masked --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_masked_bench.json
python bench.py --video-frames 8 --classes 400 --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_video_bench.json
python bench.py --precision fp16x3 --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_fp16x3_bench.json
tools/rocprof_bench.sh r3_final_ovl -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null
tools/rocprof_bench.sh r3_final_serial DYT_NO_OVERLAP=1 -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null
tools/rocprof_bench.sh r3_final_fp16x3_serial DYT_NO_OVERLAP=1 -- --precision fp16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > /dev/null
tools/pmc_step.sh r3_final
tools/pmc_bench.sh r3_final > /dev/null
EXTRA_ENV="PPREC=fp16" bash tools/probes/shape_times.sh r3_final_serial > /dev/null
python - <<'PY'
import json
for f in ("r3_final_bench","r3_final_bf16_bench","r3_final_masked_bench","r3_final_video_bench","r3_final_fp16x3_bench","r3_final_serial_bench"):
    d=json.load(open('gpurun_out/%s.json'%f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], d['dtype'], 'gemm', r['gemm_ms_per_step'], 'attn', r['attention_ms_per_step'], 'other', r['other_kernels_ms_per_step'], 'frac', r['frac'])
    for k in ('parity_mode','exact_mode','other_fast_mode'):
        m=d.get(k)
        if m: print('   ', k, m['dtype'], m['value'], m['ms_per_step'])
PY
