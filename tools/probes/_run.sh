python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head
python bench.py > gpurun_out/r3_head_bench.log 2>&1; grep '^{' gpurun_out/r3_head_bench.log > gpurun_out/r3_final_bench.json
python bench.py --precision fp16x3 --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_fp16x3_bench.json
tools/rocprof_bench.sh r3_final_fp16x3_serial DYT_NO_OVERLAP=1 -- --precision fp16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > /dev/null
python - <<'PY'
import json
for f in ("r3_final_bench","r3_final_fp16x3_bench"):
    d=json.load(open('gpurun_out/%s.json'%f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], d['dtype'], 'gemm', r['gemm_ms_per_step'], 'attn', r['attention_ms_per_step'], 'other', r['other_kernels_ms_per_step'], 'frac', r['frac'], r['achieved'])
    for k in ('parity_mode','exact_mode','other_fast_mode'):
        m=d.get(k)
        if m: print('   ', k, m['dtype'], m['value'], m['ms_per_step'])
PY
