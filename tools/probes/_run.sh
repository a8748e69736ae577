set -x
export PPREC=fp16
python bench.py > gpurun_out/r3_head_bench.log 2>&1; grep '^{' gpurun_out/r3_head_bench.log > gpurun_out/r3_final_bench.json; cut -c1-330 gpurun_out/r3_final_bench.json
python bench.py --precision bf16 --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_bf16_bench.json
python bench.py --mode masked --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_masked_bench.json
python bench.py --video-frames 8 --classes 400 --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_video_bench.json
tools/rocprof_bench.sh r3_final_ovl -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null
tools/rocprof_bench.sh r3_final_serial DYT_NO_OVERLAP=1 -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null
tools/pmc_step.sh r3_final
tools/pmc_bench.sh r3_final > /dev/null
EXTRA_ENV="PPREC=fp16" bash tools/probes/shape_times.sh r3_final_serial > /dev/null
tools/probes/marginal_cost.sh > gpurun_out/r3_final_marginal_cost.txt 2>&1; (echo serial; DYT_NO_OVERLAP=1 tools/probes/marginal_cost.sh) >> gpurun_out/r3_final_marginal_cost.txt 2>&1
python dynamic-tuning_amd/speed.py 2>&1 | tail -2
