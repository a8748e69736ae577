python bench.py > gpurun_out/r3_final_bench.log 2>&1; grep '^{' gpurun_out/r3_final_bench.log > gpurun_out/r3_final_bench.json
python bench.py --precision fp16x3 --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>&1 | grep '^{' > gpurun_out/r3_final_fp16x3_bench.json
tools/rocprof_bench.sh r3_final_fp16x3_serial DYT_NO_OVERLAP=1 -- --precision fp16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
cut -c1-300 gpurun_out/r3_final_bench.json; cut -c1-300 gpurun_out/r3_final_fp16x3_bench.json
