set -x
export PPREC=fp16
python bench.py > gpurun_out/r3_head_bench.log 2>&1; grep '^{' gpurun_out/r3_head_bench.log > gpurun_out/r3_final_bench.json; cut -c1-600 gpurun_out/r3_final_bench.json
tools/pmc_step.sh r3_final
cp gpurun_out/r3_final_gemm_traffic.json /dev/null 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or linear or scheduling or reproducible" 2>&1 | grep -E "passed|failed"
