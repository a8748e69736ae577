python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "gemm or linear" 2>&1 | grep -E "passed|failed|^E  " | head
export PPREC=fp16
for i in 1 2; do
echo -n "prev "; DYT_LIB_PATH=$(pwd)/tools/probes/_ab/prev_f16.so PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
echo -n "new  "; PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
echo -n "new one_launch "; DYT_GEMM_ROWS_ONE_LAUNCH=1 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
done
for o in 0 1; do echo -n "serial one_launch=$o "; DYT_NO_OVERLAP=1 DYT_GEMM_ROWS_ONE_LAUNCH=$o PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
