python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "gemm or linear" 2>&1 | grep -E "passed|failed|^E  " | head
export PPREC=fp16
for o in 0 1 0 1; do echo -n "one_launch=$o "; DYT_GEMM_ROWS_ONE_LAUNCH=$o PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
for o in 0 1; do echo -n "serial one_launch=$o "; DYT_NO_OVERLAP=1 DYT_GEMM_ROWS_ONE_LAUNCH=$o PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
for o in 0 1; do echo -n "fp16x3 one_launch=$o "; PPREC=fp16x3 PSTEPS=8 DYT_GEMM_ROWS_ONE_LAUNCH=$o PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
