python -m pytest tests/test_gpu_round3.py -x -q -s -k split 2>&1 | grep -E "passed|failed|^E  |fp16x3|Error" | head
for o in 1 0 1 0; do echo -n "split_bpre=$o "; DYT_SPLIT_BPRE=$o PPREC=fp16x3 PSTEPS=8 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
