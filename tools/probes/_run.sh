python -m pytest tests/test_gpu_round3.py -x -q -s -k split 2>&1 | grep -E "passed|failed|^E  |fp16x3|Error" | head
python -m pytest tests/test_gpu_parity.py -x -q -k "attention or golden" 2>&1 | grep -E "passed|failed|^E  " | head
for p in fp16x3 fp16x3 fp32; do echo -n "$p "; PPREC=$p PSTEPS=8 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
