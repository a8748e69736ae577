python -m pytest tests/test_gpu_round3.py -x -q -s -k split 2>&1 | grep -E "passed|failed|^E  |fp16x3|Error" | head -20
for p in fp16x3 fp32; do echo -n "$p "; PPREC=$p PSTEPS=6 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
python -m pytest tests/test_gpu_parity.py -x -q -k "golden or scheduling or full_size or batch_of_one" 2>&1 | grep -E "passed|failed|^E  " | head
