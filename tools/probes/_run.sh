python -m pytest tests/test_gpu_round3.py -x -q -s -k "fp16x3" 2>&1 | grep -E "worst|logits|rel-L2|passed|failed|Error|assert" | head -30
for i in 1 2; do
for p in fp16x3f; do
PPREC=$p PSTEPS=10 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
PPREC=$p PSTEPS=10 PREPS=2 DYT_LIB_PATH=tools/probes/_ab/prev_f16.so python tools/probes/ab_step.py 2>&1 | tail -1
done; done
