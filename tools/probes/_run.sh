python -m pytest tests/test_gpu_round3.py -x -q -s -k fp16x3 2>&1 | grep -E "worst|fp16x3.*logits|passed|failed|Error|assert" | head -12
for i in 1 2; do
PPREC=fp16x3 python tools/probes/ab_step.py 2>&1 | tail -1
PPREC=fp16x3 DYT_LIB_PATH=tools/probes/_ab/prev_fold.so python tools/probes/ab_step.py 2>&1 | tail -1
done
