python -m pytest tests/test_gpu_round3.py -x -q -s -k "split" 2>&1 | grep -E "passed|failed|^E  |fp16x3|attention forward|Error" | head -20
for o in 1 0 1 0; do echo -n "split_attn=$o "; DYT_SPLIT_ATTN=$o PPREC=fp16x3 PSTEPS=8 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
EXTRA_ENV="PPREC=fp16x3" bash tools/probes/shape_times.sh x3b | grep -i "attn"
