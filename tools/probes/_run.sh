python -m pytest tests/test_gpu_round3.py -x -q -s 2>&1 | grep -E "passed|failed|^E  |fp16x3|Error" | head -20
python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or linear or golden" 2>&1 | grep -E "passed|failed|^E  " | head
for p in fp16x3 fp32; do echo -n "$p "; PPREC=$p PSTEPS=6 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
export PPREC=fp16
for i in 1 2; do
echo -n "prev "; DYT_LIB_PATH=$(pwd)/tools/probes/_ab/prev_f16.so PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
echo -n "new  "; PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
done
