python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head
python -m pytest tests/test_gpu_round2.py -x -q -k "b16_vs_oracle or additive or rccl" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head
export PPREC=fp16
for i in 1 2; do
echo -n "prev "; DYT_LIB_PATH=$(pwd)/tools/probes/_ab/prev_f16.so PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
echo -n "new  "; PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
done
