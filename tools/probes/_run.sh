export PPREC=fp16
for i in 1 2 3; do
echo -n "prev "; DYT_LIB_PATH=$(pwd)/tools/probes/_ab/prev_f16.so PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
echo -n "new  "; PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
done
python -m pytest tests/test_gpu_round2.py -x -q -k "b16_vs_oracle" -s 2>&1 | grep -E "passed|failed|^E  |compact worst" | head
python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -k "fc2 or scheduling or reproducible or default_schedule" 2>&1 | grep -E "passed|failed|^E  " | head
EXTRA_ENV="PPREC=fp16" bash tools/probes/shape_times.sh tokfix | grep -E "tok_bwd|ln_bwd|EpiFc2"
