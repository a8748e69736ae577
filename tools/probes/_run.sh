export PPREC=fp16
for i in 1 2 3; do
echo -n "prev "; DYT_LIB_PATH=$(pwd)/tools/probes/_ab/prev_f16.so PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
echo -n "new  "; PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
done
echo -n "serial prev "; DYT_NO_OVERLAP=1 DYT_LIB_PATH=$(pwd)/tools/probes/_ab/prev_f16.so PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
echo -n "serial new  "; DYT_NO_OVERLAP=1 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
