for i in 1 2; do
PPREC=fp16x3f PSTEPS=10 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
DYT_SPLIT_SHORTK_SMALL=2 PPREC=fp16x3f PSTEPS=10 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
done
