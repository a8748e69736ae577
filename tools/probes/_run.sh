export PPREC=fp16
for m in none xcd a53 none xcd a53; do echo -n "mask=$m "; PMASK=$m PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1; done
