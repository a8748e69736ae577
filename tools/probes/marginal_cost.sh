#!/bin/bash
# usage: marginal_cost.sh  -> step time (B=128, fp16, compact, default overlapped schedule) with one BACKWARD kernel class dropped
# at a time (DYT_DBG_SKIP, csrc/kernels.h; no optimizer update, so the forward passes and the kept-token counts stay what they are):
# what each class costs in the overlapped schedule, where serial kernel durations do not add up.
# Needs a measurement build: make -C dynamic-tuning_amd/csrc clean all CXXFLAGS_EXTRA=-DDYT_DEBUG_HOOKS (product builds ignore DYT_DBG_SKIP / DYT_DBG_POISON)
for m in 0 1 4 8 16 64 128 256 384 511; do
    echo -n "skip(bwd)=$m  "
    DYT_DBG_SKIP=$((m + 1024)) PNOADAM=1 PPREC=${PPREC:-fp16} PREPS=1 python tools/probes/ab_step.py 2>&1 | tail -1
done
