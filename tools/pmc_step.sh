#!/bin/bash
# Whole-step counter evidence (north_star: "rocprof MFMA-utilisation and HBM GB/s against gfx950 peak"): three SEPARATE rocprofv3 --pmc
# passes (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE; only --kernel-trace next to them, as MI355X_MICROARCH.md
# prescribes) over tools/probes/ab_step.py = 4 full fine-tune steps at B=128, compact, no calibration forwards (PPREC=fp16|bf16 selects the operand type, default bf16).
# usage (GPU box, repo root): tools/pmc_step.sh <tag>     -> gpurun_out/<tag>_step_traffic.json, gpurun_out/<tag>_step_mfma.json
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=/tmp/pmcs_${tag}_${c%% *}; rm -rf $d
  env PREPS=1 PSTEPS=2 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- python $root/tools/probes/ab_step.py > $root/gpurun_out/${tag}_pmcs_${c%% *}.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_pmcs_${c%% *}.log; exit 1; }
done
f=$(find /tmp/pmcs_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmcs_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
m=$(find /tmp/pmcs_${tag}_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1)
t=$(find /tmp/pmcs_${tag}_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1)
python $root/tools/pmc_step.py 4 "$f" "$w" "$m" "$t" $root/gpurun_out/${tag}_step_traffic.json $root/gpurun_out/${tag}_step_mfma.json
