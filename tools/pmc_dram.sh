#!/bin/bash
# DRAM-side share of the step's L2 memory-side requests: TCC_EA0_RDREQ vs TCC_EA0_RDREQ_DRAM and
# TCC_EA0_WRREQ vs TCC_EA0_WRREQ_DRAM, summed over all kernels of tools/probes/ab_step.py (separate --pmc passes, --kernel-trace only).
# usage (GPU box, repo root): PPREC=fp16 tools/pmc_dram.sh <tag>  -> gpurun_out/<tag>_dram_split.txt
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/${tag}_dram_split.txt; : > $out
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
  d=/tmp/pmcd_${tag}_${c%% *}; rm -rf $d
  env PREPS=1 PSTEPS=2 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- python $root/tools/probes/ab_step.py > /tmp/pmcd.log 2>&1 || { tail -3 /tmp/pmcd.log >> $out; continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "fillBuffer" in r["Kernel_Name"]:
        continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print("%-28s %.4e requests over the profiled launches" % (k, v))
PY
done
cat $out
