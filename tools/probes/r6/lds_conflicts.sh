#!/bin/bash
# per-kernel LDS bank-conflict share of one fine-tune step (serial launches): rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE over
# tools/probes/ab_step.py, aggregated by kernel name.  usage (GPU box, repo root): PPREC=fp16 bash tools/probes/r6/lds_conflicts.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ldsc_$tag
env DYT_NO_OVERLAP=1 PREPS=1 PSTEPS=2 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/ldsc_$tag -o p --output-format csv -- python $root/tools/probes/ab_step.py > $root/gpurun_out/${tag}_ldsc.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_ldsc.log; exit 1; }
f=$(find /tmp/ldsc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" > $root/gpurun_out/${tag}_lds_conflicts.txt <<'PY'
import csv, re, sys
agg = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"gemm_(bf16_bpre|bf16_nt|f32_mfma_nt)_kernel.*?(Epi[A-Za-z0-9]+)", n)
    if m:
        t = re.search(r"ILi(\d+)ELi(\d+)", n)
        short = "%s%s %s" % (m.group(1), ("[%sx%s]" % t.groups()) if t else "", m.group(2)[:22])
    else:
        mm = re.search(r"(attn_[a-z_0-9]+|[a-z_0-9]+_kernel)", n)
        short = mm.group(1) if mm else n[:30]
    a = agg.setdefault(short, {"SQ_LDS_BANK_CONFLICT": 0.0, "SQ_LDS_IDX_ACTIVE": 0.0, "n": 0})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["n"] += 1
for k, a in sorted(agg.items(), key=lambda x: -x[1]["SQ_LDS_IDX_ACTIVE"]):
    act = a["SQ_LDS_IDX_ACTIVE"]
    if act <= 0:
        continue
    print("%-44s launches=%5d lds_active=%.3e bank_conflict=%.3e (%.1f %%)" % (k, a["n"] // 2, act, a["SQ_LDS_BANK_CONFLICT"], 100 * a["SQ_LDS_BANK_CONFLICT"] / act))
PY
cat $root/gpurun_out/${tag}_lds_conflicts.txt
