#!/bin/bash
# usage: pmc_exe.sh <tag> <kernel-substring> "<command>" <counters...>   one --pmc pass over an arbitrary command; per-kernel averages
tag=$1; pat=$2; cmd=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -o $tag --output-format csv -- $cmd > /tmp/pmc_$tag.log 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" "$pat" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    key = k.split("(")[0][-48:]
    d[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(key, r["Counter_Name"])] += 1
for key, c in d.items():
    print(key, {kk: round(v / n[(key, kk)]) for kk, v in c.items()})
PY
