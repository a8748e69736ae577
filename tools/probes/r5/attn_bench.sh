#!/bin/bash
# usage: tools/probes/r5/attn_bench.sh <tag> [VAR=1 ...]  -> gpurun_out/<tag>_attn.txt (per-kernel avg times of the isolated attention fwd / bwd, B=128)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ab_$tag
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/ab_$tag -o $tag --output-format csv -- python $root/tools/probes/r5/attn_bench.py > $root/gpurun_out/${tag}_attn.log 2>&1
f=$(find /tmp/ab_$tag -name "*kernel_stats.csv" | head -1)
{ grep -E "^B=|done|rror" $root/gpurun_out/${tag}_attn.log; python $root/tools/prof_summary.py $f 12; } > $root/gpurun_out/${tag}_attn.txt
cat $root/gpurun_out/${tag}_attn.txt
