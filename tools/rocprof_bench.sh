#!/bin/bash
# usage (on the GPU box, from the repo root): tools/rocprof_bench.sh <tag> [env VAR=1 ...] -- <bench.py args>
# rocprofv3 --kernel-trace --stats of one bench.py run; leaves gpurun_out/<tag>_kernel_stats.csv (copy to profiles/).
set -e
tag=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env "${envs[@]}" rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag --output-format csv -- \
    python "$root/bench.py" "$@" > "$root/gpurun_out/${tag}_bench.log" 2>&1 || { tail -20 "$root/gpurun_out/${tag}_bench.log"; exit 1; }
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
grep '^{' "$root/gpurun_out/${tag}_bench.log" > "$root/gpurun_out/${tag}_rocprof_bench.json" || true
head -25 "$root/gpurun_out/${tag}_kernel_stats.csv" | cut -c1-170
