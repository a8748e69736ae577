#!/bin/bash
# usage: ab_prof.sh <tag> [lib path]   -> gpurun_out/<tag>_kernel_stats.csv (serial launches)
tag=$1; lib=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
env DYT_NO_OVERLAP=1 $EXTRA_ENV ${lib:+DYT_LIB_PATH=$lib} rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag --output-format csv -- python $root/tools/probes/ab_step.py > /dev/null 2>&1
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $root/gpurun_out/${tag}_kernel_stats.csv
