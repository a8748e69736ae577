#!/bin/bash
# usage: timeline.sh <tag> -> rocprofv3 --kernel-trace of the overlapped step, analysed by timeline.py
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$tag
env PSTEPS=6 PREPS=1 $EXTRA_ENV rocprofv3 --kernel-trace -d /tmp/tl_$tag -o $tag --output-format csv -- python $root/tools/probes/ab_step.py > /dev/null 2>&1
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
head -1 $f
python $root/tools/probes/timeline.py $f 2 > $root/gpurun_out/${tag}_timeline.txt
python $root/tools/probes/timeline.py $f 2 full > $root/gpurun_out/${tag}_timeline_full.txt
cat $root/gpurun_out/${tag}_timeline.txt
