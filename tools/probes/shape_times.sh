#!/bin/bash
# usage: shape_times.sh <tag>  -> gpurun_out/<tag>_shape_times.txt (serial launches of tools/probes/ab_step.py)
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_$tag
env DYT_NO_OVERLAP=1 PSTEPS=6 PREPS=1 $EXTRA_ENV rocprofv3 --kernel-trace -d /tmp/st_$tag -o $tag --output-format csv -- python $root/tools/probes/ab_step.py > /dev/null 2>&1
f=$(find /tmp/st_$tag -name "*kernel_trace.csv" | head -1)
python $root/tools/probes/shape_times.py $f 0.3 > $root/gpurun_out/${tag}_shape_times.txt
cat $root/gpurun_out/${tag}_shape_times.txt
