#!/bin/bash
# usage: clock_watch.sh <tag> -- <command ...>   samples sclk / mclk / power (rocm-smi) every 0.5 s while <command> runs
tag=$1; shift; [ "$1" == "--" ] && shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}_clocks.txt
: > $out
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' ' >> $out
    echo >> $out
    sleep 0.5
done
wait $pid
