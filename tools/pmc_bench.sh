#!/bin/bash
# HBM traffic of the bench step from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with traces
# beyond --kernel-trace), as MI355X_MICROARCH.md's HBM section prescribes.  Writes gpurun_out/<tag>_gemm_traffic.json.
# usage (GPU box, repo root): tools/pmc_bench.sh <tag> [bench.py args]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${tag}_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${tag}_$c -o p --output-format csv -- \
      python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode "$@" > $root/gpurun_out/${tag}_pmc_$c.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_pmc_$c.log; exit 1; }
done
f=$(find /tmp/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python $root/tools/pmc_traffic.py "$f" "$w" gemm_bf16_ > $root/gpurun_out/${tag}_gemm_traffic.json 2> $root/gpurun_out/${tag}_traffic_total.txt
cat $root/gpurun_out/${tag}_gemm_traffic.json $root/gpurun_out/${tag}_traffic_total.txt
