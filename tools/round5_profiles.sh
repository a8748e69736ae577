#!/bin/bash
# Everything profiles/round5/ holds for the final code, in one gpurun call (GPU box, repo root): tools/round5_profiles.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "MALL|DRAM|HBM|EA0_RDREQ|EA0_WRREQ" | head -40) > gpurun_out/r5_counter_list.txt 2>&1
bash tools/pmc_bench.sh r5 > /dev/null 2>&1
cp gpurun_out/r5_gemm_traffic.json profiles/round5/gemm_traffic.json 2>/dev/null
python bench.py > gpurun_out/r5_final_benchline.json 2> gpurun_out/r5_final_benchline.err
bash tools/rocprof_bench.sh r5_final -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
bash tools/rocprof_bench.sh r5_final_serial DYT_NO_OVERLAP=1 -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
for p in fp16x3q fp32; do
  bash tools/rocprof_bench.sh r5_final_${p}_serial DYT_NO_OVERLAP=1 -- --precision $p --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
done
EXTRA_ENV=PPREC=fp16 bash tools/probes/shape_times.sh r5_final_serial > /dev/null 2>&1
EXTRA_ENV=PPREC=fp16x3q bash tools/probes/shape_times.sh r5_final_fp16x3q_serial > /dev/null 2>&1
PPREC=fp16 bash tools/pmc_step.sh r5_final > /dev/null 2>&1
PPREC=fp16x3q bash tools/pmc_step.sh r5_final_fp16x3q > /dev/null 2>&1
ls gpurun_out | grep r5_final
