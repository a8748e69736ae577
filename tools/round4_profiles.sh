#!/bin/bash
# Everything profiles/round4/ holds for the final code, in one gpurun call (GPU box, repo root): tools/round4_profiles.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "MALL|DRAM|HBM|EA0_RDREQ|EA0_WRREQ" | head -40) > gpurun_out/r4_counter_list.txt 2>&1
bash tools/pmc_bench.sh r4 > /dev/null 2>&1
cp gpurun_out/r4_gemm_traffic.json profiles/round4/gemm_traffic.json 2>/dev/null
python bench.py > gpurun_out/r4_final_benchline.json 2> gpurun_out/r4_final_benchline.err
bash tools/rocprof_bench.sh r4_final -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
bash tools/rocprof_bench.sh r4_final_serial DYT_NO_OVERLAP=1 -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
for p in fp16x3q fp16f8 fp16x3h fp32; do
  bash tools/rocprof_bench.sh r4_final_${p}_serial DYT_NO_OVERLAP=1 -- --precision $p --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
done
EXTRA_ENV=PPREC=fp16 bash tools/probes/shape_times.sh r4_final_serial > /dev/null 2>&1
EXTRA_ENV=PPREC=fp16x3q bash tools/probes/shape_times.sh r4_final_fp16x3q_serial > /dev/null 2>&1
EXTRA_ENV=PPREC=fp16f8 bash tools/probes/shape_times.sh r4_final_fp16f8_serial > /dev/null 2>&1
PPREC=fp16 bash tools/pmc_step.sh r4_final > /dev/null 2>&1
PPREC=fp16x3q bash tools/pmc_step.sh r4_final_fp16x3q > /dev/null 2>&1
ls gpurun_out | grep r4_final
