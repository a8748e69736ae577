#!/bin/bash
# Everything profiles/round6/ holds for the final code, in one gpurun call (GPU box, repo root): tools/round6_profiles.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "MALL|DRAM|HBM|EA0_RDREQ|EA0_WRREQ" | head -40) > gpurun_out/r6_counter_list.txt 2>&1
bash tools/pmc_bench.sh r6 > /dev/null 2>&1
cp gpurun_out/r6_gemm_traffic.json profiles/round6/gemm_traffic.json 2>/dev/null
python bench.py > gpurun_out/r6_final_benchline.json 2> gpurun_out/r6_final_benchline.err
bash tools/rocprof_bench.sh r6_final -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
bash tools/rocprof_bench.sh r6_final_serial DYT_NO_OVERLAP=1 -- --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
for p in fp16x3q fp32; do
  bash tools/rocprof_bench.sh r6_final_${p}_serial DYT_NO_OVERLAP=1 -- --precision $p --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
done
EXTRA_ENV=PPREC=fp16 bash tools/probes/shape_times.sh r6_final_serial > /dev/null 2>&1
EXTRA_ENV=PPREC=fp16x3q bash tools/probes/shape_times.sh r6_final_fp16x3q_serial > /dev/null 2>&1
PPREC=fp16 bash tools/pmc_step.sh r6_final > /dev/null 2>&1
PPREC=fp16x3q bash tools/pmc_step.sh r6_final_fp16x3q > /dev/null 2>&1
DYT_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --no-parity-mode --no-cpu-baseline --host-batches 0 > gpurun_out/r6_dist1.json 2> gpurun_out/r6_dist1.err
for f in r6_final_benchline.json r6_final_kernel_stats.csv r6_final_serial_kernel_stats.csv r6_final_fp16x3q_serial_kernel_stats.csv r6_final_fp32_serial_kernel_stats.csv \
         r6_final_serial_shape_times.txt r6_final_fp16x3q_serial_shape_times.txt r6_final_step_traffic.json r6_final_step_mfma.json \
         r6_final_fp16x3q_step_traffic.json r6_final_fp16x3q_step_mfma.json r6_dist1.json r6_counter_list.txt; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/round6/$f
done
ls -la profiles/round6 | tail -20
