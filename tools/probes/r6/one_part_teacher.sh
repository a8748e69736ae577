# teacher (complete_model) pass of fp16x3q with some GEMM classes as the hi * hi product alone: five-seed parity + step time
for m in 0 1 3 2; do
echo "=== DYT_ONE_PART_COMPLETE=$m (qkv 1, proj 2)"
DYT_ONE_PART_COMPLETE=$m python -m pytest tests/test_gpu_round4.py -q -s -k "parity_modes_vs_oracle_over_seeds and fp16x3q" 2>&1 | grep -E "seed|passed|failed|logits" | head -12
DYT_ONE_PART_COMPLETE=$m python bench.py --precision fp16x3q --no-cpu-baseline --steps 10 --warmup 3 --host-batches 0 2>&1 >/dev/null | grep "timed"
done
