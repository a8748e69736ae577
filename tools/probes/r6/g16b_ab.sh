for i in 1 2; do
for v in 0 1; do
DYT_G16_B16=$v python bench.py --precision fp16x3q --no-cpu-baseline --steps 10 --warmup 3 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/g16_b16=$v /" | cut -c1-100
done; done
DYT_G16_B16=1 python -m pytest tests/test_gpu_round4.py -q -m gpu -s -k "parity_modes and (fp16x3q or fp16x3h)" 2>&1 | grep "seed\|passed\|failed"
