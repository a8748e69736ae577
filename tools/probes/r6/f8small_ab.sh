for i in 1 2; do
for v in 1 2; do
DYT_F8_SHORTK_SMALL=$v python bench.py --precision fp16x3q --no-cpu-baseline --steps 10 --warmup 3 --host-batches 0 2>&1 >/dev/null | grep -o "timed 10 steps: [0-9.]* ms/step\|.gemm_ms_per_step.: [0-9.]*" | tr '\n' ' ' | sed "s/^/f8_shortk_small=$v /"; echo
done; done
