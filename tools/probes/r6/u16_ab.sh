for i in 1 2; do
for v in 0 1; do
DYT_TOK_U16=$v python bench.py --no-parity-mode --no-cpu-baseline --steps 20 --warmup 5 --host-batches 0 2>&1 >/dev/null | grep -o "timed 20 steps: [0-9.]* ms/step\|.other_kernels_ms_per_step.: [0-9.]*" | tr '\n' ' ' | sed "s/^/tok_u16=$v /"; echo
done; done
