for i in 1 2; do
for v in 0 1; do
DYT_G16=$v python bench.py --no-parity-mode --no-cpu-baseline --steps 20 --warmup 5 --host-batches 0 2>&1 >/dev/null | grep "timed\|other_kernels" | sed "s/^/g16=$v /" | cut -c1-2000
done; done
