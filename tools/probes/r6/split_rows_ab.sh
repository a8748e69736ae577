# DYT_SPLIT_ROWS=0: the N = 768 GEMMs that take a 256x256 body (128 KB of LDS, one workgroup per CU) + a 128x128 row tail as 128x128 tiles throughout
# (64 KB, two per CU, co-resident with the other pass's kernels)
for i in 1 2 3; do
for v in 1 0; do
DYT_SPLIT_ROWS=$v python bench.py --no-cpu-baseline --no-parity-mode --steps 20 --warmup 5 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/split_rows=$v /"
done; done
for v in 1 0; do
DYT_SPLIT_ROWS=$v python bench.py --precision fp16x3q --no-cpu-baseline --steps 10 --warmup 3 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/split_rows=$v /"
done
