for i in 1 2; do
for v in 0 1; do
DYT_FC2_CAT3=$v python bench.py --precision fp16x3q --no-cpu-baseline --steps 10 --warmup 3 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/cat3=$v /"
done; done
