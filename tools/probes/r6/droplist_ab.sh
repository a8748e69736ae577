for i in 1 2; do
for v in 0 1; do
DYT_CAT3_DROP_LIST=$v python bench.py --precision fp16x3q --no-cpu-baseline --steps 10 --warmup 3 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/drop_list=$v /"
done; done
