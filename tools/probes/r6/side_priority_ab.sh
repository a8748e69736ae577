# DYT_SIDE_PRIORITY: the teacher (complete_model) pass's stream above / below the caller's stream in priority
for i in 1 2 3; do
for v in 0 -1 1; do
DYT_SIDE_PRIORITY=$v python bench.py --no-cpu-baseline --no-parity-mode --steps 20 --warmup 5 --host-batches 0 2>&1 >/dev/null | grep "timed" | sed "s/^/side_priority=$v /"
done; done
