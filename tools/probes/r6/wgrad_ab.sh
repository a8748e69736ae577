for i in 1 2 3 4; do
python bench.py --no-parity-mode --no-cpu-baseline --steps 30 --warmup 5 --host-batches 0 2>&1 >/dev/null | grep -o "timed 30 steps: [0-9.]* ms/step" | sed "s/^/new /"
DYT_LIB_DIR=/root/repo/build_ab/prev python bench.py --no-parity-mode --no-cpu-baseline --steps 30 --warmup 5 --host-batches 0 2>&1 >/dev/null | grep -o "timed 30 steps: [0-9.]* ms/step" | sed "s/^/prev /"
done
