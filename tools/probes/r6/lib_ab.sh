# same-box alternation of two builds of the libraries: build_ab/head (the committed source) vs the tree's
# usage: lib_ab.sh [bench args...]
for i in 1 2 3; do
for v in head tree; do
if [ $v = head ]; then export DYT_LIB_DIR=$PWD/build_ab/head; else unset DYT_LIB_DIR; fi
python bench.py --no-cpu-baseline --no-parity-mode --steps 20 --warmup 5 --host-batches 0 "$@" 2>&1 >/dev/null | grep "timed" | sed "s/^/$v /"
done; done
unset DYT_LIB_DIR
