for g in 4 12; do echo "== gs 2^$g"; DYT_SPLIT_GS_LOG2=$g python -m pytest tests/test_gpu_round3.py -x -q -s -k "split and compact" 2>&1 | grep -E "passed|failed|^E  |worst" | head -4; done
