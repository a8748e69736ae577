python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | grep -E "passed|failed|^E  " | head -5
