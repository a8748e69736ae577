python -m pytest tests/test_gpu_parity.py -x -q -k "scheduling or reproducible or default_schedule or full_size" 2>&1 | grep -E "passed|failed|^E  " | head
python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | grep -E "passed|failed|^E  " | head
