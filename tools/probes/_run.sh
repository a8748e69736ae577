python -m pytest tests/test_gpu_round2.py -x -q --deselect tests/test_gpu_round2.py::test_step_at_b16_vs_oracle 2>&1 | tail -5
python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -3
