python tools/probes/dk_diag.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_round3.py -x -q -s 2>&1 | grep -E "attention backward|attention forward|passed|failed|rel" | head -20
python bench.py --precision fp16x3 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-400
