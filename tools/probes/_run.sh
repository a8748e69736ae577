python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -s 2>&1 | grep -E "worst|fp16x3.*logits|passed|failed|Error|assert" | head -12
python bench.py --precision fp16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>&1 | tail -1 | cut -c150-260
DYT_LIB_DIR=tools/probes/_ab python -c "print(1)"
python bench.py --precision fp16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>&1 | tail -1 | cut -c150-260
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>&1 | tail -1 | cut -c150-260
