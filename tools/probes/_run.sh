python -m pytest tests/test_gpu_round3.py -x -q -s -k "fp16x3" 2>&1 | grep -E "worst|logits|passed|failed|Error|assert" | head -12
for v in 0 1; do echo "== DYT_SPLIT_PROD=$v"; DYT_SPLIT_PROD=$v python bench.py --precision fp16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>&1 | tail -1 | cut -c1-260; done
DYT_SPLIT_PROD=0 python bench.py --precision fp16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>&1 | tail -1 | cut -c150-260
DYT_SPLIT_PROD=1 python bench.py --precision fp16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>&1 | tail -1 | cut -c150-260
