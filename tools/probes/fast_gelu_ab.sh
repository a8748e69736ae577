# A/B of the A&S GELU in the split forms' fc1 epilogue (DYT_SPLIT_FAST_GELU), same box: step time and logits vs the oracle
for g in 0 1 0 1; do
  for p in fp16x3q fp16f8; do
    DYT_SPLIT_FAST_GELU=$g python bench.py --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>&1 >/dev/null | grep timed | sed "s/^/[fast_gelu=$g] /"
  done
done
DYT_SPLIT_FAST_GELU=1 python -m pytest tests/test_gpu_round4.py -q -m gpu -s -k "seeds and fp16x3q" 2>&1 | grep -E "seed|passed|failed"
