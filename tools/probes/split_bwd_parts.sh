# gradient products of the fp16x3 mode's backward: GEMM contraction parts (DYT_SPLIT_BWD_PARTS) x attention gradient products
# (DYT_SPLIT_BWD_ATTN_PARTS): worst gradients vs the oracle at B=16 (tests/diag_grad_table.py) and the B=128 step time
for cfg in "3 3" "1 3" "1 1"; do
  set -- $cfg
  echo "=== gemm parts $1, attention gradient parts $2"
  DYT_SPLIT_BWD_PARTS=$1 DYT_SPLIT_BWD_ATTN_PARTS=$2 python tests/diag_grad_table.py compact 2>&1 | awk '/blocks|head/' | sort -k3 -g | tail -4
  DYT_SPLIT_BWD_PARTS=$1 DYT_SPLIT_BWD_ATTN_PARTS=$2 PPREC=fp16x3 PSTEPS=10 PREPS=2 python tools/probes/ab_step.py 2>&1 | tail -1
done
