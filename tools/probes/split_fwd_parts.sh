# products of the FORWARD GEMM classes of the fp16x3f mode (measurement knob DYT_SPLIT_FWD_PARTS="qkv,proj,fc1,fc2"; 3 = shipped):
# logits / gate decisions vs the oracle at B=16 and the B=128 step time -> profiles/round3/r3_forward_parts_probe.txt
for fp in 1,1,1,1 2,2,2,2 1,1,1,3 3,3,1,1 1,1,3,3 3,1,1,1; do
echo "=== fwd parts (qkv,proj,fc1,fc2) = $fp"
DYT_SPLIT_FWD_PARTS=$fp python -m pytest tests/test_gpu_round3.py -x -q -s -k "fp16x3_mode and compact and fp16x3f" 2>&1 | grep -E "logits|worst" | head -3
DYT_SPLIT_FWD_PARTS=$fp PPREC=fp16x3f PSTEPS=10 PREPS=1 python tools/probes/ab_step.py 2>&1 | tail -1
done
