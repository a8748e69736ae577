for sc in 1 0.05 0.003; do echo "== A scale $sc"; PA_SCALE=$sc python tools/probes/split_precision_probe.py 2>&1 | grep "fp16 qkv" | cut -c1-140; done
