#!/usr/bin/env python3
"""Per-gradient error table of the exact-fp32 and fp16x3 modes against the CPU oracle at B=16 (a debugging aid, not collected by
pytest): where along the backward pass does the split mode's error enter?  Usage: python tests/diag_grad_table.py [masked|compact]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import synth  # noqa: E402
from test_gpu_round2 import _bench_model  # noqa: E402
from oracle import dyt_oracle as O  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "compact"
B, C, r, target = 16, 100, 64, 0.5
x, y = synth.make_batch(B, C, seed=31)
g1, g2 = synth.make_noise(B, seed=32)
keep = synth.make_dropout_masks(B, r, seed=33)
sd = synth.make_state_dict(C, r, seed=0, kind="test", gate_bias=0.85)
d_ref, g_ref, _ = O.step_grads(sd, x, y, g1, g2, keep, scale=0.1, mode=mode, token_target_ratio=target)
table, rowinfo = {}, {}
for prec in ("fp32", "fp16x3"):
    m, _ = _bench_model(prec, mode, B, 0.85, classes=C, r=r, kind="test")
    m.train()
    eng = m.engine(B, torch.device("cuda", 0))
    eng.step_fwd_bwd(x.cuda(), y.cuda(), target, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                     g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous())
    for n, gr in g_ref.items():
        if gr.numel() == 1:
            continue
        got = eng.trainable_view(n, gr.shape, eng.grad).cpu()
        table.setdefault(n, []).append(float((got - gr).norm() / (gr.norm() + 1e-20)))
        if prec == "fp16x3" and n.endswith("down_proj.weight"):   # a ReLU-mask flip (pre-activation ~ 0) touches ONE row of this gradient
            rows = (got - gr).norm(dim=1) / (gr.norm() + 1e-20)
            top = torch.topk(rows, 2)
            rowinfo[n] = "row %d carries %.2e, the next one %.2e" % (int(top.indices[0]), float(top.values[0]), float(top.values[1]))
print("%-46s %10s %10s" % ("gradient", "fp32", "fp16x3"))
for n, (a, b) in table.items():
    print("%-46s %10.2e %10.2e%s" % (n, a, b, ("   <-- " + rowinfo.get(n, "")) if b > 10 * a else ""))
