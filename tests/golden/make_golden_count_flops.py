#!/usr/bin/env python3
"""Golden vectors for the FLOP-probe variant, Block.forward_count_flops (reference models/vision_transformer_IN21K.py:
167-185), switched on exactly as block_flops_dict.get_block_flops :33-55 does (apply(setattr) of `count_flops` /
`token_select_num`).  Runs the REAL reference from /root/reference on CPU behind the stand-ins of make_golden.py (build
container only); commits data only.  Usage: python tests/golden/make_golden_count_flops.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the shims, imports the reference)

synth = G.synth

if __name__ == "__main__":
    torch.set_num_threads(8)
    C, r, seed, gate_bias, B = 100, 64, 3, 0.3, 2
    sd = synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=gate_bias)
    model, tuning, select = G.build_reference(C, r, "0.1", sd)
    model.eval()
    x, _ = synth.make_batch(B, C, seed=seed)
    out = {"meta_batch": B, "meta_num_classes": C, "meta_ffn_num": r, "meta_scale": 0.1, "meta_gate_bias": gate_bias,
           "meta_seed": seed, "tokens": np.array([1, 57, 197])}
    model.apply(lambda m: setattr(m, "count_flops", True))
    for n in (1, 57, 197):
        model.apply(lambda m: setattr(m, "token_select_num", n))
        with torch.no_grad():
            t = model.patch_embed(x)     # forward_features (:343-371) unpacks a tuple per block; the probe variant returns
            t = torch.cat((model.cls_token.expand(B, -1, -1), t), dim=1) + model.pos_embed   # a tensor, so walk the blocks here
            for blk in model.blocks:
                t = blk(t)
            logits = model.head(model.norm(t)[:, 0])
        out["logits_n%d" % n] = logits.numpy()
        print("n", n, float(logits.abs().max()))
    np.savez_compressed(os.path.join(HERE, "count_flops.npz"), **out)
