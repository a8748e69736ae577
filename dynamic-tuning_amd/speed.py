#!/usr/bin/env python3
"""Inference-throughput harness of the reference (speed.py:240-275, measure_speed.sh) on MI355X.

Same protocol: eval mode, no grad, ``synchronize`` around every batch, the first 6 iterations are
skipped, stop after iteration 20, print images/s.  The model is the eval-mode forward of this repo
(deterministic gate, MLP on the compacted kept tokens == models/model_speed_test.py:274-310) on
synthetic images; weights are random like the reference's (it loads the timm checkpoint non-strictly,
so adapters/gates are random there too, speed.py:229-233)."""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from models.vision_transformer_IN21K import vit_base_patch16_224_in21k  # noqa: E402
from block_flops_dict import batch_select_flops, get_base_flops, get_block_flops  # noqa: E402


class Cfg(dict):
    __getattr__ = dict.__getitem__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--ffn_num", type=int, default=64)
    ap.add_argument("--nb_classes", type=int, default=100)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--keep", type=float, default=0.7, help="target keep ratio of the deterministic (eval) gate")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="leave the gate biases at logit(keep) (the eval gate then keeps whatever fraction has a positive logit)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    tuning = Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none",
                 ffn_adapter_init_option="lora", ffn_adapter_scalar="0.1", ffn_num=args.ffn_num, d_model=768)
    model = vit_base_patch16_224_in21k(num_classes=args.nb_classes, drop_path_rate=0.0, tuning_config=tuning,
                                       select_config=Cfg(open=True, keep_layers=0), precision=args.precision,
                                       max_batch=args.batch_size)
    model.load_state_dict(synth.make_state_dict(args.nb_classes, args.ffn_num, kind="bench",
                                                gate_bias=math.log(args.keep / (1 - args.keep))))
    model = model.to(dev).eval()
    x, _ = synth.make_batch(args.batch_size, args.nb_classes, seed=0)
    x = x.to(dev)
    if not args.no_calibrate:
        # the reference learns the keep ratio; the harness pins it: shift every block's gate bias so that the target quantile
        # of its token logits sits at the decision threshold 0 (three sweeps: shifting a block moves the ones after it)
        with torch.no_grad():
            for _ in range(3):
                _, aux = model(x)
                tl = aux["token_logits"].float()                      # [B, depth, 196, 1] or [B, depth, 196]
                tl = tl.reshape(tl.shape[0], tl.shape[1], -1).permute(1, 0, 2).reshape(tl.shape[1], -1)
                q = torch.quantile(tl, 1.0 - args.keep, dim=1)
                for i, blk in enumerate(model.blocks):
                    blk.mlp_token_select.mlp_head.bias.sub_(q[i].to(blk.mlp_token_select.mlp_head.bias.device))
    sample, total = 0, 0.0
    aux = None
    with torch.no_grad():
        for i in range(21):
            torch.cuda.synchronize()
            t0 = time.time()
            out, aux = model(x)
            torch.cuda.synchronize()
            t1 = time.time()
            if i <= 5:
                continue
            sample += x.shape[0]
            total += t1 - t0
    flops = batch_select_flops(x.shape[0], get_block_flops(ffn_num=args.ffn_num), aux["token_select"].cpu(), 12, get_base_flops())
    print("throughput {} img/s".format(sample / total))
    print("keep ratio %.4f ; average %.3f GMACs/img (%.1f %% of ViT-B/16's 17.6)" % (
        float(aux["token_select"].mean()), float(flops.mean()), 100 * float(flops.mean()) / 17.6))


if __name__ == "__main__":
    main()
