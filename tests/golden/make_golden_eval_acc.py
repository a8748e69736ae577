#!/usr/bin/env python3
"""tests/golden/eval_acc.npz: an evaluate() fixture whose metrics are NOT degenerate (eval_r64.npz's random-weight top-1 is 0.0).

Runs the REAL reference (build container only, see make_golden.py for the stand-ins): eval forward of 12 seeded images with a
10-class head, then targets are CHOSEN from the ranking of the reference's own logits -- 5 images get their top-1 class, 3 a class
ranked 3rd / 4th (top-5 but not top-1), 4 a class ranked 8th or lower -- so that the reference's evaluate()
(engine_finetune.py:208-279) returns acc1 = 41.67 and its accuracy() (util/metrics.py:4-11) acc5 = 66.67, with every decision at
least `meta_margin` away from a rank swap (so the bf16 mode must reproduce them too).  Also the mean-per-class metric
(util/metrics.py:14-25) of the same predictions.  Usage: python tests/golden/make_golden_eval_acc.py"""
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the stand-ins, imports the reference)
from util.metrics import accuracy  # noqa: E402  (the reference's)

synth = G.synth
RANKS = [0, 0, 0, 0, 0, 2, 2, 3, 7, 7, 8, 9]
MARGIN = 0.08


def main():
    B, C, r, seed, gate_bias = len(RANKS), 10, 64, 11, 0.3
    sd = synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=gate_bias)
    model, tuning, select = G.build_reference(C, r, "0.1", sd)
    x, _ = synth.make_batch(B, C, seed=seed)
    model.eval()
    with torch.no_grad():
        logits, _ = model(x)
    order = logits.argsort(dim=1, descending=True)
    srt = logits.gather(1, order)
    def margins(i, rk):   # distance of image i's top-1 / top-5 decisions from a rank swap if its target is its rank-rk class
        t = float(srt[i, rk])
        m1 = float(srt[i, 0] - srt[i, 1]) if rk == 0 else float(srt[i, 0]) - t
        m5 = t - float(srt[i, 5]) if rk < 5 else float(srt[i, 4]) - t
        return min(m1, m5)
    ranks, free = [None] * B, list(range(B))
    for rk in RANKS:   # greedy: the first image not used yet whose decisions keep the margin for this rank
        i = next(i for i in free if margins(i, rk) > MARGIN)
        ranks[i] = rk
        free.remove(i)
    y = torch.stack([order[i, ranks[i]] for i in range(B)])
    log = logging.getLogger("golden")
    out = {"meta_batch": B, "meta_num_classes": C, "meta_ffn_num": r, "meta_scale": 0.1, "meta_gate_bias": gate_bias, "meta_seed": seed,
           "meta_margin": MARGIN, "ranks": np.array(ranks), "targets": y.numpy(), "logits": logits.numpy()}
    for metric in ("accuracy", "mean_per_class_acc"):
        args = types.SimpleNamespace(metric=metric, nb_classes=C)
        rec = []
        h = model.register_forward_hook(lambda m, i, o: rec.append(o))
        status = G.engine_finetune.evaluate([(x[:5], y[:5]), (x[5:9], y[5:9]), (x[9:], y[9:])], model, torch.device("cpu"), log, None, None, args)
        h.remove()
        out["metric_" + metric] = np.float64(status["metric"])
        preds = torch.cat([o[0] for o in rec])
        assert torch.equal(preds, logits)
    a1, a5 = accuracy(logits, y, topk=(1, 5))
    assert abs(float(a1) - out["metric_accuracy"]) < 1e-9
    out["acc5"] = np.float64(a5)
    out["token_select_mean"] = np.float64(torch.cat([o[1]["token_select"] for o in rec]).float().mean())
    print("eval_acc.npz acc1 %.4f acc5 %.4f mean-per-class %.4f keep %.4f" % (out["metric_accuracy"], out["acc5"],
                                                                                out["metric_mean_per_class_acc"], out["token_select_mean"]))
    np.savez_compressed(os.path.join(HERE, "eval_acc.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    logging.basicConfig(level=logging.WARNING)
    main()
