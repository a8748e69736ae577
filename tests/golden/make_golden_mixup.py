#!/usr/bin/env python3
"""Generate tests/golden/mixup_step.npz by running the REAL reference model through its own train_one_epoch WITH a ``mixup_fn``
(engine_finetune.py:44-45: ``samples, targets = mixup_fn(samples, targets)``; the class-probability targets then reach
``criterion.base_criterion`` -- nn.CrossEntropyLoss -- for the teacher pass (:60) and, through AdaLoss (models/losses.py:53), for the student pass).
timm is not in the build container, so the callable is synth.mixup_batch: what timm.data.Mixup returns for a fixed mixing weight (mixed
samples, smoothed and mixed one-hot rows).  Same rules as make_golden.py: build container only, model / AdaLoss / train_one_epoch are the
reference's own code.
Usage:  python tests/golden/make_golden_mixup.py"""
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

synth = mg.synth


def main():
    batch, num_classes, ffn_num, seed, lam, smoothing = 4, 10, 8, 29, 0.7, 0.1
    torch.manual_seed(779)
    sd = synth.make_state_dict(num_classes, ffn_num, seed=seed, kind="test", gate_bias=0.3)
    tuning = mg.EasyDict(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                         ffn_adapter_scalar="0.1", ffn_num=ffn_num, d_model=768)
    model = mg.vit_base_patch16_224_in21k(num_classes=num_classes, drop_path_rate=0.0, tuning_config=tuning,
                                          select_config=mg.EasyDict(open=True, keep_layers=0))
    msg = model.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    for n, p in model.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    params = [p for n, p in model.named_parameters() if p.requires_grad]
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    lr, wd = 1e-3, 1e-4
    optimizer = torch.optim.AdamW(params, lr=lr, weight_decay=wd)
    criterion = mg.AdaLoss(base_criterion=nn.CrossEntropyLoss(), token_target_ratio=0.5, token_loss_ratio=2.0, token_minimal=0.0,
                           token_minimal_weight=0.0)
    scaler = mg.misc.NativeScalerWithGradNormCount()
    args = types.SimpleNamespace(accum_iter=1, lr=lr, min_lr=0.0, warmup_epochs=0, epochs=10, metric="accuracy", nb_classes=num_classes)
    grads = {}
    step_orig = optimizer.step

    def step_hook(*a, **k):
        grads.update({n: p.grad.detach().clone() for n, p in zip(names, params)})
        return step_orig(*a, **k)
    optimizer.step = step_hook
    rec = []
    h = model.register_forward_hook(lambda m, i, o: rec.append((o[0].detach().clone(), o[1]["token_select"].detach().clone())))
    x, y = synth.make_batch(batch, num_classes, seed=seed)
    keep = synth.make_dropout_masks(batch, ffn_num, seed=seed + 3)
    seen = {}

    def mixup_fn(samples, targets):
        xm, t = synth.mixup_batch(samples, targets, num_classes, lam=lam, smoothing=smoothing)
        seen["targets"] = t.clone()
        return xm, t

    with mg.Recorder(keep) as r:
        stats = mg.engine_finetune.train_one_epoch(model, criterion, [(x, y)], optimizer, torch.device("cpu"), 0, scaler, None, mixup_fn, None,
                                                   args=args, logger=logging.getLogger("golden"))
    h.remove()
    g1, g2 = r.gumbels(2, 12, batch)
    (ls, ts), (lt, _) = rec[0], rec[1]
    out = {"meta_batch": batch, "meta_num_classes": num_classes, "meta_ffn_num": ffn_num, "meta_seed": seed, "meta_gate_bias": 0.3,
           "meta_lr": lr, "meta_wd": wd, "meta_lam": lam, "meta_smoothing": smoothing, "g1": g1.numpy(), "g2": g2.numpy(),
           "soft_targets": seen["targets"].numpy(), "logits_student": ls.numpy(), "logits_teacher": lt.numpy(),
           "token_select": ts.numpy().astype(np.uint8)}
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        out["stat_" + k] = np.float64(stats[k])
    for n, gr in grads.items():
        out["gradnorm/" + n] = np.float64(gr.double().norm())
        if not ("adaptmlp" in n and n.endswith("proj.weight")) or n.startswith("blocks.6.") or n.startswith("blocks.11.") or n.startswith("blocks.0."):
            out["grad/" + n] = gr.numpy()
    for n, p in zip(names, params):
        if n.startswith("head."):
            out["param_after/" + n] = p.detach().numpy().copy()
    print("mixup_step.npz", {k: round(float(v), 6) for k, v in stats.items()})
    np.savez_compressed(os.path.join(HERE, "mixup_step.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    logging.basicConfig(level=logging.WARNING)
    main()
