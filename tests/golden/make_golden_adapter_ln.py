#!/usr/bin/env python3
"""Generate tests/golden/adapter_ln_in_step.npz / adapter_ln_out_step.npz by running the REAL reference model with
``tuning_config.ffn_adapter_layernorm_option = "in"`` / ``"out"`` on CPU (models/dynamic_adapter.py:88,95-98: the adapter owns a trainable
``nn.LayerNorm(768)`` -- default eps 1e-5 -- applied to its input (:121-122) or to its scaled output (:132-133); "in" is the Adapter class's
default, the shipped scripts pass "none").  Same rules as make_golden.py: build container only, the model / AdaLoss / train_one_epoch are the
reference's own code; the LayerNorm parameters are set to distinct values (synth.add_adapter_layernorm) before the step.
Usage:  python tests/golden/make_golden_adapter_ln.py"""
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


def one(option):
    batch, num_classes, ffn_num, seed = 3, 10, 8, 17
    torch.manual_seed(778)
    sd = synth.add_adapter_layernorm(synth.make_state_dict(num_classes, ffn_num, seed=seed, kind="test", gate_bias=0.3), seed=seed)
    tuning = mg.EasyDict(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option=option, ffn_adapter_init_option="lora",
                         ffn_adapter_scalar="0.1", ffn_num=ffn_num, d_model=768)
    model = mg.vit_base_patch16_224_in21k(num_classes=num_classes, drop_path_rate=0.0, tuning_config=tuning,
                                          select_config=mg.EasyDict(open=True, keep_layers=0))
    msg = model.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    for n, p in model.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    params = [p for n, p in model.named_parameters() if p.requires_grad]
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert sum("adapter_layer_norm_before" in n for n in names) == 24 and len(names) == 98
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
    with mg.Recorder(keep) as r:
        stats = mg.engine_finetune.train_one_epoch(model, criterion, [(x, y)], optimizer, torch.device("cpu"), 0, scaler, None, None, None,
                                                   args=args, logger=logging.getLogger("golden"))
    h.remove()
    g1, g2 = r.gumbels(2, 12, batch)
    (ls, ts), (lt, _) = rec[0], rec[1]
    out = {"meta_batch": batch, "meta_num_classes": num_classes, "meta_ffn_num": ffn_num, "meta_seed": seed, "meta_gate_bias": 0.3,
           "meta_lr": lr, "meta_wd": wd, "meta_option": np.int64({"in": 1, "out": 2}[option]), "g1": g1.numpy(), "g2": g2.numpy(),
           "logits_student": ls.numpy(), "logits_teacher": lt.numpy(), "token_select": ts.numpy().astype(np.uint8)}
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        out["stat_" + k] = np.float64(stats[k])
    for n, gr in grads.items():
        out["gradnorm/" + n] = np.float64(gr.double().norm())
        if not ("adaptmlp" in n and n.endswith("proj.weight")) or n.startswith("blocks.6.") or n.startswith("blocks.11.") or n.startswith("blocks.0."):
            out["grad/" + n] = gr.numpy()
    for n, p in zip(names, params):
        if "adapter_layer_norm_before" in n and (n.startswith("blocks.0.") or n.startswith("blocks.6.") or n.startswith("blocks.11.")):
            out["param_after/" + n] = p.detach().numpy().copy()
    print("adapter_ln_%s_step.npz" % option, {k: round(float(v), 6) for k, v in stats.items()},
          "|d gamma| blocks 0/6/11", [round(float(grads["blocks.%d.adaptmlp.adapter_layer_norm_before.weight" % i].norm()), 6) for i in (0, 6, 11)])
    np.savez_compressed(os.path.join(HERE, "adapter_ln_%s_step.npz" % option), **out)


def main():
    for option in ("in", "out"):
        one(option)


if __name__ == "__main__":
    torch.set_num_threads(8)
    logging.basicConfig(level=logging.WARNING)
    main()
