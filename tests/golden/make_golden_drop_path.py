#!/usr/bin/env python3
"""Generate tests/golden/drop_path_step.npz by running the REAL reference model with drop_path_rate > 0 on CPU.

Same rules as make_golden.py (build container only; data, not code, is committed).  The reference's blocks take ``DropPath`` from timm
(models/vision_transformer_IN21K.py:16,121,131), which is not installed here; the stand-in below restates timm==0.9.12
``timm/layers/drop.py::drop_path`` (per-sample ``bernoulli_(keep_prob)`` of shape [B,1,1], divided by keep_prob) with the Bernoulli
draws taken from RECORDED uniforms (``u < keep_prob``), in the order the reference's two forward passes call the modules
(block 1 .. 11: drop_path1 then drop_path2; block 0 holds nn.Identity because dpr[0] = 0, :285).  Everything else -- the model, AdaLoss,
``engine_finetune.train_one_epoch``, the freeze rule -- is the reference's own code, as in make_golden.py.
Usage:  python tests/golden/make_golden_drop_path.py
"""
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the third-party stand-ins, puts the reference on sys.path)

synth = mg.synth
RATE = 0.3


class Draws:
    u = None      # [passes, 2, depth, B] uniforms in [0, 1)
    calls = []    # (pass, block, branch) in call order


def _drop_path_forward(self, x):
    if self.p == 0. or not self.training:
        return x
    keep = 1.0 - self.p
    ps, blk, br = self._where
    npass = sum(1 for c in Draws.calls if c[1:] == (blk, br))
    Draws.calls.append((npass, blk, br))
    mask = (Draws.u[npass, br, blk] < keep).to(x.dtype).reshape(-1, 1, 1)
    return x * (mask / keep)


def main():
    batch, num_classes, ffn_num, scalar, seed = 3, 10, 8, "1.0", 11
    torch.manual_seed(4321)
    sd = synth.make_state_dict(num_classes, ffn_num, seed=seed, kind="test", gate_bias=0.3)
    tuning = mg.EasyDict(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none",
                         ffn_adapter_init_option="lora", ffn_adapter_scalar=scalar, ffn_num=ffn_num, d_model=768)
    select = mg.EasyDict(open=True, keep_layers=0)
    model = mg.vit_base_patch16_224_in21k(num_classes=num_classes, drop_path_rate=RATE, tuning_config=tuning, select_config=select)
    msg = model.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    for n, p in model.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    dpr = torch.linspace(0, RATE, 12)
    cls = None
    for i, blk in enumerate(model.blocks):
        if i == 0:
            assert isinstance(blk.drop_path1, nn.Identity) and isinstance(blk.drop_path2, nn.Identity)
            continue
        for br, m in enumerate((blk.drop_path1, blk.drop_path2)):
            assert abs(m.p - float(dpr[i])) < 1e-7, (i, m.p, float(dpr[i]))
            m._where = (None, i, br)
            cls = type(m)
    cls.forward = _drop_path_forward

    params = [p for n, p in model.named_parameters() if p.requires_grad]
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4)
    criterion = mg.AdaLoss(base_criterion=nn.CrossEntropyLoss(), token_target_ratio=0.5, token_loss_ratio=2.0,
                           token_minimal=0.0, token_minimal_weight=0.0)
    scaler = mg.misc.NativeScalerWithGradNormCount()
    args = types.SimpleNamespace(accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=10, metric="accuracy", nb_classes=num_classes)
    grads = {}
    step_orig = optimizer.step

    def step_hook(*a, **k):
        grads.update({n: p.grad.detach().clone() for n, p in zip(names, params)})
        return step_orig(*a, **k)
    optimizer.step = step_hook
    rec = []
    h = model.register_forward_hook(lambda m, i, o: rec.append((o[0].detach().clone(), o[1]["token_select"].detach().clone(),
                                                                o[1]["token_logits"].detach().clone())))
    x, y = synth.make_batch(batch, num_classes, seed=seed)
    keep = synth.make_dropout_masks(batch, ffn_num, seed=seed + 3)
    g = torch.Generator().manual_seed(seed + 5)
    Draws.u = torch.rand(2, 2, 12, batch, generator=g)
    Draws.calls = []
    with mg.Recorder(keep) as r:
        stats = mg.engine_finetune.train_one_epoch(model, criterion, [(x, y)], optimizer, torch.device("cpu"), 0, scaler, None, None, None,
                                                   args=args, logger=logging.getLogger("golden"))
    h.remove()
    g1, g2 = r.gumbels(2, 12, batch)
    # the order the reference called the modules in: pass 0 then pass 1, blocks 1 .. 11, branch 0 then 1
    want = [(ps, i, br) for ps in range(2) for i in range(1, 12) for br in range(2)]
    assert Draws.calls == want, Draws.calls[:6]
    keep_l = (1.0 - dpr).reshape(1, 1, 12, 1)
    scales = (Draws.u < keep_l).float() / keep_l
    scales[:, :, 0] = 1.0
    dropped = int((scales[:, :, 1:] == 0).sum())
    assert dropped >= 6, dropped   # the fixture must exercise dropped branches in both passes
    (ls, ts, tl), (lt, _, _) = rec[0], rec[1]
    out = {"meta_batch": batch, "meta_num_classes": num_classes, "meta_ffn_num": ffn_num, "meta_scale": float(scalar), "meta_seed": seed,
           "meta_gate_bias": 0.3, "meta_rate": RATE, "drop_uniforms": Draws.u.numpy(), "drop_scales": scales.numpy(),
           "g1": g1.numpy(), "g2": g2.numpy(), "logits_student": ls.numpy(), "logits_teacher": lt.numpy(),
           "token_select": ts.numpy().astype(np.uint8), "token_logits": tl.numpy()}
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        out["stat_" + k] = np.float64(stats[k])
    for n, gr in grads.items():
        out["gradnorm/" + n] = np.float64(gr.double().norm())
        if not ("adaptmlp" in n and n.endswith("proj.weight")) or n.startswith("blocks.6.") or n.startswith("blocks.11."):
            out["grad/" + n] = gr.numpy()
    print("drop_path_step.npz", {k: round(float(v), 6) for k, v in stats.items()}, "dropped branches", dropped,
          "keep", float(ts.float().mean()))
    np.savez_compressed(os.path.join(HERE, "drop_path_step.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    logging.basicConfig(level=logging.WARNING)
    main()
