"""Deterministic synthetic weights / inputs for the DyT ViT-B/16 hot path.

There is no network on the build or GPU boxes, so checkpoints (the timm IN-21K
``.pth`` that ``main_image.py:219-247`` loads) cannot be fetched.  Everything that
needs weights -- parity tests, golden fixtures, ``bench.py`` -- gets them from this
recipe instead.  The generator is numpy's PCG64 + ziggurat ``standard_normal`` (plain
C double arithmetic, no SIMD-dependent paths), so the SAME bits come out in the build
container and on the GPU box; torch's own CPU ``randn`` is avoided on purpose.

State-dict keys and shapes are exactly the reference's (SURVEY.md section 8b;
``models/vision_transformer_IN21K.py:272-320`` and ``models/dynamic_adapter.py:61,105-107``).
"""
import os
import zlib

import numpy as np
import torch

DEPTH = 12
DIM = 768
HEADS = 12
MLP_HIDDEN = 3072
NUM_PATCHES = 196
NUM_TOKENS = 197


POOL_SHAPES = {  # video model only (video_models/video_vision_transformer_IN21K.py:27-75,407-410)
    "query_token": (1, 1, DIM),
    "attentive_blocks.norm_q.weight": (DIM,), "attentive_blocks.norm_q.bias": (DIM,),
    "attentive_blocks.norm_k.weight": (DIM,), "attentive_blocks.norm_k.bias": (DIM,),
    "attentive_blocks.norm_v.weight": (DIM,), "attentive_blocks.norm_v.bias": (DIM,),
    "attentive_blocks.cross_attn.q_bias": (DIM,), "attentive_blocks.cross_attn.v_bias": (DIM,),
    "attentive_blocks.cross_attn.q.weight": (DIM, DIM), "attentive_blocks.cross_attn.k.weight": (DIM, DIM),
    "attentive_blocks.cross_attn.v.weight": (DIM, DIM),
    "attentive_blocks.cross_attn.proj.weight": (DIM, DIM), "attentive_blocks.cross_attn.proj.bias": (DIM,),
}


def param_shapes(num_classes=100, ffn_num=64, depth=DEPTH, video=False):
    """Ordered ``name -> shape`` for the reference model's ``state_dict()`` (video=True: the video model's)."""
    shapes = {
        "cls_token": (1, 1, DIM),
        "pos_embed": (1, NUM_TOKENS, DIM),
        "patch_embed.proj.weight": (DIM, 3, 16, 16),
        "patch_embed.proj.bias": (DIM,),
    }
    for i in range(depth):
        p = "blocks.%d." % i
        shapes[p + "norm1.weight"] = (DIM,)
        shapes[p + "norm1.bias"] = (DIM,)
        shapes[p + "attn.qkv.weight"] = (3 * DIM, DIM)
        shapes[p + "attn.qkv.bias"] = (3 * DIM,)
        shapes[p + "attn.proj.weight"] = (DIM, DIM)
        shapes[p + "attn.proj.bias"] = (DIM,)
        shapes[p + "norm2.weight"] = (DIM,)
        shapes[p + "norm2.bias"] = (DIM,)
        shapes[p + "mlp.fc1.weight"] = (MLP_HIDDEN, DIM)
        shapes[p + "mlp.fc1.bias"] = (MLP_HIDDEN,)
        shapes[p + "mlp.fc2.weight"] = (DIM, MLP_HIDDEN)
        shapes[p + "mlp.fc2.bias"] = (DIM,)
        shapes[p + "adaptmlp.down_proj.weight"] = (ffn_num, DIM)
        shapes[p + "adaptmlp.down_proj.bias"] = (ffn_num,)
        shapes[p + "adaptmlp.up_proj.weight"] = (DIM, ffn_num)
        shapes[p + "adaptmlp.up_proj.bias"] = (DIM,)
        shapes[p + "mlp_token_select.mlp_head.weight"] = (1, DIM)
        shapes[p + "mlp_token_select.mlp_head.bias"] = (1,)
    shapes["norm.weight"] = (DIM,)
    shapes["norm.bias"] = (DIM,)
    shapes["head.weight"] = (num_classes, DIM)
    shapes["head.bias"] = (num_classes,)
    if video:
        shapes.update(POOL_SHAPES)
    return shapes


def is_trainable(name):
    """Freeze rule of ``main_image.py:250-256``: adapters, gates and the head train."""
    return (("adaptmlp." in name) or ("mlp_token_select." in name) or name.startswith("head.") or
            name == "query_token" or name.startswith("attentive_blocks."))   # video: missing from the ViT checkpoint


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def _normal(name, shape, seed, std, mean=0.0):
    a = _rng(name, seed).standard_normal(size=shape) * std + mean
    return torch.from_numpy(a.astype(np.float32))


def make_state_dict(num_classes=100, ffn_num=64, seed=0, kind="test", depth=DEPTH,
                    gate_bias=0.0, video=False):
    """Synthetic weights.

    kind="bench": what the reference's own init gives (trunc-normal(0.02) linears with
        zero biases, LN weight 1 / bias 0, ``pos_embed ~ 0.02 N``, ``cls ~ 1e-6 N``;
        ``models/vision_transformer_IN21K.py:285,321-332``) except ``up_proj ~ N(0,0.02)``
        (the reference zero-inits it, ``dynamic_adapter.py:112-117``, which would leave the
        adapter path idle) and the gate bias set to ``gate_bias`` (keep-ratio calibration,
        SURVEY.md section 8d).
    kind="test": additionally randomises every bias and LayerNorm affine so that no term
        of the forward/backward is multiplied by an exact 0 or 1 in the parity tests.
    """
    sd = {}
    test = kind == "test"
    for name, shape in param_shapes(num_classes, ffn_num, depth, video).items():
        if name == "cls_token":
            t = _normal(name, shape, seed, 0.02 if test else 1e-6)
        elif name == "pos_embed":
            t = _normal(name, shape, seed, 0.02)
        elif name == "query_token":   # reference init is zeros (:407): LN of a constant row would be degenerate
            t = _normal(name, shape, seed, 0.02 if not test else 0.5)
        elif ".norm" in name or name.startswith("norm."):
            if name.endswith("weight"):
                t = _normal(name, shape, seed, 0.1, 1.0) if test else torch.ones(shape)
            else:
                t = _normal(name, shape, seed, 0.1) if test else torch.zeros(shape)
        elif name.endswith("mlp_token_select.mlp_head.bias"):
            t = torch.full(shape, float(gate_bias))
            if test:
                t = t + _normal(name, shape, seed, 0.1)
        elif name.endswith("bias"):
            t = _normal(name, shape, seed, 0.02) if test else torch.zeros(shape)
        elif name == "patch_embed.proj.weight":
            t = _normal(name, shape, seed, 0.02)
        elif name == "head.weight":
            t = _normal(name, shape, seed, 0.02 if test else 0.01)
        else:
            t = _normal(name, shape, seed, 0.02)
        sd[name] = t.contiguous()
    return sd


def add_learnable_scales(sd, seed=0, depth=DEPTH):
    """``blocks.i.adaptmlp.scale`` [1] for ``ffn_adapter_scalar == "learnable_scalar"`` (reference init 1.0; here distinct values in
    0.4 .. 1.6 so that a wrong block or a missing factor shows)."""
    g = torch.Generator().manual_seed(9000 + seed)
    for i in range(depth):
        sd["blocks.%d.adaptmlp.scale" % i] = 0.4 + 1.2 * torch.rand(1, generator=g)
    return sd


def add_adapter_layernorm(sd, seed=0, depth=DEPTH):
    """``blocks.i.adaptmlp.adapter_layer_norm_before.{weight,bias}`` [768] for ``ffn_adapter_layernorm_option`` "in" / "out" (reference
    init: ones / zeros, models/dynamic_adapter.py:97; here distinct values so that the parameters and their gradients are tested).  The keys
    sit right behind the block's gate (registration order of the reference's Adapter: layer norm first) -- order is irrelevant to the tests."""
    g = torch.Generator().manual_seed(1000 + seed)
    out = dict(sd)
    for i in range(depth):
        out["blocks.%d.adaptmlp.adapter_layer_norm_before.weight" % i] = 0.7 + 0.6 * torch.rand(DIM, generator=g)
        out["blocks.%d.adaptmlp.adapter_layer_norm_before.bias" % i] = 0.2 * torch.randn(DIM, generator=g)
    return out


def make_batch(batch, num_classes=100, seed=0):
    """Images ``N(0,1)`` [B,3,224,224] fp32 and int64 targets."""
    x = _normal("images", (batch, 3, 224, 224), seed, 1.0)
    y = _rng("targets", seed + 1).integers(0, num_classes, size=(batch,))
    return x, torch.from_numpy(y.astype(np.int64))


def mixup_batch(x, y, num_classes, lam=0.7, smoothing=0.1):
    """What timm.data.Mixup(mixup_alpha > 0, label_smoothing) hands the training loop (reference engine_finetune.py:44-45), with a fixed mixing
    weight instead of a Beta draw: samples lam x + (1 - lam) x.flip(0), class-probability targets lam t(y) + (1 - lam) t(y.flip(0)) with
    t(y) = one_hot (1 - smoothing) + smoothing / C.  Deterministic, so the golden recipe and the tests apply the same function."""
    import torch
    xm = lam * x + (1.0 - lam) * x.flip(0)
    off = smoothing / num_classes
    t = torch.full((y.shape[0], num_classes), off, dtype=torch.float32, device=y.device).scatter_(1, y.view(-1, 1), 1.0 - smoothing + off)
    return xm, lam * t + (1.0 - lam) * t.flip(0)


def make_noise(batch, depth=DEPTH, seed=2, passes=2):
    """Gumbel draws ``g1, g2 = -log(Exp(1))`` shaped [passes, depth, B, 196] each
    (what ``dynamic_adapter.py:30-39`` draws per block and per pass)."""
    r = _rng("gumbel", seed)
    e = r.standard_exponential(size=(2, passes, depth, batch, NUM_PATCHES))
    g = -np.log(e)
    return (torch.from_numpy(g[0].astype(np.float32)), torch.from_numpy(g[1].astype(np.float32)))


def make_dropout_masks(batch, ffn_num=64, depth=DEPTH, seed=3, p=0.1, passes=2):
    """Bernoulli(1-p) keep masks for the adapter dropout, uint8 [passes, depth, B*197, r]."""
    r = _rng("dropout", seed)
    u = r.random(size=(passes, depth, batch * NUM_TOKENS, ffn_num))
    return torch.from_numpy((u >= p).astype(np.uint8))


def available_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota.  (The GPU
    box shows 256 logical CPUs under a 16-CPU quota; torch's default 128 threads are throttled to a
    crawl there, so every CPU-side timing / oracle run sets torch.set_num_threads(available_cores()).)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n
