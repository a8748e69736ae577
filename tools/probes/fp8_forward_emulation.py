"""CPU emulation (no GPU, no library): logits / gate decisions of the whole DyT forward (the oracle's torch code) when every frozen-weight
GEMM (qkv, proj, fc1, fc2) is computed as  A_hi W_hi (IEEE half operands)  +  fp8 (e4m3) correction products  A_hi8 W_lo8 + A_lo8 W_hi8
with the FIXED power-of-two scales the HIP kernels use (activation hi part x 1, activation lo part x 2^12, weights: per-tensor exponent
from max|w|) -- against the fp32 oracle, next to plain fp16 operands (one product) and the three-product fp16 split.
    python tools/probes/fp8_forward_emulation.py [B]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-tuning_amd")]
import synth  # noqa: E402
from oracle import dyt_oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
MODE = {"kind": "exact"}
_linear = F.linear


def split16(x):
    hi = x.half()
    lo = x - hi.float()
    return hi.float(), lo


def e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def mx_block(x, mant_bits, emax, block=32):
    """MX-style: per 32-element block (last dim) a power-of-two scale putting the block max in the top binade; elements with
    `mant_bits` mantissa bits, exponent range [0, emax] + subnormals (e2m3: mant 3, emax 2; e2m1: mant 1, emax 2)."""
    sh = x.shape
    xb = x.reshape(-1, sh[-1] // block, block).double()
    mx = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-38)
    s = 2.0 ** (torch.floor(torch.log2(mx)) - emax)
    v = xb / s
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-38))).clamp(0, emax)
    q = 2.0 ** (e - mant_bits)
    top = (2.0 - 2.0 ** -mant_bits) * 2.0 ** emax
    v = (torch.round(v / q) * q).clamp(-top, top)
    return (v * s).reshape(sh).float()


_layer_norm = O.layer_norm


def tagged_layer_norm(x, w, b):
    t = _layer_norm(x, w, b)
    t._ln = (x, w, b)   # what a GEMM with the LayerNorm folded into it would read instead (kind "fp16fold")
    return t


O.layer_norm = tagged_layer_norm


def emu_linear(x, w, b=None):
    kind = MODE["kind"]
    if kind == "fp16fold":
        # LayerNorm folded into the GEMM that follows it: A = the 16-bit copy of the UN-normalised residual stream,
        # W' = half(gamma * W); y = rstd (A W'^T - mean colsum(W')) + (W beta + b), statistics from the fp32 stream
        ln = getattr(x, "_ln", None)
        frozen = w.shape[0] in (2304, 3072) or (w.shape[0] == 768 and w.shape[1] in (768, 3072))
        if not frozen:
            return _linear(x, w, b)
        if ln is None:
            return (x.half().double() @ w.half().double().T).float() + b
        xs, g, be = ln
        mu = xs.mean(dim=-1, keepdim=True)
        rstd = (xs.var(dim=-1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
        wg = (w * g).half().double()
        acc = xs.half().double() @ wg.T
        y = rstd.double() * (acc - mu.double() * wg.sum(dim=1)) + (w.double() @ be.double() + b.double())
        return y.float()
    frozen = w.shape[0] in (2304, 3072) or (w.shape[0] == 768 and w.shape[1] in (768, 3072))
    if kind == "exact" or not frozen:
        return _linear(x, w, b)
    if ":" in kind:   # "fp8c:qkv,fc1" = fp8 corrections for the named classes, three half products for the others
        cls = {(2304, 768): "qkv", (768, 768): "proj", (3072, 768): "fc1", (768, 3072): "fc2"}[tuple(w.shape)]
        kind = "fp8c" if cls in kind.split(":")[1].split(",") else "fp16x3"
    xh, xl = split16(x)
    wh, wl = split16(w)
    mm = lambda a, c: (a.double() @ c.double().T).float()
    if kind == "fp16":
        y = mm(xh, wh)
    elif kind == "fp16x3":
        y = mm(xh, wh) + mm(xh, wl.half().float()) + mm(xl.half().float(), wh)
    elif kind == "fp8c":
        ew = 7 - int(torch.ceil(torch.log2(w.abs().max())))
        sw, swl, sa, sal = 2.0 ** ew, 2.0 ** (ew + 11), 1.0, 2.0 ** 12
        y = mm(xh, wh) + mm(e4m3(xh * sa), e4m3(wl * swl)) / (sa * swl) + mm(e4m3(xl * sal), e4m3(wh * sw)) / (sal * sw)
    elif kind in ("fp6c", "fp4c"):
        mb = 3 if kind == "fp6c" else 1
        y = mm(xh, wh) + mm(mx_block(xh, mb, 2), mx_block(wl, mb, 2)) + mm(mx_block(xl, mb, 2), mx_block(wh, mb, 2))
    else:
        raise ValueError(kind)
    return y if b is None else y + b


F.linear = emu_linear
C, r = 100, 64
x, y = synth.make_batch(B, C, seed=31)
g1, g2 = synth.make_noise(B, seed=32)
keep = synth.make_dropout_masks(B, r, seed=33)
sd = synth.make_state_dict(C, r, seed=0, kind="test", gate_bias=0.85)
res = {}
with torch.no_grad():
    KINDS = ("fp16", "fp16x3", "fp8c", "fp6c", "fp4c") if len(sys.argv) < 3 else tuple(sys.argv[2:])
    for kind in ("exact",) + KINDS:
        MODE["kind"] = kind
        ls, tok = O.forward(sd, x, g1[0], g2[0], keep[0], scale=0.1, complete_model=False, training=True)[:2]
        lt = O.forward(sd, x, g1[1], g2[1], keep[1], scale=0.1, complete_model=True, training=True)[0]
        res[kind] = (ls, lt, tok["token_select"], tok["token_logits"])
ref = res["exact"]
for kind in KINDS:
    a = res[kind]
    flips = int((a[2] != ref[2]).sum())
    print("%-7s logits student %.2e teacher %.2e   gate logits %.2e   flips %d of %d" % (
        kind, float((a[0] - ref[0]).abs().max()), float((a[1] - ref[1]).abs().max()), float((a[3] - ref[3]).abs().max()), flips, ref[2].numel()))
