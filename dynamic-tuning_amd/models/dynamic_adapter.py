"""``models.dynamic_adapter`` of the reference, host side (reference: models/dynamic_adapter.py).

``TokenSelect`` and ``Adapter`` carry the trainable parameters under the reference's names
(``mlp_head.{weight,bias}``, ``down_proj.*``, ``up_proj.*``) and keep its constructor /
attribute surface (``set_tau``, ``.tau``, ``.threshold``, ``.scale``, ``.dropout``).  Inside a
``VisionTransformer`` their arithmetic is executed by the fused HIP path of libdyt_hip.so
(gate + compaction kernel, adapter GEMMs); ``TokenSelect.forward`` stand-alone calls the same
gate kernel through the C ABI.  There is no CPU implementation.
"""
import ctypes
import math

import torch
import torch.nn as nn

from _lib import DyTError, check, lib, ptr, stream_ptr


def _gumbel_sigmoid(logits, tau=1, hard=False, eps=1e-10, training=True, threshold=0.5, gumbels=None):
    """Reference models/dynamic_adapter.py:25-54 (same signature; ``gumbels=(g1, g2)`` optionally
    injects the two Gumbel draws).  Element-wise utility kept for API parity; the model path
    evaluates the same formula inside the gate kernel (csrc/rowops.hip: gate_kernel)."""
    if training:
        if gumbels is None:
            g1 = -torch.empty_like(logits).exponential_().log()
            g2 = -torch.empty_like(logits).exponential_().log()
        else:
            g1, g2 = gumbels
        y_soft = ((logits + g1 - g2) / tau).sigmoid()
    else:
        y_soft = logits.sigmoid()
    if hard:
        y_hard = torch.zeros_like(logits).masked_fill(y_soft > threshold, 1.0)
        return y_hard - y_soft.detach() + y_soft
    return y_soft


class _LinearParams(nn.Module):
    """Parameter holder with nn.Linear's names/shapes/init (weight [out,in], bias [out])."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "in_features=%d, out_features=%d, bias=%s" % (self.in_features, self.out_features, self.bias is not None)


class _AdapterLayerNormParams(nn.Module):
    """nn.LayerNorm(dim)'s parameter surface (``weight`` ones, ``bias`` zeros; eps 1e-5)."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class TokenSelect(nn.Module):
    """Reference models/dynamic_adapter.py:58-77."""

    def __init__(self, dim_in, num_sub_layer, tau=5, is_hard=True, threshold=0.5, bias=True):
        super().__init__()
        assert num_sub_layer == 1 and is_hard and bias, "the HIP gate kernel implements the configuration the reference uses"
        self.mlp_head = _LinearParams(dim_in, num_sub_layer, bias=bias)
        self.is_hard = is_hard
        self.tau = tau
        self.threshold = threshold

    def set_tau(self, tau):
        self.tau = tau

    def forward(self, x, gumbels=None):
        """x [B,197,768] on a HIP device -> (token_select [B,197,1], logits [B,196,1]); forward only."""
        if not x.is_cuda:
            raise DyTError("TokenSelect runs on the HIP device only")
        if self.training and gumbels is None:
            shape = (x.shape[0], x.shape[1] - 1)
            gumbels = (-torch.empty(shape, device=x.device).exponential_().log(),
                       -torch.empty(shape, device=x.device).exponential_().log())
        B = x.shape[0]
        u = x.detach().float().contiguous()
        mask = torch.empty(B, 196, device=x.device)
        logits = torch.empty(B, 196, device=x.device)
        keep = torch.empty(B * 197, device=x.device, dtype=torch.int32)
        counts = torch.empty(B, device=x.device, dtype=torch.int32)
        total = torch.empty(1, device=x.device, dtype=torch.int32)
        g1 = gumbels[0].float().contiguous() if self.training else None
        g2 = gumbels[1].float().contiguous() if self.training else None
        check(lib().dyt_gate_compact(ptr(u), ptr(self.mlp_head.weight.detach().reshape(-1).contiguous()),
                                     ptr(self.mlp_head.bias.detach().contiguous()), ptr(g1), ptr(g2), B,
                                     1 if self.training else 0, float(self.tau), float(self.threshold), ptr(mask),
                                     ptr(logits), ptr(keep), ptr(counts), ptr(total), stream_ptr()))
        self.last_keep_index = keep[: int(total.item())]  # flat kept rows, == nonzero() of model_speed_test.py:300
        sel = torch.cat([mask.new_ones(B, 1, 1), mask.unsqueeze(-1)], dim=1)
        return sel, logits.unsqueeze(-1)


class Adapter(nn.Module):
    """Reference models/dynamic_adapter.py:80-140.  Inside a VisionTransformer the arithmetic runs in the fused HIP path;
    ``forward`` stand-alone calls the same adapter kernels through the C ABI (forward only)."""

    def __init__(self, config=None, d_model=None, bottleneck=None, dropout=0.0, init_option="bert",
                 adapter_scalar="1.0", adapter_layernorm_option="in"):
        super().__init__()
        self.n_embd = config.d_model if d_model is None else d_model
        self.down_size = config.attn_bn if bottleneck is None else bottleneck
        self.adapter_layernorm_option = adapter_layernorm_option
        if adapter_layernorm_option not in ("none", "in", "out"):
            raise ValueError("adapter_layernorm_option=%r" % (adapter_layernorm_option,))
        # reference :95-98: nn.LayerNorm(n_embd) (default eps 1e-5, trainable: an "adaptmlp." tensor) for "in" (applied to the adapter's input,
        # :121-122) and "out" (applied to its scaled output, :132-133); the shipped scripts pass "none".  Inside a VisionTransformer it runs in
        # the HIP path (dyt_config.adapter_ln, round 6)
        self.adapter_layer_norm_before = None
        if adapter_layernorm_option in ("in", "out"):
            self.adapter_layer_norm_before = _AdapterLayerNormParams(self.n_embd)
        if adapter_scalar == "learnable_scalar":   # reference :101-102: trainable (the freeze rule keeps every "adaptmlp." tensor), DYT_OPT_LEARNABLE_SCALE
            self.scale = nn.Parameter(torch.ones(1))
        else:
            self.scale = float(adapter_scalar)
        self.down_proj = _LinearParams(self.n_embd, self.down_size)
        self.up_proj = _LinearParams(self.down_size, self.n_embd)
        self.dropout = dropout

    @property
    def adapter_ln_code(self):
        """dyt_config.adapter_ln: 0 "none", 1 "in", 2 "out"."""
        return {"none": 0, "in": 1, "out": 2}[self.adapter_layernorm_option]

    def _init_weights(self):
        with torch.no_grad():  # reference :112-117
            nn.init.kaiming_uniform_(self.down_proj.weight, a=math.sqrt(5))
            nn.init.zeros_(self.up_proj.weight)
            nn.init.zeros_(self.down_proj.bias)
            nn.init.zeros_(self.up_proj.bias)

    def forward(self, x, add_residual=True, residual=None, keep_mask=None, seed=0, precision=1):
        """Reference :120-140, stand-alone: ``up(dropout(relu(down(x)))) * scale`` (+ residual), through the product's adapter
        kernels (C ABI dyt_adapter_fwd; bf16 MFMA by default, ``precision=0`` = the exact-fp32 kernels).  Forward only -- inside
        a VisionTransformer the adapter runs in the fused path with its backward; ``keep_mask`` [rows, r] (uint8) injects the
        dropout draw, otherwise Philox(``seed``) in training mode."""
        if not x.is_cuda:
            raise DyTError("Adapter runs on the HIP device only")
        if self.adapter_layer_norm_before is not None:
            raise NotImplementedError("the stand-alone Adapter.forward runs the 'none' option; 'in' / 'out' run inside a VisionTransformer "
                                      "(dyt_config.adapter_ln)")
        shape = x.shape
        xf = x.detach().float().reshape(-1, self.n_embd).contiguous()
        res = None
        if add_residual:
            res = (x if residual is None else residual).detach().float().reshape(-1, self.n_embd).contiguous()
        out = torch.empty_like(xf)
        drop_p = float(self.dropout) if self.training else 0.0
        km = None if keep_mask is None else keep_mask.to(torch.uint8).reshape(xf.shape[0], self.down_size).contiguous()
        w = [t.detach().float().contiguous() for t in (self.down_proj.weight, self.down_proj.bias, self.up_proj.weight, self.up_proj.bias)]
        check(lib().dyt_adapter_fwd(ptr(xf), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(w[3]),
                                    ptr(res), ptr(out), xf.shape[0], self.down_size, float(self.scale), drop_p, ptr(km),
                                    ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), int(precision), stream_ptr()))
        return out.reshape(shape)
