"""``video_models.video_vision_transformer_IN21K`` of the reference on MI355X.

Same factory / parameter names (``query_token``, ``attentive_blocks.norm_{q,k,v}``,
``attentive_blocks.cross_attn.{q,k,v}.weight``, ``.q_bias``, ``.v_bias``, ``.proj``) and the same
``forward(x[b,c,t,h,w], complete_model) -> (logits[b,C], {"token_select","token_logits"})`` contract as
reference video_models/video_vision_transformer_IN21K.py:27-110,282-483.  The frames of a clip are
folded into the batch ("b c t h w -> (b t) c h w", :437), run through the same DyT blocks as the image
model, and pooled by ONE query attending over the t*197 final-norm tokens of the clip (:463-483); all
of it is one call into libdyt_hip.so (dyt_config.frames = t).  No PyTorch compute path, no CPU fallback.
"""
import torch
import torch.nn as nn

from _lib import DyTError
from models.dynamic_adapter import _LinearParams
from models.vision_transformer_IN21K import VisionTransformer as _ImageViT, _LayerNormParams


class CrossAttention(nn.Module):
    """Reference :54-110 (parameter surface; q/k/v Linear without bias + separate q_bias / v_bias)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, **_):
        super().__init__()
        if not qkv_bias:
            raise NotImplementedError("the video factory builds the pooling head with qkv_bias=True (:419)")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = _LinearParams(dim, dim, bias=False)
        self.k = _LinearParams(dim, dim, bias=False)
        self.v = _LinearParams(dim, dim, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = _LinearParams(dim, dim)


class AttentiveBlock(nn.Module):
    """Reference :27-51."""

    def __init__(self, dim, num_heads, qkv_bias=False, **_):
        super().__init__()
        self.norm_q = _LayerNormParams(dim)
        self.norm_k = _LayerNormParams(dim)
        self.norm_v = _LayerNormParams(dim)
        self.cross_attn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)

    def forward(self, x_q, x_kv):
        raise DyTError("AttentiveBlock is evaluated inside VisionTransformer's fused HIP path; call the model")


class VisionTransformer(_ImageViT):
    """Reference :282-483."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.query_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))          # :407
        self.attentive_blocks = AttentiveBlock(self.embed_dim, 12, qkv_bias=True)   # :408-410
        self.attentive_blocks.apply(self.init_weights)
        self._frames = None   # fixed by the first clip tensor (the reference reads t from the input, :436)

    def fold_input(self, x):
        """[b,c,t,h,w] -> [(b t),c,h,w] frames, clip-major (reference :437)."""
        if x.dim() != 5:
            raise DyTError("the video model takes clips [b,c,t,h,w] (got shape %s)" % (tuple(x.shape),))
        b, c, t, h, w = x.shape
        if t < 2:
            raise DyTError("the video model needs t >= 2 frames per clip; use models.vision_transformer_IN21K for images")
        if self._frames is not None and t != self._frames and self._engine is not None:
            self._engine = None   # frames per clip changed: the library context is rebuilt
        self._frames = t
        return x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w).contiguous()


def vit_base_patch16_224_in21k(**kwargs):
    """Reference :511-518."""
    model_kwargs = dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, **kwargs)
    return VisionTransformer(**model_kwargs)
