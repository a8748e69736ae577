"""``models.vision_transformer_IN21K`` of the reference on MI355X.

Same factory, constructor arguments, parameter names/shapes (checkpoint compatible) and
``forward(x, complete_model) -> (logits, {"token_select", "token_logits"})`` contract as
reference models/vision_transformer_IN21K.py:192-421 -- so main_image.py / main_vtab.py /
engine_finetune.py drop onto it -- but the arithmetic is one call into libdyt_hip.so
(hand-written HIP kernels, include/dyt_hip.h).  Sub-modules are parameter containers with the
reference's names; there is no PyTorch compute path and no CPU fallback.
"""
import os
from functools import partial

import torch
import torch.nn as nn

from _lib import DyTError, OPT_COUNT_FLOPS_TOKENS, key_to_param, is_trainable_param
from runtime import DyTEngine, parse_precision
from .dynamic_adapter import Adapter, TokenSelect, _LinearParams


def _cfg_get(cfg, key, default=None):
    """attribute-dict lookup that tolerates EasyDict (AttributeError) and plain dict subclasses (KeyError)"""
    try:
        return getattr(cfg, key)
    except (AttributeError, KeyError):
        return default


class _LayerNormParams(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _ConvParams(nn.Module):
    def __init__(self, in_chans, embed_dim, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(embed_dim, in_chans, patch, patch))
        self.bias = nn.Parameter(torch.empty(embed_dim))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        nn.init.uniform_(self.bias, -1 / (in_chans * patch * patch) ** 0.5, 1 / (in_chans * patch * patch) ** 0.5)


class PatchEmbed(nn.Module):
    """timm PatchEmbed's parameter surface (``proj.weight`` [768,3,16,16], ``proj.bias``)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = _ConvParams(in_chans, embed_dim, patch_size)


class Mlp(nn.Module):
    """timm Mlp's parameter surface (``fc1``, ``fc2``)."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = _LinearParams(in_features, hidden_features)
        self.fc2 = _LinearParams(hidden_features, in_features)


class Attention(nn.Module):
    """Reference :27-75 (parameters ``qkv``, ``proj``)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, **_):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = _LinearParams(dim, dim * 3, bias=qkv_bias)
        self.proj = _LinearParams(dim, dim)


class Block(nn.Module):
    """Reference :88-185: norm1, attn, norm2, mlp, adaptmlp, mlp_token_select (+ count_flops attrs)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, tuning_config=None, layer_id=None, select=False, **_):
        super().__init__()
        self.tuning_config = tuning_config
        self.norm1 = _LayerNormParams(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = _LayerNormParams(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.adaptmlp = Adapter(self.tuning_config, dropout=0.1, bottleneck=tuning_config.ffn_num,
                                init_option=tuning_config.ffn_adapter_init_option,
                                adapter_scalar=tuning_config.ffn_adapter_scalar,
                                adapter_layernorm_option=tuning_config.ffn_adapter_layernorm_option)
        self.mlp_token_select = TokenSelect(dim, num_sub_layer=1)
        self.count_flops = None
        self.token_select_num = None

        self.precision = os.environ.get("DYT_PRECISION", "fp16")   # arithmetic mode of a stand-alone call: the model's default (inside a model the model's precision applies)
        self._engine = None
        self._engine_state = None

    def _block_engine(self, batch, device):
        """A depth-1 libdyt_hip context holding this block's parameters (stand-alone use only)."""
        eng = self._engine
        if eng is None or eng.device != device or batch > eng.cfg.max_batch or eng.precision != parse_precision(self.precision):
            self._engine = None
            eng = DyTEngine(1, self.adaptmlp.down_size, self.adaptmlp.scale, device, precision=self.precision, max_batch=batch, depth=1,
                            slots=1, adapter_dropout=self.adaptmlp.dropout, tau=self.mlp_token_select.tau,
                            threshold=self.mlp_token_select.threshold, adapter_ln=self.adaptmlp.adapter_ln_code)
            self._engine, self._engine_state = eng, None
        state = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if state != self._engine_state:
            for n, p in self.named_parameters():
                eng.set_param("blocks.0." + n, p.data)
            self._engine_state = state
        return eng

    def forward(self, x, complete_model=False, gumbel=None, keep_mask=None, seed=0):
        """Reference :144-165 on a token tensor x [B,197,768] (how block_flops_dict.py:36-46 calls a bare Block), through the
        product's kernels in a depth-1 context: returns (x_out, dict(sub_token_select [B,197,1], token_logits [B,196,1])).
        ``count_flops`` / ``token_select_num`` select the FLOP-probe variant (:167-185).  Forward only: training a block happens
        inside the VisionTransformer's fused step.  ``gumbel=(g1, g2)`` ([B,196] each) / ``keep_mask`` ([B*197, r]) inject the
        training-mode draws."""
        if not x.is_cuda:
            raise DyTError("Block runs on the HIP device only")
        B = x.shape[0]
        if self.count_flops and self.token_select_num is None:
            # the reference asserts here too (forward_count_flops, :175): refuse instead of silently running the gated path
            raise AssertionError("Block.count_flops is set without token_select_num (block_flops_dict.get_block_flops sets both)")
        eng = self._block_engine(B, x.device)
        eng.set_option(OPT_COUNT_FLOPS_TOKENS, int(self.token_select_num) if (self.count_flops and self.token_select_num) else 0)
        g1 = g2 = None
        if self.training and gumbel is not None:
            g1, g2 = (t.to(x.device).float().reshape(1, B, 196).contiguous() for t in gumbel)
        km = None if keep_mask is None else keep_mask.to(x.device).to(torch.uint8).reshape(1, B * 197, -1).contiguous()
        out, ts, tl = eng.forward_tokens(x.detach().float().contiguous(), training=self.training, complete_model=complete_model,
                                         masked_dense=True, g1=g1, g2=g2, keep_mask=km, seed=seed)
        if self.count_flops:   # the reference's forward_count_flops returns the bare tensor (:167-185; block_flops_dict.py uses it that way)
            return out
        sel = torch.cat([ts.new_ones(B, 1, 1), ts[:, 0].unsqueeze(-1)], dim=1)
        return out, dict(sub_token_select=sel, token_logits=tl[:, 0].unsqueeze(-1))

    def __getstate__(self):   # the cached ctypes context of stand-alone calls is neither copied nor pickled (copy.deepcopy and pickle both go through here)
        d = self.__dict__.copy()
        d["_engine"] = None
        d["_engine_state"] = None
        return d



class _DyTFunction(torch.autograd.Function):
    """Autograd bridge: forward = dyt_forward (activations saved inside the library, slot 0 =
    student / 1 = complete_model pass), backward = dyt_backward -> gradients of the 74 trainables."""

    @staticmethod
    def forward(ctx, model, x, complete_model, g1, g2, keep_mask, seed, *params):
        eng = model._engine
        slot = 1 if complete_model else 0
        logits, ts, tl = eng.forward(x, slot=slot, training=model.training, complete_model=complete_model, save=True,
                                     masked_dense=(model.train_mode == "masked"), gate_always=True, g1=g1, g2=g2,
                                     keep_mask=keep_mask, seed=seed)
        ctx.model, ctx.slot, ctx.batch = model, slot, x.shape[0]
        ctx.engine, ctx.generation = eng, eng.generation[slot]
        ctx.set_materialize_grads(False)
        return logits, ts, tl

    @staticmethod
    def backward(ctx, dlogits, dts, dtl):
        model = ctx.model
        eng = model._engine
        if eng is not ctx.engine or eng.generation[ctx.slot] != ctx.generation:
            # activations are saved inside the library, ONE set per kind of pass (slot 0 student, slot 1 complete_model)
            raise DyTError("backward through a %s forward whose saved activations were overwritten by a later forward of "
                           "the same kind (or the engine was re-created): run backward before the next such forward"
                           % ("complete_model" if ctx.slot else "student"))
        gbuf = torch.zeros_like(eng.flat)
        if dlogits is None:
            dlogits = torch.zeros(ctx.batch // eng.frames, eng.num_classes, device=eng.device)
        eng.backward(ctx.slot, dlogits.float().contiguous(), gbuf,
                     dtoken_select=None if dts is None else dts.float().contiguous(),
                     dtoken_logits=None if dtl is None else dtl.float().contiguous())
        grads = tuple(eng.trainable_view(n, p.shape, gbuf) if p.requires_grad else None
                      for n, p in model._trainables)
        return (None,) * 7 + grads


class VisionTransformer(nn.Module):
    """DyT ViT-B/16 (reference :192-385)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, global_pool='token', embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True, qk_norm=False, init_values=None, class_token=True,
                 no_embed_class=False, pre_norm=False, fc_norm=None, drop_rate=0., pos_drop_rate=0., patch_drop_rate=0.,
                 proj_drop_rate=0., attn_drop_rate=0., drop_path_rate=0., weight_init='', embed_layer=None,
                 norm_layer=None, act_layer=None, block_fn=Block, mlp_layer=None, tuning_config=None, select_config=None,
                 precision=None, max_batch=None, train_mode=None):
        super().__init__()
        fixed = dict(img_size=(img_size, 224), patch_size=(patch_size, 16), in_chans=(in_chans, 3), embed_dim=(embed_dim, 768),
                     num_heads=(num_heads, 12), mlp_ratio=(mlp_ratio, 4.0), qkv_bias=(qkv_bias, True), global_pool=(global_pool, 'token'),
                     class_token=(class_token, True), qk_norm=(qk_norm, False), init_values=(init_values, None),
                     no_embed_class=(no_embed_class, False), pre_norm=(pre_norm, False))
        for k, (got, want) in fixed.items():
            if got != want:
                raise NotImplementedError("%s=%r: the HIP path implements vit_base_patch16_224_in21k (%s=%r)" % (k, got, k, want))
        if max(drop_rate, pos_drop_rate, patch_drop_rate, proj_drop_rate, attn_drop_rate) > 0:
            raise NotImplementedError("dropout rates other than the adapter's are 0 in every reference entry point")
        if not 0.0 <= float(drop_path_rate) < 1.0:
            raise ValueError("drop_path_rate=%r" % (drop_path_rate,))
        # stochastic depth (reference :285 dpr = linspace(0, drop_path_rate, depth); :121,131 DropPath(dpr[i]) on both branches of block i,
        # :148,159): per-image branch factors in the residual epilogues of the proj / fc2 GEMMs, training passes only (dyt_set_drop_path)
        self.drop_path_rate = float(drop_path_rate)
        assert tuning_config is not None and select_config is not None
        # select_config.open / keep_layers reach the reference's Block only as its `select` argument (:311), which Block.__init__ never
        # reads (:106, :138: every block gets its TokenSelect): they change nothing there, so every value is accepted here as well
        if _cfg_get(tuning_config, "ffn_option", "parallel") != "parallel":
            raise NotImplementedError("ffn_option must be 'parallel'")
        self.tuning_config = tuning_config
        self.num_classes = num_classes
        self.global_pool = global_pool
        self.num_features = self.embed_dim = embed_dim
        self.num_prefix_tokens = 1
        self.depth = depth
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, embed_dim) * .02)
        self.blocks = nn.Sequential(*[
            block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, tuning_config=tuning_config,
                     layer_id=i, select=select_config.open and i >= select_config.keep_layers) for i in range(depth)])
        self.norm = _LayerNormParams(embed_dim)
        self.head = _LinearParams(embed_dim, num_classes)
        nn.init.normal_(self.cls_token, std=1e-6)
        self.apply(self.init_weights)
        # MI355X knobs (not in the reference): arithmetic mode and training mode.
        #   train_mode "masked" (default) = the reference's training semantics: the MLP is evaluated for every token and
        #     multiplied by the straight-through mask (reference :159-162), so a DROPPED token's gate still receives the
        #     task-loss gradient <dL/dx', mlp(x)> sigma'(z)/tau and can be re-selected;
        #   train_mode "compact" = what BASELINE.json's north_star describes and bench.py measures: the student MLP runs on
        #     the kept tokens only, forward AND backward; forward values are identical, but a dropped token's gate then only
        #     sees the token-ratio loss.  Opt in explicitly.
        self.precision = parse_precision(precision if precision is not None else
                                         _cfg_get(tuning_config, "precision") or os.environ.get("DYT_PRECISION", "fp16"))
        self.train_mode = train_mode or _cfg_get(tuning_config, "dyt_train_mode") or os.environ.get("DYT_TRAIN_MODE", "masked")
        assert self.train_mode in ("compact", "masked")
        self.max_batch = max_batch
        self._engine = None
        self._sync_state = None
        self._seed_counter = 0
        self._frames = 1   # > 1 in the video subclass (video_models/video_vision_transformer_IN21K.py)

    def fold_input(self, x):
        """Image model: the batch is the input itself (the video subclass folds clips into frames here)."""
        return x

    def init_weights(self, m):  # reference :323-332
        if isinstance(m, _LinearParams):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif hasattr(m, '_init_weights'):
            m._init_weights()

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    # ---- engine plumbing ----------------------------------------------------------------------
    def _named_params(self):
        out = []
        for n, p in self.named_parameters():
            pid, _ = key_to_param(n)
            out.append((n, p, is_trainable_param(pid)))
        return out

    def engine(self, batch, device):
        """Create (or grow) the libdyt_hip context and (re)upload parameters that changed."""
        eng = self._engine
        if eng is None or eng.device != device or batch > eng.cfg.max_batch or eng.frames != (self._frames or 1):
            mb = max(batch, self.max_batch or 0)
            if self._frames and self._frames > 1:
                mb = -(-mb // self._frames) * self._frames   # whole clips
            old = self._engine
            self._engine = None
            del old
            eng = DyTEngine(self.num_classes, self.tuning_config.ffn_num, self.blocks[0].adaptmlp.scale, device,
                            precision=self.precision, max_batch=mb, depth=self.depth,
                            adapter_dropout=self.blocks[0].adaptmlp.dropout, tau=self.blocks[0].mlp_token_select.tau,
                            threshold=self.blocks[0].mlp_token_select.threshold, frames=self._frames or 1,
                            adapter_ln=self.blocks[0].adaptmlp.adapter_ln_code)
            self._engine = eng
            self._sync_state = None
        if getattr(eng, "drop_path_rate", 0.0) != self.drop_path_rate:
            eng.set_drop_path(self.drop_path_rate)
        self._sync(eng)
        return eng

    def _sync(self, eng):
        params = self._named_params()
        # frozen tensors: re-upload when any was modified in place or re-allocated
        state = tuple((p.data_ptr(), p._version) for n, p, tr in params if not tr)
        if state != self._sync_state:
            for n, p, tr in params:
                if not tr:
                    eng.set_param(n, p.data)
            self._sync_state = state
        # trainable tensors live in the engine's flat buffer; re-home any that moved
        self._trainables = []
        for n, p, tr in params:
            if not tr:
                continue
            view = eng.trainable_view(n, p.shape)
            if p.data.data_ptr() != view.data_ptr():
                view.copy_(p.data.to(eng.device, torch.float32))
                p.data = view
            self._trainables.append((n, p))

    # ---- reference API ------------------------------------------------------------------------
    def forward(self, x, complete_model=False, gumbel=None, keep_mask=None):
        """x [B,3,224,224] -> (logits [B,C], {"token_select": [B,12,196,1], "token_logits": [B,12,196,1]}).

        ``gumbel=(g1, g2)`` ([depth,B,196] each) and ``keep_mask`` ([depth,B*197,r] uint8) inject the
        training-mode random draws (parity tests); by default they come from the on-device Philox
        stream seeded from torch's seed and a per-model call counter."""
        if not x.is_cuda:
            raise DyTError("DyT VisionTransformer runs on a HIP device only (input is on %s); there is no CPU path" % x.device)
        x = self.fold_input(x.float()).contiguous()
        eng = self.engine(x.shape[0], x.device)
        # FLOP-probe variant (reference Block.forward_count_flops :167-185, set through
        # `model.apply(lambda m: setattr(m, "count_flops", True))` / `token_select_num` as block_flops_dict.get_block_flops
        # :33-55 does): every block runs its MLP on the first `token_select_num` tokens
        probe = [int(b.token_select_num or 0) if b.count_flops else 0 for b in self.blocks]
        if len(set(probe)) != 1:
            raise NotImplementedError("count_flops / token_select_num must be set on every block alike (apply(setattr) does)")
        if probe[0] != getattr(eng, "_count_flops_tokens", 0):
            from _lib import OPT_COUNT_FLOPS_TOKENS
            eng.set_option(OPT_COUNT_FLOPS_TOKENS, probe[0])
            eng._count_flops_tokens = probe[0]
        g1 = g2 = None
        if gumbel is not None:
            g1, g2 = (t.to(x.device, torch.float32).contiguous() for t in gumbel)
        if keep_mask is not None:
            keep_mask = keep_mask.to(x.device, torch.uint8).contiguous()
        self._seed_counter += 1
        seed = (torch.initial_seed() * 1000003 + self._seed_counter) & (2 ** 63 - 1)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for _, p in self._trainables)
        if need_grad:
            logits, ts, tl = _DyTFunction.apply(self, x, bool(complete_model), g1, g2, keep_mask, seed,
                                                *[p for _, p in self._trainables])
        else:
            logits, ts, tl = eng.forward(x, slot=1 if complete_model else 0, training=self.training,
                                         complete_model=bool(complete_model), save=False, gate_always=True, g1=g1, g2=g2,
                                         keep_mask=keep_mask, seed=seed)
        return logits, dict(token_select=ts.unsqueeze(-1), token_logits=tl.unsqueeze(-1))

    @torch.no_grad()
    def forward_features(self, x, complete_model=False, gumbel=None, keep_mask=None):
        """Reference :343-371: the block stack's output after the final norm, ``(tokens [B,197,768], {"token_select", "token_logits"})``.
        Runs the same HIP forward with DYT_F_TOKENS_OUT (every token of the last block is then computed: no cls-only tail) and the final
        LayerNorm through the C ABI's ``dyt_layernorm``.  Forward only -- training goes through ``forward`` / the fused step."""
        import ctypes
        from _lib import check, lib, ptr, stream_ptr
        if not x.is_cuda:
            raise DyTError("DyT VisionTransformer runs on a HIP device only (input is on %s); there is no CPU path" % x.device)
        if self._frames and self._frames > 1:
            raise NotImplementedError("forward_features of the video model: its head pools every frame's tokens inside the fused forward")
        x = self.fold_input(x.float()).contiguous()
        eng = self.engine(x.shape[0], x.device)
        g1 = g2 = None
        if gumbel is not None:
            g1, g2 = (t.to(x.device, torch.float32).contiguous() for t in gumbel)
        if keep_mask is not None:
            keep_mask = keep_mask.to(x.device, torch.uint8).contiguous()
        self._seed_counter += 1
        seed = (torch.initial_seed() * 1000003 + self._seed_counter) & (2 ** 63 - 1)
        tok, ts, tl = eng.forward_features_tokens(x, training=self.training, complete_model=bool(complete_model),
                                                  masked_dense=(self.training and self.train_mode == "masked"), g1=g1, g2=g2, keep_mask=keep_mask, seed=seed)
        out = torch.empty_like(tok)
        check(lib().dyt_layernorm(ptr(tok.view(-1, self.embed_dim)), ptr(self.norm.weight.detach().float().contiguous()),
                                  ptr(self.norm.bias.detach().float().contiguous()), ptr(out.view(-1, self.embed_dim)),
                                  tok.shape[0] * tok.shape[1], stream_ptr()))
        return out, dict(token_select=ts.unsqueeze(-1), token_logits=tl.unsqueeze(-1))

    @torch.no_grad()
    def forward_head(self, x, pre_logits: bool = False):
        """Reference :375-380: cls pooling (``global_pool='token'``; fc_norm / head_drop are identities) and the head, on the output of
        ``forward_features``.  The head GEMM runs through ``dyt_linear`` in exact fp32 (class count padded to the kernel's 128-column
        tile on the host).  Forward only."""
        from _lib import check, lib, ptr, stream_ptr
        if not x.is_cuda:
            raise DyTError("forward_head runs on a HIP device only")
        cls = x[:, 0].float().contiguous()
        if pre_logits:
            return cls
        C = self.num_classes
        Cp = -(-C // 128) * 128
        w = torch.zeros(Cp, self.embed_dim, device=x.device)
        b = torch.zeros(Cp, device=x.device)
        w[:C].copy_(self.head.weight.detach())
        b[:C].copy_(self.head.bias.detach())
        out = torch.empty(cls.shape[0], Cp, device=x.device)
        check(lib().dyt_linear(ptr(cls), ptr(w), ptr(b), ptr(out), cls.shape[0], Cp, self.embed_dim, 0, stream_ptr()))
        return out[:, :C].contiguous()


def vit_base_patch16_224_in21k(**kwargs):
    """Reference :414-421."""
    model_kwargs = dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, **kwargs)
    return VisionTransformer(**model_kwargs)
