"""Host-side owner of one libdyt_hip context: weights in, activations out.

Thin plumbing above the C ABI (include/dyt_hip.h): it owns the flat trainable buffer the
library reads (so that ONE AdamW launch and ONE all-reduce cover the 74 trainable tensors of
main_image.py:250-256) and forwards every call to the shared library on torch's current
stream.  No arithmetic happens here.
"""
import ctypes

import torch

from _lib import (Config, DyTError, F_ACCUM_GRAD, F_COMPLETE, F_GATE_ALWAYS, F_MASKED_DENSE, F_SAVE, F_TRAINING, PREC_BF16, PREC_FP32,
                  check, is_trainable_param, key_to_param, lib, ptr, stream_ptr)

NP, NT, DIM = 196, 197, 768


def parse_precision(p):
    if p in (PREC_FP32, PREC_BF16):
        return p
    p = str(p).lower()
    if p in ("fp32", "float32", "exact"):
        return PREC_FP32
    if p in ("bf16", "bfloat16", "fast"):
        return PREC_BF16
    raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (p,))


class DyTEngine:
    def __init__(self, num_classes, ffn_num, adapter_scale, device, precision=PREC_BF16, max_batch=128, depth=12,
                 slots=2, adapter_dropout=0.1, tau=5.0, threshold=0.5, frames=1):
        if torch.device(device).type != "cuda":
            raise DyTError("the DyT path runs on a HIP device only (got %s); there is no CPU path" % (device,))
        self.device = torch.device(device)
        self.cfg = Config(int(num_classes), int(ffn_num), int(depth), parse_precision(precision), int(max_batch),
                          int(slots), float(adapter_scale), float(adapter_dropout), float(tau), float(threshold),
                          int(frames))
        self.frames = max(1, int(frames))   # > 1: video model, every batch is clips * frames images
        self.L = lib()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(self.L.dyt_ctx_create(ctypes.byref(self.cfg), ctypes.byref(h)))
        self.h = h
        n = ctypes.c_int64()
        check(self.L.dyt_trainable_numel(self.h, ctypes.byref(n)))
        self.n_train = n.value
        self.flat = torch.zeros(self.n_train, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.opt_step = 0
        self.losses = torch.zeros(8, device=self.device, dtype=torch.float32)
        self.depth, self.num_classes, self.ffn_num = int(depth), int(num_classes), int(ffn_num)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.dyt_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def bytes(self):
        b = ctypes.c_int64()
        check(self.L.dyt_ctx_bytes(self.h, ctypes.byref(b)))
        return b.value

    # ---- parameters -------------------------------------------------------------------------
    def trainable_slice(self, name):
        pid, layer = key_to_param(name)
        off, num = ctypes.c_int64(), ctypes.c_int64()
        check(self.L.dyt_trainable_offset(self.h, pid, layer, ctypes.byref(off), ctypes.byref(num)))
        return off.value, num.value

    def trainable_view(self, name, shape, buf=None):
        off, num = self.trainable_slice(name)
        return (self.flat if buf is None else buf)[off:off + num].view(shape)

    def set_param(self, name, tensor):
        """One reference state_dict entry: frozen -> library copy, trainable -> flat buffer."""
        pid, layer = key_to_param(name)
        t = tensor.detach().to(device=self.device, dtype=torch.float32).contiguous()
        if is_trainable_param(pid):
            off, num = self.trainable_slice(name)
            assert t.numel() == num, (name, t.shape, num)
            self.flat[off:off + num].copy_(t.reshape(-1))
        else:
            with torch.cuda.device(self.device):
                check(self.L.dyt_set_frozen(self.h, pid, layer, ptr(t), stream_ptr()))
                torch.cuda.current_stream().synchronize()  # `t` may be a temporary

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self.set_param(k, v)

    # ---- passes -----------------------------------------------------------------------------
    def forward(self, images, slot=0, training=False, complete_model=False, save=False, masked_dense=False,
                gate_always=False, g1=None, g2=None, keep_mask=None, seed=0, want_tokens=True, trainable=None):
        B = images.shape[0]
        flags = ((F_TRAINING if training else 0) | (F_COMPLETE if complete_model else 0) | (F_SAVE if save else 0) |
                 (F_MASKED_DENSE if masked_dense else 0) | (F_GATE_ALWAYS if gate_always else 0))
        logits = torch.empty(B // self.frames, self.num_classes, device=self.device, dtype=torch.float32)
        has_tok = want_tokens and (not complete_model or gate_always)
        ts = torch.zeros(B, self.depth, NP, device=self.device, dtype=torch.float32) if has_tok else None
        tl = torch.zeros(B, self.depth, NP, device=self.device, dtype=torch.float32) if has_tok else None
        tr = self.flat if trainable is None else trainable
        with torch.cuda.device(self.device):
            check(self.L.dyt_forward(self.h, slot, ptr(images), B, flags, ptr(tr), ptr(g1), ptr(g2), ptr(keep_mask),
                                     ctypes.c_uint64(seed & (2 ** 64 - 1)), ptr(logits), ptr(ts), ptr(tl), stream_ptr()))
        return logits, ts, tl

    def backward(self, slot, dlogits, grad, dtoken_select=None, dtok=None, dtoken_logits=None):
        with torch.cuda.device(self.device):
            check(self.L.dyt_backward(self.h, slot, ptr(dlogits), ptr(dtoken_select), ptr(dtok), ptr(dtoken_logits),
                                      ptr(grad), stream_ptr()))

    def loss(self, logits_s, logits_t, targets, target_ratio, loss_ratio=2.0, token_minimal=0.0,
             token_minimal_weight=0.0, slot_student=0):
        B = logits_s.shape[0]
        dls, dlt = torch.empty_like(logits_s), torch.empty_like(logits_t)
        losses = torch.empty(8, device=self.device, dtype=torch.float32)
        dtok = torch.empty(3, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(self.L.dyt_loss(self.h, slot_student, ptr(logits_s), ptr(logits_t), ptr(targets), B, target_ratio,
                                  loss_ratio, token_minimal, token_minimal_weight, ptr(dls), ptr(dlt), ptr(losses),
                                  ptr(dtok), stream_ptr()))
        return dls, dlt, losses, dtok

    def step_fwd_bwd(self, images, targets, target_ratio=0.5, loss_ratio=2.0, token_minimal=0.0,
                     token_minimal_weight=0.0, masked_dense=False, g1=None, g2=None, keep_mask=None, seed=0,
                     logits_s=None, logits_t=None, token_select=None, losses=None, accumulate=False):
        """engine_finetune.py:47-76 up to (not including) the optimizer step; gradients land in self.grad
        (accumulate=True: are added to it -- gradient accumulation over micro-batches)."""
        B = images.shape[0]
        flags = (F_MASKED_DENSE if masked_dense else 0) | (F_ACCUM_GRAD if accumulate else 0)
        out = self.losses if losses is None else losses
        with torch.cuda.device(self.device):
            check(self.L.dyt_step_fwd_bwd(self.h, ptr(images), ptr(targets), B, flags, ptr(self.flat), ptr(g1), ptr(g2),
                                          ptr(keep_mask), ctypes.c_uint64(seed & (2 ** 64 - 1)), target_ratio, loss_ratio,
                                          token_minimal, token_minimal_weight, ptr(self.grad), ptr(out), ptr(logits_s),
                                          ptr(logits_t), ptr(token_select), stream_ptr()))
        return out

    def adamw(self, lr, weight_decay=0.01, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """torch.optim.AdamW semantics over the flat buffer (main_image.py:285)."""
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat)
        self.opt_step += 1
        with torch.cuda.device(self.device):
            check(self.L.dyt_adamw(ptr(self.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n_train,
                                   self.opt_step, lr, beta1, beta2, eps, weight_decay, grad_scale, stream_ptr()))

    def set_option(self, option, value):
        """_lib.OPT_STREAM_OVERLAP / OPT_CLS_TAIL (scheduling only; results do not change)."""
        check(self.L.dyt_ctx_set_option(self.h, int(option), int(bool(value))))

    # ---- measurement ------------------------------------------------------------------------
    def profile(self, on):
        check(self.L.dyt_profile_enable(self.h, 1 if on else 0))

    def profile_read(self, category):
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        check(self.L.dyt_profile_read(self.h, category, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
        return ms.value, n.value, fl.value
