"""Host-side owner of one libdyt_hip context: weights in, activations out.

Thin plumbing above the C ABI (include/dyt_hip.h): it owns the flat trainable buffer the
library reads (so that ONE AdamW launch and ONE all-reduce cover the 74 trainable tensors of
main_image.py:250-256) and forwards every call to the shared library on torch's current
stream.  No arithmetic happens here.
"""
import ctypes

import torch

from _lib import (Config, DyTError, F_ACCUM_GRAD, F_COMPLETE, F_DEVICE_SEED, F_GATE_ALWAYS, F_MASKED_DENSE, F_SAVE, F_TOKENS_IN, F_TOKENS_OUT,
                  F_TRAINING, OPT_F32_SPLIT16, OPT_LEARNABLE_SCALE, PREC_BF16, PREC_FP16, PREC_FP16X3, PREC_FP16F8, PREC_FP16X3F, PREC_FP16X3H, PREC_FP16X3Q, PREC_FP32,
                  check, is_trainable_param, key_to_param, lib, ptr, stream_ptr)

NP, NT, DIM = 196, 197, 768


def parse_precision(p):
    if p in (PREC_FP32, PREC_BF16, PREC_FP16, PREC_FP16X3, PREC_FP16X3F, PREC_FP16X3H, PREC_FP16F8, PREC_FP16X3Q):
        return p
    p = str(p).lower()
    if p in ("fp32", "float32", "exact"):
        return PREC_FP32
    if p in ("bf16", "bfloat16", "fast"):
        return PREC_BF16
    if p in ("fp16", "float16", "half"):
        return PREC_FP16
    if p in ("fp16x3", "split", "fp32-split"):
        return PREC_FP16X3
    if p in ("fp16x3f", "fp16x3-fwd", "split-fwd"):
        return PREC_FP16X3F
    if p in ("fp16x3h", "fp16x3-bwd16", "split-half"):
        return PREC_FP16X3H
    if p in ("fp16f8", "fp16+fp8"):
        return PREC_FP16F8
    if p in ("fp16x3q",):
        return PREC_FP16X3Q
    raise ValueError("precision must be 'fp32', 'fp16x3', 'fp16x3f', 'fp16x3h', 'fp16x3q', 'fp16f8', 'bf16' or 'fp16', got %r" % (p,))


class DyTEngine:
    def __init__(self, num_classes, ffn_num, adapter_scale, device, precision=PREC_BF16, max_batch=128, depth=12,
                 slots=2, adapter_dropout=0.1, tau=5.0, threshold=0.5, frames=1, adapter_ln=0):
        if torch.device(device).type != "cuda":
            raise DyTError("the DyT path runs on a HIP device only (got %s); there is no CPU path" % (device,))
        self.device = torch.device(device)
        self.precision = parse_precision(precision)
        # "fp16" = the second build of the library (IEEE-half operands) in ITS 16-bit mode
        # "fp16x3" = that library's fp32 mode with the frozen-weight GEMMs as three IEEE-half products (DYT_OPT_F32_SPLIT16)
        # "fp16x3f" = the same with the gradient products as the hi * hi term alone (forward bit-identical to "fp16x3")
        # "fp16x3h" = the same forward again, the backward pass on 16-bit operands with the fp16 mode's kernels
        # "fp16f8" = "fp16x3h" with the two correction products of every forward GEMM on the fp8 matrix cores (not bit-identical to fp16x3)
        split = {PREC_FP16X3: 1, PREC_FP16X3F: 2, PREC_FP16X3H: 3, PREC_FP16F8: 4, PREC_FP16X3Q: 5}.get(self.precision, 0)
        lib_prec = PREC_BF16 if self.precision == PREC_FP16 else (PREC_FP32 if split else self.precision)
        # adapter_scale: tuning_config.ffn_adapter_scalar as a number, or -- "learnable_scalar" -- the block's nn.Parameter: the scale is then
        # the trainable word DYT_P_AD_SCALE of every block (key blocks.i.adaptmlp.scale), DYT_OPT_LEARNABLE_SCALE
        self.learnable_scale = isinstance(adapter_scale, torch.Tensor)
        self.cfg = Config(int(num_classes), int(ffn_num), int(depth), lib_prec,
                          int(max_batch), int(slots), 1.0 if self.learnable_scale else float(adapter_scale), float(adapter_dropout), float(tau),
                          float(threshold), int(frames), int(adapter_ln))   # adapter_ln: 0 none / 1 "in" / 2 "out" (tuning_config.ffn_adapter_layernorm_option)
        self.adapter_ln = int(adapter_ln)
        self.frames = max(1, int(frames))   # > 1: video model, every batch is clips * frames images
        self.L = lib(fp16=self.precision == PREC_FP16 or split != 0)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_ctx_create(ctypes.byref(self.cfg), ctypes.byref(h)))
        self.h = h
        if split:
            with torch.cuda.device(self.device):
                self._ck(self.L.dyt_ctx_set_option(self.h, OPT_F32_SPLIT16, split))
        if self.learnable_scale:
            self._ck(self.L.dyt_ctx_set_option(self.h, OPT_LEARNABLE_SCALE, 1))
        n = ctypes.c_int64()
        self._ck(self.L.dyt_trainable_numel(self.h, ctypes.byref(n)))
        self.n_train = n.value
        self.flat = torch.zeros(self.n_train, device=self.device, dtype=torch.float32)
        if self.learnable_scale:   # the reference's init, nn.Parameter(torch.ones(1)) (dynamic_adapter.py:102): a zeroed word would mean "adapter off"
            from _lib import P_AD_SCALE
            off, num = ctypes.c_int64(), ctypes.c_int64()
            for layer in range(int(depth)):
                self._ck(self.L.dyt_trainable_offset(self.h, P_AD_SCALE, layer, ctypes.byref(off), ctypes.byref(num)))
                self.flat[off.value:off.value + num.value] = 1.0
        self.grad = torch.zeros_like(self.flat)
        self.losses = torch.zeros(8, device=self.device, dtype=torch.float32)
        self._graphs = {}          # captured hipGraphs of the step, keyed by its static arguments
        self._comm_stream = None   # side stream of the early (upper-half) gradient all-reduce
        self._rccl_comm = None     # ncclComm_t of dyt_allreduce_grads (created on first use)
        self.generation = [0] * int(slots)   # bumped by every saving forward into a slot (stale-backward detection)
        self.depth, self.num_classes, self.ffn_num = int(depth), int(num_classes), int(ffn_num)
        # power of two the gradient carries wherever the library holds it in 16 bits (IEEE-half builds: 2^12, dyt_ctx::gs); None: no scaling
        self.grad_scale_log2 = 12 if self.precision in (PREC_FP16, PREC_FP16X3H, PREC_FP16F8, PREC_FP16X3Q) else None

    def _ck(self, rc):
        check(rc, self.L)

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
        self._ck(self.L.dyt_ctx_bytes(self.h, ctypes.byref(b)))
        return b.value

    # ---- parameters -------------------------------------------------------------------------
    def trainable_slice(self, name):
        pid, layer = key_to_param(name)
        off, num = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.L.dyt_trainable_offset(self.h, pid, layer, ctypes.byref(off), ctypes.byref(num)))
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
                self._ck(self.L.dyt_set_frozen(self.h, pid, layer, ptr(t), stream_ptr()))
                torch.cuda.current_stream().synchronize()  # `t` may be a temporary

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self.set_param(k, v)

    def _dp_check(self, slots, batch):
        """Injected stochastic-depth factors are [2, depth, B]: a training pass of another batch size would read past / short of them."""
        for sl in slots:
            n = getattr(self, "_dp_batch", {}).get(sl)
            if n is not None and n != batch:
                raise DyTError("injected drop-path factors of slot %d are for %d images, this pass has %d" % (sl, n, batch))

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
        if training:
            self._dp_check((slot,), B)
        if save:
            self.generation[slot] += 1
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_forward(self.h, slot, ptr(images), B, flags, ptr(tr), ptr(g1), ptr(g2), ptr(keep_mask),
                                     ctypes.c_uint64(seed & (2 ** 64 - 1)), ptr(logits), ptr(ts), ptr(tl), stream_ptr()))
        return logits, ts, tl

    def forward_tokens(self, tokens, training=False, complete_model=False, masked_dense=False, g1=None, g2=None, keep_mask=None, seed=0):
        """The block stack alone on a residual-stream tensor [B,197,768] (no patch embedding, no final norm / head): what a bare
        reference Block computes when it is called on tokens (vision_transformer_IN21K.py:144-165, block_flops_dict.py:36-46).
        Forward only.  Returns (tokens_out [B,197,768], token_select [B,depth,196], token_logits [B,depth,196])."""
        B = tokens.shape[0]
        flags = ((F_TRAINING if training else 0) | (F_COMPLETE if complete_model else 0) | (F_MASKED_DENSE if masked_dense else 0) |
                 F_GATE_ALWAYS | F_TOKENS_IN | F_TOKENS_OUT)
        out = torch.empty(B, NT, DIM, device=self.device, dtype=torch.float32)
        ts = torch.zeros(B, self.depth, NP, device=self.device, dtype=torch.float32)
        tl = torch.zeros(B, self.depth, NP, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_forward(self.h, 0, ptr(tokens), B, flags, ptr(self.flat), ptr(g1), ptr(g2), ptr(keep_mask),
                                     ctypes.c_uint64(seed & (2 ** 64 - 1)), ptr(out), ptr(ts), ptr(tl), stream_ptr()))
        return out, ts, tl

    def forward_features_tokens(self, images, training=False, complete_model=False, masked_dense=False, g1=None, g2=None, keep_mask=None, seed=0):
        """Images in, the block stack's output tokens out (DYT_F_TOKENS_OUT without DYT_F_TOKENS_IN): what the reference's forward_features
        holds right before its final norm (vision_transformer_IN21K.py:343-368).  Forward only.  Returns (tokens [B,197,768],
        token_select [B,depth,196], token_logits [B,depth,196])."""
        B = images.shape[0]
        flags = ((F_TRAINING if training else 0) | (F_COMPLETE if complete_model else 0) | (F_MASKED_DENSE if masked_dense else 0) |
                 F_GATE_ALWAYS | F_TOKENS_OUT)
        if training:
            self._dp_check((0,), B)
        out = torch.empty(B, NT, DIM, device=self.device, dtype=torch.float32)
        ts = torch.zeros(B, self.depth, NP, device=self.device, dtype=torch.float32)
        tl = torch.zeros(B, self.depth, NP, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_forward(self.h, 0, ptr(images), B, flags, ptr(self.flat), ptr(g1), ptr(g2), ptr(keep_mask),
                                     ctypes.c_uint64(seed & (2 ** 64 - 1)), ptr(out), ptr(ts), ptr(tl), stream_ptr()))
        return out, ts, tl

    def backward(self, slot, dlogits, grad, dtoken_select=None, dtok=None, dtoken_logits=None):
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_backward(self.h, slot, ptr(dlogits), ptr(dtoken_select), ptr(dtok), ptr(dtoken_logits),
                                      ptr(grad), stream_ptr()))

    def loss(self, logits_s, logits_t, targets, target_ratio, loss_ratio=2.0, token_minimal=0.0,
             token_minimal_weight=0.0, slot_student=0):
        B = logits_s.shape[0]
        dls, dlt = torch.empty_like(logits_s), torch.empty_like(logits_t)
        losses = torch.empty(8, device=self.device, dtype=torch.float32)
        dtok = torch.empty(3, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_loss(self.h, slot_student, ptr(logits_s), ptr(logits_t), ptr(targets), B, target_ratio,
                                  loss_ratio, token_minimal, token_minimal_weight, ptr(dls), ptr(dlt), ptr(losses),
                                  ptr(dtok), stream_ptr()))
        return dls, dlt, losses, dtok

    def step_fwd_bwd(self, images, targets, target_ratio=0.5, loss_ratio=2.0, token_minimal=0.0,
                     token_minimal_weight=0.0, masked_dense=False, g1=None, g2=None, keep_mask=None, seed=0,
                     logits_s=None, logits_t=None, token_select=None, losses=None, accumulate=False, device_seed=False):
        """engine_finetune.py:47-76 up to (not including) the optimizer step; gradients land in self.grad
        (accumulate=True: are added to it -- gradient accumulation over micro-batches).  device_seed: the noise seed
        comes from the library's device-side word (seed_device()), advanced once per step."""
        B = images.shape[0]
        self._dp_check((0, 1), B)
        for i in range(len(self.generation)):
            self.generation[i] += 1
        flags = (F_MASKED_DENSE if masked_dense else 0) | (F_ACCUM_GRAD if accumulate else 0) | (F_DEVICE_SEED if device_seed else 0)
        out = self.losses if losses is None else losses
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_step_fwd_bwd(self.h, ptr(images), ptr(targets), B, flags, ptr(self.flat), ptr(g1), ptr(g2),
                                          ptr(keep_mask), ctypes.c_uint64(seed & (2 ** 64 - 1)), target_ratio, loss_ratio,
                                          token_minimal, token_minimal_weight, ptr(self.grad), ptr(out), ptr(logits_s),
                                          ptr(logits_t), ptr(token_select), stream_ptr()))
        return out

    def seed_device(self, seed):
        """Set the device-side seed word that `device_seed=True` steps read (and advance)."""
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_seed(self.h, ctypes.c_uint64(seed & (2 ** 64 - 1)), stream_ptr()))

    def step_graph(self, images, targets, target_ratio=0.5, loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0,
                   masked_dense=False, losses=None, accumulate=False, seed=0):
        """The same step replayed from a captured hipGraph (one graph launch instead of ~450 kernel launches: the host
        cost per step drops from ~13 ms to well under 1 ms, which is what keeps 8 ranks on a 16-core host from
        becoming launch-bound).  Inputs are copied into static device buffers; the Philox seed lives on the device
        (DYT_F_DEVICE_SEED) and is set from `seed` before every replay, exactly the value an eager step with the same `seed` uses."""
        key = (tuple(images.shape), tuple(targets.shape), float(target_ratio), float(loss_ratio), float(token_minimal),
               float(token_minimal_weight), bool(masked_dense), bool(accumulate))
        ent = self._graphs.get(key)
        cur = torch.cuda.current_stream(self.device)
        if ent is None:
            x = torch.empty_like(images)
            y = torch.empty_like(targets)
            out = torch.zeros(8, device=self.device, dtype=torch.float32)
            x.copy_(images)
            y.copy_(targets)
            self.seed_device(seed)
            kw = dict(target_ratio=target_ratio, loss_ratio=loss_ratio, token_minimal=token_minimal,
                      token_minimal_weight=token_minimal_weight, masked_dense=masked_dense, losses=out, accumulate=accumulate,
                      device_seed=True)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):     # eager warm-up: creates the library's internal streams / events, sets kernel attributes
                saved = self.grad.clone() if accumulate else None
                self.step_fwd_bwd(x, y, **kw)
                if saved is not None:
                    self.grad.copy_(saved)
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            self.seed_device(seed)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # a DataLoader pin-memory thread may call HIP meanwhile
                self.step_fwd_bwd(x, y, **kw)
            ent = (g, x, y, out)
            self._graphs[key] = ent
        g, x, y, out = ent
        # the caller's per-step seed (engine_finetune.step_seed(epoch, it)) goes into the device-side seed word before EVERY replay
        # (one tiny launch outside the graph): resumed or reordered runs draw the same noise as eager runs with the same seeds
        self.seed_device(seed)
        if images.data_ptr() != x.data_ptr():
            x.copy_(images, non_blocking=True)
        if targets.data_ptr() != y.data_ptr():
            y.copy_(targets, non_blocking=True)
        for i in range(len(self.generation)):
            self.generation[i] += 1
        g.replay()
        if losses is not None:
            losses.copy_(out, non_blocking=True)
            return losses
        self.losses.copy_(out, non_blocking=True)
        return self.losses

    def adamw(self, exp_avg, exp_avg_sq, step, lr, weight_decay=0.01, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """torch.optim.AdamW semantics over the flat buffer (main_image.py:285); the moments belong to the optimizer
        object (engine_finetune.FusedAdamW) and survive a re-created engine; `step` is the 1-based update count."""
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_adamw(ptr(self.flat), ptr(self.grad), ptr(exp_avg), ptr(exp_avg_sq), self.n_train,
                                   int(step), lr, beta1, beta2, eps, weight_decay, grad_scale, stream_ptr()))

    def adamw_guarded(self, exp_avg, exp_avg_sq, state, lr, weight_decay=0.01, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """The same update, skipped (and counted in `state`, a device int32[4]) when the gradient holds inf / NaN: GradScaler.step
        semantics (misc.py:256-272) without a host sync; the step count of the bias corrections lives in state[0]."""
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_adamw_guarded(ptr(self.flat), ptr(self.grad), ptr(exp_avg), ptr(exp_avg_sq), self.n_train, ptr(state),
                                           lr, beta1, beta2, eps, weight_decay, grad_scale, stream_ptr()))

    def clip_grad_norm(self, max_norm, pre_scale=1.0, norm_out=None):
        """torch.nn.utils.clip_grad_norm_ on the flat gradient (misc.py:262-266); pre_scale = the factor AdamW applies."""
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_clip_grad_norm(self.h, ptr(self.grad), self.n_train, float(max_norm), float(pre_scale),
                                            ptr(norm_out), stream_ptr()))

    def grad_part(self, part):
        off, num = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.L.dyt_grad_part(self.h, int(part), ctypes.byref(off), ctypes.byref(num)))
        return off.value, num.value

    def streams_concurrent(self, a, b, spin_cycles=400_000, repeats=3):
        """True when work queued on streams `a` and `b` really runs at the same time.  HIP multiplexes a process's streams onto
        GPU_MAX_HW_QUEUES hardware queues (default 4; _lib.py / bench.py ask for 8 when they are imported before the runtime starts) and two
        streams on one queue run one after the other -- a DP rank's all-reduce stream sharing the step's queue costs 30.6 instead of 25.9
        ms per step.  Measured, not assumed: one ~0.2 ms spin kernel on each stream; together they take ~1x (separate
        queues) or ~2x (same queue) the time of one.  Best of `repeats` trials, so that a busy or shared GPU (another process's kernel
        landing between the two spins) does not read as "serialised" (ADVICE round 5).  Synchronises the device: must not be reached for
        the first time under stream capture (comm_stream() is called by the first eager all-reduce)."""
        best = None
        with torch.cuda.device(self.device):
            torch.cuda._sleep(1000)                     # warm the spin kernel up
            for _ in range(max(1, int(repeats))):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
                torch.cuda.synchronize(self.device)
                cur = torch.cuda.current_stream(self.device)
                ev[0].record(cur)
                a.wait_event(ev[0]); b.wait_event(ev[0])
                with torch.cuda.stream(a):
                    ev[1].record(a); torch.cuda._sleep(spin_cycles); ev[2].record(a)
                with torch.cuda.stream(b):
                    torch.cuda._sleep(spin_cycles); ev[3].record(b)
                torch.cuda.synchronize(self.device)
                both = max(ev[1].elapsed_time(ev[2]), ev[1].elapsed_time(ev[3]))
                with torch.cuda.stream(a):
                    ev[4].record(a); torch.cuda._sleep(spin_cycles); ev[5].record(a)
                torch.cuda.synchronize(self.device)
                alone = ev[4].elapsed_time(ev[5])
                if best is None or both / alone < best[1] / best[0]:
                    best = (alone, both)
                if both < 1.5 * alone:
                    break
        self._last_concurrency = (best[0], best[0], best[1])
        return best[1] < 1.5 * best[0]

    def comm_stream(self):
        """Side stream of the early gradient all-reduce: one that is VERIFIED to run beside the step's stream (up to eight candidates; streams
        map to hardware queues in creation order).  If none does -- a launcher imported torch before GPU_MAX_HW_QUEUES could be raised and
        the queues are exhausted -- a warning names the cause; DYT_STRICT=1 turns it into an error."""
        if self._comm_stream is None:
            import os
            import warnings
            cur = torch.cuda.current_stream(self.device)
            tried = []
            for _ in range(8):
                st = torch.cuda.Stream(self.device)
                tried.append(st)
                if self.streams_concurrent(cur, st):
                    self._comm_stream = st
                    break
            if self._comm_stream is None:
                msg = ("the gradient all-reduce stream shares a hardware queue with the step's stream on %s (GPU_MAX_HW_QUEUES=%s): the "
                       "all-reduce will run after the backward pass instead of under it; export GPU_MAX_HW_QUEUES=8 before the HIP runtime "
                       "starts" % (self.device, os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)")))
                if os.environ.get("DYT_STRICT", "0") == "1":
                    raise DyTError(msg)
                warnings.warn(msg)
                self._comm_stream = tried[0]
            self.comm_stream_verified = self._comm_stream is not tried[0] or len(tried) == 1 or self.streams_concurrent(cur, self._comm_stream)
        return self._comm_stream

    def find_independent_stream(self, run_step, tries=8, slack=1.2):
        """A stream whose queued work does NOT hold up any stream the step uses -- for the host->device copies of the next batch
        (engine_finetune.DevicePrefetcher).  HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, and work on two
        streams of one queue runs in order: a copy stream that lands on the queue of the step's second pass stream serialises the two
        passes (measured: 29.6 instead of 23.7 ms per step).  The library's internal streams are not visible from here, so the test is
        end to end: `run_step()` (one side-effect-free dyt_step_fwd_bwd) is timed with a ~2.5-step spin kernel parked on the candidate;
        a candidate on any of the step's queues makes the step wait for the spin.  Returns (stream or None, report)."""
        dev = self.device
        cur = torch.cuda.current_stream(dev)

        def timed(spin_stream, cycles=0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            if spin_stream is not None:
                with torch.cuda.stream(spin_stream):
                    torch.cuda._sleep(cycles)
            e0.record(cur)
            run_step()
            e1.record(cur)
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1)
        with torch.cuda.device(dev):
            run_step()                                   # warm-up: kernel attributes, internal streams / events
            base = min(timed(None), timed(None))
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # cycles per ms of the spin kernel's counter
            c0.record(cur); torch.cuda._sleep(2_000_000); c1.record(cur); torch.cuda.synchronize(dev)
            per_ms = 2_000_000 / max(c0.elapsed_time(c1), 1e-3)
            cycles = int(2.5 * base * per_ms)
            report = dict(base_ms=round(base, 3), tried=[])
            for _ in range(tries):
                st = torch.cuda.Stream(dev)
                t = timed(st, cycles)
                report["tried"].append(round(t, 3))
                if t < slack * base:
                    report["picked"] = len(report["tried"]) - 1
                    return st, report
        return None, report

    def allreduce_native(self, overlap=True):
        """dyt_allreduce_grads: SUM of the flat gradient over the ranks on the library's own RCCL communicator (created on
        first use over the default torch.distributed group); the upper part on the communication stream when `overlap`."""
        if self._rccl_comm is None:
            from _lib import rccl_comm_shared
            self._rccl_comm = rccl_comm_shared(self.device)   # one communicator per process and device, shared by every engine
        cs = ctypes.c_void_p(self.comm_stream().cuda_stream) if overlap else None
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_allreduce_grads(self.h, self._rccl_comm, ptr(self.grad), cs, stream_ptr()))

    def stream_wait_grads(self, stream, part=0):
        """Make `stream` wait on the device until the early part of the last step's gradient is final."""
        self._ck(self.L.dyt_stream_wait_grads(self.h, int(part), ctypes.c_void_p(stream.cuda_stream)))

    def debug_dispatch(self, slot, layer, batch):
        """Token-dispatcher index arrays of the pass held in `slot` (test hook): row_src, dst_of, counts, total."""
        M = batch * NT
        row_src = torch.full((M,), -7, device=self.device, dtype=torch.int32)
        dst_of = torch.full((M,), -7, device=self.device, dtype=torch.int32)
        counts = torch.zeros(batch, device=self.device, dtype=torch.int32)
        total = torch.zeros(1, device=self.device, dtype=torch.int32)
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_debug_dispatch(self.h, int(slot), int(layer), ptr(row_src), ptr(dst_of), ptr(counts), ptr(total),
                                            stream_ptr()))
        return row_src, dst_of, counts, total

    def debug_dact(self, slot, layer):
        """Saved adapter bottleneck of one block of a saved pass as fp32 [rows, 64] on the CPU (rows = B*197, or B for the cls-only last block)."""
        out = torch.empty(int(self.cfg.max_batch) * 197 * 64, device=self.device, dtype=torch.float32)
        rows = ctypes.c_int(0)
        self._ck(self.L.dyt_debug_dact(self.h, int(slot), int(layer), ptr(out), ctypes.byref(rows), stream_ptr()))
        torch.cuda.synchronize()
        return out[:rows.value * 64].reshape(rows.value, 64).cpu()

    def set_grad_scale_log2(self, k):
        """DYT_OPT_GRAD_SCALE_LOG2: the fixed loss scale of the 16-bit gradient operands (never visible in a returned gradient)."""
        from _lib import OPT_GRAD_SCALE_LOG2
        self.set_option(OPT_GRAD_SCALE_LOG2, int(k))
        self.grad_scale_log2 = int(k)
        self._graphs = {}   # captured steps carry the old factor in their kernel arguments

    def set_option(self, option, value):
        """_lib.OPT_STREAM_OVERLAP / OPT_CLS_TAIL / OPT_SHARE_BLOCK0 (scheduling only; results do not change);
        OPT_COUNT_FLOPS_TOKENS takes the token count n (0 = off) of Block.forward_count_flops."""
        self._ck(self.L.dyt_ctx_set_option(self.h, int(option), int(value)))

    def set_drop_path(self, rate):
        """Stochastic depth of the training passes (timm DropPath as the reference's blocks use it, models/vision_transformer_IN21K.py:121,131,
        285: block l drops a sample's attention / MLP branch with probability rate * l / (depth - 1)); 0 = off, the reference scripts' default."""
        self._ck(self.L.dyt_set_drop_path(self.h, ctypes.c_float(float(rate))))
        self.drop_path_rate = float(rate)
        self._graphs = {}

    def debug_drop_path(self, slot, batch):
        """The stochastic-depth factors [2, depth, batch] the last pass of `slot` ran with (tests)."""
        out = torch.empty(2, self.depth, batch, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self.L.dyt_debug_drop_path(self.h, int(slot), ptr(out), stream_ptr()))
        return out

    def set_drop_path_scales(self, slot, scales):
        """Injected branch factors [2, depth, B] (0 or 1 / keep; [0] attention branch, [1] MLP branch) for the next training passes of `slot`
        (0 = student, 1 = complete_model pass of step_fwd_bwd) instead of the library's own draws; None restores them.  The tensor is kept alive here."""
        if not hasattr(self, "_dp_keep"):
            self._dp_keep = {}
        if scales is not None:
            assert scales.is_cuda and scales.dtype == torch.float32 and scales.is_contiguous() and scales.dim() == 3 and scales.shape[:2] == (2, self.depth)
            if scales.shape[2] > int(self.cfg.max_batch):
                raise DyTError("drop-path factors for %d images, the context holds at most %d" % (scales.shape[2], int(self.cfg.max_batch)))
        self._dp_keep[slot] = scales
        if not hasattr(self, "_dp_batch"):
            self._dp_batch = {}
        self._dp_batch[slot] = None if scales is None else int(scales.shape[2])   # the passes that read them must have exactly this batch (_dp_check)
        self._graphs = {}   # a captured step carries the old pointer in its kernel arguments
        self._ck(self.L.dyt_set_drop_path_scales(self.h, int(slot), ptr(scales)))

    def set_soft_targets(self, targets):
        """Class-probability targets [rows, num_classes] (what a ``mixup_fn`` returns, reference engine_finetune.py:44-45) for the loss evaluations
        that follow, instead of their integer labels; None restores the labels.  The tensor is kept alive here; a captured step carries its pointer."""
        if targets is not None:
            if not (targets.is_cuda and targets.dtype == torch.float32 and targets.is_contiguous() and targets.dim() == 2 and targets.shape[1] == self.num_classes):
                raise DyTError("soft targets: a contiguous fp32 [rows, %d] tensor on the HIP device" % self.num_classes)
        if targets is None and getattr(self, "_soft_keep", None) is None:
            return
        self._soft_keep = targets
        self._graphs = {}
        self._ck(self.L.dyt_set_soft_targets(self.h, ptr(targets), 0 if targets is None else int(targets.shape[0])))

    # ---- measurement ------------------------------------------------------------------------
    def profile(self, on):
        self._ck(self.L.dyt_profile_enable(self.h, 1 if on else 0))

    def profile_read(self, category):
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        self._ck(self.L.dyt_profile_read(self.h, category, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
        return ms.value, n.value, fl.value
