"""``engine_finetune`` of the reference on MI355X (reference engine_finetune.py:16-106,205-279,429-480).

``train_one_epoch`` / ``evaluate`` keep the reference's signatures and return values.  The step
body (student forward, teacher forward, CE + 2*token-ratio + teacher CE + KL, backward, AdamW;
reference :47-79) is executed by libdyt_hip in three enqueues -- dyt_step_fwd_bwd, one RCCL
all-reduce of the flat 5 MB trainable-gradient buffer (what DDP does at main_image.py:280-282),
dyt_adamw -- with NO per-step host synchronisation: the reference's ``loss.item()``,
``torch.cuda.synchronize()`` and scalar all-reduce (:70-71,81,94) are replaced by on-device
accumulation of the five loss components, read back once per ``print_freq`` steps.
"""
import math
import os
import time

import torch
import torch.distributed as dist

import util.lr_sched as lr_sched
from _lib import DyTError
from util.metrics import accuracy, mean_per_class_accuracy

LOSS_KEYS = ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss")


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def _trainable_names(model):
    """The order of main_image.py:285's `[p for name, p in model.named_parameters() if p.requires_grad]`."""
    from _lib import is_trainable_param, key_to_param
    return [(n, p) for n, p in model.named_parameters() if is_trainable_param(key_to_param(n)[0])]


def gradscaler_dict(scale_log2, growth_tracker, skipped, growth_interval):
    """The gradient-scale state in ``torch.cuda.amp.GradScaler.state_dict()``'s layout (what the reference stores under the
    checkpoint's 'scaler' key, misc.py:274-278,350-351, and feeds to ``GradScaler.load_state_dict``: scale, growth_factor,
    backoff_factor, growth_interval, _growth_tracker) plus two keys of our own: ``scale_log2`` (None: an arithmetic mode that
    carries no scale -- bf16 / fp32 operands) and ``skipped``."""
    return dict(scale=float(2.0 ** scale_log2) if scale_log2 is not None else 1.0, growth_factor=2.0, backoff_factor=0.5,
                growth_interval=int(growth_interval), _growth_tracker=int(growth_tracker),
                scale_log2=None if scale_log2 is None else int(scale_log2), skipped=int(skipped))


class FusedAdamW:
    """Stands where ``torch.optim.AdamW([trainable params], lr, weight_decay)`` stands in main_image.py:285; the update
    itself is libdyt_hip's flat AdamW kernel.  Exposes ``param_groups`` so ``lr_sched.adjust_learning_rate`` works
    unchanged.  The optimizer OWNS its state (step count and both moments, flat fp32 buffers in the library's trainable
    layout, allocated lazily on the engine's device), so it survives a re-created engine (larger eval batch) and can be
    restored right after construction, before any forward, as misc.load_model does.  ``state_dict()`` is in
    ``torch.optim.AdamW.state_dict()`` layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in named_parameters
    order, one param group), so the reference can resume from our checkpoints and we from its (misc.py:296-352)."""

    def __init__(self, model, lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8):
        self.model = getattr(model, "module", model)
        n = len(_trainable_names(self.model))
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                  foreach=None, capturable=False, differentiable=False, fused=None, params=list(range(n)))]
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self._pending = None        # per-parameter state loaded before an engine existed
        self._synced = None         # engine whose trainables were broadcast from rank 0
        # Overflow guard (the reference trains under fp16 autocast with a GradScaler that skips an update whose gradient holds
        # inf / NaN, misc.py:256-272): device int32[4] = {updates applied, updates skipped, flag of the last call, -}.  The update
        # kernel reads it, so a 16-bit overflow anywhere in a step never reaches the parameters or the moments, without a host sync.
        self.guard = True
        self.opt_state = None
        self._skips_seen = 0
        # GradScaler.update (misc.py:256-272; torch defaults): halve the scale when an update was skipped, double it after
        # `growth_interval` consecutive clean updates.  The scale here is the power of two the library carries on its 16-bit gradient
        # operands (DYT_OPT_GRAD_SCALE_LOG2, initially 2^12); it never shows in a returned gradient, so growth only trades underflow of the
        # smallest gradient operands against overflow of the largest: capped at 2^GROW_MAX_LOG2.
        self.growth_interval = 2000
        self.GROW_MAX_LOG2 = 16
        self._clean_run = 0            # applied updates since the last skip / the last growth
        self._applied_seen = 0
        self._scale_log2 = None        # last scale this optimizer set or loaded (persisted: a resumed run / a re-created engine starts from it)
        self._torch = None             # the driver's torch.optim.AdamW this optimizer was adopted from (as_fused)

    def zero_grad(self, set_to_none=True):
        pass  # dyt_step_fwd_bwd zeroes the flat gradient buffer itself

    # ---- state ---------------------------------------------------------------------------------
    def _state(self, eng):
        """Moments as flat buffers on the engine's device (moved / re-built when the engine changed)."""
        if self.exp_avg is None or self.exp_avg.numel() != eng.n_train:
            self.exp_avg = torch.zeros(eng.n_train, device=eng.device, dtype=torch.float32)
            self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        elif self.exp_avg.device != eng.device:
            self.exp_avg, self.exp_avg_sq = self.exp_avg.to(eng.device), self.exp_avg_sq.to(eng.device)
        if self.opt_state is None or self.opt_state.device != eng.device:
            self.opt_state = torch.tensor([self.step_count, 0, 0, 0], device=eng.device, dtype=torch.int32)
            self._skips_seen, self._applied_seen = 0, self.step_count   # the device counters start over with the new state word
            if self._scale_log2 is not None and getattr(eng, "grad_scale_log2", None) is not None and eng.grad_scale_log2 != int(self._scale_log2):
                eng.set_grad_scale_log2(int(self._scale_log2))       # a re-created engine (eval, resume) continues at the scale reached
        if self._pending is not None:
            for name, st in self._pending.items():
                off, num = eng.trainable_slice(name)
                self.exp_avg[off:off + num].copy_(st["exp_avg"].reshape(-1).to(eng.device, torch.float32))
                self.exp_avg_sq[off:off + num].copy_(st["exp_avg_sq"].reshape(-1).to(eng.device, torch.float32))
            self._pending = None
        return self.exp_avg, self.exp_avg_sq

    def sync_parameters(self, eng):
        """What DistributedDataParallel does at construction (main_image.py:280-282): every rank starts from rank 0's
        parameters.  Done once per engine, before its first step."""
        if self._synced is eng:
            return
        if is_dist_avail_and_initialized():
            dist.broadcast(eng.flat, src=0)
            m, v = self._state(eng)
            dist.broadcast(m, src=0)
            dist.broadcast(v, src=0)
        self._synced = eng

    def step(self, grad_scale=1.0, max_norm=0.0):
        g = self.param_groups[0]
        eng = self.model._engine
        if eng is None:
            raise DyTError("FusedAdamW.step() before any forward / step of the model")
        m, v = self._state(eng)
        if max_norm is not None and max_norm > 0:
            eng.clip_grad_norm(max_norm, grad_scale)
        self.step_count += 1
        if self.guard:   # a gradient with inf / NaN (NaN also after clipping by a NaN norm) leaves parameters and moments untouched
            eng.adamw_guarded(m, v, self.opt_state, g["lr"], g["weight_decay"], g["betas"][0], g["betas"][1], g["eps"], grad_scale)
        else:
            eng.adamw(m, v, self.step_count, g["lr"], g["weight_decay"], g["betas"][0], g["betas"][1], g["eps"], grad_scale)

    def applied_and_skipped(self):
        """(updates applied, updates skipped because of a non-finite gradient) -- reads the device-side counters (a host sync)."""
        if self.opt_state is None:
            return self.step_count, 0
        a, k = self.opt_state[:2].tolist()
        return int(a), int(k)

    def overflow_backoff(self, eng):
        """GradScaler.update (misc.py:256-272) at the loop's own sync points, from the device-side counters of dyt_adamw_guarded: updates
        skipped since the last call halve the library's gradient scale (once per call, like one GradScaler.update per skipped step would
        at most do between two of our sync points); `growth_interval` consecutive applied updates double it (up to 2^GROW_MAX_LOG2).
        Returns the number of new skips.  Works with hipGraph replay too: set_grad_scale_log2 drops the captured graphs, the next step
        re-captures."""
        applied, skipped = self.applied_and_skipped()
        self.step_count = applied
        new = skipped - self._skips_seen
        new_applied = applied - self._applied_seen
        self._skips_seen, self._applied_seen = skipped, applied
        k = getattr(eng, "grad_scale_log2", None)
        if k is None:
            return new
        if new > 0:
            self._clean_run = 0
            if k > 0:
                eng.set_grad_scale_log2(k - 1)
        else:
            self._clean_run += max(0, new_applied)
            if self._clean_run >= self.growth_interval and k < self.GROW_MAX_LOG2:
                eng.set_grad_scale_log2(k + 1)
                self._clean_run = 0
        self._scale_log2 = eng.grad_scale_log2
        return new

    def scaler_state(self):
        """What GradScaler.state_dict() holds, in its own layout (goes into the checkpoint's 'scaler' entry: the reference's
        ``loss_scaler.load_state_dict(checkpoint['scaler'])``, misc.py:350-351, accepts it)."""
        eng = self.model._engine
        k = getattr(eng, "grad_scale_log2", None) if eng is not None else None
        return gradscaler_dict(self._scale_log2 if k is None else k, self._clean_run, self._skips_seen, self.growth_interval)

    def load_scaler_state(self, st):
        """Our own dict (``scale_log2``; also the round-5 form with ``growth_tracker``) or a reference GradScaler dict (``scale`` a
        power of two, 65536 by default: log2 of it, capped at GROW_MAX_LOG2)."""
        if not st:
            return
        self._clean_run = int(st.get("_growth_tracker", st.get("growth_tracker", 0)))
        self.growth_interval = int(st.get("growth_interval", self.growth_interval))
        k = st.get("scale_log2", None)
        if k is None and "scale_log2" not in st and st.get("scale"):
            k = min(self.GROW_MAX_LOG2, max(0, int(round(math.log2(float(st["scale"]))))))
        self._scale_log2 = k
        eng = self.model._engine
        if eng is not None and self._scale_log2 is not None and getattr(eng, "grad_scale_log2", None) is not None:
            eng.set_grad_scale_log2(int(self._scale_log2))

    # ---- the driver's torch.optim.AdamW (main_image.py:285) as the handle onto this optimizer -----------------------------
    def adopt_torch_state(self):
        """Take over what ``optimizer.load_state_dict`` (misc.load_model, :347) put into the driver's torch optimizer."""
        topt = self._torch
        if topt is None:
            return
        pending, step = {}, 0
        for name, p in _trainable_names(self.model):
            st = topt.state.get(p)
            if not st or "exp_avg" not in st:
                continue
            if self.exp_avg is not None and st["exp_avg"].untyped_storage().data_ptr() == self.exp_avg.untyped_storage().data_ptr():
                return   # these are our own views (publish_torch_state): nothing was loaded
            pending[name] = dict(exp_avg=st["exp_avg"].detach().clone(), exp_avg_sq=st["exp_avg_sq"].detach().clone())
            step = max(step, int(float(st["step"])))
        if pending:
            self._pending, self.step_count, self.opt_state = pending, step, None
            if self.model._engine is not None:
                self._state(self.model._engine)

    def publish_torch_state(self):
        """``optimizer.state[p]`` of the driver's torch optimizer = views into the flat moment buffers + the applied-update count,
        so ``optimizer.state_dict()`` (misc.save_model, :306) is the fused optimizer's state at any time after a step."""
        topt, eng = self._torch, self.model._engine
        if topt is None or eng is None or self.exp_avg is None:
            return
        if self.guard and self.opt_state is not None:
            self.step_count = self.applied_and_skipped()[0]
        if self.step_count == 0 and self._pending is None:
            return
        m, v = self._state(eng)
        step = torch.tensor(float(self.step_count))
        for name, p in _trainable_names(self.model):
            off, num = eng.trainable_slice(name)
            topt.state[p] = dict(step=step.clone(), exp_avg=m[off:off + num].view(p.shape), exp_avg_sq=v[off:off + num].view(p.shape))

    def state_dict(self):
        names = _trainable_names(self.model)
        eng = self.model._engine
        state = {}
        if self.guard and self.opt_state is not None:
            self.step_count = self.applied_and_skipped()[0]   # skipped updates do not count (GradScaler.step does not call optimizer.step)
        if self.step_count > 0 or self._pending is not None:
            if eng is not None:
                m, v = self._state(eng)
            for i, (name, p) in enumerate(names):
                if eng is not None:
                    off, num = eng.trainable_slice(name)
                    ea, es = m[off:off + num].view(p.shape).clone(), v[off:off + num].view(p.shape).clone()
                else:
                    ea, es = self._pending[name]["exp_avg"], self._pending[name]["exp_avg_sq"]
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=ea, exp_avg_sq=es)
        # exactly torch.optim.AdamW.state_dict()'s keys (checkpoint interchange); the gradient-scale state travels in the checkpoint's 'scaler'
        # entry like the reference's GradScaler state (misc.save_model / load_model -> scaler_state / load_scaler_state)
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(names)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        """Accepts torch.optim.AdamW.state_dict() of the reference (or our own): tensors may live on any device
        (misc.load_model uses map_location='cpu'); they are applied when an engine exists."""
        names = _trainable_names(self.model)
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(names):
            raise ValueError("optimizer state has %d groups / %d params, the model has %d trainable tensors" % (
                len(groups), len(groups[0]["params"]) if groups else 0, len(names)))
        keep = {k: v for k, v in groups[0].items() if k != "params"}
        self.param_groups[0].update(keep)
        self.param_groups[0]["betas"] = tuple(self.param_groups[0]["betas"])
        pending, step = {}, 0
        for pos, pid in enumerate(groups[0]["params"]):
            st = sd["state"].get(pid)
            if st is None:
                continue
            name, p = names[pos]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state of %s has shape %s, parameter has %s" % (name, tuple(st["exp_avg"].shape), tuple(p.shape)))
            pending[name] = dict(exp_avg=st["exp_avg"].detach(), exp_avg_sq=st["exp_avg_sq"].detach())
            step = max(step, int(float(st["step"])))
        self.step_count = step
        self.opt_state = None   # re-created with the loaded step count (which also resets the skip / applied bookkeeping)
        self.load_scaler_state(sd.get("dyt_scaler"))
        self._pending = pending if pending else None
        if self._pending is not None and self.model._engine is not None:
            self._state(self.model._engine)


def as_fused(optimizer, model):
    """The optimizer ``train_one_epoch`` steps with.  A ``FusedAdamW`` is taken as it is; the ``torch.optim.AdamW`` the reference's
    drivers build (main_image.py:285, main_vtab.py:269, main_video.py:316: ONE group over ``[p for p in named_parameters() if
    p.requires_grad]``) is ADOPTED on first use: its hyper-parameter group becomes the fused optimizer's (one dict object, so
    ``lr_sched.adjust_learning_rate`` and any later edit reach the kernel), state it may hold from ``misc.load_model`` is taken
    over, and from then on its ``state`` aliases the fused moments -- no driver line changes.  Anything the flat AdamW kernel
    does not implement raises."""
    if isinstance(optimizer, FusedAdamW):
        return optimizer
    m = getattr(model, "module", model)
    fused = getattr(optimizer, "_dyt_fused", None)
    if fused is not None and fused.model is m:
        return fused
    if not isinstance(optimizer, torch.optim.AdamW):
        raise DyTError("train_one_epoch drives the fused HIP step: pass the torch.optim.AdamW of main_image.py:285 or an "
                       "engine_finetune.FusedAdamW (got %s)" % type(optimizer).__name__)
    if len(optimizer.param_groups) != 1:
        raise NotImplementedError("the fused AdamW kernel updates one parameter group (got %d)" % len(optimizer.param_groups))
    g = optimizer.param_groups[0]
    if g.get("amsgrad") or g.get("maximize"):
        raise NotImplementedError("amsgrad / maximize are not implemented by the fused AdamW kernel")
    names = _trainable_names(m)
    if len(g["params"]) != len(names) or any(a is not b for a, (_, b) in zip(g["params"], names)):
        raise NotImplementedError("the optimizer must hold exactly the reference's trainable tensors in named_parameters() order "
                                  "(adapters, gates, head: main_image.py:250-256,285); got %d tensors, the model trains %d"
                                  % (len(g["params"]), len(names)))
    fused = FusedAdamW(m, lr=g["lr"], weight_decay=g["weight_decay"], betas=g["betas"], eps=g["eps"])
    fused.param_groups = optimizer.param_groups
    fused._torch = optimizer
    fused.adopt_torch_state()
    optimizer._dyt_fused = fused
    return fused


_native_rccl_failed = False
_native_rccl_agreed = False   # the collective agreement below ran in this process (once, whatever engine asked: ADVICE round 5)


def allreduce_grads(engine, group=None, overlap=True):
    """Sum the flat trainable-gradient buffer over ranks (RCCL over xGMI); the 1/world factor is folded into the AdamW
    kernel.  Returns that factor.  With `overlap` the head + upper-blocks half of the buffer (final half-way through the
    backward pass, dyt_stream_wait_grads) is reduced on a side stream while the frozen-backbone backward of the lower
    blocks is still running -- what DDP's bucket hooks do inside loss.backward() (misc.py:258-259)."""
    if not is_dist_avail_and_initialized():
        return 1.0
    if engine.grad.is_cuda and group is None and os.environ.get("DYT_NATIVE_RCCL", "1") != "0" and hasattr(engine, "allreduce_native"):
        # the collective behind the C ABI (dyt_allreduce_grads): RCCL called by the library on a communicator of its own (one per
        # process).  If that communicator cannot be made (no librccl, several ranks on one GPU) the torch.distributed path below runs.
        global _native_rccl_failed, _native_rccl_agreed
        if not _native_rccl_failed:
            if _native_rccl_agreed and getattr(engine, "_rccl_comm", None) is None:
                # a re-created engine (larger eval / last batch on THIS rank only) takes the process's cached communicator without any
                # collective: the other ranks are not in this branch and would never join one
                import _lib
                engine._rccl_comm = _lib.rccl_comm_shared(engine.device)
            if getattr(engine, "_rccl_comm", None) is None:
                # Creating the communicator can fail on SOME ranks only (no librccl on one host, two ranks on one GPU); a rank that fell back
                # to torch.distributed on its own would then wait in a collective the others never issue.  The ranks agree twice, by a MIN
                # over the torch group: that librccl loads everywhere (before the unique-id broadcast of rccl_comm_create, which a rank
                # without the library would never join), and that ncclCommInitRank succeeded everywhere.
                import _lib

                def everywhere(ok):
                    flag = torch.tensor([1 if ok else 0], device=engine.grad.device, dtype=torch.int32)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    return int(flag.item()) == 1
                why = ""
                try:
                    _lib.rccl()
                    loaded = True
                except Exception as e:   # OSError from ctypes, DyTError
                    loaded, why = False, str(e)
                if not everywhere(loaded):
                    _native_rccl_failed = True
                else:
                    try:
                        engine._rccl_comm = _lib.rccl_comm_shared(engine.device)
                        made = True
                    except DyTError as e:
                        made, why = False, str(e)
                    if not everywhere(made):
                        _native_rccl_failed = True
                        engine._rccl_comm = None
                        _lib.rccl_comm_destroy_all()
                _native_rccl_agreed = True
                if _native_rccl_failed:
                    print("[dyt] native RCCL all-reduce off on every rank%s: using torch.distributed.all_reduce" % ((" (rank %d: %s)" % (dist.get_rank(), why)) if why else ""))
            if not _native_rccl_failed:
                engine.allreduce_native(overlap=overlap)   # the communicator exists on every rank: a failing collective is an error
                return 1.0 / dist.get_world_size()
    if overlap and engine.grad.is_cuda:
        off, num = engine.grad_part(0)
        comm, cur = engine.comm_stream(), torch.cuda.current_stream(engine.device)
        engine.stream_wait_grads(comm)
        with torch.cuda.stream(comm):
            dist.all_reduce(engine.grad[off:off + num], op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(engine.grad[:off], op=dist.ReduceOp.SUM, group=group)
        cur.wait_stream(comm)
    else:
        dist.all_reduce(engine.grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


def step_seed(epoch, it, base=None):
    """64-bit packing of (base seed, epoch, iteration): distinct for every step of a run."""
    base = torch.initial_seed() if base is None else base
    return ((base & 0xFFFF) << 47) | ((int(epoch) & 0xFFFF) << 31) | (int(it) & 0x7FFFFFFF)


def train_step(model, samples, targets, optimizer, criterion=None, losses_out=None, gumbel=None, keep_mask=None,
               seed=0, target_ratio=None, token_minimal=None, token_minimal_weight=None, accumulate=False, update=True,
               accum_iter=1, max_norm=0.0, graph=False):
    """One fused step (reference engine_finetune.py:47-79) on device tensors; returns the device
    tensor of loss components [loss, base, token, teacher, distillation, keep ratio, kept, 0].
    graph=True replays the forward+backward from a captured hipGraph (on-device noise only).
    ``targets``: integer labels [b], or class probabilities [b, num_classes] (a ``mixup_fn``'s output, reference :44-45) -- both
    CrossEntropyLoss terms then take the soft form (C ABI dyt_set_soft_targets)."""
    m = getattr(model, "module", model)
    samples = m.fold_input(samples.float()).contiguous()   # video: [b,c,t,h,w] -> [(b t),c,h,w]
    eng = m.engine(samples.shape[0], samples.device)
    # nn.CrossEntropyLoss(label_smoothing = e) is the soft form with t (1 - e) + e / C (torch applies the same map to class-probability targets)
    smooth = float(getattr(getattr(criterion, "base_criterion", None), "label_smoothing", 0.0) or 0.0)
    if smooth > 0.0 and not (targets.dim() == 2 and targets.is_floating_point()):
        targets = torch.nn.functional.one_hot(targets.long(), eng.num_classes).float()
    if targets.dim() == 2 and targets.is_floating_point():
        soft = targets.float().contiguous()
        if smooth > 0.0:
            soft = soft * (1.0 - smooth) + smooth / soft.shape[1]
        eng.set_soft_targets(soft)
        targets = soft.argmax(dim=1)   # (ignored by the loss while soft targets are set)
        graph = False                  # a captured step would carry this batch's pointer
    else:
        eng.set_soft_targets(None)
    optimizer.sync_parameters(eng)
    tr = criterion.token_target_ratio if target_ratio is None else target_ratio
    ratio = criterion.token_loss_ratio if criterion is not None else 2.0
    tmin = (criterion.token_minimal if criterion is not None else 0.0) if token_minimal is None else token_minimal
    tw = (criterion.token_minimal_weight if criterion is not None else 0.0) if token_minimal_weight is None else token_minimal_weight
    g1 = g2 = None
    if gumbel is not None:
        g1, g2 = gumbel
    masked = m.train_mode == "masked"
    use_graph = graph and gumbel is None and keep_mask is None
    if use_graph:
        out = eng.step_graph(samples, targets, tr, ratio, tmin, tw, masked_dense=masked, losses=losses_out, accumulate=accumulate,
                             seed=seed)
    else:
        out = eng.step_fwd_bwd(samples, targets, tr, ratio, tmin, tw, masked_dense=masked, g1=g1, g2=g2,
                               keep_mask=keep_mask, seed=seed, losses=losses_out, accumulate=accumulate)
    if update:   # reference :66-76: `loss /= accum_iter`, optimizer step on every accum_iter-th micro-batch
        scale = allreduce_grads(eng, overlap=not use_graph)   # events recorded inside a graph cannot be waited on from outside
        optimizer.step(grad_scale=scale / accum_iter, max_norm=max_norm)
    return out


def _check_supported(model, criterion):
    """The fused step hard-wires what every reference entry point uses; anything else must fail loudly."""
    from _lib import is_trainable_param, key_to_param
    base = getattr(criterion, "base_criterion", None)
    if base is not None and not (isinstance(base, torch.nn.CrossEntropyLoss) and 0.0 <= float(getattr(base, "label_smoothing", 0.0)) < 1.0
                                 and base.weight is None and base.reduction == "mean" and base.ignore_index == -100):
        raise NotImplementedError("the fused step computes nn.CrossEntropyLoss (main_image.py:292; mean reduction, no class weights, "
                                  "label_smoothing through the soft-target form); got %r" % (base,))
    for n, p in model.named_parameters():
        tr = is_trainable_param(key_to_param(n)[0])
        if tr != bool(p.requires_grad):
            raise NotImplementedError("%s.requires_grad=%s: the fused step trains exactly the reference's freeze rule "
                                      "(adapters, gates, head -- main_image.py:250-256); --fulltune / partial freezing is "
                                      "not supported" % (n, p.requires_grad))


class DevicePrefetcher:
    """Input feeding of the step loop (reference engine_finetune.py:34-42: ``samples.to(device, non_blocking=True)`` inside the
    loop, on the compute stream -- 77 MB per B=128 step, stream-ordered in front of the step's first kernel).

    Here the host->device copy of batch i+1 is issued on a COPY stream into the other of two device buffers while step i is
    still running, and handed over by events: the compute stream waits for ``ready[k]`` (the copy), the copy stream waits for
    ``free[k]`` (recorded on the compute stream after the step that read buffer k was enqueued) before it overwrites a buffer.
    The buffers are allocated once per shape, so the caching allocator never sees cross-stream reuse.  Sources are expected
    in pinned memory (the drivers' DataLoaders use ``pin_memory=True``, main_image.py:172); pageable sources still work, the
    runtime then stages them.  Tensors that already live on the device pass through untouched.

    The copy stream must not share a hardware queue with any stream of the step (HIP maps streams onto GPU_MAX_HW_QUEUES queues;
    on a shared queue the copy's wait for ``free[k]`` holds up the step's second pass stream: 29.6 instead of 23.7 ms per step,
    measured), so it is CHOSEN by measurement: ``pick_stream(samples, targets)`` is called once, with the first batch on the
    device, and returns a verified stream (DyTEngine.find_independent_stream) or None -- then, and for the first batch, the
    copy stays on the compute stream, the reference's placement.  Iterating yields ``(index, samples, targets, rest_of_batch)``;
    the yielded device tensors are valid until the iteration after next."""

    def __init__(self, loader, device, pick_stream=None, slots=2):
        self.loader, self.device, self.slots = loader, torch.device(device), int(slots)
        self.on = self.device.type == "cuda" and os.environ.get("DYT_PREFETCH", "1") != "0" and pick_stream is not None
        self.pick_stream = pick_stream
        self.copy_stream = None
        self.buf = [[None, None] for _ in range(self.slots)]
        self.ready = [None] * self.slots
        self.free = [None] * self.slots
        self.copied_bytes = 0

    def __len__(self):
        return len(self.loader)

    def _into(self, k, j, t):
        if not torch.is_tensor(t) or t.device == self.device:
            return t
        b = self.buf[k][j]
        if b is None or b.dtype != t.dtype or b.numel() < t.numel():
            b = self.buf[k][j] = torch.empty(max(t.numel(), 1), dtype=t.dtype, device=self.device)
        dst = b[:t.numel()].view(t.shape)
        dst.copy_(t, non_blocking=True)
        self.copied_bytes += t.numel() * t.element_size()
        return dst

    def __iter__(self):
        asked = False
        if self.on and self.copy_stream is None:
            self.copy_stream = self.pick_stream(None, None)   # a stream verified earlier (previous epoch): the first batch is prefetched too
        for it, batch in enumerate(self.loader):
            if self.copy_stream is None:   # the reference's placement: on the compute stream, in front of the step
                xs, ys = batch[0].to(self.device, non_blocking=True), batch[1].to(self.device, non_blocking=True)
                if self.on and not asked and xs is not batch[0]:
                    asked = True
                    self.copy_stream = self.pick_stream(xs, ys)
                yield it, xs, ys, tuple(batch[2:])
                continue
            k = it % self.slots
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                if self.free[k] is not None:
                    self.copy_stream.wait_event(self.free[k])     # the step that read this buffer has finished on the device
                elif self.buf[k][0] is None:
                    self.copy_stream.wait_stream(cur)             # first use: order the allocation behind what the compute stream holds
                xs, ys = self._into(k, 0, batch[0]), self._into(k, 1, batch[1])
                if self.ready[k] is None:
                    self.ready[k] = torch.cuda.Event()
                self.ready[k].record(self.copy_stream)
            cur.wait_event(self.ready[k])
            yield it, xs, ys, tuple(batch[2:])
            # resumed by the consumer's next(): its step on buffer k is enqueued on the compute stream by now
            if self.free[k] is None:
                self.free[k] = torch.cuda.Event()
            self.free[k].record(torch.cuda.current_stream(self.device))


def _copy_stream_picker(model, criterion, masked=None):
    """pick_stream of the training loop's DevicePrefetcher: the engine's verified copy stream, found once per engine by timing a
    side-effect-free step (gradients into the engine's buffer, which the next real step overwrites; no optimizer update)."""
    m = getattr(model, "module", model)

    def pick(samples, targets):
        if samples is None:   # only what is already known
            return getattr(getattr(m, "_engine", None), "_copy_stream_probe", (None, None))[0]
        x = m.fold_input(samples.float()).contiguous()
        eng = m.engine(x.shape[0], x.device)
        if not hasattr(eng, "_copy_stream_probe"):
            scratch = torch.zeros(8, device=x.device)
            tr = getattr(criterion, "token_target_ratio", 0.5)

            def run():
                eng.step_fwd_bwd(x, targets, tr, 2.0, 0.0, 0.0, masked_dense=(m.train_mode == "masked"), seed=12345, losses=scratch)
            eng._copy_stream_probe = eng.find_independent_stream(run)
            if eng._copy_stream_probe[0] is None:
                import warnings
                warnings.warn("no stream runs beside the step's own (GPU_MAX_HW_QUEUES=%s, probe %s): input batches are copied on the compute "
                              "stream" % (os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)"), eng._copy_stream_probe[1]))
        return eng._copy_stream_probe[0]
    return pick


def _verified_copy_stream(model):
    """pick_stream of the evaluation loops: the copy stream a training epoch has verified for this engine, if any (a stream that runs
    beside every stream of the training step also runs beside the forward pass's)."""
    m = getattr(model, "module", model)
    return lambda samples=None, targets=None: getattr(getattr(m, "_engine", None), "_copy_stream_probe", (None, None))[0]


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler=None, max_norm=0,
                    mixup_fn=None, log_writer=None, args=None, logger=None):
    """Reference engine_finetune.py:16-106.  ``optimizer`` is the ``torch.optim.AdamW`` the drivers build (main_image.py:285;
    adopted by a FusedAdamW on first use, ``as_fused``) or a FusedAdamW; ``loss_scaler`` (the drivers' ``NativeScaler()``,
    main_image.py:290) is bound to that optimizer's gradient-scale state -- the loss scaling itself (the reference's GradScaler,
    misc.py:252-272) is the library's power-of-two scale on 16-bit gradient operands plus the device-side overflow guard;
    ``max_norm`` (--clip_grad) clips the global gradient norm on the device before the update.  ``args.hip_graph`` replays the
    step from a hipGraph."""
    optimizer = as_fused(optimizer, model)
    if loss_scaler is not None and hasattr(loss_scaler, "bind"):
        loss_scaler.bind(optimizer)
    accum_iter = max(1, int(getattr(args, "accum_iter", 1) or 1)) if args is not None else 1
    use_graph = bool(getattr(args, "hip_graph", False)) if args is not None else False
    model.train(True)
    m = getattr(model, "module", model)
    _check_supported(m, criterion)
    print_freq = 20
    nsteps = len(data_loader)
    acc = torch.zeros(8, device=device)
    step_losses = torch.zeros(8, device=device)
    sums = {k: 0.0 for k in LOSS_KEYS}
    count, pending = 0, 0
    t0 = time.time()
    lr = optimizer.param_groups[0]["lr"]
    # batch i+1 travels host -> device on a copy stream while step i runs (DevicePrefetcher); the reference's loop copies on the compute stream
    for it, samples, targets, extra in DevicePrefetcher(data_loader, device, _copy_stream_picker(model, criterion)):
        # parity tests may append the random draws to inject: (samples, targets, (g1, g2), keep_mask)
        gumbel = tuple(t.to(device).contiguous() for t in extra[0]) if len(extra) > 0 and extra[0] is not None else None
        keep_mask = extra[1].to(device).contiguous() if len(extra) > 1 and extra[1] is not None else None
        if it % accum_iter == 0:   # per-iteration schedule, reference :43-46
            lr = lr_sched.adjust_learning_rate(optimizer, it / nsteps + epoch, args)
        if mixup_fn is not None:   # reference :44-45 (timm.data.Mixup or any callable of that shape): mixed samples, class-probability targets
            samples, targets = mixup_fn(samples, targets)
        train_step(model, samples, targets, optimizer, criterion, losses_out=step_losses, seed=step_seed(epoch, it),
                   gumbel=gumbel, keep_mask=keep_mask, accumulate=(it % accum_iter != 0), update=((it + 1) % accum_iter == 0),
                   accum_iter=accum_iter, max_norm=max_norm or 0.0, graph=use_graph)
        acc += step_losses
        pending += 1
        if (it + 1) % print_freq == 0 or it + 1 == nsteps:
            host = acc.tolist()  # the only host<->device sync of the loop
            before = getattr(m._engine, "grad_scale_log2", None)
            new_skips = optimizer.overflow_backoff(m._engine) if optimizer.guard else 0   # graph replay too: the scale change drops the captured graphs
            if logger is not None and (new_skips or getattr(m._engine, "grad_scale_log2", None) != before):
                logger.info("%d update(s) skipped (non-finite gradient: 16-bit overflow); gradient scale 2^%s -> 2^%s" % (
                    new_skips, before, getattr(m._engine, "grad_scale_log2", "?")))
            acc.zero_()
            for i, k in enumerate(LOSS_KEYS):
                sums[k] += host[i]
            count += pending
            if logger is not None:
                logger.info("Epoch: [%d] [%d/%d] lr: %.6f loss: %.4f keep: %.3f time/it: %.4f" % (
                    epoch, it + 1, nsteps, lr, host[0] / pending, host[5] / pending, (time.time() - t0) / (it + 1)))
            if log_writer is not None:
                epoch_1000x = int(((it + 1) / nsteps + epoch) * 1000)
                log_writer.add_scalar('loss', host[0] / pending, epoch_1000x)
                log_writer.add_scalar('lr', lr, epoch_1000x)
            pending = 0
    stats = {k: v / max(count, 1) for k, v in sums.items()}
    stats["lr"] = lr
    if mixup_fn is not None and getattr(m, "_engine", None) is not None:
        m._engine.set_soft_targets(None)   # evaluation / the next caller's steps use their integer labels
    optimizer.publish_torch_state()   # an adopted torch.optim.AdamW now reports this epoch's moments / step count (misc.save_model)
    if is_dist_avail_and_initialized():  # metric_logger.synchronize_between_processes(), reference :104
        t = torch.tensor([stats[k] for k in LOSS_KEYS], device=device, dtype=torch.float64)
        dist.all_reduce(t)
        for i, k in enumerate(LOSS_KEYS):
            stats[k] = float(t[i]) / dist.get_world_size()
    return stats


def train_video_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler=None, max_norm=0,
                          mixup_fn=None, log_writer=None, args=None, logger=None):
    """Reference engine_finetune.py:109-203: the video loop is the image loop with clip tensors
    [b,c,t,h,w]; the model folds the frames into the batch and pools them per clip."""
    return train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler, max_norm, mixup_fn,
                           log_writer, args, logger)


def all_gather_concat(tensor):
    """Reference engine_finetune.py:446-480: gather variable-length dim-0 tensors from all ranks."""
    world = get_world_size()
    if world == 1:
        return tensor
    n = torch.tensor([tensor.shape[0]], device=tensor.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = tensor.new_zeros((mx,) + tuple(tensor.shape[1:]))
    pad[: tensor.shape[0]] = tensor
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


@torch.no_grad()
def evaluate(data_loader, model, device, logger=None, base_flops=None, flops_dict=None, args=None):
    """Reference engine_finetune.py:208-279: eval-mode forward over the loader, gather across ranks,
    top-1 (or mean-per-class) accuracy in ``status["metric"]``."""
    model.eval()
    token_select, targets, predictions = [], [], []
    for _, images, target, _ in DevicePrefetcher(data_loader, device, _verified_copy_stream(model)):
        output, aux = model(images)
        token_select.append(aux["token_select"].to(torch.uint8))  # {0,1}: 4x smaller gather than the reference's fp32
        predictions.append(output)
        targets.append(target.clone())   # (the prefetcher's buffer is reused two batches later)
    targets = torch.cat(targets, dim=0)
    predictions = torch.cat(predictions, dim=0)
    token_select = torch.cat(token_select, dim=0)
    if is_dist_avail_and_initialized():
        targets = all_gather_concat(targets)
        predictions = all_gather_concat(predictions)
        token_select = all_gather_concat(token_select)
    status = {}
    metric = getattr(args, "metric", "accuracy")
    if metric == "accuracy":
        acc1, acc5 = accuracy(predictions, targets, topk=(1, 5))
        status["metric"] = status["acc1"] = acc1.item()   # 'acc1' is what the drivers' --eval branch prints (main_image.py:323)
        status["acc5"] = acc5.item()
    elif metric == "mean_per_class_acc":
        status["metric"] = mean_per_class_accuracy(predictions, targets, args.nb_classes).item()
    status["keep_ratio"] = token_select.float().mean().item()
    if logger is not None:
        logger.info("* metric %.3f keep ratio %.4f" % (status["metric"], status["keep_ratio"]))
    return status


@torch.no_grad()
def evaluate_video(data_loader, model, device, logger=None, base_flops=None, flops_dict=None, args=None):
    """Reference engine_finetune.py:281-356: every sample carries V views [B,V,c,t,h,w]; views are folded into the batch,
    the per-view logits averaged per sample (:302-305), then the same gather + accuracy as ``evaluate``."""
    model.eval()
    token_select, targets, predictions = [], [], []
    for _, images, target, _ in DevicePrefetcher(data_loader, device, _verified_copy_stream(model)):
        target = target.clone()
        B, V = images.shape[0], images.shape[1]
        output, aux = model(images.flatten(0, 1))
        predictions.append(output.view(B, V, -1).mean(dim=1))
        token_select.append(aux["token_select"].to(torch.uint8))
        targets.append(target)
    targets = torch.cat(targets, dim=0)
    predictions = torch.cat(predictions, dim=0)
    token_select = torch.cat(token_select, dim=0)
    if is_dist_avail_and_initialized():
        targets = all_gather_concat(targets)
        predictions = all_gather_concat(predictions)
        token_select = all_gather_concat(token_select)
    status = {}
    metric = getattr(args, "metric", "accuracy")
    if metric == "accuracy":
        acc1, acc5 = accuracy(predictions, targets, topk=(1, 5))
        status["metric"] = status["acc1"] = acc1.item()
        status["acc5"] = acc5.item()
    elif metric == "mean_per_class_acc":
        status["metric"] = mean_per_class_accuracy(predictions, targets, args.nb_classes).item()
    ts = token_select.float()
    status["keep_ratio"] = ts.mean().item()
    if flops_dict is not None and base_flops is not None:   # :341-345: analytic FLOPs from the masks
        from block_flops_dict import batch_select_flops
        fl = batch_select_flops(ts.shape[0], flops_dict=flops_dict, token_select=ts.unsqueeze(-1) if ts.dim() == 3 else ts,
                                block_num=12, base_flops=base_flops)
        status["gflops"] = float(fl.mean())
    if logger is not None:
        logger.info("* metric %.3f keep ratio %.4f" % (status["metric"], status["keep_ratio"]))
    return status
