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


class FusedAdamW:
    """Stands where ``torch.optim.AdamW([trainable params], lr, weight_decay)`` stands in
    main_image.py:285; the update itself is libdyt_hip's flat AdamW kernel.  Exposes
    ``param_groups`` so ``lr_sched.adjust_learning_rate`` works unchanged."""

    def __init__(self, model, lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8):
        self.model = getattr(model, "module", model)
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)]

    def zero_grad(self, set_to_none=True):
        pass  # dyt_step_fwd_bwd zeroes the flat gradient buffer itself

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        self.model._engine.adamw(g["lr"], g["weight_decay"], g["betas"][0], g["betas"][1], g["eps"], grad_scale)

    def state_dict(self):
        e = self.model._engine
        return dict(step=e.opt_step, exp_avg=e.exp_avg, exp_avg_sq=e.exp_avg_sq, param_groups=self.param_groups)

    def load_state_dict(self, sd):
        e = self.model._engine
        e.opt_step = sd["step"]
        e.exp_avg, e.exp_avg_sq = sd["exp_avg"], sd["exp_avg_sq"]
        self.param_groups = sd["param_groups"]


def allreduce_grads(engine, group=None):
    """Sum the flat trainable-gradient buffer over ranks (RCCL over xGMI); the 1/world factor is
    folded into the AdamW kernel.  Returns that factor."""
    if not is_dist_avail_and_initialized():
        return 1.0
    dist.all_reduce(engine.grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


def train_step(model, samples, targets, optimizer, criterion=None, losses_out=None, gumbel=None, keep_mask=None,
               seed=0, target_ratio=None, token_minimal=None, token_minimal_weight=None, accumulate=False, update=True,
               accum_iter=1):
    """One fused step (reference engine_finetune.py:47-79) on device tensors; returns the device
    tensor of loss components [loss, base, token, teacher, distillation, keep ratio, kept, 0]."""
    m = getattr(model, "module", model)
    samples = m.fold_input(samples.float()).contiguous()   # video: [b,c,t,h,w] -> [(b t),c,h,w]
    eng = m.engine(samples.shape[0], samples.device)
    tr = criterion.token_target_ratio if target_ratio is None else target_ratio
    ratio = criterion.token_loss_ratio if criterion is not None else 2.0
    tmin = (criterion.token_minimal if criterion is not None else 0.0) if token_minimal is None else token_minimal
    tw = (criterion.token_minimal_weight if criterion is not None else 0.0) if token_minimal_weight is None else token_minimal_weight
    g1 = g2 = None
    if gumbel is not None:
        g1, g2 = gumbel
    out = eng.step_fwd_bwd(samples, targets, tr, ratio, tmin, tw, masked_dense=(m.train_mode == "masked"), g1=g1, g2=g2,
                           keep_mask=keep_mask, seed=seed, losses=losses_out, accumulate=accumulate)
    if update:   # reference :66-76: `loss /= accum_iter`, optimizer step on every accum_iter-th micro-batch
        scale = allreduce_grads(eng)
        optimizer.step(grad_scale=scale / accum_iter)
    return out


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler=None, max_norm=0,
                    mixup_fn=None, log_writer=None, args=None, logger=None):
    """Reference engine_finetune.py:16-106.  ``optimizer`` must be a FusedAdamW; ``loss_scaler`` is
    unused (bf16 operands / fp32 accumulation need no loss scaling -- the reference's GradScaler,
    misc.py:252-272, only exists for its fp16 autocast)."""
    if not isinstance(optimizer, FusedAdamW):
        raise DyTError("train_one_epoch drives the fused HIP step; pass engine_finetune.FusedAdamW(model, ...)")
    if mixup_fn is not None:
        raise NotImplementedError("mixup is not used by train_IN21K.sh / train_vtab.sh / train_video.sh")
    accum_iter = max(1, int(getattr(args, "accum_iter", 1) or 1)) if args is not None else 1
    model.train(True)
    m = getattr(model, "module", model)
    print_freq = 20
    nsteps = len(data_loader)
    acc = torch.zeros(8, device=device)
    step_losses = torch.zeros(8, device=device)
    sums = {k: 0.0 for k in LOSS_KEYS}
    count, pending = 0, 0
    t0 = time.time()
    seed0 = (torch.initial_seed() + 7919 * epoch) & (2 ** 62 - 1)
    lr = optimizer.param_groups[0]["lr"]
    for it, batch in enumerate(data_loader):
        samples, targets = batch[0], batch[1]
        if it % accum_iter == 0:   # per-iteration schedule, reference :43-46
            lr = lr_sched.adjust_learning_rate(optimizer, it / nsteps + epoch, args)
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        train_step(model, samples, targets, optimizer, criterion, losses_out=step_losses, seed=seed0 + it,
                   accumulate=(it % accum_iter != 0), update=((it + 1) % accum_iter == 0), accum_iter=accum_iter)
        acc += step_losses
        pending += 1
        if (it + 1) % print_freq == 0 or it + 1 == nsteps:
            host = acc.tolist()  # the only host<->device sync of the loop
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
    for batch in data_loader:
        images = batch[0].to(device, non_blocking=True)
        target = batch[1].to(device, non_blocking=True)
        output, aux = model(images)
        token_select.append(aux["token_select"].to(torch.uint8))  # {0,1}: 4x smaller gather than the reference's fp32
        predictions.append(output)
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
        status["metric"] = acc1.item()
        status["acc5"] = acc5.item()
    elif metric == "mean_per_class_acc":
        status["metric"] = mean_per_class_accuracy(predictions, targets, args.nb_classes).item()
    status["keep_ratio"] = token_select.float().mean().item()
    if logger is not None:
        logger.info("* metric %.3f keep ratio %.4f" % (status["metric"], status["keep_ratio"]))
    return status
