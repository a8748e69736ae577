"""Per-iteration learning-rate schedule of the fine-tune loop: linear warm-up to ``args.lr``, then half a cosine period down to
``args.min_lr`` (what the reference's util/lr_sched.py:9-21 computes, in its floating-point evaluation order -- the values are
compared bit for bit by tests/test_host.py)."""
import math


def lr_value(epoch, args):
    """Learning rate at the (fractional) epoch ``epoch``."""
    warm, total = args.warmup_epochs, args.epochs
    peak, floor = args.lr, args.min_lr
    if epoch < warm:
        return peak * epoch / warm
    phase = math.pi * (epoch - warm) / (total - warm)
    return floor + (peak - floor) * 0.5 * (1. + math.cos(phase))


def adjust_learning_rate(optimizer, epoch, args):
    """Write the schedule's value into every parameter group (times the group's layer-decay factor ``lr_scale`` where one is set) and return it."""
    lr = lr_value(epoch, args)
    for group in optimizer.param_groups:
        group["lr"] = lr * group.get("lr_scale", 1.0)
    return lr
