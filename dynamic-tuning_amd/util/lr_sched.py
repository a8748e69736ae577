"""Per-iteration warm-up + half-cosine learning-rate schedule (reference util/lr_sched.py:9-21)."""
import math


def lr_value(epoch, args):
    if epoch < args.warmup_epochs:
        return args.lr * epoch / args.warmup_epochs
    return args.min_lr + (args.lr - args.min_lr) * 0.5 * \
        (1. + math.cos(math.pi * (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)))


def adjust_learning_rate(optimizer, epoch, args):
    lr = lr_value(epoch, args)
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr * param_group["lr_scale"] if "lr_scale" in param_group else lr
    return lr
