"""Top-k and mean-per-class accuracy (reference util/metrics.py:4-25); host-side bookkeeping on
the gathered [n,C] predictions, not part of the device hot path."""
import torch


def accuracy(output, target, topk=(1,)):
    """Percentage of rows whose label is among the k best-scored classes, one 0-dim tensor per k (the return contract of the
    reference's accuracy(), util/metrics.py:4-11, which engine_finetune.evaluate reads as acc1 / acc5)."""
    n = target.shape[0]
    kmax = min(max(topk), output.shape[1])
    ranked = output.topk(kmax, dim=1).indices                       # [n, kmax], best first (ties: lowest index, as torch.topk)
    hit_rank = (ranked == target.reshape(n, 1)).float().cumsum(1)  # hit_rank[i, j] = 1 iff the label is among the j+1 best
    return [hit_rank[:, min(k, kmax) - 1].sum() * 100.0 / n for k in topk]   # (x 100, then / n: the golden accuracy is compared to the last fp32 bit)


def mean_per_class_accuracy(pred, target, num_classes):
    """Mean over the classes of (correct top-1 predictions of the class / samples of the class), in per cent; a class without samples counts
    as 0 (the VTAB metric of the reference, util/metrics.py:14-25)."""
    top1 = pred.topk(1, dim=-1).indices.reshape(-1)
    per_class_hits = torch.bincount(target[top1 == target], minlength=num_classes)
    per_class_total = torch.bincount(target, minlength=num_classes)
    return (per_class_hits / per_class_total.clamp(min=1).float() * 100).mean(0)
