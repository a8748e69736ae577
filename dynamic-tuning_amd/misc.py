"""Checkpoint helpers of the reference (misc.py:296-352) for the MI355X module mirror.

Same file format: ``{'model': state_dict, 'optimizer': ..., 'epoch': ..., 'scaler': ..., 'args': ...}``
with the reference's parameter names, so checkpoints interchange with the reference in both
directions; the optimizer entry is in torch.optim.AdamW.state_dict() layout (FusedAdamW.state_dict), so the
reference's torch.optim.AdamW resumes from our files and FusedAdamW from the reference's.
"""
import os
from pathlib import Path

import torch
import torch.distributed as dist


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def save_on_master(*args, **kwargs):
    if get_rank() == 0:
        torch.save(*args, **kwargs)


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler=None, save_force=False):
    if get_rank() == 0 and ((epoch + 1) % getattr(args, "save_freq", 1) == 0 or (epoch + 1) == args.epochs or save_force):
        output_dir = Path(args.output_dir)
        to_save = {
            'model': {k: v.detach().cpu() for k, v in model_without_ddp.state_dict().items()},
            'optimizer': _to_cpu(optimizer.state_dict()) if optimizer is not None else None,
            'epoch': epoch,
            # the reference's GradScaler state; here the gradient scale lives in the optimizer (FusedAdamW.scaler_state: scale, growth tracker, skips)
            'scaler': loss_scaler.state_dict() if loss_scaler is not None else (optimizer.scaler_state() if hasattr(optimizer, 'scaler_state') else {}),
            'args': args,
        }
        save_on_master(to_save, output_dir / ('checkpoint-%s.pth' % epoch))
        if getattr(args, "auto_remove", False):
            for ckpt in os.listdir(args.output_dir):
                if ckpt.startswith('checkpoint-') and ckpt.endswith('.pth'):
                    try:
                        e = int(ckpt[len('checkpoint-'):-len('.pth')])
                    except ValueError:
                        continue
                    if e < epoch:
                        os.remove(os.path.join(args.output_dir, ckpt))


def load_model(args, model_without_ddp, optimizer, loss_scaler=None):
    if not getattr(args, "resume", None):
        return
    checkpoint = torch.load(args.resume, map_location='cpu', weights_only=False)
    ckp = checkpoint.get('model', checkpoint.get('module', checkpoint))
    model_without_ddp.load_state_dict(ckp)
    if 'optimizer' in checkpoint and 'epoch' in checkpoint and not getattr(args, 'eval', False):
        if optimizer is not None and checkpoint['optimizer'] is not None:
            optimizer.load_state_dict(checkpoint['optimizer'])
        args.start_epoch = checkpoint['epoch'] + 1
        if loss_scaler is not None and 'scaler' in checkpoint:
            loss_scaler.load_state_dict(checkpoint['scaler'])
        elif optimizer is not None and hasattr(optimizer, 'load_scaler_state') and isinstance(checkpoint.get('scaler'), dict) \
                and 'scale_log2' in checkpoint['scaler']:
            optimizer.load_scaler_state(checkpoint['scaler'])
