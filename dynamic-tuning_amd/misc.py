"""``misc`` of the reference for the MI355X module mirror: everything main_image.py / main_vtab.py / main_video.py /
engine_finetune.py touch in it (reference misc.py:24-363), so that this file can stand in front of the reference's.

* process-group helpers: ``init_distributed_mode`` (env:// / OMPI / SLURM -> backend "nccl" = RCCL on ROCm, reference
  :217-249), ``get_world_size / get_rank / is_main_process / is_dist_avail_and_initialized``, ``save_on_master``,
  ``all_reduce_mean`` (:355-363), ``setup_for_distributed`` (:171-186);
* meters: ``SmoothedValue`` / ``MetricLogger`` (:24-168) -- host-side bookkeeping, same attributes and format strings;
* ``NativeScalerWithGradNormCount`` (:252-278): the reference's fp16 ``GradScaler`` wrapper.  Here the loss scale is the
  power of two libdyt_hip carries on its 16-bit gradient operands and the overflow guard is the device-side check of
  ``dyt_adamw_guarded``; this object is the driver-facing handle onto that state (``state_dict`` in ``GradScaler``'s layout,
  so the reference resumes from our checkpoints) and, called like the reference calls it, runs backward + unscale/clip +
  guarded optimizer step on the generic autograd route;
* checkpoints: ``save_model`` / ``load_model`` (:296-352), same file format
  ``{'model', 'optimizer', 'epoch', 'scaler', 'args'}`` with the reference's parameter names; the optimizer entry is in
  ``torch.optim.AdamW.state_dict()`` layout whichever optimizer object the driver holds.
"""
import builtins
import datetime
import math
import os
import time
from collections import defaultdict, deque
from pathlib import Path

import torch
import torch.distributed as dist

inf = math.inf   # the reference imports it from torch._six (:21), which recent torch no longer has


# ---- process group ------------------------------------------------------------------------------------------------------
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def all_reduce_mean(x):
    """Mean of a host scalar over the ranks (reference :355-363)."""
    world = get_world_size()
    if world == 1:
        return x
    t = torch.tensor(float(x), device="cuda" if torch.cuda.is_available() else "cpu")
    dist.all_reduce(t)
    return (t / world).item()


_plain_print = builtins.print


def setup_for_distributed(is_master):
    """print() becomes a time-stamped, rank-0-only print (``force=True`` prints everywhere), reference :171-186."""
    def stamped(*args, **kwargs):
        if kwargs.pop("force", False) or is_master:
            _plain_print("[%s] " % datetime.datetime.now().time(), end="")
            _plain_print(*args, **kwargs)
    builtins.print = stamped


def init_distributed_mode(args):
    """Reference :217-249.  Rank / world / local GPU come from OpenMPI (``args.dist_on_itp``), from the torchrun environment
    (RANK / WORLD_SIZE / LOCAL_RANK) or from SLURM; otherwise the run is single-process.  Backend "nccl" is RCCL on ROCm."""
    env = os.environ
    if getattr(args, "dist_on_itp", False):
        args.rank, args.world_size = int(env["OMPI_COMM_WORLD_RANK"]), int(env["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(env["OMPI_COMM_WORLD_LOCAL_RANK"])
        args.dist_url = "tcp://%s:%s" % (env["MASTER_ADDR"], env["MASTER_PORT"])
        env["LOCAL_RANK"], env["RANK"], env["WORLD_SIZE"] = str(args.gpu), str(args.rank), str(args.world_size)
    elif all(k in env for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")):
        args.rank, args.world_size, args.gpu = int(env["RANK"]), int(env["WORLD_SIZE"]), int(env["LOCAL_RANK"])
    elif "SLURM_PROCID" in env:
        args.rank = int(env["SLURM_PROCID"])
        args.gpu = args.rank % max(1, torch.cuda.device_count())
    else:
        print("Not using distributed mode")
        setup_for_distributed(is_master=True)
        args.distributed = False
        return
    args.distributed = True
    # one process per GPU; the step's two pass streams, the gradient-sum stream, the all-reduce stream and RCCL's own need more than
    # HIP's default four hardware queues -- effective when the HIP runtime has not started yet
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl"
    print("| distributed init (rank %d): %s, gpu %d" % (args.rank, getattr(args, "dist_url", "env://"), args.gpu), flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


# ---- meters -------------------------------------------------------------------------------------------------------------
class SmoothedValue(object):
    """Windowed + global statistics of a scalar series (reference :24-82): ``median / avg`` over the last ``window_size``
    values, ``global_avg = total / count``, ``max``, ``value``; ``str()`` fills ``fmt`` with those names."""

    def __init__(self, window_size=20, fmt=None):
        self.fmt = "{median:.4f} ({global_avg:.4f})" if fmt is None else fmt
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.deque.append(value)
        self.total += value * n
        self.count += n

    def synchronize_between_processes(self):
        """count / total summed over the ranks (the window is left local, as in the reference)."""
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    """Named ``SmoothedValue`` meters + the timed iteration wrapper ``log_every`` (reference :85-168)."""

    def __init__(self, delimiter="\t", logger=None):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter
        self.logger = logger

    def update(self, **kwargs):
        for name, v in kwargs.items():
            if v is None:
                continue
            if torch.is_tensor(v):
                v = v.item()
            assert isinstance(v, (float, int)), (name, type(v))
            self.meters[name].update(v)

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError("'%s' object has no attribute '%s'" % (type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("%s: %s" % (name, meter) for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def _emit(self, text):
        (self.logger.info if self.logger is not None else print)(text)

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        n = len(iterable)
        width = len(str(n))
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        t_start = t_last = time.time()
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - t_last)
            yield obj
            iter_time.update(time.time() - t_last)
            if i % print_freq == 0 or i == n - 1:
                eta = datetime.timedelta(seconds=int(iter_time.global_avg * (n - i)))
                fields = [header, "[%*d/%d]" % (width, i, n), "eta: %s" % eta, str(self), "time: %s" % iter_time, "data: %s" % data_time]
                if torch.cuda.is_available():
                    fields.append("max mem: %.0f" % (torch.cuda.max_memory_allocated() / 2 ** 20))
                self._emit(self.delimiter.join(fields))
            t_last = time.time()
        total = time.time() - t_start
        self._emit("%s Total time: %s (%.4f s / it)" % (header, datetime.timedelta(seconds=int(total)), total / max(n, 1)))


# ---- loss scaler ----------------------------------------------------------------------------------------------------------
def get_grad_norm_(parameters, norm_type=2.0):
    """Global gradient norm over the parameters that have a gradient (reference :281-293)."""
    if torch.is_tensor(parameters):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    norm_type = float(norm_type)
    dev = grads[0].device
    if norm_type == inf:
        return max(g.abs().max().to(dev) for g in grads)
    return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, norm_type).to(dev) for g in grads]), norm_type)


class NativeScalerWithGradNormCount:
    """Stands where the reference's ``GradScaler`` wrapper stands (misc.py:252-278, built at main_image.py:290 and handed to
    ``train_one_epoch`` / ``save_model`` / ``load_model``).

    The fp16 loss scale of this implementation is the power of two libdyt_hip multiplies its 16-bit gradient operands by
    (``DYT_OPT_GRAD_SCALE_LOG2``, 2^12 initially); the skip-on-overflow and the halve / grow policy run on device-side counters in
    ``engine_finetune.FusedAdamW``.  ``train_one_epoch`` binds this object to that optimizer, after which ``state_dict()`` reports the
    live state in ``torch.cuda.amp.GradScaler.state_dict()``'s layout (+ ``scale_log2``) and ``load_state_dict`` -- the reference's
    GradScaler dicts included -- sets it.  Called the way the reference's loop calls it (generic autograd route: the model's
    forward returned autograd-connected outputs), it runs backward, clips or measures the gradient norm and steps the optimizer
    unless the gradient is non-finite."""
    state_dict_key = "amp_scaler"

    def __init__(self):
        self._fused = None      # engine_finetune.FusedAdamW whose scale / counters this handle reports
        self._loaded = None     # state loaded before an optimizer was bound
        self.skipped = 0        # updates this object itself skipped on the autograd route

    def bind(self, fused):
        if self._fused is fused:
            return
        self._fused = fused
        if self._loaded is not None:
            fused.load_scaler_state(self._loaded)
            self._loaded = None

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        loss.backward(create_graph=create_graph)   # the library's own power-of-two scale is applied inside dyt_backward
        if not update_grad:
            return None
        params = [p for p in ([parameters] if torch.is_tensor(parameters) else list(parameters or [])) if p.grad is not None]
        if clip_grad is not None and clip_grad > 0:
            norm = torch.nn.utils.clip_grad_norm_(params, clip_grad)
        else:
            norm = get_grad_norm_(params)
        if bool(torch.isfinite(norm)):   # GradScaler.step: an update whose gradient holds inf / NaN is skipped
            optimizer.step()
        else:
            self.skipped += 1
        return norm

    def state_dict(self):
        if self._fused is not None:
            return self._fused.scaler_state()
        if self._loaded is not None:
            return dict(self._loaded)
        from engine_finetune import gradscaler_dict
        return gradscaler_dict(None, 0, 0, 2000)

    def load_state_dict(self, state_dict):
        if self._fused is not None:
            self._fused.load_scaler_state(state_dict)
        else:
            self._loaded = dict(state_dict) if state_dict else None


# ---- checkpoints ----------------------------------------------------------------------------------------------------------
def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu().clone()   # moments may be views into one flat buffer: save them as tensors of their own
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def _scaler_entry(optimizer, loss_scaler):
    if loss_scaler is not None:
        return loss_scaler.state_dict()
    fused = getattr(optimizer, "_dyt_fused", optimizer)
    return fused.scaler_state() if hasattr(fused, "scaler_state") else {}


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler=None, save_force=False):
    """Reference :296-331: ``checkpoint-<epoch>.pth`` on rank 0 every ``save_freq`` epochs, at the last epoch or when forced;
    ``auto_remove`` deletes older ones."""
    due = (epoch + 1) % getattr(args, "save_freq", 1) == 0 or (epoch + 1) == args.epochs or save_force
    if not (is_main_process() and due):
        return
    out = Path(args.output_dir)
    fused = getattr(optimizer, "_dyt_fused", None)
    if fused is not None:
        fused.publish_torch_state()   # the driver's torch.optim.AdamW reports the fused optimizer's moments and step count
    save_on_master({
        'model': {k: v.detach().cpu() for k, v in model_without_ddp.state_dict().items()},
        'optimizer': _to_cpu(optimizer.state_dict()) if optimizer is not None else None,
        'epoch': epoch,
        'scaler': _scaler_entry(optimizer, loss_scaler),
        'args': args,
    }, out / ('checkpoint-%s.pth' % epoch))
    if getattr(args, "auto_remove", False):
        for name in os.listdir(args.output_dir):
            if not (name.startswith('checkpoint-') and name.endswith('.pth')):
                continue
            try:
                e = int(name[len('checkpoint-'):-len('.pth')])
            except ValueError:
                continue
            if e < epoch:
                os.remove(os.path.join(args.output_dir, name))


def load_model(args, model_without_ddp, optimizer, loss_scaler=None):
    """Reference :334-352: resume model (+ optimizer, epoch, scaler unless evaluating) from ``args.resume``."""
    if not getattr(args, "resume", None):
        return
    if str(args.resume).startswith('https'):
        checkpoint = torch.hub.load_state_dict_from_url(args.resume, map_location='cpu', check_hash=True)
    else:
        checkpoint = torch.load(args.resume, map_location='cpu', weights_only=False)
    weights = checkpoint.get('model', checkpoint.get('module', checkpoint))
    model_without_ddp.load_state_dict(weights)
    print("Resume checkpoint %s" % args.resume)
    if 'optimizer' in checkpoint and 'epoch' in checkpoint and not getattr(args, 'eval', False):
        if optimizer is not None and checkpoint['optimizer'] is not None:
            optimizer.load_state_dict(checkpoint['optimizer'])
            fused = getattr(optimizer, "_dyt_fused", None)
            if fused is not None:   # an already adopted torch optimizer: the fused one takes the loaded moments over again
                fused.adopt_torch_state()
        args.start_epoch = checkpoint['epoch'] + 1
        scaler = checkpoint.get('scaler')
        if loss_scaler is not None and scaler:
            loss_scaler.load_state_dict(scaler)
        elif optimizer is not None and scaler:
            fused = getattr(optimizer, "_dyt_fused", optimizer)
            if hasattr(fused, 'load_scaler_state'):
                fused.load_scaler_state(scaler)
        print("With optim & sched!")
