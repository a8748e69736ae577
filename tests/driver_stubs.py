"""Stand-ins that let the reference's DRIVER scripts (main_image.py, main_vtab.py, main_video.py, speed.py) be imported and
driven in the build container, where their third-party packages are not installed and no dataset exists.
Test infrastructure only; none of it is on the product path.

* third-party: ``easydict.EasyDict``, ``timm`` (only the names the drivers touch: ``trunc_normal_``, ``create_model``,
  ``Mixup``, the mean/std constants), ``torch.utils.tensorboard.SummaryWriter``, ``termcolor.colored``, ``torch._six.inf``;
* the reference's dataset builders (out of scope; they need torchvision / decord): ``datasets.image_datasets``,
  ``datasets.image_datasets_noaug`` and ``video_datasets.video_datasets`` are replaced by builders of tiny synthetic
  ``TensorDataset``s so that ``main(args)`` can run up to its first training step.
"""
import sys
import types

import torch


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class EasyDict(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class SummaryWriter:
    def __init__(self, log_dir=None, **kw):
        self.log_dir = log_dir
        self.scalars = []

    def add_scalar(self, tag, value, step=None):
        self.scalars.append((tag, float(value), step))

    def flush(self):
        pass

    def close(self):
        pass


def install(n_train=8, n_val=6, nb_classes=10, frames=2):
    def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
        return torch.nn.init.trunc_normal_(t, mean, std, a, b)

    _mod("easydict", EasyDict=EasyDict)
    timm = _mod("timm", __version__="0.9.12")
    timm.models = _mod("timm.models", create_model=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("timm stub")))
    timm.models.layers = _mod("timm.models.layers", trunc_normal_=trunc_normal_)
    timm.layers = _mod("timm.layers", trunc_normal_=trunc_normal_)
    timm.data = _mod("timm.data", Mixup=type("Mixup", (), {}), IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    timm.loss = _mod("timm.loss")
    _mod("termcolor", colored=lambda s, *a, **k: s)
    _mod("torch._six", inf=float("inf"))
    tb = _mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    torch.utils.tensorboard = tb

    g = torch.Generator().manual_seed(0)

    def image_sets(args):
        def ds(n):
            d = torch.utils.data.TensorDataset(torch.randn(n, 3, 224, 224, generator=g), torch.randint(0, nb_classes, (n,), generator=g))
            d.transform = None   # main_vtab.py:143-144 prints it
            return d
        return ds(n_train), ds(n_val), nb_classes, "accuracy"

    def video_sets(args):   # main_video.py:177: (train, val, metric); validation samples carry V views [V, c, t, h, w]
        def ds(n, views):
            x = torch.randn(n, 3, frames, 224, 224, generator=g)
            return torch.utils.data.TensorDataset(x.unsqueeze(1) if views else x, torch.randint(0, nb_classes, (n,), generator=g))
        return ds(n_train, False), ds(n_val, True), "accuracy"

    # the reference's `datasets` / `video_datasets` packages import torchvision / decord at module level: replace the three builder modules
    import importlib.machinery
    for pkg in ("datasets", "video_datasets"):
        m = _mod(pkg)
        m.__path__ = []   # a package without a search path: only the stubbed sub-modules below exist
        m.__spec__ = importlib.machinery.ModuleSpec(pkg, None, is_package=True)
    _mod("datasets.image_datasets", build_image_dataset=image_sets)
    _mod("datasets.image_datasets_noaug", build_image_dataset=image_sets)
    _mod("video_datasets.video_datasets", build_dataset=video_sets)
