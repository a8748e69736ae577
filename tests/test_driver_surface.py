"""The drop-in boundary as the reference's DRIVERS see it.

``tests/golden/driver_surface.json`` (made by tests/golden/make_driver_surface.py from the reference's main_image.py,
main_vtab.py, main_video.py, speed.py and engine_finetune.py) lists every (module, attribute) those files import or touch
on the boundary's modules, the call sites with their argument shapes, and the attributes they use on the model.  Here:

* every pair resolves in dynamic-tuning_amd/ and every call site binds to the mirror's signature (CPU, no reference needed);
* nothing the reference keeps for itself (util.pos_embed, util.logger, configs, datasets, ...) is shadowed by this package;
* where /root/reference exists (build container): the REAL drivers are imported with dynamic-tuning_amd/ in front of the
  reference root exactly as dyt_run.py arranges it, and main_image.main(args) is driven -- argument parser, model factory,
  checkpoint surgery, freeze rule, ``torch.optim.AdamW``, ``NativeScaler()``, ``misc.load_model``, the FLOPs tables,
  ``train_one_epoch`` -- up to the first fused step, which must refuse the CPU device loudly (there is no CPU path).
"""
import importlib
import inspect
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "dynamic-tuning_amd")
REF = "/root/reference"
SURFACE = json.load(open(os.path.join(HERE, "golden", "driver_surface.json")))["reference_files"]
PAIRS = sorted({(m, a) for f in SURFACE.values() for m, a in f["imports"]}, key=lambda e: (e[0], e[1] or ""))
CALLS = [(fn, c) for fn, f in SURFACE.items() for c in f["calls"]]


@pytest.mark.parametrize("module,attr", PAIRS)
def test_every_symbol_the_drivers_touch_resolves_here(module, attr):
    mod = importlib.import_module(module)
    where = os.path.realpath(getattr(mod, "__file__", None) or list(mod.__path__)[0])
    assert where.startswith(os.path.realpath(PKG)), (module, where)
    if attr is not None:
        assert hasattr(mod, attr), "%s.%s (used by the reference's drivers) is missing" % (module, attr)


def test_every_driver_call_site_binds_to_the_mirror_signature():
    bad = []
    for fn, c in CALLS:
        obj = getattr(importlib.import_module(c["module"]), c["name"])
        try:
            inspect.signature(obj).bind(*([None] * c["npos"]), **{k: None for k in c["keywords"]})
        except TypeError as e:
            bad.append("%s:%d %s.%s(%d positional, %s): %s" % (fn, c["line"], c["module"], c["name"], c["npos"], c["keywords"], e))
    assert not bad, "\n".join(bad)


def test_model_object_has_what_the_drivers_use():
    from test_host import _model
    m = _model(num_classes=7, ffn_num=8)
    for f in SURFACE.values():
        for attr in f["model_attrs"]:
            obj = m
            for part in attr.split("."):
                assert hasattr(obj, part), attr
                obj = getattr(obj, part)
    from models.model_speed_test import vit_base_patch16_224_in21k as speed_factory
    assert inspect.signature(speed_factory).bind(num_classes=1, drop_path_rate=0.0, tuning_config=None, select_config=None)


def test_reference_owned_modules_are_not_shadowed():
    """With only dynamic-tuning_amd/ on the path, the modules the reference keeps must be ABSENT (not half-present): `util`
    and `video_models` are namespace packages here, `models` extends its __path__, there is no `configs` / `datasets`."""
    owned = sorted({m for f in SURFACE.values() for m, _ in f["reference_owned"]})
    code = "import sys, importlib.util as u\nsys.path[:] = [%r] + [p for p in sys.path if 'site-packages' in p or 'lib/python' in p]\n" % PKG
    code += "for m in %r:\n    top = m.split('.')[0]\n    if top in ('datasets',):\n        continue\n" % (owned,)   # HF `datasets` sits in site-packages
    code += "    try:\n        s = u.find_spec(m)\n    except ModuleNotFoundError:\n        s = None\n    assert s is None, (m, s)\n"
    code += "import util, video_models, models\nassert util.__file__ is None and video_models.__file__ is None, 'regular packages would shadow the reference'\nprint('ok')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])


DRIVE = r'''
import os, sys, tempfile, types
sys.path.insert(0, %(tests)r)
sys.path.insert(0, %(pkg)r)
import dyt_run
sys.path[:] = dyt_run.path_for(os.path.join(%(ref)r, "main_image.py"))
assert sys.path[0] == %(pkg)r and sys.path[1] == %(ref)r, sys.path[:3]
sys.path.append(%(tests)r)
import driver_stubs
driver_stubs.install()
import torch
import main_image, main_vtab, main_video
import runpy
runpy.run_path(os.path.join(%(ref)r, "speed.py"), run_name="speed_imported")
import misc, engine_finetune, block_flops_dict, models.losses, models.vision_transformer_IN21K as vit, models.model_speed_test as mst
import util.lr_sched, util.metrics, util.pos_embed, util.logger, configs
import video_models.video_vision_transformer_IN21K as vvit
ours = [misc, engine_finetune, block_flops_dict, models.losses, vit, mst, util.lr_sched, util.metrics, vvit]
theirs = [util.pos_embed, util.logger, configs, main_image]
for m in ours:
    assert os.path.realpath(m.__file__).startswith(os.path.realpath(%(pkg)r)), m.__file__
for m in theirs:
    assert os.path.realpath(m.__file__).startswith(%(ref)r), m.__file__
import models.dynamic_adapter
import importlib
ref_twin = importlib.util.find_spec("video_models.video_model_speed_test")   # a module only the reference has, same namespace package
assert ref_twin is not None and ref_twin.origin.startswith(%(ref)r), ref_twin
print("imports ok")

# ---- drive main_image.main(args) up to the first fused step -------------------------------------------------------------------
import synth
tmp = tempfile.mkdtemp()
ck = os.path.join(tmp, "VIT_BASE_IN21K.pth")
sd = synth.make_state_dict(21843, 64, seed=0, kind="test")
torch.save({k: v for k, v in sd.items() if not synth.is_trainable(k) or k.startswith("head.")}, ck)   # a timm checkpoint: backbone + a 21k-class head
args = main_image.get_args_parser().parse_args(["--batch_size", "4", "--epochs", "2", "--blr", "0.1", "--finetune", "VIT_BASE_IN21K", "--device", "cpu",
                                                "--output_dir", tmp, "--num_workers", "0", "--dataset", "cifar100", "--ffn_adapt", "--ffn_num", "64",
                                                "--warmup_epochs", "1"])
args.data_path, args.pretrain_ckpts = configs.DATASETS, {"VIT_BASE_IN21K": ck}
seen = {}
real_step = engine_finetune.train_step
def spy(model, samples, targets, optimizer, criterion=None, **kw):
    seen.update(model=model, optimizer=optimizer, criterion=criterion, samples=samples, kw=kw)
    return real_step(model, samples, targets, optimizer, criterion, **kw)
engine_finetune.train_step = spy
from _lib import DyTError
try:
    main_image.main(args)
    raise SystemExit("main() ran a training step on the CPU: there must be no CPU path")
except DyTError as e:
    assert "HIP device only" in str(e), e
opt = seen["optimizer"]
assert isinstance(opt, engine_finetune.FusedAdamW) and isinstance(opt._torch, torch.optim.AdamW), type(opt)
assert opt.param_groups is opt._torch.param_groups and len(opt._torch.param_groups[0]["params"]) == 74
assert abs(args.lr - 0.1 * 4 / 256) < 1e-12 and opt.param_groups[0]["weight_decay"] == 0.01
assert opt.param_groups[0]["lr"] == 0.0   # first iteration of the warm-up: lr_sched wrote through the shared group
assert args.nb_classes == 10 and seen["model"].head.weight.shape == (10, 768)      # the 21k head was dropped (main_image.py:233-236)
assert seen["criterion"].token_target_ratio == 0.5 and tuple(seen["samples"].shape) == (4, 3, 224, 224)
assert sum(p.requires_grad for p in seen["model"].parameters()) == 74
print("main_image drive ok")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_real_drivers_import_and_run_to_the_first_step_with_this_package_in_front():
    code = textwrap.dedent(DRIVE) % dict(pkg=PKG, ref=REF, tests=HERE)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd="/tmp")
    assert r.returncode == 0 and "imports ok" in r.stdout and "main_image drive ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_driver_surface_fixture_is_current():
    """The committed JSON is what the recipe produces from the reference today."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_driver_surface as mk
    for name in mk.DRIVERS:
        assert mk.scan(os.path.join(REF, name)) == SURFACE[name], name


MISC_VS_REF = r'''
import os, sys, io, logging, math, types, importlib.util
sys.path.insert(0, %(tests)r)
import driver_stubs
driver_stubs.install()
import torch
def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
ref = load(os.path.join(%(ref)r, "misc.py"), "ref_misc")
ours = load(os.path.join(%(pkg)r, "misc.py"), "our_misc")
# SmoothedValue: same statistics and the same rendered strings for the format strings the loops use
g = torch.Generator().manual_seed(0)
vals = torch.rand(57, generator=g).tolist()
for win, fmt in ((20, None), (1, "{value:.6f}"), (10, "{avg:.4f}"), (5, "{median:.3f} / {max:.3f} / {global_avg:.5f}")):
    a, b = ref.SmoothedValue(window_size=win, fmt=fmt), ours.SmoothedValue(window_size=win, fmt=fmt)
    for i, v in enumerate(vals):
        a.update(v, n=1 + i %% 3); b.update(v, n=1 + i %% 3)
        assert str(a) == str(b), (win, fmt, i, str(a), str(b))
        assert (a.median, a.avg, a.global_avg, a.max, a.value, a.count, a.total) == (b.median, b.avg, b.global_avg, b.max, b.value, b.count, b.total)
# MetricLogger: update / add_meter / attribute access / str, and the lines log_every emits (timings masked)
def run(mod):
    buf = io.StringIO()
    lg = logging.getLogger("t_" + mod.__name__); lg.handlers[:] = [logging.StreamHandler(buf)]; lg.setLevel(logging.INFO); lg.propagate = False
    ml = mod.MetricLogger(delimiter="  ", logger=lg)
    ml.add_meter("lr", mod.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    seen = []
    for i, x in enumerate(ml.log_every(list(range(7)), 3, "Epoch: [0]")):
        ml.update(loss=0.5 * x, lr=1e-3 * (x + 1), skip=None, t=torch.tensor(2.0 * x))
        seen.append(x)
    assert seen == list(range(7)) and ml.loss.count == 7 and abs(ml.t.global_avg - 6.0) < 1e-9
    try:
        ml.nothing
        raise SystemExit("missing meter must raise AttributeError")
    except AttributeError:
        pass
    import re
    lines = [re.sub(r"(eta|time|data|Total time): \S+( \(\S+ s / it\))?", r"\1: _", l) for l in buf.getvalue().splitlines()]
    return lines, str(ml)
la, sa = run(ref); lb, sb = run(ours)
assert la == lb and sa == sb, ("\n".join(la), "\n".join(lb), sa, sb)
# get_grad_norm_: 2-norm, inf-norm, parameters without a gradient, a bare tensor
ps = [torch.nn.Parameter(torch.randn(3, 4, generator=g)) for _ in range(4)]
for p in ps[:3]:
    p.grad = torch.randn(3, 4, generator=g)
for nt in (2.0, 1.0, float("inf")):
    assert torch.allclose(ref.get_grad_norm_(ps, nt), ours.get_grad_norm_(ps, nt)), nt
assert float(ours.get_grad_norm_([ps[3]])) == 0.0 == float(ref.get_grad_norm_([ps[3]]))
assert torch.allclose(ref.get_grad_norm_(ps[0]), ours.get_grad_norm_(ps[0]))
# process-group helpers without a group, all_reduce_mean, init_distributed_mode's single-process branch
for m in (ref, ours):
    assert m.get_world_size() == 1 and m.get_rank() == 0 and m.is_main_process() and not m.is_dist_avail_and_initialized()
    assert m.all_reduce_mean(3.5) == 3.5
    import builtins
    keep = builtins.print
    a = types.SimpleNamespace(dist_on_itp=False, dist_url="env://")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SLURM_PROCID"):
        os.environ.pop(k, None)
    m.init_distributed_mode(a)
    assert a.distributed is False
    builtins.print = keep
assert ours.NativeScalerWithGradNormCount.state_dict_key == ref.NativeScalerWithGradNormCount.state_dict_key
print("misc vs reference ok")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_misc_helpers_equal_the_references_on_the_same_inputs():
    """``SmoothedValue`` / ``MetricLogger`` (statistics, rendered strings, the lines ``log_every`` emits), ``get_grad_norm_`` and the
    single-process branches of the process-group helpers of this package's ``misc.py`` against the REFERENCE's ``misc.py`` loaded from
    /root/reference, on the same inputs (reference misc.py:24-168,188-214,281-293,355-363)."""
    code = textwrap.dedent(MISC_VS_REF) % dict(pkg=PKG, ref=REF, tests=HERE)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and "misc vs reference ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
