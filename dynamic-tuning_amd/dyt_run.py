#!/usr/bin/env python3
"""Run one of the reference's driver scripts UNCHANGED on the MI355X path:

    python /path/to/dynamic-tuning_amd/dyt_run.py /path/to/Dynamic-Tuning/main_image.py --batch_size 128 ...
    python -m torch.distributed.run --nproc-per-node 8 /path/to/dynamic-tuning_amd/dyt_run.py /path/to/Dynamic-Tuning/main_image.py ...

Why a launcher: ``python main_image.py`` puts the script's own directory FIRST on ``sys.path``, in front of anything
``PYTHONPATH`` names, so the reference's ``models/``, ``engine_finetune.py`` and ``misc.py`` would win.  This file puts
``dynamic-tuning_amd/`` first and the script's directory right behind it, then executes the script as ``__main__`` with its
own ``sys.argv``.  The modules this directory implements (models.vision_transformer_IN21K, models.dynamic_adapter,
models.losses, models.model_speed_test, video_models.video_vision_transformer_IN21K, engine_finetune, misc, block_flops_dict,
util.lr_sched, util.metrics) then resolve here; everything else the drivers import (configs, datasets, video_datasets,
util.pos_embed, util.logger, ...) resolves in the reference tree -- ``util`` and ``video_models`` are namespace packages on
both sides and ``models`` extends its ``__path__``, so nothing of the reference is shadowed (tests/test_driver_surface.py).
"""
import os
import runpy
import sys

# before the HIP runtime starts: the step's pass streams + gradient-sum + all-reduce + RCCL streams need more than four hardware queues
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HERE = os.path.dirname(os.path.abspath(__file__))


def path_for(script):
    """sys.path with dynamic-tuning_amd/ first and the driver's directory second."""
    ref_root = os.path.dirname(os.path.abspath(script))
    rest = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (HERE, ref_root)]
    return [HERE, ref_root] + rest


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = argv[0]
    if not os.path.isfile(script):
        print("dyt_run: no such driver script: %s" % script, file=sys.stderr)
        return 2
    sys.path[:] = path_for(script)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
