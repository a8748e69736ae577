"""The fixture recipes under tests/golden/ must stay runnable: every make_golden*.py is imported in a fresh interpreter (which installs the
third-party stand-ins, imports the REAL reference from /root/reference and builds nothing yet).  Build container only: the GPU box has no
/root/reference, the test is skipped there.  (Round 3's make_golden_video.py imported this repository's mirror of `video_models` instead
of the reference's namespace package and no longer ran.)"""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
RECIPES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, "make_golden*.py")))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("recipe", RECIPES)
def test_golden_recipe_imports_the_reference(recipe):
    code = ("import sys, os; sys.path.insert(0, %r); m = __import__(%r); "
            "import models.vision_transformer_IN21K as ref; "
            "assert os.path.realpath(ref.__file__).startswith('/root/reference'), ref.__file__; "
            "v = sys.modules.get('video_models.video_vision_transformer_IN21K'); "
            "assert v is None or os.path.realpath(v.__file__).startswith('/root/reference'), v.__file__; print('ok')") % (GOLDEN, recipe)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (recipe, r.stdout[-500:], r.stderr[-1500:])


def test_every_fixture_has_a_recipe():
    made = {"step_r64.npz", "step_r8.npz", "eval_r64.npz", "eval_acc.npz", "count_flops.npz", "video_step.npz", "drop_path_step.npz", "learnable_scalar_step.npz",
            "adapter_ln_in_step.npz", "adapter_ln_out_step.npz", "mixup_step.npz"}
    have = {os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*.npz"))}
    assert have == made, (have, made)
    assert len(RECIPES) >= 4, RECIPES
