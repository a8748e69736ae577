"""Pins oracle/dyt_oracle.py to the golden vectors captured from the real reference
(tests/golden/make_golden.py).  CPU only; runs in the build container and on the GPU box."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import dyt_oracle as O


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def state(g):
    return synth.make_state_dict(int(g["meta_num_classes"]), int(g["meta_ffn_num"]), seed=int(g["meta_seed"]),
                                 kind="test", gate_bias=float(g["meta_gate_bias"]))


@pytest.fixture(scope="module")
def step64(golden_dir):
    g = load(golden_dir, "step_r64.npz")
    B, C, r = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"])
    sd = state(g)
    x, y = synth.make_batch(B, C, seed=int(g["meta_seed"]))
    keep = synth.make_dropout_masks(B, r, seed=int(g["meta_seed"]) + 3)
    g1, g2 = torch.from_numpy(g["s0_g1"]), torch.from_numpy(g["s0_g2"])
    pass
    d, grads, outs = O.step_grads(sd, x, y, g1, g2, keep, scale=float(g["meta_scale"]), mode="masked",
                                  token_target_ratio=float(g["meta_target_ratio"]))
    return g, sd, d, grads, outs, (x, y, g1, g2, keep)


def test_forward_logits_and_masks(step64):
    g, sd, d, grads, (ls, lt, tok), _ = step64
    assert np.abs(ls.detach().numpy() - g["s0_logits_student"]).max() < 2e-5
    assert np.abs(lt.detach().numpy() - g["s0_logits_teacher"]).max() < 2e-5
    assert np.abs(tok["token_logits"].detach().numpy() - g["s0_token_logits"]).max() < 2e-5
    # masks bit-exact (min |z| margin of this fixture is 1.2e-4 >> fp32 round-off of the logits)
    assert float(g["s0_min_gate_margin"]) > 1e-5
    assert np.array_equal(tok["token_select"].detach().numpy().astype(np.uint8), g["s0_token_select"])


def test_block_outputs(step64):
    g, sd, _, _, _, (x, y, g1, g2, keep) = step64
    with torch.no_grad():
        _, o = O.forward(sd, x, g1[0], g2[0], keep[0], float(g["meta_scale"]), False, True, return_blocks=True)
        _, ot = O.forward(sd, x, g1[1], g2[1], keep[1], float(g["meta_scale"]), True, True, return_blocks=True)
    sub = g["sub_tokens"].tolist()
    for name, blocks in (("s0_blocks_student", o["blocks"]), ("s0_blocks_teacher", ot["blocks"])):
        for i in range(12):
            err = np.abs(blocks[i + 1][:, sub].numpy() - g[name][i]).max()
            assert err < 5e-5, (name, i, err)


def test_loss_components(step64):
    g, _, d, _, _, _ = step64
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(float(d[k]) - float(g["s0_stat_" + k])) < 1e-5 * max(1.0, abs(float(g["s0_stat_" + k]))), k


def test_trainable_grads(step64):
    g, _, _, grads, _, _ = step64
    assert len(grads) == 74
    for n, gr in grads.items():
        ref_norm = float(g["s0_gradnorm/" + n])
        assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-9, n
        key = "s0_grad/" + n
        if key in g:
            assert np.abs(gr.numpy() - g[key]).max() <= 1e-4 * np.abs(g[key]).max() + 1e-9, n


def test_adamw_step(step64):
    g, sd, _, grads, _, _ = step64
    for n, gr in grads.items():
        key = "s0_param_after/" + n
        if key not in g:
            continue
        # pin the AdamW arithmetic itself on the reference's own gradient ...
        gref = torch.from_numpy(g["s0_grad/" + n])
        p, _, _ = O.adamw_update(sd[n], gref, torch.zeros_like(gr), torch.zeros_like(gr), 1,
                                 float(g["meta_lr"]), float(g["meta_wd"]))
        assert np.abs(p.numpy() - g[key]).max() < 1e-7, n
        # ... and end to end on the oracle's gradient (elements with |g| ~ eps=1e-8 amplify
        # round-off of g into the normalised update m/(sqrt(v)+eps): allow 1 % of lr)
        p, _, _ = O.adamw_update(sd[n], gr, torch.zeros_like(gr), torch.zeros_like(gr), 1,
                                 float(g["meta_lr"]), float(g["meta_wd"]))
        err = np.abs(p.numpy() - g[key])
        big = np.abs(gref.numpy()) > 1e-6
        assert err[big].max(initial=0.0) < 1e-5 and err.max() < 2.1 * float(g["meta_lr"]), n


def test_two_steps_vtab_shape(golden_dir):
    """r=8 / scale 1 / wd 1e-4 (main_vtab.py) + AdaLoss minimal-token term, two AdamW steps."""
    g = load(golden_dir, "step_r8.npz")
    B, C, r, seed = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"]), int(g["meta_seed"])
    sd = state(g)
    opt = {}
    for s in range(2):
        x, y = synth.make_batch(B, C, seed=seed + 10 * s)
        keep = synth.make_dropout_masks(B, r, seed=seed + 3 + 10 * s)
        g1, g2 = torch.from_numpy(g["s%d_g1" % s]), torch.from_numpy(g["s%d_g2" % s])
        d = O.train_step(sd, opt, x, y, g1, g2, keep, lr=float(g["meta_lr"]), wd=float(g["meta_wd"]),
                         scale=float(g["meta_scale"]), token_target_ratio=float(g["meta_target_ratio"]),
                         token_minimal=float(g["meta_token_minimal"]),
                         token_minimal_weight=float(g["meta_token_minimal_weight"]))
        for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
            ref = float(g["s%d_stat_%s" % (s, k)])
            assert abs(float(d[k]) - ref) < 2e-5 * max(1.0, abs(ref)), (s, k, float(d[k]), ref)
        for n in O.trainable_names(sd):
            key = "s%d_param_after/%s" % (s, n)
            if key in g:
                err = np.abs(sd[n].detach().numpy() - g[key])
                big = np.abs(g["s%d_grad/%s" % (s, n)]) > 1e-6
                assert err[big].max(initial=0.0) < 5e-5 and err.max() < 4.2 * float(g["meta_lr"]), (s, n)


def test_eval_and_gather_equivalence(golden_dir):
    g = load(golden_dir, "eval_r64.npz")
    B, C, seed = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_seed"])
    sd = state(g)
    x, y = synth.make_batch(B, C, seed=seed)
    with torch.no_grad():
        lm, tm = O.forward(sd, x, scale=float(g["meta_scale"]), training=False, mode="masked")
        lg, tg = O.forward(sd, x, scale=float(g["meta_scale"]), training=False, mode="gather")
    assert np.abs(lm.numpy() - g["logits"]).max() < 2e-5
    assert np.abs(lg.numpy() - g["logits_gathered"]).max() < 2e-5
    assert np.abs(lm.numpy() - lg.numpy()).max() < 1e-5  # the reference's two implementations agree
    assert np.array_equal(tm["token_select"].numpy().astype(np.uint8), g["token_select"])
    assert abs(float(O.accuracy(lm, y, topk=(1, 5))[0]) - float(g["metric"])) < 1e-6


def test_eval_accuracy_fixture_is_not_degenerate(golden_dir):
    """tests/golden/eval_acc.npz (make_golden_eval_acc.py): targets chosen from the reference's own ranking so that its
    evaluate() returns acc1 = 5/12, acc5 = 8/12 -- the oracle's forward + accuracy() and the mirror's metrics reproduce them."""
    from util.metrics import accuracy, mean_per_class_accuracy
    g = load(golden_dir, "eval_acc.npz")
    B, C, seed = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_seed"])
    sd = state(g)
    x, _ = synth.make_batch(B, C, seed=seed)
    y = torch.from_numpy(g["targets"])
    with torch.no_grad():
        lm, _ = O.forward(sd, x, scale=float(g["meta_scale"]), training=False, mode="masked")
    assert np.abs(lm.numpy() - g["logits"]).max() < 2e-5
    assert 0.0 < float(g["metric_accuracy"]) < float(g["acc5"]) < 100.0
    a1, a5 = O.accuracy(lm, y, topk=(1, 5))
    assert abs(float(a1) - float(g["metric_accuracy"])) < 1e-6 and abs(float(a5) - float(g["acc5"])) < 1e-6
    b1, b5 = accuracy(lm, y, topk=(1, 5))
    assert abs(float(b1) - float(g["metric_accuracy"])) < 1e-6 and abs(float(b5) - float(g["acc5"])) < 1e-6
    assert abs(float(mean_per_class_accuracy(lm, y, C)) - float(g["metric_mean_per_class_acc"])) < 1e-5


def test_compact_mode_semantics(step64):
    """compact mode: same forward values, gate gradient only from kept tokens."""
    g, sd, d, grads, outs, (x, y, g1, g2, keep) = step64
    d2, grads2, outs2 = O.step_grads(sd, x, y, g1, g2, keep, scale=float(g["meta_scale"]), mode="compact",
                                     token_target_ratio=float(g["meta_target_ratio"]))
    assert torch.equal(outs[0], outs2[0]) and abs(float(d["loss"]) - float(d2["loss"])) < 1e-6
    n = "blocks.5.mlp_token_select.mlp_head.weight"
    assert (grads[n] - grads2[n]).norm() > 1e-3 * grads[n].norm()  # the gate gradient differs...
    n = "head.weight"
    assert (grads[n] - grads2[n]).norm() < 1e-5 * grads[n].norm()  # ...the head gradient does not


# ---- video model: attentive pooling head over t*197 tokens per clip ----
@pytest.fixture(scope="module")
def video(golden_dir):
    g = load(golden_dir, "video_step.npz")
    clips, frames, C, r = int(g["meta_clips"]), int(g["meta_frames"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"])
    seed = int(g["meta_seed"])
    sd = synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=float(g["meta_gate_bias"]), video=True)
    x, _ = synth.make_batch(clips * frames, C, seed=seed)
    y = torch.from_numpy(g["targets"])
    keep = synth.make_dropout_masks(clips * frames, r, seed=seed + 3)
    g1, g2 = torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"])
    d, grads, outs = O.step_grads(sd, x, y, g1, g2, keep, scale=float(g["meta_scale"]), mode="masked",
                                  token_target_ratio=float(g["meta_target_ratio"]), frames=frames)
    return g, sd, d, grads, outs, (x, y, g1, g2, keep, frames)


def test_video_eval_forward(video):
    g, sd, _, _, _, (x, y, g1, g2, keep, frames) = video
    with torch.no_grad():
        logits, tok = O.forward(sd, x, scale=float(g["meta_scale"]), training=False, frames=frames)
    assert logits.shape == g["eval_logits"].shape
    assert np.abs(logits.numpy() - g["eval_logits"]).max() < 2e-5
    assert float(g["eval_min_gate_margin"]) > 1e-5
    assert np.array_equal(tok["token_select"].numpy().astype(np.uint8), g["eval_token_select"])


def test_video_step_losses_and_logits(video):
    g, _, d, _, (ls, lt, tok), _ = video
    assert np.abs(ls.detach().numpy() - g["logits_student"]).max() < 2e-5
    assert np.abs(lt.detach().numpy() - g["logits_teacher"]).max() < 2e-5
    assert np.array_equal(tok["token_select"].detach().numpy().astype(np.uint8), g["token_select"])
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(float(d[k]) - float(g["stat_" + k])) < 1e-5 * max(1.0, abs(float(g["stat_" + k]))), k


def test_video_trainable_grads(video):
    g, sd, _, grads, _, _ = video
    assert len(grads) == 74 + 14   # adapters/gates/head + query_token + 13 attentive_blocks tensors
    stride = int(g["meta_row_stride"])
    for n, gr in grads.items():
        ref_norm = float(g["gradnorm/" + n])
        # absolute floor 1e-7: attentive_blocks.norm_k.bias has an exactly-zero true gradient (a constant added to every
        # key shifts all scores of a clip equally); what is stored is round-off (2.7e-9) that depends on the BLAS threading
        assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-7, n
        if "grad/" + n in g:
            assert np.abs(gr.numpy() - g["grad/" + n]).max() <= 1e-4 * np.abs(g["grad/" + n]).max() + 1e-7, n
        if "gradrows/" + n in g:
            ref = g["gradrows/" + n]
            assert np.abs(gr[::stride].numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9, n


def test_oracle_count_flops_variant_vs_reference(golden_dir):
    """Block.forward_count_flops (reference :167-185) through the reference's own blocks: logits for 1 / 57 / 197 MLP tokens."""
    g = dict(np.load(os.path.join(golden_dir, "count_flops.npz")))
    sd = synth.make_state_dict(int(g["meta_num_classes"]), int(g["meta_ffn_num"]), seed=int(g["meta_seed"]), kind="test",
                               gate_bias=float(g["meta_gate_bias"]))
    x, _ = synth.make_batch(int(g["meta_batch"]), int(g["meta_num_classes"]), seed=int(g["meta_seed"]))
    for n in g["tokens"]:
        with torch.no_grad():
            logits, _ = O.forward(sd, x, scale=float(g["meta_scale"]), training=False, count_flops_tokens=int(n))
        assert np.abs(logits.numpy() - g["logits_n%d" % n]).max() < 2e-5, n


@pytest.mark.parametrize("frames", [1, 2])
def test_chunked_step_grads_equal_step_grads(frames):
    """oracle.step_grads_chunked (used for the full-size configs[4] GPU test, whose autograd graph does not fit a host) is the
    same function as the golden-pinned step_grads: 4 samples in chunks of 1 and 2 (image model) / 2 clips x 2 frames."""
    n, C, r = (4, 10, 8) if frames == 1 else (2, 7, 8)
    B = n * frames
    sd = synth.make_state_dict(C, r, seed=5, kind="test", gate_bias=0.4, video=frames > 1)
    x, y = synth.make_batch(B, C, seed=6)
    y = y[:n].contiguous()
    g1, g2 = synth.make_noise(B, seed=7)
    keep = synth.make_dropout_masks(B, r, seed=8)
    kw = dict(scale=0.5, mode="masked", token_target_ratio=0.5, frames=frames)
    d, g, (ls, lt, tok) = O.step_grads(sd, x, y, g1, g2, keep, **kw)
    for chunk in ((1, 2) if frames == 1 else (1,)):
        d2, g2_, (ls2, lt2, tok2) = O.step_grads_chunked(sd, x, y, g1, g2, keep, chunk, **kw)
        assert float((ls2 - ls.detach()).abs().max()) < 1e-5 and float((lt2 - lt.detach()).abs().max()) < 1e-5
        assert torch.equal(tok2["token_select"], tok["token_select"].detach())
        for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
            assert abs(float(d2[k]) - float(d[k])) < 1e-5 * max(1.0, abs(float(d[k]))), (k, float(d2[k]), float(d[k]))
        for k in g:
            e = float((g2_[k] - g[k]).norm() / max(float(g[k].norm()), 1e-4))   # floor: norm_k.bias has an exactly-zero true gradient
            assert e < (5e-4 if g[k].numel() == 1 else 5e-5), (chunk, k, e)   # fp32 summation order; the 1-element gate biases are sums of signed terms


def test_drop_path_step_vs_reference(golden_dir):
    """Stochastic depth: the reference model built with drop_path_rate = 0.3, one step through its own train_one_epoch with recorded
    Bernoulli uniforms (tests/golden/make_golden_drop_path.py; 31 of the 88 branch instances of the two passes dropped) against the
    oracle's restatement: factors, logits, masks, loss components and all 74 gradients."""
    g = load(golden_dir, "drop_path_step.npz")
    B, C, r = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"])
    sd = state(g)
    x, y = synth.make_batch(B, C, seed=int(g["meta_seed"]))
    keep = synth.make_dropout_masks(B, r, seed=int(g["meta_seed"]) + 3)
    u = torch.from_numpy(g["drop_uniforms"])
    scales = torch.stack([O.drop_path_scales(u[p], float(g["meta_rate"])) for p in range(2)])
    assert torch.equal(scales, torch.from_numpy(g["drop_scales"]))
    assert int((scales[:, :, 1:] == 0).sum()) >= 6 and bool((scales[:, :, 0] == 1).all())
    d, grads, outs = O.step_grads(sd, x, y, torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"]), keep, scale=float(g["meta_scale"]),
                                  mode="masked", drop_scales=scales)
    assert np.abs(outs[0].detach().numpy() - g["logits_student"]).max() < 2e-5
    assert np.abs(outs[1].detach().numpy() - g["logits_teacher"]).max() < 2e-5
    assert np.array_equal(outs[2]["token_select"].detach().numpy().astype(np.uint8), g["token_select"])
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(float(d[k]) - float(g["stat_" + k])) < 1e-5 * max(1.0, abs(float(g["stat_" + k]))), k
    assert len(grads) == 74
    for n, gr in grads.items():
        ref_norm = float(g["gradnorm/" + n])
        assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-9, n
        if "grad/" + n in g:
            assert np.abs(gr.numpy() - g["grad/" + n]).max() <= 1e-4 * np.abs(g["grad/" + n]).max() + 1e-9, n
    # and it is not the drop_path 0 step: without the factors the student logits move by far more than the tolerance
    d0, _, outs0 = O.step_grads(sd, x, y, torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"]), keep, scale=float(g["meta_scale"]), mode="masked")
    assert np.abs(outs0[0].detach().numpy() - g["logits_student"]).max() > 1e-2


def test_learnable_scalar_step_vs_reference(golden_dir):
    """ffn_adapter_scalar == "learnable_scalar": the reference model stepped through its own train_one_epoch with twelve distinct scales
    (tests/golden/make_golden_learnable_scalar.py) against the oracle: logits, masks, losses, all 86 gradients incl. the twelve d(scale)."""
    g = load(golden_dir, "learnable_scalar_step.npz")
    B, C, r = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"])
    sd = synth.add_learnable_scales(state(g), seed=int(g["meta_seed"]))
    x, y = synth.make_batch(B, C, seed=int(g["meta_seed"]))
    keep = synth.make_dropout_masks(B, r, seed=int(g["meta_seed"]) + 3)
    d, grads, outs = O.step_grads(sd, x, y, torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"]), keep, scale=float("nan"), mode="masked")
    assert np.abs(outs[0].detach().numpy() - g["logits_student"]).max() < 2e-5
    assert np.abs(outs[1].detach().numpy() - g["logits_teacher"]).max() < 2e-5
    assert np.array_equal(outs[2]["token_select"].detach().numpy().astype(np.uint8), g["token_select"])
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(float(d[k]) - float(g["stat_" + k])) < 1e-5 * max(1.0, abs(float(g["stat_" + k]))), k
    assert len(grads) == 86 and sum(n.endswith("adaptmlp.scale") for n in grads) == 12
    for n, gr in grads.items():
        ref_norm = float(g["gradnorm/" + n])
        assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-8, n
        if "grad/" + n in g:
            assert np.abs(gr.numpy() - g["grad/" + n]).max() <= 1e-4 * np.abs(g["grad/" + n]).max() + 1e-8, n


@pytest.mark.parametrize("option", ["in", "out"])
def test_adapter_layernorm_step_vs_reference(golden_dir, option):
    """ffn_adapter_layernorm_option "in" / "out" (the Adapter class's default is "in", models/dynamic_adapter.py:88,95-98,121-122,132-133):
    the reference model stepped through its own train_one_epoch with distinct LayerNorm parameters (tests/golden/make_golden_adapter_ln.py)
    against the oracle: logits, masks, losses, all 98 gradients incl. the 24 d(gamma) / d(beta)."""
    g = load(golden_dir, "adapter_ln_%s_step.npz" % option)
    B, C, r = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"])
    sd = synth.add_adapter_layernorm(state(g), seed=int(g["meta_seed"]))
    sd[O.ADAPTER_LN_KEY] = torch.tensor(int(g["meta_option"]))
    assert int(g["meta_option"]) == {"in": 1, "out": 2}[option]
    x, y = synth.make_batch(B, C, seed=int(g["meta_seed"]))
    keep = synth.make_dropout_masks(B, r, seed=int(g["meta_seed"]) + 3)
    d, grads, outs = O.step_grads(sd, x, y, torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"]), keep, scale=0.1, mode="masked")
    assert np.abs(outs[0].detach().numpy() - g["logits_student"]).max() < 2e-5
    assert np.abs(outs[1].detach().numpy() - g["logits_teacher"]).max() < 2e-5
    assert np.array_equal(outs[2]["token_select"].detach().numpy().astype(np.uint8), g["token_select"])
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(float(d[k]) - float(g["stat_" + k])) < 1e-5 * max(1.0, abs(float(g["stat_" + k]))), k
    assert len(grads) == 98 and sum("adapter_layer_norm_before" in n for n in grads) == 24
    for n, gr in grads.items():
        ref_norm = float(g["gradnorm/" + n])
        assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-8, n
        if "grad/" + n in g:
            assert np.abs(gr.numpy() - g["grad/" + n]).max() <= 1e-4 * np.abs(g["grad/" + n]).max() + 1e-8, n


def test_mixup_step_vs_reference(golden_dir):
    """A ``mixup_fn`` in the reference's train_one_epoch (engine_finetune.py:44-45): class-probability targets through nn.CrossEntropyLoss for
    both passes (tests/golden/make_golden_mixup.py; the callable is synth.mixup_batch) against the oracle with the same targets."""
    g = load(golden_dir, "mixup_step.npz")
    B, C, r = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"])
    sd = state(g)
    x, y = synth.make_batch(B, C, seed=int(g["meta_seed"]))
    xm, t = synth.mixup_batch(x, y, C, lam=float(g["meta_lam"]), smoothing=float(g["meta_smoothing"]))
    assert np.array_equal(t.numpy(), g["soft_targets"]) and abs(float(t.sum()) - B) < 1e-5
    keep = synth.make_dropout_masks(B, r, seed=int(g["meta_seed"]) + 3)
    d, grads, outs = O.step_grads(sd, xm, t, torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"]), keep, scale=0.1, mode="masked")
    assert np.abs(outs[0].detach().numpy() - g["logits_student"]).max() < 2e-5
    assert np.abs(outs[1].detach().numpy() - g["logits_teacher"]).max() < 2e-5
    assert np.array_equal(outs[2]["token_select"].detach().numpy().astype(np.uint8), g["token_select"])
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(float(d[k]) - float(g["stat_" + k])) < 1e-5 * max(1.0, abs(float(g["stat_" + k]))), k
    for n, gr in grads.items():
        ref_norm = float(g["gradnorm/" + n])
        assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-8, n
        if "grad/" + n in g:
            assert np.abs(gr.numpy() - g["grad/" + n]).max() <= 1e-4 * np.abs(g["grad/" + n]).max() + 1e-8, n
