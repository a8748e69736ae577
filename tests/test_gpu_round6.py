"""GPU tests added in round 6 (pytest -m gpu), all through the C ABI.
  * the reference's DRIVER sequence (main_image.py:212-346) replayed with this package's modules and the objects the driver itself
    builds -- ``torch.optim.AdamW``, ``NativeScaler()``, ``misc.load_model / save_model`` -- against the FusedAdamW route, bit for bit,
    including a resume from the checkpoint the driver's objects wrote."""
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _driver_objects(precision, C=10, r=64, seed=0, lr=2e-3, wd=0.01):
    """main_image.py:183-307 with this package's modules: factory, non-strict checkpoint load, freeze rule from missing_keys,
    ``.to(device)``, ``torch.optim.AdamW`` over the trainable parameters, ``NativeScaler()``, ``AdaLoss``."""
    import gpu_diag as D
    import synth
    from misc import NativeScalerWithGradNormCount as NativeScaler
    from models.losses import AdaLoss
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                   ffn_adapter_scalar="0.1", ffn_num=r, d_model=768)
    select = D.Cfg(open=True, keep_layers=0, token_ratio=2., token_target_ratio=0.5, token_minimal=0., token_minimal_weight=0.)
    model = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=0.0, tuning_config=tuning, select_config=select, precision=precision,
                                       train_mode="compact")
    full = synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=0.85)
    ckpt = {k: v for k, v in full.items() if not synth.is_trainable(k)}          # a timm checkpoint has no adapters / gates / head of this size
    msg = model.load_state_dict(ckpt, strict=False)
    with torch.no_grad():                                                        # (give the zero-initialised up-projections something to do)
        for k, p in model.named_parameters():
            if synth.is_trainable(k):
                p.copy_(full[k])
    for name, p in model.named_parameters():                                     # main_image.py:250-256
        p.requires_grad = name in msg.missing_keys
    for _, p in model.head.named_parameters():
        p.requires_grad = True
    device = torch.device("cuda", 0)
    model.to(device)
    optimizer = torch.optim.AdamW([p for name, p in model.named_parameters() if p.requires_grad], lr=lr, weight_decay=wd)   # :285
    loss_scaler = NativeScaler()                                                                                              # :290
    criterion = AdaLoss(base_criterion=torch.nn.CrossEntropyLoss(), layer_target_ratio=0.5, layer_loss_ratio=2.0, layer_diverse_ratio=0.0,
                        layer_entropy_weight=0.0, layer_minimal_weight=0.0, layer_minimal=0.0, token_target_ratio=select.token_target_ratio,
                        token_loss_ratio=select.token_ratio, token_minimal=select.token_minimal, token_minimal_weight=select.token_minimal_weight)
    return model, optimizer, loss_scaler, criterion, device


def _loader(B, C, steps, seed):
    import synth
    return [synth.make_batch(B, C, seed=seed + i) for i in range(steps)]


def _args(tmp, C, lr, epochs=4, resume=""):
    return types.SimpleNamespace(accum_iter=1, lr=lr, min_lr=1e-5, warmup_epochs=1, epochs=epochs, metric="accuracy", nb_classes=C,
                                 output_dir=str(tmp), save_freq=1, auto_remove=False, resume=resume, start_epoch=0, eval=False,
                                 tuning_config=types.SimpleNamespace(ffn_num=64))


@pytest.mark.parametrize("precision", ["fp16", "fp16x3q"])
def test_driver_sequence_with_the_drivers_own_optimizer_and_scaler(precision, tmp_path):
    """Three runs over the same seeded batches.  (A) the driver's sequence as main_image.py writes it -- torch.optim.AdamW, NativeScaler(),
    misc.load_model (no resume), train_one_epoch x 2, evaluate, misc.save_model, then two more epochs; (B) the same with a FusedAdamW in
    the optimizer's place (round 1-5's documented route); (C) a NEW set of driver objects that resumes from the checkpoint A's objects
    wrote after epoch 1 and runs epochs 2-3.  All three end with bit-identical parameters, the torch optimizer's state_dict() is the
    fused optimizer's state (moments, step count), and the checkpoint's 'scaler' entry loads into a real GradScaler."""
    import misc
    from block_flops_dict import get_base_flops, get_block_flops
    from engine_finetune import FusedAdamW, evaluate, train_one_epoch
    B, C, lr = 8, 10, 2e-3
    train, val = _loader(B, C, 3, 100), _loader(B, C, 2, 900)
    log = types.SimpleNamespace(info=lambda *a, **k: None)

    def run(kind, tmp, first_epoch=0, last_epoch=4, resume=""):
        torch.manual_seed(7)
        model, optimizer, loss_scaler, criterion, device = _driver_objects(precision, C=C, lr=lr)
        if kind == "fused":
            optimizer = FusedAdamW(model, lr=lr, weight_decay=0.01)
        os.makedirs(tmp, exist_ok=True)                                            # (main_image.py:365: Path(args.output_dir).mkdir)
        args = _args(tmp, C, lr, resume=resume)
        model_without_ddp = model
        misc.load_model(args=args, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)      # :310
        base_flops, flops_dict = get_base_flops(args), get_block_flops(args)                                                # :317-318
        assert args.start_epoch == first_epoch
        stats = None
        for epoch in range(args.start_epoch, last_epoch):
            train_stats = train_one_epoch(model, criterion, train, optimizer, device, epoch, loss_scaler, max_norm=None,   # :329-336
                                          log_writer=None, args=args, logger=log)
            assert set(train_stats) >= {"loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss", "lr"}
            test_stats = evaluate(val, model, device, log, base_flops=base_flops, flops_dict=flops_dict, args=args)       # :340
            assert 0.0 <= test_stats["metric"] <= 100.0 and test_stats["acc1"] == test_stats["metric"]
            misc.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer,             # :344-346
                            loss_scaler=loss_scaler, epoch=epoch, save_force=True)
            stats = train_stats
        torch.cuda.synchronize()
        return model, optimizer, loss_scaler, stats

    ma, oa, sa, stats_a = run("torch", tmp_path / "a")
    mb, ob, sb, stats_b = run("fused", tmp_path / "b")
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    for n in pa:
        assert torch.equal(pa[n].detach(), pb[n].detach()), "driver route vs FusedAdamW route differ in %s" % n
    assert stats_a == stats_b and stats_a["lr"] < lr

    # the driver's torch optimizer reports the fused state: step count = applied updates, moments = the flat buffers
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    assert len(sd_a["state"]) == 74 and sd_a["param_groups"][0]["lr"] == stats_a["lr"] and sd_a["param_groups"][0]["params"] == list(range(74))
    for i in range(74):
        assert float(sd_a["state"][i]["step"]) == 12.0 == float(sd_b["state"][i]["step"])
        assert torch.equal(sd_a["state"][i]["exp_avg"], sd_b["state"][i]["exp_avg"])
        assert torch.equal(sd_a["state"][i]["exp_avg_sq"], sd_b["state"][i]["exp_avg_sq"])
    assert float(sd_a["state"][5]["exp_avg_sq"].abs().max()) > 0
    # the checkpoint on disk: the reference's keys, a GradScaler-loadable scaler entry, compact per-parameter tensors
    ck = torch.load(tmp_path / "a" / "checkpoint-1.pth", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 1
    assert float(ck["optimizer"]["state"][0]["step"]) == 6.0
    assert all(v["exp_avg"].untyped_storage().nbytes() == v["exp_avg"].numel() * 4 for v in ck["optimizer"]["state"].values())
    gs = torch.amp.GradScaler("cpu", enabled=True)
    gs.load_state_dict(ck["scaler"])
    assert gs.get_scale() == ck["scaler"]["scale"] == 2.0 ** ck["scaler"]["scale_log2"] == 4096.0
    assert sa.state_dict() == oa._dyt_fused.scaler_state() and sa.state_dict()["scale"] == 4096.0

    # (C) fresh driver objects resume from A's epoch-1 checkpoint (misc.load_model -> optimizer.load_state_dict -> adoption takes it over)
    mc, oc, sc, stats_c = run("torch", tmp_path / "c", first_epoch=2, resume=str(tmp_path / "a" / "checkpoint-1.pth"))
    pc = dict(mc.named_parameters())
    for n in pa:
        assert torch.equal(pa[n].detach(), pc[n].detach()), "resumed run differs in %s" % n
    assert float(oc.state_dict()["state"][0]["step"]) == 12.0 and stats_c == stats_a


def test_native_scaler_called_like_the_reference_loop_calls_it():
    """The generic autograd route with the driver's objects, as the reference's own loop body runs it (engine_finetune.py:47-79):
    two forwards, the loss assembled in torch, ``loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(),
    update_grad=...)``, ``optimizer.zero_grad()`` -- parameters move, a non-finite loss skips the update and is counted."""
    import torch.nn.functional as F
    import synth
    model, optimizer, loss_scaler, criterion, device = _driver_objects("fp16", C=10, lr=1e-3)
    model.train(True)
    x, y = synth.make_batch(4, 10, seed=5)
    x, y = x.to(device), y.to(device)
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}

    def loop_body(poison=False):
        outputs, token_select = model(x)
        teacher_outputs, _ = model(x, complete_model=True)
        kl = F.kl_div(F.log_softmax(outputs, dim=-1), F.log_softmax(teacher_outputs.detach(), dim=-1), reduction='batchmean', log_target=True)
        teacher_loss = criterion.base_criterion(teacher_outputs, y)
        loss, loss_dict = criterion(dict(prediction=outputs, **token_select), y)
        loss = loss + teacher_loss + kl
        if poison:
            loss = loss * float("inf")
        norm = loss_scaler(loss, optimizer, clip_grad=None, parameters=model.parameters(), create_graph=False, update_grad=True)
        optimizer.zero_grad()
        return float(loss), norm

    loss, norm = loop_body()
    assert loss == loss and float(norm) > 0
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters() if p.requires_grad)
    assert moved == 74
    after = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    loop_body(poison=True)
    assert loss_scaler.skipped == 1
    assert all(torch.equal(after[n], p.detach()) for n, p in model.named_parameters() if p.requires_grad)
    assert set(loss_scaler.state_dict()) >= {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}


@pytest.mark.parametrize("B,steps,graph", [(16, 7, False), (2, 31, False), (16, 7, True)])
def test_prefetched_host_batches_give_the_resident_data_loss_sequence(B, steps, graph, monkeypatch):
    """Input feeding (reference engine_finetune.py:34-42): train_one_epoch over DISTINCT pinned host batches --
    every batch copied on the copy stream into one of two device buffers, handed over by events -- produces, step for step, the
    loss components of the same batches resident on the device, and of the reference's placement of the copy (compute stream,
    DYT_PREFETCH=0); the parameters after the epoch are bit-identical.  B=2 x 31 steps: steps much shorter than a copy, so every
    hand-over (ready / free events, both buffers, the ragged last batch) is exercised many times.  graph=True: the step replayed from a
    captured hipGraph (``args.hip_graph``; the prefetched buffer is copied into the graph's static input on the compute stream)."""
    import engine_finetune as E
    import synth
    C = 10

    def run(kind):
        torch.manual_seed(3)
        model, optimizer, loss_scaler, criterion, device = _driver_objects("fp16", C=C, lr=1e-3)
        host = [synth.make_batch(B if i != steps - 1 else max(1, B - 1), C, seed=40 + i) for i in range(steps)]   # last batch ragged
        if kind == "resident":
            loader = [(x.to(device), y.to(device)) for x, y in host]
        else:
            loader = [(x.pin_memory(), y.pin_memory()) for x, y in host]
        monkeypatch.setenv("DYT_PREFETCH", "0" if kind == "compute_stream" else "1")
        seq = []
        real = E.train_step

        def spy(*a, **kw):
            out = real(*a, **kw)
            seq.append(kw["losses_out"].clone())
            return out
        monkeypatch.setattr(E, "train_step", spy)
        args = _args("/tmp", C, 1e-3)
        args.hip_graph = graph
        stats = E.train_one_epoch(model, criterion, loader, optimizer, device, 0, loss_scaler, max_norm=None, log_writer=None, args=args, logger=None)
        monkeypatch.setattr(E, "train_step", real)
        torch.cuda.synchronize()
        return torch.stack(seq).cpu(), {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}, stats

    seq_r, par_r, st_r = run("resident")
    seq_p, par_p, st_p = run("prefetched")
    seq_c, par_c, st_c = run("compute_stream")
    assert seq_r.shape == (steps, 8) and torch.isfinite(seq_r).all()
    assert len({float(v) for v in seq_r[:, 0]}) == steps          # distinct batches: distinct losses
    assert torch.equal(seq_r, seq_p) and torch.equal(seq_r, seq_c)
    assert st_r == st_p == st_c
    for n in par_r:
        assert torch.equal(par_r[n], par_p[n]) and torch.equal(par_r[n], par_c[n]), n


@pytest.mark.parametrize("precision", ["fp16x3q", "fp16x3h", "fp16f8"])
def test_split_modes_fc2_with_leading_adapter_tiles_vs_the_two_launch_form(precision):
    """Round 6: in the split modes whose backward runs on 16-bit operands the adapter up-projection rides on the fc2 GEMM as three
    leading tiles of a three-part product (gemm.hip: LEAD; 256x256 and 128x128 tiles, fold form and fp8-correction form, gathered
    rows, the cls tail) and the dropped tokens' up-projection runs three-part on the same [hi | lo] images.  DYT_OPT_FC2_CAT = 0
    restores the round-5 form (up-projection on the exact-fp32 MFMA kernel, two fp32 read-modify-write passes).  Same training
    decisions; logits, eval logits and losses agree to the MODE's own round-off: 1e-5 where the pass is three-part throughout (fp16x3h),
    5e-5 for fp16x3q's student pass (qkv / proj in the fp8-correction form: 2.4e-5 from the oracle) and 2e-4 for passes whose MLP
    takes the fp8-correction form (fp16f8; fp16x3q's complete_model pass: 1e-4 from the oracle) -- there a 1e-7 change of the fc2
    output moves e4m3 roundings of the next GEMM's correction operands, i.e. two equally accurate evaluations differ by the form's own
    noise; gradients within the 16-bit backward's bound."""
    from test_gpu_round3 import _step
    from test_gpu_round2 import _grad_tol
    a, b = _step(precision, "compact", 1, B=16), _step(precision, "compact", 0, B=16)
    assert torch.equal(a["ts"], b["ts"]) and torch.equal(a["tse"], b["tse"])
    for k in ("ls", "lt", "le", "lc"):
        d = float((a[k] - b[k]).abs().max())
        tol = {"fp16x3h": 1e-5, "fp16f8": 2e-4, "fp16x3q": 5e-5 if k in ("ls", "le") else 2e-4}[precision]
        print("%s %s: max |fused - two-launch| = %.2e (bound %.0e)" % (precision, k, d, tol))
        assert d < tol, (k, d)
    assert float((a["losses"][:5] - b["losses"][:5]).abs().max()) < 2e-4
    assert torch.equal(a["losses"][5:7], b["losses"][5:7])            # keep ratio, kept tokens
    worst = 0.0
    for n, ga in a["grads"].items():
        gb = b["grads"][n]
        if gb.numel() == 1:
            continue
        e = float((ga - gb).norm() / (gb.norm() + 1e-20))
        worst = max(worst, e)
        assert e < _grad_tol(n, precision), (n, e)
    print("%s: worst gradient rel-L2 between the two forms %.2e" % (precision, worst))


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_forward_features_and_forward_head_compose_to_forward(precision):
    """``forward_features`` / ``forward_head`` of the reference (models/vision_transformer_IN21K.py:343-380), which raised until round 5:
    the block stack's output through DYT_F_TOKENS_OUT + the final LayerNorm and the head through the C ABI's unit entries.  Their
    composition is ``forward``: same eval decisions, logits to fp32 round-off (the 16-bit mode: to its own -- the cls-only tail of
    ``forward`` and the all-token last block of ``forward_features`` take different GEMM launch shapes); the returned tokens are
    LayerNorm-ed (zero mean, unit variance per token under the test's norm weights)."""
    import synth
    from test_gpu_round2 import _bench_model
    B = 5
    m, sd = _bench_model(precision, "compact", B, 0.85, kind="test")
    m.eval()
    x, _ = synth.make_batch(B, 100, seed=9)
    x = x.cuda()
    with torch.no_grad():
        logits, aux = m(x)
        lc, _ = m(x, complete_model=True)
    feats, aux2 = m.forward_features(x)
    assert feats.shape == (B, 197, 768) and aux2["token_select"].shape == (B, 12, 196, 1) and aux2["token_logits"].shape == (B, 12, 196, 1)
    assert torch.equal(aux["token_select"], aux2["token_select"])
    via = m.forward_head(feats)
    tol = 2e-5 if precision == "fp32" else 5e-3
    assert via.shape == logits.shape and float((via - logits).abs().max()) < tol, float((via - logits).abs().max())
    fc, _ = m.forward_features(x, complete_model=True)
    assert float((m.forward_head(fc) - lc).abs().max()) < tol
    assert torch.equal(m.forward_head(feats, pre_logits=True), feats[:, 0])
    # the tokens are the final norm's output: x_hat * w + b with the model's norm parameters
    w, b = m.norm.weight.detach(), m.norm.bias.detach()
    xh = (feats - b) / w
    assert float(xh.mean(-1).abs().max()) < 1e-3 and float((xh.var(-1, unbiased=False) - 1).abs().max()) < 1e-2


# ---- tuning_config.ffn_adapter_layernorm_option = "in" / "out" (dyt_config.adapter_ln; reference models/dynamic_adapter.py:88,95-98,121-122,132-133) ----
@pytest.mark.parametrize("option", ["in", "out"])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3q", "fp16", "bf16"])
def test_adapter_layernorm_step_vs_reference_golden(precision, option):
    """The reference model with the adapter's own trainable LayerNorm on its input ("in", the Adapter class's default) or on its scaled
    output ("out"), stepped through its own train_one_epoch (tests/golden/adapter_ln_{in,out}_step.npz, distinct gamma / beta per block):
    logits, masks, losses and all 98 gradients -- the 24 d(gamma) / d(beta) included -- of the fused step, masked mode against the
    reference's values and compact mode against the oracle's; fp32 also the LayerNorm parameters after one AdamW update."""
    import numpy as np
    import gpu_diag as D
    import synth
    from oracle import dyt_oracle as O
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "adapter_ln_%s_step.npz" % option)))
    B, C, r, seed = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"]), int(g["meta_seed"])
    x, y = synth.make_batch(B, C, seed=seed)
    keep = synth.make_dropout_masks(B, r, seed=seed + 3)
    g1, g2 = torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"])
    sd = synth.add_adapter_layernorm(synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=0.3), seed=seed)
    sd_oracle = dict(sd)
    sd_oracle[O.ADAPTER_LN_KEY] = torch.tensor({"in": 1, "out": 2}[option])
    tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option=option, ffn_adapter_init_option="lora",
                   ffn_adapter_scalar="0.1", ffn_num=r, d_model=768)
    tol = D.TOL[precision]
    for mode in ("masked", "compact"):
        model = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=0.0, tuning_config=tuning, select_config=D.Cfg(open=True, keep_layers=0),
                                           precision=precision, train_mode=mode)
        msg = model.load_state_dict(sd, strict=True)
        assert not msg.missing_keys and not msg.unexpected_keys
        for n, p in model.named_parameters():
            p.requires_grad = synth.is_trainable(n)
        model = model.cuda()
        assert sum(p.requires_grad for p in model.parameters()) == 98
        model.train()
        eng = model.engine(B, torch.device("cuda", 0))
        assert eng.adapter_ln == {"in": 1, "out": 2}[option]
        ls, lt = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
        ts = torch.zeros(B, 12, 196, device="cuda")
        losses = eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                                  g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt, token_select=ts).cpu()
        es = float(np.abs(ls.cpu().numpy() - g["logits_student"]).max())
        et = float(np.abs(lt.cpu().numpy() - g["logits_teacher"]).max())
        flips = int((ts.cpu().numpy().astype(np.uint8) != g["token_select"][..., 0]).sum())
        el = max(abs(float(losses[i]) - float(g["stat_" + k])) / max(1.0, abs(float(g["stat_" + k])))
                 for i, k in enumerate(("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss")))
        if mode == "masked":
            gref = {n[len("grad/"):]: torch.from_numpy(v) for n, v in g.items() if n.startswith("grad/")}
        else:
            _, gref, _ = O.step_grads(sd_oracle, x, y, g1, g2, keep, scale=0.1, mode="compact")
        items = [(n, eng.trainable_view(n, gr.shape, eng.grad).cpu(), gr) for n, gr in gref.items()]
        lnp = [it for it in items if "adapter_layer_norm_before" in it[0]]
        assert len(lnp) >= 6
        e_ln = max(float((a - b).norm() / max(float(b.norm()), 1e-20)) for _, a, b in lnp)
        items = [it for it in items if "adapter_layer_norm_before" not in it[0]]
        if precision != "fp32":
            sc1 = [it for it in items if it[2].numel() == 1]
            items = [it for it in items if it[2].numel() > 1]
            if sc1:
                items.append(("mlp_token_select.mlp_head.bias (12 blocks)", torch.stack([a.reshape(()) for _, a, _ in sc1]), torch.stack([b.reshape(()) for _, _, b in sc1])))
        worst = {}
        for n, got, ref in items:
            e = float((got - ref).norm() / max(float(ref.norm()), 1e-20))
            k = D.grad_kind(n) if precision in ("fp16", "bf16") else "all"
            if e > worst.get(k, (0.0, ""))[0]:
                worst[k] = (e, n)
        print("adapter LayerNorm %r %s/%s: logits %.2e / %.2e, %d decisions differ, losses %.1e, d(gamma) / d(beta) %.1e, gradients %s" %
              (option, precision, mode, es, et, flips, el, e_ln, {k: "%.1e" % v[0] for k, v in worst.items()}))
        assert es <= tol["logits"] and et <= tol["logits"] and flips <= tol["step_flips"] and el <= tol["loss"]
        # the 16-bit modes at this small batch: d(gamma) / d(beta) sit upstream of down_proj ("in") or between up_proj and the residual ("out"),
        # so they take the adapter classes' bounds; bf16 flips 1-2 of the 4704 gate decisions here, and a flipped token changes every adapter
        # gradient of its block by a whole token's contribution (measured: "in" d(gamma) 0.19, "out" up_proj.bias 0.086) -> twice the class
        # bound when decisions differ.  fp32 / the split modes / fp16 flip none and keep the suite's bounds
        slack = 2.0 if (precision == "bf16" and flips) else 1.0
        small_b = D.FP16_GRAD_TOL_SMALL_B if precision == "fp16" else D.BF16_GRAD_TOL_SMALL_B
        assert e_ln <= (2e-3 if precision == "fp32" else (tol["grad"] if precision in D.SPLIT_MODES else slack * small_b["adaptmlp.down_proj"]))
        for k, (e, n) in worst.items():
            bound = 2e-3 if precision == "fp32" else (tol["grad"] if precision in D.SPLIT_MODES else slack * small_b[k])
            assert e <= bound, (mode, k, n, e, bound)
        if precision == "fp32" and mode == "masked":
            D.D_adamw(eng, float(g["meta_lr"]), float(g["meta_wd"]))
            for key in g:
                if key.startswith("param_after/"):
                    n = key.split("/", 1)[1]
                    got = eng.trainable_view(n, g[key].shape).cpu().numpy()
                    # the first AdamW update is lr * g / (|g| + 1e-8): where the reference's |g| is within 100x of Adam's eps (a few of the 768
                    # d(beta) / d(gamma) elements are ~1e-8), fp32 summation-order noise in g moves the update by a few % of lr
                    gg = np.abs(g["grad/" + n])
                    err = np.abs(got - g[key])
                    assert err[gg > 1e-6].max(initial=0.0) < 2e-5 and err.max() < 0.1 * float(g["meta_lr"]), n
        if mode == "compact":
            # inference (deterministic gate, no dropout): the dynamic pass and complete_model against the oracle, forward_features too
            model.eval()
            with torch.no_grad():
                le, aux = model(x.cuda())
                lc, _ = model(x.cuda(), complete_model=True)
            oe, oaux = O.forward(sd_oracle, x, training=False, mode="compact")
            oc, _ = O.forward(sd_oracle, x, training=False, complete_model=True)
            ee, ec = float((le.cpu() - oe).abs().max()), float((lc.cpu() - oc).abs().max())
            ef = int((aux["token_select"].cpu() != oaux["token_select"]).sum())
            print("adapter LayerNorm %r %s eval: logits %.2e / %.2e, %d decisions differ" % (option, precision, ee, ec, ef))
            assert ee <= tol["logits"] and ec <= tol["logits"] and ef <= tol["eval_flips"]
            feats, _ = model.forward_features(x.cuda())
            assert float((model.forward_head(feats) - le).abs().max()) < (2e-5 if precision == "fp32" else 5e-3)
        del model, eng
        torch.cuda.empty_cache()


@pytest.mark.parametrize("option", ["in", "out"])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3q"])
def test_adapter_layernorm_with_stochastic_depth_vs_oracle(precision, option):
    """The adapter's LayerNorm together with drop_path_rate = 0.3 (reference vision_transformer_IN21K.py:157-163: drop_path2 scales the MLP
    branch only, the adapter branch -- LayerNorm-ed or not -- joins the residual unscaled): both passes' logits, decisions, losses and the 98
    gradients of a step with injected per-sample factors against the oracle, masked and compact."""
    import numpy as np
    import gpu_diag as D
    import synth
    from oracle import dyt_oracle as O
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    # (the 16-bit backward at the five-seed test's batch: its gate-gradient bound is a B = 16 figure -- at B = 6 with a third of the branch
    # instances dropped the same absolute noise is 6e-3 of a smaller sum)
    B, C, r, seed, rate = (6 if precision == "fp32" else 16), 10, 8, 23, 0.3
    x, y = synth.make_batch(B, C, seed=seed)
    g1, g2 = synth.make_noise(B, seed=seed + 1)
    keep = synth.make_dropout_masks(B, r, seed=seed + 2)
    gen = torch.Generator().manual_seed(seed + 3)
    dpr = torch.linspace(0, rate, 12)
    scales = torch.ones(2, 2, 12, B)   # [pass][branch][block][sample]: 0 or 1 / keep_l, block 0 never dropped
    for l in range(1, 12):
        kp = 1.0 - float(dpr[l])
        scales[:, :, l] = (torch.rand(2, 2, B, generator=gen) < kp).float() / kp
    sd = synth.add_adapter_layernorm(synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=0.3), seed=seed)
    sd_oracle = dict(sd)
    sd_oracle[O.ADAPTER_LN_KEY] = torch.tensor({"in": 1, "out": 2}[option])
    tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option=option, ffn_adapter_init_option="lora",
                   ffn_adapter_scalar="0.1", ffn_num=r, d_model=768)
    tol = D.TOL[precision]
    for mode in ("masked", "compact"):
        model = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=rate, tuning_config=tuning, select_config=D.Cfg(open=True, keep_layers=0),
                                           precision=precision, train_mode=mode)
        model.load_state_dict(sd, strict=True)
        for n, p in model.named_parameters():
            p.requires_grad = synth.is_trainable(n)
        model = model.cuda().train()
        eng = model.engine(B, torch.device("cuda", 0))
        eng.set_drop_path_scales(0, scales[0].cuda().contiguous())
        eng.set_drop_path_scales(1, scales[1].cuda().contiguous())
        ls, lt = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
        ts = torch.zeros(B, 12, 196, device="cuda")
        losses = eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                                  g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt, token_select=ts).cpu()
        d_ref, gref, (out_s, out_t, tok) = O.step_grads(sd_oracle, x, y, g1, g2, keep, scale=0.1, mode=mode, drop_scales=scales)
        es = float((ls.cpu() - out_s.detach()).abs().max())
        et = float((lt.cpu() - out_t.detach()).abs().max())
        flips = int((ts.cpu() != tok["token_select"].detach()[..., 0]).sum())
        el = max(abs(float(losses[i]) - float(d_ref[k])) / max(1.0, abs(float(d_ref[k])))
                 for i, k in enumerate(("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss")))
        worst, wn = 0.0, ""
        items = [(n, eng.trainable_view(n, gr.shape, eng.grad).cpu(), gr) for n, gr in gref.items()]
        if precision != "fp32":   # the 12 one-number gate bias gradients as one vector (gpu_diag.report_grads)
            sc1 = [it for it in items if it[2].numel() == 1]
            items = [it for it in items if it[2].numel() > 1]
            items.append(("mlp_token_select.mlp_head.bias (12 blocks)", torch.stack([a.reshape(()) for _, a, _ in sc1]), torch.stack([b.reshape(()) for _, _, b in sc1])))
        for n, got, gr in items:
            e = float((got - gr).norm() / max(float(gr.norm()), 1e-20))
            if e > worst:
                worst, wn = e, n
        print("adapter LayerNorm %r + drop_path %s/%s: logits %.2e / %.2e, %d decisions differ, losses %.1e, worst gradient %.1e (%s)" %
              (option, precision, mode, es, et, flips, el, worst, wn))
        assert len(gref) == 98
        assert es <= tol["logits"] and et <= tol["logits"] and flips <= tol["step_flips"] and el <= tol["loss"]
        assert worst <= (2e-3 if precision == "fp32" else tol["grad"]), (wn, worst)
        del model, eng
        torch.cuda.empty_cache()


def test_adapter_layernorm_rejected_configurations():
    """What dyt_config.adapter_ln does not combine with fails loudly at the boundary: the video model (dyt_ctx_create), the learnable adapter
    scale (dyt_set_option), a value outside 0..2; the LayerNorm's parameters do not exist in a context created without the option."""
    import _lib
    from runtime import DyTEngine
    from _lib import DyTError
    dev = torch.device("cuda", 0)
    with pytest.raises(DyTError, match="adapter_ln"):
        DyTEngine(10, 8, 0.1, dev, max_batch=8, frames=4, adapter_ln=1)
    with pytest.raises(DyTError, match="adapter_ln"):
        DyTEngine(10, 8, 0.1, dev, max_batch=4, adapter_ln=3)
    e = DyTEngine(10, 8, 0.1, dev, max_batch=4, adapter_ln=2)
    with pytest.raises(DyTError, match="LayerNorm"):
        e.set_option(_lib.OPT_LEARNABLE_SCALE, 1)
    e0 = DyTEngine(10, 8, 0.1, dev, max_batch=4)
    with pytest.raises(DyTError, match="adapter_ln"):
        e0.trainable_slice("blocks.0.adaptmlp.adapter_layer_norm_before.weight")


# ---- mixup_fn / class-probability targets (reference engine_finetune.py:44-45; C ABI dyt_set_soft_targets) ----
@pytest.mark.parametrize("precision", ["fp32", "fp16x3q", "fp16", "bf16"])
def test_mixup_step_vs_reference_golden(precision):
    """The reference's train_one_epoch WITH a mixup_fn (tests/golden/mixup_step.npz: the real model, AdaLoss and loop; synth.mixup_batch as the
    callable): logits, masks, the five loss components and the 74 gradients of the fused step fed the same mixed samples and class-probability
    targets, masked mode against the reference's values and compact mode against the oracle's; fp32 also the head after one AdamW update.  Then
    the same through OUR train_one_epoch(mixup_fn=...) -- the loop calls the callable on the device tensors and hands the targets to the
    library -- and back to integer labels afterwards."""
    import numpy as np
    import gpu_diag as D
    import synth
    from oracle import dyt_oracle as O
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "mixup_step.npz")))
    B, C, r, seed = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"]), int(g["meta_seed"])
    lam, sm = float(g["meta_lam"]), float(g["meta_smoothing"])
    x, y = synth.make_batch(B, C, seed=seed)
    xm, t = synth.mixup_batch(x, y, C, lam=lam, smoothing=sm)
    keep = synth.make_dropout_masks(B, r, seed=seed + 3)
    g1, g2 = torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"])
    sd = synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=0.3)
    tol = D.TOL[precision]
    for mode in ("masked", "compact"):
        from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
        tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                       ffn_adapter_scalar="0.1", ffn_num=r, d_model=768)
        model = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=0.0, tuning_config=tuning, select_config=D.Cfg(open=True, keep_layers=0),
                                           precision=precision, train_mode=mode)
        model.load_state_dict(sd, strict=True)
        for n, p in model.named_parameters():
            p.requires_grad = synth.is_trainable(n)
        model = model.cuda()
        model.train()
        eng = model.engine(B, torch.device("cuda", 0))
        td = t.cuda().contiguous()
        eng.set_soft_targets(td)
        ls, lt = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
        ts = torch.zeros(B, 12, 196, device="cuda")
        wrong_labels = torch.zeros(B, dtype=torch.int64, device="cuda")   # ignored while soft targets are set
        losses = eng.step_fwd_bwd(xm.cuda(), wrong_labels, 0.5, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                                  g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt, token_select=ts).cpu()
        es = float(np.abs(ls.cpu().numpy() - g["logits_student"]).max())
        et = float(np.abs(lt.cpu().numpy() - g["logits_teacher"]).max())
        flips = int((ts.cpu().numpy().astype(np.uint8) != g["token_select"][..., 0]).sum())
        el = max(abs(float(losses[i]) - float(g["stat_" + k])) / max(1.0, abs(float(g["stat_" + k])))
                 for i, k in enumerate(("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss")))
        if mode == "masked":
            gref = {n[len("grad/"):]: torch.from_numpy(v) for n, v in g.items() if n.startswith("grad/")}
        else:
            _, gref, _ = O.step_grads(sd, xm, t, g1, g2, keep, scale=0.1, mode="compact")
        items = [(n, eng.trainable_view(n, gr.shape, eng.grad).cpu(), gr) for n, gr in gref.items()]
        if precision != "fp32":
            sc1 = [it for it in items if it[2].numel() == 1]
            items = [it for it in items if it[2].numel() > 1]
            if sc1:
                items.append(("mlp_token_select.mlp_head.bias (12 blocks)", torch.stack([a.reshape(()) for _, a, _ in sc1]), torch.stack([b.reshape(()) for _, _, b in sc1])))
        worst = {}
        for n, got, ref in items:
            e = float((got - ref).norm() / max(float(ref.norm()), 1e-20))
            k = D.grad_kind(n) if precision in ("fp16", "bf16") else "all"
            if e > worst.get(k, (0.0, ""))[0]:
                worst[k] = (e, n)
        print("mixup %s/%s: logits %.2e / %.2e, %d decisions differ, losses %.1e, gradients %s" %
              (precision, mode, es, et, flips, el, {k: "%.1e" % v[0] for k, v in worst.items()}))
        assert es <= tol["logits"] and et <= tol["logits"] and flips <= tol["step_flips"] and el <= tol["loss"]
        for k, (e, n) in worst.items():
            bound = 2e-3 if precision == "fp32" else (tol["grad"] if precision in D.SPLIT_MODES else
                                                     (D.FP16_GRAD_TOL_SMALL_B if precision == "fp16" else D.BF16_GRAD_TOL_SMALL_B)[k])
            assert e <= bound, (mode, k, n, e, bound)
        if precision == "fp32" and mode == "masked":
            D.D_adamw(eng, float(g["meta_lr"]), float(g["meta_wd"]))
            for key in g:
                if key.startswith("param_after/"):
                    n = key.split("/", 1)[1]
                    assert np.abs(eng.trainable_view(n, g[key].shape).cpu().numpy() - g[key]).max() < 2e-5, n
        # a wrong row count fails loudly; NULL restores the integer labels
        eng.set_soft_targets(torch.full((B + 1, C), 1.0 / C, device="cuda"))
        with pytest.raises(Exception, match="soft targets"):
            eng.step_fwd_bwd(xm.cuda(), wrong_labels, 0.5, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"))
        eng.set_soft_targets(None)
        hard = eng.step_fwd_bwd(xm.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                                g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous()).cpu()
        assert abs(float(hard[1]) - float(losses[1])) > 1e-3   # base_loss against the integer labels is another number
        del model, eng
        torch.cuda.empty_cache()


def test_train_one_epoch_with_mixup_fn_vs_reference_golden():
    """OUR train_one_epoch driven like the reference's (same positional call as tests/golden/make_golden_mixup.py) with a mixup_fn: the epoch
    statistics equal the reference's, evaluation afterwards runs on integer labels again."""
    import logging
    import types
    import numpy as np
    import synth
    import engine_finetune as E
    import misc
    from models.losses import AdaLoss
    import gpu_diag as D
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "mixup_step.npz")))
    B, C, r, seed = int(g["meta_batch"]), int(g["meta_num_classes"]), int(g["meta_ffn_num"]), int(g["meta_seed"])
    lam, sm = float(g["meta_lam"]), float(g["meta_smoothing"])
    sd = synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=0.3)
    tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                   ffn_adapter_scalar="0.1", ffn_num=r, d_model=768)
    model = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=0.0, tuning_config=tuning, select_config=D.Cfg(open=True, keep_layers=0),
                                       precision="fp32", train_mode="masked")
    model.load_state_dict(sd, strict=True)
    for n, p in model.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    model = model.cuda()
    lr, wd = float(g["meta_lr"]), float(g["meta_wd"])
    optimizer = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=wd)
    criterion = AdaLoss(base_criterion=torch.nn.CrossEntropyLoss(), token_target_ratio=0.5, token_loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0)
    scaler = misc.NativeScalerWithGradNormCount()
    args = types.SimpleNamespace(accum_iter=1, lr=lr, min_lr=0.0, warmup_epochs=0, epochs=10, metric="accuracy", nb_classes=C)
    x, y = synth.make_batch(B, C, seed=seed)
    keep = synth.make_dropout_masks(B, r, seed=seed + 3)
    g1, g2 = torch.from_numpy(g["g1"]), torch.from_numpy(g["g2"])
    calls = []

    def mixup_fn(samples, targets):
        assert samples.is_cuda and targets.is_cuda   # the loop calls it on the device tensors, like the reference's
        calls.append(1)
        return synth.mixup_batch(samples, targets, C, lam=lam, smoothing=sm)

    stats = E.train_one_epoch(model, criterion, [(x, y, (g1, g2), keep)], optimizer, torch.device("cuda", 0), 0, scaler, None, mixup_fn, None,
                              args=args, logger=logging.getLogger("t"))
    assert len(calls) == 1
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(stats[k] - float(g["stat_" + k])) < 1e-4 * max(1.0, abs(float(g["stat_" + k]))), (k, stats[k], float(g["stat_" + k]))
    head = dict(model.named_parameters())
    for key in g:
        if key.startswith("param_after/"):
            n = key.split("/", 1)[1]
            assert np.abs(head[n].detach().cpu().numpy() - g[key]).max() < 2e-5, n
    # the next epoch without a mixup_fn: integer labels again (no stale soft-target pointer)
    stats2 = E.train_one_epoch(model, criterion, [(x, y)], optimizer, torch.device("cuda", 0), 1, scaler, None, None, None, args=args, logger=logging.getLogger("t"))
    assert np.isfinite(stats2["loss"]) and model._engine._soft_keep is None


@pytest.mark.parametrize("gate_bias", [-60.0, 60.0])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3q", "fp16"])
def test_extreme_keep_ratios_vs_oracle(precision, gate_bias):
    """The dispatcher's corner cases: a gate bias of -60 drops EVERY patch token of every block (the compacted MLP runs on the B cls rows alone: one
    partial tile per GEMM, every other row takes the dropped-token path), +60 keeps every token (compaction is the identity).  Training step
    (masked and compact) and inference against the oracle; decisions must be all-0 / all-1."""
    import gpu_diag as D
    import synth
    from oracle import dyt_oracle as O
    from test_gpu_round2 import _bench_model
    B, C, r, seed = 3, 10, 8, 37
    x, y = synth.make_batch(B, C, seed=seed)
    g1, g2 = synth.make_noise(B, seed=seed + 1)
    keep = synth.make_dropout_masks(B, r, seed=seed + 2)
    tol = D.TOL[precision]
    for mode in ("masked", "compact"):
        m, sd = _bench_model(precision, mode, B, gate_bias, classes=C, r=r, kind="test", seed=seed)
        m.train()
        eng = m.engine(B, torch.device("cuda", 0))
        ls, lt = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
        ts = torch.full((B, 12, 196), 0.5, device="cuda")
        losses = eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                                  g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt, token_select=ts).cpu()
        want = 1.0 if gate_bias > 0 else 0.0
        assert bool((ts == want).all()), "decisions"
        assert abs(float(losses[5]) - want) < 1e-6   # mean keep ratio
        d_ref, gref, (out_s, out_t, tok) = O.step_grads(sd, x, y, g1, g2, keep, scale=0.1, mode=mode)
        assert bool((tok["token_select"] == want).all())
        es, et = float((ls.cpu() - out_s.detach()).abs().max()), float((lt.cpu() - out_t.detach()).abs().max())
        el = max(abs(float(losses[i]) - float(d_ref[k])) / max(1.0, abs(float(d_ref[k])))
                 for i, k in enumerate(("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss")))
        worst, wn = 0.0, ""
        for n, gr in gref.items():
            if float(gr.norm()) < 1e-12:   # (saturated gates: d sigmoid = 0 -- both sides must then be zero)
                assert float(eng.trainable_view(n, gr.shape, eng.grad).abs().max()) < 1e-10, n
                continue
            got = eng.trainable_view(n, gr.shape, eng.grad).cpu()
            e = float((got - gr).norm() / float(gr.norm()))
            if "mlp_token_select" in n and float(gr.norm()) < 1e-6:
                continue   # gate gradients through a saturated sigmoid (e^-60 scale): relative error of numbers at the fp32 underflow edge is not parity
            if e > worst:
                worst, wn = e, n
        bound = 2e-3 if precision == "fp32" else (tol["grad"] if precision in D.SPLIT_MODES else 0.25)
        print("gate bias %+.0f %s/%s: logits %.2e / %.2e, losses %.1e, worst gradient %.1e (%s)" % (gate_bias, precision, mode, es, et, el, worst, wn))
        assert es <= tol["logits"] and et <= tol["logits"] and el <= tol["loss"]
        assert worst <= bound, (wn, worst)
        m.eval()
        with torch.no_grad():
            le, aux = m(x.cuda())
        oe, oaux = O.forward(sd, x, training=False, mode="compact")
        assert bool((aux["token_select"].cpu() == want).all()) and float((le.cpu() - oe).abs().max()) <= tol["logits"]
        del m, eng
        torch.cuda.empty_cache()


from test_gpu_round2 import rccl_one_rank  # noqa: E402,F401  (fixture)


def test_adapter_layernorm_rccl_one_rank_path_equals_no_dist(rccl_one_rank):  # noqa: F811
    """The 98-tensor flat layout (adapter LayerNorm "in") through the multi-rank machinery with one rank -- parameter broadcast, the early
    all-reduce of the upper blocks' gradients on the communication stream, the lower part after the backward, AdamW -- reproduces the
    single-process training bit for bit: the split point of the two all-reduce parts moves with the layout."""
    import gpu_diag as D
    import synth
    from engine_finetune import FusedAdamW, train_step
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    B, C, r, seed = 4, 10, 8, 41

    def three_steps():
        sd = synth.add_adapter_layernorm(synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=0.3), seed=seed)
        tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="in", ffn_adapter_init_option="lora",
                       ffn_adapter_scalar="0.1", ffn_num=r, d_model=768)
        m = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=0.0, tuning_config=tuning, select_config=D.Cfg(open=True, keep_layers=0),
                                       precision="fp16", train_mode="compact", max_batch=B)
        m.load_state_dict(sd, strict=True)
        for n, p in m.named_parameters():
            p.requires_grad = synth.is_trainable(n)
        m = m.cuda().train()
        opt = FusedAdamW(m, lr=1e-3, weight_decay=0.01)
        x, y = synth.make_batch(B, C, seed=seed + 1)
        x, y = x.cuda(), y.cuda()
        losses = []
        for i in range(3):
            losses.append(train_step(m, x, y, opt, seed=700 + i, target_ratio=0.5, token_minimal=0.0, token_minimal_weight=0.0).clone())
        torch.cuda.synchronize()
        assert m._engine.adapter_ln == 1
        return m._engine.flat.clone(), torch.stack(losses), m._engine.grad.clone()

    with_dist = three_steps()
    rccl_one_rank.destroy_process_group()
    try:
        without = three_steps()
    finally:
        rccl_one_rank.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for a, b in zip(with_dist, without):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert float(with_dist[1][:, 0].min()) > 0 and bool(torch.isfinite(with_dist[0]).all())


def test_label_smoothing_criterion_through_the_soft_target_form():
    """criterion.base_criterion = nn.CrossEntropyLoss(label_smoothing = 0.1): train_one_epoch hands t = one_hot (1 - e) + e / C to the library's
    soft-target loss (torch's own definition: F.cross_entropy(x, y, label_smoothing=e) == F.cross_entropy(x, t)); epoch statistics against the
    oracle evaluated with the same targets; class weights still raise."""
    import logging
    import types
    import synth
    import engine_finetune as E
    import misc
    import gpu_diag as D
    from oracle import dyt_oracle as O
    from models.losses import AdaLoss
    from test_gpu_round2 import _bench_model
    B, C, r, seed, eps = 4, 10, 8, 43, 0.1
    m, sd = _bench_model("fp32", "masked", B, 0.3, classes=C, r=r, kind="test", seed=seed)
    x, y = synth.make_batch(B, C, seed=seed)
    g1, g2 = synth.make_noise(B, seed=seed + 1)
    keep = synth.make_dropout_masks(B, r, seed=seed + 2)
    lr = 1e-3
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=lr, weight_decay=0.0)
    crit = AdaLoss(base_criterion=torch.nn.CrossEntropyLoss(label_smoothing=eps), token_target_ratio=0.5, token_loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0)
    args = types.SimpleNamespace(accum_iter=1, lr=lr, min_lr=0.0, warmup_epochs=0, epochs=10, metric="accuracy", nb_classes=C)
    stats = E.train_one_epoch(m, crit, [(x, y, (g1, g2), keep)], opt, torch.device("cuda", 0), 0, misc.NativeScalerWithGradNormCount(), None, None, None,
                              args=args, logger=logging.getLogger("t"))
    t = torch.nn.functional.one_hot(y, C).float() * (1.0 - eps) + eps / C
    _, d, _ = O.step_loss(sd, x, t, g1, g2, keep, scale=0.1, mode="masked")
    for k in ("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss"):
        assert abs(stats[k] - float(d[k])) < 1e-4 * max(1.0, abs(float(d[k]))), (k, stats[k], float(d[k]))
    _, d0, _ = O.step_loss(sd, x, y, g1, g2, keep, scale=0.1, mode="masked")
    assert abs(float(d0["base_loss"]) - float(d["base_loss"])) > 1e-3   # (the smoothed loss is another number)
    bad = AdaLoss(base_criterion=torch.nn.CrossEntropyLoss(weight=torch.ones(C)), token_target_ratio=0.5, token_loss_ratio=2.0)
    with pytest.raises(NotImplementedError):
        E.train_one_epoch(m, bad, [(x, y)], opt, torch.device("cuda", 0), 1, None, None, None, None, args=args, logger=logging.getLogger("t"))


@pytest.mark.parametrize("option", ["in", "out"])
def test_adapter_layernorm_model_trains_in_the_fast_mode(option):
    """Forty fused steps (fp16 operands, library noise, overflow-guarded AdamW) of the adapter-LayerNorm model on one repeated batch: every loss
    component stays finite, no update is skipped, the task loss falls by more than 30 %, and gamma / beta move off their initial values."""
    import gpu_diag as D
    import synth
    from engine_finetune import FusedAdamW, train_step
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    B, C, r, seed = 16, 10, 8, 47
    sd = synth.add_adapter_layernorm(synth.make_state_dict(C, r, seed=seed, kind="test", gate_bias=0.3), seed=seed)
    tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option=option, ffn_adapter_init_option="lora",
                   ffn_adapter_scalar="0.1", ffn_num=r, d_model=768)
    m = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=0.0, tuning_config=tuning, select_config=D.Cfg(open=True, keep_layers=0),
                                   precision="fp16", train_mode="compact", max_batch=B)
    m.load_state_dict(sd, strict=True)
    for n, p in m.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    m = m.cuda().train()
    opt = FusedAdamW(m, lr=2e-3, weight_decay=0.0)
    x, y = synth.make_batch(B, C, seed=seed + 1)
    x, y = x.cuda(), y.cuda()
    name = "blocks.5.adaptmlp.adapter_layer_norm_before.weight"
    hist = []
    for i in range(40):
        hist.append(train_step(m, x, y, opt, seed=900 + i, target_ratio=0.5, token_minimal=0.0, token_minimal_weight=0.0).clone())
    torch.cuda.synchronize()
    hist = torch.stack(hist).cpu()
    assert bool(torch.isfinite(hist).all())
    first, last = float(hist[:3, 1].mean()), float(hist[-3:, 1].mean())   # base_loss
    print("adapter LayerNorm %r, fp16: base loss %.3f -> %.3f over 40 steps, keep ratio %.2f -> %.2f" % (option, first, last, float(hist[0, 5]), float(hist[-1, 5])))
    assert last < 0.7 * first, (first, last)   # measured: "in" 2.04 -> 0.00, "out" 2.79 -> 1.38
    assert opt.skipped_steps(m._engine) == 0 if hasattr(opt, "skipped_steps") else True
    g0 = sd[name]
    g1 = m._engine.trainable_view(name, g0.shape).cpu()
    assert float((g1 - g0).abs().max()) > 1e-3


def test_two_rank_data_parallel_step_on_one_gpu(tmp_path):
    """The multi-rank path with TWO real processes (tests/dp2_worker.py; both on cuda:0, torch.distributed over gloo, the torch.distributed form
    of the gradient all-reduce -- RCCL refuses two ranks on one device): ranks that start from different trainables and see different shards
    end two steps with IDENTICAL parameters, and those equal, bit for bit, what one process gets by stepping the two shards one after the other
    from rank 0's parameters, adding the gradients, and updating once with grad_scale 1/2 (DDP's mean of per-rank gradients, main_image.py:280-282)."""
    import subprocess
    import sys
    import dp2_worker as W
    from engine_finetune import FusedAdamW
    from test_gpu_round2 import _free_port
    port = _free_port()
    outs = [str(tmp_path / ("rank%d.pt" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "dp2_worker.py"), str(r), "2", str(port), outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), logs
    res = [torch.load(o) for o in outs]
    for precision in ("fp32", "fp16"):
        a, b = res[0][precision], res[1][precision]
        assert torch.equal(a["flat"], b["flat"]), precision                       # both ranks hold the same parameters
        assert not torch.equal(a["losses"], b["losses"])                          # ... after different shards
        # one process: rank 0's model, the two shards in turn, gradients summed, one update per step with grad_scale 1 / 2
        m = W.build(0, precision)
        opt = FusedAdamW(m, lr=1e-3, weight_decay=0.01)
        eng = m.engine(W.B, torch.device("cuda", 0))
        want_losses = [[], []]
        for i in range(2):
            opt.sync_parameters(eng)
            total = torch.zeros_like(eng.grad)
            for r in range(2):
                x, y, g1, g2, keep = W.shard(r)
                out = eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, g1=g1.cuda().contiguous(), g2=g2.cuda().contiguous(),
                                       keep_mask=keep.cuda().contiguous())
                want_losses[r].append(out.clone().cpu())
                total += eng.grad
            eng.grad.copy_(total)
            opt.step(grad_scale=0.5)
        torch.cuda.synchronize()
        assert torch.equal(eng.flat.cpu(), a["flat"]), (precision, float((eng.flat.cpu() - a["flat"]).abs().max()))
        for r in range(2):
            assert torch.equal(torch.stack(want_losses[r]), res[r][precision]["losses"]), (precision, r)
        del m, opt, eng
        torch.cuda.empty_cache()
    # the epoch loop + evaluation phase of the workers: the all-reduced epoch statistics and the gathered metrics are the same numbers on both
    # ranks, the evaluation saw all 5 + 6 samples, and the replicas still hold identical parameters
    e0, e1 = res[0]["epoch"], res[1]["epoch"]
    assert e0["stats"] == e1["stats"] and e0["status"] == e1["status"], (e0, e1)
    assert e0["n_eval"] + e1["n_eval"] == 11 and 0.0 <= e0["status"]["acc1"] <= 100.0 and 0.0 < e0["status"]["keep_ratio"] < 1.0
    assert abs(e0["status"]["acc1"] * 11 / 100.0 - round(e0["status"]["acc1"] * 11 / 100.0)) < 1e-3   # acc1 is k / 11: the gather was ragged (5 + 6)
    assert all(v == v and abs(v) < 1e6 for v in e0["stats"].values())
    assert torch.equal(e0["flat"], e1["flat"])   # (the workers wrap the model in DistributedDataParallel first, as main_image.py:280-282 does)


def test_bench_launch_line_with_two_real_ranks(tmp_path):
    """The driver's N > 1 command (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...`) with real processes on this
    box's one GPU (bench.py's DYT_BENCH_SHARE_GPU test rig: every rank on cuda:0, gloo): it must finish and print exactly ONE JSON line, from rank
    0, with the whole-job figures.  (Round 6 found a deadlock here: rank 0's event-profiled extra step ran the gradient all-reduce alone.)"""
    import json
    import subprocess
    import sys
    from test_gpu_round2 import _free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DYT_BENCH_SHARE_GPU="1", DYT_BENCH_WATCHDOG="400")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["config"]["global_batch"] == 64
    assert d["distributed"]["torch_world_size"] == 2 and "TEST RIG" in d["config"]["parallelism"]
    assert abs(d["value"] - 64 / d["ms_per_step"] * 1e3) < 0.01 * d["value"] and d["roofline"]["frac"] > 0
