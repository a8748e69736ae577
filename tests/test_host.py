"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/dyt_hip.h
declares (no compute calls without a GPU), the module mirror keeps the reference's parameter
surface, and the small host-side mirrors (AdaLoss, lr schedule, metrics) equal the oracle."""
import os
import re
import types

import pytest
import torch

import synth
from oracle import dyt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Cfg(dict):
    def __getattr__(self, k):   # AttributeError (not KeyError) for a missing key: copy / pickle probe attributes
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _model(**kw):
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    tuning = Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                 ffn_adapter_scalar="0.1", ffn_num=kw.pop("ffn_num", 64), d_model=768)
    return vit_base_patch16_224_in21k(num_classes=kw.pop("num_classes", 100), drop_path_rate=0.0, tuning_config=tuning,
                                      select_config=Cfg(open=True, keep_layers=0), **kw)


def test_library_exports_every_declared_symbol():
    import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "dyt_hip.h")).read()
    declared = set(re.findall(r"\b(dyt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.dyt_version() >= 1


def test_ctx_create_fails_loudly_without_gpu():
    import ctypes
    import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = _lib.Config(100, 64, 12, 1, 2, 2, 0.1, 0.1, 5.0, 0.5)
    h = ctypes.c_void_p()
    assert _lib.lib().dyt_ctx_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert _lib.lib().dyt_last_error()


def test_module_keeps_reference_parameter_surface():
    m = _model()
    sd = m.state_dict()
    ref = synth.param_shapes(100, 64)
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    assert sum(p.numel() for p in m.parameters()) == 87074416
    assert sum(p.numel() for n, p in m.named_parameters() if synth.is_trainable(n)) == 1275760
    msg = m.load_state_dict({k: v for k, v in synth.make_state_dict(100, 64).items() if not synth.is_trainable(k)}, strict=False)
    assert sorted(msg.missing_keys) == sorted(k for k in ref if synth.is_trainable(k))   # freeze rule main_image.py:250-256
    assert float(m.blocks[3].adaptmlp.up_proj.weight.abs().max()) == 0.0                # LoRA-style init (dynamic_adapter.py:112-117)
    assert m.blocks[0].adaptmlp.scale == 0.1 and m.blocks[0].mlp_token_select.tau == 5
    assert hasattr(m.blocks[0], "count_flops") and m.head.weight.shape == (100, 768)


def test_no_cpu_fallback():
    from _lib import DyTError
    m = _model()
    with pytest.raises(DyTError):
        m(torch.zeros(1, 3, 224, 224))


def test_unsupported_configs_are_rejected():
    from models.vision_transformer_IN21K import VisionTransformer
    tuning = Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                 ffn_adapter_scalar="0.1", ffn_num=8, d_model=768)
    with pytest.raises(NotImplementedError):
        VisionTransformer(embed_dim=384, num_heads=6, tuning_config=tuning, select_config=Cfg(open=True, keep_layers=0))
    with pytest.raises(ValueError):
        VisionTransformer(drop_path_rate=1.0, tuning_config=tuning, select_config=Cfg(open=True, keep_layers=0))
    assert VisionTransformer(drop_path_rate=0.1, tuning_config=tuning, select_config=Cfg(open=True, keep_layers=0)).drop_path_rate == 0.1   # (round 5: supported)
    from models.dynamic_adapter import Adapter
    a = Adapter(d_model=768, bottleneck=8)   # the class default is "in" (reference models/dynamic_adapter.py:88): supported since round 6
    assert a.adapter_layernorm_option == "in" and a.adapter_ln_code == 1 and tuple(a.adapter_layer_norm_before.weight.shape) == (768,)
    assert [k for k, _ in a.named_parameters()][:2] == ["adapter_layer_norm_before.weight", "adapter_layer_norm_before.bias"]   # the reference's registration order
    assert Adapter(d_model=768, bottleneck=8, adapter_layernorm_option="none").adapter_layer_norm_before is None
    with pytest.raises(ValueError):
        Adapter(d_model=768, bottleneck=8, adapter_layernorm_option="both")
    # select_config.keep_layers / open: the reference hands them to Block as `select`, which Block.__init__ never reads
    # (models/vision_transformer_IN21K.py:106,138,311) -- every value builds the same model there, and here
    a = VisionTransformer(tuning_config=tuning, select_config=Cfg(open=True, keep_layers=4))
    b = VisionTransformer(tuning_config=tuning, select_config=Cfg(open=False, keep_layers=0))
    assert [k for k, _ in a.named_parameters()] == [k for k, _ in b.named_parameters()] and len(a.blocks) == 12
    assert all(blk.mlp_token_select is not None for blk in a.blocks)


def test_key_mapping_covers_state_dict():
    import _lib
    seen = set()
    for k in synth.param_shapes(10, 8):
        pid, layer = _lib.key_to_param(k)
        assert 0 <= pid < _lib.P_POOL_QUERY and 0 <= layer < 12
        assert _lib.is_trainable_param(pid) == synth.is_trainable(k)
        seen.add(pid)
    assert seen == set(range(_lib.P_POOL_QUERY))
    for k in synth.param_shapes(10, 8, video=True):      # video model: + the 14 pooling-head tensors, all trainable
        pid, layer = _lib.key_to_param(k)
        assert _lib.is_trainable_param(pid) == synth.is_trainable(k)
        seen.add(pid)
    assert seen == set(range(_lib.P_AD_SCALE))
    # "learnable_scalar": one more trainable word per block
    pid, layer = _lib.key_to_param("blocks.7.adaptmlp.scale")
    assert pid == _lib.P_AD_SCALE == _lib.P_COUNT - 3 and layer == 7 and _lib.is_trainable_param(pid) and synth.is_trainable("blocks.7.adaptmlp.scale")
    # ffn_adapter_layernorm_option "in" / "out" (round 6): the adapter's own LayerNorm, two more trainable tensors per block
    for k, want in (("blocks.3.adaptmlp.adapter_layer_norm_before.weight", _lib.P_AD_LN_W), ("blocks.3.adaptmlp.adapter_layer_norm_before.bias", _lib.P_AD_LN_B)):
        pid, layer = _lib.key_to_param(k)
        assert pid == want and layer == 3 and _lib.is_trainable_param(pid) and synth.is_trainable(k)
    assert _lib.P_AD_LN_B == _lib.P_COUNT - 1


def test_video_module_parameter_surface():
    """video_models.video_vision_transformer_IN21K mirror: the reference's state_dict keys/shapes
    (video_models/video_vision_transformer_IN21K.py:27-75,407-410) and its freeze rule (main_video.py:279-285)."""
    from video_models.video_vision_transformer_IN21K import vit_base_patch16_224_in21k
    tuning = Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                 ffn_adapter_scalar="0.1", ffn_num=8, d_model=768)
    m = vit_base_patch16_224_in21k(num_classes=7, drop_path_rate=0.0, tuning_config=tuning, select_config=Cfg(open=True, keep_layers=0))
    shapes = synth.param_shapes(7, 8, video=True)
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert float(m.query_token.abs().max()) == 0.0     # reference init :407
    with pytest.raises(Exception):
        m(torch.zeros(1, 3, 2, 224, 224))              # no CPU path


def test_adaloss_mirror_equals_oracle():
    from models.losses import AdaLoss
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(6, 10, generator=g)
    y = torch.randint(0, 10, (6,), generator=g)
    sel = (torch.rand(6, 12, 196, 1, generator=g) > 0.4).float()
    for tmin, tw in ((0.0, 0.0), (0.1, 1.0)):
        crit = AdaLoss(torch.nn.CrossEntropyLoss(), token_target_ratio=0.5, token_loss_ratio=2.0, token_minimal=tmin, token_minimal_weight=tw)
        loss, d = crit(dict(prediction=logits, token_select=sel, token_logits=None), y)
        ref, dref = O.ada_loss(logits, sel, y, 0.5, 2.0, tmin, tw)
        assert abs(float(loss) - float(ref)) < 1e-6 and abs(float(d["token_loss"]) - float(dref["token_loss"])) < 1e-6


def test_lr_schedule_and_metrics_mirrors():
    import util.lr_sched as lr_sched
    from util.metrics import accuracy
    args = types.SimpleNamespace(lr=1e-3, min_lr=1e-6, warmup_epochs=5, epochs=100)
    opt = types.SimpleNamespace(param_groups=[dict(lr=0.0), dict(lr=0.0, lr_scale=0.5)])
    for e in (0.0, 2.5, 5.0, 37.2, 99.9):
        lr = lr_sched.adjust_learning_rate(opt, e, args)
        assert abs(lr - O.lr_at(e, 1e-3, 1e-6, 5, 100)) < 1e-12
        assert opt.param_groups[0]["lr"] == lr and opt.param_groups[1]["lr"] == lr * 0.5
    g = torch.Generator().manual_seed(1)
    out, tgt = torch.randn(50, 10, generator=g), torch.randint(0, 10, (50,), generator=g)
    assert [float(a) for a in accuracy(out, tgt, (1, 5))] == [float(a) for a in O.accuracy(out, tgt, (1, 5))]


def test_gumbel_sigmoid_utility_matches_oracle():
    from models.dynamic_adapter import _gumbel_sigmoid
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(3, 196, 1, generator=g)
    g1 = -torch.empty(3, 196, 1).exponential_(generator=g).log()
    g2 = -torch.empty(3, 196, 1).exponential_(generator=g).log()
    a = _gumbel_sigmoid(logits, 5, True, training=True, threshold=0.5, gumbels=(g1, g2))
    b, _ = O.gumbel_sigmoid(logits, g1, g2, 5.0, 0.5, True)
    assert torch.equal(a, b)
    assert torch.equal(_gumbel_sigmoid(logits, 5, True, training=False), O.gumbel_sigmoid(logits, 0, 0, 5.0, 0.5, False)[0])


def test_flops_accounting_closed_form():
    """block_flops_dict mirror: dense ViT-B/16 comes out at the reference's 17.6 GMACs (engine_finetune.py:268)
    and dropping tokens removes exactly 2*768*3072 MACs per token per block."""
    import block_flops_dict as F
    table = F.get_block_flops(ffn_num=64)
    assert table.shape == (198,)
    full = torch.ones(2, 12, 196, 1)
    f = F.batch_select_flops(2, table, full, 12, F.get_base_flops())
    adapters = 12 * 2 * 197 * 768 * 64 / 1e9 + 12 * 196 * 768 / 1e9
    assert abs(float(f[0]) - adapters - 17.56) < 0.05, float(f[0])          # 17.6 "GFlops" in the reference's comments
    half = full.clone(); half[:, :, ::2] = 0
    g = F.batch_select_flops(2, table, half, 12, F.get_base_flops())
    assert abs(float(f[0] - g[0]) - 12 * 98 * 2 * 768 * 3072 / 1e9) < 1e-4


def test_checkpoint_roundtrip(tmp_path):
    import types
    import misc
    m = _model(num_classes=10, ffn_num=8)
    args = types.SimpleNamespace(output_dir=str(tmp_path), epochs=3, save_freq=1, auto_remove=True, resume=None)
    misc.save_model(args, 0, m, m, None)
    misc.save_model(args, 1, m, m, None)
    assert sorted(os.listdir(tmp_path)) == ["checkpoint-1.pth"]                 # auto_remove keeps the newest
    m2 = _model(num_classes=10, ffn_num=8)
    args.resume = str(tmp_path / "checkpoint-1.pth")
    misc.load_model(args, m2, None)
    assert args.start_epoch == 2
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_fused_adamw_state_dict_is_torch_adamw_layout():
    """FusedAdamW.load_state_dict / state_dict speak torch.optim.AdamW's layout (the reference's optimizer, main_image.py:285;
    checkpoint format misc.py:296-352) -- checked without a GPU: the state of a real torch.optim.AdamW over the 74 trainable
    tensors loads before any engine exists and comes back identical."""
    from engine_finetune import FusedAdamW
    m = _model()
    names = [n for n, p in m.named_parameters() if synth.is_trainable(n)]
    params = [torch.nn.Parameter(torch.zeros_like(dict(m.named_parameters())[n])) for n in names]
    topt = torch.optim.AdamW(params, lr=3e-4, weight_decay=0.05)
    g = torch.Generator().manual_seed(0)
    for _ in range(2):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)
        topt.step()
    sd = topt.state_dict()
    opt = FusedAdamW(m, lr=1.0, weight_decay=0.0)
    opt.load_state_dict(sd)
    assert opt.step_count == 2 and opt.param_groups[0]["lr"] == 3e-4 and opt.param_groups[0]["weight_decay"] == 0.05
    back = opt.state_dict()
    assert back["param_groups"][0]["params"] == list(range(74)) and len(back["state"]) == 74
    for i in range(74):
        assert torch.equal(back["state"][i]["exp_avg"], sd["state"][i]["exp_avg"])
        assert torch.equal(back["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
        assert float(back["state"][i]["step"]) == 2.0
    topt2 = torch.optim.AdamW(params, lr=1.0)
    topt2.load_state_dict(back)                       # and torch accepts what we write
    assert topt2.param_groups[0]["lr"] == 3e-4
    with pytest.raises(ValueError):
        opt.load_state_dict({"state": {}, "param_groups": [{"params": [0, 1]}]})


def test_step_seed_is_unique_across_epochs():
    from engine_finetune import step_seed
    seen = {step_seed(e, it, base=1234) for e in range(40) for it in range(9000)}
    assert len(seen) == 40 * 9000


def test_gradient_scale_backoff_and_growth_follow_gradscaler_update():
    """FusedAdamW.overflow_backoff = GradScaler.update (reference misc.py:256-272) on the device-side counters: a skipped update halves the
    scale and resets the growth tracker, `growth_interval` consecutive applied updates double it (capped), the state survives
    state_dict() / load_state_dict() and is applied to a re-created engine.  Host logic only (fake engine, no GPU)."""
    import types
    import engine_finetune as E

    class Eng:
        grad_scale_log2 = 12
        def set_grad_scale_log2(self, k):
            self.grad_scale_log2 = int(k)

    opt = E.FusedAdamW.__new__(E.FusedAdamW)
    opt.model = types.SimpleNamespace(_engine=Eng())
    opt.growth_interval, opt.GROW_MAX_LOG2 = 5, 14
    opt._clean_run = opt._applied_seen = opt._skips_seen = 0
    opt._scale_log2, opt.step_count = None, 0
    counters = [0, 0]
    opt.applied_and_skipped = lambda: tuple(counters)
    eng = opt.model._engine
    counters[:] = [3, 0]
    assert opt.overflow_backoff(eng) == 0 and eng.grad_scale_log2 == 12          # 3 clean updates: below the interval
    counters[:] = [6, 0]
    assert opt.overflow_backoff(eng) == 0 and eng.grad_scale_log2 == 13          # 6 clean: grown once, tracker reset
    counters[:] = [7, 2]
    assert opt.overflow_backoff(eng) == 2 and eng.grad_scale_log2 == 12          # skips since the last call: halved (once), tracker reset
    counters[:] = [11, 2]
    assert opt.overflow_backoff(eng) == 0 and eng.grad_scale_log2 == 12          # 4 clean after the skip
    counters[:] = [12, 2]
    assert opt.overflow_backoff(eng) == 0 and eng.grad_scale_log2 == 13
    st = opt.scaler_state()
    assert st["scale_log2"] == 13 and st["skipped"] == 2 and st["_growth_tracker"] == 0
    # ... in torch.cuda.amp.GradScaler.state_dict()'s layout: the reference's loss_scaler.load_state_dict(checkpoint['scaler'])
    # (misc.py:350-351) takes it -- checked against the real GradScaler, which runs on CPU when disabled=False is forced
    assert st["scale"] == 2.0 ** 13 and st["growth_factor"] == 2.0 and st["backoff_factor"] == 0.5 and st["growth_interval"] == 5
    ref_scaler = torch.amp.GradScaler("cpu", enabled=True)
    ref_scaler.load_state_dict(st)
    assert ref_scaler.get_scale() == 2.0 ** 13 and ref_scaler.get_growth_interval() == 5
    eng2 = Eng()
    opt.model._engine = eng2
    opt.load_scaler_state(st)
    assert eng2.grad_scale_log2 == 13
    # the other direction: a GradScaler dict written by the reference (default scale 65536) -> log2, capped at GROW_MAX_LOG2
    eng3 = Eng()
    opt.model._engine = eng3
    opt.load_scaler_state(torch.amp.GradScaler("cpu", enabled=True, init_scale=2.0 ** 10, growth_interval=77).state_dict())
    assert eng3.grad_scale_log2 == 10 and opt.growth_interval == 77
    opt.load_scaler_state(torch.amp.GradScaler("cpu", enabled=True).state_dict())
    assert eng3.grad_scale_log2 == 14                                            # 2^16 capped at this test's GROW_MAX_LOG2
    opt.load_scaler_state(dict(scale_log2=13, growth_tracker=0, skipped=2, growth_interval=5))   # round-5 checkpoints
    assert eng3.grad_scale_log2 == 13 and opt.growth_interval == 5
    opt.model._engine = eng2
    for _ in range(4):                                                            # the cap
        counters[0] += 5
        opt.overflow_backoff(eng2)
    assert eng2.grad_scale_log2 == 14
