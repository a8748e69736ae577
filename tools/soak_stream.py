"""Soak on CHANGING data: 224 fused fine-tune steps over 32 distinct synthetic batches (B=128: class-dependent
patterns under N(0,1) noise, so there is something to learn), the same images / labels / Philox seeds in every precision; the loss curves of
the fp16 (headline) and fp16x3q (parity) modes are compared with the exact-fp32 mode's step for step.
Run from the repo root on an MI355X: [DROP_PATH=0.1] [PPRECS=fp32,fp16] python tools/soak_stream.py [out.json]"""
import json, math, os, sys, time, types
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "dynamic-tuning_amd"))
import torch
import bench, synth
from engine_finetune import FusedAdamW, train_step
from models.losses import AdaLoss

NB, B, C, STEPS = 32, 128, 100, 224
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
pattern = torch.randn(C, 3, 224, 224, generator=g) * 0.35
batches = []
for i in range(NB):
    y = torch.randint(0, C, (B,), generator=g)
    x = torch.randn(B, 3, 224, 224, generator=g) + pattern[y]
    batches.append((x.to(dev), y.to(dev)))
curves, rates = {}, {}
for prec in os.environ.get("PPRECS", "fp32,fp16,fp16x3q,bf16").split(","):
    args = types.SimpleNamespace(classes=C, ffn_num=64, precision=prec, batch=B, mode="compact", video_frames=0)
    model = bench.build_model(args, dev)
    model.drop_path_rate = float(os.environ.get("DROP_PATH", "0"))   # stochastic depth: the same Philox draws in every precision (same seeds)
    bench.calibrate_gates(model, batches[0][0], 0.7)
    opt = FusedAdamW(model, lr=5e-4, weight_decay=0.01)
    opt.growth_interval = 100      # exercise the loss-scale growth path inside the soak (GradScaler default: 2000)
    crit = AdaLoss(torch.nn.CrossEntropyLoss(), token_target_ratio=0.7, token_loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0)
    out = torch.zeros(STEPS, 8, device=dev)
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(STEPS):
        x, y = batches[it % NB]
        train_step(model, x, y, opt, crit, losses_out=out[it], seed=1000 + it)
        if (it + 1) % 20 == 0:
            opt.overflow_backoff(model._engine)
    torch.cuda.synchronize(); dt = time.time() - t0
    host = out.cpu()
    assert torch.isfinite(host).all(), prec
    curves[prec] = {"loss": [round(float(v), 5) for v in host[:, 0]], "base_loss": [round(float(v), 5) for v in host[:, 1]],
                    "keep": [round(float(v), 4) for v in host[:, 5]]}
    rates[prec] = STEPS * B / dt
    a, k = opt.applied_and_skipped()
    print("%-8s loss first 8 steps %.4f -> last 8 steps %.4f | keep %.3f | %.0f img/s | updates applied %d skipped %d | gradient scale 2^%s" % (
        prec, float(host[:8, 0].mean()), float(host[-8:, 0].mean()), float(host[-8:, 5].mean()), rates[prec], a, k,
        getattr(model._engine, "grad_scale_log2", None)), flush=True)
    del model, opt
    torch.cuda.empty_cache()
if "fp32" in curves:
    ref = torch.tensor(curves["fp32"]["loss"])
    for prec, c in curves.items():
        if prec == "fp32":
            continue
        d = (torch.tensor(c["loss"]) - ref).abs()
        print("%-8s vs fp32, per-step |loss difference|: max %.4f (step %d), mean %.5f; last 32 steps max %.4f  (loss %.3f -> %.3f)" % (
            prec, float(d.max()), int(d.argmax()), float(d.mean()), float(d[-32:].max()), float(ref[0]), float(ref[-1])))
if len(sys.argv) > 1:
    json.dump({"steps": STEPS, "batches": NB, "batch": B, "curves": curves, "images_per_s": rates}, open(sys.argv[1], "w"))
