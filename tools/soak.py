"""Soak / sanity run: 3 x 100 fused fine-tune steps through engine_finetune.train_one_epoch on ONE fixed synthetic batch
(B=128): the loss must fall steadily (memorisation), stay finite, throughput and memory must stay flat.
Run from the repo root on an MI355X: python tools/soak.py"""
import sys, os, time, math, types, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "dynamic-tuning_amd"))
import bench, synth
from engine_finetune import FusedAdamW, train_one_epoch
from models.losses import AdaLoss
args = types.SimpleNamespace(classes=100, ffn_num=64, precision=os.environ.get("PPREC", "bf16"), batch=128, mode="compact", video_frames=0)
dev = torch.device("cuda", 0)
model = bench.build_model(args, dev)
x, y = synth.make_batch(128, 100, seed=1)
x, y = x.to(dev), y.to(dev)
bench.calibrate_gates(model, x, 0.7)
opt = FusedAdamW(model, lr=5e-4, weight_decay=0.01)
crit = AdaLoss(torch.nn.CrossEntropyLoss(), token_target_ratio=0.7, token_loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0)
targs = types.SimpleNamespace(accum_iter=1, lr=5e-4, min_lr=0.0, warmup_epochs=0, epochs=4, metric="accuracy", nb_classes=100)
loader = [(x, y)] * 100     # the same batch: the loss must go down steadily (memorisation)
for ep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    st = train_one_epoch(model, crit, loader, opt, dev, ep, None, args=targs)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("epoch %d: loss %.4f base %.4f token %.4f teacher %.4f kl %.5f | %.0f img/s | mem %.1f GB" % (
        ep, st["loss"], st["base_loss"], st["token_loss"], st["teacher_loss"], st["distillation_loss"], 100 * 128 / dt,
        torch.cuda.memory_allocated() / 1e9), flush=True)
    assert math.isfinite(st["loss"])
