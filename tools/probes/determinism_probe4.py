"""Is the backward of ONE pass reproducible while an unrelated kernel stream runs beside it?  (separates a data race between
the two passes of a step from a hardware / cache effect of concurrent execution)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
if os.environ.get("DYT_LIB_PATH"):
    _lib.LIB_PATH = os.environ["DYT_LIB_PATH"]
import test_gpu_round2 as T
B = int(os.environ.get("PB", "4"))
m, _ = T._bench_model("bf16", "compact", B, 0.85)
m.train()
x, y = synth.make_batch(B, 100, seed=61)
x = x.cuda()
eng = m.engine(B, x.device)
eng.set_option(_lib.OPT_STREAM_OVERLAP, 0)
g1, g2 = synth.make_noise(B, seed=42, passes=1)
g1, g2 = g1[0].cuda().contiguous(), g2[0].cuda().contiguous()
keep = synth.make_dropout_masks(B, 64, seed=43)[0].cuda().contiguous()
gen = torch.Generator(device="cuda").manual_seed(44)
dl = torch.randn(B, 100, device="cuda", generator=gen) * 0.01
dtok = torch.tensor([3e-4, 1e-4, -2e-4], device="cuda")
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
big = torch.randn(64 * 1024 * 1024, device="cuda")
def run(noise):
    logits, ts, _ = eng.forward(x, slot=0, training=True, save=True, masked_dense=False, g1=g1, g2=g2, keep_mask=keep)
    torch.cuda.synchronize()
    if noise == "gemm":
        with torch.cuda.stream(side):
            for _ in range(40):
                a @ a
    elif noise == "copy":
        with torch.cuda.stream(side):
            for _ in range(40):
                big.mul_(1.0001)
    out = torch.zeros_like(eng.flat)
    eng.backward(0, dl, out, dtok=dtok)
    torch.cuda.synchronize()
    return logits.clone(), out
for noise in ("none", "gemm", "copy"):
    rr = [run(noise) for _ in range(5)]
    print(noise, "logits equal:", [bool(torch.equal(rr[0][0], r[0])) for r in rr[1:]], "grad equal:", [bool(torch.equal(rr[0][1], r[1])) for r in rr[1:]],
          "max diff", ["%.2e" % float((rr[0][1] - r[1]).abs().max()) for r in rr[1:]])
