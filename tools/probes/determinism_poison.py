"""Stale-read detector: backward transients are filled with NaN patterns right before their producer kernels (DYT_DBG_POISON, csrc/model.hip);
a consumer that reads a row its producer has not made visible yet turns the gradient into NaN.  PPOISON=<mask> PB=128 PRUNS=6"""
import os, sys
os.environ["DYT_DBG_POISON"] = os.environ.get("PPOISON", "511")
if os.environ.get("PTNAN"):
    os.environ["DYT_DBG_TEACHER_NAN"] = "1"   # teacher backward computes on NaN; grad = the student pass only
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
import test_gpu_round2 as T
B = int(os.environ.get("PB", "128")); NRUN = int(os.environ.get("PRUNS", "6")); overlap = int(os.environ.get("POVERLAP", "1"))
def run():
    m, _ = T._bench_model("bf16", "compact", B, 0.85)
    m.train()
    x, y = synth.make_batch(B, 100, seed=61)
    x, y = x.cuda(), y.cuda()
    eng = m.engine(B, x.device)
    eng.set_option(_lib.OPT_STREAM_OVERLAP, overlap)
    out = []
    for i in range(3):
        eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=900 + i)
        torch.cuda.synchronize()
        out.append(eng.grad.clone())
    return out
runs = [run() for _ in range(NRUN)]
nan = sum(int(torch.isnan(g).any()) for r in runs for g in r)
diff = sum(int(not torch.equal(runs[0][k], runs[i][k])) for i in range(1, NRUN) for k in range(3))
print("RESULT poison=%s B=%d overlap=%d: %d of %d gradients contain NaN; %d of %d step comparisons differ"
      % (os.environ["DYT_DBG_POISON"], B, overlap, nan, 3 * NRUN, diff, 3 * (NRUN - 1)))
for r in runs[:2]:
    print([int(torch.isnan(g).sum()) for g in r])
