"""What did the student's tok_bwd actually CONSUME for the row whose output differs between identical steps?"""
import os, sys, ctypes
os.environ["DYT_DBG_SNAP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
import test_gpu_round2 as T
B = 4
M = B * 197
W = 4 * 768 + 8
L = _lib.lib()
L.dyt_debug_tok.restype = ctypes.c_int
L.dyt_debug_tok.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]
def run():
    m, _ = T._bench_model("bf16", "compact", B, 0.85)
    m.train()
    x, y = synth.make_batch(B, 100, seed=61)
    x, y = x.cuda(), y.cuda()
    eng = m.engine(B, x.device)
    for _ in range(2):
        eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=900)
        torch.cuda.synchronize()
    out = {}
    for l in range(0, 11):
        t = torch.empty(M * W, device="cuda")
        assert L.dyt_debug_tok(ctypes.c_void_p(t.data_ptr()), l, M * W) == 0
        out[l] = t.view(M, W).clone()
    for l in range(1, 11):
        t = torch.empty(M * W, device="cuda")
        assert L.dyt_debug_tok(ctypes.c_void_p(t.data_ptr()), 100 + l, M * W) == 0
        out[100 + l] = t.view(M, W).clone()
    return out, eng.grad.clone()
rr = [run() for _ in range(6)]
parts = [("du_in", 0, 768), ("dad", 768, 1536), ("dA2 row", 1536, 2304), ("u", 2304, 3072), ("mean2", 3072, 3073), ("rstd2", 3073, 3074), ("soft", 3074, 3075),
         ("maskf", 3075, 3076), ("dmask", 3076, 3077), ("ext", 3077, 3078), ("dlogit", 3078, 3079), ("r", 3079, 3080)]
for i in range(1, 6):
    if torch.equal(rr[0][1], rr[i][1]):
        print("run0 vs run%d: identical" % i); continue
    lnparts = [("dxn row", 0, 768), ("x row", 768, 1536), ("base(du)", 1536, 2304), ("g out", 2304, 3072), ("mean1", 3072, 3073), ("rstd1", 3073, 3074)]
    for l in range(10, 0, -1):   # ln_bwd(l) runs after tok_bwd(l)
        a, b = rr[0][0][100 + l], rr[i][0][100 + l]
        if not torch.equal(a, b):
            rows = (a != b).any(1).nonzero()[:, 0].tolist()
            desc = []
            for r in rows[:3]:
                d = [(n, int((a[r, s:e] != b[r, s:e]).sum()), float((a[r, s:e] - b[r, s:e]).abs().max()), (a[r, s:e] != b[r, s:e]).nonzero()[:4, 0].tolist()) for n, s, e in lnparts if not torch.equal(a[r, s:e], b[r, s:e])]
                desc.append("row %d: %s" % (r, d))
            print("run0 vs run%d: ln_bwd first differs at layer %d, %d rows %s: %s" % (i, l, len(rows), rows[:6], "; ".join(desc)))
            break
    for l in range(10, -1, -1):
        a, b = rr[0][0][l], rr[i][0][l]
        if not torch.equal(a, b):
            rows = (a != b).any(1).nonzero()[:, 0].tolist()
            desc = []
            for r in rows[:4]:
                d = [(n, int((a[r, s:e] != b[r, s:e]).sum()), float((a[r, s:e] - b[r, s:e]).abs().max())) for n, s, e in parts if not torch.equal(a[r, s:e], b[r, s:e])]
                desc.append("row %d: %s" % (r, d))
            print("run0 vs run%d: first differing consumed input at layer %d, %d rows: %s" % (i, l, len(rows), "; ".join(desc)))
            break
    else:
        print("run0 vs run%d: gradients differ but every input tok_bwd consumed is identical" % i)
