"""Which transient of the student's backward differs first between repeated identical steps (two passes overlapped)?"""
import os, sys, ctypes
os.environ["DYT_DBG_SNAP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
import test_gpu_round2 as T
B = 4
M = B * 197
L = _lib.lib()
L.dyt_debug_snap.restype = ctypes.c_int
L.dyt_debug_snap.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
names = ["ddz", "dad", "dZ[:M*D]", "dA2", "du_at", "dO", "delta", "dqkv", "dxn", "g(after ln_bwd)", "g_at", "dmask"]
sizes = [M * 64 * 2, M * 768 * 2, M * 768 * 2, M * 768 * 2, M * 768 * 2, M * 768 * 2, B * 12 * 197 * 4, M * 2304 * 2, M * 768 * 2, M * 768 * 4, M * 768 * 2, M * 4]
def run():
    m, _ = T._bench_model("bf16", "compact", B, 0.85)
    m.train()
    x, y = synth.make_batch(B, 100, seed=61)
    x, y = x.cuda(), y.cuda()
    eng = m.engine(B, x.device)
    eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=900)
    torch.cuda.synchronize()
    eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=900)
    torch.cuda.synchronize()
    snaps = {}
    for l in range(1, 11):
        for k in range(12):
            out = torch.empty(sizes[k], dtype=torch.uint8, device="cuda")
            assert L.dyt_debug_snap(ctypes.c_void_p(out.data_ptr()), l, k, sizes[k]) == 0
            snaps[(l, k)] = out
    return snaps, eng.grad.clone()
rr = [run() for _ in range(5)]
for i in range(1, 5):
    if torch.equal(rr[0][1], rr[i][1]):
        print("run0 vs run%d: identical gradients" % i)
        continue
    for l in range(10, 0, -1):
        bad = [k for k in range(12) if not torch.equal(rr[0][0][(l, k)], rr[i][0][(l, k)])]
        if bad:
            desc = []
            for k in bad:
                a, b = rr[0][0][(l, k)], rr[i][0][(l, k)]
                w = 4 if k in (6, 9, 11) else 2
                rowlen = {0: 64, 6: 197, 7: 2304, 11: 1}.get(k, 768) * w
                d = (a != b).view(-1, rowlen).any(1).nonzero()[:, 0].tolist()
                desc.append("%s: %d bytes, rows %s" % (names[k], int((a != b).sum()), d[:10]))
            print("run0 vs run%d: first divergence at layer %d:" % (i, l), "; ".join(desc))
            break
