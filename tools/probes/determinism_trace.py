"""Locates the first backward kernel launch whose output differs between two runs of the same step (DYT_DBG_CKSUM hook of
csrc/model.hip: an integer checksum of every backward launch's output, per pass, in launch order).
PB=128 PRUNS=5 python tools/probes/determinism_trace.py"""
import ctypes, os, sys
os.environ["DYT_DBG_CKSUM"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
import test_gpu_round2 as T
B = int(os.environ.get("PB", "128")); NRUN = int(os.environ.get("PRUNS", "5")); overlap = int(os.environ.get("POVERLAP", "1"))
L = _lib.lib()
L.dyt_debug_checksums.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
L.dyt_debug_checksum_label.restype = ctypes.c_char_p
L.dyt_debug_checksum_label.argtypes = [ctypes.c_int, ctypes.c_int]
def trace():
    out = []
    for slot in (0, 1):
        buf = (ctypes.c_uint64 * 1024)(); n = ctypes.c_int()
        _lib.check(L.dyt_debug_checksums(slot, buf, 1024, ctypes.byref(n)))
        out.append([(L.dyt_debug_checksum_label(slot, i).decode(), buf[i]) for i in range(n.value)])
    return out
def run():
    m, _ = T._bench_model("bf16", "compact", B, 0.85)
    m.train()
    x, y = synth.make_batch(B, 100, seed=61)
    x, y = x.cuda(), y.cuda()
    eng = m.engine(B, x.device)
    eng.set_option(_lib.OPT_STREAM_OVERLAP, overlap)
    eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=900)
    torch.cuda.synchronize()
    return eng.grad.clone(), trace()
runs = [run() for _ in range(NRUN)]
print("launches traced per pass:", [len(t) for t in runs[0][1]])
for i in range(1, NRUN):
    eq = bool(torch.equal(runs[0][0], runs[i][0]))
    msg = []
    for slot, name in ((0, "student"), (1, "teacher")):
        a, b = runs[0][1][slot], runs[i][1][slot]
        diff = [k for k in range(min(len(a), len(b))) if a[k][1] != b[k][1]]
        msg.append("%s: %d of %d entries differ, first %s" % (name, len(diff), len(a), ["#%d %s" % (k, a[k][0]) for k in diff[:6]]))
    print("run0 vs run%d grads equal %s | %s" % (i, eq, " | ".join(msg)), flush=True)
