"""What do two runs of the same step disagree on?  Keeps a copy of ONE backward launch's output (DYT_DBG_DUMP="<slot>:<label>", labels as in
determinism_trace.py) next to that launch's checksum and compares the copies of several runs row by row.
PDUMP="0:L5 tok_bwd g" PONLY="tok_bwd g" PROWBYTES=3072 python tools/probes/determinism_dump.py"""
import ctypes, os, sys
dump = os.environ.get("PDUMP", "0:L5 tok_bwd g")
os.environ["DYT_DBG_CKSUM"] = "1"; os.environ["DYT_DBG_DUMP"] = dump; os.environ["DYT_DBG_CKSUM_ONLY"] = os.environ.get("PONLY", dump[2:].split(" ", 1)[1])
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _lib, synth
import test_gpu_round2 as T
COLS = int(os.environ.get("PCOLS", "768"))
B = int(os.environ.get("PB", "128")); NRUN = int(os.environ.get("PRUNS", "5")); dt = os.environ.get("PDTYPE", "f32")
L = _lib.lib()
L.dyt_debug_dump_read.restype = ctypes.c_int64
L.dyt_debug_dump_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
def run():
    m, _ = T._bench_model("bf16", "compact", B, 0.85)
    m.train()
    x, y = synth.make_batch(B, 100, seed=61)
    eng = m.engine(B, torch.device("cuda", 0))
    eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, seed=900)
    torch.cuda.synchronize()
    buf = torch.zeros(B * 197 * 768, device="cuda", dtype=torch.float32 if dt == "f32" else torch.bfloat16)
    n = L.dyt_debug_dump_read(ctypes.c_void_p(buf.data_ptr()), buf.numel() * buf.element_size())
    return buf[: n // buf.element_size()].view(-1, COLS).float().cpu(), eng.grad.clone()
runs = [run() for _ in range(NRUN)]
print("dumped", dump, "shape", tuple(runs[0][0].shape))
for i in range(1, NRUN):
    a, b = runs[0][0], runs[i][0]
    d = (a - b).abs()
    rows = (d.max(dim=1).values > 0).nonzero()[:, 0]
    print("run0 vs run%d: grads equal %s | rows differing %d of %d" % (i, bool(torch.equal(runs[0][1], runs[i][1])), rows.numel(), a.shape[0]), end="")
    if rows.numel():
        r = rows[:12].tolist()
        rel = [float(d[k].max() / (a[k].abs().max() + 1e-30)) for k in r]
        nel = [int((d[k] > 0).sum()) for k in r]
        print(" | first rows %s (image %s, token %s) | elements differing per row %s | max rel diff per row %s"
              % (r, [k * COLS // 768 // 197 for k in r], [k * COLS // 768 % 197 for k in r], nel, ["%.1e" % v for v in rel]), end="")
    print(flush=True)
