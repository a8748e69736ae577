"""Does the run-to-run difference of the default (fully overlapped) bf16 schedule need the two passes to share CUs / XCDs?
PMASK = none | cu | xcd : the student pass (caller's stream) and the teacher pass (library side stream) get complementary CU masks
(cu: alternating CU octets inside every XCD; xcd: XCDs 0-3 vs 4-7, i.e. disjoint L2s).  Run: PB=128 PMASK=cu python tools/probes/determinism_cumask.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
mask = os.environ.get("PMASK", "none")
if mask != "none":
    os.environ["DYT_DBG_SIDE_CU_MASK"] = mask   # cu | xcd | iso (with DYT_DBG_ISO=<class bits>, see csrc/model.hip)
import torch
import _lib, synth
if os.environ.get("DYT_LIB_PATH"):   # A/B against another build of the library
    _lib.LIB_PATH = os.environ["DYT_LIB_PATH"]
    assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
import test_gpu_round2 as T
B = int(os.environ.get("PB", "128")); NRUN = int(os.environ.get("PRUNS", "6")); overlap = int(os.environ.get("POVERLAP", "1"))
_lib.lib()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
stream = None
if mask != "none":
    words = (ctypes.c_uint32 * 8)(*([{"cu": 0x00FF00FF, "xcd": 0x0F0F0F0F, "iso": 0x00FFFFFF}[mask]] * 8))
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    stream = torch.cuda.ExternalStream(h.value)
def run():
    m, _ = T._bench_model("bf16", "compact", B, 0.85)
    m.train()
    x, y = synth.make_batch(B, 100, seed=61)
    x, y = x.cuda(), y.cuda()
    eng = m.engine(B, x.device)
    eng.set_option(_lib.OPT_STREAM_OVERLAP, overlap)
    torch.cuda.synchronize()
    out = []
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for i in range(3):
            eng.step_fwd_bwd(x, y, 0.5, 2.0, 0.0, 0.0, seed=900 + i)
            torch.cuda.synchronize()
            out.append(eng.grad.clone())
    return out
import time
runs = []
for r in range(NRUN):
    t0 = time.time(); runs.append(run()); dt = time.time() - t0
bad = 0
for i in range(1, NRUN):
    eq = [bool(torch.equal(runs[0][k], runs[i][k])) for k in range(3)]
    d = [float((runs[0][k] - runs[i][k]).abs().max()) for k in range(3)]
    n = [int(((runs[0][k] - runs[i][k]).abs() > 0).sum()) for k in range(3)]
    bad += sum(not e for e in eq)
    print("mask", mask, "B", B, "run0 vs run%d:" % i, eq, d, n, flush=True)
print("RESULT mask=%s B=%d overlap=%d: %d of %d step comparisons differ" % (mask, B, overlap, bad, 3 * (NRUN - 1)))
