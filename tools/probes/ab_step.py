"""A/B timing of the fused step (B=128, bf16, compact) with the library given by DYT_LIB_PATH (default: in-tree)."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("PMASK", "none") != "none":   # the two passes on complementary CU sets: cu = alternating CU octets of every XCD, xcd = XCDs 0-3 / 4-7
    os.environ["DYT_DBG_SIDE_CU_MASK"] = os.environ["PMASK"]
import torch
import _lib, synth
if os.environ.get("DYT_LIB_PATH"):
    if os.environ.get("PPREC") in ("fp16", "fp16x3", "fp16x3f"):
        _lib.LIB_PATH_F16 = os.environ["DYT_LIB_PATH"]   # an IEEE-half build (libdyt_hip_f16.so twin)
    else:
        _lib.LIB_PATH = os.environ["DYT_LIB_PATH"]
    L = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
    probe = ctypes.CDLL(os.environ["DYT_LIB_PATH"])
    _lib.SYMBOLS = {k: v for k, v in _lib.SYMBOLS.items() if hasattr(probe, k)}
import test_gpu_round2 as T
B = 128
mode = os.environ.get("PMODE", "compact")
m, _ = T._bench_model(os.environ.get("PPREC", "bf16"), mode, B, 0.85)
m.train()
x, y = synth.make_batch(B, 100, seed=61)
x, y = x.cuda(), y.cuda()
eng = m.engine(B, x.device)
mm, vv = torch.zeros_like(eng.flat), torch.zeros_like(eng.flat)
if os.environ.get("DYT_NO_OVERLAP"):
    eng.set_option(_lib.OPT_STREAM_OVERLAP, 0)
if os.environ.get("DYT_OVERLAP"):
    eng.set_option(_lib.OPT_STREAM_OVERLAP, int(os.environ["DYT_OVERLAP"]))
for kv in filter(None, os.environ.get("DYT_OPTS", "").split(",")):   # "6=0,2=1": dyt_ctx_set_option(option, value)
    k, v = kv.split("=")
    eng.set_option(int(k), int(v))
_mask_stream = None
if os.environ.get("PMASK", "none") != "none":
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    words = (ctypes.c_uint32 * 8)(*([{"cu": 0x00FF00FF, "xcd": 0x0F0F0F0F, "a53": 0x07070707}[os.environ["PMASK"]]] * 8))
    h = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words) == 0
    _mask_stream = torch.cuda.ExternalStream(h.value)
    torch.cuda.set_stream(_mask_stream)
def step(i):
    eng.step_fwd_bwd(x, y, 0.7, 2.0, 0.0, 0.0, seed=900 + i, masked_dense=(mode == "masked"))
    if os.environ.get("PNOADAM"):
        return
    _lib.check(eng.L.dyt_adamw(_lib.ptr(eng.flat), _lib.ptr(eng.grad), _lib.ptr(mm), _lib.ptr(vv), eng.n_train, i + 1, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1.0, _lib.stream_ptr()))
NS = int(os.environ.get("PSTEPS", "20"))
for i in range(min(5, NS)):
    step(i)
torch.cuda.synchronize()
for rep in range(int(os.environ.get("PREPS", "3"))):
    t0 = time.perf_counter()
    for i in range(NS):
        step(5 + i)
    torch.cuda.synchronize()
    print("%s: %.2f ms/step" % (os.environ.get("DYT_LIB_PATH", "in-tree"), (time.perf_counter() - t0) / NS * 1e3), flush=True)
