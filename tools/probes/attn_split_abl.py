"""split attention backward (dq + dkv kernels) with staging / compute switched off (DYT_DBG_ATTN_ABL bits: 1 = stage the first
head only, 2 = no tile loop); whole dyt_attention call (qkv split + forward + dq + dkv), B=128."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch, _lib
L = _lib.lib(fp16=True)
_lib.check(L.dyt_set_global_option(_lib.OPT_F32_SPLIT16, 1))
B = 128
qkv = torch.randn(B * 197, 2304, device="cuda")
out = torch.empty(B * 197, 768, device="cuda")
dout = torch.randn(B * 197, 768, device="cuda") * 1e-3
dqkv = torch.empty(B * 197, 2304, device="cuda")
def run(bwd):
    _lib.check(L.dyt_attention(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(dout) if bwd else None, _lib.ptr(dqkv) if bwd else None, B, 0, _lib.stream_ptr()))
for bwd in (False, True):
    for _ in range(3):
        run(bwd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(bwd)
    e1.record(); torch.cuda.synchronize()
    print("abl=%s gp=%s %s: %.1f us" % (os.environ.get("DYT_DBG_ATTN_ABL", "0"), os.environ.get("DYT_DBG_ATTN_GP", "3"), "fwd+bwd" if bwd else "fwd only", e0.elapsed_time(e1) * 100))
