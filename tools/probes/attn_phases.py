import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
os.environ["DYT_DBG_ATTN_ABL"] = "9"
import torch, _lib
L = _lib.lib()
B = 128
qkv = torch.randn(B * 197, 2304, device="cuda")
out = torch.empty(B * 197, 768, device="cuda")
buf = (ctypes.c_ulonglong * 8)()
L.dyt_debug_attn(buf)
for _ in range(3):
    _lib.check(L.dyt_attention(_lib.ptr(qkv), _lib.ptr(out), None, None, B, 1, _lib.stream_ptr()))
L.dyt_debug_attn(buf)
n = max(1, buf[7])
names = ["stage_store+barrier", "prefetch issue", "S (QK^T)", "softmax", "PV", "O store", "end barrier"]
print("heads timed:", n)
for i, nm in enumerate(names):
    print("%-22s %8.0f cycles" % (nm, buf[i] / n))
print("sum %.0f" % (sum(buf[:7]) / n))
