import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch, _lib
L = _lib.lib()
B = 128
qkv = torch.randn(B * 197, 2304, device="cuda")
out = torch.empty(B * 197, 768, device="cuda")
for _ in range(12):
    _lib.check(L.dyt_attention(_lib.ptr(qkv), _lib.ptr(out), None, None, B, 1, _lib.stream_ptr()))
torch.cuda.synchronize()
