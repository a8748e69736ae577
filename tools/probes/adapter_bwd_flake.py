"""dyt_adapter_bwd (precision 1, r = 8) against a torch fp32 evaluation of the same formula on the device: where do dx / d_down_b differ?
(run in several fresh processes: the deviation differs from process to process, not from call to call)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch, _lib
L = _lib.lib()
precision, r, scale = int(os.environ.get("PP", "1")), int(os.environ.get("PR", "8")), 1.0
M = 3 * 197
g = torch.Generator().manual_seed(10 + r)
x = torch.randn(M, 768, generator=g).cuda(); dw = (torch.randn(r, 768, generator=g) * 0.03).cuda(); db = (torch.randn(r, generator=g) * 0.05).cuda()
uw = (torch.randn(768, r, generator=g) * 0.05).cuda(); dout = (torch.randn(M, 768, generator=g) * 0.01).cuda()
keep = (torch.rand(M, r, generator=g) > 0.1)
if os.environ.get("PWARM"):   # what the pytest does before: a context is created and destroyed (its arenas go back to the HIP pool)
    junk = torch.full((256 << 20,), 3.0e4, device="cuda"); del junk; torch.cuda.empty_cache()
pre = x @ dw.t() + db
ddz_ref = (dout @ uw) * scale * (pre > 0) * keep.cuda() / 0.9
dx_ref = ddz_ref @ dw
dx = torch.zeros(M, 768, device="cuda"); gdw = torch.zeros(r, 768, device="cuda"); gdb = torch.zeros(r, device="cuda")
guw = torch.zeros(768, r, device="cuda"); gub = torch.zeros(768, device="cuda")
k8 = keep.to(torch.uint8).cuda()
_lib.check(L.dyt_adapter_bwd(_lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(uw), _lib.ptr(dout), _lib.ptr(dx), _lib.ptr(gdw), _lib.ptr(gdb),
                             _lib.ptr(guw), _lib.ptr(gub), M, r, scale, 0.1, _lib.ptr(k8), ctypes.c_uint64(0), precision, _lib.stream_ptr()))
torch.cuda.synchronize()
e = (dx - dx_ref)
rows = e.norm(dim=1) / dx_ref.norm(dim=1).clamp_min(1e-20)
bad = (rows > 0.05).nonzero().flatten().tolist()
print("dx rel-L2 %.3e; rows off by > 5%%: %d of %d %s; d_down_b rel %.3e" % (float(e.norm() / dx_ref.norm()), len(bad), M, bad[:20],
      float((gdb - ddz_ref.sum(0)).norm() / ddz_ref.sum(0).norm())))
