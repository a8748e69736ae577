"""Isolated attention fwd + bwd at B=128 through the unit entry (fp16 library).  Run under rocprofv3 --kernel-trace --stats
(tools/probes/r5/attn_bench.sh) for per-kernel times; prints max errors vs an fp64 reference at B=2 first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch, _lib
L = _lib.lib(fp16=True)
B = int(os.environ.get("AB", "128"))
reps = int(os.environ.get("AREPS", "5"))


def ref(qkv, B, dout):
    x = qkv.double().reshape(B, 197, 3, 12, 64).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)
    q, k, v = x[0], x[1], x[2]
    s = (q * 0.125) @ k.transpose(-1, -2)
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * 197, 768)
    o.backward(dout.double())
    g = x.grad.permute(1, 3, 0, 2, 4).reshape(B * 197, 2304)
    return o.detach(), g


def relerr(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


g = torch.Generator().manual_seed(5)
for Bs in (1, 3):
    qkv = torch.randn(Bs * 197, 2304, generator=g) * 1.5
    dout = torch.randn(Bs * 197, 768, generator=g)
    ro, rg = ref(qkv, Bs, dout)
    out = torch.full((Bs * 197, 768), float("nan"), device="cuda")
    dq = torch.full((Bs * 197, 2304), float("nan"), device="cuda")
    _lib.check(L.dyt_attention(_lib.ptr(qkv.cuda()), _lib.ptr(out), _lib.ptr(dout.cuda()), _lib.ptr(dq), Bs, 1, _lib.stream_ptr()))
    print("B=%d fwd %.2e dq %.2e dk %.2e dv %.2e" % (Bs, relerr(out.cpu(), ro), relerr(dq.cpu()[:, :768], rg[:, :768]),
                                                   relerr(dq.cpu()[:, 768:1536], rg[:, 768:1536]), relerr(dq.cpu()[:, 1536:], rg[:, 1536:])))
qkv = torch.randn(B * 197, 2304, device="cuda") * 1.5
dout = torch.randn(B * 197, 768, device="cuda")
out = torch.empty(B * 197, 768, device="cuda")
dq = torch.empty(B * 197, 2304, device="cuda")
for _ in range(reps):
    _lib.check(L.dyt_attention(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(dout), _lib.ptr(dq), B, 1, _lib.stream_ptr()))
torch.cuda.synchronize()
print("done", float(out.abs().mean()), float(dq.abs().mean()))
