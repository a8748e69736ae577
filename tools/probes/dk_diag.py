import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, _lib, gpu_diag as D
from _lib import check, ptr, stream_ptr
L = _lib.lib(fp16=True)
B = 3
g = torch.Generator().manual_seed(23)
qkv = torch.randn(B * 197, 2304, generator=g) * 1.5
dout = torch.randn(B * 197, 768, generator=g) * 1e-3
ref_o, ref = D.attn_ref(qkv, B, dout)
check(L.dyt_set_global_option(_lib.OPT_F32_SPLIT16, 1))
out = torch.full((B * 197, 768), float("nan"), device="cuda")
dqkv = torch.full((B * 197, 2304), float("nan"), device="cuda")
qd, dd = qkv.cuda(), dout.cuda()
check(L.dyt_attention(ptr(qd), ptr(out), ptr(dd), ptr(dqkv), B, 0, stream_ptr()))
torch.cuda.synchronize()
dk = dqkv.cpu()[:, 768:1536].reshape(B, 197, 12, 64).double()
dk_ref = ref[:, 768:1536].reshape(B, 197, 12, 64).double()
err = dk - dk_ref
q3 = qkv.double().reshape(B, 197, 3, 12, 64).permute(2, 0, 3, 1, 4)
q, k, v = q3[0] * 0.125, q3[1], q3[2]
dO = dout.double().reshape(B, 197, 12, 64).permute(0, 2, 1, 3)
a = (q @ k.transpose(-2, -1)).softmax(-1)
dp = dO @ v.transpose(-2, -1)
delta = (a * dp).sum(-1, keepdim=True)
dS = a * (dp - delta)
flat = err.abs().amax(dim=3)   # [B,197,12]
for _ in range(4):
    idx = int(flat.argmax()); b_, key, h_ = idx // (197 * 12), (idx // 12) % 197, idx % 12
    e = err[b_, key, h_]
    Q = q[b_, h_]                                  # [197,64]
    coef = (Q @ e) / (Q * Q).sum(1)                # least-squares coefficient per q
    resid = ((e[None, :] - coef[:, None] * Q) ** 2).sum(1)
    qs = int(resid.argmin())
    print("b %d key %d head %d: |err| %.2e; best single-query explanation q* = %d (residual %.1f %% of |err|^2), implied dS error %.3e; true dS %.3e, p %.3e, dP %.3e, delta %.3e, s %.2f"
          % (b_, key, h_, float(e.abs().max()), qs, 100 * float(resid[qs] / (e ** 2).sum()), float(coef[qs]), float(dS[b_, h_, qs, key]), float(a[b_, h_, qs, key]),
             float(dp[b_, h_, qs, key]), float(delta[b_, h_, qs, 0]), float((q[b_, h_, qs] * k[b_, h_, key]).sum())))
    flat[b_, key, h_] = 0
