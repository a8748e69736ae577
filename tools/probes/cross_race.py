"""Which kernel is perturbed when the adapter weight-gradient kernel shares CUs with a GEMM?  Stream 1 loops dyt_wgrad_raw, stream 2
loops one bf16 GEMM variant (both at the B=128 size, M = 25216 rows), back to back without host syncs so that they overlap;
every output is compared bit for bit with the serial result of the same launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch
from _lib import check, lib, ptr
L = lib()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
M = int(os.environ.get("PM", 25216)); NREP = int(os.environ.get("PREP", 40))
torch.manual_seed(0)
X = (torch.randn(M, 768, device="cuda") * 1e-3).bfloat16()
Y = torch.randn(M, 64, device="cuda").bfloat16()
part = torch.zeros(int(L.dyt_wgrad_scratch_floats(M)), device="cuda")
def wgrad(st, w, xs, ys):
    w.zero_(); xs.zero_(); ys.zero_()
    check(L.dyt_wgrad_raw(ptr(X), ptr(Y), M, 64, 1, ptr(part), ptr(w), ptr(xs), ptr(ys), st))
wref = [torch.zeros(768 * 64, device="cuda"), torch.zeros(768, device="cuda"), torch.zeros(64, device="cuda")]
wgrad(0, *wref)
torch.cuda.synchronize()
for (N, K, v, name) in [(768, 768, 0, "128x128"), (768, 3072, 10, "256x256"), (768, 768, 30, "product split-row"), (3072, 768, 70, "pre-shuffled-W 128x256"),
                        (64, 768, -1, "wgrad only (control)")]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    wt = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    cref = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    if v >= 0:
        check(L.dyt_gemm_bf16_raw(ptr(a), ptr(wt), ptr(cref), M, N, K, v, 0))
    torch.cuda.synchronize()
    wouts = [[torch.zeros_like(t) for t in wref] for _ in range(NREP)]
    couts = [torch.zeros_like(cref) for _ in range(NREP if M * N < 4e7 else min(NREP, 12))]
    bad_w = bad_c = 0
    for rep in range(3):
        with torch.cuda.stream(s1):
            for j in range(NREP):
                wgrad(s1.cuda_stream, *wouts[j])
        if v >= 0:
            with torch.cuda.stream(s2):
                for j in range(NREP):
                    check(L.dyt_gemm_bf16_raw(ptr(a), ptr(wt), ptr(couts[j % len(couts)]), M, N, K, v, s2.cuda_stream))
        torch.cuda.synchronize()
        bad_w += sum(not all(torch.equal(x, y) for x, y in zip(wouts[j], wref)) for j in range(NREP))
        if v >= 0:
            bad_c += sum(not torch.equal(c, cref) for c in couts)
    print("wgrad next to GEMM %-24s N=%d K=%d: wgrad results differing from serial %d / %d ; GEMM results differing %d / %d"
          % (name, N, K, bad_w, 3 * NREP, bad_c, 3 * len(couts)), flush=True)
