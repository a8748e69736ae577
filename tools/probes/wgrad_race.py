"""The adapter weight-gradient kernel alone, on two streams at once, repeated: is its own output reproducible?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
import torch
from _lib import check, lib, ptr
L = lib()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for B in (4, 16, 128):
    M = B * 197
    torch.manual_seed(B)
    X = [(torch.randn(M, 768, device="cuda") * 1e-3).bfloat16() for _ in range(2)]
    Y = [(torch.randn(M, 64, device="cuda")).bfloat16() for _ in range(2)]
    n = int(L.dyt_wgrad_scratch_floats(M))
    part = [torch.zeros(n, device="cuda") for _ in range(2)]
    def run(i, st):
        w = torch.zeros(768 * 64, device="cuda"); xs = torch.zeros(768, device="cuda"); ys = torch.zeros(64, device="cuda")
        check(L.dyt_wgrad_raw(ptr(X[i]), ptr(Y[i]), M, 64, 1, ptr(part[i]), ptr(w), ptr(xs), ptr(ys), st))
        return w, xs, ys
    ref = [run(i, 0) for i in range(2)]
    torch.cuda.synchronize()
    r64 = (X[0].double().t() @ Y[0].double()).float()
    print("B=%d: max rel err of the serial result vs fp64: %.2e" % (B, float((ref[0][0].view(768, 64) - r64).abs().max() / r64.abs().max())))
    bad = 0
    for rep in range(200):
        outs = []
        for i, st in ((0, s1), (1, s2)):
            with torch.cuda.stream(st):
                outs.append(run(i, st.cuda_stream))
        torch.cuda.synchronize()
        for i in range(2):
            if not all(torch.equal(a, b) for a, b in zip(outs[i], ref[i])):
                bad += 1
    print("B=%d: %d of 400 concurrent results differ from the serial one" % (B, bad), flush=True)
