"""CPU emulation (no GPU, no library): how accurate is a GEMM whose hi x hi product runs in IEEE half and whose two correction products
A_hi W_lo + A_lo W_hi run in fp8 (e4m3, power-of-two scaled lo parts)?  Compared with plain fp16 operands and with the three-product
fp16 split (the fp16x3 mode) against fp64.  Activation-like A (unit normal with a few large channels), weight-like W (std 0.02)."""
import torch

torch.manual_seed(0)
M, K, N = 512, 768, 2304
A = torch.randn(M, K, dtype=torch.float64)
A[:, :4] *= 30.0                                    # a few massive channels, as in a ViT residual stream after LayerNorm
W = torch.randn(N, K, dtype=torch.float64) * 0.02
ref = A @ W.T
scale = ref.abs().max()


def split16(x):
    hi = x.float().half()
    lo = (x.float() - hi.float()).half()
    return hi.double(), lo.double()


def q8(x):                                          # e4m3 with a per-tensor power-of-two scale that puts max|x| near 256
    s = 2.0 ** torch.floor(torch.log2(256.0 / x.abs().max()))
    return (x * s).float().to(torch.float8_e4m3fn).double() / s


Ah, Al = split16(A)
Wh, Wl = split16(W)
err = lambda c: float((c - ref).abs().max() / scale)
print("plain fp16 operands            max err / max|C| = %.2e" % err(Ah @ Wh.T))
print("fp16 split, three products     max err / max|C| = %.2e" % err(Ah @ Wh.T + Ah @ Wl.T + Al @ Wh.T))
print("fp16 hi x hi + fp8 corrections max err / max|C| = %.2e" % err(Ah @ Wh.T + q8(Ah) @ q8(Wl).T + q8(Al) @ q8(Wh).T))
print("fp16 hi x hi + e5m2 corrections (no scaling)     = %.2e" % err(
    Ah @ Wh.T + Ah.float().to(torch.float8_e5m2).double() @ Wl.float().to(torch.float8_e5m2).double().T
    + Al.float().to(torch.float8_e5m2).double() @ Wh.float().to(torch.float8_e5m2).double().T))
