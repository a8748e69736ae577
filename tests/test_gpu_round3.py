"""GPU parity tests added in round 3 (pytest -m gpu), all through the C ABI:
  * the K-concatenated fc2 GEMM (adapter up-projection as the leading k-tile, DYT_OPT_FC2_CAT): the raw kernel forms bitwise
    against the same contraction written as ONE plain GEMM over [A2 | A], and the whole step / inference forward with the
    option on and off;
  * the fused attention backward kernel, bit for bit against the two-kernel form;
  * precision "fp16x3" (DYT_OPT_F32_SPLIT16: frozen-weight GEMMs as three IEEE-half products) against the oracle at the
    parity bars of the exact-fp32 mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import synth  # noqa: E402
from test_gpu_round2 import _bench_model, _grad_tol  # noqa: E402


@pytest.mark.parametrize("M", [128, 3152, 17690, 25216])
def test_gemm_leading_ktile_form_is_the_concatenated_gemm(M):
    """C = A2 W2^T + A W^T with the second pair staged as k-tile 0 (csrc/gemm.hip CatArgs; the fc2 + adapter-up GEMM of the
    16-bit modes: models/vision_transformer_IN21K.py:157-163 + models/dynamic_adapter.py:128-137 in one accumulator chain):
    the 128x128 kernel, the 256x256 pipelined kernel and the product dispatch (row-split, both kernels) must give the
    bits of the plain 128x128 GEMM over the K-concatenated operands [A2 | A], [W2 | W]."""
    from _lib import check, lib, ptr, stream_ptr
    N, K = 768, 3072
    g = torch.Generator(device="cuda").manual_seed(M)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    a2 = torch.randn(M, 64, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    w2 = (torch.randn(N, 64, device="cuda", generator=g) * 0.05).bfloat16()
    abuf = torch.cat([a.reshape(-1), a2.reshape(-1)]).contiguous()      # A2 behind A, W2 behind W (the raw hook's convention)
    wbuf = torch.cat([w.reshape(-1), w2.reshape(-1)]).contiguous()
    acat = torch.cat([a2, a], dim=1).contiguous()
    wcat = torch.cat([w2, w], dim=1).contiguous()
    ref = torch.zeros(2 * M, N, device="cuda", dtype=torch.bfloat16)
    check(lib().dyt_gemm_bf16_raw(ptr(acat), ptr(wcat), ptr(ref), M, N, K + 64, 0, stream_ptr()))
    torch.cuda.synchronize()
    want = a2.float() @ w2.float().t() + a.float() @ w.float().t()
    assert float((ref[:M].float() - want).abs().max() / want.abs().max()) < 1e-2
    assert float(ref[M:].float().abs().max()) == 0.0
    for variant in (40, 41, 42, 41, 42):
        c = torch.zeros(2 * M, N, device="cuda", dtype=torch.bfloat16)
        check(lib().dyt_gemm_bf16_raw(ptr(abuf), ptr(wbuf), ptr(c), M, N, K, variant, stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(c, ref), (M, variant, int((c != ref).sum()))


def _step(prec, mode, cat, B=6):
    import _lib
    x, y = synth.make_batch(B, 100, seed=71)
    g1, g2 = synth.make_noise(B, seed=72)
    keep = synth.make_dropout_masks(B, 64, seed=73)
    m, _ = _bench_model(prec, mode, B, 0.85, kind="test")
    m.train()
    eng = m.engine(B, torch.device("cuda", 0))
    eng.set_option(_lib.OPT_FC2_CAT, cat)
    ls = torch.empty(B, 100, device="cuda")
    lt = torch.empty(B, 100, device="cuda")
    ts = torch.zeros(B, 12, 196, device="cuda")
    losses = eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                              g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt,
                              token_select=ts).clone()
    m.eval()
    with torch.no_grad():
        le, ae = m(x.cuda())                       # inference: compacted student pass, no saved activations
        lc, _ = m(x.cuda(), complete_model=True)
    torch.cuda.synchronize()
    names = [n for n, p in m.named_parameters() if synth.is_trainable(n)]
    grads = {n: eng.trainable_view(n, dict(m.named_parameters())[n].shape, eng.grad).clone().cpu() for n in names}
    out = dict(losses=losses.cpu(), ls=ls.cpu(), lt=lt.cpu(), ts=ts.cpu(), le=le.cpu(), lc=lc.cpu(), tse=ae["token_select"].cpu(), grads=grads)
    del m, eng
    torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize("precision,mode", [("fp16", "compact"), ("bf16", "compact"), ("fp16", "masked"), ("fp32", "compact")])
def test_fc2_leading_ktile_option_does_not_change_results(precision, mode):
    """DYT_OPT_FC2_CAT on / off: every compacted / complete pass computes x_out = u + adapter + mlp in one GEMM instead of two
    launches (training student pass: the saved MLP output then includes the adapter and tok_bwd subtracts <g, adapter> from the
    gate gradient) -- same training masks, logits / losses / gradients (the gate's included) within the 16-bit modes' own
    round-off; the masked mode and the fp32 mode keep the two-launch form (fp32: bitwise)."""
    a, b = _step(precision, mode, 1), _step(precision, mode, 0)
    tflips = int((a["ts"] != b["ts"]).sum())   # training-mode decisions of 6 x 2352 (student pass: its x_out differs by round-off)
    assert tflips <= (0 if precision != "bf16" else 6), tflips
    if tflips:   # a flipped token changes what the later blocks compute: the comparisons below only hold decision for decision
        assert float((a["lt"] - b["lt"]).abs().max()) < 1.5e-2 and float((a["lc"] - b["lc"]).abs().max()) < 1.5e-2
        return
    if precision == "fp32":
        for k in ("losses", "ls", "lt", "le", "lc", "tse"):
            assert torch.equal(a[k], b[k]), k
        for n in a["grads"]:
            assert torch.equal(a["grads"][n], b["grads"][n]), n
        return
    tol = 2e-3 if precision == "fp16" else 1.5e-2
    assert float((a["ls"] - b["ls"]).abs().max()) < tol, float((a["ls"] - b["ls"]).abs().max())
    assert float((a["lt"] - b["lt"]).abs().max()) < tol, float((a["lt"] - b["lt"]).abs().max())
    assert float((a["lc"] - b["lc"]).abs().max()) < tol
    flips = int((a["tse"] != b["tse"]).sum())
    assert flips <= (2 if precision == "fp16" else 12), flips   # of 6 x 2352 eval-mode decisions; bf16 vs the oracle: 13 of 37 632
    if flips == 0:
        assert float((a["le"] - b["le"]).abs().max()) < tol, float((a["le"] - b["le"]).abs().max())
    assert float((a["losses"][:5] - b["losses"][:5]).abs().max()) < tol * 5
    for n, ga in a["grads"].items():
        gb = b["grads"][n]
        if gb.numel() == 1:
            continue
        e = float((ga - gb).norm() / (gb.norm() + 1e-20))
        assert e < _grad_tol(n, precision), (n, e)


@pytest.mark.parametrize("B", [1, 3, 128])
def test_fused_attention_backward_is_bitwise_the_two_kernel_form(B):
    """attn_bwd_fused_bf16_kernel (dK/dV phase and dQ phase of a head in one persistent workgroup pass, csrc/attention.hip) runs
    the arithmetic of the two separate kernels on the same operands: dq, dk, dv must be the same bits, in both 16-bit operand
    types, at 1 / 3 / 1536 (image, head) pairs per launch (B=128 = the bench size: 6 heads per persistent workgroup).
    Reference op: Attention.forward's autograd backward, models/vision_transformer_IN21K.py:60-70."""
    import _lib
    from _lib import check, ptr, stream_ptr
    g = torch.Generator(device="cuda").manual_seed(B)
    qkv = torch.randn(B * 197, 2304, device="cuda", generator=g) * 1.5
    dout = torch.randn(B * 197, 768, device="cuda", generator=g)
    for fp16 in (False, True):
        L = _lib.lib(fp16=fp16)
        res = []
        check(L.dyt_set_global_option(_lib.OPT_ATTN_V2, 0))   # the round 1-4 kernels (round 5's are compared with them in test_gpu_round5.py)
        for mode in (0, 1, 2, 1):
            check(L.dyt_set_global_option(_lib.OPT_ATTN_BWD_FUSED, mode))
            out = torch.full((B * 197, 768), float("nan"), device="cuda")
            dqkv = torch.full((B * 197, 2304), float("nan"), device="cuda")
            check(L.dyt_attention(ptr(qkv), ptr(out), ptr(dout), ptr(dqkv), B, 1, stream_ptr()))
            torch.cuda.synchronize()
            res.append(dqkv.clone())
        check(L.dyt_set_global_option(_lib.OPT_ATTN_BWD_FUSED, 1))
        check(L.dyt_set_global_option(_lib.OPT_ATTN_V2, 3))
        assert torch.isfinite(res[0]).all()
        for r in res[1:]:
            assert torch.equal(r, res[0]), (B, fp16, int((r != res[0]).sum()))


@pytest.mark.parametrize("prec", ["fp16x3", "fp16x3f", "fp16x3h"])
@pytest.mark.parametrize("mode", ["masked", "compact"])
def test_split_fp16x3_mode_meets_the_fp32_parity_bars(mode, prec):
    """precision "fp16x3" (DYT_OPT_F32_SPLIT16: the fp32 mode with every frozen-weight GEMM computed on the 16-bit matrix cores as
    hi*hi + hi*lo + lo*hi of IEEE-half parts, fp32 accumulate and epilogues; attention / LayerNorm / adapters exact fp32) against
    the CPU oracle at BASELINE configs[0] size (B=16): the bars of the exact-fp32 mode -- logits within 1e-3 (north_star), gate
    decisions bit-exact outside fp32 round-off of the threshold, losses 1e-4, all 74 gradients 2e-3 relative.
    "fp16x3f" (DYT_OPT_F32_SPLIT16 = 2): the same forward, every gradient product as the hi * hi term alone -- the same bars."""
    from oracle import dyt_oracle as O
    B, C, r, target = 16, 100, 64, 0.5
    x, y = synth.make_batch(B, C, seed=31)
    g1, g2 = synth.make_noise(B, seed=32)
    keep = synth.make_dropout_masks(B, r, seed=33)
    sd = synth.make_state_dict(C, r, seed=0, kind="test", gate_bias=0.85)
    d_ref, g_ref, (ref_ls, ref_lt, tok) = O.step_grads(sd, x, y, g1, g2, keep, scale=0.1, mode=mode, token_target_ratio=target)
    ref_ts = tok["token_select"].detach()
    z = ((tok["token_logits"].detach()[..., 0].permute(1, 0, 2) + g1[0] - g2[0]) / 5.0).abs()   # decision margins [12,B,196]
    m, _ = _bench_model(prec, mode, B, 0.85, classes=C, r=r, kind="test")
    m.train()
    eng = m.engine(B, torch.device("cuda", 0))
    ls = torch.empty(B, C, device="cuda"); lt = torch.empty(B, C, device="cuda"); ts = torch.zeros(B, 12, 196, device="cuda")
    losses = eng.step_fwd_bwd(x.cuda(), y.cuda(), target, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                              g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt, token_select=ts).cpu()
    es, et = float((ls.cpu() - ref_ls.detach()).abs().max()), float((lt.cpu() - ref_lt.detach()).abs().max())
    flip = ts.cpu() != ref_ts[..., 0].float()
    print("%s/%s: logits %.2e / %.2e, gate flips %d of %d" % (prec, mode, es, et, int(flip.sum()), flip.numel()))
    assert es < 1e-3 and et < 1e-3, (es, et)
    import parity_rules as PR
    band = PR.tie_band(sd, x, g1[0], g2[0], keep[0], mode, tok["token_logits"].detach()[..., 0], key=("step", B, C, r, mode, 31))
    nflip, outside, zmax, blk = PR.judge_decisions(flip, z.permute(1, 0, 2), band)
    assert outside == 0, (nflip, outside, zmax, blk)   # the one tie rule (tests/parity_rules.py)
    for i, k in enumerate(("loss", "base_loss", "token_loss", "teacher_loss", "distillation_loss")):
        assert abs(float(losses[i]) - float(d_ref[k])) < 1e-4 * max(1.0, abs(float(d_ref[k]))), (k, float(losses[i]), float(d_ref[k]))
    worst, wname = 0.0, ""
    for n, gr in g_ref.items():
        if gr.numel() == 1:
            continue
        got = eng.trainable_view(n, gr.shape, eng.grad).cpu()
        e = float((got - gr).norm() / (gr.norm() + 1e-20))
        if e > worst:
            worst, wname = e, n
        assert e < 2e-3, (n, e)
    # measured: 71 gradients <= 6e-6; down_proj of blocks 1, 2, 4 0.8-1.5e-4, each from ONE ReLU-mask flip (one row of the gradient
    # carries the whole error: tests/diag_grad_table.py).  fp16x3f: worst 6.3e-4 (gate of block 0), spread over all rows
    print("%s/%s worst gradient rel-L2 %.2e (%s)" % (prec, mode, worst, wname))
    assert worst < {"fp16x3": 3e-4, "fp16x3f": 1.5e-3, "fp16x3h": 2e-3}[prec], (wname, worst)   # ~2x measured


@pytest.mark.parametrize("B", [16, 128])
def test_fp16x3f_forward_is_bitwise_the_fp16x3_forward(B):
    """DYT_OPT_F32_SPLIT16 = 2 changes gradient products only: logits of both passes, token-keep decisions and the
    loss components of a training step equal those of value 1 bit for bit (B = 16 and the bench size); the gradients differ
    (at the 1e-4 level) and stay finite."""
    import _lib
    x, y = synth.make_batch(B, 100, seed=71)
    res = {}
    # "fp16x3h+fused": round 6 -- fp16x3h's default runs the adapter up-projection as three leading tiles of the fc2 GEMM (a three-part product
    # instead of the exact-fp32 MFMA kernel's): the same decisions, logits to the three-part products' round-off; DYT_OPT_FC2_CAT = 0 is the
    # fp16x3 forward bit for bit, as before
    for prec in ("fp16x3", "fp16x3f", "fp16x3h", "fp16x3h+fused"):
        m, _ = _bench_model(prec.split("+")[0], "compact", B, 0.85)
        m.train()
        eng = m.engine(B, torch.device("cuda", 0))
        if prec == "fp16x3h":
            eng.set_option(_lib.OPT_FC2_CAT, 0)
        ls = torch.empty(B, 100, device="cuda"); lt = torch.empty(B, 100, device="cuda")
        ts = torch.zeros(B, 12, 196, device="cuda")
        losses = eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.7, 2.0, 0.0, 0.0, seed=7, logits_s=ls, logits_t=lt, token_select=ts)
        torch.cuda.synchronize()
        res[prec] = (ls.clone(), lt.clone(), ts.clone(), losses.clone(), eng.grad.clone())
        del eng, m
        torch.cuda.empty_cache()
    a = res["fp16x3"]
    f = res["fp16x3h+fused"]
    assert torch.equal(a[2], f[2])
    d_s, d_t = float((a[0] - f[0]).abs().max()), float((a[1] - f[1]).abs().max())
    print("B=%d: fp16x3h with the fused up-projection vs fp16x3: logits %.2e student / %.2e teacher" % (B, d_s, d_t))
    assert d_s < 1e-5 and d_t < 1e-5 and float((a[3][:5] - f[3][:5]).abs().max()) < 1e-5
    for other in ("fp16x3f", "fp16x3h"):
        b = res[other]
        for i, name in enumerate(("student logits", "teacher logits", "token_select")):
            assert torch.equal(a[i], b[i]), (other, name)
        assert torch.equal(a[3][:5], b[3][:5]), (other, a[3], b[3])
        assert torch.isfinite(b[4]).all()
        rel = float((a[4] - b[4]).norm() / a[4].norm())
        print("B=%d: %s vs fp16x3 flat gradient rel-L2 %.2e" % (B, other, rel))
        assert 0.0 < rel < (2e-3 if other == "fp16x3f" else 4e-3), (other, rel)


@pytest.mark.parametrize("B", [1, 3, 64])
def test_split_attention_matches_fp64(B):
    """attn_fwd_split_kernel / attn_bwd_dq_split_kernel / attn_bwd_dkv_split_kernel (fp32 q / k / v / dO / probabilities / dS as two
    IEEE-half parts, three MFMA products per fp32-class product; the fp16x3 mode's attention, Attention.forward
    models/vision_transformer_IN21K.py:60-70 and its autograd backward) against the fp64 reference at the tolerance of the
    exact-fp32 MFMA kernels, and against those kernels."""
    import gpu_diag as D
    import _lib
    from _lib import check, ptr, stream_ptr
    L = _lib.lib(fp16=True)
    g = torch.Generator().manual_seed(20 + B)
    qkv = torch.randn(B * 197, 2304, generator=g) * 1.5
    dout = torch.randn(B * 197, 768, generator=g) * 1e-3   # gradient-sized: the split kernels scale dO by 2^12 before splitting
    ref_o, ref_dq = D.attn_ref(qkv, B, dout)
    res = []
    for split in (0, 1):
        check(L.dyt_set_global_option(_lib.OPT_F32_SPLIT16, split))
        out = torch.full((B * 197, 768), float("nan"), device="cuda")
        dqkv = torch.full((B * 197, 2304), float("nan"), device="cuda")
        qd, dd = qkv.cuda(), dout.cuda()   # named: a temporary would be freed (and its block handed to the next allocation) before the launch
        check(L.dyt_attention(ptr(qd), ptr(out), ptr(dd), ptr(dqkv), B, 0, stream_ptr()))
        torch.cuda.synchronize()
        res.append((out.cpu(), dqkv.cpu()))
    check(L.dyt_set_global_option(_lib.OPT_F32_SPLIT16, 0))
    e_exact, e_split = D.relerr(res[0][0], ref_o), D.relerr(res[1][0], ref_o)
    print("attention forward vs fp64: exact-fp32 kernel %.2e, split kernel %.2e" % (e_exact, e_split))
    assert e_split < 2e-5 and e_split < 4 * e_exact + 1e-6, (e_exact, e_split)
    errs = {}
    for i, nm in enumerate(("dq", "dk", "dv")):
        sl = slice(i * 768, (i + 1) * 768)
        ee, es = D.relerr(res[0][1][:, sl], ref_dq[:, sl]), D.relerr(res[1][1][:, sl], ref_dq[:, sl])
        print("attention backward %s vs fp64: exact-fp32 kernels %.2e, split kernels %.2e" % (nm, ee, es))
        errs[nm] = (ee, es)
    # measured: dq 6e-7, dk 7e-7-1.1e-6, dv 6e-7 (the exact kernels: 9e-7 / 1.2e-6 / 7e-7).  dk was 1.5e-5 / 8.7e-5 with single dS entries
    # off by one 16-bit ulp until `split2` pinned its source value (dyt_common.h: the compiler had folded the conversion into the
    # producing multiply for the lo part only); this bound is what catches that class of error.
    for nm, (ee, es) in errs.items():
        assert es < 4 * ee + 1e-6, (nm, ee, es)
