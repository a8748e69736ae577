"""The ONE tie rule and the ONE ReLU-side rule of the GPU parity tests.

Tie rule.  A token-keep decision m = [(logit + g) / tau > 0] of the reference is only defined up to the fp32 round-off of the
reference's OWN gate logit.  That round-off is measured, not guessed: the oracle is run a second time in float64 on the same
inputs and draws; sigma_l = max over the tokens of block l of |logit_fp32 - logit_fp64|.  A second correct fp32 implementation
(another summation order) carries an independent error of that size, so two of them can differ by ~2 sigma_l; the band is

    |z_ref| <= TIE_K * sigma_l / tau,   TIE_K = 4                      (z = (logit + g) / tau, the decision margin)

(0.5 ... 2e-6 in z for the test weights: sigma grows from 7e-7 in block 0 to 2.4e-6 in block 11.)  EVERY mode that claims bit-exact
masks -- fp32, fp16x3, fp16x3f, fp16x3h, fp16x3q -- and fp16f8 are judged by this one rule: a decision may differ from the reference
only inside the band.  A flip inside the band changes the token set the later blocks see, so the draw's later decisions, logits and
gradients are then consequences and are not compared (the caller skips them).

ReLU-side rule.  d(down_proj) has a jump wherever a bottleneck pre-activation crosses zero.  A row of that gradient may be set aside
only if the unit REALLY is on the other side of the ReLU for a token whose reference pre-activation lies within the forward's
round-off of zero -- verified from the library's saved bottleneck (dyt_debug_dact) against the oracle's pre-activation --, at most
RELU_MAX_UNITS rows per tensor and RELU_MAX_LAYERS adapter layers per step.  The budget counts LAYERS (round 6; round 5 counted tensors,
cap 2): one unit on the other side shows in BOTH the weight and the bias gradient of its layer, so "2 tensors" was exactly one event and
two shipped cases sat at the cap with zero headroom.  The measured distribution (tools/probes/r6/relu_side_events.py,
profiles/round6/r6_relu_side_events.txt: fp16x3q, B=16, 24 seeds, both passes) is what the cap is set from; every use prints the
headroom and a use that exhausts it warns."""
import torch

TIE_K = 4.0
RELU_MAX_UNITS = 2
RELU_MAX_LAYERS = 2     # adapter layers per step that may each set aside <= RELU_MAX_UNITS rows (in their down_proj weight AND bias)
RELU_MAX_TENSORS = 2 * RELU_MAX_LAYERS
_band_cache = {}


def tie_band(sd, x, g1s, g2s, keeps, mode, tl32, key=None, tau=5.0):
    """Per-block tie band in z units, [depth].  g1s / g2s / keeps: the STUDENT pass's draws ([depth,B,196] / [depth,B*197,r]);
    tl32: the fp32 oracle's token logits [B,depth,196]."""
    from oracle import dyt_oracle as O
    if key is not None and key in _band_cache:
        return _band_cache[key]
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        _, o64 = O.forward(sd64, x.double(), g1s.double(), g2s.double(), keeps, scale=0.1, training=True, mode=mode)
    tl64 = o64["token_logits"][..., 0]
    d = (tl32.double() - tl64).abs().amax(dim=(0, 2))            # [depth]: sigma_l
    z64 = (tl64 + (g1s - g2s).permute(1, 0, 2).double()) / tau
    z32 = (tl32.double() + (g1s - g2s).permute(1, 0, 2).double()) / tau
    split = ((z64 > 0) != (z32 > 0)).any(dim=2).any(dim=0)       # blocks where fp32 and fp64 already disagree: later sigmas are consequences
    if bool(split.any()):
        first = int(split.nonzero()[0])
        d[first + 1:] = d[:first + 1].max()
    band = TIE_K * d / tau
    if key is not None:
        _band_cache[key] = band
    return band


def judge_decisions(flip, z, band):
    """flip, z: [B,depth,196] (decisions that differ from the reference; the reference's |margin|).  Only the FIRST block that differs is
    judged: a flipped token changes what every later block sees, so later differences are consequences.  Returns (n_flips in all blocks,
    n outside the band in the first differing block, largest margin of a flip in that block, that block)."""
    n = int(flip.sum())
    if not n:
        return 0, 0, 0.0, -1
    blk = int(flip.any(dim=2).any(dim=0).nonzero()[0])
    f, zb = flip[:, blk], z[:, blk]
    return n, int((f & (zb > float(band[blk]))).sum()), float(zb[f].max()), blk


class ReluSideBudget:
    """Per-step accounting of the ReLU-side rule (one instance per step comparison).  The step's gradient sums the student pass (slot 0:
    g1[0], g2[0], keep[0]) and the teacher pass (slot 1: complete model, g1[1], g2[1], keep[1]); a unit may be on the other side in either."""

    def __init__(self, eng, sd, x, g1, g2, keep, mode):
        self.eng, self.sd, self.x, self.g1, self.g2, self.keep, self.mode = eng, sd, x, g1, g2, keep, mode
        self.used = 0
        self.layers = set()
        self._pre = {}

    def _oracle_preacts(self, p_):
        """Reference bottleneck pre-activations of every block of pass p_ (0 student, 1 teacher), [depth][B*197, r] (one more oracle forward
        per pass; only when the rule is invoked)."""
        if p_ not in self._pre:
            from oracle import dyt_oracle as O
            import torch.nn.functional as F
            with torch.no_grad():
                _, out = O.forward(self.sd, self.x, self.g1[p_], self.g2[p_], self.keep[p_], scale=0.1, complete_model=bool(p_), training=True,
                                   mode=self.mode, return_blocks=True)
                pre = []
                for l, xin in enumerate(out["blocks"][:-1]):
                    p = "blocks.%d." % l
                    u = xin + O.attention(self.sd, p, O.layer_norm(xin, self.sd[p + "norm1.weight"], self.sd[p + "norm1.bias"]))
                    w, b = self.sd[p + "adaptmlp.down_proj.weight"], self.sd[p + "adaptmlp.down_proj.bias"]
                    pre.append(F.linear(u, w, b).reshape(-1, b.numel()))
            self._pre[p_] = pre
        return self._pre[p_]

    def without_side_units(self, name, got, gr, e, what, fwd_roundoff):
        """Relative L2 error of the down_proj gradient `name` ([r,768] weight or [r] bias) without the rows of units that are verifiably on
        the other side of the ReLU.  fwd_roundoff: bound on the forward's error of a pre-activation (absolute)."""
        layer = int(name.split(".")[1])
        r = gr.shape[0]
        units, ntok, maxpre = set(), 0, 0.0
        for p_ in (0, 1):
            pre = self._oracle_preacts(p_)[layer]                    # [B*197, r]
            act = self.eng.debug_dact(p_, layer)                     # [rows, 64] fp32: relu(pre) (* dropout scale), rows = B*197 or B (cls tail)
            keep = self.keep[p_][layer]
            if act.shape[0] != pre.shape[0]:                         # last block on the cls rows only
                pre = pre.reshape(-1, 197, r)[:, 0, :]
                keep = keep.reshape(-1, 197, r)[:, 0, :]
            differs = (keep != 0) & ((pre > 0) != (act[:, :r] != 0))   # [rows, r]
            near = pre.abs() <= fwd_roundoff
            assert bool((differs <= near).all()), (what, "pass %d: a ReLU side differs where the reference pre-activation is %.2e from zero" % (
                p_, float(pre.abs()[differs & ~near].min())))
            units |= set(differs.any(dim=0).nonzero()[:, 0].tolist())
            ntok += int(differs.sum())
            if bool(differs.any()):
                maxpre = max(maxpre, float(pre.abs()[differs].max()))
        rows = (got - gr).reshape(r, -1).norm(dim=1)
        worst = rows.argsort(descending=True)[:RELU_MAX_UNITS].tolist()
        drop = [j for j in worst if j in units]
        top = rows.argsort(descending=True)[:4].tolist()
        assert drop, (what, "rel-L2 %.2e over the bound, but the worst rows %s (error norms %s of %.2e total) are not units on the other side of the ReLU %s" % (
            e, top, ["%.1e" % float(rows[j]) for j in top], float(rows.norm()), sorted(units)))
        self.used += 1
        self.layers.add(layer)
        assert len(self.layers) <= RELU_MAX_LAYERS and self.used <= RELU_MAX_TENSORS, (
            what, "ReLU-side rule invoked for more than %d adapter layers of one step (layers %s)" % (RELU_MAX_LAYERS, sorted(self.layers)))
        headroom = RELU_MAX_LAYERS - len(self.layers)
        if headroom == 0:
            import warnings
            warnings.warn("%s: the ReLU-side budget of this step is exhausted (layers %s of at most %d)" % (what, sorted(self.layers), RELU_MAX_LAYERS))
        keep_rows = torch.ones(r, dtype=torch.bool)
        keep_rows[drop] = False
        e2 = float((got - gr)[keep_rows].norm() / (gr[keep_rows].norm() + 1e-20))
        print("%s: rel-L2 %.2e, %.2e without bottleneck unit(s) %s -- verified on the other side of the ReLU in the library's forward (%d token(s), |reference "
              "pre-activation| <= %.1e); ReLU-side budget: %d of %d layers used, headroom %d" % (what, e, e2, drop, ntok, maxpre, len(self.layers),
                                                                                                     RELU_MAX_LAYERS, headroom))
        return e2
