#!/usr/bin/env python3
"""How often does an adapter bottleneck unit come out on the other side of the ReLU than the reference's?  (tests/parity_rules.py: the
ReLU-side rule's budget is set from this.)  fp16x3q (bench.py's parity_mode), B=16, both passes of a training step, SEEDS draws: for every
block the library's saved bottleneck (dyt_debug_dact) is compared with the oracle's pre-activation; an EVENT = a (pass, block) with at
least one kept unit whose side differs.  Prints per seed the events, the units involved and the largest |reference pre-activation| among
them, then the distribution of events / layers per step.   usage (GPU box): python tools/probes/r6/relu_side_events.py [SEEDS]"""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (ROOT, os.path.join(ROOT, "dynamic-tuning_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import synth  # noqa: E402
from oracle import dyt_oracle as O  # noqa: E402
from test_gpu_round2 import _bench_model  # noqa: E402

SEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B, C, r, prec, mode = 16, 100, 64, os.environ.get("PPREC", "fp16x3q"), "compact"
torch.set_num_threads(synth.available_cores())
sd = synth.make_state_dict(C, r, seed=0, kind="test", gate_bias=0.85)
m, _ = _bench_model(prec, mode, B, 0.85, classes=C, r=r, kind="test")
m.train()
eng = m.engine(B, torch.device("cuda", 0))
layers_per_step, events_per_step, worst_pre = Counter(), Counter(), 0.0
over_per_step = Counter()   # layers whose down_proj weight or bias gradient exceeds the 16-bit-backward bound (= where the tests would invoke the rule)
BOUND = 3e-3
for i in range(SEEDS):
    seed = 31 + 10 * i
    x, y = synth.make_batch(B, C, seed=seed)
    g1, g2 = synth.make_noise(B, seed=seed + 1)
    keep = synth.make_dropout_masks(B, r, seed=seed + 2)
    eng.step_fwd_bwd(x.cuda(), y.cuda(), 0.5, 2.0, 0.0, 0.0, g1=g1.cuda().contiguous(), g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous())
    torch.cuda.synchronize()
    ev, layers = [], set()
    for p_ in (0, 1):
        with torch.no_grad():
            _, out = O.forward(sd, x, g1[p_], g2[p_], keep[p_], scale=0.1, complete_model=bool(p_), training=True, mode=mode, return_blocks=True)
        for l, xin in enumerate(out["blocks"][:-1]):
            pf = "blocks.%d." % l
            with torch.no_grad():
                u = xin + O.attention(sd, pf, O.layer_norm(xin, sd[pf + "norm1.weight"], sd[pf + "norm1.bias"]))
                pre = F.linear(u, sd[pf + "adaptmlp.down_proj.weight"], sd[pf + "adaptmlp.down_proj.bias"]).reshape(-1, r)
            act = eng.debug_dact(p_, l)
            kp = keep[p_][l]
            if act.shape[0] != pre.shape[0]:
                pre, kp = pre.reshape(-1, 197, r)[:, 0, :], kp.reshape(-1, 197, r)[:, 0, :]
            differs = (kp != 0) & ((pre > 0) != (act[:, :r] != 0))
            if bool(differs.any()):
                mp = float(pre.abs()[differs].max())
                worst_pre = max(worst_pre, mp)
                ev.append("pass %d block %d unit(s) %s (%d token(s), |pre| <= %.1e)" % (p_, l, differs.any(dim=0).nonzero()[:, 0].tolist(), int(differs.sum()), mp))
                layers.add(l)
    layers_per_step[len(layers)] += 1
    events_per_step[len(ev)] += 1
    _, g_ref, _ = O.step_grads(sd, x, y, g1, g2, keep, scale=0.1, mode=mode, token_target_ratio=0.5)
    over = {}
    for n, gr in g_ref.items():
        if "down_proj" not in n:
            continue
        e = float((eng.trainable_view(n, gr.shape, eng.grad).cpu() - gr).norm() / (gr.norm() + 1e-20))
        if e > BOUND:
            over.setdefault(int(n.split(".")[1]), []).append("%s %.1e" % (n.split(".")[-1], e))
    over_per_step[len(over)] += 1
    print("seed %3d: %d event(s) in %d layer(s); down_proj gradients over %.0e in %d layer(s) %s%s" % (
        seed, len(ev), len(layers), BOUND, len(over), over if over else "", ("  -- " + "; ".join(ev)) if ev else ""), flush=True)
print("%s, B=%d, %d seeds: LAYERS WHOSE down_proj GRADIENT EXCEEDS %.0e PER STEP (what the ReLU-side rule's budget must cover): %s" % (
    prec, B, SEEDS, BOUND, dict(sorted(over_per_step.items()))))
print("%s, B=%d, %d seeds: layers with a ReLU-side difference per step: %s; events per step: %s; largest |reference pre-activation| of a differing unit %.1e" % (
    prec, B, SEEDS, dict(sorted(layers_per_step.items())), dict(sorted(events_per_step.items())), worst_pre))
