"""Parity of one fused step per arithmetic mode against the CPU oracle on the same seeded inputs / draws (B = 16 = BASELINE.json
configs[0], reference-style masked training): logits, gate decisions, losses, gradient error per tensor kind -- the rows of
DESIGN.md section 3.  PB=16 PMODES=fp32,bf16,fp16 python tools/probes/precision_table.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
import test_gpu_round2 as T
from oracle import dyt_oracle as O
B = int(os.environ.get("PB", "16")); C, r, mode, target = 100, 64, os.environ.get("PTRAIN", "masked"), 0.5
torch.set_num_threads(synth.available_cores())
x, y = synth.make_batch(B, C, seed=31)
g1, g2 = synth.make_noise(B, seed=32)
keep = synth.make_dropout_masks(B, r, seed=33)
sd = synth.make_state_dict(C, r, seed=0, kind="test", gate_bias=0.85)
d_ref, g_ref, (ref_ls, ref_lt, tok) = O.step_grads(sd, x, y, g1, g2, keep, scale=0.1, mode=mode, token_target_ratio=target)
ref_ts = tok["token_select"].detach()[..., 0].float()
rows = {}
for prec in os.environ.get("PMODES", "fp32,bf16,fp16").split(","):
    m, _ = T._bench_model(prec, mode, B, 0.85, classes=C, r=r, kind="test")
    m.train()
    eng = m.engine(B, torch.device("cuda", 0))
    if os.environ.get("PGS") and prec != "fp32":
        import _lib
        eng.set_option(_lib.OPT_GRAD_SCALE_LOG2, int(os.environ["PGS"]))
    ls = torch.empty(B, C, device="cuda"); lt = torch.empty(B, C, device="cuda"); ts = torch.zeros(B, 12, 196, device="cuda")
    losses = eng.step_fwd_bwd(x.cuda(), y.cuda(), target, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                              g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt, token_select=ts).cpu()
    worst = {}
    for n, gr in g_ref.items():
        got = eng.trainable_view(n, gr.shape, eng.grad).cpu()
        kind = n.split(".", 2)[-1]
        if gr.numel() == 1:
            worst.setdefault(kind + " (12-vector)", []).append((float(got), float(gr)))
            continue
        worst[kind] = max(worst.get(kind, 0.0), float((got - gr).norm() / (gr.norm() + 1e-20)))
    for k in [k for k in worst if isinstance(worst[k], list)]:
        a, b = torch.tensor(worst[k], dtype=torch.float64).unbind(1)
        worst[k] = float((a - b).norm() / (b.norm() + 1e-20))
    rows[prec] = {"logits_student_max_abs": float((ls.cpu() - ref_ls.detach()).abs().max()), "logits_teacher_max_abs": float((lt.cpu() - ref_lt.detach()).abs().max()),
                  "gate_flips": int((ts.cpu() != ref_ts).sum()), "decisions": ts.numel(),
                  "loss_abs_err": abs(float(losses[0]) - float(d_ref["loss"])), "nan_in_grad": bool(torch.isnan(eng.grad).any()),
                  "grad_rel_l2_worst_per_kind": {k: round(v, 5) for k, v in sorted(worst.items())}}
    print(prec, json.dumps(rows[prec]), flush=True)
    del m, eng
    torch.cuda.empty_cache()
