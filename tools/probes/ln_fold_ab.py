"""A/B of the 16-bit modes with LayerNorm-2 folded into fc1 (default) vs as its own kernel (DYT_LN_FOLD=0), against the CPU oracle at
B=16 over seeds: logits, token-keep decisions that differ, worst gradient rel-L2 per tensor kind.  Run on an MI355X from the repo root:
    python tools/probes/ln_fold_ab.py [fp16|bf16] [seeds...]"""
import os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "dynamic-tuning_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import synth
from oracle import dyt_oracle as O
from test_gpu_round2 import _bench_model

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
seeds = [int(a) for a in sys.argv[2:]] or [31, 41, 51, 61, 71]
B, C, r, target = 16, 100, 64, 0.7
for mode in ("masked", "compact"):
    for seed in seeds:
        x, y = synth.make_batch(B, C, seed=seed)
        g1, g2 = synth.make_noise(B, seed=seed + 1)
        keep = synth.make_dropout_masks(B, r, seed=seed + 2)
        sd = synth.make_state_dict(C, r, seed=0, kind="test", gate_bias=0.85)
        _, g_ref, (ref_ls, ref_lt, tok) = O.step_grads(sd, x, y, g1, g2, keep, scale=0.1, mode=mode, token_target_ratio=target)
        ref_ts = tok["token_select"].detach()[..., 0].float()
        for fold in ("0", "1"):
            os.environ["DYT_LN_FOLD"] = fold
            m, _ = _bench_model(prec, mode, B, 0.85, classes=C, r=r, kind="test")
            m.train()
            eng = m.engine(B, torch.device("cuda", 0))
            ls = torch.empty(B, C, device="cuda"); lt = torch.empty(B, C, device="cuda"); ts = torch.zeros(B, 12, 196, device="cuda")
            eng.step_fwd_bwd(x.cuda(), y.cuda(), target, 2.0, 0.0, 0.0, masked_dense=(mode == "masked"), g1=g1.cuda().contiguous(),
                             g2=g2.cuda().contiguous(), keep_mask=keep.cuda().contiguous(), logits_s=ls, logits_t=lt, token_select=ts)
            worst = {}
            for n, gr in g_ref.items():
                if gr.numel() == 1:
                    continue
                got = eng.trainable_view(n, gr.shape, eng.grad).cpu()
                kind = "gate" if "token_select" in n else ("down" if "down_proj" in n else ("up" if "up_proj" in n else "head"))
                worst[kind] = max(worst.get(kind, 0.0), float((got - gr).norm() / (gr.norm() + 1e-20)))
            print("%s %-7s seed %d fold %s: logits %.2e / %.2e, flips %d, grads %s" % (
                prec, mode, seed, fold, float((ls.cpu() - ref_ls.detach()).abs().max()), float((lt.cpu() - ref_lt.detach()).abs().max()),
                int((ts.cpu() != ref_ts).sum()), {k: "%.1e" % v for k, v in sorted(worst.items())}), flush=True)
            del m, eng
            torch.cuda.empty_cache()
