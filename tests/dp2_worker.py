"""One rank of the two-process data-parallel GPU test (tests/test_gpu_round6.py::test_two_rank_data_parallel_step_on_one_gpu).
usage: python tests/dp2_worker.py <rank> <world> <port> <out.pt>
Both ranks use cuda:0 (the test boxes have one GPU), torch.distributed over gloo, and the torch.distributed form of the gradient all-reduce
(DYT_NATIVE_RCCL=0: RCCL refuses two ranks on one device).  Everything else is the product's multi-rank path: parameter broadcast at optimizer
construction, a per-rank shard through the fused HIP step, SUM all-reduce of the flat gradient, 1 / world folded into AdamW."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

B, C, R, SEED = 4, 10, 8, 53


def shard(rank):
    import synth
    x, y = synth.make_batch(B, C, seed=SEED + 10 * rank)
    g1, g2 = synth.make_noise(B, seed=SEED + 10 * rank + 1)
    keep = synth.make_dropout_masks(B, R, seed=SEED + 10 * rank + 2)
    return x, y, g1, g2, keep


def build(rank, precision):
    import synth
    import gpu_diag as D
    from models.vision_transformer_IN21K import vit_base_patch16_224_in21k
    sd = synth.make_state_dict(C, R, seed=SEED + 100 * rank, kind="test", gate_bias=0.3)   # ranks start from DIFFERENT trainables: the broadcast must fix that
    sd0 = synth.make_state_dict(C, R, seed=SEED, kind="test", gate_bias=0.3)
    for k in sd:
        if not synth.is_trainable(k):
            sd[k] = sd0[k]                                                                  # (one frozen backbone)
    tuning = D.Cfg(ffn_adapt=True, ffn_option="parallel", ffn_adapter_layernorm_option="none", ffn_adapter_init_option="lora",
                   ffn_adapter_scalar="0.1", ffn_num=R, d_model=768)
    m = vit_base_patch16_224_in21k(num_classes=C, drop_path_rate=0.0, tuning_config=tuning, select_config=D.Cfg(open=True, keep_layers=0),
                                   precision=precision, train_mode="compact", max_batch=B)
    m.load_state_dict(sd, strict=True)
    for n, p in m.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    return m.cuda().train()


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["DYT_NATIVE_RCCL"] = "0"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from engine_finetune import FusedAdamW, train_step
    res = {}
    for precision in ("fp32", "fp16"):
        m = build(rank, precision)
        opt = FusedAdamW(m, lr=1e-3, weight_decay=0.01)          # broadcasts rank 0's trainables
        x, y, g1, g2, keep = shard(rank)
        losses = []
        for i in range(2):
            losses.append(train_step(m, x.cuda(), y.cuda(), opt, gumbel=(g1.cuda(), g2.cuda()), keep_mask=keep.cuda(), target_ratio=0.5,
                                     token_minimal=0.0, token_minimal_weight=0.0).clone())
        torch.cuda.synchronize()
        res[precision] = dict(flat=m._engine.flat.cpu(), losses=torch.stack(losses).cpu())
        del m, opt
        torch.cuda.empty_cache()
    # the epoch loop and the evaluation with two ranks: train_one_epoch over three batches whose last one is ragged (the engine of the smaller batch
    # is a re-created context on BOTH ranks), its end-of-epoch statistics all-reduce, then evaluate() over shards of DIFFERENT lengths (rank r: 5 + r
    # samples -> the ragged all_gather_concat, reference engine_finetune.py:446-480).  Both ranks must return the same epoch statistics and metrics.
    import logging
    import types
    import synth
    import engine_finetune as E
    import misc
    from models.losses import AdaLoss
    m = build(rank, "fp16")
    m_plain = m
    m = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0])     # main_image.py:280-282: the drivers wrap the model, then build the optimizer
    optimizer = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.01)   # (the drivers' optimizer, adopted)
    crit = AdaLoss(base_criterion=torch.nn.CrossEntropyLoss(), token_target_ratio=0.5, token_loss_ratio=2.0, token_minimal=0.0, token_minimal_weight=0.0)
    args = types.SimpleNamespace(accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=4, metric="accuracy", nb_classes=C)
    loader = []
    for i, b in enumerate((B, B, 2)):
        xb, yb = synth.make_batch(b, C, seed=SEED + 1000 + 10 * rank + i)
        loader.append((xb, yb))
    stats = E.train_one_epoch(m, crit, loader, optimizer, torch.device("cuda", 0), 0, misc.NativeScalerWithGradNormCount(), None, None, None,
                              args=args, logger=logging.getLogger("dp2"))
    ev = []
    n_eval = 5 + rank
    xe, ye = synth.make_batch(n_eval, C, seed=SEED + 2000 + rank)
    for i in range(0, n_eval, 3):
        ev.append((xe[i:i + 3], ye[i:i + 3]))
    status = E.evaluate(ev, m, torch.device("cuda", 0), args=args)
    res["epoch"] = dict(stats={k: float(v) for k, v in stats.items()}, status={k: float(v) for k, v in status.items()},
                        flat=m_plain._engine.flat.cpu(), n_eval=n_eval)
    dist.barrier()
    torch.save(res, out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
