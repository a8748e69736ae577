"""world_size-2 tests of the data-parallel glue on CPU (gloo): ragged all_gather_concat
(reference engine_finetune.py:446-480) and the flat-gradient all-reduce + 1/world scaling that
replaces DDP's bucket all-reduce (reference main_image.py:280-282), and the whole update leg (parameter broadcast,
per-shard gradients -> all-reduce -> AdamW) against a per-shard oracle mean."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import engine_finetune as E
    out = {}
    # ragged gather: rank r contributes r+2 rows
    t = torch.arange((rank + 2) * 3, dtype=torch.float32).reshape(rank + 2, 3) + 100 * rank
    g = E.all_gather_concat(t)
    out["gather_shape"] = tuple(g.shape)
    out["gather_ok"] = bool(torch.equal(g[:2], torch.arange(6.).reshape(2, 3)) and torch.equal(g[2:], torch.arange(9.).reshape(3, 3) + 100))
    # flat gradient all-reduce: per-shard gradients -> mean
    class Eng:
        pass
    e = Eng()
    e.grad = torch.full((1000,), float(rank + 1))
    scale = E.allreduce_grads(e)
    out["scale"] = scale
    out["mean_ok"] = bool(torch.allclose(e.grad * scale, torch.full((1000,), 1.5)))
    out["world"] = E.get_world_size()
    # the update leg of train_step (all-reduce of per-shard gradients + FusedAdamW.step, incl. the construction-time
    # parameter broadcast DDP does, main_image.py:280-282) on a CPU stand-in for the engine: the flat-buffer AdamW is the
    # oracle's restatement instead of the HIP kernel, everything else is the product's host code
    from oracle import dyt_oracle as O

    class Head(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.zeros(3, 4))
            self.bias = torch.nn.Parameter(torch.zeros(3))

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = Head()
            self._engine = None

    class CpuEngine:
        n_train, device = 15, torch.device("cpu")

        def __init__(self, rank):
            g = torch.Generator().manual_seed(5 + rank)          # ranks start from DIFFERENT parameters
            self.flat = torch.randn(15, generator=g)
            self.grad = torch.zeros(15)

        def trainable_slice(self, name):
            return (0, 12) if name == "head.weight" else (12, 3)

        def adamw(self, m, v, step, lr, wd, b1, b2, eps, grad_scale):
            p, m2, v2 = O.adamw_update(self.flat, self.grad * grad_scale, m, v, step, lr, wd, b1, b2, eps)
            self.flat.copy_(p); m.copy_(m2); v.copy_(v2)

        def adamw_guarded(self, m, v, state, lr, wd, b1, b2, eps, grad_scale):   # dyt_adamw_guarded's contract on the CPU stand-in
            if not bool(torch.isfinite(self.grad).all()):
                state[1] += 1; state[2] = 1
                return
            state[2] = 0
            self.adamw(m, v, int(state[0]) + 1, lr, wd, b1, b2, eps, grad_scale)
            state[0] += 1

    model = Model()
    model._engine = eng = CpuEngine(rank)
    opt = E.FusedAdamW(model, lr=1e-2, weight_decay=0.1)
    shard_grads = [torch.arange(15.) * 0.1 + 1.0, -torch.arange(15.) * 0.3 + 0.5]     # per-shard gradients of the two ranks
    for step in range(2):
        opt.sync_parameters(eng)                                   # first call: broadcast rank 0's parameters
        eng.grad.copy_(shard_grads[rank] * (step + 1))
        scale = E.allreduce_grads(eng)
        opt.step(grad_scale=scale)
    out["flat"] = eng.flat.clone()
    out["opt_sd_step"] = int(opt.state_dict()["state"][1]["step"])
    # an overflowed step on ONE rank: the SUM all-reduce carries the NaN to every rank, every rank skips the update (GradScaler.step)
    eng.grad.copy_(shard_grads[rank])
    if rank == 1:
        eng.grad[3] = float("inf")
    scale = E.allreduce_grads(eng)
    opt.step(grad_scale=scale)
    out["flat_after_overflow"] = eng.flat.clone()
    out["applied_skipped"] = opt.applied_and_skipped()
    out["sd_step_after_overflow"] = int(opt.state_dict()["state"][1]["step"])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:   # a port that is free right now (a pid-derived one collided with a lingering listener once)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r]["gather_shape"] == (5, 3) and res[r]["gather_ok"]
        assert res[r]["scale"] == 0.5 and res[r]["mean_ok"] and res[r]["world"] == 2
    # replicas stay identical, and equal the oracle's AdamW on the MEAN of the per-shard gradients from rank 0's start
    sys.path.insert(0, ROOT)
    from oracle import dyt_oracle as O
    p = torch.randn(15, generator=torch.Generator().manual_seed(5))
    m, v = torch.zeros(15), torch.zeros(15)
    for step in range(2):
        gmean = 0.5 * ((torch.arange(15.) * 0.1 + 1.0) + (-torch.arange(15.) * 0.3 + 0.5)) * (step + 1)
        p, m, v = O.adamw_update(p, gmean, m, v, step + 1, 1e-2, 0.1)
    assert torch.equal(res[0]["flat"], res[1]["flat"])
    assert float((res[0]["flat"] - p).abs().max()) < 1e-6
    assert res[0]["opt_sd_step"] == 2
    for r in (0, 1):
        assert torch.equal(res[r]["flat_after_overflow"], res[r]["flat"]) and res[r]["applied_skipped"] == (2, 1)
        assert res[r]["sd_step_after_overflow"] == 2


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
    import engine_finetune as E

    class Eng:
        grad = torch.ones(4)
    assert E.allreduce_grads(Eng()) == 1.0 and E.get_world_size() == 1
    t = torch.arange(4.)
    assert E.all_gather_concat(t) is t


# ------------------------------------------------------------------------------------------------------------------
# the loops' own collectives: train_one_epoch's end-of-epoch statistics all-reduce (reference engine_finetune.py:104,
# misc.py:42-53) and evaluate's three ragged all_gather_concat calls (reference :245-248), world 2 on CPU.  The device step /
# the model forward are stand-ins (there is no CPU path of the HIP library); everything between them is the product's host code.
# ------------------------------------------------------------------------------------------------------------------
def _eval_data(rank):
    """rank r holds 2 + r batches of different sizes: 5 + 3 images (rank 0), 4 + 2 + 1 (rank 1); 6 classes"""
    g = torch.Generator().manual_seed(40 + rank)
    sizes = [5, 3] if rank == 0 else [4, 2, 1]
    return [(torch.randn(n, 3, 8, 8, generator=g), torch.randint(0, 6, (n,), generator=g)) for n in sizes]


class _EvalModel(torch.nn.Module):
    """deterministic stand-in for the DyT model: logits and a {0,1} token mask derived from the pixels"""

    def forward(self, x):
        f = x.flatten(1)
        logits = torch.stack([f[:, i::6].sum(1) for i in range(6)], dim=1)
        ts = (f[:, : 12 * 196 % f.shape[1] or f.shape[1]].mean(1, keepdim=True) > 0).float().reshape(-1, 1, 1, 1).expand(-1, 12, 196, 1)
        return logits, {"token_select": ts.contiguous()}


def _loop_worker(rank, world, port, q):
    import types
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import engine_finetune as E
    out = {}
    # ---- evaluate(): ragged shards -> gathered predictions / targets / masks -> one accuracy on every rank ----
    for metric in ("accuracy", "mean_per_class_acc"):
        st = E.evaluate(_eval_data(rank), _EvalModel(), torch.device("cpu"), None, None, None,
                        types.SimpleNamespace(metric=metric, nb_classes=6))
        out["eval_" + metric] = st
    # ---- train_one_epoch(): per-rank loss statistics -> mean over the ranks at the end of the epoch ----
    class Head(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.zeros(3, 4))
            self.bias = torch.nn.Parameter(torch.zeros(3))

    class Model(torch.nn.Module):
        train_mode = "masked"

        def __init__(self):
            super().__init__()
            self.head = Head()
            self._engine = None

    class CpuEngine:
        n_train, device = 15, torch.device("cpu")

        def __init__(self):
            self.flat, self.grad = torch.zeros(15), torch.zeros(15)

        def trainable_slice(self, name):
            return (0, 12) if name == "head.weight" else (12, 3)

    model = Model()
    model._engine = CpuEngine()
    opt = E.FusedAdamW(model, lr=1e-2)
    calls = []

    def fake_step(model, samples, targets, optimizer, criterion=None, losses_out=None, seed=0, **kw):
        calls.append((int(seed), kw["accumulate"], kw["update"]))
        it = len(calls)                                                    # loss components that depend on rank and iteration
        losses_out.copy_(torch.tensor([10.0 * rank + it, 1.0 + rank, 2.0, 3.0 * (rank + 1), 0.5, 0.7 - 0.1 * rank, 0.0, 0.0]))
        return losses_out
    E.train_step = fake_step
    from models.losses import AdaLoss
    crit = AdaLoss(torch.nn.CrossEntropyLoss(), token_target_ratio=0.5, token_loss_ratio=2.0)
    args = types.SimpleNamespace(accum_iter=2, lr=1e-2, min_lr=0.0, warmup_epochs=0, epochs=4)
    loader = [(torch.zeros(2, 3, 4, 4), torch.zeros(2, dtype=torch.int64)) for _ in range(4)]
    stats = E.train_one_epoch(model, crit, loader, opt, torch.device("cpu"), 1, None, 0, None, None, args=args)
    out["stats"] = stats
    out["calls"] = calls
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_loop_collectives_world2():
    import socket
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
    from util.metrics import accuracy, mean_per_class_accuracy
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # evaluate: what one process computes on the concatenation of both shards (rank order), on both ranks
    data = _eval_data(0) + _eval_data(1)
    x = torch.cat([b[0] for b in data]); y = torch.cat([b[1] for b in data])
    logits, aux = _EvalModel()(x)
    a1, a5 = accuracy(logits, y, topk=(1, 5))
    for r in (0, 1):
        st = res[r]["eval_accuracy"]
        assert abs(st["metric"] - float(a1)) < 1e-6 and abs(st["acc5"] - float(a5)) < 1e-6, (r, st)
        assert abs(st["keep_ratio"] - float(aux["token_select"].mean())) < 1e-6
        assert abs(res[r]["eval_mean_per_class_acc"]["metric"] - float(mean_per_class_accuracy(logits, y, 6))) < 1e-5
    # train_one_epoch: per-rank means over 4 iterations, then the mean over the two ranks (reference :104)
    want = {"loss": (sum(range(1, 5)) / 4 + (10 + sum(range(1, 5)) / 4)) / 2, "base_loss": 1.5, "token_loss": 2.0,
            "teacher_loss": 4.5, "distillation_loss": 0.5}
    for r in (0, 1):
        for k, v in want.items():
            assert abs(res[r]["stats"][k] - v) < 1e-9, (r, k, res[r]["stats"][k], v)
        assert [c[1:] for c in res[r]["calls"]] == [(False, False), (True, True), (False, False), (True, True)]   # accum_iter = 2
        assert len({c[0] for c in res[r]["calls"]}) == 4                                                            # a fresh seed per step
