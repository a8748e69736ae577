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


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
    import engine_finetune as E

    class Eng:
        grad = torch.ones(4)
    assert E.allreduce_grads(Eng()) == 1.0 and E.get_world_size() == 1
    t = torch.arange(4.)
    assert E.all_gather_concat(t) is t
