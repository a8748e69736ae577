"""world_size-2 tests of the data-parallel glue on CPU (gloo): ragged all_gather_concat
(reference engine_finetune.py:446-480) and the flat-gradient all-reduce + 1/world scaling that
replaces DDP's bucket all-reduce (reference main_image.py:280-282)."""
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
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
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


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "dynamic-tuning_amd"))
    import engine_finetune as E

    class Eng:
        grad = torch.ones(4)
    assert E.allreduce_grads(Eng()) == 1.0 and E.get_world_size() == 1
    t = torch.arange(4.)
    assert E.all_gather_concat(t) is t
