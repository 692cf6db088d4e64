"""The N > 1 path on CPU: world_size-2 gloo processes, pairs sharded by rank, ONE all-reduce of the flat gradient buffer,
then the identical clip + Adam update on every replica (voicemap_amd/parallel.py).  The CPU oracle stands in for the HIP
engine here (no GPU in this container); what is tested is the data-parallel logic itself."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voicemap_amd import parallel


def test_shard_range_partitions_exactly():
    for n in (1, 7, 128, 1023):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import voicemap_oracle as O
    torch.set_num_threads(1)
    r_, w_, _ = parallel.init_distributed("gloo")
    assert (r_, w_) == (rank, world)
    arch = O.EncoderArch.baseline(8, 8, dropout=0.0)
    p = O.init_params(arch, seed=3)
    g = np.random.default_rng(5)
    pairs = 4
    x1 = O.whiten(g.normal(0, 0.05, (pairs, 400, 1)))
    x2 = O.whiten(g.normal(0, 0.05, (pairs, 400, 1)))
    y = np.array([[0.0], [1.0], [0.0], [1.0]])
    lo, hi = parallel.shard_range(pairs, rank, world)
    res = O.siamese_train_step(arch, p, None, torch.tensor(x1[lo:hi]), torch.tensor(x2[lo:hi]), torch.tensor(y[lo:hi]))
    names = O.param_names(arch)
    flat = torch.cat([res["grads"][k].reshape(-1) for k in names]).to(torch.float32)
    local = flat.clone()

    class FakeEngine:
        pass
    eng = FakeEngine()
    parallel.attach(eng, world)
    eng.grad_sync(flat)               # the one collective of a training step
    avg = flat * eng.grad_prescale
    # gather every rank's local gradient to check the sum, and the averaged result for bit-equality across ranks
    locals_ = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(locals_, local)
    avgs = [torch.zeros_like(avg) for _ in range(world)]
    dist.all_gather(avgs, avg)
    if rank == 0:
        ok_sum = torch.allclose(flat, sum(locals_), rtol=1e-6, atol=1e-9)
        ok_same = all(torch.equal(avgs[0], a) for a in avgs)
        t = parallel.max_over_ranks(1.0 + rank, torch.device("cpu"))
        torch.save({"ok_sum": ok_sum, "ok_same": ok_same, "prescale": eng.grad_prescale, "max": t}, out)
    else:
        parallel.max_over_ranks(1.0 + rank, torch.device("cpu"))
    parallel.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_gloo(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["ok_sum"] and res["ok_same"]
    assert res["prescale"] == 0.5 and res["max"] == 2.0
