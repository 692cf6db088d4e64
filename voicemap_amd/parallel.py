"""Data parallelism for the siamese training step: one process per GPU, pairs sharded across ranks, ONE sum
all-reduce of the flat fp32 gradient buffer per step (RCCL over xGMI; backend "nccl" is RCCL on ROCm), then every
rank applies the identical clip + Adam update to its replica.  BatchNorm statistics stay local to a rank's towers
(SURVEY.md section 8e).  The reference has no multi-GPU code at all -- this is new, not a port.

At cfg-A the buffer is ~4.1 MB: latency-bound, so it is a single collective on the whole buffer (no bucketing).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradAllReduce:
    """engine.grad_sync hook: sum the flat gradient buffer over ranks; the 1/world average is folded into the
    optimizer kernel's prescale so the clip uses the norm of the *averaged* gradient on every rank."""

    def __init__(self, world: int):
        self.world = world

    def __call__(self, flat_grad: torch.Tensor):
        if self.world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)


def attach(engine, world: int):
    engine.grad_sync = GradAllReduce(world)
    engine.grad_prescale = 1.0 / world


def broadcast_state(engine, src: int = 0):
    """Make every replica start from rank `src`'s parameters / Adam slots / moving statistics."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in (engine.P, engine.M, engine.V, engine.NT):
            dist.broadcast(t, src)
        engine.refresh_weights()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
