"""Data parallelism for the siamese training step and the k-way evaluation: one process per GPU (torchrun), pairs / tasks
sharded across ranks.  The reference has no multi-GPU code at all -- this is new, not a port.

Training: every rank draws its own batches (weak scaling: global batch = world x batchsize), BatchNorm statistics stay
local to a rank's towers, and the flat fp32 gradient buffer is summed over ranks (RCCL over xGMI; backend "nccl" is RCCL
on ROCm) in TWO collectives so that the large one hides behind the rest of backward:

  1. ``begin_tail`` -- called by ``HipEncoderEngine.backward`` as soon as the gradients of blocks 2..n, the dense layer and
     the head are final (the wgrad GEMMs run on the engine's side stream, so the collective is enqueued behind THAT stream
     plus an event of the main stream): an asynchronous all-reduce of G[conv2.kernel:] (99.6 % of the buffer at cfg-A)
     that runs while the main stream still computes the block-2 dgrad and the whole block-1 backward;
  2. ``__call__`` (the engine's ``grad_sync`` hook in ``optimizer_step``) -- the all-reduce of the small block-1 head of
     the buffer, then the wait for (1).

The 1/world average is folded into the optimizer kernel's prescale, so the clip uses the norm of the *averaged* gradient
and every replica applies the identical update: replicas stay bit-identical without ever re-broadcasting parameters.
At cfg-A the buffer is ~4.1 MB: latency-bound on xGMI, so there is no bucketing beyond that split.

Evaluation (``voicemap_amd.utils.n_shot_task_evaluation``): tasks are sharded over ranks with ``shard_range`` and
``n_correct`` is summed with ``sum_over_ranks`` -- one integer per evaluation is the only exchange.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: str | None = None, timeout_s: float | None = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).  ``timeout_s`` bounds the
    rendezvous (and, for gloo, every collective): a rank that never shows up is an exception, not a hang."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:   # VOICEMAP_DIST_BACKEND=gloo: rehearsal of the multi-rank code paths on a box with fewer GPUs than ranks
            backend = os.environ.get("VOICEMAP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            if os.environ.get("VOICEMAP_DIST_SHARE_DEVICE"):   # rehearsal on a box with fewer GPUs than ranks (RCCL is expected to refuse)
                local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
        kw = {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def rank_world() -> Tuple[int, int]:
    """(rank, world) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _comm_device(device=None) -> torch.device:
    if device is not None:
        return torch.device(device)
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class GradAllReduce:
    """``engine.grad_sync`` hook (see the module docstring).  ``split_name``: first tensor of the flat buffer that belongs
    to the early collective; everything before it (block 1) is reduced in the hook itself."""

    replayable = True   # engine._train_step may record a step that contains this hook (its collectives are host calls of the program)

    def __init__(self, world: int, overlap: bool = True, split_name: str = "conv2.kernel"):
        self.world = world
        self.overlap = overlap
        self.split_name = split_name
        self._work = None
        self._split = 0
        self.collectives = 0  # issued so far (tests / diagnostics)
        # bench.py: with ``time_wait`` on, every optimizer-step hook is bracketed by two events on the current stream -- the time
        # the step's stream spends in (and waiting for) the collectives, i.e. what the gradient exchange costs that is NOT hidden
        self.time_wait = False
        self.wait_events = []

    # -- called from HipEncoderEngine.backward ------------------------------------------------------------
    def begin_tail(self, engine, main_event=None):
        """Enqueue the all-reduce of G[split:] behind the engine's side stream (weight-gradient GEMMs) and ``main_event``
        (recorded on the main stream after the last main-stream write into that range)."""
        if self.world <= 1 or not self.overlap or self.split_name not in engine.offsets:
            return
        if self._work is not None:  # a backward without an optimizer step in between: finish the old collective first
            self._work.wait()
            self._work = None
        split = engine.offsets[self.split_name][0]
        if split <= 0 or split >= engine.G.numel():
            return
        tail = engine.G[split:]
        side = getattr(engine, "side_stream", None)
        if tail.is_cuda and side is not None:
            if main_event is not None:
                side.wait_event(main_event)
            else:
                side.wait_stream(torch.cuda.current_stream(tail.device))
            with torch.cuda.stream(side):
                self._work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
        self._split = split
        self.collectives += 1

    def ordered_begin(self, engine):
        """``begin_tail`` for a caller that has already ordered the side stream behind the main stream's last write into the range
        (engine._begin_grad_tail: through the engine's own event calls, so that a recorded step replays the ordering)."""
        if self.world <= 1 or not self.overlap or self.split_name not in engine.offsets:
            return
        if self._work is not None:
            self._work.wait()
            self._work = None
        split = engine.offsets[self.split_name][0]
        if split <= 0 or split >= engine.G.numel():
            return
        tail = engine.G[split:]
        if tail.is_cuda:
            with torch.cuda.stream(engine.side_stream):
                self._work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
        self._split = split
        self.collectives += 1

    # -- called from HipEncoderEngine.optimizer_step ------------------------------------------------------
    def __call__(self, flat_grad: torch.Tensor):
        if self.world <= 1:
            return
        if self.time_wait and flat_grad.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._reduce(flat_grad)
            e1.record()
            self.wait_events.append((e0, e1))
            return
        self._reduce(flat_grad)

    def _reduce(self, flat_grad: torch.Tensor):
        if self._work is None:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
            self.collectives += 1
            return
        dist.all_reduce(flat_grad[:self._split], op=dist.ReduceOp.SUM)
        self.collectives += 1
        self._work.wait()  # the current stream waits for the early collective
        self._work = None


def attach(engine, world: int, overlap: bool = True):
    engine.grad_sync = GradAllReduce(world, overlap=overlap)
    engine.grad_prescale = 1.0 / world
    engine.sync_bn_world = world   # used only where engine.sync_bn is set (SyncBN: BatchNorm over the global batch, engine.py)


def attach_if_distributed(engine) -> Tuple[int, int]:
    """What ``fit_generator`` calls: under torchrun, hook the gradient sum into the engine and start every replica from rank
    0's state.  Idempotent."""
    rank, world = rank_world()
    if world > 1 and not isinstance(getattr(engine, "grad_sync", None), GradAllReduce):
        attach(engine, world)
        broadcast_state(engine)
    return rank, world


def broadcast_state(engine, src: int = 0):
    """Make every replica start from rank `src`'s parameters / Adam slots / moving statistics / step counter."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in (engine.P, engine.M, engine.V, engine.NT):
            dist.broadcast(t, src)
        if hasattr(engine, "ZD"):   # zero-debias accumulators of the BatchNorm moving statistics + their step counter
            dist.broadcast(engine.ZD, src)
        it = torch.tensor([int(engine.iterations), int(getattr(engine, "bn_steps", 0))], dtype=torch.int64, device=engine.P.device)
        dist.broadcast(it, src)
        engine.iterations = int(it[0].item())
        if hasattr(engine, "ZD"):
            engine.bn_steps = int(it[1].item())
        engine.refresh_weights()


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    """Sum of one number over ranks (n_correct of the sharded k-way evaluation)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def weighted_mean_logs(sums: Dict[str, float], weight: float, device=None) -> Dict[str, float]:
    """Per-epoch metrics of data-parallel training: ``sums[k]`` = sum over this rank's samples of metric k, ``weight`` = this
    rank's sample count; returns the global sample-weighted means -- the SAME numbers on every rank, so callbacks that act on
    them (ReduceLROnPlateau, ModelCheckpoint's best-so-far) decide identically everywhere."""
    keys = sorted(sums)
    vals = [float(sums[k]) for k in keys] + [float(weight)]
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor(vals, dtype=torch.float64, device=_comm_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        vals = t.tolist()
    w = max(vals[-1], 1.0)
    return {k: v / w for k, v in zip(keys, vals[:-1])}


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":   # name the device: no guess from the rank number at the communicator's first use
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
