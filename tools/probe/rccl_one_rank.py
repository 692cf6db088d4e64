"""RCCL under the data-parallel step on a ONE-GPU box: a process group of one rank over backend "nccl" (= RCCL) with the gradient hook
forced on (GradAllReduce.world = 2 gates the collectives in; the prescale stays 1, and a sum over one rank is the identity).  What runs is
the production path of voicemap_amd/parallel.py -- the early all-reduce of G[conv2.kernel:] on the side stream behind the weight-gradient
GEMMs (async work handle), the late one of the block-1 range inside the optimizer hook, Work.wait() on the step's stream, the barrier with
device_ids, and the recorded-step replay whose host-call slots issue the collectives -- with RCCL's own kernels on the device between
this library's launches.  Prints one JSON line:
  python tools/probe/rccl_one_rank.py [small|cfgA] [pairs] [steps]
``same_bits``: parameters / Adam slots / gradients after the steps equal those of an engine without the hook, bit for bit;
``ms_per_step`` of that engine with the hook on and off, interleaved (the price of two RCCL launches per step where there is nobody to
talk to) and the un-hidden time of the hook (events around it on the step's stream)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))

from voicemap_amd import parallel  # noqa: E402
from voicemap_amd.engine import HipEncoderEngine, _Program  # noqa: E402

CFG = {"small": ([(32, 16, 4), (3, 32, 2), (3, 48, 2), (3, 64, 2)], 16, 1600),
       "cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 12000)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "small"
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    blocks, E, l0 = CFG[name]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.barrier(device_ids=[0])
    out = {"backend": dist.get_backend(), "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "world": dist.get_world_size(),
           "config": name, "pairs": pairs, "steps": steps}
    ea = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=9)
    eb = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=9)
    parallel.attach(ea, 1)
    ea.grad_sync.world = 2          # gate the collectives in; grad_prescale stays 1 / 1
    ea.grad_sync.time_wait = True
    g = np.random.default_rng(3)
    y = np.concatenate([np.zeros(pairs // 2), np.ones(pairs - pairs // 2)])[:, None]
    batches = [(g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32), g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32)) for _ in range(min(steps, 4))]
    for k in range(steps):
        xa, xb = batches[k % len(batches)]
        for e in (ea, eb):
            e.siamese_train_step(xa, xb, y, drop_masks=None)
    torch.cuda.synchronize()
    out["same_bits"] = bool(all(torch.equal(getattr(ea, n).view(torch.int32), getattr(eb, n).view(torch.int32)) for n in ("P", "M", "V", "G")))
    out["collectives"] = ea.grad_sync.collectives
    out["collectives_per_step"] = ea.grad_sync.collectives / steps
    out["replayed_programs"] = sum(isinstance(p_, _Program) for p_ in ea._programs.values())
    waits = [a.elapsed_time(b) for a, b in ea.grad_sync.wait_events]
    out["hook_ms_on_step_stream_median"] = float(np.median(waits)) if waits else None
    ea.grad_sync.time_wait = False
    # timing: resident input, the bench's step
    x = torch.from_numpy(g.normal(0, 0.05, (2 * pairs, 4 * l0)).astype(np.float32)).cuda()
    yd = torch.as_tensor(y, dtype=torch.float32).reshape(pairs).cuda()

    def block(e, k=30):
        pl = e.plan(2 * pairs, l0, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            e.train_step_resident(pl, pairs, yd, "contrastive", raw=x, input_ready=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3

    # the SAME engine with the hook on and off, interleaved (two engines of one process differ by ~1 % from where their buffers lie)
    hook, t = ea.grad_sync, {0: [], 1: []}
    for v in (0, 1):
        ea.grad_sync = hook if v else None
        block(ea, 10)
    for _ in range(5):
        for v in (0, 1):
            ea.grad_sync = hook if v else None
            t[v].append(block(ea))
    ea.grad_sync = hook
    out["ms_per_step_with_rccl_hook"] = float(np.median(t[1]))
    out["ms_per_step_without"] = float(np.median(t[0]))
    out["finite"] = bool(torch.isfinite(ea.P).all().item())
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
