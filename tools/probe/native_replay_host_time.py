"""Host time of a replayed step: the library's own runner (vm_program_run, one call per step) against the Python loop of ctypes calls.
python tools/probe/native_replay_host_time.py [cfgA|cfgB] [pairs]  -> host (enqueue) and wall ms per step, interleaved."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfgB"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
blocks, E, drop = CFG[name]
eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype="f16", seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)
masks = eng.make_drop_masks(2 * pairs)   # fixed masks: the three torch launches that draw them are not part of the program


def block(k=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, input_ready=True, drop_masks=masks)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / k * 1e3, (time.perf_counter() - t0) / k * 1e3


# host time with the device idle in front of every step: what the host needs to enqueue one step
def solo(k=50):
    hs = []
    for _ in range(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, input_ready=True, drop_masks=masks)
        hs.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(hs))


res = {0: [], 1: []}
for v in (0, 1):
    eng.native_replay = bool(v)
    block(20)
for rep in range(5):
    for v in (0, 1):
        eng.native_replay = bool(v)
        res[v].append(block())
so = {}
for v in (0, 1):
    eng.native_replay = bool(v)
    so[v] = solo()
for v, nm in ((0, "python loop of ctypes calls"), (1, "vm_program_run")):
    h, w = np.median(np.array(res[v]), 0)
    print("%s %d pairs  %-28s host %.3f ms  wall %.3f ms per step back to back;  %.3f ms to enqueue one step on an idle device" % (name, pairs, nm, h, w, so[v]))
