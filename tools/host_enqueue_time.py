"""Host time to enqueue one training step (ctypes calls, events, stream switches) against the GPU time of the step: is the bench
GPU-bound?   python tools/host_enqueue_time.py"""
import time
import numpy as np
import torch
from voicemap_amd.engine import HipEncoderEngine

F, E, pairs = 128, 64, 128
blocks = [(32, F, 4), (3, 2 * F, 2), (3, 3 * F, 2), (3, 4 * F, 2)]
eng = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="bf16", seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)


def step():
    eng.preprocess(pl, x, 4, True, pairs)
    eng.forward(pl, pairs, None)
    eng.siamese_head(pl, y, "contrastive")
    eng.backward(pl, sync_tail=True)
    eng.optimizer_step()


for _ in range(10):
    step()
torch.cuda.synchronize()
for n in (1, 5, 20):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%2d steps: host enqueue %.3f ms/step, until the GPU is done %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
