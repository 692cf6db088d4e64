"""Config 4 (log-mel + 2-D CNN, 128 pairs of 3 s clips, f16 storage): what the two-plane image and the two-plane block-1 output cost, interleaved on one
box: python tools/probe/config4_planes_ab.py  -> ms per step for (split_image, split_z) in (0,0) (1,0) (1,1)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.spectro_engine import HipSpectrogramEncoderEngine  # noqa: E402

pairs = 128
g = np.random.default_rng(1)
x = torch.from_numpy(g.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs // 2)]).cuda()
eng = HipSpectrogramEncoderEngine(32, 64, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=1234)
pl = eng.plan(2 * pairs, 48000, True)


def step():
    eng.features(pl, x)
    eng.forward(pl, pairs, None)
    eng.siamese_head(pl, y, "contrastive")
    eng.backward(pl)
    eng.optimizer_step()


def block(k=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


modes = [(False, False), (True, False), (True, True)]
res = {m: [] for m in modes}
for m in modes:
    eng.split_image, eng.split_z = m
    block(5)
for _ in range(5):
    for m in modes:
        eng.split_image, eng.split_z = m
        res[m].append(block())
for m in modes:
    print("split_image=%d split_z=%d: %.3f ms  (%s)" % (m[0], m[1], float(np.median(res[m])), " ".join("%.3f" % v for v in res[m])))
