"""Two streams beside the main one, or one?  Interleaved A/B on one box of the engine's default (tower stream + side stream) against the
weight-gradient chain enqueued on the TOWER stream (one extra stream: the tower stream is idle during the backward, the side stream during
the forward).  python tools/probe/stream_merge_ab.py [cfgA|cfgB] [pairs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfgA"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
blocks, E, drop = CFG[name]
eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype="f16", seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)
side, tower = eng.side_stream, eng.tower_stream


def set_value(v):
    eng.side_stream = tower if v else side


def block(k=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, input_ready=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


res = {0: [], 1: []}
for v in (0, 1):
    set_value(v)
    block(10)
p0 = None
for rep in range(7):
    for v in (0, 1):
        set_value(v)
        res[v].append(block())
print("%s %d pairs  two streams: %.3f ms   one stream (side = tower): %.3f ms   (all: %s | %s)" % (name, pairs, float(np.median(res[0])), float(np.median(res[1])),
      " ".join("%.3f" % t for t in res[0]), " ".join("%.3f" % t for t in res[1])))
