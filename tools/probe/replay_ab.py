"""Interleaved A/B on one box: the training step eager vs replayed (engine._Program), ms per step (wall, GPU-synchronised blocks)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
for name, pairs in (("cfgA", 128), ("cfgA", 64), ("cfgA", 32), ("cfgB", 32), ("cfgB", 128)):
    blocks, E, drop = CFG[name]
    eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype="f16", seed=1)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
    y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
    pl = eng.plan(2 * pairs, 12000, True)

    def block(k=40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            eng.train_step_resident(pl, pairs, y, "contrastive", raw=x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3, (t1 - t0) / k * 1e3
    res = {True: [], False: []}
    for mode in (True, False):
        eng.replay = mode
        block(10)
    for rep in range(5):
        for mode in (True, False):
            eng.replay = mode
            res[mode].append(block())
    fmt = lambda v: "wall %.3f (host %.3f)" % (float(np.median([a for a, _ in v])), float(np.median([b for _, b in v])))
    print("%s %3d pairs   replay %s   eager %s" % (name, pairs, fmt(res[True]), fmt(res[False])), flush=True)
    del eng, pl
    torch.cuda.empty_cache()
