"""Interleaved A/B of one engine attribute on one box: python tools/probe/ab_switch.py <attr> <python literal A> <python literal B> [cfg pairs]
prints the median wall ms per step (GPU-synchronised blocks of 40 steps, 7 rounds) for each value."""
import ast
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
attr, va, vb = sys.argv[1], ast.literal_eval(sys.argv[2]), ast.literal_eval(sys.argv[3])
name = sys.argv[4] if len(sys.argv) > 4 else "cfgA"
pairs = int(sys.argv[5]) if len(sys.argv) > 5 else 128
blocks, E, drop = CFG[name]
eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype=os.environ.get("VM_DTYPE", "f16"), seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)


def set_value(v):
    if attr.startswith("tune:"):       # a kernel-selection knob of the library (vm_set_tuning) instead of an engine attribute
        eng.lib.call("vm_set_tuning", attr[5:].encode(), int(v))
    else:
        setattr(eng, attr, v)


def block(k=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, input_ready=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


res = {0: [], 1: []}
for i, v in enumerate((va, vb)):
    set_value(v)
    block(10)
for rep in range(7):
    for i, v in enumerate((va, vb)):
        set_value(v)
        res[i].append(block())
print("%s %d pairs  %s=%r: %.3f ms   %s=%r: %.3f ms   (all: %s | %s)" % (name, pairs, attr, va, float(np.median(res[0])), attr, vb, float(np.median(res[1])),
      " ".join("%.3f" % t for t in res[0]), " ".join("%.3f" % t for t in res[1])))
