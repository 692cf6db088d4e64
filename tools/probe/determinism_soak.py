"""Run-to-run determinism of the final tree's default path (replayed by vm_program_run, one stream beside the main one): two engines, same
seed, the same 1 500 batches (cfg-A, 32 pairs, f16 storage, dropout 0.05 so that the mask draws are in it) -- parameters, Adam slots and
moving statistics must be bit-identical at the end, the loss finite throughout.  python tools/probe/determinism_soak.py [steps] [pairs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine, _Program  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
blocks = [(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)]
g = np.random.default_rng(7)
corpus = torch.from_numpy(g.normal(0, 0.05, (64, 2 * pairs, 48000)).astype(np.float32)).cuda()   # 64 different batches, cycled
y = torch.from_numpy((g.random((64, pairs)) > 0.5).astype(np.float32)).cuda()
out = []
for run in range(2):
    torch.manual_seed(11)
    eng = HipEncoderEngine(blocks, 64, dropout=0.05, head="uniform_euclidean", dtype="f16", seed=3)
    pl = eng.plan(2 * pairs, 12000, True)
    t0 = time.perf_counter()
    worst = 0.0
    for k in range(steps):
        eng.train_step_resident(pl, pairs, y[k % 64], "contrastive", raw=corpus[k % 64], input_ready=True)
        if k % 250 == 249:
            l = float(pl["loss_acc"][0].item())
            assert np.isfinite(l), (k, l)
            worst = max(worst, l)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    progs = [p for p in eng._programs.values() if isinstance(p, _Program)]
    out.append((eng.P.clone(), eng.M.clone(), eng.V.clone(), eng.NT.clone(), float(pl["loss_acc"][0].item()), eng.loss_scale, eng.skipped_steps if hasattr(eng, "skipped_steps") else None))
    print("run %d: %d steps in %.2f s (%.3f ms per step), final loss %.6f, loss scale %g, native programs %d" % (
        run, steps, dt, dt / steps * 1e3, out[-1][4], out[-1][5], sum(p.native is not None for p in progs)))
same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(out[0][:4], out[1][:4]))
print("bit-identical parameters / Adam slots / moving statistics after %d steps: %s" % (steps, same))
assert same
