"""Does a long run on ONE batch (the power / convergence probes' loop) skip optimizer steps?  Prints loss, loss scale and the skipped-step
count every 100 steps:  python tools/probe/long_run_skips.py [steps] [cfgA|cfgB] [pairs]"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

warnings.simplefilter("ignore")
CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
name = sys.argv[2] if len(sys.argv) > 2 else "cfgA"
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 128
blocks, E, drop = CFG[name]
eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype=os.environ.get("VM_DTYPE", "f16"), seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)
for s in range(steps):
    eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, input_ready=True)
    if (s + 1) % int(os.environ.get("SYNC_EVERY", "100")) == 0:
        torch.cuda.synchronize()
        print("step %5d  loss %.6f  loss_scale 2^%.0f  skipped %d" % (s + 1, float(pl["loss_acc"][0].item()), np.log2(eng.loss_scale), eng.skipped_steps()), flush=True)
