"""N training steps of one configuration, for rocprofv3 --kernel-trace --stats:  python tools/probe/steps_for_profile.py cfgB 32 [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
name, pairs = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
blocks, E, drop = CFG[name]
eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype=os.environ.get("VM_DTYPE", "f16"), seed=1)
if os.environ.get("VM_SERIAL"):
    eng.split_towers = eng.overlap_wgrad = False
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)
for _ in range(steps):
    eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, input_ready=True)
torch.cuda.synchronize()
print("done", name, pairs, steps)
