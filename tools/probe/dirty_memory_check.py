"""Does a step read memory it never wrote?  Fill the device allocator's pool with NaN (fp32 and half patterns), free it, build the engine on
the recycled bytes, train:  python tools/probe/dirty_memory_check.py [steps]   (knobs through VOICEMAP_TUNE)."""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

warnings.simplefilter("ignore")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
junk = [torch.full((1 << 28,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(12)]   # 12 GB of NaN
junk.append(torch.full((1 << 29,), float("nan"), dtype=torch.float16, device="cuda"))
torch.cuda.synchronize()
del junk   # back to torch's caching allocator: the engine's torch.empty buffers come out of these bytes
blocks = [(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)]
eng = HipEncoderEngine(blocks, 64, dropout=0.0, head="uniform_euclidean", dtype=os.environ.get("VM_DTYPE", "f16"), seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (256, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(64), torch.ones(64)]).cuda()
pl = eng.plan(256, 12000, True)
for s in range(steps):
    eng.train_step_resident(pl, 128, y, "contrastive", raw=x, input_ready=True)
torch.cuda.synchronize()
print("after %d steps on NaN-recycled memory: loss %.6f  skipped %d  G finite %s" % (steps, float(pl["loss_acc"][0].item()), eng.skipped_steps(),
      bool(torch.isfinite(eng.G).all())))
