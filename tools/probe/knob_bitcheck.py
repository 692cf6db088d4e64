"""Does a vm_set_tuning knob leave the training step bit-identical?  python tools/probe/knob_bitcheck.py <knob> <value A> <value B> [cfgA|cfgB] [pairs]
Two engines from the same seed run 4 steps on the same batch, one under each value; the parameters and Adam slots are compared bit for bit."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
knob, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
name = sys.argv[4] if len(sys.argv) > 4 else "cfgA"
pairs = int(sys.argv[5]) if len(sys.argv) > 5 else 128
blocks, E, drop = CFG[name]
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
out = []
for v in (va, vb):
    eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype=os.environ.get("VM_DTYPE", "f16"), seed=1)
    eng.lib.call("vm_set_tuning", knob.encode(), v)
    pl = eng.plan(2 * pairs, 12000, True)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    for _ in range(4):
        masks = eng.make_drop_masks(2 * pairs, g) if drop > 0 else None
        eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, drop_masks=masks)
    torch.cuda.synchronize()
    out.append((eng.P.clone(), eng.M.clone(), eng.V.clone(), float(pl["loss_acc"][0].item())))
same = all(torch.equal(a, b) for a, b in zip(out[0][:3], out[1][:3]))
print("%s %d pairs  %s=%d vs %d: loss %.9g / %.9g  parameters + Adam slots bit-identical: %s" % (name, pairs, knob, va, vb, out[0][3], out[1][3], same))
sys.exit(0 if same else 1)
