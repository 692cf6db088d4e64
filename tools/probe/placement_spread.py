"""Does it matter WHERE the step's buffers lie?  N engines of one process (same seed, same batch: the same work on buffers at different
addresses), timed interleaved.  python tools/probe/placement_spread.py [engines] [pairs] [pad_mb]
pad_mb: a throw-away allocation of that many MB between two engines (shifts everything that follows)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine  # noqa: E402

n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
pad_mb = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
blocks, E, l0 = [(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 12000
g = np.random.default_rng(3)
x = torch.from_numpy(g.normal(0, 0.05, (2 * pairs, 4 * l0)).astype(np.float32)).cuda()
yd = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()


def block(e, k=30):
    pl = e.plan(2 * pairs, l0, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        e.train_step_resident(pl, pairs, yd, "contrastive", raw=x, input_ready=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


engines, pads = [], []
for i in range(n_eng):
    e = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=9)
    block(e, 8)
    engines.append(e)
    if pad_mb > 0:
        pads.append(torch.empty(int(pad_mb * (i + 1) * 2 ** 20), dtype=torch.uint8, device="cuda"))
res = [[] for _ in engines]
for rep in range(5):
    for i, e in enumerate(engines):
        res[i].append(block(e))
for i, e in enumerate(engines):
    pl = e.plan(2 * pairs, l0, True)
    addr = " ".join("%s=%x" % (k, pl[b][k].data_ptr()) for b in (1, 2) for k in ("ep", "o", "du", "dp") if k in pl[b])
    print("engine %d: median %.3f ms  (%s)  %s" % (i, float(np.median(res[i])), " ".join("%.3f" % v for v in res[i]), addr))

# which launches differ?  Serial per-entry-point times (HIP events on the launch stream) of the fastest and the slowest engine
if os.environ.get("VM_PLACEMENT_BREAKDOWN", "1") != "0":
    med = [float(np.median(r)) for r in res]
    fast, slow = int(np.argmin(med)), int(np.argmax(med))
    names = ["vm_conv1_fused_fwd", "vm_conv1_fused_bwd", "vm_conv_fwd", "vm_conv_dgrad", "vm_conv_wgrad", "vm_bn_pool_bwd_apply", "vm_bn_pool_bwd_apply_gmax",
             "vm_bn_drop_pool_gmax_partials", "vm_decimate_whiten"]
    rows = {}
    for tag, i in (("fast", fast), ("slow", slow)):
        e = engines[i]
        pl = e.plan(2 * pairs, l0, True)
        e.timed = {nm: [] for nm in names}
        e.overlap_wgrad, e.split_towers = False, False
        for _ in range(4):
            e.train_step_resident(pl, pairs, yd, "contrastive", raw=x, input_ready=True)
        torch.cuda.synchronize()
        for nm in names:
            calls = e.timed[nm]
            per = len(calls) // 4
            for j in range(per):
                ts = [calls[k * per + j][0].elapsed_time(calls[k * per + j][1]) for k in range(1, 4)]
                rows.setdefault((nm, j), {})[tag] = float(np.median(ts))
        e.timed = {}
    print("serial launches, engine %d (fast) against engine %d (slow), us:" % (fast, slow))
    for (nm, j), v in rows.items():
        print("  %-32s #%d  %8.1f  %8.1f   %+5.1f %%" % (nm, j, 1e3 * v["fast"], 1e3 * v["slow"], 100 * (v["slow"] / v["fast"] - 1)))
