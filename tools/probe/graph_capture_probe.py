"""A hipGraph of the whole training step against the recorded-step replay (engine._Program): the eager step is captured through
torch.cuda.graph (three streams: the tower and side streams join the capture through the engine's own event calls) and replayed.
Timing only -- the by-value scalars of a step (Adam's lr_t, the BatchNorm zero-debias factor) stay what they were at capture.
  python tools/probe/graph_capture_probe.py [cfgA|cfgB] [pairs]
Round 6 (profiles/r06_hipgraph_probe.txt): the graph halves the HOST time of a step and the device runs it SLOWER than the same launches
enqueued one by one -- cfg-B 32 pairs 0.461 -> 0.507 ms, cfg-A 32 pairs 0.862 -> 0.931 ms: not adopted."""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from voicemap_amd.engine import HipEncoderEngine
name = sys.argv[1] if len(sys.argv) > 1 else "cfgB"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
CFG = {"cfgA": ([(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 0.0), "cfgB": ([(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 0.05)}
blocks, E, drop = CFG[name]
eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype="f16", seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)
masks = eng.make_drop_masks(2 * pairs)

def timeit(f, k=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3

print("replayed program: host %.3f wall %.3f ms" % timeit(lambda: eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, input_ready=True, drop_masks=masks)))
eng.replay = False
eng.pre_overlap = False
eng.loss_scaled_poll = False
print("eager, no pre_overlap: host %.3f wall %.3f ms" % timeit(lambda: eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, drop_masks=masks)))
poll = eng._poll_loss_scale
eng._poll_loss_scale = lambda: None
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
try:
    with torch.cuda.graph(g, stream=s):
        eng.train_step_resident(pl, pairs, y, "contrastive", raw=x, drop_masks=masks)
        cur = torch.cuda.current_stream()
        for st in (eng.side_stream, eng.tower_stream, eng.misc_stream):
            with torch.cuda.stream(st):
                cap = torch.cuda.is_current_stream_capturing()
            if cap:
                cur.wait_stream(st)
    print("captured")
    print("graph replay: host %.3f wall %.3f ms" % timeit(lambda: g.replay()))
    p0 = eng.P.clone()
    g.replay(); torch.cuda.synchronize()
    print("params move under replay:", bool((eng.P != p0).any().item()), "finite:", bool(torch.isfinite(eng.P).all().item()))
except Exception as e:
    import traceback; traceback.print_exc()
