"""Timing probe: one training step replayed from a HIP graph vs enqueued eagerly (Adam's lr_t is frozen in the captured
graph, so this is a timing experiment only)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from voicemap_amd.engine import HipEncoderEngine
dev = torch.device("cuda", 0)
F, E, pairs = 128, 64, 128
blocks = [(32, F, 4), (3, 2 * F, 2), (3, 3 * F, 2), (3, 4 * F, 2)]
eng = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="bf16", device=dev, seed=1234)
rng = np.random.default_rng(0)
xcat = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).to(dev)
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).to(dev)
pl = eng.plan(2 * pairs, 12000, True)
def step():
    eng.preprocess(pl, xcat, 4, True, pairs)
    eng.forward(pl, pairs, None)
    eng.siamese_head(pl, y, "contrastive")
    eng.backward(pl)
    eng.optimizer_step()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager  %.4f ms" % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print("graph  %.4f ms" % timeit(g.replay))
print("eager  %.4f ms" % timeit(step))
print("graph  %.4f ms" % timeit(g.replay))
