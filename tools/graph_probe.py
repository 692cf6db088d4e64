"""Timing probe: one training step replayed from a HIP graph vs enqueued eagerly (Adam's lr_t and the BatchNorm zero-debias factor are
kernel arguments, i.e. frozen in the captured graph: a timing experiment only).  PYTHONPATH=$PWD python tools/graph_probe.py [pairs ...]"""
import sys, time
import numpy as np, torch
from voicemap_amd.engine import HipEncoderEngine
dev = torch.device("cuda", 0)
F, E = 128, 64
blocks = [(32, F, 4), (3, 2 * F, 2), (3, 3 * F, 2), (3, 4 * F, 2)]
for pairs in [int(a) for a in sys.argv[1:]] or [8, 64, 128]:
    eng = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="f16", device=dev, seed=1234)
    eng._poll_loss_scale = lambda: None   # host logic (pinned copies, event queries): not capturable, not needed for a timing
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
    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    e1 = timeit(step)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g1 = timeit(g.replay); e2 = timeit(step); g2 = timeit(g.replay)
    print("%3d pairs: eager %.4f / %.4f ms   graph replay %.4f / %.4f ms" % (pairs, e1, e2, g1, g2))
