"""Probe: does running the two towers (two half batches) on two streams, one kernel apart, beat one full-batch pass?
Training-mode forward + backward (no optimizer) of 128-window half batches on two engines / two streams vs one 256-window batch."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from voicemap_amd.engine import HipEncoderEngine
dev = torch.device("cuda", 0)
F, E = 128, 64
blocks = [(32, F, 4), (3, 2 * F, 2), (3, 3 * F, 2), (3, 4 * F, 2)]
def mk():
    return HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="bf16", device=dev, seed=1234)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (256, 48000)).astype(np.float32)).to(dev)
full = mk(); a = mk(); b = mk()
plf = full.plan(256, 12000, True); pla = a.plan(128, 12000, True); plb = b.plan(128, 12000, True)
def run(eng, pl, xs, wpt):
    eng.preprocess(pl, xs, 4, True, wpt)
    eng.forward(pl, wpt, None)
    pl["demb"].fill_(0.01)
    eng.backward(pl)
def t(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one(): run(full, plf, x, 128)
def seq():
    run(a, pla, x[:128], 128); run(b, plb, x[128:], 128)
def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): run(a, pla, x[:128], 128)
    with torch.cuda.stream(s2): run(b, plb, x[128:], 128)
    cur.wait_stream(s1); cur.wait_stream(s2)
for name, fn in (("one 256-window pass", one), ("two 128-window passes, same stream", seq), ("two 128-window passes, two streams", par)):
    print("%-40s %.3f ms" % (name, t(fn)))
