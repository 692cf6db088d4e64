"""Where does the host's time per training step go at the reference's own batch sizes?  (cfg-B 32 pairs, cfg-A 64 pairs.)
Prints: host enqueue and wall ms per step, the floor of a ctypes call and of one tiny kernel launch through the C ABI, cProfile's top."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from voicemap_amd import _lib
from voicemap_amd.engine import HipEncoderEngine

which = sys.argv[1] if len(sys.argv) > 1 else "cfgB"
if which == "cfgB":
    blocks, E, pairs, drop = [(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)], 128, 32, 0.05
else:
    blocks, E, pairs, drop = [(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)], 64, 64, 0.0
eng = HipEncoderEngine(blocks, E, dropout=drop, head="uniform_euclidean", dtype="f16", seed=1)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (2 * pairs, 48000)).astype(np.float32)).cuda()
y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).cuda()
pl = eng.plan(2 * pairs, 12000, True)


def step():
    eng.train_step_resident(pl, pairs, y, "contrastive", raw=x)


if len(sys.argv) > 2 and sys.argv[2] == "eager":
    eng.replay = False

for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s %d pairs: host enqueue %.3f ms/step, wall %.3f ms/step" % (which, pairs, (t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
calls = []
orig = eng.lib.call


def counting(name, *a):
    calls.append(name)
    return orig(name, *a)


eng.lib.call = counting
step()
eng.lib.call = orig
torch.cuda.synchronize()
print("C-ABI calls per step: %d" % len(calls))
lib = _lib.lib()
f = lib.cdll.vm_abi_version
t0 = time.perf_counter()
for _ in range(20000):
    f()
print("ctypes call, no args: %.2f us" % ((time.perf_counter() - t0) / 20000 * 1e6))
buf = torch.zeros(1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
g = lib.cdll.vm_fill_zero
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    g(buf.data_ptr(), 4096, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("one tiny launch through ctypes (vm_fill_zero 4 KB), host side: %.2f us (2000 back to back, queue not full? wall %.2f us)" %
      ((t1 - t0) / 2000 * 1e6, (time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    buf.data_ptr()
print("tensor.data_ptr(): %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    buf[16:32].data_ptr()
print("tensor[a:b].data_ptr(): %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(2000):
    ev.record()
print("event.record(): %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
s2 = torch.cuda.Stream()
t0 = time.perf_counter()
for _ in range(2000):
    with torch.cuda.stream(s2):
        pass
print("with torch.cuda.stream(s): %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000):
    torch.cuda.current_stream().wait_stream(s2)
print("current.wait_stream(s): %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(200):
    eng.make_drop_masks(2 * pairs)
print("make_drop_masks: %.2f us" % ((time.perf_counter() - t0) / 200 * 1e6))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
