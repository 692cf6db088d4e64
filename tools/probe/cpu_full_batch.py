import sys, time, os
sys.path.insert(0, os.getcwd())
from oracle import voicemap_oracle as O
import torch
arch = O.EncoderArch.baseline(128, 64, dropout=0.0)
print("cpus", os.cpu_count())
for th in (32, 64, 128):
    t0=time.time()
    sec, t = O.time_cpu_train_steps(arch, 128, 1, threads=th, budget_s=5.0)
    print("128 pairs threads", th, "sec/step", round(sec,2), "wall", round(time.time()-t0,1), flush=True)
for th in (16, 32):
    sec, t = O.time_cpu_train_steps(arch, 8, 2, threads=th, budget_s=5.0)
    print("8 pairs threads", th, "sec/step", round(sec,3), flush=True)
