#!/usr/bin/env python
"""BASELINE.json config 4 evidence: the log-mel + 2-D CNN training step (128 pairs of raw 3 s clips, filters 32, embedding 64), timed
per C-ABI entry point with HIP events (serial: side stream off) and, when run under `rocprofv3 --kernel-trace --stats`, per kernel.
    python tools/spectro_profile.py [--dtype f16] [--steps 10] [--breakdown out.csv]
Prints one JSON line with the step time and the algorithmic work of the step."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def algorithmic_work(pairs, F, E, T=298, M=64, esize=2):
    """Layer-granular compulsory traffic (the convention of SURVEY 8d: every block reads its input once, writes its full-resolution
    z once and reads it back once in forward; backward reads pooled dY, z, writes + reads dZ, reads the block input, writes dX) and the
    FLOPs of the four Conv2D 3x3 blocks + the front-end, for 2 * pairs clips."""
    n = 2 * pairs
    chans = [F, 2 * F, 3 * F, 4 * F]
    t, m, cin = T, M, 1
    flops = n * 2.0 * T * (512 * 400 + 256 * M)          # DFT-basis GEMM (re | im) + mel GEMM per frame
    nbytes = n * (48000 * 4 + T * M * esize)              # raw clip in, log-mel out
    for c in chans:
        pos = t * m
        flops += 3 * n * 2.0 * pos * 9 * cin * c          # forward + dgrad + wgrad
        act_in, z, pooled = pos * cin, pos * c, (t // 2) * (m // 2) * c
        nbytes += n * esize * ((act_in + z) + (z + pooled) + (pooled + z + z) + (z + act_in) + (z + act_in))
        t, m, cin = t // 2, m // 2, c
    return flops, nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--breakdown", default="")
    a = ap.parse_args()
    from voicemap_amd.spectro_engine import HipSpectrogramEncoderEngine
    dev = torch.device("cuda", 0)
    pairs = a.pairs
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.normal(0.0, 0.05, size=(2 * pairs, 48000)).astype(np.float32)).to(dev)
    y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).to(dev)
    eng = HipSpectrogramEncoderEngine(32, 64, dropout=0.0, head="uniform_euclidean", dtype=a.dtype, device=dev, seed=1234)
    pl = eng.plan(2 * pairs, 48000, True)

    def step():
        eng.features(pl, x)
        eng.forward(pl, pairs, None)
        eng.siamese_head(pl, y, "contrastive")
        eng.backward(pl)
        eng.optimizer_step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    flops, nbytes = algorithmic_work(pairs, 32, 64, esize=2 if a.dtype in ("bf16", "f16") else 4)
    out = {"config": "log-mel 298 x 64 + Conv2D 3x3 x4 (32-64-96-128) + dense 64, %d pairs of 3 s clips, %s storage" % (pairs, a.dtype),
           "ms_per_step": ms, "audio_s_per_s": 2 * pairs * 3.0 / (ms * 1e-3), "algorithmic_gflop_per_step": flops / 1e9,
           "algorithmic_mb_per_step": nbytes / 1e6, "hbm_frac_of_8TBs": nbytes / (ms * 1e-3) / 8e12,
           "mfma_frac_of_2.5PF": flops / (ms * 1e-3) / 2.5e15, "ms_at_hbm_roofline": nbytes / 8e12 * 1e3}
    if a.breakdown:
        names = ["vm_stft_logmel", "vm_stft_logmel_f16s", "vm_conv2d_first_fwd", "vm_conv2d_first_wgrad", "vm_stack_windows", "vm_bn_pool2d_stack_fwd", "vm_fold_pool_windows_bwd", "vm_bn_bwd_from_sums_finalize", "vm_colsum_strided", "vm_fold_windows", "vm_pool_windows_fwd", "vm_pool_windows_bwd", "vm_clip_max_fwd", "vm_clip_max_bwd",
                 "vm_conv_fwd", "vm_conv_fwd_flat", "vm_conv_dgrad", "vm_conv_wgrad", "vm_bn_finalize", "vm_bn_drop_pool_fwd", "vm_bn_drop_pool_gmax_fwd", "vm_bn_pool_bwd_reduce",
                 "vm_bn_pool_bwd_reduce_pooled", "vm_bn_pool_bwd_reduce_gmax", "vm_bn_bwd_finalize", "vm_bn_pool_bwd_apply", "vm_bn_pool_bwd_apply_gmax", "vm_colsum",
                 "vm_dense_fwd", "vm_dense_bwd", "vm_siamese_head_loss", "vm_grad_sqnorm", "vm_adam_clip_step", "vm_prep_conv_weights", "vm_fill_zero"]
        eng.timed = {nm: [] for nm in names}
        eng.overlap_wgrad = False
        reps = 3
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        rows = [(nm, len(eng.timed[nm]) // reps, sum(e0.elapsed_time(e1) for e0, e1, _ in eng.timed[nm]) / reps) for nm in names]
        eng.timed = {}
        with open(a.breakdown, "w") as f:
            f.write("entry_point,launches_per_step,ms_per_step\n")
            for nm, cnt, tot in sorted(rows, key=lambda r: -r[2]):
                f.write("%s,%d,%.4f\n" % (nm, cnt, tot))
            f.write("TOTAL_EVENT_MS,,%.4f\nWALL_MS_PER_STEP,,%.4f\n" % (sum(r[2] for r in rows), ms))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
