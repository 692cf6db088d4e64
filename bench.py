#!/usr/bin/env python
"""Headline benchmark of the voicemap hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one ``train_on_batch`` of the siamese script at cfg-A (experiments/train_siamese.py:20-25: filters 128,
embedding 64, dropout 0; contrastive loss per BASELINE.json config 2), 128 pairs (256 windows) of 3 s @ 16 kHz per
GPU, bf16 storage / fp32 accumulate: decimate x4 + whiten on the GPU, twin forward, loss, backward, (gradient
all-reduce), global-norm clip + Adam, GEMM-layout weight refresh.  Raw windows are synthetic and resident
in HBM before the timed region.  value = audio-seconds embedded per second = N * 256 windows * 3 s * K / wall time,
wall time = max over ranks between barrier+synchronize brackets.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP events on the launch stream inside the timed
region) and "cpu_baseline" (the CPU oracle's fp32 training step on a bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak
TRAIN_BYTES_PER_WINDOW = 44.02e6   # algorithmic HBM bytes per 3 s window, training, S4k, bf16
TRAIN_FLOPS_PER_WINDOW = 7.373e9


def conv_launch_work(name, args, esize):
    """Algorithmic bytes and FLOPs of one conv entry-point launch (layer-granular compulsory traffic of SURVEY 8d:
    each operand tensor is read once and each result written once; weights excluded).  ``args`` may end with the name of the fused
    form that actually ran (engine._call books vm_conv_fwd_e / vm_conv_dgrad_bnred under the plain entry points): its extra tensor
    -- the pooled extreme written by the forward epilogue, the tensor the dgrad epilogue takes its BatchNorm sums against -- counts."""
    base = {"vm_conv_fwd": 3, "vm_conv_dgrad": 2, "vm_conv_wgrad": 2}[name]
    n, L, cin, cout = args[base:base + 4]
    fused = args[-1] if args and isinstance(args[-1], str) else ""
    shape = {"n_windows": n, "L": L, "c_in": cin, "c_out": cout, "fused": fused}
    extra = {"vm_conv_fwd_e": n * (L // 2) * cout, "vm_conv_dgrad_bnred": n * L * cin}.get(fused, 0)
    return (n * L * (cin + cout) + extra) * esize, 2.0 * n * L * 3 * cin * cout, shape


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=128, help="pairs per GPU (cfg: 128)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--loss", default="contrastive")
    ap.add_argument("--dominant", default="auto", help="entry point whose launches the roofline object describes: auto = the "
                    "GEMM family (vm_conv_fwd / vm_conv_dgrad / vm_conv_wgrad) with the largest share of the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side figures (other loss, embed-only pass)")
    ap.add_argument("--no-overlap-wgrad", action="store_true",
                    help="keep the weight-gradient GEMMs on the main stream (default: side stream, concurrent with dgrad)")
    ap.add_argument("--tune", default="", help="extra tuning knobs key=value,key=value (vm_set_tuning)")
    ap.add_argument("--breakdown", default="", help="write a per-entry-point time breakdown (extra untimed steps) to this file")
    a = ap.parse_args()

    from voicemap_amd import parallel
    from voicemap_amd.engine import HipEncoderEngine
    rank, world, local = parallel.init_distributed()
    assert world == a.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"
    n_gpus = world
    if os.environ.get("VOICEMAP_DIST_BACKEND") == "gloo":   # rehearsal: more ranks than GPUs, the replicas share devices
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    F, E = 128, 64
    blocks = [(32, F, 4), (3, 2 * F, 2), (3, 3 * F, 2), (3, 4 * F, 2)]
    eng = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype=a.dtype, device=dev, seed=1234)
    eng.overlap_wgrad = not a.no_overlap_wgrad
    for kv in [t for t in a.tune.split(",") if t]:
        k, v = kv.split("=")
        if k == "overlap_wgrad":
            eng.overlap_wgrad = bool(int(v))
        elif k == "pooled_reduce":
            eng.pooled_reduce = bool(int(v))
        elif k == "split_towers":
            eng.split_towers = bool(int(v))
        elif k == "fused_bn_reduce":
            eng.fused_bn_reduce = bool(int(v))
        elif k == "wgrad_after_dgrad":
            eng.wgrad_after_dgrad = bool(int(v))
        elif k == "fused_pool_extreme":
            eng.fused_pool_extreme = bool(int(v))
        elif k == "tower_stagger":
            eng.tower_stagger = int(v)
        else:
            eng.lib.call("vm_set_tuning", k.encode(), int(v))
    parallel.attach(eng, n_gpus)
    parallel.broadcast_state(eng)

    # synthetic raw windows, a different shard per rank, resident in HBM
    pairs = a.pairs
    rng = np.random.default_rng(1234 + rank)

    def raw():
        x = rng.normal(0.0, 0.05, size=(pairs, 48000)).astype(np.float32)
        return torch.from_numpy(x + rng.uniform(-0.01, 0.01, size=(pairs, 1)).astype(np.float32)).to(dev)
    x1, x2 = raw(), raw()
    y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).to(dev)
    xcat = torch.cat([x1, x2], 0).contiguous()
    l0 = 12000
    pl = eng.plan(2 * pairs, l0, True)

    def step():
        eng.preprocess(pl, xcat, 4, True, pairs)
        eng.forward(pl, pairs, None)
        eng.siamese_head(pl, y, a.loss)
        eng.backward(pl, sync_tail=True)   # N > 1: the large gradient all-reduce starts before block 1's backward
        eng.optimizer_step()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    # Inside the timed region only the forward GEMMs are event-bracketed when the weight-gradient GEMMs run on the side stream
    # (the backward kernels then overlap and their durations are not attributable); all three GEMM families are attributed by
    # a SERIAL pass after the timed region (below) and the roofline object describes the family that is largest there.
    families = ["vm_conv_fwd"] if eng.overlap_wgrad else ["vm_conv_fwd", "vm_conv_dgrad", "vm_conv_wgrad"]
    eng.timed = {nm: [] for nm in families}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)
    in_region = {nm: sum(e0.elapsed_time(e1) for e0, e1, _ in eng.timed[nm]) / a.steps for nm in families}
    eng.timed = {}
    loss = float(pl["loss_acc"][0].item())
    # a tuning set may switch kernels, never results: a non-finite loss is an error there too, except for the ablation switches of
    # -DVM_ENABLE_ABLATION builds (wrong results by design; note that corrupted tensors also make every later launch FASTER -- the
    # chip clocks higher on NaN / zero operands -- so their timings overstate what removing the ablated part would save)
    assert np.isfinite(loss) or "ablate" in a.tune, "training diverged"

    windows = 2 * pairs * n_gpus * a.steps
    value = windows * 3.0 / dt
    ms = dt / a.steps * 1e3

    # ---- serial attribution pass (untimed, after the timed region): the same step with the weight-gradient GEMMs on the main
    # stream, every GEMM launch bracketed by HIP events on its launch stream; median over the repetitions per launch shape ----
    esize = 2 if a.dtype in ("bf16", "f16") else 4
    gemm = ["vm_conv_fwd", "vm_conv_dgrad", "vm_conv_wgrad"]
    was_overlap, eng.overlap_wgrad = eng.overlap_wgrad, False
    was_split, eng.split_towers = eng.split_towers, False   # one launch per GEMM of the step, nothing else in flight
    snap0 = (eng.P.clone(), eng.M.clone(), eng.V.clone(), eng.NT.clone(), eng.iterations, eng.ZD.clone(), eng.bn_steps)
    eng.timed = {nm: [] for nm in gemm}
    att_reps = max(5, min(a.steps, 20))
    for _ in range(att_reps):
        step()
    torch.cuda.synchronize()
    fam = {}
    for nm in gemm:
        by_shape = {}
        for e0, e1, args in eng.timed[nm]:
            nb_, nf_, shp = conv_launch_work(nm, args, esize)
            by_shape.setdefault(tuple(shp.values()), []).append(e0.elapsed_time(e1) * 1e-3)
        launches = []
        for key, ts in by_shape.items():
            t_med = float(np.median(ts))
            nb_, nf_, shp = conv_launch_work(nm, (None,) * {"vm_conv_fwd": 3, "vm_conv_dgrad": 2, "vm_conv_wgrad": 2}[nm] + key[:4] + (key[4],), esize)
            launches.append({"shape": shp, "ms": t_med * 1e3, "tflops": nf_ / t_med / 1e12, "gbs": nb_ / t_med / 1e9,
                             "algorithmic_bytes": nb_, "flops": nf_})
        fam[nm] = {"ms_per_step": sum(l["ms"] for l in launches), "launches": launches}
    eng.timed = {}
    eng.overlap_wgrad = was_overlap
    eng.split_towers = was_split
    eng.P.copy_(snap0[0]); eng.M.copy_(snap0[1]); eng.V.copy_(snap0[2]); eng.NT.copy_(snap0[3]); eng.iterations = snap0[4]
    eng.ZD.copy_(snap0[5]); eng.bn_steps = snap0[6]
    eng.refresh_weights()
    if a.dominant == "auto":
        a.dominant = max(fam, key=lambda k: fam[k]["ms_per_step"])
    worst = max(fam[a.dominant]["launches"], key=lambda l: l["ms"])
    t_avg = worst["ms"] * 1e-3
    nbytes, nflops, shape = worst["algorithmic_bytes"], worst["flops"], worst["shape"]
    ai = nflops / nbytes
    ridge = MFMA_BF16_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
    if ai < ridge:
        roof = {"bound": "hbm", "achieved": nbytes / t_avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": nflops / t_avg / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["algorithmic_bytes"] = nbytes
    roof["traffic"] = None
    try:  # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py), if this shape was profiled
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            key = "%s|%d|%d|%d|%d" % (a.dominant, shape["n_windows"], shape["L"], shape["c_in"], shape["c_out"])
            roof["traffic"] = json.load(f)["kernels"][key]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    roof["kernel"] = a.dominant
    roof["launch_ms"] = t_avg * 1e3
    roof["launch_shape"] = shape
    # every GEMM launch of the step from the serial pass: ms, TFLOP/s (against the dense bf16 peak) and algorithmic GB/s
    roof["families_serial"] = {nm: {"ms_per_step": round(f["ms_per_step"], 4),
                                    "frac_of_mfma_peak": round(sum(l["flops"] for l in f["launches"]) / (f["ms_per_step"] * 1e-3) / 1e12
                                                               / MFMA_BF16_PEAK_TF, 4),
                                    "launches": [{"L": l["shape"]["L"], "c_in": l["shape"]["c_in"], "c_out": l["shape"]["c_out"], "fused": l["shape"]["fused"],
                                                  "ms": round(l["ms"], 4), "tflops": round(l["tflops"], 1),
                                                  "algorithmic_gbs": round(l["gbs"], 1)} for l in f["launches"]]}
                               for nm, f in fam.items()}
    roof["family_ms_per_step_in_timed_region"] = in_region
    roof["step_hbm_frac"] = TRAIN_BYTES_PER_WINDOW * (2 * pairs * a.steps / dt) / (HBM_PEAK_GBS * 1e9)
    roof["step_mfma_frac"] = TRAIN_FLOPS_PER_WINDOW * (2 * pairs * a.steps / dt) / (MFMA_BF16_PEAK_TF * 1e12)

    out = {"metric": "audio-sec/s embedded, 3s@16kHz siamese batch (training step)", "value": value, "unit": "audio-s/s",
           "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "siamese cfg-A train step (filters 128, embed 64, %s loss), %d pairs/GPU of 3 s @ 16 kHz "
                                  "decimated x4 (L=12000), Adam(clipnorm 1)" % (a.loss, pairs),
                      "global_pairs": pairs * n_gpus, "parallelism": "dp%d" % n_gpus, "final_loss": loss},
           "roofline": roof}

    if a.breakdown and rank == 0 and n_gpus == 1:  # its steps contain collectives: single-process runs only
        names = ["vm_decimate_whiten", "vm_conv1_fused_fwd", "vm_conv1_fused_bwd", "vm_conv1_fwd", "vm_conv_fwd", "vm_bn_finalize", "vm_bn_drop_pool_fwd", "vm_bn_drop_pool_gmax_fwd",
                 "vm_global_maxpool_fwd", "vm_dense_fwd", "vm_siamese_head_loss", "vm_dense_bwd", "vm_global_maxpool_bwd",
                 "vm_bn_pool_bwd_reduce", "vm_bn_pool_bwd_reduce_pooled", "vm_bn_pool_bwd_reduce_gmax", "vm_bn_bwd_from_sums", "vm_bn_bwd_finalize", "vm_bn_pool_bwd_apply",
                 "vm_bn_pool_bwd_apply_gmax", "vm_colsum", "vm_conv_wgrad",
                 "vm_conv_dgrad", "vm_conv1_wgrad", "vm_grad_sqnorm", "vm_adam_clip_step", "vm_prep_conv_weights"]
        eng.timed = {nm: [] for nm in names}
        was_overlap, eng.overlap_wgrad = eng.overlap_wgrad, False  # serial, so that every entry point is attributable
        was_split2, eng.split_towers = eng.split_towers, False
        reps = 3
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        rows = []
        for nm in names:
            r_ = eng.timed[nm]
            tot = sum(e0.elapsed_time(e1) for e0, e1, _ in r_) / reps
            rows.append((nm, len(r_) // reps, tot))
        eng.timed = {}
        eng.overlap_wgrad = was_overlap
        eng.split_towers = was_split2
        with open(a.breakdown, "w") as f:
            f.write("entry_point,launches_per_step,ms_per_step\n")
            for nm, cnt, tot in sorted(rows, key=lambda r: -r[2]):
                f.write("%s,%d,%.4f\n" % (nm, cnt, tot))
            f.write("TOTAL_EVENT_MS,,%.4f\nWALL_MS_PER_STEP,,%.4f\n" % (sum(r[2] for r in rows), ms))
            f.write("# breakdown steps run with the wgrad side stream off; WALL is the timed region (overlap %s)\n" % ("on" if was_overlap else "off"))

    if rank == 0 and n_gpus == 1 and not a.no_extras:
        # Side figures of SURVEY 8(d), measured after (and outside) the timed region: the same step with the other loss of
        # the two training scripts, and the embed-only (inference) pass of the same 256 windows.
        def timed(fn, reps=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps
        other = "bce" if a.loss == "contrastive" else "contrastive"
        snap = (eng.P.clone(), eng.M.clone(), eng.V.clone(), eng.iterations)

        def step_other():
            eng.preprocess(pl, xcat, 4, True, pairs)
            eng.forward(pl, pairs, None)
            eng.siamese_head(pl, y, other)
            eng.backward(pl)
            eng.optimizer_step()
        t_other = timed(step_other)
        pli = eng.plan(2 * pairs, l0, False)

        def embed_only():
            eng.preprocess(pli, xcat, 4, True, 2 * pairs)
            eng.forward(pli, 2 * pairs, None)
        t_embed = timed(embed_only)
        # PCIe-inclusive step: the boundary handed host buffers (pinned int16 PCM of the same windows) -- never `value`
        host16 = (xcat.clamp(-1, 1) * 32767.0).round().to(torch.int16).cpu().pin_memory()
        dev16 = torch.empty_like(host16, device=dev)

        def step_h2d():
            dev16.copy_(host16, non_blocking=True)
            eng.preprocess(pl, dev16, 4, True, pairs)
            eng.forward(pl, pairs, None)
            eng.siamese_head(pl, y, a.loss)
            eng.backward(pl)
            eng.optimizer_step()
        t_h2d = timed(step_h2d)
        # device-resident corpus (voicemap_amd/shards.py): the 256 windows are start offsets into an int16 buffer in HBM and
        # the crop happens inside the preprocessing kernel -- the data path of experiments/train_siamese.py --device-data
        corpus = (torch.randn(64 * 1024 * 1024, device=dev) * 0.05 * 32767.0).clamp(-32767, 32767).to(torch.int16)
        offs = torch.randint(0, corpus.numel() - 48000, (2 * pairs,), device=dev, dtype=torch.int64)

        def step_offsets():
            eng.preprocess(pl, corpus, 4, True, pairs, offsets=offs, raw_len=48000)
            eng.forward(pl, pairs, None)
            eng.siamese_head(pl, y, a.loss)
            eng.backward(pl)
            eng.optimizer_step()
        t_off = timed(step_offsets)
        del corpus
        eng.P.copy_(snap[0]); eng.M.copy_(snap[1]); eng.V.copy_(snap[2]); eng.iterations = snap[3]
        eng.refresh_weights()
        # BASELINE.json config 4: the log-mel + 2-D CNN variant (not in the reference): one siamese training step
        # of 128 pairs of RAW 3 s clips -- vm_stft_logmel + four Conv2D 3x3 blocks (filters 32) + loss + backward + Adam
        spectro_extras = {}
        try:
            from voicemap_amd.spectro_engine import HipSpectrogramEncoderEngine
            seng = HipSpectrogramEncoderEngine(32, 64, dropout=0.0, head="uniform_euclidean", dtype=a.dtype, device=dev, seed=1234)
            spl = seng.plan(2 * pairs, 48000, True)

            def spectro_step():
                seng.features(spl, xcat)
                seng.forward(spl, pairs, None)
                seng.siamese_head(spl, y, a.loss)
                seng.backward(spl)
                seng.optimizer_step()

            def spectro_features():
                seng.features(spl, xcat)
            t_sp = timed(spectro_step)
            t_ft = timed(spectro_features)
            spectro_extras = {"logmel_2dcnn_train_ms_per_step": t_sp * 1e3, "logmel_2dcnn_audio_s_per_s": 2 * pairs * 3.0 / t_sp,
                              "logmel_frontend_ms_per_256_clips": t_ft * 1e3,
                              "logmel_2dcnn_config": "log-mel 298 x 64 (25 ms / 10 ms frames), Conv2D 3x3 channels 32-64-96-128, embedding 64, "
                                                     "%d pairs, %s storage" % (pairs, a.dtype)}
            del seng, spl
        except Exception as e:  # the side figure must never take the headline line down
            spectro_extras = {"logmel_2dcnn_error": repr(e)}
        out["extras"] = {"%s_loss_ms_per_step" % other: t_other * 1e3,
                         "embed_only_audio_s_per_s": 2 * pairs * 3.0 / t_embed, "embed_only_ms_per_256_windows": t_embed * 1e3,
                         "pcie_inclusive_ms_per_step_int16_host_windows": t_h2d * 1e3,
                         "device_resident_corpus_ms_per_step_int16_offsets": t_off * 1e3}
        out["extras"].update(spectro_extras)
        # BASELINE config 5: the n-shot k-way evaluation loop of experiments/k_way_accuracy.py (5-way 1-shot, siamese distances) over a
        # device-resident synthetic corpus -- tasks are start offsets, embedded in batches by the cfg-A encoder of this run's shape
        try:
            import tempfile
            from voicemap_amd import models as VM, shards as VS, utils as VU
            from voicemap_amd.librispeech import SyntheticSpeechDataset
            with tempfile.TemporaryDirectory() as td:
                VS.write_shards(SyntheticSpeechDataset(num_speakers=40, files_per_speaker=4, seconds=3, seed=3), td)
                sd = VS.ShardedSpeechDataset(td, 3, stochastic=True)
                enc = VM.get_baseline_convolutional_encoder(F, E, dropout=0.0, dtype=a.dtype)
                net = VM.build_siamese_net(enc, (sd.fragment_length // 4, 1))
                net.compile(loss="binary_crossentropy", optimizer="adam")
                bp = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
                sd.to_device("cuda")
                np.random.seed(5)
                VU.n_shot_task_evaluation(net, sd, bp, 50, 1, 5, network_type="siamese", distance="euclidean")
                torch.cuda.synchronize()
                tasks = 400
                t1 = time.perf_counter()
                VU.n_shot_task_evaluation(net, sd, bp, tasks, 1, 5, network_type="siamese", distance="euclidean")
                torch.cuda.synchronize()
                t_k = time.perf_counter() - t1
                out["extras"].update({"kway_eval_5way_1shot_tasks_per_s": tasks / t_k,
                                      "kway_eval_audio_s_per_s": tasks * 6 * 3.0 / t_k,
                                      "kway_eval_config": "%d tasks of 6 x 3 s windows from a device-resident int16 corpus, host draws the tasks" % tasks})
                del net, enc, sd
        except Exception as e:
            out["extras"]["kway_eval_error"] = repr(e)
        if a.dtype in ("bf16", "f16"):
            # the exact-parity storage mode (fp32 activations, split-precision MFMAs) on the same windows: its step time and how far the
            # bf16 embeddings / gradients of THIS run are from it (north star: embeddings within 1e-3 of the reference arithmetic)
            try:
                e32 = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="f32", device=dev, seed=1234)
                e32.set_params({k: v for k, v in eng.get_params().items()})
                p32 = e32.plan(2 * pairs, l0, True)

                def step32(update=True):
                    e32.preprocess(p32, xcat, 4, True, pairs)
                    e32.forward(p32, pairs, None)
                    e32.siamese_head(p32, y, a.loss)
                    e32.backward(p32)
                    if update:
                        e32.optimizer_step()
                step32(False)
                eng.preprocess(pl, xcat, 4, True, pairs)
                eng.forward(pl, pairs, None)
                eng.siamese_head(pl, y, a.loss)
                eng.backward(pl)
                torch.cuda.synchronize()
                rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-30))
                cos = lambda u, v: float((u * v).sum() / (u.norm() * v.norm()).clamp_min(1e-30))
                gs = eng.G / float(eng.loss_scale)   # f16 storage keeps loss_scale x the gradients in G
                out["extras"].update({"%s_vs_f32_embedding_rel_err" % a.dtype: rel(pl["emb"], p32["emb"]),
                                      "%s_vs_f32_gradient_rel_err" % a.dtype: rel(gs, e32.G), "%s_vs_f32_gradient_cosine" % a.dtype: cos(gs, e32.G)})
                emb32, g32 = p32["emb"].clone(), e32.G.clone()  # before the timed steps move the parameters
                t32 = timed(step32, reps=5)
                out["extras"]["f32_storage_ms_per_step"] = t32 * 1e3
                del e32, p32
                # fp32 storage with split-bf16 products in the k=3 GEMMs (dtype "f32s"): step time and distance from the fp32 mode
                es = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="f32s", device=dev, seed=1234)
                es.set_params({k: v for k, v in eng.get_params().items()})
                ps = es.plan(2 * pairs, l0, True)

                def step32s(update=True):
                    es.preprocess(ps, xcat, 4, True, pairs)
                    es.forward(ps, pairs, None)
                    es.siamese_head(ps, y, a.loss)
                    es.backward(ps)
                    if update:
                        es.optimizer_step()
                step32s(False)
                torch.cuda.synchronize()
                out["extras"].update({"f32s_vs_f32_embedding_rel_err": rel(ps["emb"], emb32), "f32s_vs_f32_gradient_rel_err": rel(es.G, g32)})
                out["extras"]["f32s_storage_ms_per_step"] = timed(step32s, reps=5) * 1e3
                del es, ps
            except Exception as e:
                out["extras"]["f32_storage_error"] = repr(e)

    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        from oracle import voicemap_oracle as O
        arch = O.EncoderArch.baseline(F, E, dropout=0.0)
        cpu_pairs, cpu_steps = 8, 4
        sec, threads = O.time_cpu_train_steps(arch, cpu_pairs, cpu_steps, loss=a.loss, threads=None)
        trials = getattr(O.time_cpu_train_steps, "last_trials", {})
        out["cpu_baseline"] = {"value": 2 * cpu_pairs * 3.0 / sec, "unit": "audio-s/s", "cores": threads, "kind": "port",
                               "host_cpu_count": os.cpu_count() or 1,
                               "threads_tried_ms_per_step": {str(k): round(v * 1e3, 1) for k, v in sorted(trials.items())},
                               "sample": "<=%d steps of %d pairs, cfg-A, same step definition (fp32 torch-CPU oracle, best of the "
                                         "intra-op thread counts listed in threads_tried on a %d-cpu host: a batch of %d pairs does "
                                         "not scale to every core; %.0f ms/step)"
                                         % (cpu_steps, cpu_pairs, os.cpu_count() or 1, cpu_pairs, sec * 1e3)}
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
