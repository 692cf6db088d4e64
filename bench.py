#!/usr/bin/env python
"""Headline benchmark of the voicemap hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one ``train_on_batch`` of the siamese script at cfg-A (experiments/train_siamese.py:20-25: filters 128,
embedding 64, dropout 0; contrastive loss per BASELINE.json config 2), 128 pairs (256 windows) of 3 s @ 16 kHz per
GPU, 16-bit storage / fp32 accumulate: decimate x4 + whiten on the GPU, twin forward, loss, backward, (gradient
all-reduce), global-norm clip + Adam, GEMM-layout weight refresh.  Raw windows are synthetic and resident
in HBM before the timed region.  value = audio-seconds embedded per second = N * 256 windows * 3 s * K / wall time.

Storage mode of the headline (--dtype, default "f16"): IEEE half tensors on the v_mfma_f32_32x32x16_f16 pipe with fp32
accumulation -- the same bytes and the same matrix-pipe rate as the bf16 that BASELINE.json names for this config, with 11
instead of 8 significand bits per stored value: it is the 16-bit mode that meets the north star's "embeddings within 1e-3 of the
reference arithmetic" (7.4e-4 against the float64 oracle at this very size, tests/test_gpu_fullsize_oracle.py; bf16: 6e-3).  The
bf16 step is timed the same way and reported under extras.bf16_mode; the line's "precision" object states mode and tolerance.

Timing: ``--blocks`` (default 5) blocks of exactly K steps, each bracketed by barrier + torch.cuda.synchronize() on both sides, the
maximum over ranks per block; the line reports the MEDIAN block (steps = K, ms_per_step = median block / K) and lists all blocks:
one 60 ms block alone decides nothing below the +-3 % the boxes differ by.

Extra objects on the JSON line: "roofline" (dominant GEMM family, HIP events on the launch stream), "precision", "timing",
"data_parallel" (N > 1: backend, per-rank ms, un-hidden all-reduce time) and "cpu_baseline" (the CPU oracle's fp32 training step on
a bounded sample, rank 0 at N = 1 only).
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
import torch.distributed

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_16BIT_PEAK_TF = 2500.0  # dense bf16 / f16 MFMA peak
TRAIN_BYTES_PER_WINDOW = 44.02e6   # algorithmic HBM bytes per 3 s window, training, S4k, 16-bit storage
TRAIN_FLOPS_PER_WINDOW = 7.373e9
F, E, L0 = 128, 64, 12000
BLOCKS = [(32, F, 4), (3, 2 * F, 2), (3, 3 * F, 2), (3, 4 * F, 2)]
# rocprofv3 kernel names of the GEMM families under the default dispatch (what profiles/r03_rocprofv3_kernel_stats.csv lists)
KERNEL_SYMBOL = {"vm_conv_fwd": "vm::conv_nt2r_kernel<{T}, 0>", "vm_conv_dgrad": "vm::conv_nt2r_kernel<{T}, 1>",
                 "vm_conv_wgrad": "vm::conv_tn9_kernel<{T}>"}
# round 4: with packed weights (the default for 16-bit storage) forward / dgrad run conv_nt3_kernel<T, EPI, K-side channels / 32, true>
NT3_EPI = {"vm_conv_fwd": 3, "vm_conv_dgrad": 1}
CTYPE = {"bf16": "__bf16", "f16": "_Float16", "f32": "float", "f32s": "float"}


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
    if fused == "vm_conv_fwd_fold" and args[base + 8] is not None:   # (in_e, wf, bias, n, L, c_in, c_out, dtype, z, sum, sq, e, ...)
        pairs = args[base + 5] is None   # no z: the output is the (extreme, other element) pair, n * L * c_out values as z would be
        extra = 0 if pairs else n * (L // 2) * cout
        shape["fused"] = "vm_conv_fwd_fold+" + ("pairs" if pairs else "e")
    return (n * L * (cin + cout) + extra) * esize, 2.0 * n * L * 3 * cin * cout, shape


def timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t1) / reps


def snapshot(eng):
    return (eng.P.clone(), eng.M.clone(), eng.V.clone(), eng.NT.clone(), eng.iterations, eng.ZD.clone(), eng.bn_steps)


def restore(eng, s):
    eng.P.copy_(s[0]); eng.M.copy_(s[1]); eng.V.copy_(s[2]); eng.NT.copy_(s[3]); eng.iterations = s[4]
    eng.ZD.copy_(s[5]); eng.bn_steps = s[6]
    eng.refresh_weights()


def fail(msg, code=2):
    """A loud failure: one JSON error line on stdout (so a record of the run says why there is no number) and a non-zero exit."""
    print(json.dumps({"error": msg, "metric": None, "value": None}), flush=True)
    sys.stderr.write("bench.py: " + msg + "\n")
    sys.stderr.flush()
    os._exit(code)


class Watchdog:
    """Turns a hang of the rendezvous or of the first collective into an error line: ``arm(what)`` starts a timer that, unless
    ``disarm()`` comes first, reports ``what`` and ends the process (a hung RCCL call cannot be interrupted from Python)."""

    def __init__(self, seconds):
        self.seconds = seconds
        self._timer = None

    def arm(self, what):
        import threading
        self.disarm()
        self._timer = threading.Timer(self.seconds, fail, args=("watchdog: no progress in %.0f s during %s (rank %s)"
                                                                % (self.seconds, what, os.environ.get("RANK", "0")), 3))
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


def self_launch(n):
    """``python bench.py --gpus N`` without a torchrun environment: re-exec this very command line under
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` (the driver's own launch form), one rank per GPU over RCCL.
    Fewer than N visible devices is an error unless the transport is the gloo rehearsal (ranks then share devices)."""
    import socket
    import subprocess
    if os.environ.get("VOICEMAP_DIST_BACKEND") != "gloo" and not os.environ.get("VOICEMAP_DIST_SHARE_DEVICE"):
        have = torch.cuda.device_count()
        if have < n:
            fail("bench.py --gpus %d: only %d device(s) visible on this node -- refusing to report a smaller run as n_gpus %d"
                 % (n, have, n))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def live_pmc_traffic(key, a):
    """HBM bytes per launch of the dominant kernel, counted NOW: two ``rocprofv3 --pmc`` passes (FETCH_SIZE, then WRITE_SIZE: separate
    passes as MI355X_MICROARCH.md prescribes) over three serial steps of this very script in a child process, folded by
    tools/pmc_traffic.py (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction).  Returns (bytes or None, how / why not)."""
    import contextlib
    import io
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):
        return None, "this process runs under a profiler itself"
    tmp = tempfile.mkdtemp(prefix="vm_pmc_", dir="/tmp")
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    tune = ",".join(t for t in (a.tune, "split_towers=0") if t)
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--blocks", "1", "--no-cpu-baseline", "--no-extras",
             "--no-overlap-wgrad", "--no-live-pmc", "--tune", tune, "--dtype", a.dtype, "--pairs", str(a.pairs), "--loss", a.loss]
    t0 = time.perf_counter()
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            r = subprocess.run([exe, "--pmc", c, "--output-format", "csv", "-d", os.path.join(tmp, c), "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=150)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s ended with rc %d: %s" % (c, r.returncode, r.stderr.decode(errors="replace")[-200:])
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_traffic
        with contextlib.redirect_stdout(io.StringIO()):
            pmc_traffic.main(os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"), os.path.join(tmp, "traffic.json"))
        with open(os.path.join(tmp, "traffic.json")) as f:
            got = json.load(f)["kernels"].get(key)
        if got is None:
            return None, "the PMC passes hold no launch of shape %s" % key
        return got["hbm_bytes"], ("counted in this run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a child process running three serial "
                                  "steps of this script on this device, %.0f s; 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction "
                                  "(tools/pmc_traffic.py)" % (time.perf_counter() - t0))
    except Exception as e:   # noqa: BLE001  (a timeout, a parse error: the committed figure serves, labelled)
        return None, "live PMC passes failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def vendor_gemm_ms(kernel, shape, dtype, dev):
    """What the vendor GEMM library (hipBLASLt / rocBLAS behind torch.matmul) needs for the PLAIN GEMM of the dominant launch -- same
    M, N, K and storage type, no im2col addressing, no bias / ReLU / statistics / pool-pair epilogue -- on this device, now."""
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    n, L, cin, cout = shape["n_windows"], shape["L"], shape["c_in"], shape["c_out"]
    if kernel == "vm_conv_wgrad":
        a_ = torch.randn(n * L, 3 * cin, device=dev, dtype=tdt)
        b_ = torch.randn(n * L, cout, device=dev, dtype=tdt)
        f = lambda: torch.matmul(a_.t(), b_)   # noqa: E731
        mnk = (3 * cin, cout, n * L)
    else:
        nn, kk = (cout, 3 * cin) if kernel == "vm_conv_fwd" else (cin, 3 * cout)
        a_ = torch.randn(n * L, kk, device=dev, dtype=tdt)
        b_ = torch.randn(nn, kk, device=dev, dtype=tdt)
        f = lambda: torch.matmul(a_, b_.t())   # noqa: E731
        mnk = (n * L, nn, kk)
    for _ in range(3):
        f()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), mnk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--pairs", type=int, default=128, help="pairs per GPU (cfg: 128)")
    ap.add_argument("--dtype", default="f16", help="storage mode of the headline: f16 (default) | bf16 | f32s | f32")
    ap.add_argument("--loss", default="contrastive")
    ap.add_argument("--dominant", default="auto", help="entry point whose launches the roofline object describes: auto = the "
                    "GEMM family (vm_conv_fwd / vm_conv_dgrad / vm_conv_wgrad) with the largest share of the serial pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side figures (other modes, embed-only pass, evaluation)")
    ap.add_argument("--no-overlap-wgrad", action="store_true",
                    help="keep the weight-gradient GEMMs on the main stream (default: side stream, concurrent with dgrad)")
    ap.add_argument("--tune", default="", help="extra tuning knobs key=value,key=value (engine switches / vm_set_tuning)")
    ap.add_argument("--breakdown", default="", help="write a per-entry-point time breakdown (extra untimed steps) to this file")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed profiles/pmc_traffic.json instead of two "
                    "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) run by this process on the serial step")
    ap.add_argument("--allow-nonfinite", action="store_true", help="timing experiments with ablated builds (tools/build_variant.sh): results are wrong by design")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # plain ``python bench.py --gpus N``: become the launcher of N ranks (one per GPU) -- never a silent 1-GPU run
        sys.exit(self_launch(a.gpus))

    from voicemap_amd import parallel
    from voicemap_amd.engine import HipEncoderEngine
    # (the rendezvous gets twice the collectives' limit: ranks of a fresh box finish ``import torch`` at different times)
    watchdog = Watchdog(2 * float(os.environ.get("VOICEMAP_DIST_WATCHDOG_S", "120")))
    watchdog.arm("torch.distributed rendezvous / init_process_group (world %s)" % os.environ.get("WORLD_SIZE", "1"))
    rank, world, local = parallel.init_distributed(timeout_s=watchdog.seconds)
    watchdog.disarm()
    watchdog.seconds = float(os.environ.get("VOICEMAP_DIST_WATCHDOG_S", "120"))
    if world != a.gpus:
        fail("bench.py --gpus %d was started with WORLD_SIZE=%d: launch with --nproc-per-node == --gpus (or without torchrun: "
             "bench.py spawns the ranks itself)" % (a.gpus, world))
    n_gpus = world
    share = bool(os.environ.get("VOICEMAP_DIST_SHARE_DEVICE"))   # rehearsal of the RCCL path itself on a box with fewer GPUs than ranks:
    # RCCL refuses two ranks of a communicator on one device -- tests/test_gpu_bench.py asserts that the refusal comes back as an
    # error line within the watchdog's limit, not as a hang
    if n_gpus > 1 and os.environ.get("VOICEMAP_DIST_BACKEND") != "gloo" and not share and torch.cuda.device_count() < n_gpus:
        fail("bench.py --gpus %d: only %d device(s) visible on this node" % (n_gpus, torch.cuda.device_count()))
    if os.environ.get("VOICEMAP_DIST_BACKEND") == "gloo" or share:   # rehearsal: more ranks than GPUs, the replicas share devices
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def make_engine(dtype):
        e = HipEncoderEngine(BLOCKS, E, dropout=0.0, head="uniform_euclidean", dtype=dtype, device=dev, seed=1234)
        e.overlap_wgrad = not a.no_overlap_wgrad
        return e
    eng = make_engine(a.dtype)
    ENGINE_SWITCHES = {"overlap_wgrad": bool, "pooled_reduce": bool, "split_towers": bool, "fused_bn_reduce": bool, "wgrad_after_dgrad": bool,
                       "fused_pool_extreme": bool, "tower_stagger": int, "loss_scale": float, "fused_sums_finalize": bool, "fold_affine": bool, "fold_pairs": bool, "side_priority": int}
    for kv in [t for t in a.tune.split(",") if t]:
        k, v = kv.split("=")
        if k == "main_priority":   # experiment: the whole step on a stream of this HIP priority (-1 = above the side stream's 0)
            torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(v)))
        elif k in ENGINE_SWITCHES:
            setattr(eng, k, ENGINE_SWITCHES[k](float(v)))
        else:
            eng.lib.call("vm_set_tuning", k.encode(), int(v))
    parallel.attach(eng, n_gpus)
    if n_gpus > 1:
        watchdog.arm("the first collectives (barrier + state broadcast, backend %s)" % torch.distributed.get_backend())
        parallel.barrier()
    parallel.broadcast_state(eng)
    if n_gpus > 1:
        torch.cuda.synchronize()
        watchdog.disarm()

    # synthetic raw windows, a different shard per rank, resident in HBM
    pairs = a.pairs
    rng = np.random.default_rng(1234 + rank)

    def raw():
        x = rng.normal(0.0, 0.05, size=(pairs, 48000)).astype(np.float32)
        return torch.from_numpy(x + rng.uniform(-0.01, 0.01, size=(pairs, 1)).astype(np.float32)).to(dev)
    x1, x2 = raw(), raw()
    y = torch.cat([torch.zeros(pairs // 2), torch.ones(pairs - pairs // 2)]).to(dev)
    xcat = torch.cat([x1, x2], 0).contiguous()

    def make_step(e, loss=a.loss, sync_tail=True):
        p_ = e.plan(2 * pairs, L0, True)

        def step():
            if sync_tail or e.grad_sync is None:
                # the product's step (what siamese_train_step runs behind its host-to-device copies): preprocess -> forward -> head ->
                # backward (N > 1: the large gradient all-reduce starts before block 1's backward) -> Adam; on one GPU its enqueue
                # sequence is recorded once and replayed (engine._Program) -- same launches, same arguments
                # (input_ready: the synthetic windows are resident and complete before the timed region starts -- the bench contract --
                # so the step's preprocessing may run beside the previous step's optimizer tail; engine.pre_overlap)
                e.train_step_resident(p_, pairs, y, loss, raw=xcat, drop_masks=None, input_ready=True)
                return
            e.preprocess(p_, xcat, 4, True, pairs)
            e.forward(p_, pairs, None, defer_tail=True)
            e.siamese_head(p_, y, loss)
            e.backward(p_, sync_tail=False)
            e.optimizer_step()
        return step, p_

    host_enqueue_s = []

    def time_blocks(step, n_blocks, k_steps):
        """n_blocks blocks of exactly k_steps steps, barrier + synchronize on both sides of each, max over ranks per block."""
        out, mine = [], []
        for _ in range(n_blocks):
            torch.cuda.synchronize()
            parallel.barrier()
            t0 = time.perf_counter()
            for _ in range(k_steps):
                step()
            host_enqueue_s.append(time.perf_counter() - t0)   # the host is done enqueueing: how far ahead of the GPU it runs
            torch.cuda.synchronize()
            t_local = time.perf_counter() - t0
            parallel.barrier()
            dt = time.perf_counter() - t0
            mine.append(t_local)
            out.append(parallel.max_over_ranks(dt, dev))
        return out, mine

    step, pl = make_step(eng)
    if n_gpus > 1:
        watchdog.arm("the warm-up steps (first gradient all-reduce)")
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    watchdog.disarm()
    if n_gpus > 1 and eng.grad_sync is not None:
        eng.grad_sync.time_wait = True
    block_s, mine_s = time_blocks(step, max(1, a.blocks), a.steps)
    dt = float(np.median(block_s))
    loss = float(pl["loss_acc"][0].item())
    # a tuning set may switch kernels, never results: a non-finite loss is an error
    assert np.isfinite(loss) or a.allow_nonfinite, "training diverged"
    # f16 storage: a step whose loss-scaled gradient norm is not finite runs every launch and leaves the parameters alone (dynamic loss
    # scaling, engine.adjust_loss_scale).  On synthetic noise a collapsed state can ask for that for a few steps in a row (BatchNorm
    # layers with no variance amplify the gradient 30 x each: profiles/r06_block1_products.txt); the count goes on the line -- a run in
    # which more than a tenth of the steps did not update is not a training run
    n_skipped = int(eng.skipped_steps())
    assert n_skipped * 10 <= a.warmup + a.steps * len(block_s) or a.allow_nonfinite, "loss-scaled steps were skipped (non-finite gradients): %d" % n_skipped

    value = 2 * pairs * n_gpus * a.steps * 3.0 / dt
    ms = dt / a.steps * 1e3
    out = {"metric": "audio-sec/s embedded, 3s@16kHz siamese batch (training step)", "value": value, "unit": "audio-s/s",
           "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "siamese cfg-A train step (filters 128, embed 64, %s loss), %d pairs/GPU of 3 s @ 16 kHz "
                                  "decimated x4 (L=12000), Adam(clipnorm 1), %s storage" % (a.loss, pairs, a.dtype),
                      "global_pairs": pairs * n_gpus, "parallelism": "dp%d" % n_gpus, "final_loss": loss},
           "timing": {"blocks": len(block_s), "steps_per_block": a.steps, "reported": "median block",
                      "block_ms_per_step": [round(b / a.steps * 1e3, 4) for b in block_s],
                      "host_enqueue_ms_per_step": round(float(np.median(host_enqueue_s)) / a.steps * 1e3, 4),
                      "optimizer_steps_not_applied": n_skipped}}
    if n_gpus > 1:
        import torch.distributed as dist
        gs = eng.grad_sync
        gs.time_wait = False
        wait_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in gs.wait_events])) if gs.wait_events else 0.0
        per_rank = torch.tensor([float(np.median(mine_s)) / a.steps * 1e3, wait_ms], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(per_rank) for _ in range(n_gpus)]
        dist.all_gather(allr, per_rank)
        # what makes the first real N > 1 line self-verifying: the transport's version, WHICH devices the ranks ran on, and whether
        # the replicas still hold the same parameters after the timed steps (they must: same reduced gradient, same update)
        props = torch.cuda.get_device_properties(dev)
        ident = {"rank": rank, "host": os.uname().nodename, "device_index": int(dev.index), "name": props.name,
                 "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))}
        idents = [None] * n_gpus
        dist.all_gather_object(idents, ident)
        psum = torch.stack([eng.P.double().sum(), eng.P.double().abs().sum(), eng.M.double().sum()])
        psums = [torch.zeros_like(psum) for _ in range(n_gpus)]
        dist.all_gather(psums, psum)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception as e:   # noqa: BLE001
            rccl = "unavailable: %r" % (e,)
        out["data_parallel"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": rccl,
                                "rank_devices": idents,
                                "distinct_devices": len({(i["host"], i["uuid"] or i["pci_bus_id"], i["device_index"]) for i in idents}),
                                "replicas_bit_identical_after_timed_steps": bool(all(torch.equal(psums[0], q) for q in psums)),
                                "collectives_per_step": gs.collectives / max(1, a.warmup + a.steps * len(block_s)),
                                "flat_gradient_bytes": int(eng.n_flat * 4),
                                "rank_ms_per_step": [round(float(t[0]), 4) for t in allr],
                                "rank_allreduce_unhidden_ms_per_step": [round(float(t[1]), 4) for t in allr],
                                "note": "rank_ms: each rank's own median block (before the closing barrier); unhidden: the optimizer-step "
                                        "stream's time inside / waiting for the two collectives (HIP events)"}

    # ---- serial attribution pass (untimed, after the timed region): the same step with the weight-gradient GEMMs on the main
    # stream, every GEMM launch bracketed by HIP events on its launch stream; median over the repetitions per launch shape ----
    esize = 2 if a.dtype in ("bf16", "f16") else 4
    gemm = ["vm_conv_fwd", "vm_conv_dgrad", "vm_conv_wgrad"]
    was_overlap, eng.overlap_wgrad = eng.overlap_wgrad, False
    was_split, eng.split_towers = eng.split_towers, False   # one launch per GEMM of the step, nothing else in flight
    snap0 = snapshot(eng)
    eng.timed = {nm: [] for nm in gemm}
    attr_steps = max(5, min(a.steps, 20))
    for _ in range(attr_steps):
        step()
    torch.cuda.synchronize()
    fam = {}
    for nm in gemm:
        by_shape = {}
        for e0, e1, args in eng.timed[nm]:
            work = conv_launch_work(nm, args, esize)
            by_shape.setdefault(tuple(work[2].values()), (work, []))[1].append(e0.elapsed_time(e1) * 1e-3)
        launches = []
        for key, ((nb_, nf_, shp), ts) in by_shape.items():
            t_med = float(np.median(ts))
            # (the folded forward launches every layer once per tower: ``per_step`` launches of this shape in a step)
            launches.append({"shape": shp, "ms": t_med * 1e3, "tflops": nf_ / t_med / 1e12, "gbs": nb_ / t_med / 1e9,
                             "algorithmic_bytes": nb_, "flops": nf_, "per_step": len(ts) // attr_steps})
        fam[nm] = {"ms_per_step": sum(l["ms"] * l["per_step"] for l in launches), "launches": launches}
    eng.timed = {}
    eng.overlap_wgrad = was_overlap
    eng.split_towers = was_split
    restore(eng, snap0)
    if a.dominant == "auto":
        a.dominant = max(fam, key=lambda k: fam[k]["ms_per_step"])
    worst = max(fam[a.dominant]["launches"], key=lambda l: l["ms"])
    t_avg = worst["ms"] * 1e-3
    nbytes, nflops, shape = worst["algorithmic_bytes"], worst["flops"], worst["shape"]
    ridge = MFMA_16BIT_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
    if nflops / nbytes < ridge:
        roof = {"bound": "hbm", "achieved": nbytes / t_avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": nflops / t_avg / 1e12, "peak": MFMA_16BIT_PEAK_TF, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["algorithmic_bytes"] = nbytes
    roof["traffic"] = None
    key = "%s|%d|%d|%d|%d" % (a.dominant, shape["n_windows"], shape["L"], shape["c_in"], shape["c_out"])
    committed = None
    try:  # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py), if this shape was profiled
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            committed = json.load(f)["kernels"][key]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    why = "--no-live-pmc" if a.no_live_pmc else ("N > 1 or not rank 0" if n_gpus > 1 or rank else "tools/pmc_traffic.py maps the 256-window launches only")
    if rank == 0 and n_gpus == 1 and not a.no_live_pmc and 2 * pairs == 256:
        roof["traffic"], why = live_pmc_traffic(key, a)
        if roof["traffic"] is not None:
            roof["traffic_source"] = why
            roof["traffic_committed"] = committed   # profiles/pmc_traffic.json (another box, another day): the two must agree
    if roof["traffic"] is None and committed is not None:
        roof["traffic"] = committed
        roof["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch shape on another "
                                  "box (tools/pmc_traffic.py; 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction) -- NOT measured in this run "
                                  "(%s)" % why)
    roof["kernel"] = a.dominant
    def symbol(nm, shp=None):   # the folded forward (vm_conv_fwd_fold) is epilogue variant 3 of the same kernel
        sym = KERNEL_SYMBOL[nm].format(T=CTYPE.get(a.dtype, a.dtype))
        folded = nm == "vm_conv_fwd" and any(str(l["shape"]["fused"]).startswith("vm_conv_fwd_fold") for l in fam[nm]["launches"])
        if folded:
            sym = sym.replace(", 0>", ", 3>")
        fused = nm == "vm_conv_dgrad" and any(l["shape"]["fused"] == "vm_conv_dgrad_bnred" for l in fam[nm]["launches"])
        nt3 = nm in NT3_EPI and getattr(eng, "packed_weights", False) and (folded or fused) and "nt3=0" not in a.tune
        if nt3:   # K-side channel chunks: c_in for the forward, c_out for dgrad (a family: one instantiation per layer)
            ck = None if shp is None else (shp["c_in"] if nm == "vm_conv_fwd" else shp["c_out"]) // 32
            sym = "vm::conv_nt3_kernel<%s, %d, %s, true>" % (CTYPE.get(a.dtype, a.dtype), NT3_EPI[nm], "*" if ck is None else str(ck))
        return sym
    roof["kernel_symbol"] = symbol(a.dominant, shape)
    roof["source"] = ("bench.py serial attribution pass of this run: HIP events on the launch stream around every GEMM launch, median per "
                      "launch shape; committed counterparts: profiles/r06_conv_kernels_by_layer.txt (rocprofv3 --kernel-trace of the "
                      "serial step), profiles/r06_rocprofv3_kernel_stats.csv (default, overlapped step), profiles/pmc_traffic.json (traffic)")
    roof["launch_ms"] = t_avg * 1e3
    roof["launch_shape"] = shape
    # every GEMM launch of the step from the serial pass: ms, TFLOP/s (against the dense 16-bit peak) and algorithmic GB/s
    roof["families_serial"] = {nm: {"ms_per_step": round(f["ms_per_step"], 4), "kernel_symbol": symbol(nm),
                                    "frac_of_mfma_peak": round(sum(l["flops"] * l["per_step"] for l in f["launches"]) / (f["ms_per_step"] * 1e-3) / 1e12
                                                               / MFMA_16BIT_PEAK_TF, 4),
                                    "launches": [{"L": l["shape"]["L"], "c_in": l["shape"]["c_in"], "c_out": l["shape"]["c_out"], "fused": l["shape"]["fused"],
                                                  "n_windows": l["shape"]["n_windows"], "per_step": l["per_step"],
                                                  "ms": round(l["ms"], 4), "tflops": round(l["tflops"], 1),
                                                  "algorithmic_gbs": round(l["gbs"], 1)} for l in f["launches"]]}
                               for nm, f in fam.items()}
    # the dense 16-bit MFMA rate THIS device sustains from registers, measured here beside the step (vm_mfma_rate_probe: every SIMD
    # issuing v_mfma_f32_32x32x16 back to back on non-zero operands, ~3 ms per launch, the chip warm from the timed steps): the step
    # runs on the package power limit (profiles/r06_kernel_power.txt), so `peak` -- the nominal-clock figure the contract asks
    # for -- is not a rate the part holds; frac_of_sustained prices the dominant launch against what it does hold
    if rank == 0 and a.dtype in ("f16", "bf16"):
        try:
            vm_dt = {"bf16": 1, "f16": 3}[a.dtype]
            iters = 12000
            sink = torch.empty(512 * 256, dtype=torch.float32, device=dev)
            st_ = torch.cuda.current_stream(dev).cuda_stream
            probe_flops = float(eng.lib.query("vm_mfma_rate_probe_flops", iters))
            for _ in range(3):
                eng.lib.call("vm_mfma_rate_probe", vm_dt, iters, sink.data_ptr(), st_)
            torch.cuda.synchronize()
            rates = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.lib.call("vm_mfma_rate_probe", vm_dt, iters, sink.data_ptr(), st_)
                e1.record()
                torch.cuda.synchronize()
                rates.append(probe_flops / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            sustained = float(np.median(rates))
            roof["mfma_sustained"] = {"tflops": sustained, "frac_of_nominal_peak": sustained / MFMA_16BIT_PEAK_TF,
                                      "dominant_launch_frac_of_sustained": (nflops / t_avg / 1e12) / sustained,
                                      "source": "vm_mfma_rate_probe in this run: 512 workgroups x 4 waves x %d x 8 dense v_mfma_f32_32x32x16_%s "
                                                "from registers, median of 5 launches (HIP events)" % (iters, a.dtype)}
        except Exception as e:   # noqa: BLE001
            roof["mfma_sustained"] = {"error": repr(e)}
    if rank == 0 and a.dtype in ("f16", "bf16"):
        try:   # the vendor library on the plain GEMM of the same shape, beside the fused launch it is compared with
            v_ms, mnk = vendor_gemm_ms(a.dominant, shape, a.dtype, dev)
            roof["vendor_gemm"] = {"ms": v_ms, "frac": nflops / (v_ms * 1e-3) / 1e12 / MFMA_16BIT_PEAK_TF, "m_n_k": list(mnk),
                                   "this_launch_ms": t_avg * 1e3,
                                   "source": "torch.matmul (%s -> %s, fp32 accumulation) on the plain GEMM of the dominant launch's shape in this run, "
                                             "median of 10 (HIP events): no im2col addressing, no bias / ReLU / statistics / pool-pair epilogue"
                                             % (a.dtype, a.dtype)}
        except Exception as e:   # noqa: BLE001
            roof["vendor_gemm"] = {"error": repr(e)}
    roof["step_hbm_frac"] = TRAIN_BYTES_PER_WINDOW * (2 * pairs * a.steps / dt) / (HBM_PEAK_GBS * 1e9)
    roof["step_mfma_frac"] = TRAIN_FLOPS_PER_WINDOW * (2 * pairs * a.steps / dt) / (MFMA_16BIT_PEAK_TF * 1e12)
    out["roofline"] = roof

    if a.breakdown and rank == 0 and n_gpus == 1:  # its steps contain collectives: single-process runs only
        names = ["vm_decimate_whiten", "vm_conv1_fused_fwd", "vm_conv1_fused_bwd", "vm_conv1_fwd", "vm_conv_fwd", "vm_bn_finalize", "vm_bn_drop_pool_fwd", "vm_bn_drop_pool_gmax_fwd",
                 "vm_global_maxpool_fwd", "vm_dense_fwd", "vm_siamese_head_loss", "vm_dense_bwd", "vm_global_maxpool_bwd",
                 "vm_bn_pool_bwd_reduce", "vm_bn_pool_bwd_reduce_pooled", "vm_bn_pool_bwd_reduce_gmax", "vm_bn_bwd_gmax_finalize", "vm_bn_bwd_from_sums", "vm_bn_bwd_from_sums_finalize", "vm_bn_bwd_finalize", "vm_bn_pool_bwd_apply",
                 "vm_bn_pool_bwd_apply_gmax", "vm_colsum", "vm_du_tower_sums", "vm_fold_bn_weights", "vm_conv_wgrad",
                 "vm_conv_dgrad", "vm_conv1_wgrad", "vm_grad_sqnorm", "vm_adam_clip_step", "vm_prep_conv_weights", "vm_prep_conv_weights_batch",
                 "vm_bn_drop_pool_gmax_partials", "vm_tail_fwd_bwd", "vm_tail_param_grads", "vm_pack_nt_weights_batch", "vm_conv_wgrad_fold_finish"]
        eng.timed = {nm: [] for nm in names}
        was_overlap, eng.overlap_wgrad = eng.overlap_wgrad, False  # serial, so that every entry point is attributable
        was_split2, eng.split_towers = eng.split_towers, False
        snap1 = snapshot(eng)
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
        restore(eng, snap1)
        with open(a.breakdown, "w") as f:
            f.write("entry_point,launches_per_step,ms_per_step\n")
            for nm, cnt, tot in sorted(rows, key=lambda r: -r[2]):
                f.write("%s,%d,%.4f\n" % (nm, cnt, tot))
            f.write("TOTAL_EVENT_MS,,%.4f\nWALL_MS_PER_STEP,,%.4f\n" % (sum(r[2] for r in rows), ms))
            f.write("# breakdown steps run with the wgrad side stream off; WALL is the timed region (overlap %s); dtype %s\n" % ("on" if was_overlap else "off", a.dtype))

    if rank == 0 and n_gpus == 1 and not a.no_extras:
        out["extras"] = extras(a, eng, step, pl, xcat, y, pairs, dev, make_engine, make_step, time_blocks, out)

    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        from oracle import voicemap_oracle as O
        arch = O.EncoderArch.baseline(F, E, dropout=0.0)
        cpu_pairs, cpu_steps = 8, 4
        sec, threads = O.time_cpu_train_steps(arch, cpu_pairs, cpu_steps, loss=a.loss, threads=None)
        trials = getattr(O.time_cpu_train_steps, "last_trials", {})
        out["cpu_baseline"] = {"value": 2 * cpu_pairs * 3.0 / sec, "unit": "audio-s/s", "cores": threads, "kind": "port",
                               "host_cpu_count": os.cpu_count() or 1,
                               "threads_tried_ms_per_step": {str(k): round(v * 1e3, 1) for k, v in sorted(trials.items())},
                               "full_batch_note": "the 128-pair batch of the GPU step is the CPU path's WORSE operating point: 10.0 / 11.3 / 17.8 s "
                                                  "per step on 32 / 64 / 128 threads of the same 256-cpu host = 77 audio-s/s at best "
                                                  "(tools/probe/cpu_full_batch.py, round 6) against 150-200 at 8 pairs -- the sample below is "
                                                  "the batch that favours the CPU",
                               "sample": "<=%d steps of %d pairs, cfg-A, same step definition (fp32 torch-CPU oracle, best of the "
                                         "intra-op thread counts listed in threads_tried on a %d-cpu host: a batch of %d pairs does "
                                         "not scale to every core; %.0f ms/step)"
                                         % (cpu_steps, cpu_pairs, os.cpu_count() or 1, cpu_pairs, sec * 1e3)}
        # the other two legs BASELINE.md 3 names, on the same host with the thread count the search above chose (bounded: ~10 s):
        # the inference embedding pass (voicemap/utils.py:141-156) and BASELINE.json config 1, the classifier step at batch 8
        # (experiments/train_classifier.py:110-127) -- each beside the GPU figure of the same work from this run's extras
        try:
            legs = {}
            n_emb = 16
            t_e = O.time_cpu_embed_only(arch, n_emb, 3, threads)
            legs["embed_only"] = {"value": n_emb * 3.0 / t_e, "unit": "audio-s/s", "cores": threads, "kind": "port",
                                  "sample": "%d windows of 3 s per pass, inference BatchNorm, fp32 (%.0f ms per pass)" % (n_emb, t_e * 1e3),
                                  "gpu_value_this_run": (out.get("extras") or {}).get("embed_only_audio_s_per_s")}
            n_cls, classes = 8, 40
            t_c = O.time_cpu_classifier_steps(arch, n_cls, classes, 3, threads)
            legs["classifier_step_batch8"] = {"value": n_cls * 3.0 / t_c, "unit": "audio-s/s", "cores": threads, "kind": "port",
                                              "sample": "BASELINE.json config 1: train_on_batch of the speaker classifier (cfg-A encoder + "
                                                        "Dense(%d, softmax), categorical CE, Adam(clipnorm 1)), batch %d of 3 s windows, "
                                                        "fp32 (%.0f ms per step)" % (classes, n_cls, t_c * 1e3),
                                              "gpu_value_this_run": (out.get("extras") or {}).get("classifier_batch8_audio_s_per_s")}
            out["cpu_baseline"]["legs"] = legs
        except Exception as e:   # a side figure never takes the line down
            out["cpu_baseline"]["legs_error"] = repr(e)
    if rank == 0:
        print(json.dumps(out))


def extras(a, eng, step, pl, xcat, y, pairs, dev, make_engine, make_step, time_blocks, out):
    """Side figures of SURVEY 8(d), measured after (and outside) the timed region.  Each group is fenced: a failure is reported
    under its own *_error key and never takes the headline line down."""
    ex = {}
    rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-30))
    cos = lambda u, v: float((u * v).sum() / (u.norm() * v.norm()).clamp_min(1e-30))
    snap = snapshot(eng)

    # ---- the other storage / arithmetic modes on the same windows and weights: step time and distance from the fp32 mode ----------
    try:
        params = eng.get_params()

        def grads_of(e):
            s_, p_ = make_step(e, sync_tail=False)
            e.preprocess(p_, xcat, 4, True, pairs)
            e.forward(p_, pairs, None, defer_tail=True)
            e.siamese_head(p_, y, a.loss)
            e.backward(p_)
            torch.cuda.synchronize()
            return p_["emb"].clone(), (e.G / float(e.loss_scale)).clone(), s_
        e32 = make_engine("f32")
        e32.set_params(params)
        emb32, g32, step32 = grads_of(e32)
        ex["f32_storage_ms_per_step"] = timed(step32, reps=5) * 1e3
        del e32, step32
        torch.cuda.empty_cache()
        modes = {}
        for mode in ("f16", "bf16", "f32s"):
            if mode == a.dtype:
                restore(eng, snap)
                em, gm, _ = grads_of(eng)
                modes[mode] = {"ms_per_step": out["ms_per_step"], "audio_s_per_s": out["value"]}
            else:
                e_ = make_engine(mode)
                e_.set_params(params)
                em, gm, st_ = grads_of(e_)
                for _ in range(3):
                    st_()
                k_ = a.steps if mode != "f32s" else 5
                bl, _ = time_blocks(st_, 3 if mode != "f32s" else 1, k_)
                modes[mode] = {"ms_per_step": float(np.median(bl)) / k_ * 1e3, "audio_s_per_s": 2 * pairs * 3.0 * k_ / float(np.median(bl))}
                del e_, st_
                torch.cuda.empty_cache()
            modes[mode].update({"embedding_rel_err_vs_f32_mode": rel(em, emb32), "gradient_rel_err_vs_f32_mode": rel(gm, g32),
                                "gradient_cosine_vs_f32_mode": cos(gm, g32)})
        for mode, d in modes.items():
            ex["%s_mode" % mode] = d
        m = modes.get(a.dtype)
        if m is not None:
            out["precision"] = {"mode": a.dtype, "tolerance": 1e-3, "embedding_rel_err_vs_f32_mode": m["embedding_rel_err_vs_f32_mode"],
                                "meets_1e-3": bool(m["embedding_rel_err_vs_f32_mode"] < 1e-3),
                                "reference": "north star: embeddings within 1e-3 rel-tol of the reference arithmetic; measured here against the "
                                             "fp32-storage / fp32-MFMA mode of this library on the timed batch (that mode is 1e-6 from the float64 "
                                             "CPU oracle at this size); against_oracle: this mode against the CPU oracle itself, from the committed "
                                             "parity report of the -m gpu tests (not measured in this run)"}
            out["precision"]["against_oracle"] = oracle_referenced_figures(a.dtype)
        restore(eng, snap)
    except Exception as e:
        ex["modes_error"] = repr(e)
        restore(eng, snap)

    # ---- other loss, embed-only pass, data-path variants ----------------------------------------------------------------------
    try:
        other = "bce" if a.loss == "contrastive" else "contrastive"
        step_other, _ = make_step(eng, loss=other, sync_tail=False)
        ex["%s_loss_ms_per_step" % other] = timed(step_other) * 1e3
        pli = eng.plan(2 * pairs, L0, False)

        def embed_only():
            eng.preprocess(pli, xcat, 4, True, 2 * pairs)
            eng.forward(pli, 2 * pairs, None)
        t_embed = timed(embed_only, reps=20)
        ex["embed_only_audio_s_per_s"] = 2 * pairs * 3.0 / t_embed
        ex["embed_only_ms_per_256_windows"] = t_embed * 1e3
        # BASELINE.json config 1 on the GPU (the CPU leg of cpu_baseline.legs times the same step): the speaker classifier at batch 8
        ecls = type(eng)(BLOCKS, E, dropout=0.0, head="classifier", num_classes=40, dtype=a.dtype, device=dev, seed=1234)
        xc, lab = xcat[:8].contiguous(), torch.arange(8, device=dev, dtype=torch.int32) % 40
        pc = ecls.plan(8, L0, True)

        def cls_step():
            ecls.train_step_resident(pc, 8, lab, None, raw=xc, input_ready=True)
        t_cls = timed(cls_step, reps=20)
        ex["classifier_batch8_ms_per_step"] = t_cls * 1e3
        ex["classifier_batch8_audio_s_per_s"] = 8 * 3.0 / t_cls
        del ecls, pc
        # PCIe-inclusive step: the boundary handed host buffers (pinned int16 PCM of the same windows) -- never `value`
        host16 = (xcat.clamp(-1, 1) * 32767.0).round().to(torch.int16).cpu().pin_memory()
        dev16 = torch.empty_like(host16, device=dev)

        def step_h2d():
            dev16.copy_(host16, non_blocking=True)
            eng.preprocess(pl, dev16, 4, True, pairs)
            eng.forward(pl, pairs, None, defer_tail=True)
            eng.siamese_head(pl, y, a.loss)
            eng.backward(pl)
            eng.optimizer_step()
        ex["pcie_inclusive_ms_per_step_int16_host_windows"] = timed(step_h2d) * 1e3
        # device-resident corpus (voicemap_amd/shards.py): the 256 windows are start offsets into an int16 buffer in HBM and
        # the crop happens inside the preprocessing kernel -- the data path of experiments/train_siamese.py --device-data
        corpus = (torch.randn(64 * 1024 * 1024, device=dev) * 0.05 * 32767.0).clamp(-32767, 32767).to(torch.int16)
        offs = torch.randint(0, corpus.numel() - 48000, (2 * pairs,), device=dev, dtype=torch.int64)

        def step_offsets():
            eng.preprocess(pl, corpus, 4, True, pairs, offsets=offs, raw_len=48000)
            eng.forward(pl, pairs, None, defer_tail=True)
            eng.siamese_head(pl, y, a.loss)
            eng.backward(pl)
            eng.optimizer_step()
        ex["device_resident_corpus_ms_per_step_int16_offsets"] = timed(step_offsets) * 1e3
        del corpus
        restore(eng, snap)
    except Exception as e:
        ex["variants_error"] = repr(e)
        restore(eng, snap)

    # ---- cfg-B: the reference's CONTRASTIVE-LOSS script as it runs it (experiments/siamese_contrastive_loss.py:19-23,67-70): filters 32,
    # embedding 128, SpatialDropout1D 0.05 (masks drawn on the device), 32 pairs, contrastive loss.  SURVEY 8(d): 0.516 GFLOP and
    # 11.04 MB of algorithmic traffic per window in training -> 64 windows are 33 GFLOP / 0.71 GB: 13 us of MFMA, 88 us of HBM; the step is
    # bounded by its launch chain (dropout masks are per (window, channel), so the BatchNorm fold is off and the unfolded kernel sequence
    # runs), not by any tile shape.  Same engine at the headline's 128 pairs beside it. ----
    try:
        cb_blocks = [(32, 32, 4), (3, 64, 2), (3, 96, 2), (3, 128, 2)]
        cfgb = {}
        for pr in (32, pairs):
            eb = type(eng)(cb_blocks, 128, dropout=0.05, head="uniform_euclidean", dtype=a.dtype, device=dev, seed=1234)
            pb = eb.plan(2 * pr, L0, True)
            xb = torch.cat([xcat[:pr], xcat[pairs:pairs + pr]], 0).contiguous()
            yb = torch.cat([torch.zeros(pr // 2, device=dev), torch.ones(pr - pr // 2, device=dev)]).contiguous()

            def cfgb_step():
                eb.train_step_resident(pb, pr, yb, "contrastive", raw=xb, input_ready=True)     # (masks drawn on the device, every step)
            t_b = timed(cfgb_step, reps=20)
            t0 = time.perf_counter()
            for _ in range(50):
                cfgb_step()
            t_host = (time.perf_counter() - t0) / 50
            torch.cuda.synchronize()
            key = "pairs_%d" % pr
            cfgb[key] = {"ms_per_step": t_b * 1e3, "audio_s_per_s": 2 * pr * 3.0 / t_b, "host_enqueue_ms_per_step": t_host * 1e3,
                         "step_hbm_frac": 11.04e6 * 2 * pr / t_b / 8e12, "step_mfma_frac": 0.516e9 * 2 * pr / t_b / 2.5e15}
            del eb, pb
        cfgb["config"] = ("filters 32 (channels 32-64-96-128), embedding 128, dropout 0.05 with device-drawn masks, contrastive loss, "
                          "%s storage; pairs_32 is the reference script's own batch" % a.dtype)
        cfgb["kernels"] = ("block 1 conv1_fused; blocks 2-3 forward and all dgrads on the 128-wide tiles (conv_nt_glds_kernel: N-side "
                           "channels 32 / 64 / 96 are not multiples of the 256 x 128 tile's 128), block 4 forward conv_nt2r_kernel; wgrad "
                           "conv_tn_kernel family; unfolded BatchNorm / dropout / pool passes (bn_drop_pool_fwd, bn_pool_bwd_*)")
        ex["cfgB"] = cfgb
        torch.cuda.empty_cache()
    except Exception as e:
        ex["cfgB_error"] = repr(e)

    # ---- BASELINE.json config 4: the log-mel + 2-D CNN variant (not in the reference): one siamese training step
    # of 128 pairs of RAW 3 s clips -- vm_stft_logmel + four Conv2D 3x3 blocks (filters 32) + loss + backward + Adam ----
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
        t_sp = timed(spectro_step)
        t_ft = timed(lambda: seng.features(spl, xcat))
        ex.update({"logmel_2dcnn_train_ms_per_step": t_sp * 1e3, "logmel_2dcnn_audio_s_per_s": 2 * pairs * 3.0 / t_sp,
                   "logmel_frontend_ms_per_256_clips": t_ft * 1e3,
                   "logmel_2dcnn_config": "log-mel 298 x 64 (25 ms / 10 ms frames), Conv2D 3x3 channels 32-64-96-128, embedding 64, "
                                          "%d pairs, %s storage" % (pairs, a.dtype),
                   "logmel_2dcnn_precision": config4_precision(a.dtype)})
        del seng, spl
        torch.cuda.empty_cache()
    except Exception as e:  # the side figure must never take the headline line down
        ex["logmel_2dcnn_error"] = repr(e)

    # ---- BASELINE.json config 5: k-way n-shot evaluation.  (a) the reference-faithful loop of voicemap/utils.py:104-216 (tasks drawn
    # one by one with the reference's np.random sequence, k*n + 1 windows embedded per task); (b) the cached form: corpus embedded
    # once, tasks as row indices through vm_nshot_indexed, the pairwise-distance matrix through vm_pairdist_argmin.  Plus the accuracy
    # figures of the BASELINE metric: the reference's known-answer task and held-out synthetic speakers after a short training run. ----
    try:
        import tempfile
        from voicemap_amd import models as VM, retrieval as R, shards as VS, utils as VU
        from voicemap_amd.librispeech import SyntheticSpeechDataset
        with tempfile.TemporaryDirectory() as td:
            VS.write_shards(SyntheticSpeechDataset(num_speakers=48, files_per_speaker=8, seconds=3, seed=3), td)
            sd = VS.ShardedSpeechDataset(td, 3, stochastic=False)
            enc = VM.get_baseline_convolutional_encoder(F, E, dropout=0.0, dtype=a.dtype)
            net = VM.build_siamese_net(enc, (sd.fragment_length // 4, 1))
            net.compile(loss="binary_crossentropy", optimizer="adam")
            bp = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
            sd.to_device("cuda")
            tasks = 2040   # a multiple of the 51 tasks per launch: the warm-up run below creates every plan the timed run uses
            np.random.seed(5)
            VU.n_shot_task_evaluation(net, sd, bp, tasks, 1, 5, network_type="siamese", distance="euclidean")
            torch.cuda.synchronize()
            np.random.seed(6)
            t1 = time.perf_counter()
            VU.n_shot_task_evaluation(net, sd, bp, tasks, 1, 5, network_type="siamese", distance="euclidean")
            torch.cuda.synchronize()
            t_k = time.perf_counter() - t1
            np.random.seed(6)
            t1 = time.perf_counter()
            for _ in range(tasks):
                sd.build_n_shot_task_offsets(5, 1)
            t_draw = time.perf_counter() - t1
            ex.update({"kway_eval_5way_1shot_tasks_per_s": tasks / t_k, "kway_eval_audio_s_per_s": tasks * 6 * 3.0 / t_k,
                       "kway_eval_host_task_draw_share": t_draw / t_k,
                       "kway_eval_config": "%d tasks of 6 x 3 s windows from a device-resident int16 corpus (384 files), the reference's task-by-task "
                                           "semantics; the host draws the tasks with the reference's np.random sequence (that share of the time is "
                                           "kway_eval_host_task_draw_share)" % tasks})
            # (b) cached form
            R.embed_corpus(net, sd, bp)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cache = R.embed_corpus(net, sd, bp)
            torch.cuda.synchronize()
            t_emb = time.perf_counter() - t1
            np.random.seed(6)
            t1 = time.perf_counter()
            q_, s_ = R.draw_tasks_reference(sd, tasks, 5, 1)
            R.evaluate_tasks(cache, q_, s_, 5, 1, "euclidean")
            t_ref = time.perf_counter() - t1
            sampler = R.DeviceTaskSampler(sd, dev, seed=7)
            big = 200000
            sampler.draw(big, 5, 1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            q2, s2 = sampler.draw(big, 5, 1)
            R.evaluate_tasks(cache, q2, s2, 5, 1, "euclidean")
            torch.cuda.synchronize()
            t_dev = time.perf_counter() - t1
            ex.update({"kway_cached_embed_corpus_files_per_s": len(sd) / t_emb, "kway_cached_embed_corpus_audio_s_per_s": len(sd) * 3.0 / t_emb,
                       "kway_cached_tasks_per_s_reference_order_draws": tasks / t_ref, "kway_cached_tasks_per_s_device_sampler": big / t_dev,
                       "kway_cached_config": "corpus of %d files embedded once (each window whitened alone), tasks = row indices through vm_nshot_indexed; "
                                             "reference-order draws: %d tasks; device sampler: %d tasks in one launch" % (len(sd), tasks, big)})
            del net, enc, cache
        # pairwise-distance matrix at train-clean-360's size on one GPU's row shard (N = 104 014 files, E = 64; 8 ranks -> 13 002 rows each)
        n360, rows8 = 104014, 13002
        embm = torch.randn(n360, E, device=dev)
        ws = torch.empty(eng.lib.query("vm_pairdist_workspace_bytes", rows8, n360) // 4 + 16, device=dev)
        bv, bi = torch.empty(rows8, device=dev), torch.empty(rows8, dtype=torch.int32, device=dev)
        st_ = torch.cuda.current_stream(dev).cuda_stream
        fn = lambda: eng.lib.call("vm_pairdist_argmin", embm.data_ptr(), embm.data_ptr(), rows8, n360, E, 0, 0, None, bv.data_ptr(), bi.data_ptr(),
                                  ws.data_ptr(), st_)
        t_pd = timed(fn, reps=5, warm=1)
        ex.update({"pairdist_ms_per_rank_shard_13002x104014x64": t_pd * 1e3,
                   "pairdist_gpairs_per_s": rows8 * n360 / t_pd / 1e9,
                   "pairdist_config": "vm_pairdist_argmin, euclidean, argmin only: one rank's 1/8 row shard of a train-clean-360-sized (104 014 x 64) "
                                      "embedding matrix against all of it",
                   # the kernel's bound is fp32 VALU issue: the direct form costs a subtract and an FMA per component pair, a wave-64
                   # instruction occupies its SIMD for 2 clocks (packed fp32 forms issue at half that rate: no gain on gfx950)
                   "pairdist_roofline": {"bound": "valu", "unit": "G wave-instructions/s",
                                         "achieved": 2.0 * rows8 * n360 * E / 64 / t_pd / 1e9, "peak": 1024 * 2.4 / 2.0,
                                         "frac": (2.0 * rows8 * n360 * E / 64 / t_pd / 1e9) / (1024 * 2.4 / 2.0),
                                         "note": "2 x M x N x E / 64 wave instructions (subtract + FMA per component pair; addressing, LDS reads "
                                                 "and the argmin are not counted) against 1024 SIMDs x 2.4 GHz / 2 clocks"}})
        del embm, ws
        torch.cuda.empty_cache()
    except Exception as e:
        ex["kway_eval_error"] = repr(e)
    try:
        ex.update(accuracy_figures(a, dev))
    except Exception as e:
        ex["accuracy_error"] = repr(e)
    restore(eng, snap)
    return ex


def oracle_referenced_figures(dtype):
    """Embedding error of storage mode ``dtype`` against the CPU ORACLE, read from the parity report the -m gpu tests wrote
    (profiles/r06_parity_report.csv, committed: one full run of the suite on an MI355X, NOT this run): the bench batch at fresh-init
    weights (tests/test_gpu_fullsize_oracle.py), the same batch at a trained-like BatchNorm / bias state, the reference's shipped
    checkpoint on its own 8 LibriSpeech clips in training mode (tests/test_gpu_golden_step.py) and -- f16, round 6 -- the spread of the
    guard sweep (tests/test_gpu_f16_guard.py: five seeds of each synthetic state and a state trained for 1 200 steps).
    ``meets_1e-3`` per entry."""
    want = {"full_size_oracle[%s]" % dtype: ("emb_rel_err_vs_fp64_oracle", "bench_batch_fresh_init"),
            "full_size_oracle_trained_state[%s]" % dtype: ("emb_rel_err_vs_fp32_oracle", "bench_batch_trained_like_batchnorm_and_bias_state"),
            "golden_step_cfgCK_real_clips[%s]" % dtype: ("emb_rel_err", "reference_checkpoint_on_its_8_librispeech_clips_training_mode")}
    guard = {"f16_guard[fresh_init]": "guard_5_seeds_fresh_init", "f16_guard[trained_like]": "guard_5_seeds_trained_like_state",
             "f16_guard[trained_1200_steps]": "guard_state_trained_1200_steps_noise_and_unseen_speakers"}
    out = {"source": "profiles/r06_parity_report.csv (tests/test_gpu_fullsize_oracle.py, tests/test_gpu_golden_step.py, tests/test_gpu_f16_guard.py "
                     "on an MI355X; NOT this run)"}
    try:
        spread = {}
        with open(os.path.join(ROOT, "profiles", "r06_parity_report.csv")) as f:
            for line in f:
                parts = line.strip().split(",")
                if len(parts) != 3:
                    continue
                if parts[0] in want and parts[1] == want[parts[0]][0]:
                    v = float(parts[2])
                    out[want[parts[0]][1]] = {"embedding_rel_err": v, "meets_1e-3": bool(v < 1e-3)}
                elif dtype == "f16" and parts[0] in guard and parts[1] in ("emb_rel_err_min", "emb_rel_err_max", "cases"):
                    spread.setdefault(guard[parts[0]], {})[parts[1]] = float(parts[2])
        for k, v in spread.items():
            if "emb_rel_err_max" in v:
                out[k] = {"embedding_rel_err_min": v.get("emb_rel_err_min"), "embedding_rel_err_max": v["emb_rel_err_max"],
                          "cases": int(v.get("cases", 0)), "meets_1e-3": bool(v["emb_rel_err_max"] < 1e-3)}
    except (OSError, ValueError):
        out["error"] = "parity report not found"
    return out


def config4_precision(dtype):
    """BASELINE.json config 4 (log-mel + 2-D CNN; NOT in the reference, so its only oracle is this repository's own float64
    restatement: parity unpinned by construction) at its own size, from the same committed parity report."""
    key = {"f16": "spectro_step_f16_drop0_298x64_F32", "bf16": "spectro_step_bf16_drop0_298x64_F32", "f32": "spectro_step_f32_drop0_298x64_F32"}.get(dtype)
    out = {"source": "profiles/r06_parity_report.csv (tests/test_gpu_spectro.py at 298 x 64; NOT this run)", "oracle": "this repository's own float64 "
           "restatement (oracle/voicemap_oracle.py: the variant does not exist in the reference) -- parity unpinned"}
    guard = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r06_parity_report.csv")) as f:
            for line in f:
                parts = line.strip().split(",")
                if len(parts) == 3 and parts[0] == key and parts[1] == "emb_rel_err":
                    v = float(parts[2])
                    out.update({"embedding_rel_err": v, "meets_1e-3": bool(v < 1e-3)})
                if dtype == "f16" and len(parts) == 3 and parts[0] == "config4_f16_guard":   # other parameter / clip seeds, forward only (round 6)
                    guard.setdefault("one_plane_image" if parts[1].startswith("one_plane") else "two_plane_image", []).append(float(parts[2]))
        for k, vs in guard.items():
            out["guard_seeds_" + k] = {"embedding_rel_err_min": min(vs), "embedding_rel_err_max": max(vs), "cases": len(vs), "meets_1e-3": bool(max(vs) < 1e-3)}
        if dtype in ("f16", "bf16"):
            out["image"] = ("the log-mel image on two planes of the storage type, block 1's boundary recomputed from it (ABI 10, DESIGN.md section 2); "
                            "one plane measured 1.27e-3 in f16 (rounds 3-5)")
        if out.get("meets_1e-3") is False:
            out["note"] = "config 4 has no 16-bit storage mode inside the 1e-3 tolerance (its fp32 mode is: 1e-6); the figure stands as measured"
    except (OSError, ValueError):
        out["error"] = "parity report not found"
    return out


def accuracy_figures(a, dev):
    """The "k-way verif acc" half of the BASELINE metric, two figures that need no LibriSpeech on the box:
    (1) the reference's own known-answer 5-way 1-shot task (notebooks/Human_Evaluation.ipynb cell 8: "The correct answer was 5") on its
        only shipped checkpoint (tests/golden/, extracted by tests/golden/extract_reference_fixtures.py) in this run's storage mode;
    (2) 5-way 1-shot accuracy on HELD-OUT synthetic speakers after a short cfg-A training run on other synthetic speakers (device-
        resident corpus, the pipeline of experiments/train_siamese.py --device-data), evaluated on the cached embeddings."""
    import tempfile
    from voicemap_amd import models as VM, retrieval as R, shards as VS, utils as VU
    from voicemap_amd.keras_like import Adam
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    fig = {}
    g = os.path.join(ROOT, "tests", "golden")
    ck = VM.load_keras_checkpoint_npz(os.path.join(g, "ckpt_cfgCK_weights.npz"), dtype=a.dtype)
    c = np.load(os.path.join(g, "clips_human_eval.npz"))
    q = c["query"].astype(np.float32) / 32768.0
    s = c["support"].astype(np.float32) / 32768.0
    bp = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
    ([i1, i2], _) = bp(([np.stack([q] * 5)[:, :, None], s[:, :, None]], []))   # each side whitened as its own batch (utils.py:126-133)
    pred = np.asarray(ck.predict([i1, i2]))[:, 0]
    fig["kway_known_answer_task_pick_1based"] = int(np.argmin(pred)) + 1
    fig["kway_known_answer_task_correct"] = bool(int(np.argmin(pred)) + 1 == int(c["correct_answer_1based"]))
    del ck
    steps, bs = 240, 64
    with tempfile.TemporaryDirectory() as td:
        VS.write_shards(SyntheticSpeechDataset(num_speakers=64, files_per_speaker=8, seconds=3, seed=0), os.path.join(td, "train"))
        VS.write_shards(SyntheticSpeechDataset(num_speakers=40, files_per_speaker=6, seconds=3, seed=1, subset="heldout"), os.path.join(td, "valid"))
        train = VS.ShardedSpeechDataset(os.path.join(td, "train"), 3, stochastic=True)
        valid = VS.ShardedSpeechDataset(os.path.join(td, "valid"), 3, stochastic=False)
        train.to_device("cuda")
        valid.to_device("cuda")
        torch.manual_seed(1)
        np.random.seed(1)
        enc = VM.get_baseline_convolutional_encoder(F, E, dropout=0.0, dtype=a.dtype)
        net = VM.build_siamese_net(enc, (train.fragment_length // 4, 1))
        net.compile(loss="binary_crossentropy", optimizer=Adam(clipnorm=1.0), metrics=["accuracy"])
        sampler = R.DeviceTaskSampler(valid, dev, seed=3)
        q0, s0 = sampler.draw(5000, 5, 1)
        acc0 = R.evaluate_tasks(R.embed_corpus(net, valid, bp), q0, s0, 5, 1, "euclidean") / 5000.0
        gen = (bp(b) for b in train.yield_verification_batches_device(bs))
        t1 = time.perf_counter()
        for _ in range(steps):
            x, yb = next(gen)
            net.train_on_batch(x, yb)
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t1
        cache = R.embed_corpus(net, valid, bp)
        acc1 = R.evaluate_tasks(cache, q0, s0, 5, 1, "euclidean") / 5000.0
        nn = R.pairwise_retrieval(cache, "euclidean")["accuracy"]
    fig.update({"kway_5way_1shot_acc_heldout_synthetic_untrained": acc0, "kway_5way_1shot_acc_heldout_synthetic_after_training": acc1,
                "nearest_neighbour_same_speaker_acc_heldout_synthetic_after_training": nn,
                "kway_acc_config": "%d Adam steps of %d pairs (BCE, cfg-A, %s storage) on 64 synthetic speakers from a device-resident corpus "
                                   "(%.1f ms per step incl. the host's pair draws), then 5000 5-way 1-shot tasks on 40 held-out synthetic speakers "
                                   "(cached embeddings, euclidean); chance = 0.2" % (steps, bs, a.dtype, t_train / steps * 1e3)})
    return fig


if __name__ == "__main__":
    main()
