"""The N > 1 paths on CPU: world_size-2 gloo processes driving the SAME hooks the HIP engine drives on a GPU node
(voicemap_amd/parallel.py): the two-collective gradient sum (``begin_tail`` from backward + the ``grad_sync`` hook from the
optimizer step) over the engine's own flat-buffer layout (``engine.flat_layout`` / ``FlatState``), ``broadcast_state``,
the data-parallel ``fit_generator`` loop of ``models._TrainableModel`` (per-rank batches, reduced epoch logs, rank-0-only
file callbacks) and the sharded k-way evaluation of ``utils.n_shot_task_evaluation``.  There is no GPU in this container,
so the arithmetic of a step (gradients, Adam) comes from the CPU oracle; everything around it is the product code."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voicemap_amd import parallel


def test_shard_range_partitions_exactly():
    for n in (1, 7, 128, 1023):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_layout_matches_keras_trainable_order():
    from voicemap_amd.engine import flat_layout
    blocks = [(32, 8, 4), (3, 16, 2), (3, 24, 2), (3, 32, 2)]
    spec, offsets, n_flat, nt_off, n_nt = flat_layout(blocks, 8, "uniform_euclidean")
    names = [n for n, _ in spec]
    assert names[:4] == ["conv1.kernel", "conv1.bias", "bn1.gamma", "bn1.beta"] and names[-4:] == ["dense.kernel", "dense.bias",
                                                                                                  "head.kernel", "head.bias"]
    assert all(o % 64 == 0 for o, _, _ in offsets.values()) and n_flat % 64 == 0
    assert offsets["conv2.kernel"][0] > 0 and offsets["conv2.kernel"][2] == (3, 8, 16)
    assert list(nt_off) == [f"bn{i}.{s}" for i in range(1, 5) for s in ("moving_mean", "moving_variance")]
    with pytest.raises(NotImplementedError):
        flat_layout(blocks, 8, "cosine_distance")


# ---------------------------------------------------------------------------------------------------------
class OracleBackedSiamese:
    """models._TrainableModel with the oracle as its arithmetic: the flat buffers are the engine's (FlatState on the CPU), a
    train step writes the oracle's gradients into G exactly where the HIP kernels would, then runs the engine-side sequence
    ``begin_tail`` (from backward) -> ``grad_sync`` -> clip + Adam with the 1/world prescale (optimizer_step)."""

    def __new__(cls, *a, **kw):
        from voicemap_amd.models import _TrainableModel

        class _M(_TrainableModel):
            def __init__(self, arch, seed):
                super().__init__()
                from oracle import voicemap_oracle as O
                from voicemap_amd.engine import FlatState
                self.O, self.arch = O, arch
                self.names = O.param_names(arch)
                st = FlatState(arch.blocks, arch.embedding_dimension, "uniform_euclidean", 0, "cpu")
                st.grad_sync, st.grad_prescale, st.side_stream = None, 1.0, None
                st.lr = 1e-3
                p = O.init_params(arch, head="uniform_euclidean", seed=seed)
                for k, v in p.items():
                    st.view(k).copy_(v.to(torch.float32).reshape(st.view(k).shape))
                self.engine = st
                self.adam = O.AdamState()
                self.loss = "contrastive_loss"

            def _ensure_engine(self):
                return self.engine

            def _params(self):
                st = self.engine
                return {k: st.view(k).to(torch.float64).clone() for k in list(st.offsets) + list(st.nt_off)}

            def train_on_batch(self, x, y):
                O, st = self.O, self.engine
                p = self._params()
                res = O.siamese_train_step(self.arch, p, None, torch.tensor(x[0]), torch.tensor(x[1]), torch.tensor(y))
                st.G.zero_()
                for k in self.names:   # what backward() leaves in the flat gradient buffer
                    st.view(k, st.G).copy_(res["grads"][k].to(torch.float32).reshape(st.view(k, st.G).shape))
                if st.grad_sync is not None:
                    st.grad_sync.begin_tail(st, None)   # HipEncoderEngine.backward, after block 2's gradients
                    st.grad_sync(st.G)                  # HipEncoderEngine.optimizer_step
                g = {k: st.view(k, st.G).to(torch.float64) * st.grad_prescale for k in self.names}
                self.adam.lr, self.adam.clipnorm = st.lr, 1.0
                new = O.adam_step(self.adam, {k: p[k] for k in self.names}, g)
                for k in self.names:
                    st.view(k).copy_(new[k].to(torch.float32).reshape(st.view(k).shape))
                st.iterations += 1
                return float(res["loss"]), float(res["acc"])

            def test_on_batch(self, x, y):
                O = self.O
                pr, _, _ = O.siamese_forward(self.arch, self._params(), torch.tensor(x[0]), torch.tensor(x[1]), training=False)
                pr = pr.reshape(-1)
                yt = torch.tensor(y).reshape(-1)
                return float(O.contrastive_loss(yt, pr)), float(O.binary_accuracy(yt, pr))

            def get_lr(self):
                return self.engine.lr

            def set_lr(self, lr):
                self.engine.lr = float(lr)

        return _M(*a, **kw)


def _batches(rank, pairs, length):
    from oracle import voicemap_oracle as O
    g = np.random.default_rng(100 + rank)   # every rank draws its OWN batches
    while True:
        x1 = O.whiten(g.normal(0, 0.05, (pairs, length, 1)))
        x2 = O.whiten(g.normal(0, 0.05, (pairs, length, 1)))
        y = np.concatenate([np.zeros(pairs // 2), np.ones(pairs - pairs // 2)])[:, None]
        yield [x1, x2], y


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle import voicemap_oracle as O
    from voicemap_amd import keras_like as K
    from voicemap_amd import utils as U
    r_, w_, _ = parallel.init_distributed("gloo")
    assert (r_, w_) == (rank, world) and parallel.rank_world() == (rank, world)
    arch = O.EncoderArch.baseline(8, 8, dropout=0.0)
    model = OracleBackedSiamese(arch, seed=3 + rank)      # replicas start DIFFERENT: broadcast_state must fix that
    st = model.engine
    res = {}

    # ---- 1. one step by hand: sum of local gradients, identical replicas --------------------------------------
    parallel.attach(st, world)
    parallel.broadcast_state(st)
    p0 = [torch.zeros_like(st.P) for _ in range(world)]
    dist.all_gather(p0, st.P)
    res["broadcast_same"] = all(torch.equal(p0[0], t) for t in p0)
    (x, y) = next(_batches(rank, 4, 400))
    ref = O.siamese_train_step(arch, model._params(), None, torch.tensor(x[0]), torch.tensor(x[1]), torch.tensor(y))
    local = torch.zeros_like(st.G)
    for k in model.names:
        o, n, _ = st.offsets[k]
        local[o:o + n] = ref["grads"][k].to(torch.float32).reshape(-1)
    model.train_on_batch(x, y)
    locals_ = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(locals_, local)
    res["sum_ok"] = torch.allclose(st.G, sum(locals_), rtol=1e-6, atol=1e-9)   # G now holds the SUM (prescale is in the optimizer)
    res["collectives"] = st.grad_sync.collectives                                # two per step: tail early, block-1 head in the hook
    res["split"] = st.grad_sync._split == st.offsets["conv2.kernel"][0]
    ps = [torch.zeros_like(st.P) for _ in range(world)]
    dist.all_gather(ps, st.P)
    res["replicas_same_after_step"] = all(torch.equal(ps[0], t) for t in ps)
    res["prescale"] = st.grad_prescale

    # ---- 2. the data-parallel fit_generator loop ---------------------------------------------------------------
    class Marker(K.Callback):
        rank0_only = True

        def on_epoch_end(self, epoch, logs=None):
            open(os.path.join(outdir, "marker_rank%d_epoch%d" % (rank, epoch)), "w").close()

    class Everywhere(K.Callback):
        def on_epoch_end(self, epoch, logs=None):
            logs["seen_by_all"] = 1.0

    csv = K.CSVLogger(os.path.join(outdir, "log.csv"))
    plateau = K.ReduceLROnPlateau(monitor="val_loss", patience=1, factor=0.5, min_delta=10.0)  # never "better": lr halves on epoch 2
    hist = model.fit_generator(_batches(rank, 4, 400), steps_per_epoch=2, epochs=3, verbose=0, workers=0,
                               validation_data=_batches(10 + rank, 4, 400), validation_steps=2,
                               callbacks=[Everywhere(), Marker(), csv, plateau])
    logs = torch.tensor([hist.history[k] for k in ("loss", "acc", "val_loss", "val_acc", "lr")], dtype=torch.float64)
    alll = [torch.zeros_like(logs) for _ in range(world)]
    dist.all_gather(alll, logs)
    res["logs_same"] = all(torch.equal(alll[0], t) for t in alll)
    res["lr_final"] = model.get_lr()
    ps = [torch.zeros_like(st.P) for _ in range(world)]
    dist.all_gather(ps, st.P)
    res["replicas_same_after_fit"] = all(torch.equal(ps[0], t) for t in ps)
    res["seen_by_all"] = hist.history.get("seen_by_all") == [1.0, 1.0, 1.0]

    # ---- 3. sharded k-way evaluation: every rank evaluates its share, one all-reduce of n_correct ---------------
    seen = []
    orig = U._n_shot_local
    U._n_shot_local = lambda model_, ds, pre, num, n, k, kind, distance: (seen.append(num), num - 1 if num else 0)[1]
    try:
        total = U.n_shot_task_evaluation(None, None, None, 11, 1, 5)
    finally:
        U._n_shot_local = orig
    res["nshot_total"] = total          # (6 - 1) + (5 - 1) with two ranks
    res["nshot_local"] = seen[0]
    res["max"] = parallel.max_over_ranks(1.0 + rank)
    res["sum"] = parallel.sum_over_ranks(1.0 + rank)
    # ---- 4. the row all-gather of the cached-embedding evaluation (retrieval.embed_corpus): unequal shards, rank order ----------
    from voicemap_amd import retrieval
    full = torch.arange(5 * 3, dtype=torch.float32).reshape(5, 3)
    lo, hi = parallel.shard_range(5, rank, world)
    res["gather_rows"] = bool(torch.equal(retrieval.all_gather_rows(full[lo:hi].clone(), 5), full))
    torch.save(res, os.path.join(outdir, "res%d.pt" % rank))
    parallel.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(str(tmp_path / ("res%d.pt" % r))) for r in (0, 1))
    for r in (r0, r1):
        assert r["broadcast_same"] and r["sum_ok"] and r["split"] and r["replicas_same_after_step"]
        assert r["collectives"] == 2 and r["prescale"] == 0.5
        assert r["logs_same"] and r["replicas_same_after_fit"] and r["seen_by_all"]
        assert r["lr_final"] == pytest.approx(1e-3 * 0.5 * 0.5)   # patience 1: reductions after epochs 2 and 3 -- on BOTH ranks
        assert r["nshot_total"] == 9 and r["max"] == 2.0 and r["sum"] == 3.0 and r["gather_rows"]
    assert (r0["nshot_local"], r1["nshot_local"]) == (6, 5)
    files = sorted(os.listdir(str(tmp_path)))
    assert [f for f in files if f.startswith("marker_")] == ["marker_rank0_epoch%d" % e for e in range(3)]   # rank-0-only callback
    rows = open(str(tmp_path / "log.csv")).read().strip().splitlines()
    assert len(rows) == 4 and rows[0].split(",")[0] == "epoch"


@pytest.mark.gpu
def test_two_rank_hip_engine_replicas_bit_identical(tmp_path):
    """Two ranks on two visible GPUs with the HIP engine (skipped when the box has one): bit-identical P/M/V across ranks
    after data-parallel steps on different per-rank batches, and G == sum of the local gradients."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 visible GPUs")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        res = torch.load(str(tmp_path / ("gres%d.pt" % r)))
        assert res["same_P"] and res["same_M"] and res["same_V"] and res["sum_ok"] and res["finite"]


@pytest.mark.gpu
def test_two_rank_hip_engine_on_one_gpu_over_gloo(tmp_path):
    """The same check on ONE GPU: two processes share cuda:0 and exchange the flat gradient buffer through gloo (RCCL refuses two ranks
    on one device).  Everything but the transport is the production path: HipEncoderEngine, broadcast_state, the early tail
    all-reduce enqueued from the side stream, the head all-reduce in the optimizer hook, the 1/world prescale in the Adam kernel."""
    port = 29500 + ((os.getpid() + 7) % 2000)
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path), "gloo"), nprocs=2, join=True)
    for r in (0, 1):
        res = torch.load(str(tmp_path / ("gres%d.pt" % r)))
        assert res["same_P"] and res["same_M"] and res["same_V"] and res["sum_ok"] and res["finite"]
        assert res["collectives_per_step"] == 2
        assert res["replay_programs"] == 1 and res["replay_same_as_eager"] and res["replay_collectives"] == (14, 14)
        assert res["cache_rows"] == 45 and res["cache_same"]
        assert res["nshot_sharded"] == res["nshot_single"] and res["retrieval_sharded"] == res["retrieval_single"]


@pytest.mark.gpu
def test_sync_bn_two_ranks_equal_one_device_with_the_global_batch(tmp_path):
    """SyncBN (engine.sync_bn): two ranks x 4 pairs == one device x 8 pairs -- loss, gradients and the weights after an Adam
    step -- because the BatchNorm statistics and the two BatchNorm-backward means are all-reduced; without it the replicas normalise
    over their own 4 pairs and the gradients differ visibly (asserted too)."""
    port = 29500 + ((os.getpid() + 13) % 2000)
    mp.spawn(_sync_bn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)   # both storage modes in one pair of processes
    from tests.gpu_util import report
    for dtype in ("f32", "f16"):
        res = torch.load(str(tmp_path / ("syncbn_%s.pt" % dtype)))
        for k, v in res.items():
            report("sync_bn_2_ranks_x_4_pairs_vs_1_device_x_8_pairs[%s]" % dtype, k, v)
        assert res["grad_rel_err_sync"] < {"f32": 2e-5, "f16": 3e-2}[dtype], res
        assert res["param_rel_err_sync"] < {"f32": 1e-6, "f16": 1e-3}[dtype], res
        assert abs(res["loss_sync"] - res["loss_single"]) < {"f32": 1e-6, "f16": 2e-3}[dtype], res
        assert res["moving_rel_err_sync"] < {"f32": 1e-6, "f16": 2e-3}[dtype], res
        assert res["grad_rel_err_plain"] > 10 * res["grad_rel_err_sync"] and res["grad_rel_err_plain"] > 1e-2, res
        assert res["collectives_sync"] > res["collectives_plain"] == 2


def _sync_bn_worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from voicemap_amd.engine import HipEncoderEngine
    parallel.init_distributed("gloo")
    torch.cuda.set_device(0)
    blocks = [(32, 16, 4), (3, 32, 2), (3, 48, 2), (3, 64, 2)]
    g = np.random.default_rng(77)
    pairs, l0 = 8, 1600
    x1 = g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32)
    x2 = g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32)
    y = (g.random(pairs) < 0.5).astype(np.float32)[:, None]
    lo, hi = rank * pairs // world, (rank + 1) * pairs // world

    def trained_like(e):   # non-trivial BatchNorm parameters and biases, the same on every engine
        gg = torch.Generator().manual_seed(3)
        for i in range(1, 5):
            e.view("bn%d.gamma" % i).copy_(1.0 + 0.25 * torch.randn(e.view("bn%d.gamma" % i).shape, generator=gg))
            e.view("bn%d.beta" % i).copy_(0.2 * torch.randn(e.view("bn%d.beta" % i).shape, generator=gg))
            e.view("conv%d.bias" % i).copy_(0.1 * torch.randn(e.view("conv%d.bias" % i).shape, generator=gg))
        e.refresh_weights()

    for dtype in ("f32", "f16"):
        def run(sync_bn):
            e = HipEncoderEngine(blocks, 16, dropout=0.0, head="uniform_euclidean", dtype=dtype, seed=5)
            trained_like(e)
            e.sync_bn = sync_bn
            parallel.attach(e, world)
            parallel.broadcast_state(e)
            c0 = dist_collectives()
            pl = e.siamese_train_step(x1[lo:hi], x2[lo:hi], y[lo:hi], drop_masks=None, apply_update=True)
            torch.cuda.synchronize()
            losses = [torch.zeros(1, device="cuda") for _ in range(world)]
            dist.all_gather(losses, pl["loss_acc"][:1].float().clone())
            return e, float(sum(losses).item() / world), dist_collectives() - c0

        counter = {"n": 0}
        real_all_reduce = dist.all_reduce

        def counting_all_reduce(*a, **k):
            counter["n"] += 1
            return real_all_reduce(*a, **k)
        dist.all_reduce = counting_all_reduce
        dist_collectives = lambda: counter["n"]
        e_sync, loss_sync, n_sync = run(True)
        e_plain, loss_plain, n_plain = run(False)
        dist.all_reduce = real_all_reduce
        if rank == 0:
            single = HipEncoderEngine(blocks, 16, dropout=0.0, head="uniform_euclidean", dtype=dtype, seed=5)
            trained_like(single)
            pl = single.siamese_train_step(x1, x2, y, drop_masks=None, apply_update=True)
            torch.cuda.synchronize()
            rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
            scale = lambda e: e.G / float(e.loss_scale)
            mov = lambda e: torch.cat([e.view("bn%d.moving_%s" % (i, k)).reshape(-1) for i in range(1, 5) for k in ("mean", "variance")])
            torch.save({"grad_rel_err_sync": rel(scale(e_sync) * 0.5, scale(single)), "grad_rel_err_plain": rel(scale(e_plain) * 0.5, scale(single)),
                        "param_rel_err_sync": rel(e_sync.P, single.P), "loss_sync": loss_sync, "loss_plain": loss_plain,
                        "loss_single": float(pl["loss_acc"][0].item()), "moving_rel_err_sync": rel(mov(e_sync), mov(single)),
                        "collectives_sync": n_sync, "collectives_plain": n_plain}, os.path.join(outdir, "syncbn_%s.pt" % dtype))
    dist.barrier()
    dist.destroy_process_group()


def _gpu_worker(rank, world, port, outdir, backend="nccl"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if backend == "nccl" else 0), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from voicemap_amd.engine import HipEncoderEngine
    parallel.init_distributed(backend)
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    blocks = [(32, 16, 4), (3, 32, 2), (3, 48, 2), (3, 64, 2)]
    eng = HipEncoderEngine(blocks, 16, dropout=0.0, head="uniform_euclidean", dtype="bf16", seed=5 + rank)
    g = np.random.default_rng(50 + rank)
    pairs, l0 = 8, 1600
    x1 = g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32)
    x2 = g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32)
    y = np.concatenate([np.zeros(pairs // 2), np.ones(pairs // 2)])[:, None]
    # local gradient of step 1 without the hook (same weights on both ranks after the broadcast)
    parallel.attach_if_distributed(eng)
    sync, eng.grad_sync = eng.grad_sync, None
    eng.siamese_train_step(x1, x2, y, drop_masks=None, apply_update=False)
    local = eng.G.clone()
    eng.grad_sync = sync
    c0 = sync.collectives
    pl = eng.siamese_train_step(x1, x2, y, drop_masks=None, apply_update=True)
    c1 = sync.collectives
    torch.cuda.synchronize()
    locals_ = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(locals_, local)
    res = {"sum_ok": bool(torch.allclose(eng.G, sum(locals_), rtol=1e-5, atol=1e-7)), "collectives_per_step": c1 - c0}
    for _ in range(3):
        pl = eng.siamese_train_step(x1, x2, y, drop_masks=None)
    torch.cuda.synchronize()
    for name in ("P", "M", "V"):
        t = getattr(eng, name)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        res["same_" + name] = bool(all(torch.equal(parts[0], q) for q in parts))
    res["finite"] = bool(torch.isfinite(pl["loss_acc"]).all().item())
    # round 6: a data-parallel step is recorded and replayed like a single-GPU one (the two collectives are host calls of the program,
    # engine._host_call): a replaying engine and an eager one, same seed, same batches, hold the same bits after 7 steps
    from voicemap_amd.engine import _Program
    ea = HipEncoderEngine(blocks, 16, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=9)
    eb = HipEncoderEngine(blocks, 16, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=9)
    eb.replay = False
    for e in (ea, eb):
        parallel.attach_if_distributed(e)
    for k in range(7):
        xa = g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32)
        xb = g.normal(0, 0.05, (pairs, l0, 1)).astype(np.float32)
        for e in (ea, eb):
            e.siamese_train_step(xa, xb, y, drop_masks=None)
    torch.cuda.synchronize()
    res["replay_programs"] = sum(isinstance(p_, _Program) for p_ in ea._programs.values())
    res["replay_same_as_eager"] = bool(all(torch.equal(getattr(ea, nm_).view(torch.int32), getattr(eb, nm_).view(torch.int32)) for nm_ in ("P", "M", "V", "G")))
    res["replay_collectives"] = (ea.grad_sync.collectives, eb.grad_sync.collectives)
    # BASELINE.json config 5 sharded: every rank embeds its rows of the corpus, the (N, E) matrix is all-gathered, tasks / query rows
    # are split over ranks and one integer is summed -- every rank must hold the single-process answers
    from voicemap_amd import models as VM, retrieval as R, utils as VU
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    torch.manual_seed(4)
    ds = SyntheticSpeechDataset(num_speakers=9, files_per_speaker=5, seconds=1, stochastic=False, seed=5)
    enc = VM.get_baseline_convolutional_encoder(16, 32, dropout=0.0, dtype="f32")
    net = VM.build_siamese_net(enc, (ds.fragment_length // 4, 1))
    net.compile(loss="binary_crossentropy", optimizer="adam")
    pre = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
    cache = R.embed_corpus(net, ds, pre)                      # sharded rows + all-gather
    np.random.seed(21)                                        # the same stream on both ranks: each evaluates its own share of ...
    q, s_ = R.draw_tasks_reference(ds, 40, 4, 2)              # ... these 40 tasks below (drawn here once to have the single-rank answer)
    res["cache_rows"] = int(cache.emb.shape[0])
    res["nshot_single"] = R.evaluate_tasks(cache, q, s_, 4, 2, "cosine")
    lo, hi = parallel.shard_range(40, rank, world)
    res["nshot_sharded"] = int(round(parallel.sum_over_ranks(float(R.evaluate_tasks(cache, q[lo:hi], s_[lo:hi], 4, 2, "cosine")))))
    res["retrieval_sharded"] = R.pairwise_retrieval(cache, "euclidean")["n_correct"]
    res["retrieval_single"] = R.pairwise_retrieval(cache, "euclidean", rows=(0, cache.n))["n_correct"]
    sums = [torch.zeros(1, device="cuda") for _ in range(world)]
    dist.all_gather(sums, cache.emb.double().sum().float().reshape(1))
    res["cache_same"] = bool(all(torch.equal(sums[0], t) for t in sums))
    torch.save(res, os.path.join(outdir, "gres%d.pt" % rank))
    parallel.barrier()
    dist.destroy_process_group()
