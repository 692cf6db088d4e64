"""-m gpu parity tests of the cached-embedding evaluation (BASELINE.json config 5; voicemap_amd/retrieval.py, vm_nshot_indexed,
vm_pairdist_argmin) against the float64 oracle (O.n_shot_prediction = voicemap/utils.py:159-206) and numpy."""
import numpy as np
import pytest
import torch

from oracle import voicemap_oracle as O
from tests.gpu_util import L, dev, p, rel_err, report, stream

pytestmark = pytest.mark.gpu
DIST = {"euclidean": 0, "cosine": 1, "dot_product": 2}


def _pairwise_ref(q, r, dist):
    q, r = q.astype(np.float64), r.astype(np.float64)
    if dist == "euclidean":
        return np.sqrt(((q[:, None, :] - r[None, :, :]) ** 2).sum(-1))
    if dist == "cosine":
        return 1.0 - (q @ r.T) / (np.linalg.norm(q, axis=1)[:, None] * np.linalg.norm(r, axis=1)[None, :])
    return -(q @ r.T)


@pytest.mark.parametrize("dist", ["euclidean", "cosine", "dot_product"])
@pytest.mark.parametrize("k,n,E", [(5, 1, 64), (20, 5, 64), (3, 2, 128), (7, 3, 100), (2, 4, 256)])
def test_nshot_indexed_matches_oracle(dist, k, n, E):
    r = np.random.default_rng(k * 100 + n)
    rows, tasks = 300, 257
    emb = r.normal(0, 1, (rows, E)).astype(np.float32)
    qi = r.integers(0, rows, tasks).astype(np.int32)
    si = r.integers(0, rows, (tasks, k * n)).astype(np.int32)
    si[3, :n] = qi[3]          # a task whose first class is the query itself (distance ~0 / an exact tie candidate)
    pred = torch.empty(tasks, k, device="cuda")
    am = torch.empty(tasks, dtype=torch.int32, device="cuda")
    L().call("vm_nshot_indexed", p(dev(emb)), rows, p(dev(qi, torch.int32)), p(dev(si, torch.int32)), tasks, k, n, E, DIST[dist], p(pred),
             p(am), stream())
    got, gam = pred.cpu().numpy(), am.cpu().numpy()
    for t in range(tasks):
        ref = O.n_shot_prediction(emb[qi[t]], emb[si[t]], n, k, dist)
        assert np.abs(got[t] - ref).max() < 2e-6 * max(1.0, np.abs(ref).max()), t
        order = np.sort(ref)
        if len(order) < 2 or order[1] - order[0] > 1e-6 * max(1.0, abs(order[0])):
            assert gam[t] == int(np.argmin(ref)), t
    # the same launch without the distance output
    am2 = torch.empty_like(am)
    L().call("vm_nshot_indexed", p(dev(emb)), rows, p(dev(qi, torch.int32)), p(dev(si, torch.int32)), tasks, k, n, E, DIST[dist], None,
             p(am2), stream())
    assert torch.equal(am, am2)


def test_nshot_indexed_nan_distance_comes_first_like_numpy():
    emb = np.ones((6, 64), np.float32)
    emb[2] = np.nan
    qi = np.array([0], np.int32)
    si = np.array([[1, 2, 3]], np.int32)
    am = torch.empty(1, dtype=torch.int32, device="cuda")
    L().call("vm_nshot_indexed", p(dev(emb)), 6, p(dev(qi, torch.int32)), p(dev(si, torch.int32)), 1, 3, 1, 64, 0, None, p(am), stream())
    assert am.item() == int(np.argmin(O.n_shot_prediction(emb[0], emb[si[0]], 1, 3, "euclidean"))) == 1


@pytest.mark.parametrize("dist", ["euclidean", "cosine", "dot_product"])
@pytest.mark.parametrize("M,N,E,row0", [(70, 333, 64, -1), (64, 64, 64, 0), (130, 1000, 128, 200), (5, 77, 100, 3), (256, 4100, 64, 1024),
                                        (33, 200, 256, -1)])
def test_pairdist_argmin_vs_numpy(dist, M, N, E, row0):
    r = np.random.default_rng(M + N)
    ref = r.normal(0, 1, (N, E)).astype(np.float32)
    q = ref[row0:row0 + M].copy() if row0 >= 0 else r.normal(0, 1, (M, E)).astype(np.float32)
    ws = torch.empty(L().query("vm_pairdist_workspace_bytes", M, N) // 4 + 16, device="cuda")
    d = torch.full((M, N), float("nan"), device="cuda")
    bv = torch.empty(M, device="cuda")
    bi = torch.empty(M, dtype=torch.int32, device="cuda")
    L().call("vm_pairdist_argmin", p(dev(q)), p(dev(ref)), M, N, E, DIST[dist], row0, p(d), p(bv), p(bi), p(ws), stream())
    want = _pairwise_ref(q, ref, dist)
    got = d.cpu().numpy().astype(np.float64)
    if dist == "euclidean" and row0 >= 0:   # sqrt near 0 (a row against itself): compare squares there
        self_mask = np.zeros((M, N), bool)
        self_mask[np.arange(M), row0 + np.arange(M)] = True
        assert np.abs(got[self_mask]).max() < 1e-3
        got[self_mask] = want[self_mask] = 0.0
    assert np.abs(got - want).max() < 3e-6 * max(1.0, np.abs(want).max())
    # the argmin is the first minimum of the kernel's own matrix, the row's own entry excluded
    own = d.clone()
    if row0 >= 0:
        own[torch.arange(M), row0 + torch.arange(M)] = float("inf")
    assert torch.equal(bi.long().cpu(), own.argmin(dim=1).cpu())
    assert torch.equal(bv.cpu(), own.min(dim=1).values.cpu())
    # argmin-only launch: same answer, nothing else written
    bv2, bi2 = torch.empty_like(bv), torch.empty_like(bi)
    L().call("vm_pairdist_argmin", p(dev(q)), p(dev(ref)), M, N, E, DIST[dist], row0, None, p(bv2), p(bi2), p(ws), stream())
    assert torch.equal(bi, bi2) and torch.equal(bv, bv2)


def test_pairdist_argmin_full_shard_planted_neighbours():
    """BASELINE.json config 5 at the size bench.py times (one rank's 13 002 x 104 014 x 64 shard of a train-clean-360-sized matrix;
    experiments/k_way_accuracy.py:52-69 is the loop it replaces): a size-independent property instead of a float64 matrix of 1.35e9
    entries -- every query has ONE planted reference 1e-2 away (everything else is ~11 away), outside the query rows and different
    for every query, so the answer is known: argmin = the plant, its distance = the float64 distance to it."""
    N, M, E, row0 = 104014, 13002, 64, 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    ref = torch.randn(N, E, device="cuda", generator=g)
    perm = torch.randperm(N - (row0 + M), device="cuda", generator=g)[:M] + row0 + M     # distinct targets behind the query rows
    lo = torch.randperm(row0, device="cuda", generator=g)[:M // 4]                         # ... and a quarter of them in front
    perm[:M // 4] = lo
    noise = torch.randn(M, E, device="cuda", generator=g) * (1e-2 / 8.0)
    ref[perm] = ref[row0:row0 + M] + noise
    q = ref[row0:row0 + M].contiguous()
    ws = torch.empty(L().query("vm_pairdist_workspace_bytes", M, N) // 4 + 16, device="cuda")
    bv = torch.empty(M, device="cuda")
    bi = torch.empty(M, dtype=torch.int32, device="cuda")
    L().call("vm_pairdist_argmin", p(q), p(ref), M, N, E, DIST["euclidean"], row0, None, p(bv), p(bi), p(ws), stream())
    torch.cuda.synchronize()
    assert torch.equal(bi.long(), perm)
    want = (q.double() - ref[perm].double()).pow(2).sum(1).sqrt()
    assert (bv.double() - want).abs().max().item() < 3e-6 * 1.0 + 1e-7
    # without the exclusion a query finds itself
    L().call("vm_pairdist_argmin", p(q), p(ref), M, N, E, DIST["euclidean"], -1, None, p(bv), p(bi), p(ws), stream())
    torch.cuda.synchronize()
    assert torch.equal(bi.long(), torch.arange(row0, row0 + M, device="cuda")) and bv.abs().max().item() == 0.0


def _model_and_data(dtype="f32", speakers=14, files=6):
    from voicemap_amd import models as VM, utils as VU
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    torch.manual_seed(4)
    ds = SyntheticSpeechDataset(num_speakers=speakers, files_per_speaker=files, seconds=1, stochastic=False, seed=5)
    enc = VM.get_baseline_convolutional_encoder(16, 32, dropout=0.0, dtype=dtype)
    net = VM.build_siamese_net(enc, (ds.fragment_length // 4, 1))
    net.compile(loss="binary_crossentropy", optimizer="adam")
    pre = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
    return net, ds, pre


def test_cached_evaluation_matches_the_oracle_on_the_cached_embeddings_and_the_task_by_task_path():
    from voicemap_amd import retrieval as R, utils as VU
    net, ds, pre = _model_and_data()
    cache = R.embed_corpus(net, ds, pre)
    assert cache.emb.shape == (len(ds), 32) and torch.isfinite(cache.emb).all()
    # a row of the cache = the file's first fragment embedded on its own (whitened alone), whatever batch it was embedded in
    eng = net._ensure_engine()
    inst = pre.instance_preprocessor
    for i in (0, 7, len(ds) - 1):
        lazy = inst(R._first_fragment(ds, i)[None, :, None])
        one = eng.embed(torch.as_tensor(np.ascontiguousarray(lazy.raw, dtype=np.float32)), preprocessed=False, downsampling=lazy.downsampling,
                        whitening=lazy.whitening, windows_per_tower=1)
        assert rel_err(cache.emb[i].cpu().numpy(), one[0].cpu().numpy()) < 1e-6
    emb = cache.emb.cpu().numpy()
    for dist, k, n in (("euclidean", 5, 1), ("cosine", 4, 3), ("dot_product", 6, 2)):
        np.random.seed(11)
        q, s = R.draw_tasks_reference(ds, 30, k, n)
        got, pred = R.evaluate_tasks(cache, q, s, k, n, dist, return_pred=True)
        want = sum(int(np.argmin(O.n_shot_prediction(emb[q[t]], emb[s[t]], n, k, dist)) == 0) for t in range(len(q)))
        assert got == want, (dist, got, want)
        # the drawn tasks obey the reference's invariants (tests/tests.py:31-68 of the reference)
        spk = ds._code
        assert (spk[s[:, :n]] == spk[q][:, None]).all() and not (s[:, :n] == q[:, None]).any()
        classes = spk[s].reshape(len(q), k, n)
        assert (classes == classes[:, :, :1]).all() and all(len(set(c[:, 0])) == k for c in classes)
        # same seed through the wrapper (which embeds the corpus itself when no cache is passed)
        np.random.seed(11)
        through = R.n_shot_task_evaluation_cached(net, ds, pre, 30, n, k, "siamese", dist, cache=cache)
        if n == 1:
            # a siamese net's 1-shot cells are ranked by its verification head (voicemap/utils.py:121-137), not by ``distance``:
            # sigmoid(w * ||q - s|| + b) of the uniform_euclidean head, restated in float64 on the cached rows
            w_, b_ = eng.get_params()["head.kernel"].reshape(-1)[0].astype(np.float64), eng.get_params()["head.bias"].reshape(-1)[0].astype(np.float64)
            d = np.linalg.norm(emb[q][:, None, :].astype(np.float64) - emb[s].astype(np.float64), axis=2)
            p_head = 1.0 / (1.0 + np.exp(-(w_ * d + b_)))
            n_head, pred_head = R.evaluate_tasks_head(eng, cache, q, s, k, return_pred=True)
            assert np.abs(pred_head - p_head).max() < 1e-6
            assert through == n_head == int((np.argmin(pred_head, axis=1) == 0).sum())
            if w_ > 0:   # a monotone function of the euclidean distance: the same ranking
                assert through == got
        else:
            assert through == got
    # against the reference-faithful task-by-task evaluation on the SAME tasks (same seed): the only difference is the whitening
    # scalar of the support windows (per task batch there, per window here), so the accuracies are close, not equal
    np.random.seed(3)
    faithful = VU.n_shot_task_evaluation(net, ds, pre, 30, 5, 5, network_type="siamese", distance="euclidean")
    np.random.seed(3)
    cached = R.n_shot_task_evaluation_cached(net, ds, pre, 30, 5, 5, "siamese", "euclidean", cache=cache)
    report("cached_eval", "acc_task_by_task_5way_5shot", faithful / 30.0)
    report("cached_eval", "acc_cached_5way_5shot", cached / 30.0)
    # (untrained net on synthetic speakers of very different loudness: 0.51 vs 0.77 measured -- batch-level whitening keeps the
    # loudness differences between a task's support windows, per-window whitening removes them; both must beat chance = 0.2)
    assert faithful > 0.3 * 30 and cached > 0.3 * 30


def test_device_task_sampler_draws_valid_tasks_with_the_reference_distribution():
    from voicemap_amd import retrieval as R
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    ds = SyntheticSpeechDataset(num_speakers=9, files_per_speaker=5, seconds=1, stochastic=False, seed=2)
    sm = R.DeviceTaskSampler(ds, "cuda", seed=1)
    k, n, tasks = 4, 3, 20000
    q, s = sm.draw(tasks, k, n)
    q, s = q.cpu().numpy(), s.cpu().numpy()
    spk = ds._code
    assert (spk[s[:, :n]] == spk[q][:, None]).all() and not (s[:, :n] == q[:, None]).any()
    classes = spk[s].reshape(tasks, k, n)
    assert (classes == classes[:, :, :1]).all()
    assert all(len(set(c)) == k for c in classes[:500, :, 0])
    files = s.reshape(tasks, k, n)
    assert all(len(set(f)) == n for f in files[:500].reshape(-1, n))
    # query files are drawn with probability ~ length (voicemap/librispeech.py:218)
    freq = np.bincount(q, minlength=len(ds)) / tasks
    want = ds._len / ds._len.sum()
    assert np.abs(freq - want).max() < 4 * np.sqrt(want.max() / tasks)
    # every other speaker is equally likely to be a distractor
    others = np.bincount(classes[:, 1:, 0].ravel(), minlength=9) / (tasks * (k - 1))
    assert np.abs(others - 1.0 / 9).max() < 0.01
    with pytest.raises(ValueError):
        sm.draw(4, 9, 1)      # k must be smaller than the number of speakers
    with pytest.raises(ValueError):
        sm.draw(4, 3, 5)      # a speaker has only 4 other files


def test_pairwise_retrieval_accuracy_and_matrix():
    from voicemap_amd import retrieval as R
    net, ds, pre = _model_and_data()
    cache = R.embed_corpus(net, ds, pre)
    emb = cache.emb.cpu().numpy()
    for dist in ("euclidean", "cosine", "dot_product"):
        out = R.pairwise_retrieval(cache, dist, return_matrix=True)
        ref = _pairwise_ref(emb, emb, dist)
        np.fill_diagonal(ref, np.inf)
        nn = ref.argmin(1)
        want = int((ds._code[nn] == ds._code).sum())
        got_nn = out["best_idx"].cpu().numpy()
        clear = np.sort(ref, 1)[:, 1] - np.sort(ref, 1)[:, 0] > 1e-5
        assert (got_nn[clear] == nn[clear]).all()
        assert abs(out["n_correct"] - want) <= int((~clear).sum()) and out["n_rows"] == len(ds)
        m = out["matrix"].cpu().numpy()
        off = ~np.eye(len(ds), dtype=bool)
        assert np.abs(m[off] - _pairwise_ref(emb, emb, dist)[off]).max() < 1e-4 * max(1.0, np.abs(ref[off]).max())
        # a row shard against the whole matrix (what one rank of a data-parallel run computes)
        part = R.pairwise_retrieval(cache, dist, rows=(10, 31))
        assert torch.equal(part["best_idx"], out["best_idx"][10:31])
