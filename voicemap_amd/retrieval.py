"""Evaluation over a cached embedding matrix -- BASELINE.json config 5 as it is worded ("batched embedding forward + pairwise-distance
matrix over train-clean-360, sharded across 8 x MI355X") and SURVEY.md 8(f).2.

The reference evaluates task by task: ``n_shot_task_evaluation`` (voicemap/utils.py:104-216) embeds the k*n + 1 windows of every
task with two ``predict`` calls, and ``experiments/k_way_accuracy.py:52-69`` does that 38 x 1000 times.  Its evaluation datasets are
built with ``stochastic=False`` (experiments/train_siamese.py:44, k_way_accuracy.py:38): a file's window is its first
``fragment_length`` samples, always -- so the corpus has ONE embedding per file.  Here it is embedded once, in batches of a few hundred
windows straight from the device-resident corpus, into an (N, E) fp32 matrix in HBM (104 K x 64 x 4 B = 27 MB for train-clean-360),
and

* a k-way n-shot task is k*n + 1 row indices: ``vm_nshot_indexed`` evaluates any number of tasks in one launch (float64 arithmetic
  of voicemap/utils.py:159-206, all three distances);
* the pairwise-distance matrix / nearest-neighbour retrieval is ``vm_pairdist_argmin`` on this rank's rows against all rows.

Data parallel (one process per GPU): ranks embed disjoint row ranges, all-gather the matrix (the only bulk exchange: N x E fp32), then
take disjoint task ranges / query-row ranges and sum one integer.

ONE deviation from the reference, which is why this mode is opt-in and the task-by-task path stays the default: ``whiten`` scales a
batch by ONE scalar (voicemap/utils.py:94-99); the reference whitens a task's query alone but its support set as one batch
of k*n windows (utils.py:153-154), so a support window's scale depends on the task it is drawn into.  A cached embedding cannot: here
every window is whitened alone (what the reference does to queries).  With real speech (windows of similar RMS) the two differ by
the spread of the per-window RMS; ``tests/test_gpu_retrieval.py`` measures the accuracy of both on the same tasks.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import parallel

_DIST = {"euclidean": 0, "cosine": 1, "dot_product": 2}


def _encoder_engine(model, network_type: str):
    """The engine that embeds: ``model.layers[2]`` of a siamese net (voicemap/utils.py:141), a classifier minus its last layer
    (:143-145), or an encoder as it is."""
    if network_type == "siamese" and hasattr(model, "layers") and len(getattr(model, "layers", [])) > 2 and hasattr(model, "_ensure_engine") \
            and hasattr(model.layers[2], "_ensure_engine") and model.layers[2] is not model:
        enc = model.layers[2]
        enc.engine = model._ensure_engine()
        return enc._ensure_engine()
    if network_type == "classifier":
        enc = model.clone()
        enc.set_weights(model.get_weights())
        enc.pop()
        return enc._ensure_engine()
    return model._ensure_engine()


class EmbeddingCache:
    """(N, E) fp32 embeddings of a dataset's files on the device + the speaker code of every row (``dataset._code`` order)."""

    def __init__(self, emb: torch.Tensor, speaker: np.ndarray):
        self.emb = emb
        self.speaker = np.asarray(speaker)
        self.speaker_dev = torch.as_tensor(self.speaker.astype(np.int32)).to(emb.device)

    @property
    def n(self) -> int:
        return int(self.emb.shape[0])

    @property
    def E(self) -> int:
        return int(self.emb.shape[1])


def embed_corpus(model, dataset, preprocessor, network_type: str = "siamese", batch: int = 256) -> EmbeddingCache:
    """Embed the first-fragment window of every file of ``dataset`` (inference mode, each window whitened alone).  With a
    device-resident corpus (``ShardedSpeechDataset.to_device``) windows are start offsets and the crop happens in the preprocessing
    kernel; otherwise windows are loaded on the host like ``dataset[i]`` with ``stochastic=False``.  Under torchrun every rank embeds
    ``parallel.shard_range`` of the rows and the matrix is all-gathered."""
    eng = _encoder_engine(model, network_type)
    inst = preprocessor.instance_preprocessor if hasattr(preprocessor, "instance_preprocessor") else preprocessor
    n_files = len(dataset)
    rank, world = parallel.rank_world()
    lo, hi = parallel.shard_range(n_files, rank, world)
    T = dataset.fragment_length
    local = torch.empty(hi - lo, eng.E, dtype=torch.float32, device=eng.device)
    dev_audio = getattr(dataset, "device_audio", None)
    probe = inst(np.zeros((1, 8, 1)))
    ds, wh = getattr(probe, "downsampling", 1), getattr(probe, "whitening", False)
    lazy_ok = hasattr(probe, "raw")
    for b0 in range(lo, hi, batch):
        idx = np.arange(b0, min(b0 + batch, hi))
        if dev_audio is not None and lazy_ok:
            if np.any(dataset.file_length[idx] < T):
                raise ValueError("the device path cannot pad: a file is shorter than the fragment length")
            offs = torch.as_tensor(dataset.global_offset[idx])
            e = eng.embed_from_offsets(dev_audio, offs, T, ds, wh, windows_per_tower=1)
        else:
            win = np.stack([_first_fragment(dataset, int(i)) for i in idx])[:, :, np.newaxis]
            lazy = inst(win)
            if hasattr(lazy, "raw"):
                e = eng.embed(torch.as_tensor(np.ascontiguousarray(lazy.raw, dtype=np.float32)), preprocessed=False,
                              downsampling=lazy.downsampling, whitening=lazy.whitening, windows_per_tower=1)
            else:   # a preprocessor that already returned numbers: whiten-alone semantics are the caller's business
                e = eng.embed(np.asarray(lazy, dtype=np.float32))
        local[b0 - lo:b0 - lo + len(idx)].copy_(e)
    emb = all_gather_rows(local, n_files)
    return EmbeddingCache(emb, dataset._code)


def _first_fragment(dataset, index: int) -> np.ndarray:
    """``dataset[index][0]`` with ``stochastic=False`` (voicemap/librispeech.py:103-137: start 0, optional zero padding) whatever the
    dataset's own setting, and without touching ``np.random``."""
    x = np.asarray(dataset._load(index))[:dataset.fragment_length]
    if len(x) < dataset.fragment_length:
        if not dataset.pad:
            raise ValueError("file %d is shorter than the fragment length and the dataset does not pad" % index)
        x = np.pad(x, (0, dataset.fragment_length - len(x)), "constant")   # stochastic=False pads at the end (:123-127)
    return x


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate the ranks' row blocks (``parallel.shard_range`` sizes, which differ by at most one row) in rank order."""
    rank, world = parallel.rank_world()
    if world == 1:
        return local
    import torch.distributed as dist
    sizes = [parallel.shard_range(n_total, r, world) for r in range(world)]
    most = max(h - l for l, h in sizes)
    pad = torch.zeros(most, local.shape[1], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]].copy_(local)
    comm = pad if dist.get_backend() != "gloo" else pad.cpu()
    parts = [torch.empty_like(comm) for _ in range(world)]
    dist.all_gather(parts, comm)
    return torch.cat([p[:h - l] for p, (l, h) in zip(parts, sizes)]).to(local.device)


# ---- tasks as indices -----------------------------------------------------------------------------------------------------------
def draw_tasks_reference(dataset, num_tasks: int, k: int, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """(query_idx (num_tasks,), support_idx (num_tasks, k*n)): the tasks ``dataset.build_n_shot_task(k, n)`` would build, drawn with
    the very same ``np.random`` calls in the same order (query file ~ length, n other files of its speaker, k-1 other speakers
    uniformly, n files each; voicemap/librispeech.py:204-240) -- including, for a ``stochastic`` dataset, the fragment-start draws
    that ``__getitem__`` makes and this mode ignores, so that a seed selects the same files either way."""
    if k >= dataset.unique_speakers:
        raise ValueError('k must be smaller than the number of unique speakers in this dataset!')
    if k <= 1:
        raise ValueError('k must be greater than or equal to one!')
    q = np.empty(num_tasks, dtype=np.int32)
    s = np.empty((num_tasks, k * n), dtype=np.int32)
    T = dataset.fragment_length
    for t in range(num_tasks):
        qi = int(dataset._weighted(1, dataset._len)[0])
        if dataset.stochastic:
            np.random.randint(0, max(int(dataset._len[qi]) - T, 1))
        si = dataset._n_shot_support(qi, k, n)
        if dataset.stochastic:
            np.random.randint(0, np.maximum(dataset._len[si].astype(np.int64) - T, 1))
        q[t] = qi
        s[t] = si
    return q, s


class DeviceTaskSampler:
    """The same task DISTRIBUTION drawn on the GPU, thousands of tasks per call (not the reference's random stream): query file with
    probability ~ length; n other files of its speaker and n files of each of k-1 other speakers, weighted by length without
    replacement (Gumbel top-n: keys log w + Gumbel noise, the n largest -- the Plackett-Luce law of successive weighted draws);
    the k-1 other speakers uniformly without replacement (uniform keys, top k-1).  Host time per task drops from ~80 us (the
    reference-order draws) to nothing measurable, which is what lets the cached evaluation run at the kernel's rate."""

    def __init__(self, dataset, device, seed: int = 0):
        self.device = torch.device(device)
        code = np.asarray(dataset._code)
        cnt, start, files = np.asarray(dataset._cnt), np.asarray(dataset._start), np.asarray(dataset._files)
        S, fmax = len(cnt), int(cnt.max())
        table = np.full((S, fmax), -1, dtype=np.int64)
        logw = np.full((S, fmax), -np.inf, dtype=np.float32)
        for c in range(S):
            f = files[start[c]:start[c] + cnt[c]]
            table[c, :cnt[c]] = f
            logw[c, :cnt[c]] = np.log(np.maximum(dataset._len[f], 1e-30))
        self.table = torch.as_tensor(table).to(self.device)
        self.logw = torch.as_tensor(logw).to(self.device)
        self.cnt = torch.as_tensor(cnt.astype(np.int64)).to(self.device)
        self.code = torch.as_tensor(code.astype(np.int64)).to(self.device)
        self.file_w = torch.as_tensor(np.asarray(dataset._len, dtype=np.float32)).to(self.device)
        self.S = S
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def _gumbel(self, shape):
        u = torch.rand(shape, device=self.device, generator=self.gen).clamp_(1e-20, 1.0)
        return -torch.log(-torch.log(u))

    def draw(self, num_tasks: int, k: int, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        if k >= self.S:
            raise ValueError('k must be smaller than the number of unique speakers in this dataset!')
        if k <= 1:
            raise ValueError('k must be greater than or equal to one!')
        q = torch.multinomial(self.file_w, num_tasks, replacement=True, generator=self.gen)
        qs = self.code[q]
        if bool((self.cnt[qs] - 1 < n).any()):
            raise ValueError("Fewer non-zero entries in p than size")   # np.random.choice's refusal (a speaker with < n other files)
        # k - 1 other speakers, uniformly without replacement; only speakers with >= n files can fill a class
        keys = torch.rand(num_tasks, self.S, device=self.device, generator=self.gen)
        keys.scatter_(1, qs[:, None], -1.0)
        keys[:, self.cnt < n] = -1.0
        top = keys.topk(k - 1, dim=1)
        if bool((top.values < 0).any()):
            raise ValueError("Fewer non-zero entries in p than size")
        spk = torch.cat([qs[:, None], top.indices], 1)                  # (tasks, k): class 1 = the query's speaker
        lw = self.logw[spk]                                             # (tasks, k, fmax)
        own = self.table[qs] == q[:, None]                              # the query file itself never supports its own class
        lw[:, 0][own] = float("-inf")
        pick = (lw + self._gumbel(lw.shape)).topk(n, dim=2).indices     # (tasks, k, n)
        sup = torch.gather(self.table[spk], 2, pick).reshape(num_tasks, k * n)
        return q.to(torch.int32), sup.to(torch.int32)


def evaluate_tasks(cache: EmbeddingCache, query_idx, support_idx, k: int, n: int, distance: str = "euclidean",
                   return_pred: bool = False):
    """n_correct (and optionally the (tasks, k) distances) of tasks given as row indices -- correct iff argmin == 0, the support set
    being laid out with the query's speaker first (voicemap/utils.py:208-210)."""
    from . import _lib
    if distance not in _DIST:
        raise ValueError("Distance must be in (euclidean, cosine, dot_product)")
    dev = cache.emb.device
    q = torch.as_tensor(query_idx).to(dev, torch.int32).contiguous()
    s = torch.as_tensor(support_idx).to(dev, torch.int32).contiguous().reshape(-1)
    tasks = int(q.numel())
    if tasks == 0:
        return (0, None) if return_pred else 0
    assert s.numel() == tasks * k * n
    if int(torch.minimum(q.min(), s.min())) < 0 or int(torch.maximum(q.max(), s.max())) >= cache.n:
        raise IndexError("task indices outside the cached matrix (0 .. %d)" % (cache.n - 1))
    am = torch.empty(tasks, dtype=torch.int32, device=dev)
    pred = torch.empty(tasks, k, dtype=torch.float32, device=dev) if return_pred else None
    _lib.lib().call("vm_nshot_indexed", cache.emb.data_ptr(), cache.n, q.data_ptr(), s.data_ptr(), tasks, k, n, cache.E, _DIST[distance],
                    None if pred is None else pred.data_ptr(), am.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    n_correct = int((am == 0).sum().item())
    return (n_correct, pred) if return_pred else n_correct


def _siamese_head_engine(model):
    """The engine of a siamese net that owns the verification head's weights (None for a bare encoder)."""
    if hasattr(model, "_ensure_engine") and hasattr(model, "layers") and len(getattr(model, "layers", [])) > 2:
        eng = model._ensure_engine()
        if "head.kernel" in getattr(eng, "offsets", {}) and getattr(eng, "head", None) in ("uniform_euclidean", "weighted_l1"):
            return eng
    return None


def evaluate_tasks_head(eng, cache: EmbeddingCache, query_idx, support_idx, k: int, return_pred: bool = False):
    """1-shot tasks ranked the way voicemap/utils.py:121-137 ranks them for a siamese network: the model's own verification head on
    the k (query, support) pairs of every task -- ``vm_siamese_head_loss`` in its predict-only form on rows of the cached matrix --
    and correct iff the smallest output is the first pair's (:137).  n_correct (and optionally the (tasks, k) outputs)."""
    from .engine import HEADS, _p
    dev = cache.emb.device
    q = torch.as_tensor(query_idx).to(dev, torch.int64).reshape(-1)
    s = torch.as_tensor(support_idx).to(dev, torch.int64).reshape(-1)
    tasks = int(q.numel())
    if tasks == 0:
        return (0, None) if return_pred else 0
    assert s.numel() == tasks * k
    if int(torch.minimum(q.min(), s.min())) < 0 or int(torch.maximum(q.max(), s.max())) >= cache.n:
        raise IndexError("task indices outside the cached matrix (0 .. %d)" % (cache.n - 1))
    pairs = tasks * k
    both = torch.empty(2 * pairs, cache.E, dtype=torch.float32, device=dev)   # rows [0, pairs): the query k times per task; then the supports
    torch.index_select(cache.emb, 0, q.repeat_interleave(k), out=both[:pairs])
    torch.index_select(cache.emb, 0, s, out=both[pairs:])
    pred = torch.empty(pairs, dtype=torch.float32, device=dev)
    eng.lib.call("vm_siamese_head_loss", _p(both), _p(eng.view("head.kernel")), _p(eng.view("head.bias")), None, pairs, cache.E,
                 HEADS[eng.head], 0, 1.0, _p(pred), None, None, None, None, None, torch.cuda.current_stream(dev).cuda_stream)
    p = pred.reshape(tasks, k).cpu().numpy()
    n_correct = int((np.argmin(p, axis=1) == 0).sum())
    return (n_correct, p) if return_pred else n_correct


def n_shot_task_evaluation_cached(model, dataset, preprocessor, num_tasks, n, k, network_type="siamese", distance="euclidean",
                                  cache: Optional[EmbeddingCache] = None, sampler="reference"):
    """``n_shot_task_evaluation`` (voicemap/utils.py:104-216: same arguments, same return value ``n_correct``) on a cached embedding
    matrix: the corpus is embedded once (or ``cache`` re-used: experiments/k_way_accuracy.py sweeps 38 (k, n) cells over one model),
    tasks are row indices, all of them go through one launch.  ``sampler``: "reference" draws them with the reference's
    ``np.random`` sequence (``draw_tasks_reference``), a ``DeviceTaskSampler`` draws them on the GPU.  For a siamese net with n = 1
    the reference ranks the k candidates by the verification HEAD's output, whatever ``distance`` says (:121-137): so does this
    (``evaluate_tasks_head``: the head kernel on the cached embedding pairs); every other cell takes the embedding route
    (prototype distances, :138-212).  Under torchrun tasks are sharded over ranks and the counts summed."""
    if n < 1:
        raise ValueError("n must be >= 1")
    if network_type not in ("siamese", "classifier"):
        raise ValueError("mode must be one of (siamese, classifier)")
    if distance not in _DIST:
        raise ValueError("Distance must be in (euclidean, cosine, dot_product)")
    if cache is None:
        cache = embed_corpus(model, dataset, preprocessor, network_type)
    rank, world = parallel.rank_world()
    lo, hi = parallel.shard_range(num_tasks, rank, world)
    if hi > lo:
        if isinstance(sampler, DeviceTaskSampler):
            q, s = sampler.draw(hi - lo, k, n)
        else:
            q, s = draw_tasks_reference(dataset, hi - lo, k, n)
        head_eng = _siamese_head_engine(model) if (network_type == "siamese" and n == 1) else None
        if head_eng is not None:
            local = evaluate_tasks_head(head_eng, cache, q, s, k)
        else:
            local = evaluate_tasks(cache, q, s, k, n, distance)
    else:
        local = 0
    return int(round(parallel.sum_over_ranks(float(local)))) if world > 1 else local


def pairwise_retrieval(cache: EmbeddingCache, distance: str = "euclidean", return_matrix: bool = False, rows: Optional[Sequence[int]] = None):
    """The pairwise-distance matrix of the corpus against itself, this rank's rows (``parallel.shard_range`` of N, or ``rows`` =
    (lo, hi)): nearest OTHER utterance per query row with ``vm_pairdist_argmin`` and the retrieval form of verification accuracy --
    the share of utterances whose nearest neighbour is the same speaker -- summed over ranks.  Returns a dict with ``n_correct``,
    ``n_rows`` (global), ``accuracy``, this rank's ``best_idx`` / ``best_val`` and, on request, its (rows, N) block of the matrix."""
    from . import _lib
    if distance not in _DIST:
        raise ValueError("Distance must be in (euclidean, cosine, dot_product)")
    rank, world = parallel.rank_world()
    lo, hi = rows if rows is not None else parallel.shard_range(cache.n, rank, world)
    dev = cache.emb.device
    M, N = hi - lo, cache.n
    out = {"rows": (lo, hi)}
    if M > 0:
        lib = _lib.lib()
        ws = torch.empty(lib.query("vm_pairdist_workspace_bytes", M, N) // 4 + 16, dtype=torch.float32, device=dev)
        bv = torch.empty(M, dtype=torch.float32, device=dev)
        bi = torch.empty(M, dtype=torch.int32, device=dev)
        dist = torch.empty(M, N, dtype=torch.float32, device=dev) if return_matrix else None
        q = cache.emb[lo:hi]
        lib.call("vm_pairdist_argmin", q.data_ptr(), cache.emb.data_ptr(), M, N, cache.E, _DIST[distance], lo,
                 None if dist is None else dist.data_ptr(), bv.data_ptr(), bi.data_ptr(), ws.data_ptr(),
                 torch.cuda.current_stream(dev).cuda_stream)
        same = cache.speaker_dev[bi.clamp_min(0).long()] == cache.speaker_dev[lo:hi]
        local = int((same & (bi >= 0)).sum().item())
        out.update(best_idx=bi, best_val=bv, matrix=dist)
    else:
        local = 0
    total = int(round(parallel.sum_over_ranks(float(local)))) if world > 1 and rows is None else local
    n_rows = cache.n if rows is None else M
    out.update(n_correct=total, n_rows=n_rows, accuracy=total / max(n_rows, 1))
    return out
