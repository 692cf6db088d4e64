"""Drop-in mirror of ``voicemap/utils.py``: same names, arguments and error behaviour.

    get_bottleneck(classifier, samples)                                                        utils.py:9
    preprocess_instances(downsampling, whitening=True)                                         utils.py:22
    BatchPreProcessor(mode, instance_preprocessor, target_preprocessor=identity)               utils.py:37
    contrastive_loss(y_true, y_pred)                                                           utils.py:77
    whiten(batch, rms=0.038021)                                                                utils.py:88
    n_shot_task_evaluation(model, dataset, preprocessor, num_tasks, n, k, network_type, distance)   utils.py:104
    NShotEvaluationCallback(num_tasks, n_shot, k_way, dataset, preprocessor, mode)             utils.py:219

What is different underneath: ``preprocess_instances`` returns a *lazy* batch (``LazyWindows``) that remembers the raw
16 kHz windows; the HIP models decimate + whiten it on the GPU (vm_decimate_whiten) instead of in three numpy passes on
the host, and the n-shot evaluation runs all tasks of a run through batched launches instead of ``num_tasks``
sequential ``predict`` calls.  A ``LazyWindows`` converts to the reference's float64 array on demand (``np.asarray``).
"""
from __future__ import annotations

import numpy as np

from .keras_like import Callback


# ---------------------------------------------------------------------------------------------------------
class LazyWindows:
    """Result of ``preprocess_instances(d)(instances)``: (n, T, 1) raw windows + the preprocessing still to apply.
    ``np.asarray(lazy)`` gives exactly what the reference's preprocessor returns (float64, decimated, whitened)."""

    def __init__(self, raw, downsampling: int, whitening: bool):
        if type(raw).__name__ != "DeviceWindows":  # shards.DeviceWindows: offsets into a device-resident buffer, kept as is
            raw = np.asarray(raw)
        if raw.ndim != 3:
            raise ValueError("Input must be a 3D array of shape (n_segments, n_timesteps, 1).")
        self.raw, self.downsampling, self.whitening = raw, int(downsampling), bool(whitening)

    @property
    def shape(self):
        n, t, c = self.raw.shape
        return (n, (t + self.downsampling - 1) // self.downsampling, c)

    def __len__(self):
        return self.raw.shape[0]

    def __array__(self, dtype=None, copy=None):
        x = np.asarray(self.raw)[:, ::self.downsampling, :]
        if self.whitening:
            x = whiten(x)
        return np.asarray(x, dtype=dtype) if dtype is not None else np.asarray(x)


def whiten(batch, rms=0.038021):
    """voicemap/utils.py:88-101: subtract each window's mean, then multiply the WHOLE batch by one scalar
    rms / sqrt(mean(batch**2)) (mean of squares of the un-centred batch, utils.py:98).  Host numpy, float64 in / out like
    the reference; the GPU path for training is vm_decimate_whiten."""
    batch = np.asarray(batch)
    if len(batch.shape) != 3:
        raise ValueError("Input must be a 3D array of shape (n_segments, n_timesteps, 1).")
    centred = batch - batch.mean(axis=1, keepdims=True)
    return centred * (rms / np.sqrt(np.power(batch, 2).mean()))


def preprocess_instances(downsampling, whitening=True):
    """voicemap/utils.py:22-34: the canonical preprocessing -- take every ``downsampling``-th sample, then whiten."""
    def preprocess_instances_(instances):
        return LazyWindows(instances, downsampling, whitening)

    return preprocess_instances_


class BatchPreProcessor(object):
    """voicemap/utils.py:37-74: applies the instance preprocessor to classifier batches ``(inputs, outputs)`` and to
    siamese batches ``([input_1, input_2], outputs)`` -- each tower separately (utils.py:59-60) -- and the target
    preprocessor to the labels."""

    def __init__(self, mode, instance_preprocessor, target_preprocessor=lambda x: x):
        assert mode in ("siamese", "classifier")
        self.mode = mode
        self.instance_preprocessor = instance_preprocessor
        self.target_preprocessor = target_preprocessor

    def __call__(self, batch):
        if self.mode == "siamese":
            ([input_1, input_2], labels) = batch
            return [self.instance_preprocessor(input_1), self.instance_preprocessor(input_2)], self.target_preprocessor(labels)
        elif self.mode == "classifier":
            instances, labels = batch
            return self.instance_preprocessor(instances), self.target_preprocessor(labels)
        raise ValueError


def contrastive_loss(y_true, y_pred):
    """voicemap/utils.py:77-85 (Hadsell et al. '06, margin 1): mean((1-y) p^2 + y max(1-p, 0)^2).  Pass it to
    ``model.compile(loss=contrastive_loss)``: the siamese model recognises it and evaluates loss + gradient inside the
    fused head kernel (vm_siamese_head_loss).  Called directly it evaluates the same expression with numpy."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_pred = np.asarray(y_pred, dtype=np.float64)
    margin = 1
    return np.mean((1 - y_true) * np.square(y_pred) + y_true * np.square(np.maximum(margin - y_pred, 0)))


def get_bottleneck(classifier, samples):
    """voicemap/utils.py:9-19: activations of the layer before the classification layer (= the embedding) in
    inference mode."""
    enc = classifier.clone()
    enc.set_weights(classifier.get_weights())
    enc.pop()
    return enc.predict(samples)


# ---------------------------------------------------------------------------------------------------------
_DIST = {"euclidean": 0, "cosine": 1, "dot_product": 2}


def _as_raw(pre, windows):
    """windows: (m, T, 1) raw -> what to hand to the engine."""
    out = pre(windows)
    return out


def n_shot_task_evaluation(model, dataset, preprocessor, num_tasks, n, k, network_type="siamese", distance="euclidean"):
    """voicemap/utils.py:104-216 with the same task semantics and return value (``n_correct``):

    * n == 1 and a siamese network (:121-137): the query repeated k times against the k support windows through the
      verification head; each side is preprocessed (whitened) as its own batch of k (:131); correct iff
      ``argmin(pred[:, 0]) == 0`` (:135).
    * n > 1 or a classifier (:138-212): embed query and support with the encoder (``model.layers[2]`` for the siamese
      net :141, the classifier minus its last layer :143-145), whitening the query alone and the support set as one
      batch (:153-154), class prototypes + distance (:159-206) on the GPU (vm_nshot_distances), correct iff argmin == 0.

    Tasks are sampled one by one with ``dataset.build_n_shot_task(k, n)`` exactly like the reference, but embedded in
    batched launches (the per-task whitening batches are kept as towers of the preprocessing kernel).

    Under torchrun (BASELINE.json config 5) the ``num_tasks`` tasks are sharded over the ranks: every rank samples and
    evaluates its own ``parallel.shard_range`` share (tasks are independent draws) and the counts are summed with one
    all-reduce, so every rank returns the global ``n_correct``."""
    import torch
    from . import parallel
    rank, world = parallel.rank_world()
    if world > 1:
        lo, hi = parallel.shard_range(num_tasks, rank, world)
        local = _n_shot_local(model, dataset, preprocessor, hi - lo, n, k, network_type, distance)
        return int(round(parallel.sum_over_ranks(float(local))))
    return _n_shot_local(model, dataset, preprocessor, num_tasks, n, k, network_type, distance)


def _n_shot_local(model, dataset, preprocessor, num_tasks, n, k, network_type="siamese", distance="euclidean"):
    """This rank's share of ``n_shot_task_evaluation``."""
    import torch
    if n < 1:
        raise ValueError("n must be >= 1")
    if network_type not in ("siamese", "classifier"):
        raise ValueError("mode must be one of (siamese, classifier)")
    if not (n == 1 and network_type == "siamese") and distance not in _DIST:
        raise ValueError("Distance must be in (euclidean, cosine, dot_product)")
    inst = preprocessor.instance_preprocessor if hasattr(preprocessor, "instance_preprocessor") else None

    if num_tasks == 0:
        return 0
    if getattr(dataset, "device_audio", None) is not None and hasattr(dataset, "build_n_shot_tasks_device"):
        # device-resident corpus (shards.py): tasks are start offsets, windows are cropped by the preprocessing kernel
        return _n_shot_device(model, dataset, preprocessor, num_tasks, n, k, network_type, distance)
    queries, supports = [], []
    for _ in range(num_tasks):
        query_sample, support_set_samples = dataset.build_n_shot_task(k, n)
        queries.append(np.asarray(query_sample[0]))
        supports.append(np.asarray(support_set_samples[0]))

    n_correct = 0
    chunk = max(1, 256 // max(k * n, 1))  # tasks per launch
    if n == 1 and network_type == "siamese":
        eng = model._ensure_engine()
        for t0 in range(0, num_tasks, chunk):
            qs, ss = queries[t0:t0 + chunk], supports[t0:t0 + chunk]
            nt = len(qs)
            in1 = np.concatenate([np.stack([q] * k) for q in qs])[:, :, np.newaxis]   # (nt*k, T, 1)
            in2 = np.concatenate(ss)[:, :, np.newaxis]
            lazy1, lazy2 = _lazy_pair(preprocessor, in1, in2)
            pred = _siamese_predict_towers(eng, lazy1, lazy2, tower=k).reshape(nt, k)
            n_correct += int((pred.argmin(axis=1) == 0).sum())
        return n_correct

    # embedding route
    if network_type == "siamese":
        encoder = model.layers[2]
        encoder.engine = model._ensure_engine()
    else:
        encoder = model.clone()
        encoder.set_weights(model.get_weights())
        encoder.pop()
    eng = encoder._ensure_engine()
    for t0 in range(0, num_tasks, chunk):
        qs, ss = queries[t0:t0 + chunk], supports[t0:t0 + chunk]
        nt = len(qs)
        qraw = np.stack(qs)[:, :, np.newaxis]            # (nt, T, 1): each query is whitened alone (tower = 1)
        sraw = np.concatenate(ss)[:, :, np.newaxis]      # (nt*k*n, T, 1): each task's support set is one whitening batch
        ql, sl = inst(qraw), inst(sraw)
        qe = _embed_towers(eng, ql, tower=1)
        se = _embed_towers(eng, sl, tower=k * n)
        pred = torch.empty(nt, k, dtype=torch.float32, device=eng.device)
        am = torch.empty(nt, dtype=torch.int32, device=eng.device)
        eng._call("vm_nshot_distances", qe.data_ptr(), se.data_ptr(), nt, k, n, eng.E, _DIST[distance], pred.data_ptr(),
                  am.data_ptr(), eng.stream())
        n_correct += int((am == 0).sum().item())
    return n_correct


def _n_shot_device(model, dataset, preprocessor, num_tasks, n, k, network_type, distance):
    """n_shot_task_evaluation over ``ShardedSpeechDataset.build_n_shot_tasks_device``: identical task semantics and
    whitening batches (query alone / repeated k times, support set of a task as one batch), windows addressed by offset."""
    import torch
    q, s, _, _ = dataset.build_n_shot_tasks_device(num_tasks, k, n)
    lazy = preprocessor.instance_preprocessor(q) if hasattr(preprocessor, "instance_preprocessor") else None
    if not isinstance(lazy, LazyWindows):
        raise ValueError("the device n-shot path needs a BatchPreProcessor over preprocess_instances")
    ds, wh, T = lazy.downsampling, lazy.whitening, q.length
    n_correct = 0
    chunk = max(1, 256 // max(k * n, 1))  # tasks per launch
    siamese_1shot = n == 1 and network_type == "siamese"
    if siamese_1shot:
        eng = model._ensure_engine()
    elif network_type == "siamese":
        encoder = model.layers[2]
        encoder.engine = model._ensure_engine()
        eng = encoder._ensure_engine()
    else:
        encoder = model.clone()
        encoder.set_weights(model.get_weights())
        encoder.pop()
        eng = encoder._ensure_engine()
    for t0 in range(0, num_tasks, chunk):
        nt = min(chunk, num_tasks - t0)
        qo = q.offsets[t0:t0 + nt]
        so = s.offsets[t0 * k * n:(t0 + nt) * k * n]
        if siamese_1shot:
            off = torch.cat([qo.repeat_interleave(k), so])  # input_1 = the query k times, input_2 = the k support windows
            eng.embed_from_offsets(q.audio, off, T, ds, wh, windows_per_tower=k)
            pl = eng.plan(2 * nt * k, eng.last_infer_l0, False)
            pred = eng.siamese_head(pl, None).reshape(nt, k)
            n_correct += int((pred.argmin(dim=1) == 0).sum().item())
            continue
        qe = eng.embed_from_offsets(q.audio, qo, T, ds, wh, windows_per_tower=1).clone()
        se = eng.embed_from_offsets(q.audio, so, T, ds, wh, windows_per_tower=k * n).clone()
        pred = torch.empty(nt, k, dtype=torch.float32, device=eng.device)
        am = torch.empty(nt, dtype=torch.int32, device=eng.device)
        eng._call("vm_nshot_distances", qe.data_ptr(), se.data_ptr(), nt, k, n, eng.E, _DIST[distance], pred.data_ptr(),
                  am.data_ptr(), eng.stream())
        n_correct += int((am == 0).sum().item())
    return n_correct


def _lazy_pair(preprocessor, in1, in2):
    ([a, b], _) = preprocessor(([in1, in2], []))
    return a, b


def _embed_towers(eng, lazy, tower: int):
    """Embed a LazyWindows batch in inference mode with whitening statistics taken per group of ``tower`` windows."""
    import torch
    if isinstance(lazy, LazyWindows):
        return eng.embed(torch.as_tensor(np.ascontiguousarray(lazy.raw, dtype=np.float32)), preprocessed=False,
                         downsampling=lazy.downsampling, whitening=lazy.whitening, windows_per_tower=tower).clone()
    return eng.embed(np.asarray(lazy, dtype=np.float32)).clone()


def _siamese_predict_towers(eng, lazy1, lazy2, tower: int):
    import torch
    if isinstance(lazy1, LazyWindows) and isinstance(lazy2, LazyWindows):
        x = np.concatenate([lazy1.raw, lazy2.raw]).astype(np.float32)
        pairs = lazy1.raw.shape[0]
        eng.embed(torch.as_tensor(x), preprocessed=False, downsampling=lazy1.downsampling, whitening=lazy1.whitening,
                  windows_per_tower=tower)
        pl = eng.plan(2 * pairs, eng.last_infer_l0, False)
        return eng.siamese_head(pl, None).cpu().numpy().copy()
    return eng.siamese_predict(np.asarray(lazy1, dtype=np.float32), np.asarray(lazy2, dtype=np.float32)).cpu().numpy()[:, 0]


class NShotEvaluationCallback(Callback):
    """voicemap/utils.py:219-252: after every epoch evaluate ``num_tasks`` k-way n-shot tasks and store the accuracy in
    ``logs['val_{n}-shot_acc']`` (later callbacks in the list monitor that key)."""

    def __init__(self, num_tasks, n_shot, k_way, dataset, preprocessor=lambda x: x, mode="siamese"):
        super(NShotEvaluationCallback, self).__init__()
        self.num_tasks = num_tasks
        self.n_shot = n_shot
        self.k_way = k_way
        self.dataset = dataset
        self.preprocessor = preprocessor
        assert mode in ("siamese", "classifier")
        self.mode = mode

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        n_correct = n_shot_task_evaluation(self.model, self.dataset, self.preprocessor, self.num_tasks, self.n_shot,
                                           self.k_way, network_type=self.mode)
        n_shot_acc = n_correct * 1. / self.num_tasks
        logs["val_{}-shot_acc".format(self.n_shot)] = n_shot_acc
        from . import parallel
        if parallel.rank_world()[0] == 0:   # every rank holds the same (summed) figure; one line like the reference's
            print("val_{}-shot_acc: {:.4f}".format(self.n_shot, n_shot_acc))
