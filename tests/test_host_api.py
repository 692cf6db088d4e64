"""CPU tests of the host-side mirror of the reference API (voicemap_amd.{models,utils,librispeech,keras_like}): the
reference's own dataset tests (tests/tests.py:16-68) restated on a synthetic speaker table, its whitening test
(:71-90), and the signatures / error behaviour of the build functions.  No GPU needed."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import voicemap_oracle as O
from voicemap_amd import keras_like as K
from voicemap_amd import models, utils
from voicemap_amd.librispeech import LibriSpeechDataset, SyntheticSpeechDataset


@pytest.fixture(scope="module")
def dataset():
    np.random.seed(0)
    return SyntheticSpeechDataset(num_speakers=40, files_per_speaker=6, seconds=3)  # dev-clean has 40 speakers


# ---- tests/tests.py:16-29 -------------------------------------------------------------------------------
def test_verification_pairs(dataset):
    alike = dataset.get_alike_pairs(16)
    assert len(alike) == 16
    assert all(dataset[i][1] == dataset[j][1] for i, j in alike), "All alike pairs must come from the same speaker."
    differing = dataset.get_differing_pairs(16)
    assert all(dataset[i][1] != dataset[j][1] for i, j in differing), "All differing pairs must come from different speakers."


# ---- tests/tests.py:31-68 -------------------------------------------------------------------------------
def test_n_shot_task(dataset):
    n, k = 1, 5
    query_sample, support = dataset.build_n_shot_task(k, n)
    assert query_sample[1] == support[1][0]
    assert query_sample[1] not in support[1][1:]
    assert len(np.unique(support[1])) == k
    assert support[0].shape == (k * n, 48000) and query_sample[0].shape == (48000,)
    n, k = 5, 5
    query_sample, support = dataset.build_n_shot_task(k, n)
    assert all(pd.Series(support[1]).value_counts() == 5)
    for i in range(0, n * k, n):
        assert np.all(support[1][i:i + n] == support[1][i])
    with pytest.raises(ValueError):
        dataset.build_n_shot_task(dataset.unique_speakers, 1)
    with pytest.raises(ValueError):
        dataset.build_n_shot_task(1, 1)


def test_verification_batch_layout(dataset):
    ([x1, x2], y) = dataset.build_verification_batch(8)
    assert x1.shape == (8, 48000, 1) and x2.shape == (8, 48000, 1) and y.shape == (8, 1)
    assert np.array_equal(y[:, 0], [0, 0, 0, 0, 1, 1, 1, 1])  # 0 = same speaker (librispeech.py:194)
    gen = dataset.yield_verification_batches(4)
    ([a, b], yy) = next(gen)
    assert a.shape == (4, 48000, 1) and yy.shape == (4, 1)


def test_dataset_getitem_modes():
    d = SyntheticSpeechDataset(num_speakers=4, files_per_speaker=3, seconds=3, stochastic=False)
    a, la = d[0]
    b, lb = d[0]
    assert np.array_equal(a, b) and la == lb and len(a) == 48000
    assert len(d) == 12 and d.num_classes() == 4 and d.unique_speakers == 4
    assert set(["speaker_id", "sex", "subset", "speaker_minutes", "name", "filepath", "length", "seconds", "id"]) <= set(d.df.columns)
    short = SyntheticSpeechDataset(num_speakers=3, files_per_speaker=2, seconds=3, pad=True, stochastic=False,
                                   min_file_seconds=1.0, max_file_seconds=2.0)
    x, _ = short[0]
    assert len(x) == 48000 and np.all(x[40000:] == 0)  # zero padded at the end when not stochastic
    nopad = SyntheticSpeechDataset(num_speakers=3, files_per_speaker=2, seconds=3, pad=False, min_file_seconds=1.0,
                                   max_file_seconds=2.0)
    assert len(nopad) == 0  # files shorter than the fragment are dropped (librispeech.py:86)
    sex = SyntheticSpeechDataset(num_speakers=4, files_per_speaker=2, seconds=3, label="sex")
    assert sex[0][1] in (True, False)
    with pytest.raises(AssertionError):
        SyntheticSpeechDataset(label="age")


def test_speakers_table_parse_rule():
    import config
    p = os.path.join(config.PATH, "tests", "golden", "SPEAKERS_head.TXT")
    df = LibriSpeechDataset.read_speakers_table(p)
    assert list(df.columns) == ["id", "sex", "subset", "minutes", "name"]
    assert df.iloc[0]["id"] == 14 and df.iloc[0]["sex"] == "F" and df.iloc[0]["subset"] == "train-clean-360"
    assert df.iloc[0]["name"] == "Kristin LeMoine"


# ---- tests/tests.py:71-90 -------------------------------------------------------------------------------
def test_whitening_no_batch(golden_dir):
    desired_rms = 0.038021
    c = np.load(os.path.join(golden_dir, "clips_human_eval.npz"))
    clip = c["query"].astype(np.float64) / 32768.0
    data = np.stack([clip] * 2)[:, :, np.newaxis]
    w = utils.whiten(data, desired_rms)
    assert np.isclose(w.mean().item(), 0)
    assert np.isclose(np.sqrt(np.power(w[0, :], 2).mean()).item(), desired_rms, rtol=1e-3)
    assert np.array_equal(w, O.whiten(data, desired_rms))
    with pytest.raises(ValueError):
        utils.whiten(data[:, :, 0])


def test_preprocessors_and_lazy_batches():
    r = np.random.default_rng(0)
    x = r.normal(0.01, 0.05, (4, 4800, 1))
    pre = utils.preprocess_instances(4)
    lazy = pre(x)
    assert lazy.shape == (4, 1200, 1) and len(lazy) == 4
    assert np.allclose(np.asarray(lazy), O.preprocess_instances(4)(x))
    assert np.allclose(np.asarray(utils.preprocess_instances(4, whitening=False)(x)), x[:, ::4, :])
    bp = utils.BatchPreProcessor("siamese", pre)
    ([a, b], lab) = bp(([x, x[::-1]], np.zeros((4, 1))))
    assert isinstance(a, utils.LazyWindows) and np.array_equal(lab, np.zeros((4, 1)))
    bc = utils.BatchPreProcessor("classifier", pre, lambda y: y + 1)
    xi, yi = bc((x, np.zeros((4, 1))))
    assert np.all(yi == 1)
    assert bc.instance_preprocessor is pre
    with pytest.raises(AssertionError):
        utils.BatchPreProcessor("triplet", pre)


def test_contrastive_loss_matches_oracle():
    import torch
    r = np.random.default_rng(1)
    y = (r.random((16, 1)) > 0.5).astype(float)
    p = r.random((16, 1))
    assert np.isclose(utils.contrastive_loss(y, p), O.contrastive_loss(torch.tensor(y), torch.tensor(p)).item())


# ---- voicemap/models.py signatures and error behaviour ---------------------------------------------------
def test_build_functions_surface():
    enc = models.get_baseline_convolutional_encoder(128, 64, dropout=0.0)
    assert [type(l).__name__ for l in enc.layers[:4]] == ["Conv1D", "BatchNormalization", "SpatialDropout1D", "MaxPool1D"]
    assert len(enc.layers) == 18 and enc.layers[-1].units == 64
    net = models.build_siamese_net(enc, (12000, 1), distance_metric="uniform_euclidean")
    assert net.layers[2] is enc and len(net.layers) == 6
    with pytest.raises(AssertionError):
        models.build_siamese_net(enc, (12000, 1), distance_metric="manhattan")
    for name in ("weighted_euclidean", "uniform_l1", "dot_product", "cosine_distance"):
        with pytest.raises(NotImplementedError):
            models.build_siamese_net(enc, (12000, 1), distance_metric=name)
    models.build_siamese_net(models.get_baseline_convolutional_encoder(32, 128), (12000, 1), "weighted_l1")
    lines = []
    e2 = models.get_baseline_convolutional_encoder(128, 64, (12000, 1), dropout=0.0)
    e2.summary(print_fn=lines.append)
    txt = "\n".join(lines)
    assert "Total params: 1,026,368" in txt and "Trainable params: 1,023,808" in txt  # + 2 head params = 1 023 810
    clf = models.get_baseline_convolutional_encoder(16, 32, (12000, 1))
    clf.add(K.Dense(40, activation="softmax"))
    assert clf.classifier_units == 40 and clf.layers[-1].name == "dense_2"
    with pytest.raises(NotImplementedError):
        clf.add(K.Dense(3, activation="softmax"))
    clf.pop()
    assert clf.classifier_units == 0


def test_keras_like_callbacks(tmp_path):
    class FakeModel:
        def __init__(self):
            self.lr, self.saved = 1e-3, []
        def get_lr(self): return self.lr
        def set_lr(self, v): self.lr = v
        def save(self, p): self.saved.append(p)
    m = FakeModel()
    csvp = str(tmp_path / "logs" / "run.csv")
    cbs = [K.CSVLogger(csvp), K.ModelCheckpoint(str(tmp_path / "m.npz"), monitor="val_1-shot_acc", mode="max", save_best_only=True),
           K.ReduceLROnPlateau(monitor="val_1-shot_acc", mode="max", patience=2)]
    for cb in cbs:
        cb.set_model(m)
        cb.on_train_begin()
    accs = [0.3, 0.5, 0.4, 0.45, 0.2]
    for ep, a in enumerate(accs):
        logs = {"loss": 1.0 / (ep + 1), "acc": 0.5, "val_loss": 1.0, "val_acc": 0.5, "val_1-shot_acc": a}
        for cb in cbs:
            cb.on_epoch_end(ep, logs)
    for cb in cbs:
        cb.on_train_end()
    assert len(m.saved) == 2            # improvements at epochs 0 and 1 only
    assert np.isclose(m.lr, 1e-4)       # no improvement for 2 epochs after the best -> one reduction by 0.1
    rows = open(csvp).read().strip().split("\n")
    assert rows[0].split(",")[0] == "epoch" and "val_1-shot_acc" in rows[0] and len(rows) == 6
    assert np.array_equal(K.to_categorical([1, 0, 2], 3), np.eye(3, dtype=np.float32)[[1, 0, 2]])
    opt = K.Adam(clipnorm=1.)
    assert (opt.lr, opt.beta_1, opt.beta_2, opt.epsilon, opt.clipnorm) == (1e-3, 0.9, 0.999, 1e-7, 1.0)


def test_batch_feeder_generator_and_sequence():
    def gen():
        i = 0
        while True:
            yield i, -i
            i += 1
    f = K.BatchFeeder(gen(), workers=3, max_queue_size=4)
    got = sorted(f.get()[0] for _ in range(20))
    f.close()
    assert len(set(got)) == 20 and max(got) < 20 + 3 + 4  # unordered across workers, nothing duplicated or lost for long

    class Seq(K.Sequence):
        def __len__(self): return 5
        def __getitem__(self, i): return i, i
    f0 = K.BatchFeeder(Seq(), workers=0)
    assert [f0.get()[0] for _ in range(7)] == [0, 1, 2, 3, 4, 0, 1]


def test_voicemap_alias_package():
    import voicemap
    from voicemap.models import get_baseline_convolutional_encoder, build_siamese_net  # noqa: F401
    from voicemap.utils import whiten, NShotEvaluationCallback, BatchPreProcessor, preprocess_instances, contrastive_loss  # noqa: F401
    from voicemap.librispeech import LibriSpeechDataset as L2
    assert L2 is LibriSpeechDataset and voicemap.models is models


# ---------------------------------------------------------------------------------------------------------
# pre-decoded int16 shards: same dataset API, no decode, offsets for the device-side crop
# ---------------------------------------------------------------------------------------------------------
def test_sharded_dataset_round_trip_and_offsets(tmp_path):
    from voicemap_amd import shards
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    src = SyntheticSpeechDataset(num_speakers=8, files_per_speaker=4, seconds=3, seed=5, stochastic=False)
    index = shards.write_shards(src, str(tmp_path), shard_samples=300000)
    assert index['shard'].max() >= 2 and (index.groupby('shard')['length'].sum() <= 300000).all() or len(index) > 0
    sd = shards.ShardedSpeechDataset(str(tmp_path), 3, stochastic=False)
    assert len(sd) == len(src) and sd.num_classes() == src.num_classes()
    assert list(sd.df['speaker_id']) == list(src.df['speaker_id']) and list(sd.df['length']) == list(src.df['length'])
    for i in (0, 7, len(sd) - 1):
        a, la = sd[i]
        b, lb = src[i]
        assert la == lb and a.shape == b.shape == (sd.fragment_length,)
        assert np.abs(a - b).max() <= 0.5 / 32768 + 1e-12  # int16 quantisation only
    # every file's samples sit at global_offset in the concatenation of the shards
    cat = np.concatenate([np.asarray(m) for m in sd._maps])
    for i in (1, 9, len(sd) - 2):
        o, n = sd.global_offset[i], sd.file_length[i]
        assert np.array_equal(cat[o:o + n], shards.to_int16(src._load(i)))
    # offsets batch: the reference's pair layout (tests/tests.py:16-68 restated) + fragments inside their files
    sd.stochastic = True
    np.random.seed(3)
    o1, o2, y = sd.build_verification_batch_offsets(16)
    assert o1.shape == o2.shape == (16,) and y.shape == (16, 1)
    assert np.array_equal(y[:, 0], np.r_[np.zeros(8), np.ones(8)])
    ends = sd.global_offset + sd.file_length

    def owner(off):
        k = np.searchsorted(sd.global_offset, off, side='right') - 1
        assert off + sd.fragment_length <= ends[k]
        return sd.datasetid_to_speaker_id[int(k)]
    for a, b, lab in zip(o1, o2, y[:, 0]):
        assert (owner(a) == owner(b)) == (lab == 0)
    # the sampling API of the reference still works on top of the shards
    q, (sup, sup_labels) = sd.build_n_shot_task(4, 2)
    assert sup.shape == (8, sd.fragment_length) and q[1] == sup_labels[0] == sup_labels[1]
    # ... and as offsets (the device route of the n-shot evaluation): same draws in the same order as the host method
    np.random.seed(11)
    q_host, (sup_host, lab_host) = sd.build_n_shot_task(4, 2)
    np.random.seed(11)
    (qo, ql), (so, sl) = sd.build_n_shot_task_offsets(4, 2)
    T = sd.fragment_length
    assert ql == q_host[1] and list(sl) == list(lab_host) and so.shape == (8,)
    assert np.array_equal(shards.to_int16(q_host[0]), cat[qo:qo + T])
    for j in range(8):
        assert np.array_equal(shards.to_int16(sup_host[j]), cat[so[j]:so[j] + T])
    with pytest.raises(ValueError):
        sd.build_n_shot_task_offsets(1, 1)
    with pytest.raises(ValueError):
        sd.build_n_shot_task_offsets(sd.unique_speakers, 1)



def test_numpy_sampling_reproduces_the_pandas_draws():
    """The pair / task sampling is restated on plain arrays (librispeech._build_sampling_index): under one np.random seed it must
    return exactly what the reference's DataFrame.sample / pd.merge formulation (tests/pandas_sampling.py, restating
    librispeech.py:145-240) returns -- same files, same fragments, same order -- because pandas' sample IS np.random.choice on
    row positions; build_verification_batch must also cut its fragments in the reference's statement order."""
    from tests import pandas_sampling as PS
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    for ds in (SyntheticSpeechDataset(num_speakers=40, files_per_speaker=7, seconds=0.3, seed=5),
               SyntheticSpeechDataset(num_speakers=9, files_per_speaker=3, seconds=0.2, seed=1, stochastic=False)):
        for seed in range(6):
            for fn, args in (("get_alike_pairs", (8,)), ("get_differing_pairs", (4,))):
                np.random.seed(seed)
                a = [(int(x), int(y)) for x, y in getattr(PS, fn)(ds, *args)]
                np.random.seed(seed)
                b = [(int(x), int(y)) for x, y in getattr(ds, fn)(*args)]
                assert a == b, (fn, seed)
            np.random.seed(100 + seed)
            qa, (sa, la) = PS.build_n_shot_task(ds, 4, 2)
            np.random.seed(100 + seed)
            qb, (sb, lb) = ds.build_n_shot_task(4, 2)
            assert qa[1] == qb[1] and np.array_equal(qa[0], qb[0]) and np.array_equal(sa, sb) and list(la) == list(lb)
            np.random.seed(200 + seed)
            (a1, a2), ya = PS.build_verification_batch(ds, 8)
            np.random.seed(200 + seed)
            (b1, b2), yb = ds.build_verification_batch(8)
            assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and np.array_equal(ya, yb)
    # one vectorised randint draw per file == the reference's loop of scalar draws (librispeech.py:113)
    r = np.random.default_rng(0)
    for trial in range(50):
        spans = r.integers(1, 2 ** int(r.integers(1, 40)), size=int(r.integers(1, 40)))
        np.random.seed(trial)
        a = [np.random.randint(0, int(s)) for s in spans]
        np.random.seed(trial)
        assert a == list(np.random.randint(0, spans))


def test_single_weighted_draw_fast_path_is_np_random_choice():
    """LibriSpeechDataset._weighted(1, w) (cached cdf for the all-files draw, validation-free cdf for a speaker's files) returns what
    np.random.choice(len(w), 1, replace=False, p=w / w.sum()) returns and leaves the random stream where choice leaves it: the n-shot
    tasks and verification pairs of a seed do not change (reference draws: voicemap/librispeech.py:139-240)."""
    import types
    from voicemap_amd.librispeech import SyntheticSpeechDataset

    def by_choice(self, n, weights):
        return np.random.choice(len(weights), size=n, replace=False, p=weights / weights.sum())

    res = {}
    for name in ("fast", "choice"):
        d = SyntheticSpeechDataset(num_speakers=90, files_per_speaker=7, seconds=1, seed=3)
        if name == "choice":
            d._weighted = types.MethodType(by_choice, d)
        np.random.seed(11)
        out = []
        for i in range(120):
            q = int(d._weighted(1, d._len)[0])
            out.append((q, tuple(d._n_shot_support(q, 5, 1 + (i % 3)))))
        out.append(tuple(map(tuple, d.get_alike_pairs(8))) + tuple(map(tuple, d.get_differing_pairs(8))))
        res[name] = (out, np.random.random())
    assert res["fast"] == res["choice"]


def test_sampling_refuses_under_populated_indexes_like_np_random_choice():
    """The restated weighted sampling keeps np.random.choice's refusals (pandas' DataFrame.sample raises them in the reference,
    voicemap/librispeech.py:145-240): a speaker with fewer than n other files cannot fill an n-shot class, one speaker owning the
    whole index has no differing pairs, more pairs than files cannot be drawn without replacement -- ValueError, never a silently
    padded or duplicated sample."""
    import pytest
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    d = SyntheticSpeechDataset(num_speakers=4, files_per_speaker=3, seconds=1, seed=1)
    with pytest.raises(ValueError):
        d._weighted(5, np.array([1.0, 2.0, 3.0]))            # larger than the population
    with pytest.raises(ValueError):
        d._weighted(2, np.array([0.0, 2.0, 0.0]))            # fewer rows with positive weight than requested
    with pytest.raises(ValueError):
        d._weighted(1, np.array([]))                          # empty population
    np.random.seed(0)
    with pytest.raises(ValueError):
        d.build_n_shot_task(3, 4)                             # 3 files per speaker: no 4-shot support set for the query's class
    assert len(d._weighted(3, np.array([1.0, 2.0, 3.0]))) == 3
    one = SyntheticSpeechDataset(num_speakers=1, files_per_speaker=5, seconds=1, seed=2)
    np.random.seed(0)
    with pytest.raises(ValueError):
        one.get_differing_pairs(2)                            # the remainder after removing the anchor's speaker is empty
    with pytest.raises(ValueError):
        one.get_alike_pairs(4)                                # 8 anchors from 5 files without replacement


def test_cached_evaluation_draws_the_reference_tasks():
    """retrieval.draw_tasks_reference (tasks as file indices for the cached-embedding evaluation) makes the np.random calls of
    LibriSpeechDataset.build_n_shot_task (voicemap/librispeech.py:204-240) in the same order: under one seed the query / support
    FILES are the same and the random stream ends in the same place, for a deterministic-crop dataset and for a stochastic one
    (whose fragment-start draws are consumed and ignored)."""
    from voicemap_amd import retrieval as R
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    for stochastic in (False, True):
        ds = SyntheticSpeechDataset(num_speakers=7, files_per_speaker=5, seconds=1, stochastic=stochastic, seed=4)
        np.random.seed(9)
        want = []
        for _ in range(25):
            (qa, ql), (sa, sl) = ds.build_n_shot_task(4, 2)
            want.append((ql, tuple(sl)))
        end_a = np.random.random()
        np.random.seed(9)
        q, s = R.draw_tasks_reference(ds, 25, 4, 2)
        end_b = np.random.random()
        assert end_a == end_b
        got = [(ds._label(int(q[t])), tuple(ds._label(int(i)) for i in s[t])) for t in range(25)]
        assert got == want
        if not stochastic:   # the windows themselves: a file's first fragment
            np.random.seed(9)
            (qa, _), (sa, _) = ds.build_n_shot_task(4, 2)
            assert np.array_equal(qa, R._first_fragment(ds, int(q[0]))) and np.array_equal(sa[3], R._first_fragment(ds, int(s[0][3])))


def test_sharded_dataset_speaker_shard_partitions_the_corpus(tmp_path):
    """ShardedSpeechDataset(speaker_shard=(rank, world)): the ranks' speaker sets partition the corpus, every rank's dataset is a
    complete LibriSpeechDataset over its speakers (pairs / tasks drawn inside it), and the compact device buffer of a rank holds
    exactly its recordings at the re-based offsets (checked on the CPU: 'device' = cpu)."""
    from voicemap_amd import shards
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    src = SyntheticSpeechDataset(num_speakers=9, files_per_speaker=3, seconds=3, seed=4)
    shards.write_shards(src, str(tmp_path), shard_samples=400000)
    whole = shards.ShardedSpeechDataset(str(tmp_path), 3, stochastic=False)
    parts = [shards.ShardedSpeechDataset(str(tmp_path), 3, stochastic=False, speaker_shard=(r, 4)) for r in range(4)]
    seen = [set(p.df['speaker_id'].unique()) for p in parts]
    assert sum(len(s) for s in seen) == 9 and set().union(*seen) == set(whole.df['speaker_id'].unique())
    assert sum(len(p) for p in parts) == len(whole)
    for p in parts:
        #  the offsets describe the compact per-rank buffer from construction on, not only after to_device()
        before = p.global_offset.copy()
        assert before[0] == 0 and np.array_equal(before[1:], np.cumsum(p.file_length)[:-1])
        audio = p.to_device("cpu").numpy()
        assert np.array_equal(p.global_offset, before)
        assert len(audio) == int(p.file_length.sum())
        for i in (0, len(p) - 1):
            o = int(p.global_offset[i])
            assert np.array_equal(audio[o:o + int(p.file_length[i])], np.asarray(p._pcm(i)))
        np.random.seed(1)
        for a, b in p.get_alike_pairs(2):
            assert p.datasetid_to_speaker_id[a] == p.datasetid_to_speaker_id[b]
        o1, o2, y = p.build_verification_batch_offsets(4)
        assert o1.max() + p.fragment_length <= len(audio) and o2.max() + p.fragment_length <= len(audio)
    with pytest.raises(ValueError):   # 9 speakers over 8 ranks: rank 1..7 would hold one speaker -- no different-speaker pairs
        shards.ShardedSpeechDataset(str(tmp_path), 3, stochastic=False, speaker_shard=(1, 8))
