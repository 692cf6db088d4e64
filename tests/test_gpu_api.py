"""-m gpu tests of the drop-in Python surface (voicemap_amd.models / utils / librispeech driving the HIP engine): a
miniature of experiments/train_siamese.py with all four callbacks, n-shot evaluation against the oracle on identical
tasks, checkpoint round trip, the classifier script's flow, and the reference's known-answer task through the public API."""
import os

import numpy as np
import pytest
import torch

from oracle import voicemap_oracle as O
from voicemap_amd import keras_like as K
from voicemap_amd import models, utils
from voicemap_amd.librispeech import SyntheticSpeechDataset

pytestmark = pytest.mark.gpu


def _oracle_params(eng):
    return {k: torch.tensor(v, dtype=torch.float64) for k, v in eng.get_params().items()}


def test_fit_generator_like_train_siamese(tmp_path):
    np.random.seed(0)
    train = SyntheticSpeechDataset(num_speakers=30, files_per_speaker=4, seconds=0.6, pad=True)
    valid = SyntheticSpeechDataset(num_speakers=12, files_per_speaker=4, seconds=0.6, stochastic=False, pad=True, seed=5)
    bp = utils.BatchPreProcessor("siamese", utils.preprocess_instances(4))
    train_gen = (bp(b) for b in train.yield_verification_batches(8))
    valid_gen = (bp(b) for b in valid.yield_verification_batches(8))
    enc = models.get_baseline_convolutional_encoder(16, 32, dropout=0.05, dtype="f32")
    net = models.build_siamese_net(enc, (2400, 1), distance_metric="uniform_euclidean")
    net.compile(loss="binary_crossentropy", optimizer=K.Adam(clipnorm=1.), metrics=["accuracy"])
    csvp = str(tmp_path / "logs" / "run.csv")
    ckpt = str(tmp_path / "models" / "best.hdf5")  # Keras-2.2.2 HDF5, as the reference's ModelCheckpoint writes
    hist = net.fit_generator(generator=train_gen, steps_per_epoch=3, validation_data=valid_gen, validation_steps=2, epochs=2,
                             workers=2, use_multiprocessing=True, verbose=0,
                             callbacks=[utils.NShotEvaluationCallback(6, 1, 5, valid, preprocessor=bp),
                                        K.CSVLogger(csvp),
                                        K.ModelCheckpoint(ckpt, monitor="val_1-shot_acc", mode="max", save_best_only=True),
                                        K.ReduceLROnPlateau(monitor="val_1-shot_acc", mode="max", verbose=0)])
    assert set(hist.history) >= {"loss", "acc", "val_loss", "val_acc", "val_1-shot_acc", "lr"}
    assert len(hist.history["loss"]) == 2 and all(np.isfinite(hist.history["loss"]))
    assert net.engine.iterations == 6
    header = open(csvp).readline().strip().split(",")
    # like Keras: CSVLogger fixes its columns at the first epoch, BEFORE ReduceLROnPlateau (later in the list) adds 'lr'
    assert header == ["epoch", "acc", "loss", "val_1-shot_acc", "val_acc", "val_loss"]
    assert os.path.exists(ckpt)
    # checkpoint round trip: same predictions, optimizer state restored
    ([x1, x2], _) = bp(valid.build_verification_batch(4))
    loaded = models.load_model(ckpt)
    assert loaded.engine.iterations in (3, 6)
    net2 = models.load_model(ckpt)
    assert np.array_equal(loaded.predict([x1, x2]), net2.predict([x1, x2]))
    # encoder shares the trained weights (utils.py:141 uses model.layers[2])
    e = net.layers[2].predict(x1)
    assert e.shape == (4, 32) and np.isfinite(e).all()


def test_fit_generator_deferred_batch_logs_equal_the_per_batch_ones():
    """fit_generator reads the per-batch (loss, acc) in blocks when no callback has an on_batch_end (the host keeps enqueueing instead of
    waiting for every step): the epoch logs and the weights are those of the per-batch loop, bit for bit; a callback with an on_batch_end
    gets every batch's numbers as before."""
    train = SyntheticSpeechDataset(num_speakers=12, files_per_speaker=3, seconds=0.5, pad=True, seed=1)
    bp = utils.BatchPreProcessor("siamese", utils.preprocess_instances(4))
    np.random.seed(5)
    batches = [bp(train.build_verification_batch(8)) for _ in range(6)]

    class Seen(K.Callback):
        def __init__(self):
            super().__init__()
            self.losses = []

        def on_batch_end(self, batch, logs=None):
            self.losses.append(logs["loss"])

    def run(defer, cbs):
        torch.manual_seed(3)
        enc = models.get_baseline_convolutional_encoder(16, 24, dropout=0.0, dtype="f16")
        net = models.build_siamese_net(enc, (2000, 1), distance_metric="uniform_euclidean")
        net.compile(loss="binary_crossentropy", optimizer=K.Adam(clipnorm=1.), metrics=["accuracy"])
        net.defer_batch_logs = defer
        h = net.fit_generator(generator=iter(batches), steps_per_epoch=3, epochs=2, workers=0, verbose=0, callbacks=cbs)
        return h.history, net.engine.P.clone()
    h1, p1 = run(True, [])
    h0, p0 = run(False, [])
    seen = Seen()
    h2, p2 = run(True, [seen])
    assert h1["loss"] == h0["loss"] == h2["loss"] and h1["acc"] == h0["acc"] and torch.equal(p1, p0) and torch.equal(p1, p2)
    assert len(seen.losses) == 6 and np.isclose(np.mean(seen.losses[:3]), h2["loss"][0])


@pytest.mark.parametrize("ext", ["hdf5", "npz"])
def test_checkpoint_resume_is_bit_identical(tmp_path, ext):
    """model.save -> load_model restores weights, moving statistics, Adam slots and the iteration counter exactly: the next
    training step of the loaded model equals the next step of the original, bit for bit (SURVEY 8f.3 for ext == hdf5)."""
    train = SyntheticSpeechDataset(num_speakers=12, files_per_speaker=3, seconds=0.5, pad=True, seed=1)
    bp = utils.BatchPreProcessor("siamese", utils.preprocess_instances(4))
    enc = models.get_baseline_convolutional_encoder(16, 24, dropout=0.0, dtype="f32")
    net = models.build_siamese_net(enc, (2000, 1), distance_metric="weighted_l1")
    net.compile(loss=utils.contrastive_loss, optimizer=K.Adam(lr=2e-3, clipnorm=1.), metrics=["accuracy"])
    np.random.seed(5)
    batches = [bp(train.build_verification_batch(8)) for _ in range(3)]
    for x, y in batches[:2]:
        net.train_on_batch(x, y)
    path = str(tmp_path / ("resume." + ext))
    net.save(path)
    loaded = models.load_model(path)
    le, ne = loaded._ensure_engine(), net.engine
    assert le.iterations == ne.iterations == 2 and le.lr == ne.lr and le.clipnorm == ne.clipnorm
    for a, b in ((le.P, ne.P), (le.M, ne.M), (le.V, ne.V), (le.NT, ne.NT)):
        assert torch.equal(a, b)
    assert loaded._loss_name() == "contrastive" and loaded.distance_metric == "weighted_l1"
    la = net.train_on_batch(*batches[2])
    lb = loaded.train_on_batch(*batches[2])
    assert la == lb and torch.equal(le.P, ne.P) and torch.equal(le.M, ne.M) and torch.equal(le.V, ne.V)
    if ext == "hdf5":
        # weights-only round trip through model.save_weights / load_weights, and into a separately built encoder
        w = str(tmp_path / "w.h5")
        net.save_weights(w)
        enc2 = models.get_baseline_convolutional_encoder(16, 24, dropout=0.0, dtype="f32")
        net2 = models.build_siamese_net(enc2, (2000, 1), distance_metric="weighted_l1")
        net2.compile(loss=utils.contrastive_loss, optimizer=K.Adam())
        net2.load_weights(w)
        x, _ = batches[0]
        assert np.array_equal(net2.predict(x), net.predict(x))


def test_load_model_reads_a_libhdf5_written_keras_file(golden_dir):
    """A classifier stored by h5py in Keras' layout (tests/golden/make_h5py_fixture.py) loads into a working model whose
    weights are the stored arrays."""
    m = models.load_model(os.path.join(golden_dir, "keras_layout_h5py.hdf5"), dtype="f32")
    exp = {k.replace("|", "/"): v for k, v in np.load(os.path.join(golden_dir, "keras_layout_h5py_expected.npz")).items()}
    eng = m._ensure_engine()
    got = eng.get_params()
    assert np.array_equal(got["conv2.kernel"], exp["conv1d_2/kernel:0"]) and np.array_equal(got["head.kernel"], exp["dense_2/kernel:0"])
    assert np.array_equal(got["bn4.moving_variance"], exp["batch_normalization_4/moving_variance:0"])
    assert eng.iterations == 37 and abs(eng.lr - 0.0005) < 1e-12 and eng.clipnorm == 1.0
    assert np.array_equal(eng.view("dense.kernel", eng.M).cpu().numpy(), exp["optimizer/training/Adam/Variable_16:0"])
    p = m.predict(np.random.default_rng(0).normal(0, 0.1, (3, 800, 1)))
    assert p.shape == (3, 5) and np.allclose(p.sum(1), 1.0, atol=1e-4)


@pytest.mark.parametrize("n,k,dist", [(1, 5, "euclidean"), (5, 5, "euclidean"), (3, 4, "cosine"), (2, 6, "dot_product")])
def test_n_shot_evaluation_matches_oracle(n, k, dist):
    valid = SyntheticSpeechDataset(num_speakers=14, files_per_speaker=8, seconds=0.5, stochastic=False, seed=3)
    enc = models.get_baseline_convolutional_encoder(16, 32, dropout=0.0, dtype="f32")
    net = models.build_siamese_net(enc, (2000, 1))
    net.compile(loss=utils.contrastive_loss, optimizer=K.Adam(clipnorm=1.))
    eng = net._ensure_engine()
    # make the moving statistics non-trivial
    r = np.random.default_rng(0)
    eng.set_params({f"bn{i}.moving_mean": r.normal(0.05, 0.02, c) for i, (_, c, _) in enumerate(eng.blocks, 1)})
    eng.set_params({f"bn{i}.moving_variance": r.uniform(0.01, 0.1, c) for i, (_, c, _) in enumerate(eng.blocks, 1)})
    bp = utils.BatchPreProcessor("siamese", utils.preprocess_instances(4))
    num_tasks = 6
    np.random.seed(11)
    got = utils.n_shot_task_evaluation(net, valid, bp, num_tasks, n, k, network_type="siamese", distance=dist)
    # the oracle on the very same tasks (same RNG stream)
    arch = O.EncoderArch(blocks=eng.blocks, embedding_dimension=32, dropout=0.0)
    p = _oracle_params(eng)
    pre = O.preprocess_instances(4)
    np.random.seed(11)
    want = 0
    for _ in range(num_tasks):
        q, s = valid.build_n_shot_task(k, n)
        if n == 1:
            i1 = pre(np.stack([q[0]] * k)[:, :, None])
            i2 = pre(s[0][:, :, None])
            pred, _, _ = O.siamese_forward(arch, p, torch.tensor(i1), torch.tensor(i2), False)
            want += int(pred[:, 0].argmin() == 0)
        else:
            qe = O.encoder_forward(arch, p, torch.tensor(pre(q[0].reshape(1, -1, 1))), False).numpy()
            se = O.encoder_forward(arch, p, torch.tensor(pre(s[0][:, :, None])), False).numpy()
            want += int(np.argmin(O.n_shot_prediction(qe, se, n, k, dist)) == 0)
    assert got == want


def test_classifier_flow_like_train_classifier():
    np.random.seed(1)
    train = SyntheticSpeechDataset(num_speakers=10, files_per_speaker=4, seconds=0.5)
    speakers = sorted(train.df["speaker_id"].unique())
    mapping = {s: i for i, s in enumerate(speakers)}

    def label_pre(y):
        return K.to_categorical(np.array([mapping[i] for i in y[:, 0]])[:, None], train.num_classes())

    bp = utils.BatchPreProcessor("classifier", utils.preprocess_instances(4), label_pre)

    class Batched(K.Sequence):
        def __len__(self):
            return len(train) // 8

        def __getitem__(self, item):
            idx = range(item * 8, item * 8 + 8)
            X = np.stack([train[i][0][:, None] for i in idx])
            y = np.stack([train[i][1] for i in idx])[:, None]
            return bp((X, y))

    clf = models.get_baseline_convolutional_encoder(16, 32, (2000, 1), dropout=0.0, dtype="f32")
    clf.add(K.Dense(train.num_classes(), activation="softmax"))
    clf.compile(loss="categorical_crossentropy", optimizer=K.Adam(clipnorm=1.), metrics=["accuracy"])
    valid = SyntheticSpeechDataset(num_speakers=10, files_per_speaker=4, seconds=0.5, stochastic=False, seed=9)
    hist = clf.fit_generator(Batched(), steps_per_epoch=4, epochs=2, verbose=0, workers=0,
                             callbacks=[utils.NShotEvaluationCallback(5, 1, 5, valid, preprocessor=bp, mode="classifier")])
    assert np.isfinite(hist.history["loss"]).all() and "val_1-shot_acc" in hist.history
    prob = clf.predict(Batched()[0][0])
    assert prob.shape == (8, 10) and np.allclose(prob.sum(1), 1, atol=1e-5)
    emb = utils.get_bottleneck(clf, Batched()[0][0])
    assert emb.shape == (8, 32)


def test_known_answer_task_through_public_api(golden_dir):
    """notebooks/Human_Evaluation.ipynb cell 8 via load_keras_checkpoint_npz + n_shot_task_evaluation."""
    c = np.load(os.path.join(golden_dir, "clips_human_eval.npz"))

    class OneTask:
        unique_speakers = 6

        def build_n_shot_task(self, k, n=1):
            q = c["query"].astype(np.float64) / 32768.0
            s = c["support"].astype(np.float64) / 32768.0
            order = [4, 0, 1, 2, 3]  # correct speaker first, as build_n_shot_task lays tasks out
            return (q, 0), (s[order], np.arange(5))

    net = models.load_keras_checkpoint_npz(os.path.join(golden_dir, "ckpt_cfgCK_weights.npz"), dtype="f32")
    bp = utils.BatchPreProcessor("siamese", utils.preprocess_instances(4))
    assert utils.n_shot_task_evaluation(net, OneTask(), bp, 1, 1, 5, network_type="siamese") == 1


@pytest.mark.gpu
def test_device_side_crop_equals_host_crop(tmp_path):
    """windows cropped on the device from resident int16 shards (vm_crop_decimate_whiten) give bit-identical
    preprocessed inputs, and a bit-identical training step, to the same windows cropped on the host and fed as int16."""
    import torch
    from voicemap_amd import shards
    from voicemap_amd.engine import HipEncoderEngine
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    src = SyntheticSpeechDataset(num_speakers=10, files_per_speaker=3, seconds=3, seed=9)
    shards.write_shards(src, str(tmp_path), shard_samples=500000)
    sd = shards.ShardedSpeechDataset(str(tmp_path), 3, stochastic=True)
    audio = sd.to_device("cuda")
    np.random.seed(4)
    o1, o2, y = sd.build_verification_batch_offsets(8)
    T = sd.fragment_length
    cat = np.concatenate([np.asarray(m) for m in sd._maps])
    x1 = np.stack([cat[o:o + T] for o in o1])
    x2 = np.stack([cat[o:o + T] for o in o2])
    blocks = [(32, 16, 4), (3, 32, 2), (3, 48, 2), (3, 64, 2)]
    res = []
    for mode in ("host", "device"):
        eng = HipEncoderEngine(blocks, 24, dropout=0.0, head="uniform_euclidean", dtype="f32", seed=7)
        if mode == "host":
            pl = eng.siamese_train_step(torch.from_numpy(x1), torch.from_numpy(x2), y, preprocessed=False, downsampling=4,
                                        drop_masks=None)
        else:
            pl = eng.siamese_train_step_from_offsets(audio, torch.from_numpy(o1).cuda(), torch.from_numpy(o2).cuda(), y, T,
                                                     downsampling=4, drop_masks=None)
        torch.cuda.synchronize()
        res.append((pl["x0"].clone(), pl["loss_acc"].clone(), eng.G.clone(), eng.P.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,mode,dist", [(1, 5, "siamese", "euclidean"), (3, 4, "siamese", "cosine"), (2, 5, "classifier", "dot_product")])
def test_device_resident_n_shot_equals_host_n_shot(tmp_path, n, k, mode, dist):
    """n_shot_task_evaluation over a device-resident corpus (tasks = start offsets, crop in the preprocessing
    kernel) returns the same n_correct as the host route fed with the same tasks (same RNG seed => same files and fragments)."""
    from voicemap_amd import models, shards, utils
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    src = SyntheticSpeechDataset(num_speakers=9, files_per_speaker=5, seconds=3, seed=2)
    shards.write_shards(src, str(tmp_path), shard_samples=600000)
    sd = shards.ShardedSpeechDataset(str(tmp_path), 3, stochastic=True)
    enc = models.get_baseline_convolutional_encoder(16, 24, dropout=0.0, dtype="f32")
    if mode == "siamese":
        net = models.build_siamese_net(enc, (sd.fragment_length // 4, 1))
        net.compile(loss="binary_crossentropy", optimizer="adam")
    else:
        from voicemap_amd.keras_like import Dense
        enc.add(Dense(sd.num_classes(), activation="softmax"))
        enc.compile(loss="categorical_crossentropy", optimizer="adam")
        net = enc
    bp = utils.BatchPreProcessor(mode, utils.preprocess_instances(4))
    tasks = 12
    np.random.seed(21)
    host = utils.n_shot_task_evaluation(net, sd, bp, tasks, n, k, network_type=mode, distance=dist)
    sd.to_device("cuda")
    np.random.seed(21)
    dev = utils.n_shot_task_evaluation(net, sd, bp, tasks, n, k, network_type=mode, distance=dist)
    assert 0 <= host <= tasks and dev == host


def test_train_siamese_script_with_device_resident_data(tmp_path, monkeypatch):
    """experiments/train_siamese.py --device-data: the training windows exist only as offsets into an HBM-resident int16
    buffer; the script runs end to end (fit_generator, validation, n-shot callback, checkpoint) and the loss is finite."""
    import config
    from experiments import _common as C
    from experiments import train_siamese
    monkeypatch.setattr(config, "PATH", str(tmp_path))
    monkeypatch.setattr(C, "PATH", str(tmp_path))
    os.makedirs(os.path.join(str(tmp_path), "logs"), exist_ok=True)
    os.makedirs(os.path.join(str(tmp_path), "models"), exist_ok=True)
    hist = train_siamese.main(["--synthetic", "--device-data", os.path.join(str(tmp_path), "shards"), "--filters", "16",
                               "--embedding-dimension", "16", "--batchsize", "16", "--epochs", "2", "--steps-per-epoch", "3",
                               "--validation-steps", "2", "--num-evaluation-tasks", "4", "--n-seconds", "3", "--dtype", "f32"])
    assert len(hist.history["loss"]) == 2 and all(np.isfinite(v) for v in hist.history["loss"])
    assert os.path.exists(os.path.join(str(tmp_path), "shards", "index.csv"))


def test_f32s_model_tracks_f32_model_through_the_public_api(tmp_path):
    """dtype='f32s' (fp32 storage, split-bf16 GEMM products) through the drop-in surface: the same batches train a model whose
    losses, predictions and saved checkpoint stay within 1e-4 of the dtype='f32' model's, and the checkpoint reloads in its own mode."""
    train = SyntheticSpeechDataset(num_speakers=12, files_per_speaker=3, seconds=0.5, pad=True, seed=1)
    bp = utils.BatchPreProcessor("siamese", utils.preprocess_instances(4))
    np.random.seed(11)
    batches = [bp(train.build_verification_batch(8)) for _ in range(3)]
    nets = {}
    for dt in ("f32", "f32s"):
        enc = models.get_baseline_convolutional_encoder(16, 24, dropout=0.0, dtype=dt)
        net = models.build_siamese_net(enc, (2000, 1), distance_metric="uniform_euclidean")
        net.compile(loss=utils.contrastive_loss, optimizer=K.Adam(lr=1e-3, clipnorm=1.), metrics=["accuracy"])
        nets[dt] = net
    nets["f32s"].set_weights(nets["f32"].get_weights())
    for x, y in batches[:2]:
        la, lb = nets["f32"].train_on_batch(x, y), nets["f32s"].train_on_batch(x, y)
        assert abs(la[0] - lb[0]) < 1e-4 * max(1.0, abs(la[0]))
    x, _ = batches[2]
    pa, pb = nets["f32"].predict(x), nets["f32s"].predict(x)
    assert np.abs(pa - pb).max() < 1e-4
    path = str(tmp_path / "f32s.hdf5")
    nets["f32s"].save(path)
    loaded = models.load_model(path)
    assert loaded._ensure_engine().dtype == nets["f32s"].engine.dtype
    assert np.array_equal(loaded.predict(x), pb)


def test_k_way_accuracy_script_cached_sweep(tmp_path, monkeypatch):
    """experiments/k_way_accuracy.py (the reference's sweep, experiments/k_way_accuracy.py:52-69) in its --cached form: the model is
    saved as a Keras HDF5 file, loaded back, the evaluation set is embedded ONCE and every (k, n) cell runs on the cached matrix --
    with the reference-order task draws and with the device sampler; the CSV has the reference's columns, and under the script's seed
    the cached cells equal retrieval.n_shot_task_evaluation_cached called by hand."""
    import config
    from experiments import k_way_accuracy
    from voicemap_amd import models as VM, retrieval as R, utils as VU
    from voicemap_amd.librispeech import SyntheticSpeechDataset
    monkeypatch.setattr(config, "PATH", str(tmp_path))
    monkeypatch.setattr(k_way_accuracy, "PATH", str(tmp_path))
    os.makedirs(os.path.join(str(tmp_path), "logs"), exist_ok=True)
    torch.manual_seed(2)
    enc = VM.get_baseline_convolutional_encoder(16, 16, dropout=0.0, dtype="f32")
    net = VM.build_siamese_net(enc, (4000, 1))   # 1 s windows: the sweep, not the window length, is under test
    net.compile(loss="binary_crossentropy", optimizer="adam")
    path = os.path.join(str(tmp_path), "siamese.hdf5")
    net.save(path)
    args = ["--siamese", path, "--synthetic", "--k-way", "2", "5", "--n-shot", "1", "3", "--num-tasks", "16", "--distance", "cosine", "--cached", "--n-seconds", "1"]
    df = k_way_accuracy.main(args)       # (the script seeds np.random itself: experiments/_common.setup)
    assert list(df.columns) == ["method", "n_correct", "n_tasks", "n", "k"] and len(df) == 4
    assert ((df["n_correct"] >= 0) & (df["n_correct"] <= 16)).all()
    rows = open(os.path.join(str(tmp_path), "logs", "k-way_n-shot_accuracy_dev-clean_cosine.csv")).read().strip().splitlines()
    # (the finished DataFrame replaces the intermediate lines at the end, as in the reference :71-72: its header is the frame's)
    assert rows[0] == "method,n_correct,n_tasks,n,k" and len(rows) == 5
    # the same cells by hand, same seed
    valid = SyntheticSpeechDataset(num_speakers=40, files_per_speaker=12, seconds=1, stochastic=False, seed=1)
    pre = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
    loaded = VM.load_model(path)
    cache = R.embed_corpus(loaded, valid, pre)
    from experiments import _common as C
    C.seed_everything(0, 0)              # what setup() did before the script's first draw
    want = [R.n_shot_task_evaluation_cached(loaded, valid, pre, 16, n, k, "siamese", "cosine", cache=cache) for k in (2, 5) for n in (1, 3)]
    assert list(df["n_correct"]) == want
    args2 = ["--siamese", path, "--synthetic", "--k-way", "5", "--n-shot", "1", "--num-tasks", "16", "--distance", "cosine", "--cached",
             "--device-sampler", "--n-seconds", "1"]
    df2 = k_way_accuracy.main(args2)
    assert len(df2) == 1 and ((df2["n_correct"] >= 0) & (df2["n_correct"] <= 16)).all()
