"""Keras-2.2.2 HDF5 checkpoint interop without h5py: voicemap_amd/hdf5_lite.py (format) and
voicemap_amd/keras_hdf5.py (Keras' layout).  CPU only.

Pins: (1) a file written by libhdf5 itself in Keras' layout (tests/golden/keras_layout_h5py.hdf5, made by
tests/golden/make_h5py_fixture.py under the container's h5py) reads back exactly; (2) in the build container the
reference's own shipped checkpoint reads to the arrays that h5py extracted from it (tests/golden/ckpt_cfgCK_weights.npz);
(3) what this package writes is read identically by its own reader and -- where an h5py interpreter exists -- by libhdf5."""
import json
import os
import subprocess

import numpy as np
import pytest

from voicemap_amd import hdf5_lite as H
from voicemap_amd import keras_hdf5 as KH

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_CKPT = "/root/reference/models/n_seconds/siamese__nseconds_3.0__filters_32__embed_64__drop_0.05__r_0.hdf5"
H5PY_PYTHON = "/opt/conda/bin/python3.9"


def _random_state(kind, f=4, e=3, classes=5, seed=0):
    r = np.random.default_rng(seed)
    params = {}
    cin = 1
    for i, (k, mult) in enumerate([(32, 1), (3, 2), (3, 3), (3, 4)], 1):
        cout = mult * f
        params["conv%d.kernel" % i] = r.normal(size=(k, cin, cout)).astype(np.float32)
        params["conv%d.bias" % i] = r.normal(size=(cout,)).astype(np.float32)
        for s in KH.BN_SLOTS:
            params["bn%d.%s" % (i, s)] = (np.abs(r.normal(size=(cout,))) + 0.1).astype(np.float32)
        cin = cout
    params["dense.kernel"] = r.normal(size=(cin, e)).astype(np.float32)
    params["dense.bias"] = r.normal(size=(e,)).astype(np.float32)
    geo = {"filters": f, "embedding_dimension": e, "dropout": 0.05, "first_pool": 4, "input_shape": (800, 1), "classifier_units": 0}
    if kind == "siamese":
        geo["distance_metric"] = "uniform_euclidean"
        params["head.kernel"], params["head.bias"] = r.normal(size=(1, 1)).astype(np.float32), r.normal(size=(1,)).astype(np.float32)
    elif kind == "classifier":
        geo["classifier_units"] = classes
        params["head.kernel"] = r.normal(size=(e, classes)).astype(np.float32)
        params["head.bias"] = r.normal(size=(classes,)).astype(np.float32)
    names = KH.trainable_names(kind != "encoder")
    opt = {"config": {"lr": 0.001, "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7, "decay": 0.0, "amsgrad": False, "clipnorm": 1.0},
           "iterations": 123, "m": {n: (params[n] * 1e-3).astype(np.float32) for n in names},
           "v": {n: (params[n] ** 2 * 1e-4).astype(np.float32) for n in names}}
    return geo, params, opt, {"loss": "binary_crossentropy" if kind == "siamese" else "categorical_crossentropy", "metrics": ["accuracy"]}


def test_reader_on_a_file_written_by_libhdf5():
    ck = KH.read_checkpoint(os.path.join(GOLDEN, "keras_layout_h5py.hdf5"))
    exp = {k.replace("|", "/"): v for k, v in np.load(os.path.join(GOLDEN, "keras_layout_h5py_expected.npz")).items()}
    assert ck["kind"] == "classifier"
    g = ck["config"]
    assert (g["filters"], g["embedding_dimension"], g["classifier_units"], g["first_pool"], g["dropout"]) == (8, 3, 5, 4, 0.05)
    assert g["input_shape"] == (800, 1) and (g["bn_eps"], g["bn_momentum"]) == (0.001, 0.99)
    for i in range(1, 5):
        assert np.array_equal(ck["params"]["conv%d.kernel" % i], exp["conv1d_%d/kernel:0" % i])
        assert np.array_equal(ck["params"]["conv%d.bias" % i], exp["conv1d_%d/bias:0" % i])
        for s in KH.BN_SLOTS:
            assert np.array_equal(ck["params"]["bn%d.%s" % (i, s)], exp["batch_normalization_%d/%s:0" % (i, s)])
    assert np.array_equal(ck["params"]["dense.kernel"], exp["dense_1/kernel:0"])
    assert np.array_equal(ck["params"]["head.kernel"], exp["dense_2/kernel:0"]) and ck["params"]["head.kernel"].shape == (3, 5)
    names = KH.trainable_names(True)
    assert ck["optimizer"]["iterations"] == 37 and ck["optimizer"]["config"]["lr"] == 0.0005
    for k, n in enumerate(names):
        assert np.array_equal(ck["optimizer"]["m"][n], exp["optimizer/training/Adam/Variable%s:0" % ("" if k == 0 else "_%d" % k)])
        assert np.array_equal(ck["optimizer"]["v"][n], exp["optimizer/training/Adam/Variable_%d:0" % (len(names) + k)])
    assert ck["training"] == {"loss": "categorical_crossentropy", "metrics": ["accuracy"]}
    # weights-only view of the same file (model.load_weights)
    w = KH.read_weights(os.path.join(GOLDEN, "keras_layout_h5py.hdf5"))
    assert list(w) == list(ck["params"]) and all(np.array_equal(w[k], ck["params"][k]) for k in w)
    # the low-level API
    f = H.File(os.path.join(GOLDEN, "keras_layout_h5py.hdf5"))
    assert f.keys() == ["model_weights", "optimizer_weights"] and "nope" not in f
    d = f["model_weights/conv1d_2/conv1d_2/kernel:0"]
    assert d.shape == (3, 8, 16) and d.dtype == np.float32 and np.array_equal(d[()], exp["conv1d_2/kernel:0"])
    assert f["optimizer_weights/Adam/iterations:0"][()] == 37 and f["optimizer_weights/Adam/iterations:0"].shape == ()
    with pytest.raises(KeyError):
        f["model_weights/conv1d_9"]


@pytest.mark.skipif(not os.path.exists(REF_CKPT), reason="the reference tree is only mounted in the build container")
def test_reader_on_the_reference_checkpoint_equals_h5py_extraction():
    ck = KH.read_checkpoint(REF_CKPT)
    ref = np.load(os.path.join(GOLDEN, "ckpt_cfgCK_weights.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "ckpt_cfgCK_meta.json")))
    g = ck["config"]
    assert ck["kind"] == "siamese" and g["distance_metric"] == "weighted_l1"
    assert (g["filters"], g["embedding_dimension"], g["first_pool"], g["dropout"], g["input_shape"]) == (32, 128, 2, 0.05, (12000, 1))
    assert meta["backend"] == "tensorflow"
    for i in range(1, 5):
        assert np.array_equal(ck["params"]["conv%d.kernel" % i], ref["conv1d_%d/kernel" % i])
        assert np.array_equal(ck["params"]["conv%d.bias" % i], ref["conv1d_%d/bias" % i])
        for s in KH.BN_SLOTS:
            assert np.array_equal(ck["params"]["bn%d.%s" % (i, s)], ref["batch_normalization_%d/%s" % (i, s)])
    for a, b in (("dense", "dense_1"), ("head", "dense_2")):
        assert np.array_equal(ck["params"][a + ".kernel"], ref[b + "/kernel"]) and np.array_equal(ck["params"][a + ".bias"], ref[b + "/bias"])
    assert ck["optimizer"]["iterations"] == int(ref["adam_iterations"]) == 11000
    assert ck["optimizer"]["config"]["clipnorm"] == 1.0 and ck["training"]["loss"] == "binary_crossentropy"
    assert set(ck["optimizer"]["m"]) == set(KH.trainable_names(True))
    assert all(ck["optimizer"]["v"][n].shape == ck["params"][n].shape and (ck["optimizer"]["v"][n] >= 0).all() for n in ck["optimizer"]["v"])


@pytest.mark.parametrize("kind", ["siamese", "encoder", "classifier"])
def test_write_then_read_round_trip(tmp_path, kind):
    geo, params, opt, training = _random_state(kind)
    p = str(tmp_path / ("%s.hdf5" % kind))
    KH.write_checkpoint(p, kind, geo, params, opt, training)
    assert KH.is_hdf5(p)
    ck = KH.read_checkpoint(p)
    assert ck["kind"] == kind
    for k in ("filters", "embedding_dimension", "dropout", "first_pool", "input_shape", "classifier_units"):
        assert ck["config"][k] == geo[k], k
    if kind == "siamese":
        assert ck["config"]["distance_metric"] == "uniform_euclidean"
    assert list(ck["params"]) == list(params if kind != "encoder" else params)
    assert all(np.array_equal(ck["params"][k], params[k]) for k in params)
    assert ck["optimizer"]["iterations"] == 123 and ck["optimizer"]["config"] == opt["config"]
    assert all(np.array_equal(ck["optimizer"]["m"][k], opt["m"][k]) and np.array_equal(ck["optimizer"]["v"][k], opt["v"][k]) for k in opt["m"])
    assert ck["training"] == training
    # weights only (model.save_weights): no optimizer / training sections
    q = str(tmp_path / "w.h5")
    KH.write_checkpoint(q, kind, geo, params, None, None)
    f = H.File(q)
    assert "optimizer_weights" not in f and "training_config" not in f.attrs
    assert all(np.array_equal(v, params[k]) for k, v in KH.read_weights(q).items())


def test_hdf5_lite_tree_with_many_links_and_attribute_kinds(tmp_path):
    root = H.NodeSpec()
    root.attrs["text"] = "plain str"
    root.attrs["bytes"] = b"raw bytes"
    root.attrs["json"] = json.dumps({"k": list(range(2000))}).encode()
    root.attrs["ints"] = np.arange(5, dtype=np.int32)
    root.attrs["f64"] = np.float64(2.5)
    g = root.require_group("a/b")
    g.attrs["names"] = np.array([b"x", b"longer_name", b""])
    r = np.random.default_rng(1)
    want = {}
    for i in range(40):  # > 8 links: several symbol-table nodes under one B-tree node
        want["a/b/d%02d" % i] = r.normal(size=(i % 4 + 1, 3)).astype(np.float32 if i % 2 else np.float64)
        root.create_dataset("a/b/d%02d" % i, want["a/b/d%02d" % i])
    root.create_dataset("scalar", np.array(-7, dtype=np.int64))
    root.create_dataset("empty", np.zeros((0, 3), dtype=np.float32))
    root.create_dataset("u8", np.arange(10, dtype=np.uint8))
    p = str(tmp_path / "t.h5")
    H.write_file(p, root)
    f = H.File(p)
    assert f.keys() == ["a", "empty", "scalar", "u8"]
    assert f.attrs["text"] == b"plain str" and f.attrs["bytes"] == b"raw bytes" and f.attrs["f64"] == 2.5
    assert json.loads(f.attrs["json"].decode())["k"][-1] == 1999 and list(f.attrs["ints"]) == [0, 1, 2, 3, 4]
    assert list(f["a/b"].attrs["names"]) == [b"x", b"longer_name", b""]
    assert len(f["a/b"]) == 40 and all(np.array_equal(np.asarray(f[k]), v) and f[k].dtype == v.dtype for k, v in want.items())
    assert f["scalar"][()] == -7 and f["scalar"].shape == () and f["empty"].shape == (0, 3) and list(f["u8"][()]) == list(range(10))
    assert sorted(k for k, _ in f.visit_datasets()) == sorted(list(want) + ["scalar", "empty", "u8"])
    with pytest.raises(ValueError):
        big = H.NodeSpec()
        big.attrs["too_big"] = b"x" * 70000  # object-header messages are limited to 64 KB (Keras splits such attributes)
        H.write_file(str(tmp_path / "big.h5"), big)
    with pytest.raises(H.Hdf5FormatError):
        open(str(tmp_path / "junk.h5"), "wb").write(b"not hdf5" * 100)
        H.File(str(tmp_path / "junk.h5"))


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py on this machine")
def test_files_written_here_are_valid_for_libhdf5(tmp_path):
    geo, params, opt, training = _random_state("siamese", seed=3)
    p = str(tmp_path / "ours.hdf5")
    KH.write_checkpoint(p, "siamese", geo, params, opt, training)
    np.savez(str(tmp_path / "want.npz"), **{k.replace(".", "__"): v for k, v in params.items()})
    code = r'''
import sys, json, h5py, numpy as np
f = h5py.File(sys.argv[1], "r"); want = np.load(sys.argv[2])
assert json.loads(f.attrs["model_config"])["class_name"] == "Model" and f.attrs["keras_version"] == b"2.2.2"
mw = f["model_weights"]
assert [n.decode() for n in mw.attrs["layer_names"]] == ["input_1", "input_2", "sequential_1", "subtract_embeddings", "euclidean_distance", "dense_2"]
names = [n.decode() for n in mw["sequential_1"].attrs["weight_names"]]
assert len(names) == 26 and names[0] == "sequential_1/conv1d_1/kernel:0" and names[-1].endswith("batch_normalization_4/moving_variance:0")
assert np.array_equal(mw["sequential_1"][names[0]][()], want["conv1__kernel"])
assert np.array_equal(mw["sequential_1/sequential_1/batch_normalization_3/beta:0"][()], want["bn3__beta"])
assert np.array_equal(mw["dense_2/dense_2/kernel:0"][()], want["head__kernel"]) and len(mw["euclidean_distance"].attrs["weight_names"]) == 0
ow = f["optimizer_weights"]; wn = [n.decode() for n in ow.attrs["weight_names"]]
assert len(wn) == 61 and ow["Adam/iterations:0"][()] == 123 and ow[wn[-1]].shape == (1,)
assert np.allclose(ow[wn[1]][()], want["conv1__kernel"] * 1e-3)
print("LIBHDF5_OK")
'''
    out = subprocess.run([H5PY_PYTHON, "-c", code, p, str(tmp_path / "want.npz")], capture_output=True, text=True, timeout=120)
    assert "LIBHDF5_OK" in out.stdout, out.stderr[-2000:]


def _random_tree(seed):
    r = np.random.default_rng(seed)
    root = H.NodeSpec()
    want, attrs = {}, {}
    dtypes = [np.float32, np.float64, np.int32, np.int64, np.uint8, np.int16]

    def name():
        return "".join(r.choice(list("abcdefghijklmnopqrstuvwxyz_0123456789:"), size=int(r.integers(1, 14))))

    def fill(node, path, depth):
        for _ in range(int(r.integers(1, 12 if depth == 0 else 6))):
            nm = name()
            if nm in node.children:
                continue
            full = path + "/" + nm if path else nm
            if depth < 3 and r.random() < 0.35:
                g = node.require_group(nm)
                if r.random() < 0.5:
                    g.attrs["note"] = ("group " + full).encode()
                    attrs[full] = ("note", ("group " + full).encode())
                fill(g, full, depth + 1)
            else:
                shape = tuple(int(x) for x in r.integers(0, 5, size=int(r.integers(0, 4))))
                dt = dtypes[int(r.integers(len(dtypes)))]
                a = (r.normal(size=shape) * 50).astype(dt)
                d = node.create_dataset(nm, a)
                want[full] = a
                if r.random() < 0.4:
                    v = r.integers(-5, 5, size=int(r.integers(1, 6))).astype(np.int64)
                    d.attrs["tags"] = v
                    attrs[full] = ("tags", v)
    fill(root, "", 0)
    return root, want, attrs


@pytest.mark.parametrize("seed", range(8))
def test_hdf5_lite_random_trees_round_trip(tmp_path, seed):
    root, want, attrs = _random_tree(seed)
    p = str(tmp_path / "r.h5")
    H.write_file(p, root)
    f = H.File(p)
    got = dict(f.visit_datasets())
    assert sorted(got) == sorted(want)
    for k, a in want.items():
        b = np.asarray(got[k])
        assert b.shape == a.shape and b.dtype == a.dtype and np.array_equal(a, b), k
    for k, (an, av) in attrs.items():
        v = f[k].attrs[an]
        assert (v == av) if isinstance(av, bytes) else np.array_equal(v, av), k
    if os.path.exists(H5PY_PYTHON) and seed < 3:  # libhdf5 reads the same values
        np.savez(str(tmp_path / "want.npz"), **{k.replace("/", "|"): v for k, v in want.items()})
        code = r'''
import sys, h5py, numpy as np
f = h5py.File(sys.argv[1], "r"); want = np.load(sys.argv[2])
n = 0
for k in want.files:
    d = f[k.replace("|", "/")]
    assert d.shape == want[k].shape and d.dtype == want[k].dtype and np.array_equal(d[()], want[k]), k
    n += 1
print("LIBHDF5_OK", n)
'''
        out = subprocess.run([H5PY_PYTHON, "-c", code, p, str(tmp_path / "want.npz")], capture_output=True, text=True, timeout=120)
        assert "LIBHDF5_OK" in out.stdout, out.stderr[-2000:]
