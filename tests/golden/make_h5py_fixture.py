#!/usr/bin/env python
"""Write ``keras_layout_h5py.hdf5`` + ``keras_layout_h5py_expected.npz``: a tiny model (filters 8, embedding 3, classifier
with 5 classes) stored in Keras 2.2.2's HDF5 layout BY libhdf5 (h5py), with seeded random arrays -- the independent
fixture for voicemap_amd/hdf5_lite.py's reader on machines without h5py.  The h5py calls are the ones
keras/engine/saving.py makes (bytes attributes, arrays of bytes, ``create_dataset(name, shape, dtype)`` + assignment).

Run once in the build container:  /opt/conda/bin/python3.9 tests/golden/make_h5py_fixture.py
"""
import json
import os

import h5py
import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
F, E, C = 8, 3, 5
r = np.random.RandomState(1234)
expected = {}


def arr(*shape):
    return r.normal(0, 1, shape).astype(np.float32)


def layer(name, cls, **cfg):
    return {"class_name": cls, "config": dict(name=name, **cfg)}


layers, weights = [], []  # weights: (layer, [(weight name, array)])
cin = 1
for i, (k, mult, pool) in enumerate([(32, 1, 4), (3, 2, 2), (3, 3, 2), (3, 4, 2)], 1):
    cout = mult * F
    cfg = dict(filters=cout, kernel_size=[k], strides=[1], padding="same", activation="relu", use_bias=True)
    if i == 1:
        cfg["batch_input_shape"] = [None, 800, 1]
    layers.append(layer("conv1d_%d" % i, "Conv1D", **cfg))
    weights.append(("conv1d_%d" % i, [("conv1d_%d/kernel:0" % i, arr(k, cin, cout)), ("conv1d_%d/bias:0" % i, arr(cout))]))
    layers.append(layer("batch_normalization_%d" % i, "BatchNormalization", axis=-1, momentum=0.99, epsilon=0.001))
    weights.append(("batch_normalization_%d" % i, [("batch_normalization_%d/%s:0" % (i, s), np.abs(arr(cout)) + 0.1)
                                                   for s in ("gamma", "beta", "moving_mean", "moving_variance")]))
    layers.append(layer("spatial_dropout1d_%d" % i, "SpatialDropout1D", rate=0.05))
    weights.append(("spatial_dropout1d_%d" % i, []))
    layers.append(layer("max_pooling1d_%d" % i, "MaxPooling1D", pool_size=[pool], strides=[pool], padding="valid"))
    weights.append(("max_pooling1d_%d" % i, []))
    cin = cout
layers.append(layer("global_max_pooling1d_1", "GlobalMaxPooling1D"))
weights.append(("global_max_pooling1d_1", []))
layers.append(layer("dense_1", "Dense", units=E, activation="linear", use_bias=True))
weights.append(("dense_1", [("dense_1/kernel:0", arr(cin, E)), ("dense_1/bias:0", arr(E))]))
layers.append(layer("dense_2", "Dense", units=C, activation="softmax", use_bias=True))
weights.append(("dense_2", [("dense_2/kernel:0", arr(E, C)), ("dense_2/bias:0", arr(C))]))

path = os.path.join(OUT, "keras_layout_h5py.hdf5")
with h5py.File(path, "w") as f:
    f.attrs["keras_version"] = "2.2.2".encode("utf8")
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["model_config"] = json.dumps({"class_name": "Sequential", "config": layers}).encode("utf8")
    mw = f.create_group("model_weights")
    mw.attrs["layer_names"] = [n.encode("utf8") for n, _ in weights]
    mw.attrs["backend"] = "tensorflow".encode("utf8")
    mw.attrs["keras_version"] = "2.2.2".encode("utf8")
    for lname, ws in weights:
        g = mw.create_group(lname)
        g.attrs["weight_names"] = [n.encode("utf8") for n, _ in ws]
        for n, a in ws:
            d = g.create_dataset(n, a.shape, dtype=a.dtype)
            d[...] = a
            expected[n] = a
    f.attrs["training_config"] = json.dumps({
        "optimizer_config": {"class_name": "Adam", "config": {"lr": 0.0005, "beta_1": 0.9, "beta_2": 0.999, "decay": 0.0,
                                                              "epsilon": 1e-07, "amsgrad": False, "clipnorm": 1.0}},
        "loss": "categorical_crossentropy", "metrics": ["accuracy"], "sample_weight_mode": None, "loss_weights": None}).encode("utf8")
    ow = f.create_group("optimizer_weights")
    trainable = [a for _, ws in weights for n, a in ws if "moving_" not in n]
    names, vals = ["Adam/iterations:0"], [np.array(37, dtype=np.int64)]
    k = 0
    for slot in ("m", "v", "vhat"):
        for a in trainable:
            names.append("training/Adam/Variable%s:0" % ("" if k == 0 else "_%d" % k))
            vals.append(np.zeros((1,), np.float32) if slot == "vhat" else np.abs(arr(*a.shape)) * 1e-3)
            k += 1
    ow.attrs["weight_names"] = [n.encode("utf8") for n in names]
    for n, a in zip(names, vals):
        d = ow.create_dataset(n, a.shape, dtype=a.dtype)
        d[...] = a if a.shape else a[()]
        expected["optimizer/" + n] = a
np.savez(os.path.join(OUT, "keras_layout_h5py_expected.npz"), **{k.replace("/", "|"): v for k, v in expected.items()})
print("wrote", path, os.path.getsize(path), "bytes")
