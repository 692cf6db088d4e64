"""Keras-2.2.2 HDF5 checkpoint interop: the files the reference writes with ``ModelCheckpoint`` /
``model.save`` (experiments/train_siamese.py:74-80, siamese_contrastive_loss.py) and reads with
``keras.models.load_model`` (experiments/k_way_accuracy.py:45-46), through voicemap_amd/hdf5_lite.py.

Layout (keras/engine/saving.py of Keras 2.2.2, restated from the shipped checkpoint
models/n_seconds/siamese__nseconds_3.0__filters_32__embed_64__drop_0.05__r_0.hdf5):

  /                     attrs keras_version, backend, model_config (JSON), training_config (JSON)
  /model_weights        attrs layer_names, backend, keras_version
      /<layer>          attr weight_names = [b"<weight.name>", ...]   (one group per layer, also weight-less ones)
          /<weight.name>     e.g. sequential_1/conv1d_1/kernel:0 -- nested groups, float32, Keras' own array layouts
  /optimizer_weights    attr weight_names; Adam/iterations:0 (int64 scalar), training/Adam/Variable[_k]:0 =
                        [m_1..m_T, v_1..v_T, vhat_1..vhat_T] over the T trainable tensors in model order (vhat = zeros(1))

This module is pure data plumbing (numpy + hdf5_lite): ``read_checkpoint`` / ``write_checkpoint`` work without a GPU;
``voicemap_amd.models`` turns the result into a model (``load_model``) and collects a model's state for ``model.save``.
Parameter names on this side: conv{i}.kernel|bias, bn{i}.gamma|beta|moving_mean|moving_variance, dense.kernel|bias (the
embedding layer), head.kernel|bias (siamese Dense(1, sigmoid) or the classifier's Dense(num_classes, softmax)).
"""
from __future__ import annotations

import json
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np

from . import hdf5_lite as H

KERAS_VERSION = b"2.2.2"
BACKEND = b"tensorflow"
BN_SLOTS = ("gamma", "beta", "moving_mean", "moving_variance")


def _text(v) -> str:
    if isinstance(v, bytes):
        return v.decode("utf8")
    if isinstance(v, np.ndarray) and v.shape == ():
        return _text(v[()])
    return str(v)


def is_hdf5(path: str) -> bool:
    with open(path, "rb") as f:
        return f.read(8) == H.SIGNATURE


# =========================================================================================================
# reading
# =========================================================================================================
def _layer_list(config):
    """Sequential config: a list in Keras 2.2.2, {'name':..., 'layers': [...]} from 2.2.3 on."""
    return config["layers"] if isinstance(config, dict) else config


def _encoder_geometry(layers: List[dict]) -> dict:
    convs = [l for l in layers if l["class_name"] == "Conv1D"]
    bns = [l for l in layers if l["class_name"] == "BatchNormalization"]
    pools = [l for l in layers if l["class_name"] == "MaxPooling1D"]
    drops = [l for l in layers if l["class_name"] == "SpatialDropout1D"]
    denses = [l for l in layers if l["class_name"] == "Dense"]
    if len(convs) != 4 or len(bns) != 4 or len(pools) != 4 or not denses:
        raise ValueError("not the baseline convolutional encoder (voicemap/models.py:6-41): %d Conv1D, %d BN, %d pools"
                         % (len(convs), len(bns), len(pools)))
    f = convs[0]["config"]["filters"]
    kernels = [c["config"]["kernel_size"][0] for c in convs]
    chans = [c["config"]["filters"] for c in convs]
    if kernels != [32, 3, 3, 3] or chans != [f, 2 * f, 3 * f, 4 * f]:
        raise ValueError("unexpected conv geometry %s / %s" % (kernels, chans))
    pool = [p["config"]["pool_size"][0] for p in pools]
    if pool[1:] != [2, 2, 2]:
        raise ValueError("unexpected pool sizes %s" % pool)
    shape = convs[0]["config"].get("batch_input_shape")
    return {"filters": f, "embedding_dimension": denses[0]["config"]["units"], "first_pool": pool[0],
            "dropout": drops[0]["config"]["rate"] if drops else 0.0,
            "input_shape": tuple(shape[1:]) if shape else None,
            "bn_eps": bns[0]["config"]["epsilon"], "bn_momentum": bns[0]["config"]["momentum"],
            "conv_names": [c["config"]["name"] for c in convs], "bn_names": [b["config"]["name"] for b in bns],
            "dense_names": [d["config"]["name"] for d in denses],
            "dense_units": [d["config"]["units"] for d in denses],
            "dense_activations": [d["config"].get("activation") for d in denses]}


def _stored_arrays(mw) -> Dict[str, np.ndarray]:
    """Every array of a model_weights group keyed "<innermost layer>/<weight>" ("conv1d_1/kernel"), via weight_names."""
    arrays: Dict[str, np.ndarray] = {}
    for lname in [_text(x) for x in np.atleast_1d(mw.attrs["layer_names"])]:
        g = mw[lname]
        for wn in [_text(x) for x in np.atleast_1d(g.attrs.get("weight_names", []))]:
            parts = wn.split("/")
            arrays["%s/%s" % (parts[-2], parts[-1].split(":")[0])] = np.asarray(g[wn])
    return arrays


def read_weights(path: str) -> "OrderedDict[str, np.ndarray]":
    """``model.load_weights``: the arrays of a full-model file or of a ``save_weights`` file (whose root is the
    model_weights group), mapped to this side's names by layer order (conv1d_* -> conv1..4, batch_normalization_* ->
    bn1..4, the first Dense -> dense, a second Dense -> head)."""
    f = H.File(path)
    arrays = _stored_arrays(f["model_weights"] if "model_weights" in f else f)

    def ordered(prefix):
        ls = sorted({k.split("/")[0] for k in arrays if k.startswith(prefix)}, key=lambda n: int(n.rsplit("_", 1)[1]))
        return ls
    convs, bns, denses = ordered("conv1d_"), ordered("batch_normalization_"), ordered("dense_")
    if len(convs) != 4 or len(bns) != 4 or not denses:
        raise ValueError("%s does not hold the baseline encoder's weights" % path)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(4):
        out["conv%d.kernel" % (i + 1)], out["conv%d.bias" % (i + 1)] = arrays[convs[i] + "/kernel"], arrays[convs[i] + "/bias"]
        for sl in BN_SLOTS:
            out["bn%d.%s" % (i + 1, sl)] = arrays["%s/%s" % (bns[i], sl)]
    out["dense.kernel"], out["dense.bias"] = arrays[denses[0] + "/kernel"], arrays[denses[0] + "/bias"]
    if len(denses) > 1:
        out["head.kernel"], out["head.bias"] = arrays[denses[1] + "/kernel"], arrays[denses[1] + "/bias"]
    return out


def read_checkpoint(path: str) -> dict:
    """-> {"kind": "siamese" | "encoder" | "classifier", "config": geometry, "params": {name: array},
           "optimizer": None | {"config": {...}, "iterations": int, "m": {...}, "v": {...}},
           "training": None | {"loss":..., "metrics": [...]}}"""
    f = H.File(path)
    if "model_config" not in f.attrs:
        raise ValueError("%s has no model_config attribute (a weights-only file: build the model and use load_weights)" % path)
    mc = json.loads(_text(f.attrs["model_config"]))
    arrays = _stored_arrays(f["model_weights"])

    if mc["class_name"] == "Model":
        top = mc["config"]["layers"]
        seq = [l for l in top if l["class_name"] == "Sequential"]
        if len(seq) != 1:
            raise ValueError("expected one shared Sequential encoder in the siamese model")
        geo = _encoder_geometry(_layer_list(seq[0]["config"]))
        heads = [l for l in top if l["class_name"] == "Dense"]
        if len(heads) != 1 or heads[0]["config"]["units"] != 1:
            raise ValueError("expected a Dense(1) verification head (voicemap/models.py:58-69)")
        head_name = heads[0]["config"]["name"]
        inputs = [l for l in top if l["class_name"] == "InputLayer"]
        if inputs:
            geo["input_shape"] = tuple(inputs[0]["config"]["batch_input_shape"][1:])
        kind = "siamese"
        hk = arrays["%s/kernel" % head_name]
        # both implemented heads are Subtract -> Lambda -> Dense(1): |e1-e2| keeps E inputs, the euclidean norm has one
        geo["distance_metric"] = "uniform_euclidean" if hk.shape[0] == 1 and geo["embedding_dimension"] != 1 else "weighted_l1"
        geo["classifier_units"] = 0
    elif mc["class_name"] == "Sequential":
        geo = _encoder_geometry(_layer_list(mc["config"]))
        if len(geo["dense_names"]) == 2:
            kind, head_name = "classifier", geo["dense_names"][1]
            geo["classifier_units"] = geo["dense_units"][1]
        else:
            kind, head_name = "encoder", None
            geo["classifier_units"] = 0
    else:
        raise ValueError("unsupported model class %s" % mc["class_name"])

    params: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(4):
        params["conv%d.kernel" % (i + 1)] = arrays["%s/kernel" % geo["conv_names"][i]]
        params["conv%d.bias" % (i + 1)] = arrays["%s/bias" % geo["conv_names"][i]]
        for s in BN_SLOTS:
            params["bn%d.%s" % (i + 1, s)] = arrays["%s/%s" % (geo["bn_names"][i], s)]
    params["dense.kernel"] = arrays["%s/kernel" % geo["dense_names"][0]]
    params["dense.bias"] = arrays["%s/bias" % geo["dense_names"][0]]
    if head_name is not None:
        params["head.kernel"] = arrays["%s/kernel" % head_name]
        params["head.bias"] = arrays["%s/bias" % head_name]

    training, optimizer = None, None
    if "training_config" in f.attrs:
        tc = json.loads(_text(f.attrs["training_config"]))
        training = {"loss": tc.get("loss"), "metrics": tc.get("metrics") or []}
        oc = tc.get("optimizer_config") or {}
        optimizer = {"class_name": oc.get("class_name"), "config": oc.get("config", {}), "iterations": 0, "m": None, "v": None}
        if "optimizer_weights" in f:
            ow = f["optimizer_weights"]
            vals = [np.asarray(ow[_text(n)]) for n in np.atleast_1d(ow.attrs["weight_names"])]
            names = trainable_names(head_name is not None)
            T = len(names)
            if oc.get("class_name") == "Adam" and len(vals) >= 1 + 2 * T:
                optimizer["iterations"] = int(np.asarray(vals[0]).reshape(-1)[0])
                optimizer["m"] = OrderedDict(zip(names, vals[1:1 + T]))
                optimizer["v"] = OrderedDict(zip(names, vals[1 + T:1 + 2 * T]))
    geo["dtype"] = _text(f.attrs["voicemap_storage_dtype"]) if "voicemap_storage_dtype" in f.attrs else None
    return {"kind": kind, "config": geo, "params": params, "optimizer": optimizer, "training": training}


def trainable_names(with_head: bool) -> List[str]:
    """Keras ``model.trainable_weights`` order of these models (= the order of the optimizer slots)."""
    out = []
    for i in range(1, 5):
        out += ["conv%d.kernel" % i, "conv%d.bias" % i, "bn%d.gamma" % i, "bn%d.beta" % i]
    out += ["dense.kernel", "dense.bias"]
    if with_head:
        out += ["head.kernel", "head.bias"]
    return out


# =========================================================================================================
# writing
# =========================================================================================================
def _init(cls, **cfg):
    return {"class_name": cls, "config": cfg}


_GLOROT = _init("VarianceScaling", scale=1.0, mode="fan_avg", distribution="uniform", seed=None)


def _conv_cfg(name, filters, k, input_shape=None):
    c = OrderedDict(name=name, trainable=True)
    if input_shape is not None:
        c["batch_input_shape"] = [None] + list(input_shape)
        c["dtype"] = "float32"
    c.update(filters=filters, kernel_size=[k], strides=[1], padding="same", data_format="channels_last", dilation_rate=[1],
             activation="relu", use_bias=True, kernel_initializer=_GLOROT, bias_initializer=_init("Zeros"),
             kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
             bias_constraint=None)
    return {"class_name": "Conv1D", "config": c}


def _dense_cfg(name, units, activation):
    return {"class_name": "Dense", "config": OrderedDict(
        name=name, trainable=True, units=units, activation=activation, use_bias=True, kernel_initializer=_GLOROT,
        bias_initializer=_init("Zeros"), kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None,
        kernel_constraint=None, bias_constraint=None)}


def encoder_layer_configs(filters, embedding_dimension, dropout, first_pool, input_shape, classifier_units=0):
    """The Sequential config of get_baseline_convolutional_encoder (voicemap/models.py:6-41) in Keras' own vocabulary."""
    layers = []
    for i, (k, mult, pool) in enumerate([(32, 1, first_pool), (3, 2, 2), (3, 3, 2), (3, 4, 2)]):
        layers.append(_conv_cfg("conv1d_%d" % (i + 1), mult * filters, k, input_shape if i == 0 else None))
        layers.append({"class_name": "BatchNormalization", "config": OrderedDict(
            name="batch_normalization_%d" % (i + 1), trainable=True, axis=-1, momentum=0.99, epsilon=0.001, center=True,
            scale=True, beta_initializer=_init("Zeros"), gamma_initializer=_init("Ones"),
            moving_mean_initializer=_init("Zeros"), moving_variance_initializer=_init("Ones"), beta_regularizer=None,
            gamma_regularizer=None, beta_constraint=None, gamma_constraint=None)})
        layers.append({"class_name": "SpatialDropout1D", "config": OrderedDict(
            name="spatial_dropout1d_%d" % (i + 1), trainable=True, rate=dropout, noise_shape=None, seed=None)})
        layers.append({"class_name": "MaxPooling1D", "config": OrderedDict(
            name="max_pooling1d_%d" % (i + 1), trainable=True, pool_size=[pool], strides=[pool], padding="valid",
            data_format="channels_last")})
    layers.append({"class_name": "GlobalMaxPooling1D", "config": OrderedDict(name="global_max_pooling1d_1", trainable=True,
                                                                               data_format="channels_last")})
    layers.append(_dense_cfg("dense_1", embedding_dimension, "linear"))
    if classifier_units:
        layers.append(_dense_cfg("dense_2", classifier_units, "softmax"))
    return layers


def head_layer_names(distance_metric):
    """Names Keras gives the two weight-less head layers of build_siamese_net: auto-named for 'weighted_l1'
    (voicemap/models.py:58-59), explicit for 'uniform_euclidean' (voicemap/models.py:64-68)."""
    if distance_metric == "uniform_euclidean":
        return "subtract_embeddings", "euclidean_distance"
    return "subtract_1", "lambda_1"


def _model_config(kind, geo):
    enc = encoder_layer_configs(geo["filters"], geo["embedding_dimension"], geo["dropout"], geo["first_pool"],
                                geo["input_shape"], geo.get("classifier_units", 0))
    if kind != "siamese":
        return {"class_name": "Sequential", "config": enc}
    shape = [None] + list(geo["input_shape"])
    inp = lambda n: {"name": n, "class_name": "InputLayer", "inbound_nodes": [],
                     "config": {"batch_input_shape": shape, "dtype": "float32", "sparse": False, "name": n}}
    sub, lam = head_layer_names(geo["distance_metric"])
    layers = [inp("input_1"), inp("input_2"),
              {"name": "sequential_1", "class_name": "Sequential", "config": enc,
               "inbound_nodes": [[["input_1", 0, 0, {}]], [["input_2", 0, 0, {}]]]},
              {"name": sub, "class_name": "Subtract", "config": {"name": sub, "trainable": True},
               "inbound_nodes": [[["sequential_1", 1, 0, {}], ["sequential_1", 2, 0, {}]]]},
              # Keras stores a Lambda as marshalled Python-2 bytecode; this writer records WHICH of the reference's two lambdas
              # it is (voicemap/models.py:55-69) instead.  Consequence: keras.models.load_model cannot deserialize THIS layer
              # of a file written here (model_weights / optimizer_weights / training_config are plain Keras); rebuild the
              # model with build_siamese_net and model.load_weights(file), or use this package's load_model
              {"name": lam, "class_name": "Lambda", "inbound_nodes": [[[sub, 0, 0, {}]]],
               "config": {"name": lam, "trainable": True, "function": geo["distance_metric"],
                          "function_type": "voicemap_distance_metric", "output_shape": None, "output_shape_type": "raw",
                          "arguments": {}}},
              dict(_dense_cfg("dense_2", 1, "sigmoid"), name="dense_2", inbound_nodes=[[[lam, 0, 0, {}]]])]
    return {"class_name": "Model", "config": {"name": "model_1", "layers": layers,
                                              "input_layers": [["input_1", 0, 0], ["input_2", 0, 0]],
                                              "output_layers": [["dense_2", 0, 0]]}}


def write_checkpoint(path: str, kind: str, geo: dict, params: Dict[str, np.ndarray], optimizer: Optional[dict] = None,
                     training: Optional[dict] = None):
    """Inverse of ``read_checkpoint``.  ``geo`` needs filters, embedding_dimension, dropout, first_pool, input_shape
    (+ distance_metric for kind 'siamese', classifier_units for 'classifier'); ``optimizer`` = {"config": Adam config,
    "iterations": int, "m": {...}, "v": {...}} or None; ``training`` = {"loss": str, "metrics": [...]}."""
    assert kind in ("siamese", "encoder", "classifier")
    if geo.get("input_shape") is None:
        raise ValueError("input_shape is needed to write a Keras model_config")
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    root = H.NodeSpec()
    root.attrs["keras_version"], root.attrs["backend"] = KERAS_VERSION, BACKEND
    root.attrs["model_config"] = json.dumps(_model_config(kind, geo)).encode("utf8")
    if geo.get("dtype"):
        root.attrs["voicemap_storage_dtype"] = str(geo["dtype"]).encode("utf8")  # not Keras': this package's activation storage mode
    mw = root.require_group("model_weights")
    mw.attrs["backend"], mw.attrs["keras_version"] = BACKEND, KERAS_VERSION

    def encoder_weights(prefix):
        """[(weight name, array)] in Keras' order: trainable weights layer by layer, then the non-trainable ones."""
        tr, nt = [], []
        for i in range(1, 5):
            tr += [("%sconv1d_%d/kernel:0" % (prefix, i), params["conv%d.kernel" % i]),
                   ("%sconv1d_%d/bias:0" % (prefix, i), params["conv%d.bias" % i]),
                   ("%sbatch_normalization_%d/gamma:0" % (prefix, i), params["bn%d.gamma" % i]),
                   ("%sbatch_normalization_%d/beta:0" % (prefix, i), params["bn%d.beta" % i])]
            nt += [("%sbatch_normalization_%d/moving_mean:0" % (prefix, i), params["bn%d.moving_mean" % i]),
                   ("%sbatch_normalization_%d/moving_variance:0" % (prefix, i), params["bn%d.moving_variance" % i])]
        tr += [("%sdense_1/kernel:0" % prefix, params["dense.kernel"]), ("%sdense_1/bias:0" % prefix, params["dense.bias"])]
        return tr, nt

    def put(layer, weights):
        g = mw.require_group(layer)
        g.attrs["weight_names"] = np.array([n.encode("utf8") for n, _ in weights]) if weights else np.zeros((0,), dtype="S1")
        for n, a in weights:
            g.create_dataset(n, f32(a))

    if kind == "siamese":
        tr, nt = encoder_weights("sequential_1/")
        sub, lam = head_layer_names(geo.get("distance_metric"))
        layer_names = ["input_1", "input_2", "sequential_1", sub, lam, "dense_2"]
        put("input_1", [])
        put("input_2", [])
        put("sequential_1", tr + nt)
        put(sub, [])
        put(lam, [])
        put("dense_2", [("dense_2/kernel:0", params["head.kernel"]), ("dense_2/bias:0", params["head.bias"])])
    else:
        # a Sequential saved on its own: one group per layer, weights named "<layer>/<weight>:0"
        layer_names = [l["config"]["name"] for l in encoder_layer_configs(
            geo["filters"], geo["embedding_dimension"], geo["dropout"], geo["first_pool"], geo["input_shape"],
            geo.get("classifier_units", 0))]
        for ln in layer_names:
            put(ln, [])
        for i in range(1, 5):
            put("conv1d_%d" % i, [("conv1d_%d/kernel:0" % i, params["conv%d.kernel" % i]),
                                  ("conv1d_%d/bias:0" % i, params["conv%d.bias" % i])])
            put("batch_normalization_%d" % i, [("batch_normalization_%d/%s:0" % (i, s), params["bn%d.%s" % (i, s)])
                                               for s in BN_SLOTS])
        put("dense_1", [("dense_1/kernel:0", params["dense.kernel"]), ("dense_1/bias:0", params["dense.bias"])])
        if kind == "classifier":
            put("dense_2", [("dense_2/kernel:0", params["head.kernel"]), ("dense_2/bias:0", params["head.bias"])])
    mw.attrs["layer_names"] = np.array([n.encode("utf8") for n in layer_names])

    if training is not None:
        ocfg = dict((optimizer or {}).get("config") or {})
        tc = {"optimizer_config": {"class_name": "Adam", "config": ocfg}, "loss": training.get("loss"),
              "metrics": list(training.get("metrics") or []), "sample_weight_mode": None, "loss_weights": None}
        root.attrs["training_config"] = json.dumps(tc).encode("utf8")
        if optimizer is not None and optimizer.get("m") is not None:
            names = trainable_names(kind != "encoder")
            ow = root.require_group("optimizer_weights")
            wn = ["Adam/iterations:0"]
            ow.create_dataset("Adam/iterations:0", np.array(int(optimizer.get("iterations", 0)), dtype=np.int64))
            k = 0
            for slot in ("m", "v", None):
                for n in names:
                    nm = "training/Adam/Variable%s:0" % ("" if k == 0 else "_%d" % k)
                    ow.create_dataset(nm, f32(optimizer[slot][n]) if slot else np.zeros((1,), np.float32))
                    wn.append(nm)
                    k += 1
            ow.attrs["weight_names"] = np.array([n.encode("utf8") for n in wn])
    H.write_file(path, root)
