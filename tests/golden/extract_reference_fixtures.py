#!/usr/bin/env python
"""Extract the *data* fixtures the reference tree holds for the siamese hot path.

Run once in the build container (needs /root/reference and h5py, i.e.
``/opt/conda/bin/python3.9 tests/golden/extract_reference_fixtures.py``).
Outputs (committed, data only -- no reference source text):

* ``ckpt_cfgCK_weights.npz``  -- the 28 weight arrays (+ the Adam iteration
  counter) of the one Keras HDF5 checkpoint the reference ships
  (models/n_seconds/siamese__nseconds_3.0__filters_32__embed_64__drop_0.05__r_0.hdf5),
  keyed ``<layer>/<name>``.
* ``ckpt_cfgCK_adam_slots.npz`` -- the optimizer state of the same file (``optimizer_weights/training/Adam/Variable*``):
  the 20 first-moment (``m/<layer>/<name>``) and 20 second-moment (``v/<layer>/<name>``) accumulators after its 11 000
  iterations, keyed like the weights (Keras ``trainable_weights`` order), plus ``vhat_max`` (the 20 amsgrad
  placeholders are all zero).  These and the BatchNorm moving statistics are the only numbers in the reference tree that
  its own forward / backward pass computed; tests/test_oracle_reference_pin.py holds the oracle to them.
* ``ckpt_cfgCK_meta.json``    -- layer geometry read from its ``model_config``
  attribute (kernel sizes, pool sizes, BN eps/momentum, head type) and the
  optimizer config from ``training_config``.
* ``clips_human_eval.npz``    -- query + 5 support clips (int16, 48000 frames,
  16 kHz mono) embedded as WAV in notebooks/Human_Evaluation.ipynb cell 8, with
  the recorded correct answer (5 -> support index 4).
* ``clips_embedding_vis.npz`` -- the 2 clips of
  notebooks/Embedding_Space_Visualisation.ipynb cell 27.
"""
import base64
import io
import json
import os
import re
import struct
import sys

import numpy as np

REF = os.environ.get("VOICEMAP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def wav_bytes_to_int16(b):
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos = 12
    fmt = None
    while pos < len(b):
        cid = b[pos:pos + 4]
        (sz,) = struct.unpack("<I", b[pos + 4:pos + 8])
        body = b[pos + 8:pos + 8 + sz]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            assert fmt is not None and fmt[0] == 1 and fmt[1] == 1 and fmt[5] == 16, fmt
            return np.frombuffer(body, dtype="<i2").copy(), fmt[2]
        pos += 8 + sz + (sz & 1)
    raise ValueError("no data chunk")


def audio_outputs(nb_path, cell_index):
    nb = json.load(open(nb_path))
    cell = nb["cells"][cell_index]
    clips = []
    for o in cell.get("outputs", []):
        html = o.get("data", {}).get("text/html")
        if html is None:
            continue
        html = "".join(html)
        m = re.search(r"data:audio/(?:x-)?wav;base64,([A-Za-z0-9+/=]+)", html)
        if m:
            pcm, sr = wav_bytes_to_int16(base64.b64decode(m.group(1)))
            clips.append((pcm, sr))
    return clips, cell


def main():
    import h5py

    path = os.path.join(REF, "models/n_seconds/siamese__nseconds_3.0__filters_32__embed_64__drop_0.05__r_0.hdf5")
    f = h5py.File(path, "r")
    weights = {}
    enc = f["model_weights/sequential_1/sequential_1"]
    for layer in enc:
        for name in enc[layer]:
            weights["%s/%s" % (layer, name.split(":")[0])] = np.asarray(enc[layer][name])
    head = f["model_weights/dense_2/dense_2"]
    for name in head:
        weights["dense_2/%s" % name.split(":")[0]] = np.asarray(head[name])
    weights["adam_iterations"] = np.asarray(f["optimizer_weights/Adam/iterations:0"])
    np.savez_compressed(os.path.join(OUT, "ckpt_cfgCK_weights.npz"), **weights)

    # Adam slots: Keras 2.2.2 Adam.get_updates stores self.weights = [iterations] + ms + vs + vhats, each list in
    # model.trainable_weights order = the order of model_weights' weight_names with the moving statistics dropped.
    order = []
    for grp in ("model_weights/sequential_1", "model_weights/dense_2"):
        for n in f[grp].attrs["weight_names"]:
            n = n.decode() if isinstance(n, bytes) else str(n)
            if "moving_" not in n:
                order.append((grp, n))
    assert len(order) == 20, order
    adam = f["optimizer_weights/training/Adam"]
    slot = lambda i: np.asarray(adam["Variable:0" if i == 0 else "Variable_%d:0" % i])
    slots = {}
    for i, (grp, n) in enumerate(order):
        key = "/".join(n.split(":")[0].split("/")[-2:])          # "conv1d_1/kernel", ..., "dense_2/bias": the keys of the weights file
        assert slot(i).shape == f[grp][n].shape == slot(20 + i).shape, (key, slot(i).shape)
        slots["m/" + key], slots["v/" + key] = slot(i), slot(20 + i)
    slots["vhat_max"] = np.array([float(np.abs(slot(40 + i)).max()) for i in range(20)])
    slots["order"] = np.array([k[2:] for k in slots if k.startswith("m/")])
    np.savez_compressed(os.path.join(OUT, "ckpt_cfgCK_adam_slots.npz"), **slots)

    mc = json.loads(f.attrs["model_config"])
    tc = json.loads(f.attrs["training_config"])
    layers = []
    head_cfg = []
    for l in mc["config"]["layers"]:
        if l["class_name"] == "Sequential":
            for s in l["config"]:
                c = s["config"]
                keep = {k: c[k] for k in ("name", "filters", "kernel_size", "padding", "activation", "strides",
                                          "pool_size", "epsilon", "momentum", "rate", "units", "use_bias",
                                          "data_format", "axis") if k in c}
                layers.append({"class_name": s["class_name"], **keep})
        elif l["class_name"] not in ("InputLayer",):
            c = l["config"]
            keep = {k: c[k] for k in ("name", "units", "activation", "function_type") if k in c}
            head_cfg.append({"class_name": l["class_name"], **keep})
    meta = {"keras_version": f.attrs["keras_version"].decode() if isinstance(f.attrs["keras_version"], bytes) else str(f.attrs["keras_version"]),
            "backend": f.attrs["backend"].decode() if isinstance(f.attrs["backend"], bytes) else str(f.attrs["backend"]),
            "input_shape": [12000, 1], "encoder_layers": layers, "head_layers": head_cfg,
            "training_config": tc}
    json.dump(meta, open(os.path.join(OUT, "ckpt_cfgCK_meta.json"), "w"), indent=1, sort_keys=True)

    clips, cell = audio_outputs(os.path.join(REF, "notebooks/Human_Evaluation.ipynb"), 8)
    assert len(clips) == 6, len(clips)
    text = "".join("".join(o["text"]) for o in cell["outputs"] if "text" in o)
    m = re.search(r"The correct answer was (\d+)", text)
    answer = int(m.group(1))
    names = re.findall(r"^(\d): (.+)$", text, flags=re.M)
    np.savez_compressed(os.path.join(OUT, "clips_human_eval.npz"),
                        query=clips[0][0], support=np.stack([c[0] for c in clips[1:]]),
                        sample_rate=np.int32(clips[0][1]), correct_answer_1based=np.int32(answer),
                        speaker_names=np.array([n for _, n in names]))
    clips2, _ = audio_outputs(os.path.join(REF, "notebooks/Embedding_Space_Visualisation.ipynb"), 27)
    assert len(clips2) == 2
    np.savez_compressed(os.path.join(OUT, "clips_embedding_vis.npz"),
                        clips=np.stack([c[0] for c in clips2]), sample_rate=np.int32(clips2[0][1]),
                        dataset_ids=np.array([11637, 15334]))
    print("answer", answer, names, [c[0].shape for c in clips], clips[0][1])
    for k, v in sorted(weights.items()):
        print(k, v.shape, float(np.abs(v).max()))


if __name__ == "__main__":
    sys.exit(main())
