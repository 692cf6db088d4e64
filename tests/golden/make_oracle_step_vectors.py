#!/usr/bin/env python
"""Golden vectors of the TRAINING step: what the CPU oracle (oracle/voicemap_oracle.py, float64) computes for
one and three ``train_on_batch`` calls -- per-block activations, embeddings, both losses, all 20 gradients, the weights after 1 and 3
Adam(clipnorm 1) steps and the BatchNorm moving statistics (TF-1.10 zero-debiased).  Written once, committed as data; the CPU test
``tests/test_oracle_golden.py::test_oracle_reproduces_committed_step_vectors`` re-runs the oracle against them, so a silent change of
the oracle's arithmetic (the BatchNorm moving-average form changed in round 2 and nothing committed would have noticed) fails a test,
and the GPU tests compare the HIP path with the same numbers.

  python tests/golden/make_oracle_step_vectors.py          # rewrites the two .npz files next to this script

* ``oracle_vectors_step_tiny.npz``  -- a 4-block encoder of the reference's shape (kernel 32 / 3 / 3 / 3, pools 4 / 2 / 2 / 2) with
  8-16-24-32 filters, embedding 16, 4 pairs of 2048 pre-processed samples, randomised gamma (some negative) / beta / biases;
  both script configurations: contrastive loss + uniform_euclidean head (experiments/siamese_contrastive_loss.py) and binary
  cross-entropy + weighted_l1 head (experiments/train_siamese.py:57).  float64 throughout.
* ``oracle_vectors_step_cfgCK.npz`` -- the reference's shipped checkpoint (tests/golden/ckpt_cfgCK_weights.npz: filters 32, embedding
  128, weighted_l1 head) on the 8 REAL LibriSpeech clips the reference tree holds (tests/golden/clips_*.npz) as 4 training pairs,
  decimated x4 and whitened per tower batch exactly as the scripts do; binary cross-entropy (the loss the checkpoint was trained with).
  Large arrays (gradients, weights) are stored as float32 to keep the fixture small; scalars and embeddings as float64.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import voicemap_oracle as O  # noqa: E402


def tiny_case():
    arch = O.EncoderArch([(32, 8, 4), (3, 16, 2), (3, 24, 2), (3, 32, 2)], 16, 0.0)
    r = np.random.default_rng(20260927)
    x1 = r.normal(0.0, 0.04, (4, 2048, 1))
    x2 = r.normal(0.0, 0.04, (4, 2048, 1))
    y = np.array([0.0, 1.0, 1.0, 0.0])[:, None]
    return arch, r, x1, x2, y


def randomise(p, r, names):
    """Non-trivial BatchNorm / bias state: gamma with 15 % negative entries, beta, conv and dense biases (what a trained net has)."""
    for k in names:
        if k.endswith(".gamma"):
            g = r.normal(1.0, 0.25, tuple(p[k].shape)) * np.where(r.random(tuple(p[k].shape)) < 0.15, -1.0, 1.0)
            p[k] = torch.tensor(g, dtype=p[k].dtype)
        elif k.endswith(".beta") or k.endswith(".bias"):
            p[k] = torch.tensor(r.normal(0.0, 0.2, tuple(p[k].shape)), dtype=p[k].dtype)
    return p


def run_steps(arch, p, x1, x2, y, loss, head, n_steps=3):
    """n_steps train_on_batch calls on the same batch; returns the first step's forward / gradient quantities and the parameters after
    steps 1 and n_steps."""
    state = O.AdamState()
    out = OrderedDict()
    a, b, yt = torch.tensor(x1), torch.tensor(x2), torch.tensor(y)
    bn = "fresh"
    cur = p
    for s in range(n_steps):
        st = O.siamese_train_step(arch, cur, state, a, b, yt, loss=loss, distance_metric=head, bn_state=bn)
        bn = st["bn_state"]
        cur = st["params"]
        if s == 0:
            out["loss"], out["acc"], out["grad_norm"] = st["loss"].numpy(), st["acc"].numpy(), st["grad_norm"].numpy()
            out["pred"], out["e1"], out["e2"] = st["pred"].numpy(), st["e1"].numpy(), st["e2"].numpy()
            for i in range(len(arch.blocks)):
                pooled = st["collect1"]["pooled"][i].numpy()
                out["t1_block%d_pooled_sum" % (i + 1)] = pooled.sum()
                out["t1_block%d_pooled_abssum" % (i + 1)] = np.abs(pooled).sum()
                out["t1_block%d_pooled_head" % (i + 1)] = pooled[0, :6, :4].copy()
                out["t1_block%d_bn_mean" % (i + 1)] = st["collect1"]["bn_mean"][i].numpy()
                out["t1_block%d_bn_var" % (i + 1)] = st["collect1"]["bn_var"][i].numpy()
            for k, g in st["grads"].items():
                out["grad/" + k] = g.numpy()
        if s in (0, n_steps - 1):
            for k, v in cur.items():
                out["params_after_%d/%s" % (s + 1, k)] = v.numpy()
    return out


def main():
    torch.set_num_threads(1)   # the float64 sums of a conv depend on the thread partition in the last bits: pin it for the record
    # ---- tiny ----
    arch, r, x1, x2, y = tiny_case()
    vec = OrderedDict(x1=x1, x2=x2, y=y)
    for loss, head in (("contrastive", "uniform_euclidean"), ("bce", "weighted_l1")):
        p = O.init_params(arch, head=head, seed=7)
        p = randomise(p, np.random.default_rng(11), O.param_names(arch))
        for k, v in p.items():
            vec["%s/params0/%s" % (loss, k)] = v.numpy()
        for k, v in run_steps(arch, p, x1, x2, y, loss, head).items():
            vec["%s/%s" % (loss, k)] = v
    np.savez_compressed(os.path.join(HERE, "oracle_vectors_step_tiny.npz"), **vec)
    # ---- cfg-CK on the reference's own clips ----
    w = np.load(os.path.join(HERE, "ckpt_cfgCK_weights.npz"))
    arch, p = O.params_from_checkpoint(w)
    h, v = np.load(os.path.join(HERE, "clips_human_eval.npz")), np.load(os.path.join(HERE, "clips_embedding_vis.npz"))
    f = lambda c: c.astype(np.float64) / 32768.0
    left = np.stack([f(h["query"]), f(h["support"][0]), f(h["support"][1]), f(v["clips"][0])])[:, :, None]
    right = np.stack([f(h["support"][4]), f(h["support"][2]), f(h["support"][3]), f(v["clips"][1])])[:, :, None]
    y = np.array([0.0, 1.0, 1.0, 1.0])[:, None]    # pair 0: the notebook's query and its recorded same-speaker answer
    pre = O.preprocess_instances(4)
    x1, x2 = pre(left), pre(right)                  # each tower whitened as its own batch (voicemap/utils.py:29-34)
    out = run_steps(arch, p, x1, x2, y, "bce", "weighted_l1")
    vec = OrderedDict(y=y, pair_clips=np.array(["query|support5", "support1|support3", "support2|support4", "vis1|vis2"]))
    for k, val in out.items():
        big = k.startswith("grad/") or k.startswith("params_after")
        vec[k] = np.asarray(val, dtype=np.float32) if big else val
    np.savez_compressed(os.path.join(HERE, "oracle_vectors_step_cfgCK.npz"), **vec)
    for name in ("oracle_vectors_step_tiny.npz", "oracle_vectors_step_cfgCK.npz"):
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
