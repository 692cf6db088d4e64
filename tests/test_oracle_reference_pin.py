"""Pins the CPU oracle to numbers the REFERENCE ITSELF COMPUTED (VERDICT r4 "What's missing" #1).

The reference cannot run here (Python 2 / Keras 2.2.2 / TF 1.10) and its tests hold no golden vectors for the
encoder, the losses, the gradients or the optimizer.  The only numbers in its tree that its own forward and backward pass
produced are inside the checkpoint it ships, models/n_seconds/siamese__nseconds_3.0__filters_32__embed_64__drop_0.05__r_0.hdf5
(11 000 Adam iterations of experiments/train_siamese.py's loop on LibriSpeech):

* the 4 x 2 BatchNormalization moving statistics (640 numbers) -- an average of what the reference's FORWARD pass saw in
  front of every BatchNorm layer: they pin the preprocessing (decimation, whitening scale), the convolution arithmetic
  (cross-correlation, bias, ReLU before BatchNorm), the use of batch statistics in training mode, and the pool geometry;
* the 20 Adam second-moment accumulators v (an average of the squared, CLIPPED gradient of its BACKWARD pass) and the
  first-moment accumulators m: they pin the global-norm clip, the scale of every layer's gradient through four
  BatchNorm backward passes, SpatialDropout1D's per-tower masks, and the structural zeros of the weighted-L1 head;
* a weight that only the optimizer's epsilon rule moved (dense_1/bias: its gradient is rounding noise).

The audio is the 8 LibriSpeech clips the reference's notebooks embed (tests/golden/clips_*.npz) -- not the training set, so
the comparisons are statistical: every bound below is a measured figure with the measured figure of each WRONG variant
beside it, and the test asserts both sides (the oracle inside, the variant outside), so it is known to discriminate.
Runs on the CPU in a few seconds.  tests/golden/extract_reference_fixtures.py wrote the fixtures from the checkpoint.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import voicemap_oracle as O


@pytest.fixture(scope="module")
def ck(golden_dir):
    w = np.load(f"{golden_dir}/ckpt_cfgCK_weights.npz")
    s = np.load(f"{golden_dir}/ckpt_cfgCK_adam_slots.npz")
    arch, p = O.params_from_checkpoint(w)
    names = O.param_names(arch, head="weighted_l1")
    assert len(names) == len(s["order"]) == 20
    m = {n: s["m/" + k].astype(np.float64) for n, k in zip(names, s["order"])}
    v = {n: s["v/" + k].astype(np.float64) for n, k in zip(names, s["order"])}
    for n in names:   # Keras trainable_weights order == the oracle's param_names order, shape by shape
        assert m[n].shape == v[n].shape == tuple(p[n].shape), n
    h, e = np.load(f"{golden_dir}/clips_human_eval.npz"), np.load(f"{golden_dir}/clips_embedding_vis.npz")
    f = lambda c: c.astype(np.float64) / 32768.0
    clips = np.concatenate([f(h["query"])[None], f(h["support"]), f(e["clips"])])[:, :, None]     # (8, 48000, 1)
    # the pairs of tests/golden/make_oracle_step_vectors.py: (query, support 5) is the same speaker ("The correct answer was 5")
    left = clips[[0, 1, 2, 6]]
    right = clips[[5, 3, 4, 7]]
    y = torch.tensor([[0.0], [1.0], [1.0], [1.0]], dtype=torch.float64)
    return dict(w=w, arch=arch, p=p, names=names, m=m, v=v, vhat_max=s["vhat_max"], clips=clips, left=left, right=right, y=y,
                iterations=int(w["adam_iterations"]))


# ---------------------------------------------------------------------------------------------
# forward pass: BatchNorm moving statistics
# ---------------------------------------------------------------------------------------------


def _batch_statistics(ck, first_pool=2, relu_before_bn=True, pad32=None, decimate=4, whiten="batch", flip_kernel=False,
                      use_bias=True, train_mode_bn=True):
    """Training-mode forward of the encoder over the 8 clips, composed from the oracle's own layer functions, with ONE
    switch per reference rule so that a wrong variant is a one-word change.  Returns [(mean, var)] in front of each BatchNorm.
    With the defaults it is O.encoder_forward (asserted in the test)."""
    p, arch = ck["p"], ck["arch"]
    x = ck["clips"][:, ::decimate, :]                                           # voicemap/utils.py:29
    if whiten == "batch":
        x = O.whiten(x)                                                         # utils.py:88-101: ONE scale for the batch
    elif whiten == "per_sample":
        x = (x - x.mean(axis=1, keepdims=True)) * (0.038021 / np.sqrt(np.power(x, 2).mean(axis=(1, 2), keepdims=True)))
    h = torch.tensor(x)
    out = []
    for i, (k, c, pool) in enumerate(arch.blocks):
        pool = first_pool if i == 0 else pool
        kern, bias = p[f"conv{i+1}.kernel"], p[f"conv{i+1}.bias"]
        if flip_kernel:
            kern = kern.flip(0)
        if not use_bias:
            bias = torch.zeros_like(bias)
        pl, pr = O.same_padding(k) if (pad32 is None or k != 32) else pad32
        z = F.conv1d(F.pad(h.transpose(1, 2), (pl, pr)), kern.permute(2, 1, 0), bias).transpose(1, 2)
        if relu_before_bn:
            z = torch.relu(z)                                                   # models.py:13 activation='relu' inside Conv1D
        out.append((z.mean(dim=(0, 1)).numpy(), z.var(dim=(0, 1), unbiased=False).numpy()))
        g, b = p[f"bn{i+1}.gamma"], p[f"bn{i+1}.beta"]
        if train_mode_bn:
            yv, _, _ = O.batchnorm_train(z, g, b, arch.bn_eps)
        else:
            yv = O.batchnorm_infer(z, g, b, p[f"bn{i+1}.moving_mean"], p[f"bn{i+1}.moving_variance"], arch.bn_eps)
        if not relu_before_bn:
            yv = torch.relu(yv)
        h = O.maxpool1d(yv, pool)
    return out


def _agreement(ck, stats):
    """Per BatchNorm layer: correlation of the batch means with the checkpoint's moving means over the channels, correlation
    of the log variances, and the median ratio of each."""
    rows = []
    for i, (mean, var) in enumerate(stats):
        mm = ck["w"][f"batch_normalization_{i+1}/moving_mean"].astype(np.float64)
        mv = ck["w"][f"batch_normalization_{i+1}/moving_variance"].astype(np.float64)
        var = np.maximum(var, 1e-30)                                           # a dead channel of a wrong variant
        rows.append((np.corrcoef(mean, mm)[0, 1], np.corrcoef(np.log(var), np.log(mv))[0, 1],
                     float(np.median(mean / mm)), float(np.median(var / mv))))
    return np.array(rows)


def _pinned(a):
    return bool((a[:, 0] >= 0.9).all() and (a[:, 1] >= 0.9).all() and (np.abs(np.log(a[:, 2:])) <= math.log(1.25)).all())


def test_batch_statistics_on_reference_clips_match_the_checkpoint_moving_statistics(ck):
    """Measured (oracle rules): mean correlations 0.991 / 0.976 / 0.915 / 0.938, log-variance correlations 0.975 / 0.967 /
    0.965 / 0.944, median ratios 0.92 ... 1.04 -- on 8 clips that are not the training set."""
    stats = _batch_statistics(ck)
    # the composition above IS the oracle's encoder: same statistics from O.encoder_forward
    col = {}
    O.encoder_forward(ck["arch"], ck["p"], torch.tensor(O.preprocess_instances(4)(ck["clips"])), True, None, col)
    for (mean, var), om, ov in zip(stats, col["bn_mean"], col["bn_var"]):
        assert np.array_equal(mean, om.numpy()) and np.array_equal(var, ov.numpy())
    a = _agreement(ck, stats)
    assert _pinned(a), a
    assert a[:, 0].min() > 0.91 and a[:, 1].min() > 0.94 and 0.9 < a[:, 2:].min() and a[:, 2:].max() < 1.06, a


@pytest.mark.parametrize("variant,kw,worst_corr_below,note", [
    ("current models.py first pool 4 (the checkpoint was trained with 2)", dict(first_pool=4), 0.75, "blocks 2-4: 0.54-0.73"),
    ("ReLU after BatchNorm instead of inside Conv1D", dict(relu_before_bn=False), 0.0, "log-variance correlation goes NEGATIVE"),
    ("no whitening", dict(whiten=None), 0.7, "block-1 variance ratio 9.1"),
    ("no decimation (16 kHz into the net)", dict(decimate=1), 0.7, "block-1 variance ratio 1.5"),
    ("true convolution (kernel flipped in time) instead of cross-correlation", dict(flip_kernel=True), 0.8,
     "weak: speech statistics are nearly time-symmetric; only block 1's log-variance correlation drops, 0.975 -> 0.795"),
    ("convolution bias dropped", dict(use_bias=False), 0.0, "block-1 mean correlation -0.17"),
    ("BatchNorm normalising with the MOVING statistics in training mode", dict(train_mode_bn=False), 0.65, "blocks 3-4"),
])
def test_wrong_forward_variants_do_not_match_the_moving_statistics(ck, variant, kw, worst_corr_below, note):
    a = _agreement(ck, _batch_statistics(ck, **kw))
    assert not _pinned(a), (variant, a)
    assert min(a[:, 0].min(), a[:, 1].min()) < worst_corr_below, (variant, note, a)


def test_what_the_moving_statistics_do_not_discriminate(ck):
    """Stated so nobody reads more into the pin than it holds: a 16/15 instead of TensorFlow's 15/16 SAME split of the
    32-tap kernel shifts the output by one sample (statistics unchanged to 3 digits), and a PER-SAMPLE whitening scale
    instead of the reference's one-scalar-per-batch (utils.py:98) is invisible on 8 peak-normalised clips of similar level.
    Those two rules rest on the reference's source text (utils.py:94-99) and TensorFlow's documented SAME rule."""
    for kw in (dict(pad32=(16, 15)), dict(whiten="per_sample")):
        assert _pinned(_agreement(ck, _batch_statistics(ck, **kw)))


# ---------------------------------------------------------------------------------------------
# backward pass + optimizer: Adam accumulators
# ---------------------------------------------------------------------------------------------


def _step(ck, masks_seed=None, arch=None, loss="bce", **kw):
    arch = arch or ck["arch"]
    pre = O.preprocess_instances(4)
    m1 = m2 = None
    if masks_seed is not None:   # SpatialDropout1D(0.05): keep-mask (N, 1, C), drawn independently for each encoder call
        r = np.random.default_rng(masks_seed)
        m1 = [torch.tensor((r.random((4, 1, c)) >= arch.dropout).astype(np.float64)) for (_, c, _) in arch.blocks]
        m2 = [torch.tensor((r.random((4, 1, c)) >= arch.dropout).astype(np.float64)) for (_, c, _) in arch.blocks]
    return O.siamese_train_step(arch, ck["p"], None, torch.tensor(pre(ck["left"])), torch.tensor(pre(ck["right"])), ck["y"],
                                loss=loss, distance_metric="weighted_l1", drop_masks1=m1, drop_masks2=m2, **kw)


def _clipped(grads, clipnorm=1.0):
    n = float(O.global_norm(grads))
    c = clipnorm / n if n >= clipnorm else 1.0
    return {k: g.numpy() * c for k, g in grads.items()}, n


def test_adam_second_moments_sum_to_the_clip_norm(ck):
    """sum over ALL 20 tensors of v = 0.99852 after 11 000 iterations (1 - 0.999^11000 = 1 - 1.7e-5): v averages the squared
    CLIPPED gradient, so its total is E[min(1, |g|^2)] -- it sits just under 1.0 iff (a) the clip is ONE global norm over all
    gradients (standalone Keras 2.2.2 `clip_norm(g, clipnorm, norm)` with norm over the whole list; tf.keras clips per tensor),
    (b) clipnorm is 1, (c) almost every batch's raw gradient norm exceeded it.  The oracle: raw norm 5.6 on these clips,
    clipped total exactly 1; per-tensor clipping of the same gradient would total 3.3."""
    total_v = sum(x.sum() for x in ck["v"].values())
    assert 0.995 < total_v <= 1.0, total_v
    out = _step(ck)
    g, n = _clipped(out["grads"])
    assert n > 3.0
    assert abs(sum((x ** 2).sum() for x in g.values()) - 1.0) < 1e-12
    # the oracle's own optimizer applies exactly this clip: the m it leaves after one step is (1 - beta_1) * g_clipped
    st = O.AdamState()
    O.adam_step(st, {k: ck["p"][k].clone() for k in ck["names"]}, out["grads"])
    assert abs(sum(float((st.m[k] ** 2).sum()) for k in ck["names"]) - 0.1 ** 2) < 1e-12
    per_tensor = sum(min(1.0, float((x ** 2).sum())) for x in out["grads"].values())
    assert per_tensor > 3.0, per_tensor                     # what per-tensor clipping would have accumulated: not ~1
    assert float(ck["vhat_max"].max()) == 0.0              # amsgrad=False: the 20 vhat placeholders stayed zero


def _log_profile(ck, mean_sq, skip=("dense.bias",)):
    return np.array([math.log10(max(mean_sq[n], 1e-300) / ck["v"][n].mean()) for n in ck["names"] if n not in skip])


def _mean_sq(ck, seeds, **kw):
    """E[g_clipped^2] per tensor (mean over elements and over dropout-mask draws): what Adam's v estimates."""
    acc = {n: 0.0 for n in ck["names"]}
    for seed in seeds:
        g, _ = _clipped(_step(ck, masks_seed=seed, **kw)["grads"])
        for n in ck["names"]:
            acc[n] += float((g[n] ** 2).mean()) / len(seeds)
    return acc


def test_gradient_profile_matches_adam_second_moments_over_four_decades(ck):
    """mean(v) per tensor runs from 2.1e-7 (conv4 bias) to 5.1e-3 (conv1 bias, head kernel): 4.4 decades.  The oracle's clipped
    BCE gradient on 4 pairs of reference clips, with SpatialDropout1D(0.05) masks as the reference trained (mean square over
    8 mask draws), is within a factor 10^0.37 = 2.3 of mean(v) for ALL 19 tensors that have a gradient, 16 of them within
    1.45 (log10 rms 0.14; a single draw: rms 0.16-0.32).  What this pins: every layer's backward scale through four
    BatchNorm-in-training-mode backward passes, the max-pool routing, the weighted-L1 head and the BCE gradient.  A BatchNorm
    backward that treats the batch mean / variance as constants puts 12 of the 19 tensors further off than the oracle's
    worst one (log10 rms 0.65, worst 12 x; asserted below)."""
    lp = _log_profile(ck, _mean_sq(ck, range(8)))
    assert len(lp) == 19 and np.abs(lp).max() < 0.45 and np.sqrt((lp ** 2).mean()) < 0.18, lp
    assert (np.abs(lp) < 0.17).sum() >= 16, lp
    orig = O.batchnorm_train

    def detached(z, gamma, beta, eps):
        mean, var = z.mean(dim=(0, 1)).detach(), z.var(dim=(0, 1), unbiased=False).detach()
        inv = gamma * torch.rsqrt(var + eps)
        return z * inv + (beta - mean * inv), mean, var
    O.batchnorm_train = detached
    try:
        lp = _log_profile(ck, _mean_sq(ck, range(2)))
    finally:
        O.batchnorm_train = orig
    assert np.abs(lp).max() > 1.0 and np.sqrt((lp ** 2).mean()) > 0.55, lp       # measured 1.08 / 0.65 against 0.37 / 0.14


def test_structural_zeros_of_the_weighted_l1_head(ck):
    """|e1 - e2| cancels anything added to BOTH embeddings.  dense_1/bias: gradient identically 0 -> the checkpoint's v is
    1e-20 (rounding noise of fp32 backprop; every other tensor's is >= 2e-7) and the oracle's gradient is exactly 0.
    batch_normalization_4/beta shifts a channel's global max in both towers alike -> 0 as well WITHOUT dropout; the
    checkpoint's v there is 5.5e-7 (37 x below gamma's): only SpatialDropout1D dropping a channel in ONE tower breaks the
    cancellation.  With independent per-tower masks the oracle's mean square over 8 draws is 0.98 x that v -- so the masks are
    drawn per encoder call, whole channels at a time."""
    v = ck["v"]
    assert v["dense.bias"].max() < 1e-18 and min(v[n].mean() for n in ck["names"] if n != "dense.bias") > 1e-7
    assert np.abs(ck["m"]["dense.bias"]).max() < 1e-9
    plain = _step(ck)["grads"]
    assert float(plain["dense.bias"].abs().max()) == 0.0 and float(plain["bn4.beta"].abs().max()) == 0.0
    assert 1e-7 < v["bn4.beta"].mean() < 0.1 * v["bn4.gamma"].mean()
    ms = _mean_sq(ck, range(8))
    assert ms["dense.bias"] == 0.0
    assert 0.7 < ms["bn4.beta"] / v["bn4.beta"].mean() < 1.4, ms["bn4.beta"]      # measured 0.98 (single draws 0.14 ... 2.2)
    # the same mask in both towers would keep the cancellation: the reference's masks are NOT shared between the towers
    r = np.random.default_rng(0)
    m = [torch.tensor((r.random((4, 1, c)) >= 0.05).astype(np.float64)) for (_, c, _) in ck["arch"].blocks]
    pre = O.preprocess_instances(4)
    shared = O.siamese_train_step(ck["arch"], ck["p"], None, torch.tensor(pre(ck["left"])), torch.tensor(pre(ck["right"])), ck["y"],
                                  loss="bce", distance_metric="weighted_l1", drop_masks1=m, drop_masks2=m)["grads"]
    assert float(shared["bn4.beta"].abs().max()) < 1e-12


def test_adam_epsilon_rule_from_the_drift_of_a_gradient_free_bias(ck):
    """dense_1/bias starts at 0 (Keras `zeros`) and its gradient is rounding noise (v ~ 1e-20, sqrt(v) ~ 1e-10 << epsilon), so
    all that moved it in 11 000 iterations is lr_t * m / (sqrt(v) + EPSILON): the checkpoint's values have rms 9.36e-5.
    Driving the ORACLE's adam_step for 11 000 iterations with noise of the checkpoint's own v gives rms 0.9e-4 ... 1.3e-4 with
    Keras' K.epsilon() = 1e-7 OUTSIDE the square root; 1e-8 (torch / TF default) gives 1e-3, no epsilon 0.1, epsilon inside
    the root (sqrt(v + 1e-7) = 3e-4) 3e-8 -- each an order of magnitude or more away."""
    obs = float(np.sqrt((ck["w"]["dense_1/bias"].astype(np.float64) ** 2).mean()))
    assert 8e-5 < obs < 1.1e-4
    sig = torch.tensor(np.sqrt(ck["v"]["dense.bias"]))

    def drift(eps, steps=ck["iterations"]):
        g = torch.Generator().manual_seed(0)
        st = O.AdamState(epsilon=eps, clipnorm=None)
        p = {"b": torch.zeros(128, dtype=torch.float64)}
        for _ in range(steps):
            p = O.adam_step(st, p, {"b": torch.randn(128, generator=g, dtype=torch.float64) * sig})
        return float(torch.sqrt((p["b"] ** 2).mean()))
    got = drift(O.KERAS_EPSILON)
    assert obs / 1.6 < got < obs * 1.6, (got, obs)
    assert drift(1e-8) > 5 * obs and drift(1e-6) < obs / 5
