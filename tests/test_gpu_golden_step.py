"""-m gpu: the HIP training step against the COMMITTED golden vectors of the oracle (tests/golden/oracle_vectors_step_*.npz, written
by tests/golden/make_oracle_step_vectors.py; tests/test_oracle_golden.py holds the oracle itself to them on the CPU).

* cfg-CK: the reference's shipped checkpoint (filters 32, embedding 128, first pool 2, weighted_l1 head) on the 8 REAL LibriSpeech
  clips the reference tree holds, as ONE training-mode batch of 4 pairs (VERDICT r3 weak #1b): raw int16 clips -> decimate x4 +
  whiten per tower on the GPU -> twin forward with batch statistics -> binary cross-entropy -> all 20 gradients -> three
  Adam(clipnorm 1) steps with the zero-debiased moving statistics.  Reference arithmetic: voicemap/models.py:6-81,
  voicemap/utils.py:22-34, 88-101, experiments/train_siamese.py:54-57.
* tiny: a 4-block encoder with randomised BatchNorm parameters (negative gammas), both script configurations.

Tolerances are stated per storage mode below; measured figures go to the parity report (gpurun_out/parity_report.csv)."""
import os

import numpy as np
import pytest
import torch

from oracle import voicemap_oracle as O
from tests.gpu_util import cosine, max_err, rel_err, report

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# embeddings (relative 2-norm), loss (absolute), whole-gradient cosine, parameters after 1 / 3 steps.  An Adam step moves every
# parameter by ~lr = 1e-3 whatever its gradient's size (m / sqrt(v) is +-1 on the first step), so the parameter error is NOT
# proportional to the gradient error: an element whose gradient is of the order of Adam's epsilon (1e-7; the trained checkpoint has such
# elements) turns a 1e-8 gradient difference into a tenth of a step, and a sign flip of a tiny gradient costs 2 lr per step.  Hence two
# bounds per step: ``p?`` on the 99.9 % quantile of |parameter error| (the bulk), ``p?max`` on the worst element.
TOL = {"f32": dict(emb=1e-4, loss=1e-5, cos=0.99999, p1=2e-5, p1max=3e-4, p3=1e-4, p3max=1e-3),
       "f32s": dict(emb=1e-4, loss=1e-5, cos=0.9999, p1=5e-5, p1max=5e-4, p3=2e-4, p3max=1.5e-3),
       "f16": dict(emb=1e-3, loss=1e-3, cos=0.98, p1=2.1e-3, p1max=2.1e-3, p3=6.1e-3, p3max=6.1e-3),
       "bf16": dict(emb=2e-2, loss=1e-2, cos=0.9, p1=2.1e-3, p1max=2.1e-3, p3=6.1e-3, p3max=6.1e-3)}


def _check(tag, dt, eng, step_fn, g, prefix, names, tol=None):
    """Run three steps (``step_fn()`` -> plan of one forward/backward without update), compare with the golden arrays under ``prefix``."""
    t = dict(TOL[dt], **(tol or {}))
    pl = step_fn()
    torch.cuda.synchronize()
    emb = pl["emb"].cpu().numpy()
    e_ref = np.concatenate([g[prefix + "e1"], g[prefix + "e2"]])
    d_emb = rel_err(emb, e_ref)
    report(tag, "emb_rel_err", d_emb)
    loss = float(pl["loss_acc"][0].item())
    report(tag, "loss_abs_err", abs(loss - float(g[prefix + "loss"])))
    grads = eng.get_grads()
    flat_h = np.concatenate([np.asarray(grads[k], dtype=np.float64).ravel() for k in names])
    flat_o = np.concatenate([g[prefix + "grad/" + k].astype(np.float64).ravel() for k in names])
    cos = cosine(flat_h, flat_o)
    report(tag, "grad_cosine", cos)
    report(tag, "grad_rel_err", rel_err(flat_h, flat_o))
    worst, worst_k = 0.0, ""
    for k in names:
        e = rel_err(grads[k], g[prefix + "grad/" + k])
        report(tag, "grad_rel_err[%s]" % k, e)
        if e > worst:
            worst, worst_k = e, k
    report(tag, "worst_grad_rel_err_tensor[%s]" % worst_k, worst)
    assert d_emb < t["emb"], (tag, d_emb)
    assert abs(loss - float(g[prefix + "loss"])) < t["loss"] * max(1.0, abs(float(g[prefix + "loss"])))
    assert cos > t["cos"], (tag, cos)
    if dt == "f32":
        for k in names:
            ref = g[prefix + "grad/" + k].astype(np.float64)
            assert np.abs(np.asarray(grads[k], dtype=np.float64) - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, k
    # three optimizer steps on the same batch: Adam slots, lr_t, global-norm clip, refreshed weight copies, moving statistics
    for s in (1, 2, 3):
        if s > 1:
            step_fn()
        eng.optimizer_step()
        if s in (1, 3):
            torch.cuda.synchronize()
            got = eng.get_params()
            worst_m, errs = 0.0, []
            for k in got:
                ref = g[prefix + "params_after_%d/%s" % (s, k)].astype(np.float64)
                if "moving" in k:
                    worst_m = max(worst_m, rel_err(got[k], ref))
                else:
                    errs.append(np.abs(np.asarray(got[k], dtype=np.float64) - ref).ravel())
            errs = np.concatenate(errs)
            worst_p, bulk_p = float(errs.max()), float(np.quantile(errs, 0.999))
            report(tag, "params_after_%d_max_abs_err" % s, worst_p)
            report(tag, "params_after_%d_q999_abs_err" % s, bulk_p)
            report(tag, "moving_stats_after_%d_rel_err" % s, worst_m)
            assert bulk_p < t["p%d" % s] and worst_p < t["p%dmax" % s], (tag, s, bulk_p, worst_p)
            assert worst_m < 10 * t["emb"], (tag, s, worst_m)
    assert eng.iterations == 3


@pytest.mark.parametrize("dt", ["f32", "f32s", "f16", "bf16"])
def test_cfgCK_training_step_on_the_reference_clips(dt):
    from voicemap_amd.engine import HipEncoderEngine
    g = np.load(os.path.join(GOLDEN, "oracle_vectors_step_cfgCK.npz"))
    arch, p = O.params_from_checkpoint(np.load(os.path.join(GOLDEN, "ckpt_cfgCK_weights.npz")))
    h, v = np.load(os.path.join(GOLDEN, "clips_human_eval.npz")), np.load(os.path.join(GOLDEN, "clips_embedding_vis.npz"))
    f = lambda c: c.astype(np.float32) / np.float32(32768.0)
    left = np.stack([f(h["query"]), f(h["support"][0]), f(h["support"][1]), f(v["clips"][0])])
    right = np.stack([f(h["support"][4]), f(h["support"][2]), f(h["support"][3]), f(v["clips"][1])])
    eng = HipEncoderEngine(arch.blocks, arch.embedding_dimension, dropout=0.0, head="weighted_l1", dtype=dt)
    eng.set_params({k: val.numpy() for k, val in p.items()})
    y = g["y"]
    step = lambda: eng.siamese_train_step(left, right, y, loss="bce", preprocessed=False, downsampling=4, drop_masks=None, apply_update=False)
    _check("golden_step_cfgCK_real_clips[%s]" % dt, dt, eng, step, g, "", O.param_names(arch, head="weighted_l1"))


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("loss,head", [("contrastive", "uniform_euclidean"), ("bce", "weighted_l1")])
def test_tiny_training_step_against_golden_vectors(dt, loss, head):
    from voicemap_amd.engine import HipEncoderEngine
    g = np.load(os.path.join(GOLDEN, "oracle_vectors_step_tiny.npz"))
    arch = O.EncoderArch([(32, 8, 4), (3, 16, 2), (3, 24, 2), (3, 32, 2)], 16, 0.0)
    names = O.param_names(arch, head=head)
    p = {k[len(loss) + 9:]: g[k] for k in g.files if k.startswith(loss + "/params0/")}
    eng = HipEncoderEngine(arch.blocks, arch.embedding_dimension, dropout=0.0, head=head, dtype=dt)
    eng.set_params(p)
    x1, x2, y = g["x1"], g["x2"], g["y"]
    step = lambda: eng.siamese_train_step(x1, x2, y, loss=loss, drop_masks=None, apply_update=False)
    # 8..32 channels: the folded / packed kernels do not serve these widths (the pass-per-block path runs, its block-1 extreme is
    # rounded twice) and so few channels average little: half storage measures 4.1e-3 here against 4e-4..1e-3 at the real widths
    _check("golden_step_tiny[%s-%s]" % (dt, loss), dt, eng, step, g, loss + "/", names,
           tol=dict(emb=8e-3, loss=2e-2, cos=0.97) if dt == "f16" else None)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_hip_batch_statistics_match_the_reference_checkpoints_moving_statistics(dt):
    """The PRODUCT against numbers the reference itself computed, with no oracle in between (cf. tests/test_oracle_reference_pin.py,
    which holds the oracle to the same figures): the HIP encoder in training mode over the 8 LibriSpeech clips of the reference's
    notebooks (raw int16 -> decimate x4 + whiten on the GPU, one encoder call of 8 windows) -- its four BatchNorm layers' batch
    means / variances against the moving means / variances in the reference's shipped checkpoint: per-layer correlation >= 0.9
    (means, log variances), median ratios within 1.25.  Oracle rules measured 0.92-0.99 / 0.92-1.04; the current models.py pool
    geometry (first pool 4) 0.54-0.73."""
    from voicemap_amd.engine import HipEncoderEngine
    w = np.load(os.path.join(GOLDEN, "ckpt_cfgCK_weights.npz"))
    arch, p = O.params_from_checkpoint(w)
    h, v = np.load(os.path.join(GOLDEN, "clips_human_eval.npz")), np.load(os.path.join(GOLDEN, "clips_embedding_vis.npz"))
    clips = np.concatenate([h["query"][None], h["support"], v["clips"]]).astype(np.int16)          # (8, 48000)
    rows = []
    for blocks in (arch.blocks, [(32, 32, 4)] + list(arch.blocks[1:])):
        eng = HipEncoderEngine(blocks, arch.embedding_dimension, dropout=0.0, head=None, dtype=dt)
        eng.set_params({k: t.numpy() for k, t in p.items() if not k.startswith("head.")})
        pl = eng.plan(8, 12000, True)
        eng.preprocess(pl, torch.as_tensor(clips).to("cuda"), 4, True, 8)
        eng.forward(pl, 8, None)
        torch.cuda.synchronize()
        agree = []
        for i in range(4):
            mean = pl[i]["mean"][0].cpu().numpy().astype(np.float64)
            var = np.maximum(1.0 / pl[i]["invstd"][0].cpu().numpy().astype(np.float64) ** 2 - arch.bn_eps, 1e-30)
            mm = w[f"batch_normalization_{i+1}/moving_mean"].astype(np.float64)
            mv = w[f"batch_normalization_{i+1}/moving_variance"].astype(np.float64)
            agree.append((np.corrcoef(mean, mm)[0, 1], np.corrcoef(np.log(var), np.log(mv))[0, 1], np.median(mean / mm), np.median(var / mv)))
        rows.append(np.array(agree))
        del eng, pl
    good, pool4 = rows
    tag = "hip_vs_checkpoint_moving_statistics[%s]" % dt
    for i in range(4):
        report(tag, "bn%d_mean_correlation" % (i + 1), good[i, 0])
        report(tag, "bn%d_logvar_correlation" % (i + 1), good[i, 1])
        report(tag, "bn%d_median_var_ratio" % (i + 1), good[i, 3])
    assert (good[:, :2] >= 0.9).all() and (np.abs(np.log(good[:, 2:])) <= np.log(1.25)).all(), good
    assert pool4[1:, :2].min() < 0.8, pool4           # the wrong geometry does NOT match: the check discriminates


def test_hip_gradient_profile_matches_the_reference_checkpoints_adam_slots():
    """... and the backward pass: the HIP path's clipped BCE gradient on 4 pairs of the reference's clips, SpatialDropout1D(0.05) masks
    drawn per tower (mean square over 8 draws), against mean(v) of the 20 Adam second-moment accumulators in the reference's
    checkpoint (2e-7 ... 5e-3, 4.4 decades): every tensor within 10^0.45 (the oracle: 10^0.37), dense_1/bias -- a structural zero of
    the weighted-L1 head -- at rounding noise.  Same clips, masks and bounds as tests/test_oracle_reference_pin.py."""
    from voicemap_amd.engine import HipEncoderEngine
    w = np.load(os.path.join(GOLDEN, "ckpt_cfgCK_weights.npz"))
    s = np.load(os.path.join(GOLDEN, "ckpt_cfgCK_adam_slots.npz"))
    arch, p = O.params_from_checkpoint(w)
    names = O.param_names(arch, head="weighted_l1")
    v = {n: s["v/" + k].astype(np.float64) for n, k in zip(names, s["order"])}
    h, e = np.load(os.path.join(GOLDEN, "clips_human_eval.npz")), np.load(os.path.join(GOLDEN, "clips_embedding_vis.npz"))
    clips = np.concatenate([h["query"][None], h["support"], e["clips"]]).astype(np.float32) / 32768.0
    left, right = clips[[0, 1, 2, 6]][:, :, None], clips[[5, 3, 4, 7]][:, :, None]
    y = np.array([[0.0], [1.0], [1.0], [1.0]], dtype=np.float32)
    eng = HipEncoderEngine(arch.blocks, arch.embedding_dimension, dropout=arch.dropout, head="weighted_l1", dtype="f32")
    eng.set_params({k: t.numpy() for k, t in p.items()})
    acc = {n: 0.0 for n in names}
    for seed in range(8):
        r = np.random.default_rng(seed)       # the draws of tests/test_oracle_reference_pin.py::_step
        m1 = [(r.random((4, 1, c)) >= arch.dropout) for (_, c, _) in arch.blocks]
        m2 = [(r.random((4, 1, c)) >= arch.dropout) for (_, c, _) in arch.blocks]
        dm = [torch.as_tensor(np.concatenate([a[:, 0, :], b[:, 0, :]], 0).astype(np.float32) / (1.0 - arch.dropout)).to("cuda") for a, b in zip(m1, m2)]
        eng.siamese_train_step(left, right, y, loss="bce", preprocessed=False, downsampling=4, drop_masks=dm, apply_update=False)
        torch.cuda.synchronize()
        g = eng.get_grads()
        norm = np.sqrt(sum(float((np.asarray(g[n], dtype=np.float64) ** 2).sum()) for n in names))
        assert norm > 3.0                     # the clip is active, as it was on almost every batch of the reference's run (sum v = 0.9985)
        for n in names:
            acc[n] += float(((np.asarray(g[n], dtype=np.float64) / norm) ** 2).mean()) / 8
    lp = np.array([np.log10(max(acc[n], 1e-300) / v[n].mean()) for n in names if n != "dense.bias"])
    report("hip_vs_checkpoint_adam_slots[f32]", "max_abs_log10_ratio", float(np.abs(lp).max()))
    report("hip_vs_checkpoint_adam_slots[f32]", "rms_log10_ratio", float(np.sqrt((lp ** 2).mean())))
    assert len(lp) == 19 and np.abs(lp).max() < 0.45 and np.sqrt((lp ** 2).mean()) < 0.18, lp
    assert acc["dense.bias"] < 1e-12 * acc["dense.kernel"]
