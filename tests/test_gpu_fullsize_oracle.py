"""-m gpu, slow (about a minute of host CPU): the bench's own batch -- BASELINE.json configs[1], cfg-A (filters 128, embedding 64),
128 pairs of 3 s @ 16 kHz from the synthetic generator of SURVEY 8(d) -- through the CPU oracle AT FULL SIZE, and every storage mode
of the HIP path against it.  This is the full-size check that is not HIP-vs-HIP: embeddings, loss and the
BatchNorm batch statistics against the float64 oracle forward, gradients against the oracle's fp32 autograd step (a float64
backward of 256 windows would need ~25 GB of saved activations; fp32 is what the reference itself computes in).

Reference arithmetic: voicemap/models.py:6-41 (encoder), :61-69 (head), voicemap/utils.py:77-101 (loss, whiten),
experiments/train_siamese.py:54-57.  Figures land in gpurun_out/parity_report.csv (kept as profiles/r03_parity_report.csv).

Bounds: the north star asks for embeddings within 1e-3 (relative) of the reference arithmetic.  'f32' / 'f32s' meet it by three
orders of magnitude, 'f16' (half storage, the bf16 kernels with 11 significand bits) meets it, 'bf16' does not (5e-3) and is held to
its own documented bound.  Gradient errors at this size are max-pool re-routing: direction (cosine) is asserted,
the per-tensor figures are reported.
"""
import os
import time

import numpy as np
import pytest
import torch

from oracle import voicemap_oracle as O
from tests.gpu_util import cosine, rel_err, report

pytestmark = pytest.mark.gpu

PAIRS, F, E = 128, 128, 64
EMB_TOL = {"f32": 1e-4, "f32s": 1e-4, "f16": 1e-3, "bf16": 3e-2}
GRAD_COS = {"f32": 0.9999, "f32s": 0.9999, "f16": 0.99, "bf16": 0.9}
# second state (VERDICT r3 weak #1a): the same batch through a net whose BatchNorm gamma (15 % negative), beta and conv / dense biases
# are those of a trained network, not of a fresh one -- the 16-bit modes are held to what they MEASURE there (bounds below; the
# figures go to the parity report and bench.py's precision object quotes them), the fp32-storage modes to the same 1e-4
# (measured: f16 1.06e-3, bf16 8.4e-3 -- with block 1's pool extreme stored centred, round 4; 1.3e-2 / 8.9e-2 before: the whitened
# waveform makes conv-1 outputs ~0.03, a bias of N(0, 0.2) is a pedestal seven times that)
# round 5: f16 holds the north star's 1e-3 here too -- block 2's tile is computed and stored centred on its per-channel pedestal
# (vm_conv_fwd_fold e_center; this state has channels whose pedestal is 10 x their spread: measured 1.06e-3 before, 0.66e-3 in the
# CPU emulation of the centred storage)
EMB_TOL_TRAINED = {"f32": 1e-4, "f32s": 1e-4, "f16": 1e-3, "bf16": 1.5e-2}
GRAD_COS_TRAINED = {"f32": 0.9999, "f32s": 0.9999, "f16": 0.99, "bf16": 0.95}


def _trained_like(p, seed=77):
    """gamma ~ N(1, 0.25) with 15 % of the channels negated, beta and every bias ~ N(0, 0.2): what _fold_arch_case of
    tests/test_gpu_fold.py does at 4 pairs, here on the bench batch."""
    r = np.random.default_rng(seed)
    q = {k: v.clone() for k, v in p.items()}
    for k in q:
        shape = tuple(q[k].shape)
        if k.endswith(".gamma"):
            q[k] = torch.tensor(r.normal(1.0, 0.25, shape) * np.where(r.random(shape) < 0.15, -1.0, 1.0), dtype=q[k].dtype)
        elif k.endswith(".beta") or (k.endswith(".bias") and not k.startswith("head")):
            q[k] = torch.tensor(r.normal(0.0, 0.2, shape), dtype=q[k].dtype)
    return q


@pytest.fixture(scope="module")
def oracle_full_size():
    """float64 forward (no autograd) + fp32 autograd step of the oracle on the bench batch; ~1 min on 16+ host cores."""
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, int(os.environ.get("VOICEMAP_TEST_ORACLE_THREADS", "32")))))
    try:
        arch = O.EncoderArch.baseline(F, E, dropout=0.0)
        p = O.init_params(arch, head="uniform_euclidean", seed=1234)
        x1, x2, y = O.synthetic_pairs(PAIRS, seed=1234)
        pre = O.preprocess_instances(4)
        a, b = torch.tensor(pre(x1.astype(np.float64))), torch.tensor(pre(x2.astype(np.float64)))
        t0 = time.time()
        c1, c2 = {}, {}
        with torch.no_grad():
            pred, e1, e2 = O.siamese_forward(arch, p, a, b, True, "uniform_euclidean", None, None, c1, c2)
            loss = O.contrastive_loss(torch.tensor(y, dtype=torch.float64), pred)
        stats = []
        for i in range(len(arch.blocks)):
            stats.append((torch.stack([c1["bn_mean"][i], c2["bn_mean"][i]]).numpy(), torch.stack([c1["bn_var"][i], c2["bn_var"][i]]).numpy()))
        del c1, c2
        t_fwd = time.time() - t0
        t0 = time.time()
        p32 = {k: v.float() for k, v in p.items()}
        step32 = O.siamese_train_step(arch, p32, None, a.float(), b.float(), torch.tensor(y), loss="contrastive")
        grads = {k: g.double().numpy() for k, g in step32["grads"].items()}
        e32 = np.concatenate([step32["e1"].numpy(), step32["e2"].numpy()])
        del step32
        report("full_size_oracle", "cpu_seconds_fp64_forward", t_fwd)
        report("full_size_oracle", "cpu_seconds_fp32_train_step", time.time() - t0)
        emb = np.concatenate([e1.numpy(), e2.numpy()])
        report("full_size_oracle", "oracle_fp32_vs_fp64_emb_rel_err", rel_err(e32, emb))
        # the trained-like state: the oracle's fp32 autograd step only (its embeddings are 1e-6 from the float64 forward, the figure
        # reported just above -- three orders under the bounds they are used for); saves the second float64 pass
        t0 = time.time()
        pt = _trained_like(p)
        st = O.siamese_train_step(arch, {k: v.float() for k, v in pt.items()}, None, a.float(), b.float(), torch.tensor(y), loss="contrastive")
        trained = {"p": pt, "emb": np.concatenate([st["e1"].numpy(), st["e2"].numpy()]).astype(np.float64), "loss": float(st["loss"]),
                   "grads": {k: g.double().numpy() for k, g in st["grads"].items()}}
        del st
        report("full_size_oracle", "cpu_seconds_fp32_train_step_trained_state", time.time() - t0)
        return {"arch": arch, "p": p, "x1": x1, "x2": x2, "y": y, "emb": emb, "loss": float(loss), "stats": stats, "grads": grads,
                "trained": trained}
    finally:
        torch.set_num_threads(threads)


@pytest.mark.parametrize("dtype", ["f32", "f32s", "f16", "bf16"])
def test_cfgA_full_batch_against_the_cpu_oracle(dtype, oracle_full_size):
    from voicemap_amd.engine import HipEncoderEngine
    o = oracle_full_size
    arch = o["arch"]
    eng = HipEncoderEngine(arch.blocks, E, dropout=0.0, head="uniform_euclidean", dtype=dtype)
    eng.set_params({k: v.numpy() for k, v in o["p"].items()})
    pl = eng.siamese_train_step(o["x1"], o["x2"], o["y"], loss="contrastive", preprocessed=False, downsampling=4, drop_masks=None,
                                apply_update=False)
    torch.cuda.synchronize()
    tag = "full_size_oracle[%s]" % dtype
    emb = pl["emb"].cpu().numpy()
    d_emb = rel_err(emb, o["emb"])
    report(tag, "emb_rel_err_vs_fp64_oracle", d_emb)
    report(tag, "emb_max_abs_err_over_max_abs", float(np.abs(emb - o["emb"]).max() / np.abs(o["emb"]).max()))
    loss = float(pl["loss_acc"][0].item())
    report(tag, "loss_abs_err_vs_fp64_oracle", abs(loss - o["loss"]))
    assert d_emb < EMB_TOL[dtype], (dtype, d_emb)
    assert abs(loss - o["loss"]) < max(EMB_TOL[dtype], 1e-5) * max(1.0, abs(o["loss"]))
    # BatchNorm batch statistics per tower (biased variance, voicemap/models.py:14-17 [3P])
    for i, (mean_ref, var_ref) in enumerate(o["stats"]):
        mean = pl[i]["mean"].cpu().numpy().astype(np.float64)
        var = 1.0 / pl[i]["invstd"].cpu().numpy().astype(np.float64) ** 2 - arch.bn_eps
        d_m = float(np.abs(mean - mean_ref).max() / max(np.sqrt(var_ref).max(), 1e-30))
        d_v = rel_err(var, var_ref)
        report(tag, "bn%d_mean_abs_err_over_max_std" % (i + 1), d_m)
        report(tag, "bn%d_var_rel_err" % (i + 1), d_v)
        tol = 10 * EMB_TOL[dtype]
        assert d_m < tol and d_v < tol, (dtype, i, d_m, d_v)
    # gradients vs the oracle's fp32 autograd step
    grads = eng.get_grads()
    assert all(np.isfinite(g).all() for g in grads.values())
    flat_h = np.concatenate([grads[k].ravel() for k in o["grads"]])
    flat_o = np.concatenate([o["grads"][k].ravel() for k in o["grads"]])
    cos = cosine(flat_h, flat_o)
    report(tag, "grad_cosine_vs_fp32_oracle", cos)
    report(tag, "grad_rel_err_vs_fp32_oracle", rel_err(flat_h, flat_o))
    for k, g in o["grads"].items():
        report(tag, "grad_rel_err[%s]" % k, rel_err(grads[k], g))
    assert cos > GRAD_COS[dtype], (dtype, cos)
    del eng, pl
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f32", "f32s", "f16", "bf16"])
def test_cfgA_full_batch_trained_like_batchnorm_state(dtype, oracle_full_size):
    """The bench batch in TRAINING mode through a network with non-trivial BatchNorm parameters (negative gammas included: the
    pool-extreme / folded-weight paths take their minimum branches) and biases: embeddings, loss and gradients against the oracle."""
    from voicemap_amd.engine import HipEncoderEngine
    o, t = oracle_full_size, oracle_full_size["trained"]
    eng = HipEncoderEngine(o["arch"].blocks, E, dropout=0.0, head="uniform_euclidean", dtype=dtype)
    eng.set_params({k: v.numpy() for k, v in t["p"].items()})
    pl = eng.siamese_train_step(o["x1"], o["x2"], o["y"], loss="contrastive", preprocessed=False, downsampling=4, drop_masks=None,
                                apply_update=False)
    torch.cuda.synchronize()
    tag = "full_size_oracle_trained_state[%s]" % dtype
    emb = pl["emb"].cpu().numpy()
    d_emb = rel_err(emb, t["emb"])
    report(tag, "emb_rel_err_vs_fp32_oracle", d_emb)
    report(tag, "emb_max_abs_err_over_max_abs", float(np.abs(emb - t["emb"]).max() / np.abs(t["emb"]).max()))
    loss = float(pl["loss_acc"][0].item())
    report(tag, "loss_abs_err_vs_fp32_oracle", abs(loss - t["loss"]))
    grads = eng.get_grads()
    assert all(np.isfinite(g).all() for g in grads.values())
    flat_h = np.concatenate([grads[k].ravel() for k in t["grads"]])
    flat_o = np.concatenate([t["grads"][k].ravel() for k in t["grads"]])
    cos = cosine(flat_h, flat_o)
    report(tag, "grad_cosine_vs_fp32_oracle", cos)
    report(tag, "grad_rel_err_vs_fp32_oracle", rel_err(flat_h, flat_o))
    for k, g in t["grads"].items():
        report(tag, "grad_rel_err[%s]" % k, rel_err(grads[k], g))
    assert d_emb < EMB_TOL_TRAINED[dtype], (dtype, d_emb)
    assert abs(loss - t["loss"]) < max(EMB_TOL_TRAINED[dtype], 1e-5) * max(1.0, abs(t["loss"]))
    assert cos > GRAD_COS_TRAINED[dtype], (dtype, cos)
    del eng, pl
    torch.cuda.empty_cache()
