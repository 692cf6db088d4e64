"""-m gpu, about a minute of host CPU: the 1e-3 claim of the benchmarked mode (f16 storage) held against more than one state.

The north star asks for embeddings within 1e-3 (relative) of the reference arithmetic (voicemap/models.py:6-41 is fp32 throughout).
tests/test_gpu_fullsize_oracle.py checks ONE seed per state; round 5 measured 7.0e-4 / 6.8e-4 there -- a thin margin.  Here the same
comparison -- the bench's batch size (128 pairs of 3 s @ 16 kHz, cfg-A), training-mode forward, HIP f16 against the CPU oracle's fp32
forward (1e-6 from its float64 forward at this size: full_size_oracle,oracle_fp32_vs_fp64_emb_rel_err) -- over
  * five seeds (weights AND batch) of the fresh-init state,
  * five seeds of the trained-like BatchNorm / bias state (gamma ~ N(1, 0.25), 15 % negative; beta, biases ~ N(0, 0.2)),
  * a TRAINED state: tests/golden/trained_cfgA_state.npz, 1 200 Adam steps of this repository's own training loop on synthetic
    speakers (tests/golden/make_trained_state.py), on noise windows and on windows of synthetic speakers it was not trained on,
and the maximum must stay under 1e-3.  The spread goes to the parity report (bench.py's precision.against_oracle quotes it)."""
import os

import numpy as np
import pytest
import torch

from oracle import voicemap_oracle as O
from tests.gpu_util import rel_err, report
from tests.test_gpu_fullsize_oracle import _trained_like

pytestmark = pytest.mark.gpu

PAIRS, F, E = 128, 128, 64
TOL = 1e-3
_seen = {}


def _oracle_fp32_embeddings(arch, p, x1, x2):
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, int(os.environ.get("VOICEMAP_TEST_ORACLE_THREADS", "32")))))
    try:
        pre = O.preprocess_instances(4)
        a, b = torch.tensor(pre(x1.astype(np.float64))).float(), torch.tensor(pre(x2.astype(np.float64))).float()
        with torch.no_grad():
            _, e1, e2 = O.siamese_forward(arch, {k: v.float() for k, v in p.items()}, a, b, True, "uniform_euclidean", None, None, {}, {})
        return np.concatenate([e1.numpy(), e2.numpy()]).astype(np.float64)
    finally:
        torch.set_num_threads(threads)


def _hip_f16_embeddings(arch, p, x1, x2, y):
    from voicemap_amd.engine import HipEncoderEngine
    eng = HipEncoderEngine(arch.blocks, E, dropout=0.0, head="uniform_euclidean", dtype="f16")
    eng.set_params({k: v.numpy() for k, v in p.items() if "moving" not in k})
    pl = eng.siamese_train_step(x1, x2, y, loss="contrastive", preprocessed=False, downsampling=4, drop_masks=None, apply_update=False)
    torch.cuda.synchronize()
    emb = pl["emb"].cpu().numpy()
    del eng, pl
    torch.cuda.empty_cache()
    return emb


def _check(state, tag, arch, p, x1, x2, y):
    ref = _oracle_fp32_embeddings(arch, p, x1, x2)
    d = rel_err(_hip_f16_embeddings(arch, p, x1, x2, y), ref)
    report("f16_guard[%s]" % state, "emb_rel_err_vs_fp32_oracle[%s]" % tag, d)
    _seen.setdefault(state, []).append(d)
    assert d < TOL, (state, tag, d)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15])
@pytest.mark.parametrize("state", ["fresh_init", "trained_like"])
def test_f16_embeddings_within_1e3_over_seeds(state, seed):
    arch = O.EncoderArch.baseline(F, E, dropout=0.0)
    p = O.init_params(arch, head="uniform_euclidean", seed=seed)
    if state == "trained_like":
        p = _trained_like(p, seed=100 + seed)
    x1, x2, y = O.synthetic_pairs(PAIRS, seed=seed)
    _check(state, "seed%d" % seed, arch, p, x1, x2, y)


def _trained_params(golden_dir):
    z = np.load(os.path.join(golden_dir, "trained_cfgA_state.npz"))
    return {k: torch.tensor(z[k].astype(np.float64)) for k in z.files if not k.startswith("__")}


@pytest.mark.parametrize("inputs", ["noise_windows", "synthetic_speakers"])
def test_f16_embeddings_within_1e3_on_a_trained_state(inputs, golden_dir):
    arch = O.EncoderArch.baseline(F, E, dropout=0.0)
    p = _trained_params(golden_dir)
    if inputs == "noise_windows":
        x1, x2, y = O.synthetic_pairs(PAIRS, seed=31)
    else:
        from voicemap_amd.librispeech import SyntheticSpeechDataset
        ds = SyntheticSpeechDataset(num_speakers=48, files_per_speaker=8, seconds=3, seed=9, subset="guard")   # speakers the state never saw
        np.random.seed(5)
        ([x1, x2], y) = ds.build_verification_batch(PAIRS)
        x1, x2, y = np.asarray(x1, dtype=np.float32), np.asarray(x2, dtype=np.float32), np.asarray(y, dtype=np.float32).reshape(-1, 1)
    _check("trained_1200_steps", inputs, arch, p, x1, x2, y)


def test_f16_guard_spread_is_reported():
    """(runs last in this file) the spread per state, for the parity report and bench.py."""
    if not _seen:
        pytest.skip("the guard cases did not run in this session")
    for state, v in _seen.items():
        report("f16_guard[%s]" % state, "emb_rel_err_min", float(np.min(v)))
        report("f16_guard[%s]" % state, "emb_rel_err_max", float(np.max(v)))
        report("f16_guard[%s]" % state, "cases", float(len(v)))
        assert np.max(v) < TOL
