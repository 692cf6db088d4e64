"""CPU checks of the log-mel specification (voicemap_amd/spectro.py) and its oracle (oracle.logmel_features, encoder2d_*): the
variant is not in the reference, so what can be pinned without a GPU is internal consistency -- the DFT-basis GEMM
the HIP kernel evaluates equals numpy.fft on the same frames, the filterbank has the stated shape, the 2-D oracle's hand-derived
pieces agree with autograd / finite differences."""
import numpy as np
import torch

from oracle import voicemap_oracle as O
from voicemap_amd import spectro as S


def test_frames_and_filterbank_shape():
    assert S.n_frames(48000) == 298 and S.n_frames(399) == 0 and S.n_frames(400) == 1 and S.n_frames(560) == 2
    w = S.mel_filterbank(64)
    assert w.shape == (256, 64) and w.dtype == np.float32 and w.min() >= 0 and w.max() <= 1.0
    assert np.all(w[0] == 0)                                  # DC gets no weight (fmin = 0 is the first filter's lower edge)
    peaks = w.argmax(0)
    assert np.all(np.diff(peaks) > 0)                         # one band after the other
    assert np.all(w.sum(1)[1:250] > 0)                        # no spectral hole between the first and the last filter
    edges = S.mel_to_hz(np.linspace(S.hz_to_mel(0.0), S.hz_to_mel(8000.0), 66))
    assert abs(edges[-1] - 8000.0) < 1e-6 and abs(S.mel_to_hz(S.hz_to_mel(1234.5)) - 1234.5) < 1e-9
    b = S.dft_basis()
    assert b.shape == (400, 512) and abs(b[0]).max() == 0.0   # periodic Hann starts at 0
    assert np.allclose(b[200, :256], S.hann_periodic(400)[200] * np.cos(2 * np.pi * np.arange(256) * 200 / 512), atol=1e-6)


def test_basis_gemm_equals_fft_logmel():
    r = np.random.default_rng(0)
    raw = r.normal(0, 0.05, (3, 5000)) + 0.1 * np.sin(np.arange(5000) * 0.2)[None, :]
    ref = O.logmel_features(raw)
    T = S.n_frames(5000)
    idx = np.arange(T)[:, None] * S.HOP + np.arange(S.WIN_LENGTH)[None, :]
    d = raw[:, idx] @ S.dft_basis().astype(np.float64)
    got = np.log((d[..., :256] ** 2 + d[..., 256:] ** 2) @ S.mel_filterbank().astype(np.float64) + S.LOG_FLOOR)
    assert ref.shape == (3, T, 64) and np.abs(got - ref).max() < 1e-5
    # a pure tone lands in the band whose triangle covers it
    tone = np.sin(2 * np.pi * 1000.0 * np.arange(4000) / 16000.0)[None, :]
    band = O.logmel_features(tone)[0].mean(0).argmax()
    w = S.mel_filterbank()
    assert w[32, band] > 0.3                                  # bin 32 = 1000 Hz


def test_encoder2d_oracle_gradients_finite_difference():
    arch = O.Encoder2dArch(4, 4, dropout=0.0)
    p = O.init_params2d(arch, seed=2)
    r = np.random.default_rng(1)
    f1, f2 = torch.tensor(r.normal(0, 1, (2, 16, 16))), torch.tensor(r.normal(0, 1, (2, 16, 16)))
    y = torch.tensor([[0.0], [1.0]])
    res = O.siamese2d_train_step(arch, p, None, f1, f2, y)
    for name in ("conv2.kernel", "bn3.gamma", "dense.kernel"):
        g = res["grads"][name]
        idx = tuple(int(v) for v in np.unravel_index(int(g.abs().argmax()), g.shape))
        eps = 1e-6
        vals = []
        for s in (+1, -1):
            q = {k: v.clone() for k, v in p.items()}
            q[name][idx] += s * eps
            vals.append(O.siamese2d_train_step(arch, q, None, f1, f2, y)["loss"].item())
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - g[idx].item()) < 1e-5 * max(1.0, abs(fd)), name
    # inference mode uses the moving statistics
    e = O.encoder2d_forward(arch, res["params"], f1, training=False)
    assert e.shape == (2, 4) and torch.isfinite(e).all()
