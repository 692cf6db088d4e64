"""Host-side constants of the log-mel front-end (vm_stft_logmel): the windowed DFT basis and the mel filterbank, plus the
specification of the spectrogram variant (not in the reference -- SURVEY.md D9 -- so these numbers are this
repository's own choice of a conventional 16 kHz speech front-end):

    frames      25 ms (400 samples) every 10 ms (160 samples), no centre padding: T = 1 + (n - 400) // 160   (3 s -> 298)
    window      periodic Hann(400), zero-extended to n_fft = 512
    spectrum    power |X[k]|^2 of bins k = 0 .. 255
    mel         64 triangular filters (peak 1, HTK mel scale 2595 log10(1 + f / 700)) between 0 Hz and 8000 Hz
    log         natural log(mel + 1e-6)
"""
import numpy as np

SAMPLE_RATE = 16000
N_FFT = 512
WIN_LENGTH = 400
HOP = 160
N_MELS = 64
LOG_FLOOR = 1e-6
N_BINS = 256  # bins 0 .. 255: with fmin = 0 and fmax = Nyquist the triangular filters give bin 0 and bin 256 zero weight


def n_frames(n_samples: int, win_length: int = WIN_LENGTH, hop: int = HOP) -> int:
    return 0 if n_samples < win_length else 1 + (n_samples - win_length) // hop


def hann_periodic(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def dft_basis(win_length: int = WIN_LENGTH, n_fft: int = N_FFT) -> np.ndarray:
    """(win_length, 2 * N_BINS) float32: row n = window[n] * [cos(2 pi k n / n_fft) for k < 256 | -sin(2 pi k n / n_fft) for k < 256]."""
    n = np.arange(win_length, dtype=np.float64)[:, None]
    k = np.arange(N_BINS, dtype=np.float64)[None, :]
    ang = 2.0 * np.pi * k * n / n_fft
    w = hann_periodic(win_length)[:, None]
    return np.ascontiguousarray(np.concatenate([w * np.cos(ang), -w * np.sin(ang)], axis=1).astype(np.float32))


def hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE, n_fft: int = N_FFT, fmin: float = 0.0,
                   fmax: float = None) -> np.ndarray:
    """(N_BINS, n_mels) float32 triangular filters with unit peak on the HTK mel scale."""
    fmax = sample_rate / 2.0 if fmax is None else fmax
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    freqs = np.arange(N_BINS, dtype=np.float64) * sample_rate / n_fft
    lo, mid, hi = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    f = freqs[:, None]
    w = np.maximum(0.0, np.minimum((f - lo) / (mid - lo), (hi - f) / (hi - mid)))
    return np.ascontiguousarray(w.astype(np.float32))
