"""Helpers for the -m gpu parity tests: call the C ABI on torch device buffers, compare with the CPU oracle."""
import numpy as np
import torch

from voicemap_amd import _lib
from voicemap_amd._lib import VM_BF16, VM_F16, VM_F32, VM_F32S

DTYPES = {"f32": (VM_F32, torch.float32), "bf16": (VM_BF16, torch.bfloat16), "f32s": (VM_F32S, torch.float32), "f16": (VM_F16, torch.float16)}


def L():
    return _lib.lib()


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


# Kernels are enqueued asynchronously from raw pointers, so a temporary device tensor must outlive the launch:
# otherwise the caching allocator hands its block to the next temporary and the H2D copy of THAT one lands on
# top of it before the kernel runs.  Everything made here is kept until the test ends (conftest clears it).
_KEEP = []


def release():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _KEEP.clear()


def dev(a, dtype=torch.float32):
    t = torch.as_tensor(np.asarray(a)).to("cuda", dtype).contiguous()
    _KEEP.append(t)
    return t


def padded(x, tdt):
    """(N, L, C) -> (N, L+2, C) with zero halo rows, on the device in the storage dtype."""
    x = torch.as_tensor(np.asarray(x), dtype=torch.float64)
    n, l, c = x.shape
    out = torch.zeros(n, l + 2, c, dtype=torch.float64)
    out[:, 1:l + 1] = x
    t = out.to("cuda", tdt).contiguous()
    _KEEP.append(t)
    return t


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_err(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def quant(x, name):
    """Round a float64 array through the storage dtype (so the oracle sees the same inputs the kernel saw)."""
    t = torch.as_tensor(np.asarray(x), dtype=torch.float64)
    if name == "bf16":
        return t.to(torch.bfloat16).to(torch.float64)
    if name == "f16":
        return t.to(torch.float16).to(torch.float64)
    return t.to(torch.float32).to(torch.float64)


def grad_close(a, b, rtol, atol=1e-7):
    """relative L2 error below rtol, or (for gradients that are analytically ~0, e.g. the dense bias of the twin
    towers whose two contributions cancel) absolute max error below atol."""
    return rel_err(a, b) < rtol or max_err(a, b) < atol


def cosine(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def report(test, key, value):
    """Append a measured parity figure to gpurun_out/parity_report.csv (kept under profiles/ per round)."""
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.csv"), "a") as f:
            f.write("%s,%s,%.6e\n" % (test, key, value))
    except OSError:
        pass
