"""Helpers for the -m gpu parity tests: call the C ABI on torch device buffers, compare with the CPU oracle."""
import numpy as np
import torch

from voicemap_amd import _lib
from voicemap_amd._lib import VM_BF16, VM_F32

DTYPES = {"f32": (VM_F32, torch.float32), "bf16": (VM_BF16, torch.bfloat16)}


def L():
    return _lib.lib()


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to("cuda", dtype).contiguous()


def padded(x, tdt):
    """(N, L, C) -> (N, L+2, C) with zero halo rows, on the device in the storage dtype."""
    x = torch.as_tensor(np.asarray(x), dtype=torch.float64)
    n, l, c = x.shape
    out = torch.zeros(n, l + 2, c, dtype=torch.float64)
    out[:, 1:l + 1] = x
    return out.to("cuda", tdt).contiguous()


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
    return t.to(torch.float32).to(torch.float64)
