"""Where a wave of conv1_fused_fwd_kernel spends its clocks at the cfg-A shape (256 windows, L 12000, 128 filters, pool 4, padded extreme),
against a -DVM_EXPERIMENT_PROFILE build of conv1_fused.hip:
  bash tools/build_variant.sh c1prof -DVM_EXPERIMENT_PROFILE conv1_fused.hip
  VOICEMAP_HIP_LIB=voicemap_amd/lib/libvoicemap_hip_c1prof.so PYTHONPATH=$PWD python tools/probe/conv1_prof.py [f1_fwd_blocks=N]"""
import ctypes, sys
import numpy as np, torch
from voicemap_amd import _lib
L = _lib.lib()
vm, tdt = 3, torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
for key, val in [kv.split("=") for kv in sys.argv[1:]]:
    L.call("vm_set_tuning", key.encode(), int(val))
n, l, f, pool = 256, 12000, 128, 4
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(n, l + 31, device="cuda", generator=g) * 0.04
w = torch.randn(32, f, device="cuda", generator=g) * 0.1
b = torch.randn(f, device="cuda", generator=g) * 0.02
gam = torch.ones(f, device="cuda")
rows = L.query("vm_conv1_stat_rows", l)
ss = torch.empty(n * rows, f, device="cuda"); sq = torch.empty_like(ss)
e = torch.zeros(n, l // pool + 2, f, dtype=tdt, device="cuda")
run = lambda: L.call("vm_conv1_fused_fwd", p(x), p(w), p(b), p(gam), None, n, l, f, pool, 2, vm, p(e), p(ss), p(sq), st())
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 10
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
NS = 4096 * 4
buf = np.zeros((NS, 8), np.uint32)
L.cdll.vm_debug_prof1_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.cdll.vm_debug_prof1_read(buf.ctypes.data, NS)
d = buf.astype(np.float64)
d = d[d[:, 0] > 0]
m = d.mean(0)
tiles = 47 * 8 * n / (len(d) / 1.0) * 1.0  # row tiles per wave (every wave walks its workgroup's chunks x 8 row tiles)
print("%.1f us/launch, %d waves stamped, %.1f tiles per wave" % (us, len(d), tiles))
print("per wave (ticks): total %.0f | prologue %.0f | first fetch+stash %.0f | barriers %.0f | conv (reads+MFMA) %.0f (%.0f/tile) | "
      "epilogue+stores %.0f (%.0f/tile) | stash %.0f | tail %.0f" % (m[0], m[1], m[2], m[3], m[4], m[4] / tiles, m[5], m[5] / tiles, m[6], m[7]))
print("waves resident per SIMD if the launch were one steady state: %.1f" % (m[0] * len(d) / 4 / 256 / (us * 1e-6 * 100e6 * 1.0) if False else (m[0] * len(d) / (256 * 4) / (us * 100.0))))
