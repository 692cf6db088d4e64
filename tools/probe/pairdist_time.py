"""vm_pairdist_argmin on one rank's shard of BASELINE.json config 5 (13 002 x 104 014 x 64, euclidean, argmin only): ms per launch and
the fraction of the fp32 VALU issue bound.   python tools/probe/pairdist_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voicemap_amd import _lib
L = _lib.lib()
N, M, E = 104014, 13002, 64
ref = torch.randn(N, E, device="cuda")
ws = torch.empty(L.query("vm_pairdist_workspace_bytes", M, N) // 4 + 16, device="cuda")
bv, bi = torch.empty(M, device="cuda"), torch.empty(M, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for kind, name in ((0, "euclidean"), (1, "cosine"), (2, "dot_product")):
    run = lambda: L.call("vm_pairdist_argmin", ref.data_ptr(), ref.data_ptr(), M, N, E, kind, 0, None, bv.data_ptr(), bi.data_ptr(), ws.data_ptr(), st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ops = (2.0 if kind == 0 else 1.0) * M * N * E / 64
    print("%-12s %7.3f ms   %.1f G pairs/s   %.0f G wave-instr/s = %.2f of 1228.8 (1024 SIMDs x 2.4 GHz / 2 clk)" % (name, ms, M * N / ms / 1e6, ops / ms / 1e6, ops / ms / 1e6 / 1228.8))
