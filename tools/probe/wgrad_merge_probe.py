"""vm_conv_wgrad on the 2-D variant's layers with g consecutive windows taken as one (their zero halo rows make the concatenation a
valid k = 3 sequence; the sum is the same): time per merge factor g.  python tools/probe/wgrad_merge_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from voicemap_amd import _lib
lib = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
for (nw, L, cin, cout, M) in [(8192, 149, 96, 64, 32), (4096, 74, 192, 96, 16), (2048, 37, 288, 128, 8)]:
    x = torch.randn(nw, L + 2, cin, device="cuda").half(); x[:, 0] = 0; x[:, -1] = 0
    du = torch.randn(nw, L + 2, cout, device="cuda").half(); du[:, 0] = 0; du[:, -1] = 0
    gw = torch.empty(3, cin, cout, device="cuda")
    ref = None
    for g in (1, 2, 4, 8, M, 2 * M, 8 * M):
        n2, L2 = nw // g, g * (L + 2) - 2
        ws = torch.empty(lib.query("vm_conv_wgrad_workspace_bytes", n2, L2, cin, cout) // 4 + 16, device="cuda")
        f = lambda: lib.call("vm_conv_wgrad", x.data_ptr(), du.data_ptr(), n2, L2, cin, cout, _lib.VM_F16, ws.data_ptr(), gw.data_ptr(), st())
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = gw.clone()
        err = ((gw - ref).norm() / ref.norm()).item()
        print("layer %dx%d->%d L=%d: g=%4d  n=%5d L'=%6d splits=%3d  %.1f us  rel diff vs g=1 %.1e" % (
            nw, cin, cout, L, g, n2, L2, lib.query("vm_conv_wgrad_splits", n2, L2, cin, cout), e0.elapsed_time(e1) * 100, err))
