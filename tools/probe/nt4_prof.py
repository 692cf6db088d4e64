"""Where the compute and the drain waves of conv_nt4_kernel spend their ticks (s_memtime sums over a workgroup's tiles), against a
-DVM_EXPERIMENT_PROFILE build:  bash tools/build_profile_lib.sh && VOICEMAP_HIP_LIB=voicemap_amd/lib/libvoicemap_hip_prof.so PYTHONPATH=. python tools/probe/nt4_prof.py"""
import ctypes, sys
import numpy as np, torch
from voicemap_amd import _lib
L = _lib.lib()
vm, tdt = 3, torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
NS = 8192 * 4
buf = np.zeros((NS, 8), np.uint32)
L.cdll.vm_debug_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = 256
for key, val in [kv.split("=") for kv in sys.argv[1:]]:
    L.call("vm_set_tuning", key.encode(), int(val))
for (l, cin, cout) in [(3000, 128, 256), (1500, 256, 384), (750, 384, 512)]:
    for kind in ("fwd", "dgrad"):
        g = torch.Generator(device="cuda").manual_seed(1)
        if kind == "fwd":
            a = torch.zeros(n, l + 2, cin, dtype=tdt, device="cuda"); a[:, 1:l + 1] = torch.randn(n, l, cin, device="cuda", generator=g).abs().to(tdt)
            w = (torch.randn(2 * cout * 3 * cin, device="cuda", generator=g) * 0.05).to(tdt)
            wp = torch.empty_like(w)
            L.call("vm_pack_nt_weights", p(w), 2, cout, cin, vm, p(wp), st())
            bias, hb, gam = torch.zeros(cout, device="cuda"), torch.zeros(2 * 4 * cout, device="cuda"), torch.ones(cout, device="cuda")
            rows = L.query("vm_conv_stat_rows", l)
            ss = torch.empty(n * rows, cout, device="cuda"); sq = torch.empty_like(ss)
            e = torch.zeros(n, l // 2 + 2, cout, dtype=tdt, device="cuda"); o = torch.empty(n, l // 2, cout, dtype=tdt, device="cuda")
            run = lambda: L.call("vm_conv_fwd_fold", p(a), p(w), p(bias), p(hb), p(gam), n, n // 2, l, cin, cout, vm, None, p(ss), p(sq), p(e), p(o), p(wp), st())
        else:
            a = torch.zeros(n, l + 2, cout, dtype=tdt, device="cuda"); a[:, 1:l + 1] = torch.randn(n, l, cout, device="cuda", generator=g).to(tdt)
            w = (torch.randn(cin * 3 * cout, device="cuda", generator=g) * 0.05).to(tdt)
            wp = torch.empty_like(w)
            L.call("vm_pack_nt_weights", p(w), 1, cin, cout, vm, p(wp), st())
            z = torch.empty(n, l, cin, dtype=tdt, device="cuda")
            ra = torch.randn(n, l + 2, cin, device="cuda", generator=g).to(tdt)
            rows = L.query("vm_conv_dgrad_bnred_rows", l)
            s0 = torch.empty(n * rows, cin, device="cuda"); s1 = torch.empty_like(s0)
            run = lambda: L.call("vm_conv_dgrad_bnred", p(a), p(w), n, l, cin, cout, vm, p(z), p(ra), 1, p(s0), p(s1), p(wp), st())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 10
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        L.cdll.vm_debug_prof_read(buf.ctypes.data, NS)
        d = buf[:256 * 8].astype(np.float64).reshape(256, 8, 8)
        c, dr = d[:, :4].reshape(-1, 8).mean(0), d[:, 4:].reshape(-1, 8).mean(0)
        tiles = c[6]
        us = e0.elapsed_time(e1) * 1e3 / reps
        nk = 3 * ((cin if kind == "fwd" else cout) // 32)
        print("%-5s L%-4d %3d->%3d: %6.1f us | %2.0f tiles/WG | compute wave per tile: K loop %6.0f (%4.0f per K tile), next-tile setup+prefetch %5.0f, wait B1 %5.0f, "
              "tile write %5.0f, B2+prefetch wait %5.0f | drain wave per tile: drain %6.0f, idle at chunk barriers %6.0f, B1+B2 %5.0f | ticks/us %.0f"
              % (kind, l, cin, cout, us, tiles, c[1] / tiles, c[1] / tiles / nk, c[2] / tiles, c[3] / tiles, c[4] / tiles, c[5] / tiles,
                 dr[1] / tiles, dr[2] / tiles, dr[3] / tiles, c[0] / us))
