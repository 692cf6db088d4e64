"""Where a wave of conv_nt3_kernel spends its clocks (the hot entry points of a cfg-A step: vm_conv_fwd_fold with the pair epilogue,
vm_conv_dgrad_bnred), against a -DVM_EXPERIMENT_PROFILE build:
  bash tools/build_profile_lib.sh && VOICEMAP_HIP_LIB=voicemap_amd/lib/libvoicemap_hip_prof.so python tools/probe/nt3_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
            ctr = torch.zeros(2 * cout, device="cuda") if os.environ.get("CTR") else None   # CTR=1: the centred-tile form (block 2's default)
            run = lambda: L.call("vm_conv_fwd_fold", p(a), p(w), p(bias), p(hb), p(gam), n, n // 2, l, cin, cout, vm, None, p(ss), p(sq), p(e), p(o), p(wp),
                                 p(ctr) if ctr is not None else None, st())
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
        if os.environ.get("EPI"):     # -DVM_EXPERIMENT_PROFILE_EPI build: raw stamps start, [K loop end], sync, tile write, sync, pool pairs / stores, statistics, drained
            tilesN = (cout if kind == "fwd" else cin) // 128
            nwg = min(8192, n * ((l + 253) // 254) * tilesN)
            raw = buf[:nwg * 4].astype(np.int64)
            dd = np.diff(raw, axis=1) % (1 << 32)
            m = dd.mean(0)
            us = e0.elapsed_time(e1) * 1e3 / reps
            if os.environ.get("PER_WAVE"):   # which of the four waves arrives late at the epilogue's first barrier
                pw = dd[:, 1].reshape(-1, 4)
                k0 = dd[:, 0].reshape(-1, 4)
                print("%-5s L%-4d per-wave (w = 2 wm + wn) mean clocks start -> end of K loop %s | wait at barrier 1 %s | share of tiles where wave w waits least %s" % (
                    kind, l, np.round(k0.mean(0)).astype(int), np.round(pw.mean(0)).astype(int), np.round(np.bincount(pw.argmin(1), minlength=4) / len(pw), 2)))
            if os.environ.get("PER_WAVE"):
                spread = k0.max(1) - k0.min(1)
                print("      per-tile spread of the four waves' arrival (max - min of start -> end of K loop): quantiles 10/50/90/99 %% = %s; wait quantiles %s; "
                      "tiles whose slowest wave is > 1000 clocks behind the fastest: %.2f" % (
                          np.percentile(spread, [10, 50, 90, 99]).astype(int), np.percentile(pw, [10, 50, 90, 99]).astype(int), (spread > 1000).mean()))
            if os.environ["EPI"] == "2":
                if kind == "dgrad":
                    print("dgrad L%-4d %3d->%3d: %6.1f us/launch | start -> tile stores issued %6.0f | dp transposing reads %5.0f | barrier %5.0f | 16 LDS-DMA issued %5.0f | "
                          "16 MFMAs + vmcnt(0): second operand landed AND output stores acknowledged %5.0f | barrier %5.0f | 32 reads + 16 MFMAs + sums stored + drained %5.0f | total %6.0f"
                          % (l, cin, cout, us, m[0], m[1], m[2], m[3], m[4], m[5], m[6], dd.sum(1).mean()))
                continue
            print("%-5s L%-4d %3d->%3d: %6.1f us/launch | start -> end of K loop %6.0f | wait at barrier 1 %5.0f | tile write %5.0f | wait at barrier 2 %5.0f | "
                  "%s %5.0f | %s %5.0f | store drain (vmcnt 0) %5.0f | total %6.0f" % (
                      kind, l, cin, cout, us, m[0], m[1], m[2], m[3], "pool pairs + stores" if kind == "fwd" else "tile stores", m[4],
                      "statistics" if kind == "fwd" else "(none)", m[5], m[6], dd.sum(1).mean()))
            continue
        tilesN = (cout if kind == "fwd" else cin) // 128
        nwg = min(8192, n * ((l + 253) // 254) * tilesN)
        d = buf[:nwg * 4].astype(np.float64)
        nk = 3 * ((cin if kind == "fwd" else cout) // 32)
        m = d.mean(0)
        us = e0.elapsed_time(e1) * 1e3 / reps
        clock = m[0] * (nwg / 512.0) / us / 1e3
        print("%-5s L%-4d %3d->%3d: %6.1f us/launch | per wave-tile %6.0f clk: setup %4.0f, prologue issue %4.0f, init + first wait + barrier %4.0f, "
              "first K tile %4.0f | other %2d K tiles %6.0f (%4.0f each; 1024 = two waves sharing a SIMD at full MFMA rate) | epilogue + store drain %4.0f "
              "| clock %.2f GHz if a slot's tiles ran back to back" % (kind, l, cin, cout, us, m[0], m[4], m[5], m[6], m[7], nk - 1, m[2],
                                                                      m[2] / (nk - 1), m[3], clock))
