"""Where and when the workgroups of a conv_nt3_kernel launch ran: per CU workgroup slot the sequence of tiles with their start / end
stamps (s_memtime) and the hardware ids (HW_ID, XCC_ID) -- the effective shader clock of the launch, the idle time of a slot between two
workgroups, and the phase of the two workgroups that share a CU.  Against a -DVM_EXPERIMENT_PROFILE -DVM_EXPERIMENT_PROFILE_HW build:
  VM_PROF_EXTRA=-DVM_EXPERIMENT_PROFILE_HW bash tools/build_profile_lib.sh
  VOICEMAP_HIP_LIB=voicemap_amd/lib/libvoicemap_hip_prof.so python tools/probe/nt3_slots.py [key=value tuning knobs]"""
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
            run = lambda: L.call("vm_conv_fwd_fold", p(a), p(w), p(bias), p(hb), p(gam), n, n // 2, l, cin, cout, vm, None, p(ss), p(sq), p(e), p(o), p(wp), None, st())
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
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        L.cdll.vm_debug_prof_read(buf.ctypes.data, NS)
        tilesN = (cout if kind == "fwd" else cin) // 128
        nwg = min(8192, n * ((l + 253) // 254) * tilesN)
        raw = buf[:nwg * 4].astype(np.int64).reshape(nwg, 4, 8)
        start = raw[:, :, 0]; end = raw[:, :, 1]; hw = raw[:, 0, 2]; xcc = raw[:, 0, 3] & 0xf
        cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; tg = (hw >> 16) & 0xf; simd = (hw >> 4) & 3; wv = hw & 0xf
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        # s_memtime counters are not synchronised across the chip: every CU's stamps are taken from ITS first workgroup start (found as
        # the stamp after the largest circular gap, so that a 32-bit wrap inside the launch does not matter)
        ws = np.zeros(nwg, np.int64); we = np.zeros(nwg, np.int64); bar = np.zeros(nwg, np.int64); loop = np.zeros(nwg, np.int64)
        spans = []
        for x in np.unique(cuid):
            m = cuid == x
            allst = np.sort(np.unique(start[m].ravel()))
            gaps_c = np.diff(np.concatenate([allst, [allst[0] + (1 << 32)]]))
            t0 = allst[(np.argmax(gaps_c) + 1) % len(allst)]
            ws[m] = ((start[m] - t0) % (1 << 32)).min(1); we[m] = ((end[m] - t0) % (1 << 32)).max(1)
            bar[m] = ((raw[m][:, :, 4] - t0) % (1 << 32)).max(1); loop[m] = ((raw[m][:, :, 5] - t0) % (1 << 32)).max(1)
            spans.append(we[m].max())
        span = float(np.median(spans))
        ncu = len(np.unique(cuid))
        gaps, lives, per_cu_n, phase = [], [], [], []
        for c in np.unique(cuid):
            idx = np.where(cuid == c)[0]
            per_cu_n.append(len(idx))
            order = idx[np.argsort(ws[idx])]
            # slots: greedy assignment of the CU's workgroups to two lanes by start time
            lanes = [[], []]
            for i in order:
                k = 0 if (not lanes[0] or we[lanes[0][-1]] <= ws[i]) else 1
                if lanes[k] and we[lanes[k][-1]] > ws[i]:
                    k = 1 - k
                lanes[k].append(i)
            for ln in lanes:
                for a_, b_ in zip(ln[:-1], ln[1:]):
                    gaps.append(ws[b_] - we[a_])
                lives += [we[i] - ws[i] for i in ln]
            # phase: for every workgroup of lane 1, where in the concurrent lane-0 workgroup's life it started (0..1)
            for i in lanes[1]:
                for j in lanes[0]:
                    if ws[j] <= ws[i] < we[j]:
                        phase.append((ws[i] - ws[j]) / max(1, we[j] - ws[j]))
                        break
        gaps, lives, phase = np.array(gaps), np.array(lives), np.array(phase)
        print("%-5s L%-4d %3d->%3d: %6.1f us | %d workgroups on %d CUs (%d..%d per CU) | launch span %d clk (median over CUs: first start .. last end on a CU) -> %.2f GHz | workgroup life %d clk (p10 %d p90 %d) | "
              "slot idle between workgroups: median %d p90 %d clk (%d gaps, %d negative) | sum life / (2 slots x span) = %.2f | K loop share of life %.2f" % (
                  kind, l, cin, cout, us, nwg, ncu, min(per_cu_n), max(per_cu_n), span, span / us / 1e3, lives.mean(), np.percentile(lives, 10), np.percentile(lives, 90),
                  np.median(gaps), np.percentile(gaps, 90), len(gaps), (gaps < 0).sum(), lives.sum() / (2.0 * ncu * span), float(np.mean((loop - bar) / np.maximum(1, we - ws)))))
        print("      phase of the second slot's workgroup start within the first slot's concurrent workgroup: quantiles 10/25/50/75/90 %% = %s   first-round tg ids: %s, wave ids %s" % (
            np.round(np.percentile(phase, [10, 25, 50, 75, 90]), 2), np.bincount(tg[:512], minlength=4)[:6], np.bincount(wv[:512], minlength=4)[:6]))
        del a, w, wp
        torch.cuda.empty_cache()
