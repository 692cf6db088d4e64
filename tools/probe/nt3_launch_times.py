"""The six hot conv_nt3_kernel launches of a cfg-A step (vm_conv_fwd_fold with the pair epilogue, vm_conv_dgrad_bnred; f16, 256 windows),
each timed alone with HIP events for every value of one vm_set_tuning knob, values interleaved, outputs compared bit for bit with the
first value's (NOCHECK=1: experiment builds whose results are wrong by design, tools/probe/nt3_ablate.sh).
  python tools/probe/nt3_launch_times.py [knob v0 v1 ...]       (default: nt3_lean 3 -- i.e. just the times)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from voicemap_amd import _lib
L = _lib.lib()
vm, tdt = 3, torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
knob = sys.argv[1] if len(sys.argv) > 1 else "nt3_lean"
pcts = [int(v) for v in sys.argv[2:]] or [3]
n = 256
print("conv_nt3_kernel, us per launch alone (median of 5 x 10 launches), vm_set_tuning %s =" % knob)
print("%-24s" % "launch" + "".join("%10d" % v for v in pcts))
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
            outs = (e, o, ss, sq)
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
            outs = (z, s0, s1)
            run = lambda: L.call("vm_conv_dgrad_bnred", p(a), p(w), n, l, cin, cout, vm, p(z), p(ra), 1, p(s0), p(s1), p(wp), st())
        res = {v: [] for v in pcts}
        ref = None
        for rep in range(5):
            for v in pcts:
                L.call("vm_set_tuning", knob.encode(), v)
                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record(); torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) * 100.0)
                if rep == 0:
                    cur = [t.clone() for t in outs]
                    if ref is None:
                        ref = cur
                    elif not os.environ.get("NOCHECK"):
                        assert all(torch.equal(x.view(torch.int16) if x.dtype == tdt else x.view(torch.int32), y.view(torch.int16) if y.dtype == tdt else y.view(torch.int32))
                                   for x, y in zip(ref, cur)), "the knob changed a result"
        print("%-24s" % ("%s L%d %d->%d" % (kind, l, cin, cout)) + "".join("%10.1f" % float(np.median(res[v])) for v in pcts))
        del a, w, wp
        torch.cuda.empty_cache()
