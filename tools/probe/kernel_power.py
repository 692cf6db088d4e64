"""Socket power and shader clock while ONE kernel of the cfg-A step runs back to back for ~2.5 s (rocm-smi sampled from a thread beside
the launch loop): which launches sit on the package power limit (1400 W) and which have headroom.
  python tools/probe/kernel_power.py"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from voicemap_amd import _lib
from voicemap_amd.engine import HipEncoderEngine
L = _lib.lib()
vm, tdt = 3, torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
n = 256


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
            ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            if pw and ck:
                out.append((float(pw.group(1)), float(ck.group(1))))
        except Exception:
            pass


def measure(name, run, seconds=2.5, flop=0.0, nbytes=0.0):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.perf_counter(); k = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            run()
        k += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / k
    o = np.array(out[1:]) if len(out) > 2 else np.array(out or [(0, 0)])
    pw, ck = float(np.median(o[:, 0])), float(np.median(o[:, 1]))
    print("%-34s %8.1f us  %6.0f W  %5.0f MHz  -> %6.1f mJ per launch%s%s" % (
        name, us, pw, ck, pw * us * 1e-3, ("  %6.0f TFLOP/s" % (flop / us * 1e-6)) if flop else "", ("  %5.2f TB/s" % (nbytes / us * 1e-6)) if nbytes else ""), flush=True)


print("one launch shape of the cfg-A step (128 pairs, f16) back to back; median of the rocm-smi samples taken meanwhile")
for (l, cin, cout) in [(3000, 128, 256), (1500, 256, 384), (750, 384, 512)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.zeros(n, l + 2, cin, dtype=tdt, device="cuda"); a[:, 1:l + 1] = torch.randn(n, l, cin, device="cuda", generator=g).abs().to(tdt)
    w = (torch.randn(2 * cout * 3 * cin, device="cuda", generator=g) * 0.05).to(tdt)
    wp = torch.empty_like(w)
    L.call("vm_pack_nt_weights", p(w), 2, cout, cin, vm, p(wp), st())
    bias, hb, gam = torch.zeros(cout, device="cuda"), torch.zeros(2 * 4 * cout, device="cuda"), torch.ones(cout, device="cuda")
    rows = L.query("vm_conv_stat_rows", l)
    ss = torch.empty(n * rows, cout, device="cuda"); sq = torch.empty_like(ss)
    e = torch.zeros(n, l // 2 + 2, cout, dtype=tdt, device="cuda"); o = torch.empty(n, l // 2, cout, dtype=tdt, device="cuda")
    flop = 2.0 * n * l * 3 * cin * cout
    measure("fwd   L%d %d->%d" % (l, cin, cout), lambda: L.call("vm_conv_fwd_fold", p(a), p(w), p(bias), p(hb), p(gam), n, n // 2, l, cin, cout, vm, None, p(ss), p(sq), p(e), p(o), p(wp), None, st()), flop=flop)
    du = torch.zeros(n, l + 2, cout, dtype=tdt, device="cuda"); du[:, 1:l + 1] = torch.randn(n, l, cout, device="cuda", generator=g).to(tdt)
    wd = (torch.randn(cin * 3 * cout, device="cuda", generator=g) * 0.05).to(tdt)
    wdp = torch.empty_like(wd)
    L.call("vm_pack_nt_weights", p(wd), 1, cin, cout, vm, p(wdp), st())
    z = torch.empty(n, l, cin, dtype=tdt, device="cuda")
    ra = torch.randn(n, l + 2, cin, device="cuda", generator=g).to(tdt)
    rows2 = L.query("vm_conv_dgrad_bnred_rows", l)
    s0 = torch.empty(n * rows2, cin, device="cuda"); s1 = torch.empty_like(s0)
    measure("dgrad L%d %d->%d" % (l, cin, cout), lambda: L.call("vm_conv_dgrad_bnred", p(du), p(wd), n, l, cin, cout, vm, p(z), p(ra), 1, p(s0), p(s1), p(wdp), st()), flop=flop)
    ws = torch.empty(L.query("vm_conv_wgrad_fold_workspace_bytes", n, n // 2, l, cin, cout) // 4 + 16, device="cuda")
    measure("wgrad L%d %d->%d" % (l, cin, cout), lambda: L.call("vm_conv_wgrad_fold", p(a), p(du), n, n // 2, l, cin, cout, vm, None, None, None, p(ws), None, st()), flop=flop)
    del a, w, wp, du, wd, wdp, z, ra, ws, e, o
    torch.cuda.empty_cache()
# a streaming copy and a pure read for scale
x = torch.empty(256 * 1024 * 1024, dtype=torch.float16, device="cuda"); y = torch.empty_like(x)
measure("torch copy 512 MB -> 512 MB", lambda: y.copy_(x), nbytes=2.0 * x.numel() * 2)
del x, y
torch.cuda.empty_cache()
# the whole step, and its serial form
blocks = [(32, 128, 4), (3, 256, 2), (3, 384, 2), (3, 512, 2)]
for serial in (False, True):
    eng = HipEncoderEngine(blocks, 64, dropout=0.0, head="uniform_euclidean", dtype="f16", seed=1)
    if serial:
        eng.split_towers = eng.overlap_wgrad = False
    rng = np.random.default_rng(0)
    xr = torch.from_numpy(rng.normal(0, 0.05, (256, 48000)).astype(np.float32)).cuda()
    yy = torch.cat([torch.zeros(64), torch.ones(64)]).cuda()
    pl = eng.plan(256, 12000, True)
    measure("train step, 128 pairs%s" % (" (one stream)" if serial else ""), lambda: eng.train_step_resident(pl, 128, yy, "contrastive", raw=xr, input_ready=True))
    if not serial:
        eng.timed = {}
