"""Probe: the two towers as two half-batch pipelines on two streams, tower B ONE STAGE behind tower A, so that a GEMM of one tower
runs next to a streaming BatchNorm pass of the other (forward only, training mode).  Compared with the one-launch-per-stage
forward of the full 256-window batch."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from voicemap_amd.engine import HipEncoderEngine, _p
dev = torch.device("cuda", 0)
F, E = 128, 64
blocks = [(32, F, 4), (3, 2 * F, 2), (3, 3 * F, 2), (3, 4 * F, 2)]
eng = HipEncoderEngine(blocks, E, dropout=0.0, head="uniform_euclidean", dtype="bf16", device=dev, seed=1234)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.normal(0, 0.05, (256, 48000)).astype(np.float32)).to(dev)
plf = eng.plan(256, 12000, True)
# two independent half-batch plans (plan() caches by key: make the second by hand)
pla = eng.plan(128, 12000, True)
eng._plans.pop((128, 12000, True))
plb = eng.plan(128, 12000, True)


def stages(pl, xs):
    """the forward of one tower as a list of closures, one per kernel group"""
    n, dt = pl["n"], eng.dtype
    out = [lambda: eng.preprocess(pl, xs, 4, True, n)]
    for i, (k, c, pool) in enumerate(eng.blocks):
        b, L = pl[i], pl["L"][i]
        ssum, ssq = _p(b["ssum"]), _p(b["ssq"])
        bias = _p(eng.view(f"conv{i+1}.bias"))
        gam, bet = _p(eng.view(f"bn{i+1}.gamma")), _p(eng.view(f"bn{i+1}.beta"))
        fin = lambda b=b, ssum=ssum, ssq=ssq, c=c, L=L, gam=gam, bet=bet: eng._call(
            "vm_bn_finalize", ssum, ssq, n * b["stat_rows"], 1, c, float(n * L), gam, bet, eng.bn_eps, eng.bn_momentum, 1, None, None,
            _p(b["mean"]), _p(b["invstd"]), _p(b["scale"]), _p(b["shift"]), _p(pl["cr_ws"]), None, 0.0, None, None, None, eng.stream())
        if i == 0:
            out.append(lambda b=b, ssum=ssum, ssq=ssq, c=c, L=L, pool=pool, bias=bias, gam=gam: eng._call(
                "vm_conv1_fused_fwd", _p(pl["x0"]), _p(eng.view("conv1.kernel")), bias, gam, None, n, L, c, pool, 0, _p(b["e"]), ssum, ssq,
                eng.stream()))
            out.append(fin)
            out.append(lambda b=b, c=c: eng._call("vm_bn_drop_pool_fwd", _p(b["e"]), _p(b["scale"]), _p(b["shift"]), None, n, n,
                                                 pl["L"][1], c, 1, dt, _p(b["act"]), eng.stream()))
            continue
        cin = eng.blocks[i - 1][1]
        out.append(lambda b=b, i=i, cin=cin, c=c, L=L, bias=bias, ssum=ssum, ssq=ssq: eng._call(
            "vm_conv_fwd", _p(pl[i - 1]["act"]), _p(eng.wf[i]), bias, n, L, cin, c, dt, _p(b["z"]), ssum, ssq, eng.stream()))
        out.append(fin)
        if i == eng.nb - 1:
            out.append(lambda b=b, c=c, L=L, pool=pool: eng._call("vm_bn_drop_pool_gmax_fwd", _p(b["z"]), _p(b["scale"]), _p(b["shift"]), None,
                                                                 n, n, L, c, pool, dt, _p(pl["gmax"]), _p(pl["gidx"]), _p(pl["gmax_ws"]),
                                                                 eng.stream()))
        else:
            out.append(lambda b=b, c=c, L=L, pool=pool: eng._call("vm_bn_drop_pool_fwd", _p(b["z"]), _p(b["scale"]), _p(b["shift"]), None, n, n,
                                                                 L, c, pool, dt, _p(b["act"]), eng.stream()))
    return out


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def one():
    eng.preprocess(plf, x, 4, True, 128)
    eng.forward(plf, 128, None)


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
A, B = stages(pla, x[:128]), stages(plb, x[128:])
evs = [torch.cuda.Event() for _ in A]


def seq():
    for f in A:
        f()
    for f in B:
        f()


def pipelined(lag=1):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    for k in range(len(A)):
        with torch.cuda.stream(sa):
            A[k]()
            evs[k].record(sa)
        if k - lag + 1 >= 0:
            j = k - lag + 1      # B's stage j may start once A's stage j + lag - 1 ... keep B `lag` GROUPS behind
        with torch.cuda.stream(sb):
            sb.wait_event(evs[k])
            B[k]()
    cur.wait_stream(sa)
    cur.wait_stream(sb)


def free_running():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        for f in A:
            f()
    with torch.cuda.stream(sb):
        for f in B:
            f()
    cur.wait_stream(sa)
    cur.wait_stream(sb)


for name, fn in (("one 256-window forward", one), ("two 128-window forwards, one stream", seq),
                 ("two streams, B one stage behind A", pipelined), ("two streams, free running", free_running)):
    print("%-44s %.3f ms" % (name, t(fn)))
