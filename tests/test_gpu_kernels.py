"""-m gpu parity tests, one per C-ABI entry point: HIP kernel vs the CPU oracle (oracle/voicemap_oracle.py) on the
same seeded inputs.  Tolerances: fp32 storage -> 2e-5 relative (L2) unless stated; bf16 storage -> the oracle is fed
the bf16-rounded inputs and outputs are compared at 1e-2 relative (bf16 has 8 mantissa bits: 2^-8 = 3.9e-3 per
rounding)."""
import numpy as np
import pytest
import torch

from oracle import voicemap_oracle as O
from tests.gpu_util import DTYPES, L, dev, max_err, p, padded, quant, rel_err, stream

pytestmark = pytest.mark.gpu
# "f32s": fp32 storage, split-bf16 products (3 bf16 MFMAs per product; ~2^-17 per product, measured <= 1e-5 on these shapes)
# f16: 11 significand bits, 2^-12 = 2.4e-4 per rounding (inputs are pre-rounded, the output is rounded once)
TOL = {"f32": 2e-5, "bf16": 1e-2, "f32s": 5e-5, "f16": 1.5e-3}


def rng(seed):
    return np.random.default_rng(seed)


# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,l,f", [(3, 700, 16), (2, 256, 128), (1, 37, 8)])
def test_conv1_fwd(dt, n, l, f):
    vm, tdt = DTYPES[dt]
    r = rng(1)
    x = r.normal(0, 0.05, (n, l)).astype(np.float32)
    w = r.normal(0, 0.2, (32, 1, f)).astype(np.float32)
    b = r.normal(0, 0.05, (f,)).astype(np.float32)
    xp = np.zeros((n, l + 31), np.float32)
    xp[:, 15:15 + l] = x
    rows = L().query("vm_conv1_stat_rows", l)
    z = torch.empty(n, l, f, dtype=tdt, device="cuda")
    ss = torch.zeros(n * rows, f, device="cuda")
    sq = torch.zeros(n * rows, f, device="cuda")
    L().call("vm_conv1_fwd", p(dev(xp)), p(dev(w)), p(dev(b)), n, l, f, vm, p(z), p(ss), p(sq), stream())
    ref = O.conv1d_same_relu(torch.tensor(x, dtype=torch.float64)[:, :, None], torch.tensor(w, dtype=torch.float64),
                             torch.tensor(b, dtype=torch.float64)).numpy()
    zz = z.float().cpu().numpy()
    assert rel_err(zz, ref) < TOL[dt]
    # statistics are taken over the stored (rounded) values
    assert rel_err(ss.cpu().numpy().reshape(n, rows, f).sum(1), zz.astype(np.float64).sum(1)) < 1e-5
    assert rel_err(sq.cpu().numpy().reshape(n, rows, f).sum(1), (zz.astype(np.float64) ** 2).sum(1)) < 1e-5


def _conv_ref(x, w, b):
    return O.conv1d_same_relu(x, w, b)


GEMM_DEFAULTS = {"nt_n2": 3, "nt_glds": 1, "tn_x": 1, "tn_tile": 256, "tn9": 1, "tn9_stages": 1}   # the library's defaults (conv_gemm.hip / conv_wgrad.hip)


@pytest.fixture
def gemm_kernels(request):
    """Pins a kernel selection (vm_set_tuning) for one test and restores the defaults: every selectable kernel must agree with
    the oracle on every shape."""
    for k, v in request.param.items():
        L().call("vm_set_tuning", k.encode(), v)
    yield request.param
    for k, v in GEMM_DEFAULTS.items():
        L().call("vm_set_tuning", k.encode(), v)


GEMM_SHAPES = [(3, 200, 16, 24), (2, 300, 128, 256), (1, 129, 8, 136), (2, 5, 24, 8), (2, 260, 32, 64), (3, 131, 96, 32),
               (8, 140, 64, 384), (3, 520, 256, 512), (8, 300, 64, 256), (16, 1030, 256, 256)]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,l,cin,cout", GEMM_SHAPES)
def test_conv_fwd_dgrad_wgrad(dt, n, l, cin, cout):
    """The default dispatch of the three conv GEMM entry points (see the headers of conv_gemm.hip / conv_wgrad.hip) against
    the float64 oracle: K tails (c_in = 8, 16, 24, 96), ragged t-tiles, N tails (136), windows shorter than a tile (5)."""
    _conv_fwd_dgrad_wgrad(dt, n, l, cin, cout)


@pytest.mark.parametrize("gemm_kernels", [{"nt_n2": 0, "tn_x": 0}, {"nt_n2": 0, "nt_glds": 0, "tn_x": 0, "tn_tile": 128}],
                         indirect=True, ids=["lds-dma-128+tn256", "register-staged-128"])
@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("n,l,cin,cout", [GEMM_SHAPES[i] for i in (1, 2, 3, 5, 7, 9)])
def test_conv_fwd_dgrad_wgrad_fallback_kernels(dt, n, l, cin, cout, gemm_kernels):
    """... and the kernels the default dispatch does not pick on these shapes (the fallbacks of other shapes / storage types), pinned
    through vm_set_tuning: every selectable kernel agrees with the oracle wherever it can be selected.  (bf16 differs from f16 in
    the MFMA instruction only, which the default-dispatch test covers.)"""
    _conv_fwd_dgrad_wgrad(dt, n, l, cin, cout)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("n,l,cin,cout", [(2, 700, 128, 256), (3, 760, 256, 256), (2, 650, 64, 512), (1, 129, 32, 128), (2, 5, 128, 128),
                                          (3, 131, 192, 320), (1, 62, 64, 64), (4, 3000, 128, 256), (4, 1500, 256, 384), (4, 750, 384, 512)])
def test_conv_input_resident_kernel_shapes(dt, n, l, cin, cout):
    """conv_nt2r_kernel / conv_tn8x_kernel under the default dispatch on the shapes that exercise their tiling: 254-position tiles
    with ragged last tiles (700, 760, 650), windows whose second statistics row of the last tile does not exist (129, 650: the
    forward falls back to the 128 x 128 kernel there, dgrad does not), a window shorter than one tile (5), cfg-A's own geometries."""
    _conv_fwd_dgrad_wgrad(dt, n, l, cin, cout)


@pytest.mark.parametrize("gemm_kernels", [{"tn9": 0}, {"tn9": 2}], indirect=True, ids=["tn8x-slots", "tn9-producer-waves"])
@pytest.mark.parametrize("n,l,cin,cout", [(2, 700, 128, 256), (2, 650, 64, 512), (2, 5, 128, 128), (3, 131, 192, 320), (1, 62, 64, 64),
                                          (4, 750, 384, 512)])
def test_conv_wgrad_kernel_variants(n, l, cin, cout, gemm_kernels):
    """The two other forms of the input-resident wgrad tile (the default is conv_tn9_kernel): conv_tn8x_kernel (READ / MFMA slots) and
    conv_tn9_kernel with producer waves, on ragged stages (700, 650, 131), a window shorter than a stage (5, 62), half-filled tiles
    (64, 192, 320 channels) and cfg-A's block 4."""
    _conv_fwd_dgrad_wgrad("f16", n, l, cin, cout)


@pytest.mark.parametrize("gemm_kernels", [{"tn9_stages": 1}, {"tn9_stages": 0}, {"tn9": 2, "tn9_stages": 1}], indirect=True,
                         ids=["stage-splits", "window-splits", "stage-splits+producer-waves"])
@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("n,wpt,l,cin,cout", [(6, 3, 190, 64, 64), (10, 5, 64, 128, 128), (3, 3, 62, 64, 128), (14, 7, 1000, 128, 128),
                                              (2, 1, 3000, 128, 256), (9, 9, 129, 192, 64), (22, 11, 318, 64, 64)])
def test_conv_wgrad_split_granularity(dt, n, wpt, l, cin, cout, gemm_kernels):
    """conv_tn9_kernel's split-K ranges: 64-position stages of a tower's window stream (default; a split may begin and end inside a
    window, and one that ends inside stages one stage more than it computes, for taps 1 and 2 of its last two positions) against whole
    windows.  Both the plain entry point (one tower) and vm_conv_wgrad_fold's slabs per tower (identity affine, zero shift: the
    folded gradient IS the plain one) must match the float64 definition; windows of 1, 2, 3, 5, 16 and 47 stages, stage counts that
    do and do not divide the split length, towers of 1 .. 11 windows."""
    vm, tdt = DTYPES[dt]
    r = rng(n * l + cin)
    x = quant(r.normal(0, 1.0, (n, l, cin)), dt).numpy()
    du = quant(r.normal(0, 1.0, (n, l, cout)), dt).numpy()
    xp_ = np.zeros((n, l + 2, cin))
    xp_[:, 1:l + 1] = x
    want = np.stack([np.einsum("nli,nlo->io", xp_[:, k:k + l], du) for k in range(3)])
    xp, dup = padded(x, tdt), padded(du, tdt)
    ws = torch.empty(L().query("vm_conv_wgrad_workspace_bytes", n, l, cin, cout) // 4 + 16, device="cuda")
    gw = torch.empty(3, cin, cout, device="cuda")
    L().call("vm_conv_wgrad", p(xp), p(dup), n, l, cin, cout, vm, p(ws), p(gw), stream())
    assert rel_err(gw.cpu().numpy(), want) < 2e-5
    towers = n // wpt
    one, zero = torch.ones(towers, cin, device="cuda"), torch.zeros(towers, cin, device="cuda")
    dsum = torch.zeros(towers, 3, cout, device="cuda")
    wsf = torch.empty(L().query("vm_conv_wgrad_fold_workspace_bytes", n, wpt, l, cin, cout) // 4 + 16, dtype=torch.float32, device="cuda")
    gwf = torch.empty(3, cin, cout, device="cuda")
    L().call("vm_conv_wgrad_fold", p(xp), p(dup), n, wpt, l, cin, cout, vm, p(one), p(zero), p(dsum), p(wsf), p(gwf), stream())
    assert rel_err(gwf.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("n,l,cin,cout", [(3, 200, 16, 24), (2, 300, 128, 256), (1, 129, 8, 136), (2, 5, 24, 8), (3, 131, 96, 32),
                                          (8, 140, 64, 384), (3, 520, 256, 512), (4, 3000, 128, 256), (4, 1500, 256, 384),
                                          (4, 750, 384, 512)])
def test_conv_f32_storage_split_bf16_products(n, l, cin, cout):
    """dtype VM_F32S: fp32 operands staged as bf16 hi + lo halves, a*b = a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix pipe
    (conv_nt_kernel / conv_tn_kernel / conv_tn256_kernel with SPLIT): forward + statistics, dgrad and wgrad against the float64
    oracle; ragged tiles, K tails (c_in = 8, 16, 24, 96), both wgrad tilings and cfg-A's own geometries."""
    _conv_fwd_dgrad_wgrad("f32s", n, l, cin, cout)


def _conv_fwd_dgrad_wgrad(dt, n, l, cin, cout):
    vm, tdt = DTYPES[dt]
    r = rng(2)
    x = quant(r.normal(0, 1.0, (n, l, cin)), dt)
    w = quant(r.normal(0, 0.1, (3, cin, cout)), dt)
    b = torch.tensor(r.normal(0, 0.3, (cout,)).astype(np.float32), dtype=torch.float64)
    wf = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda")
    wd = torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
    L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
    xp = padded(x, tdt)
    rows = L().query("vm_conv_stat_rows", l)
    z = torch.empty(n, l, cout, dtype=tdt, device="cuda")
    ss = torch.zeros(n * rows, cout, device="cuda")
    sq = torch.zeros(n * rows, cout, device="cuda")
    L().call("vm_conv_fwd", p(xp), p(wf), p(dev(b)), n, l, cin, cout, vm, p(z), p(ss), p(sq), stream())
    ref = _conv_ref(x, w, b).numpy()
    zz = z.float().cpu().numpy()
    assert rel_err(zz, ref) < TOL[dt]
    assert rel_err(ss.cpu().numpy().reshape(n, rows, cout).sum(1), zz.astype(np.float64).sum(1)) < 1e-5
    assert rel_err(sq.cpu().numpy().reshape(n, rows, cout).sum(1), (zz.astype(np.float64) ** 2).sum(1)) < 1e-5
    # inference launch (no statistics) gives the same z
    z2 = torch.empty_like(z)
    L().call("vm_conv_fwd", p(xp), p(wf), p(dev(b)), n, l, cin, cout, vm, p(z2), None, None, stream())
    assert torch.equal(z, z2)

    # dgrad / wgrad against autograd of the linear conv (du given)
    du = quant(r.normal(0, 1.0, (n, l, cout)), dt)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    pl_, pr_ = O.same_padding(3)
    y = torch.nn.functional.conv1d(torch.nn.functional.pad(xr.transpose(1, 2), (pl_, pr_)), wr.permute(2, 1, 0)).transpose(1, 2)
    gx, gw = torch.autograd.grad((y * du).sum(), [xr, wr])
    dup = padded(du, tdt)
    dx = torch.empty(n, l, cin, dtype=tdt, device="cuda")
    L().call("vm_conv_dgrad", p(dup), p(wd), n, l, cin, cout, vm, p(dx), stream())
    assert rel_err(dx.float().cpu().numpy(), gx.numpy()) < TOL[dt]
    ws = torch.empty(L().query("vm_conv_wgrad_workspace_bytes", n, l, cin, cout) // 4 + 16, device="cuda")
    gwd = torch.empty(3, cin, cout, device="cuda")
    L().call("vm_conv_wgrad", p(xp), p(dup), n, l, cin, cout, vm, p(ws), p(gwd), stream())
    # fp32 accumulation of exact products of the (rounded) operands: tight in both modes
    assert rel_err(gwd.cpu().numpy(), gw.numpy()) < (5e-5 if dt == "f32s" else 2e-5)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_conv1_wgrad(dt):
    vm, tdt = DTYPES[dt]
    r = rng(3)
    n, l, f = 3, 1500, 24
    x = r.normal(0, 0.05, (n, l)).astype(np.float32)
    du = quant(r.normal(0, 1, (n, l, f)), dt)
    xp = np.zeros((n, l + 31), np.float32)
    xp[:, 15:15 + l] = x
    ws = torch.empty(L().query("vm_conv1_wgrad_workspace_bytes", n, f) // 4 + 16, device="cuda")
    gw = torch.empty(32, 1, f, device="cuda")
    L().call("vm_conv1_wgrad", p(dev(xp)), p(padded(du, tdt)), n, l, f, vm, p(ws), p(gw), stream())
    xt = torch.tensor(xp, dtype=torch.float64)
    ref = torch.stack([(xt[:, k:k + l, None] * du).sum((0, 1)) for k in range(32)])[:, None, :]
    assert rel_err(gw.cpu().numpy(), ref.numpy()) < 2e-5


# ----------------------------------------------------------------------------------------------------------
def _bn_block_oracle(z, gamma, beta, drop, pool, wpt, eps=1e-3):
    """BN(train, per tower) -> dropout -> maxpool on float64 tensors with autograd."""
    outs = []
    stats = []
    for t0 in range(0, z.shape[0], wpt):
        zt = z[t0:t0 + wpt]
        y, mean, var = O.batchnorm_train(zt, gamma, beta, eps)
        if drop is not None:
            y = y * drop[t0:t0 + wpt, None, :]
        outs.append(O.maxpool1d(y, pool))
        stats.append((mean, var))
    return torch.cat(outs, 0), stats


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,wpt,l,c,pool,use_drop", [(4, 2, 50, 16, 2, True), (2, 1, 64, 136, 4, False), (4, 4, 31, 8, 2, True),
                                                     (2, 2, 9, 24, 4, False)])
def test_bn_drop_pool_fwd_bwd(dt, n, wpt, l, c, pool, use_drop):
    vm, tdt = DTYPES[dt]
    r = rng(4)
    z = quant(np.maximum(r.normal(0.2, 1.0, (n, l, c)), 0.0), dt)  # post-ReLU activations, many exact zeros
    gamma = torch.tensor(r.normal(1.0, 0.3, c) * np.where(r.random(c) < 0.2, -1, 1), dtype=torch.float64)
    beta = torch.tensor(r.normal(0, 0.3, c), dtype=torch.float64)
    drop = None
    if use_drop:
        drop = torch.tensor((r.random((n, c)) > 0.3) / 0.7, dtype=torch.float64)
    lq = l // pool
    towers = n // wpt
    # forward statistics straight from z (as the conv epilogue would produce them: one partial row per window)
    zd = z.to("cuda", tdt).contiguous()
    ssum = z.sum(1).to("cuda", torch.float32).contiguous()
    ssq = (z * z).sum(1).to("cuda", torch.float32).contiguous()
    f32 = dict(dtype=torch.float32, device="cuda")
    mean, invstd, scale, shift = (torch.empty(towers, c, **f32) for _ in range(4))
    mm = torch.zeros(c, **f32)
    mv = torch.ones(c, **f32)
    crws = torch.empty(L().query("vm_colreduce_workspace_bytes", towers, c) // 8, dtype=torch.float64, device="cuda")
    L().call("vm_bn_finalize", p(ssum), p(ssq), wpt, towers, c, float(wpt * l), p(dev(gamma)), p(dev(beta)), 1e-3, 0.99, 1,
             p(mm), p(mv), p(mean), p(invstd), p(scale), p(shift), p(crws), None, 0.0, None, None, None, None, stream())
    zr = z.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    out_ref, stats = _bn_block_oracle(zr, gr, br, drop, pool, wpt)
    for t, (m_, v_) in enumerate(stats):
        assert max_err(mean[t].cpu().numpy(), m_.detach().numpy()) < 1e-5
        assert rel_err(invstd[t].cpu().numpy(), (1 / torch.sqrt(v_ + 1e-3)).detach().numpy()) < 1e-5
    # moving statistics: sequential update per tower, Keras unbiased-variance factor
    mm_ref, mv_ref = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    for (m_, v_) in stats:
        mm_ref = O.moving_update(mm_ref, m_.detach(), 0.99)
        mv_ref = O.moving_update(mv_ref, O.bn_unbiased_variance(v_.detach(), wpt * l, 1e-3), 0.99)
    assert max_err(mm.cpu().numpy(), mm_ref.numpy()) < 1e-6
    assert max_err(mv.cpu().numpy(), mv_ref.numpy()) < 1e-6
    # the same two calls with Keras 2.2.2 / TF 1.10 zero-debias accumulators (two consecutive training steps on the same batch)
    zdb = torch.zeros(towers * 2 * c, **f32)
    mm2, mv2 = torch.zeros(c, **f32), torch.ones(c, **f32)
    state = "fresh"
    newp = {"bn1.moving_mean": torch.zeros(c, dtype=torch.float64), "bn1.moving_variance": torch.ones(c, dtype=torch.float64)}
    for step in (1, 2):
        L().call("vm_bn_finalize", p(ssum), p(ssq), wpt, towers, c, float(wpt * l), p(dev(gamma)), p(dev(beta)), 1e-3, 0.99, 1,
                 p(mm2), p(mv2), p(mean), p(invstd), p(scale), p(shift), p(crws), p(zdb), 1.0 / (1.0 - 0.99 ** step), None, None, None, None, stream())
        collects = [{"bn_mean": [m_.detach()], "bn_var": [v_.detach()], "bn_count": [wpt * l]} for (m_, v_) in stats]
        state = O.apply_moving_updates(newp, collects, 1, 1e-3, 0.99, True, state)
        assert rel_err(mm2.cpu().numpy(), newp["bn1.moving_mean"].numpy()) < 1e-5
        assert rel_err(mv2.cpu().numpy(), newp["bn1.moving_variance"].numpy()) < 1e-5
    # after one or two steps on the same batch the de-biased average IS the last tower's batch statistic
    assert rel_err(mm2.cpu().numpy(), stats[-1][0].detach().numpy()) < 1e-5

    dropd = dev(drop) if drop is not None else None
    out = torch.zeros(n, lq + 2, c, dtype=tdt, device="cuda")
    L().call("vm_bn_drop_pool_fwd", p(zd), p(scale), p(shift), p(dropd), n, wpt, l, c, pool, vm, p(out), stream())
    o = out.float().cpu().numpy()
    assert np.all(o[:, 0] == 0) and np.all(o[:, -1] == 0)
    assert rel_err(o[:, 1:-1], out_ref.detach().numpy()) < (2e-5 if dt == "f32" else 6e-3)

    # backward
    dp = quant(r.normal(0, 1, (n, lq, c)), dt)
    gz, gg, gb = torch.autograd.grad((out_ref * dp).sum(), [zr, gr, br])
    gz = gz * (z > 0)  # ReLU of the producing conv is fused into this backward
    rows = L().query("vm_bn_part_rows")
    pa, pb, pdu = (torch.zeros(n * rows, c, **f32) for _ in range(3))
    c1, c2 = torch.empty(towers, c, **f32), torch.empty(towers, c, **f32)
    ggam, gbet = torch.empty(c, **f32), torch.empty(c, **f32)
    dpd = dp.to("cuda", tdt).contiguous()
    common = (p(zd), p(dpd), p(scale), p(shift), p(mean), p(invstd), p(dropd))
    L().call("vm_bn_pool_bwd_reduce", *common, n, wpt, l, c, pool, vm, p(pa), p(pb), stream())
    L().call("vm_bn_bwd_finalize", p(pa), p(pb), n, wpt, c, float(wpt * l), p(c1), p(c2), p(ggam), p(gbet), p(crws), stream())
    du = torch.zeros(n, l + 2, c, dtype=tdt, device="cuda")
    L().call("vm_bn_pool_bwd_apply", *common, p(c1), p(c2), n, wpt, l, c, pool, vm, p(du), p(pdu), stream())
    gbias = torch.empty(c, **f32)
    L().call("vm_colsum", p(pdu), n * rows, c, p(gbias), p(crws), stream())
    tol = 5e-5 if dt == "f32" else 1e-2
    assert rel_err(ggam.cpu().numpy(), gg.numpy()) < tol
    assert rel_err(gbet.cpu().numpy(), gb.numpy()) < tol
    d = du.float().cpu().numpy()
    assert np.all(d[:, 0] == 0) and np.all(d[:, -1] == 0)
    assert rel_err(d[:, 1:-1], gz.numpy()) < tol
    assert rel_err(gbias.cpu().numpy(), d[:, 1:-1].astype(np.float64).sum((0, 1))) < 1e-5


def test_bn_infer_affine():
    r = rng(5)
    c = 40
    g, b, mm = (r.normal(0, 1, c).astype(np.float32) for _ in range(3))
    mv = r.random(c).astype(np.float32)
    sc, sh = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    L().call("vm_bn_infer_affine", p(dev(g)), p(dev(b)), p(dev(mm)), p(dev(mv)), 1e-3, c, p(sc), p(sh), stream())
    z = torch.tensor(r.normal(0, 1, (2, 5, c)))
    ref = O.batchnorm_infer(z, torch.tensor(g, dtype=torch.float64), torch.tensor(b, dtype=torch.float64),
                            torch.tensor(mm, dtype=torch.float64), torch.tensor(mv, dtype=torch.float64), 1e-3)
    got = z * sc.cpu().double() + sh.cpu().double()
    assert rel_err(got.numpy(), ref.numpy()) < 1e-5


# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,l,c", [(3, 375, 64), (2, 7, 136), (1, 1, 8)])
def test_global_maxpool(dt, n, l, c):
    vm, tdt = DTYPES[dt]
    r = rng(6)
    a = quant(np.round(r.normal(0, 1, (n, l, c)) * 4) / 4, dt)  # coarse grid -> ties, first max must win
    gmax = torch.empty(n, c, device="cuda")
    gidx = torch.empty(n, c, dtype=torch.int32, device="cuda")
    L().call("vm_global_maxpool_fwd", p(padded(a, tdt)), n, l, c, vm, p(gmax), p(gidx), stream())
    assert np.array_equal(gmax.cpu().numpy(), a.max(1).values.float().numpy())
    assert np.array_equal(gidx.cpu().numpy(), a.numpy().argmax(1).astype(np.int32))
    dg = r.normal(0, 1, (n, c)).astype(np.float32)
    dp = torch.empty(n, l, c, dtype=tdt, device="cuda")
    L().call("vm_global_maxpool_bwd", p(dev(dg)), p(gidx), n, l, c, vm, p(dp), stream())
    ref = np.zeros((n, l, c))
    ii, jj = np.meshgrid(np.arange(n), np.arange(c), indexing="ij")
    ref[ii, a.numpy().argmax(1), jj] = quant(dg, dt).numpy()
    assert np.array_equal(dp.float().cpu().numpy().astype(np.float64), ref)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,wpt,l,c,pool,use_drop", [(4, 2, 750, 64, 2, True), (2, 1, 37, 136, 4, False), (3, 3, 9, 8, 2, True),
                                                     (2, 2, 64, 512, 1, False)])
def test_bn_drop_pool_gmax_fused_equals_two_pass(dt, n, wpt, l, c, pool, use_drop):
    """Last block: the fused BN-apply + dropout + max-pool + global max must return exactly what vm_bn_drop_pool_fwd followed
    by vm_global_maxpool_fwd returns (values AND first-argmax indices; coarse value grid -> many ties)."""
    vm, tdt = DTYPES[dt]
    r = rng(31)
    towers = n // wpt
    z = quant(np.maximum(np.round(r.normal(0.2, 1.0, (n, l, c)) * 2) / 2, 0.0), dt).to("cuda", tdt).contiguous()
    scale = dev(np.round(r.normal(1.0, 0.3, (towers, c)) * 4) / 4 * np.where(r.random((towers, c)) < 0.3, -1, 1))
    shift = dev(np.round(r.normal(0, 0.3, (towers, c)) * 4) / 4)
    drop = dev((r.random((n, c)) > 0.2) / 0.8) if use_drop else None
    lq = l // pool
    act = torch.zeros(n, lq + 2, c, dtype=tdt, device="cuda")
    L().call("vm_bn_drop_pool_fwd", p(z), p(scale), p(shift), p(drop), n, wpt, l, c, pool, vm, p(act), stream())
    g0, i0 = torch.empty(n, c, device="cuda"), torch.empty(n, c, dtype=torch.int32, device="cuda")
    L().call("vm_global_maxpool_fwd", p(act), n, lq, c, vm, p(g0), p(i0), stream())
    g1, i1 = torch.empty_like(g0), torch.empty_like(i0)
    ws = torch.empty(L().query("vm_bn_drop_pool_gmax_workspace_bytes", n, c) // 4, device="cuda")
    L().call("vm_bn_drop_pool_gmax_fwd", p(z), p(scale), p(shift), p(drop), n, wpt, l, c, pool, vm, p(g1), p(i1), p(ws), stream())
    assert torch.equal(g0, g1)
    assert torch.equal(i0, i1)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,wpt,l,c,pool,use_drop", [(4, 2, 300, 64, 2, True), (2, 1, 64, 136, 4, False), (3, 3, 19, 8, 2, True)])
def test_bn_pool_bwd_reduce_pooled_form(dt, n, wpt, l, c, pool, use_drop):
    """vm_bn_pool_bwd_reduce_pooled (extreme of z recovered from the pooled forward output) against the z form: equal up to the
    storage rounding of the pooled tensor; channels with scale == 0 (fallback to z) and dropped channels included."""
    vm, tdt = DTYPES[dt]
    r = rng(41)
    towers = n // wpt
    z = quant(np.maximum(r.normal(0.3, 1.0, (n, l, c)), 0.0), dt).to("cuda", tdt).contiguous()
    sc = r.normal(1.0, 0.3, (towers, c)) * np.where(r.random((towers, c)) < 0.3, -1, 1)
    sc[:, 3] = 0.0  # not invertible: this 8-channel vector must come from z
    scale, shift = dev(sc), dev(r.normal(0, 0.3, (towers, c)))
    mean, invstd = dev(r.normal(0.4, 0.1, (towers, c))), dev(r.uniform(0.5, 2.0, (towers, c)))
    drop = dev((r.random((n, c)) > 0.25) / 0.75) if use_drop else None
    lq = l // pool
    act = torch.zeros(n, lq + 2, c, dtype=tdt, device="cuda")
    L().call("vm_bn_drop_pool_fwd", p(z), p(scale), p(shift), p(drop), n, wpt, l, c, pool, vm, p(act), stream())
    dp = quant(r.normal(0, 1.0, (n, lq, c)), dt).to("cuda", tdt).contiguous()
    rows = L().query("vm_bn_part_rows")
    out = []
    for pooled in (False, True):
        pa, pb = (torch.zeros(n * rows, c, device="cuda") for _ in range(2))
        if pooled:
            L().call("vm_bn_pool_bwd_reduce_pooled", p(z), p(act), p(dp), p(scale), p(shift), p(mean), p(invstd), p(drop), n, wpt, l,
                     c, pool, vm, p(pa), p(pb), stream())
        else:
            L().call("vm_bn_pool_bwd_reduce", p(z), p(dp), p(scale), p(shift), p(mean), p(invstd), p(drop), n, wpt, l, c, pool, vm,
                     p(pa), p(pb), stream())
        out.append((pa.view(n, rows, c).sum(1).cpu().numpy().astype(np.float64), pb.view(n, rows, c).sum(1).cpu().numpy().astype(np.float64)))
    (a0, b0), (a1, b1) = out
    assert np.array_equal(a0, a1)  # sum of dy does not involve z at all
    tol = 2e-5 if dt == "f32" else 1.5e-2
    assert rel_err(b1, b0) < tol
    assert np.array_equal(b0[:, 3], b1[:, 3])  # the scale == 0 channel took the z path: bit-identical


def test_dense_fwd_bwd():
    r = rng(7)
    rows, ni, no = 10, 72, 33
    x, w, b, dout = (r.normal(0, 1, s).astype(np.float32) for s in [(rows, ni), (ni, no), (no,), (rows, no)])
    out = torch.empty(rows, no, device="cuda")
    L().call("vm_dense_fwd", p(dev(x)), p(dev(w)), p(dev(b)), rows, ni, no, p(out), stream())
    assert rel_err(out.cpu().numpy(), x.astype(np.float64) @ w + b) < 1e-6
    gw, gb, din = torch.empty(ni, no, device="cuda"), torch.empty(no, device="cuda"), torch.empty(rows, ni, device="cuda")
    L().call("vm_dense_bwd", p(dev(x)), p(dev(w)), p(dev(dout)), rows, ni, no, p(gw), p(gb), p(din), stream())
    assert rel_err(gw.cpu().numpy(), x.astype(np.float64).T @ dout) < 1e-6
    assert rel_err(gb.cpu().numpy(), dout.astype(np.float64).sum(0)) < 1e-6
    assert rel_err(din.cpu().numpy(), dout.astype(np.float64) @ w.T) < 1e-6


@pytest.mark.parametrize("head", ["uniform_euclidean", "weighted_l1"])
@pytest.mark.parametrize("loss", ["contrastive", "bce"])
@pytest.mark.parametrize("pairs,e,gscale", [(6, 32, 1.0), (300, 8, 1.0), (6, 32, 4096.0)])
def test_siamese_head_loss(head, loss, pairs, e, gscale):
    from voicemap_amd.engine import HEADS, LOSSES
    r = rng(8)
    emb = r.normal(0, 0.4, (2 * pairs, e)).astype(np.float32)
    hw = r.normal(0.5, 0.3, (1, 1) if head == "uniform_euclidean" else (e, 1)).astype(np.float32)
    hb = r.normal(-0.5, 0.1, (1,)).astype(np.float32)
    y = (r.random(pairs) > 0.5).astype(np.float32)
    pred, la = torch.empty(pairs, device="cuda"), torch.empty(2, device="cuda")
    demb = torch.empty(2 * pairs, e, device="cuda")
    ghw, ghb = torch.empty(hw.size, device="cuda"), torch.empty(1, device="cuda")
    L().call("vm_siamese_head_loss", p(dev(emb)), p(dev(hw)), p(dev(hb)), p(dev(y)), pairs, e, HEADS[head], LOSSES[loss], gscale,
             p(pred), p(la), p(demb), p(ghw), p(ghb), p(torch.empty(4 * pairs, device="cuda")), stream())
    et = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    prm = {"head.kernel": torch.tensor(hw, dtype=torch.float64, requires_grad=True),
           "head.bias": torch.tensor(hb, dtype=torch.float64, requires_grad=True)}
    pr = O.siamese_head(prm, et[:pairs], et[pairs:], head)
    yt = torch.tensor(y, dtype=torch.float64)[:, None]
    lo = O.contrastive_loss(yt, pr) if loss == "contrastive" else O.binary_crossentropy(yt, pr)
    ge, gw, gb = torch.autograd.grad(lo, [et, prm["head.kernel"], prm["head.bias"]])
    assert rel_err(pred.cpu().numpy(), pr.detach().numpy()[:, 0]) < 1e-5
    assert abs(la[0].item() - lo.item()) < 2e-5 * max(1.0, abs(lo.item()))
    assert abs(la[1].item() - O.binary_accuracy(yt, pr).item()) < 1e-6
    # grad_scale (the loss scale of f16 storage) multiplies every gradient output and nothing else
    assert rel_err(demb.cpu().numpy() / gscale, ge.numpy()) < 1e-4
    assert rel_err(ghw.cpu().numpy() / gscale, gw.numpy().ravel()) < 1e-4
    assert rel_err(ghb.cpu().numpy() / gscale, gb.numpy()) < 1e-4
    # predict-only launch leaves the training outputs alone and gives the same pred
    pred2 = torch.empty(pairs, device="cuda")
    L().call("vm_siamese_head_loss", p(dev(emb)), p(dev(hw)), p(dev(hb)), None, pairs, e, HEADS[head], LOSSES[loss], 1.0,
             p(pred2), None, None, None, None, None, stream())
    assert torch.equal(pred, pred2)


def test_siamese_head_rejects_unimplemented_metric():
    from voicemap_amd._lib import VoicemapHipError
    d = torch.zeros(8, device="cuda")
    with pytest.raises(VoicemapHipError):
        L().call("vm_siamese_head_loss", p(d), p(d), p(d), None, 2, 2, 5, 0, 1.0, p(d), None, None, None, None, None, stream())


@pytest.mark.parametrize("rows,nc,gscale", [(5, 40, 1.0), (3, 1172, 1.0), (5, 40, 4096.0)])
def test_softmax_cce(rows, nc, gscale):
    r = rng(9)
    logits = r.normal(0, 2, (rows, nc)).astype(np.float32)
    labels = r.integers(0, nc, rows).astype(np.int32)
    prob, dl = torch.empty(rows, nc, device="cuda"), torch.empty(rows, nc, device="cuda")
    la, ws = torch.empty(2, device="cuda"), torch.empty(2 * rows, device="cuda")
    L().call("vm_softmax_cce", p(dev(logits)), p(dev(labels, torch.int32)), rows, nc, gscale, p(prob), p(la), p(dl), p(ws), stream())
    lt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    pr = torch.softmax(lt, -1)
    oh = torch.nn.functional.one_hot(torch.tensor(labels, dtype=torch.int64), nc).double()
    lo = O.categorical_crossentropy(oh, pr)
    (g,) = torch.autograd.grad(lo, [lt])
    assert rel_err(prob.cpu().numpy(), pr.detach().numpy()) < 1e-5
    assert abs(la[0].item() - lo.item()) < 1e-5 * max(1, abs(lo.item()))
    assert abs(la[1].item() - O.categorical_accuracy(oh, pr).item()) < 1e-6
    assert rel_err(dl.cpu().numpy() / gscale, g.numpy()) < 1e-4


def test_adam_clip_step():
    r = rng(10)
    n = 5000
    for gscale in (0.001, 3.0):  # below and above the clip threshold
        pv, g = r.normal(0, 1, n).astype(np.float32), (r.normal(0, 1, n) * gscale).astype(np.float32)
        m0, v0 = r.normal(0, 0.01, n).astype(np.float32), (r.random(n) * 1e-3).astype(np.float32)
        P_, G_, M_, V_ = dev(pv), dev(g), dev(m0), dev(v0)
        ws = torch.empty(L().query("vm_sqnorm_workspace_bytes", n) // 8, dtype=torch.float64, device="cuda")
        sq = torch.empty(1, device="cuda")
        L().call("vm_grad_sqnorm", p(G_), n, p(ws), p(sq), stream())
        assert abs(sq.item() - float((g.astype(np.float64) ** 2).sum())) < 1e-5 * float((g.astype(np.float64) ** 2).sum())
        st = O.AdamState(iterations=6)
        st.m["w"], st.v["w"] = torch.tensor(m0, dtype=torch.float64), torch.tensor(v0, dtype=torch.float64)
        ref = O.adam_step(st, {"w": torch.tensor(pv, dtype=torch.float64)}, {"w": torch.tensor(g, dtype=torch.float64)})["w"]
        t = 7
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        L().call("vm_adam_clip_step", p(P_), p(G_), p(M_), p(V_), n, lr_t, 0.9, 0.999, 1e-7, 1.0, 1.0, p(sq), None, 0, None, stream())
        assert max_err(P_.cpu().numpy(), ref.numpy()) < 2e-6
        assert rel_err(M_.cpu().numpy(), st.m["w"].numpy()) < 1e-5
        assert rel_err(V_.cpu().numpy(), st.v["w"].numpy()) < 1e-5
        # the one-launch-less form: partials only, the optimizer kernel adds them (same order) and publishes the norm -- same bits
        P3, M3, V3 = dev(pv), dev(m0), dev(v0)
        sq3 = torch.zeros(1, device="cuda")
        L().call("vm_grad_sqnorm", p(G_), n, p(ws), None, stream())
        L().call("vm_adam_clip_step", p(P3), p(G_), p(M3), p(V3), n, lr_t, 0.9, 0.999, 1e-7, 1.0, 1.0, p(sq3), p(ws), 0, None, stream())
        assert torch.equal(P3, P_) and torch.equal(M3, M_) and torch.equal(V3, V_) and torch.equal(sq3, sq)
        # loss-scaled form (f16 storage): G holds 4096 x the gradients, grad_prescale divides it out before the clip -- the same
        # update to rounding; skip_nonfinite with a finite norm changes nothing and reports 0
        P2, G2, M2, V2 = dev(pv), dev(g * np.float32(4096.0)), dev(m0), dev(v0)
        flag = torch.full((1,), 7, dtype=torch.int32, device="cuda")
        L().call("vm_grad_sqnorm", p(G2), n, p(ws), p(sq), stream())
        L().call("vm_adam_clip_step", p(P2), p(G2), p(M2), p(V2), n, lr_t, 0.9, 0.999, 1e-7, 1.0, 1.0 / 4096.0, p(sq), None, 1, p(flag), stream())
        assert max_err(P2.cpu().numpy(), ref.numpy()) < 2e-6 and flag.item() == 7     # a running count: untouched by a good step
        # a non-finite gradient norm: the step is skipped on the device (p, m, v untouched) and flagged; without the switch the
        # NaN goes through like in Keras
        G2[17] = float("inf")
        before = (P2.clone(), M2.clone(), V2.clone())
        L().call("vm_grad_sqnorm", p(G2), n, p(ws), p(sq), stream())
        L().call("vm_adam_clip_step", p(P2), p(G2), p(M2), p(V2), n, lr_t, 0.9, 0.999, 1e-7, 1.0, 1.0 / 4096.0, p(sq), None, 1, p(flag), stream())
        assert flag.item() == 8 and torch.equal(P2, before[0]) and torch.equal(M2, before[1]) and torch.equal(V2, before[2])
        L().call("vm_adam_clip_step", p(P2), p(G2), p(M2), p(V2), n, lr_t, 0.9, 0.999, 1e-7, 1.0, 1.0 / 4096.0, p(sq), None, 0, None, stream())
        assert not torch.isfinite(P2).all()


@pytest.mark.parametrize("i16", [False, True])
def test_decimate_whiten(i16):
    r = rng(11)
    n, wpt, raw_len, ds = 6, 3, 4801, 4
    raw = (r.normal(0, 0.05, (n, raw_len)) + r.uniform(-0.01, 0.01, (n, 1)))
    if i16:
        raw_i = np.clip(np.round(raw * 32768), -32768, 32767).astype(np.int16)
        raw = raw_i.astype(np.float64) / 32768.0
        rd = dev(raw_i, torch.int16)
    else:
        raw = raw.astype(np.float32)
        rd = dev(raw)
    l0 = (raw_len + ds - 1) // ds
    out = torch.zeros(n, l0 + 31, device="cuda")
    ws = torch.empty(L().query("vm_decimate_whiten_workspace_bytes", n) // 8, dtype=torch.float64, device="cuda")
    L().call("vm_decimate_whiten", p(rd), int(i16), n, raw_len, ds, 1, 0.038021, wpt, p(out), p(ws), stream())
    pre = O.preprocess_instances(ds)
    ref = np.concatenate([pre(raw.astype(np.float64)[t:t + wpt, :, None]) for t in range(0, n, wpt)])[:, :, 0]
    o = out.cpu().numpy()
    assert np.all(o[:, :15] == 0) and np.all(o[:, 15 + l0:] == 0)
    assert max_err(o[:, 15:15 + l0], ref) < 1e-7
    L().call("vm_decimate_whiten", p(rd), int(i16), n, raw_len, ds, 0, 0.038021, wpt, p(out), p(ws), stream())
    assert max_err(out.cpu().numpy()[:, 15:15 + l0], raw[:, ::ds]) < 1e-7


@pytest.mark.parametrize("dist", ["euclidean", "cosine", "dot_product"])
@pytest.mark.parametrize("k,n", [(5, 1), (20, 5), (70, 2)])
def test_nshot_distances(dist, k, n):
    kinds = {"euclidean": 0, "cosine": 1, "dot_product": 2}
    r = rng(12)
    tasks, e = 7, 24
    q = r.normal(0, 1, (tasks, e)).astype(np.float32)
    s = r.normal(0, 1, (tasks, k * n, e)).astype(np.float32)
    pred = torch.empty(tasks, k, device="cuda")
    am = torch.empty(tasks, dtype=torch.int32, device="cuda")
    L().call("vm_nshot_distances", p(dev(q)), p(dev(s)), tasks, k, n, e, kinds[dist], p(pred), p(am), stream())
    ref = np.stack([O.n_shot_prediction(q[t], s[t], n, k, dist) for t in range(tasks)])
    assert rel_err(pred.cpu().numpy(), ref) < 1e-5
    assert np.array_equal(am.cpu().numpy(), ref.argmin(1).astype(np.int32))


# ----------------------------------------------------------------------------------------------------------
@pytest.fixture
def f1_splits(request):
    """Workgroups per window of the fused block-1 kernels: small test batches would otherwise always get one chunk per
    workgroup and never walk the double-buffered multi-chunk loop that the full-size batch runs."""
    per_window = request.param
    if per_window:
        n = request.node.callspec.params["n"]
        L().call("vm_set_tuning", b"f1_fwd_blocks", per_window * n)
        L().call("vm_set_tuning", b"f1_blocks", per_window * n)
    yield per_window
    L().call("vm_set_tuning", b"f1_fwd_blocks", 1024)   # (the defaults)
    L().call("vm_set_tuning", b"f1_blocks", 1024)


@pytest.mark.parametrize("f1_splits", [0, 1, 2], indirect=True)
@pytest.mark.parametrize("neg", [0.25, 0.0])  # 0.0: every gamma positive -> the wave-uniform all-maximum paths (the training default)
@pytest.mark.parametrize("n,wpt,l,f,pool,use_drop", [(4, 2, 700, 16, 4, True), (2, 1, 1200, 128, 4, False), (2, 2, 530, 40, 2, True),
                                                     (2, 1, 300, 160, 4, False), (3, 3, 2100, 64, 4, False)])
def test_conv1_fused_block(n, wpt, l, f, pool, use_drop, neg, f1_splits):
    """Fused block 1 (conv k=32 -> relu -> BN -> dropout -> maxpool) forward, inference forward and backward vs
    the float64 oracle.  z1 never reaches HBM, so it has no storage rounding: statistics, arg-max routing and the
    backward see the fp32 accumulator; the stored tensors are the rounded pooled extreme of z and the rounded activation."""
    _conv1_fused_block("bf16", n, wpt, l, f, pool, use_drop, neg)


@pytest.mark.parametrize("neg", [0.25, 0.0])
@pytest.mark.parametrize("n,wpt,l,f,pool,use_drop", [(4, 2, 700, 16, 4, True), (2, 1, 1200, 128, 4, False), (2, 2, 530, 40, 2, True),
                                                     (2, 1, 300, 160, 4, False), (3, 3, 2100, 64, 4, False)])
def test_conv1_fused_block_f16(n, wpt, l, f, pool, use_drop, neg):
    """The same kernels with half storage (dtype VM_F16): the stored tensors carry 11 significand bits instead of 8, and (round 6,
    f1_products = 2) the convolution takes the waveform ROUNDED TO HALF -- the precision every other layer's input has in this mode --
    against filters split hi + lo: held to the oracle on that rounded waveform at the old tolerances, and to the oracle on the
    un-rounded waveform at 6e-4 / 1.2e-3 (the end-to-end distance from the reference arithmetic is tests/test_gpu_fullsize_oracle.py's
    and tests/test_gpu_f16_guard.py's to hold: < 1e-3 in every state)."""
    _conv1_fused_block("f16", n, wpt, l, f, pool, use_drop, neg, round_x=True)
    # (forward only: a waveform that differs in its 12th bit re-routes a few pool windows, and the bias gradient is a cancelling sum)
    _conv1_fused_block("f16", n, wpt, l, f, pool, use_drop, neg, round_x=False, tols=(6e-4, 1.2e-3, 3e-4), backward=False)


@pytest.mark.parametrize("n,wpt,l,f,pool,use_drop", [(2, 1, 1200, 128, 4, False), (2, 2, 530, 40, 2, True)])
def test_conv1_fused_block_f16_three_product_form(n, wpt, l, f, pool, use_drop):
    """vm_set_tuning("f1_products", 3): the round-1..5 form (waveform and filters split hi + lo in bf16, three products) is still there
    and still meets the old tolerances against the oracle on the un-rounded waveform."""
    L().call("vm_set_tuning", b"f1_products", 3)
    try:
        _conv1_fused_block("f16", n, wpt, l, f, pool, use_drop, 0.25)
    finally:
        L().call("vm_set_tuning", b"f1_products", 2)


def _conv1_fused_block(dt, n, wpt, l, f, pool, use_drop, neg, round_x=False, tols=None, backward=True):
    vm, tdt = DTYPES[dt]
    tol_e, tol_act = (2e-3, 8e-3) if dt == "bf16" else (3e-4, 1e-3)
    tol_stat = 1e-4
    if tols is not None:
        tol_e, tol_act, tol_stat = tols
    r = rng(20)
    x = r.normal(0, 0.05, (n, l)).astype(np.float32)
    w = r.normal(0, 0.2, (32, 1, f)).astype(np.float32)
    b = r.normal(0, 0.05, (f,)).astype(np.float32)
    gamma = (np.abs(r.normal(1.0, 0.3, f)) * np.where(r.random(f) < neg, -1, 1)).astype(np.float32)
    beta = r.normal(0, 0.3, f).astype(np.float32)
    drop = ((r.random((n, f)) > 0.3) / 0.7).astype(np.float32) if use_drop else None
    xp = np.zeros((n, l + 31), np.float32)
    xp[:, 15:15 + l] = x
    lq = l // pool
    towers = n // wpt
    f32 = dict(dtype=torch.float32, device="cuda")
    rows = L().query("vm_conv1_stat_rows", l)
    xd, wd_, bd, gd, btd = dev(xp), dev(w), dev(b), dev(gamma), dev(beta)
    dropd = dev(drop) if drop is not None else None
    e = torch.empty(n, lq, f, dtype=tdt, device="cuda")
    ss, sq = torch.zeros(n * rows, f, **f32), torch.zeros(n * rows, f, **f32)
    L().call("vm_conv1_fused_fwd", p(xd), p(wd_), p(bd), p(gd), None, n, l, f, pool, 0, vm, p(e), p(ss), p(sq), stream())

    # ---- oracle: z (unrounded), BN per tower, dropout, pool
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    wr, br = T(w).requires_grad_(True), T(b).requires_grad_(True)
    gr, btr = T(gamma).requires_grad_(True), T(beta).requires_grad_(True)
    x_o = x.astype(np.float16).astype(np.float32) if round_x else x   # what the kernel's convolution is specified on
    z = O.conv1d_same_relu(T(x_o)[:, :, None], wr, br)
    zz = z.detach()
    pooled_max = O.maxpool1d(zz, pool)
    pooled_min = -O.maxpool1d(-zz, pool)
    e_ref = quant(torch.where(T(gamma) >= 0, pooled_max, pooled_min), dt)
    assert rel_err(e.float().cpu().numpy(), e_ref.numpy()) < tol_e
    assert rel_err(ss.cpu().numpy().reshape(n, rows, f).sum(1), zz.sum(1).numpy()) < tol_stat
    assert rel_err(sq.cpu().numpy().reshape(n, rows, f).sum(1), (zz * zz).sum(1).numpy()) < tol_stat

    # ---- BN finalize + affine/dropout on the pooled tensor == pool(BN(z)*drop)
    mean, invstd, scale, shift = (torch.empty(towers, f, **f32) for _ in range(4))
    crws = torch.empty(L().query("vm_colreduce_workspace_bytes", towers, f) // 8, dtype=torch.float64, device="cuda")
    L().call("vm_bn_finalize", p(ss), p(sq), wpt * rows, towers, f, float(wpt * l), p(gd), p(btd), 1e-3, 0.99, 1, None, None,
             p(mean), p(invstd), p(scale), p(shift), p(crws), None, 0.0, None, None, None, None, stream())
    # the centred form (vm_conv1_fused_fwd mode 2 stores e - max(bias, 0)): shift_adj / mean_adj carry the offset
    sh_adj, mean_adj = torch.empty_like(shift), torch.empty_like(mean)
    mean2, invstd2, scale2, shift2 = (torch.empty(towers, f, **f32) for _ in range(4))
    L().call("vm_bn_finalize", p(ss), p(sq), wpt * rows, towers, f, float(wpt * l), p(gd), p(btd), 1e-3, 0.99, 1, None, None,
             p(mean2), p(invstd2), p(scale2), p(shift2), p(crws), None, 0.0, p(bd), p(sh_adj), p(mean_adj), None, stream())
    ctr = torch.clamp(bd, min=0.0)[None, :]
    assert torch.equal(mean2, mean) and torch.equal(scale2, scale) and torch.equal(shift2, shift)
    # (the kernel forms shift_adj with ONE rounding, fma(scale, ctr, shift); torch rounds the product and the sum: where the two terms
    # cancel the difference is an ulp of the LARGER term)
    assert ((sh_adj - (shift + scale * ctr)).abs() <= 2.4e-7 * (shift.abs() + (scale * ctr).abs()) + 1e-7).all()
    assert torch.allclose(mean_adj, mean - ctr, rtol=1e-6, atol=1e-7)
    # mode 2: the extreme as a padded tensor, stored CENTRED (e - max(bias, 0)) in the storage type; same statistics; halo rows untouched
    ep = torch.zeros(n, lq + 2, f, dtype=tdt, device="cuda")
    ss2, sq2 = torch.zeros_like(ss), torch.zeros_like(sq)
    L().call("vm_conv1_fused_fwd", p(xd), p(wd_), p(bd), p(gd), None, n, l, f, pool, 2, vm, p(ep), p(ss2), p(sq2), stream())
    e_exact = torch.where(T(gamma) >= 0, pooled_max, pooled_min)
    ep_ref = quant(e_exact - torch.clamp(T(b), min=0.0), dt)
    assert rel_err(ep[:, 1:-1].float().cpu().numpy(), ep_ref.numpy()) < tol_e
    assert torch.equal(ss2, ss) and torch.equal(sq2, sq) and not ep[:, 0].any() and not ep[:, -1].any()
    act = torch.zeros(n, lq + 2, f, dtype=tdt, device="cuda")
    L().call("vm_bn_drop_pool_fwd", p(e), p(scale), p(shift), p(dropd), n, wpt, lq, f, 1, vm, p(act), stream())
    out_ref, _ = _bn_block_oracle(z, gr, btr, T(drop) if drop is not None else None, pool, wpt)
    assert rel_err(act.float().cpu().numpy()[:, 1:-1], out_ref.detach().numpy()) < tol_act

    # ---- backward
    dpq = quant(r.normal(0, 1, (n, lq, f)), dt)
    gw_ref, gb_ref, gg_ref, gbt_ref = torch.autograd.grad((out_ref * dpq).sum(), [wr, br, gr, btr])
    prow = L().query("vm_bn_part_rows")
    pa, pb = torch.zeros(n * prow, f, **f32), torch.zeros(n * prow, f, **f32)
    c1, c2 = torch.empty(towers, f, **f32), torch.empty(towers, f, **f32)
    ggam, gbet = torch.empty(f, **f32), torch.empty(f, **f32)
    dpd = dpq.to("cuda", tdt).contiguous()
    L().call("vm_bn_pool_bwd_reduce", p(e), p(dpd), p(scale), p(shift), p(mean), p(invstd), p(dropd), n, wpt, lq, f, 1, vm,
             p(pa), p(pb), stream())
    L().call("vm_bn_bwd_finalize", p(pa), p(pb), n, wpt, f, float(wpt * l), p(c1), p(c2), p(ggam), p(gbet), p(crws), stream())
    ws = torch.empty(L().query("vm_conv1_fused_bwd_workspace_bytes", n, l, f) // 4 + 16, **f32)
    gw, gb = torch.empty(32, 1, f, **f32), torch.empty(f, **f32)
    L().call("vm_conv1_fused_bwd", p(xd), p(wd_), p(bd), p(dpd), p(scale), p(mean), p(invstd), p(dropd), p(c1), p(c2), n, wpt,
             l, f, pool, vm, p(ws), p(gw), p(gb), stream())
    if backward:
        assert rel_err(ggam.cpu().numpy(), gg_ref.numpy()) < 2e-2
        assert rel_err(gbet.cpu().numpy(), gbt_ref.numpy()) < 2e-2
        assert rel_err(gw.cpu().numpy(), gw_ref.numpy()) < 2e-2
        assert rel_err(gb.cpu().numpy(), gb_ref.numpy()) < 2e-2

    # ---- inference forward: moving-statistics affine applied in the epilogue, padded output
    mm, mv = r.normal(0.1, 0.05, f).astype(np.float32), (r.random(f) * 0.01 + 1e-4).astype(np.float32)
    sc_i, sh_i = torch.empty(f, **f32), torch.empty(f, **f32)
    L().call("vm_bn_infer_affine", p(gd), p(btd), p(dev(mm)), p(dev(mv)), 1e-3, f, p(sc_i), p(sh_i), stream())
    act_i = torch.zeros(n, lq + 2, f, dtype=tdt, device="cuda")
    L().call("vm_conv1_fused_fwd", p(xd), p(wd_), p(bd), p(sc_i), p(sh_i), n, l, f, pool, 1, vm, p(act_i), None, None, stream())
    y_i = O.batchnorm_infer(zz, T(gamma), T(beta), T(mm), T(mv), 1e-3)
    ref_i = O.maxpool1d(y_i, pool).numpy()
    a = act_i.float().cpu().numpy()
    assert np.all(a[:, 0] == 0) and np.all(a[:, -1] == 0)
    assert rel_err(a[:, 1:-1], ref_i) < tol_act


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_bn_pool_bwd_sparse_gmax_form_equals_dense(dt):
    """The last block's BN-backward passes fed with (dg, gidx) must equal the passes fed with the dense tensor that
    vm_global_maxpool_bwd would have written: the apply pass bit for bit, the reduce pass (a gather in the sparse form, with
    its one term per window in partial row 0) per window up to fp32 rounding."""
    vm, tdt = DTYPES[dt]
    r = rng(30)
    n, wpt, l, c, pool = 4, 2, 46, 24, 2
    lq = l // pool
    z = quant(np.maximum(r.normal(0.2, 1.0, (n, l, c)), 0.0), dt).to("cuda", tdt).contiguous()
    f32 = dict(dtype=torch.float32, device="cuda")
    scale = dev(r.normal(1.0, 0.3, (2, c)) * np.where(r.random((2, c)) < 0.3, -1, 1))
    shift, mean = dev(r.normal(0, 0.3, (2, c))), dev(r.normal(0.3, 0.1, (2, c)))
    invstd = dev(r.uniform(0.5, 2.0, (2, c)))
    c1, c2 = dev(r.normal(0, 0.01, (2, c))), dev(r.normal(0, 0.01, (2, c)))
    dg = dev(r.normal(0, 1, (n, c)))
    gidx = dev(r.integers(0, lq, (n, c)), torch.int32)
    dense = torch.empty(n, lq, c, dtype=tdt, device="cuda")
    L().call("vm_global_maxpool_bwd", p(dg), p(gidx), n, lq, c, vm, p(dense), stream())
    rows = L().query("vm_bn_part_rows")
    outs = []
    for sparse in (False, True):
        pa, pb, pdu = (torch.zeros(n * rows, c, **f32) for _ in range(3))
        du = torch.zeros(n, l + 2, c, dtype=tdt, device="cuda")
        if sparse:
            head = (p(z), p(dg), p(gidx))
            L().call("vm_bn_pool_bwd_reduce_gmax", *head, p(scale), p(shift), p(mean), p(invstd), None, n, wpt, l, c, pool, vm,
                     p(pa), p(pb), stream())
            L().call("vm_bn_pool_bwd_apply_gmax", *head, p(scale), p(shift), p(mean), p(invstd), None, p(c1), p(c2), n, wpt, l, c,
                     pool, vm, p(du), p(pdu), stream())
        else:
            head = (p(z), p(dense))
            L().call("vm_bn_pool_bwd_reduce", *head, p(scale), p(shift), p(mean), p(invstd), None, n, wpt, l, c, pool, vm, p(pa),
                     p(pb), stream())
            L().call("vm_bn_pool_bwd_apply", *head, p(scale), p(shift), p(mean), p(invstd), None, p(c1), p(c2), n, wpt, l, c, pool,
                     vm, p(du), p(pdu), stream())
        outs.append((pa.clone(), pb.clone(), pdu.clone(), du.clone()))
    for a, b in zip(outs[0][2:], outs[1][2:]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0][:2], outs[1][:2]):
        wa, wb = a.view(n, rows, c).sum(1), b.view(n, rows, c).sum(1)
        assert torch.allclose(wa, wb, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,wpt,l,c,pool,with_drop", [(4, 2, 46, 24, 2, False), (70, 35, 30, 40, 2, True), (256, 128, 750, 512, 2, False), (6, 6, 32, 16, 4, True)])
def test_bn_bwd_gmax_finalize_equals_reduce_then_finalize(dt, n, wpt, l, c, pool, with_drop):
    """vm_bn_bwd_gmax_finalize (round 6: the last block's sparse BatchNorm-backward sums and their finalize in one launch;
    voicemap/models.py:32-37 backward) against the two entry points it replaces: c1, c2, grad_gamma, grad_beta agree to the last
    bit or, where the fp64 summation order shows, to one fp32 rounding."""
    vm, tdt = DTYPES[dt]
    r = rng(n + c)
    lq = l // pool
    z = quant(np.maximum(r.normal(0.2, 1.0, (n, l, c)), 0.0), dt).to("cuda", tdt).contiguous()
    towers = n // wpt
    scale = dev(r.normal(1.0, 0.3, (towers, c)) * np.where(r.random((towers, c)) < 0.3, -1, 1))
    shift, mean = dev(r.normal(0, 0.3, (towers, c))), dev(r.normal(0.3, 0.1, (towers, c)))
    invstd = dev(r.uniform(0.5, 2.0, (towers, c)))
    drop = dev((r.random((n, c)) > 0.2) / 0.8) if with_drop else None
    dg = dev(r.normal(0, 1, (n, c)))
    gi = r.integers(0, lq, (n, c))
    gi[0, :3] = -1                      # "no position" entries contribute nothing
    gidx = dev(gi, torch.int32)
    rows = L().query("vm_bn_part_rows")
    f32 = dict(dtype=torch.float32, device="cuda")
    pa, pb = torch.zeros(n * rows, c, **f32), torch.zeros(n * rows, c, **f32)
    ws = torch.empty(L().query("vm_colreduce_workspace_bytes", towers, c) // 4 + 64, **f32)
    ref = [torch.full((towers, c), float("nan"), **f32), torch.full((towers, c), float("nan"), **f32), torch.full((c,), float("nan"), **f32),
           torch.full((c,), float("nan"), **f32)]
    got = [t.clone() for t in ref]
    head = (p(z), p(dg), p(gidx), p(scale), p(shift), p(mean), p(invstd), p(drop))
    L().call("vm_bn_pool_bwd_reduce_gmax", *head, n, wpt, l, c, pool, vm, p(pa), p(pb), stream())
    L().call("vm_bn_bwd_finalize", p(pa), p(pb), n, wpt, c, float(wpt * l), p(ref[0]), p(ref[1]), p(ref[2]), p(ref[3]), p(ws), stream())
    L().call("vm_bn_bwd_gmax_finalize", *head, n, wpt, l, c, pool, vm, float(wpt * l), p(got[0]), p(got[1]), p(got[2]), p(got[3]), stream())
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.isfinite(b).all()
        assert torch.allclose(a, b, rtol=3e-7, atol=1e-12), (a - b).abs().max().item()


# ----------------------------------------------------------------------------------------------------------
# cfg-A's own GEMM geometries (experiments/train_siamese.py:20-25: filters 128 -> blocks 2..4 are 128->256 @ L=3000,
# 256->384 @ L=1500, 384->512 @ L=750) under the DEFAULT dispatch -- no vm_set_tuning call in these tests, so they
# exercise exactly the kernels the bench line runs (conv_nt2r_kernel, the three-chunk c_in = 384 walk of conv_tn8x_kernel, ...).
# Reference arithmetic: voicemap/models.py:22-35.
CFG_A_GEMMS = [(3000, 128, 256), (1500, 256, 384), (750, 384, 512)]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("l,cin,cout", CFG_A_GEMMS)
def test_conv_cfgA_geometry_default_dispatch(dt, l, cin, cout):
    _conv_fwd_dgrad_wgrad(dt, 4, l, cin, cout)


def _window_slices(n):
    return sorted({0, n // 2, n - 1})   # first / last window and the first window of the second tower


# (fp32 storage -- f32, f32s -- on the first shape only: its kernels do not depend on the channel counts beyond % 128; wall-clock budget)
@pytest.mark.parametrize("dt,l,cin,cout", [(dt,) + g for g in CFG_A_GEMMS for dt in ("f32", "f32s", "bf16", "f16")
                                           if dt in ("bf16", "f16") or g == CFG_A_GEMMS[0]])
def test_conv_cfgA_full_batch_sampled_windows(dt, l, cin, cout):
    """The bench launch itself (256 windows = 128 pairs): forward + statistics and dgrad are per-window independent, so the
    oracle is run on a handful of sampled windows of the very same launch; wgrad sums over windows, so it is checked (a)
    by linearity -- du non-zero in the sampled windows only must give the oracle's wgrad of those windows, with the other
    251 windows' inputs random (they walk every split/slab of the launch) -- and (b) with a dense du against a float64
    torch.matmul restatement of the same sum on the device."""
    vm, tdt = DTYPES[dt]
    n = 256
    g = torch.Generator(device="cuda").manual_seed(77)
    xq = torch.randn(n, l, cin, device="cuda", generator=g).to(tdt)
    duq = torch.randn(n, l, cout, device="cuda", generator=g).to(tdt)
    r = rng(5)
    w = quant(r.normal(0, 0.1, (3, cin, cout)), dt)
    b = torch.tensor(r.normal(0, 0.3, (cout,)).astype(np.float32), dtype=torch.float64)
    wf = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda")
    wd = torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
    L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
    xp = torch.zeros(n, l + 2, cin, dtype=tdt, device="cuda")
    xp[:, 1:l + 1] = xq
    dup = torch.zeros(n, l + 2, cout, dtype=tdt, device="cuda")
    dup[:, 1:l + 1] = duq
    rows = L().query("vm_conv_stat_rows", l)
    z = torch.empty(n, l, cout, dtype=tdt, device="cuda")
    ss = torch.zeros(n * rows, cout, device="cuda")
    sq = torch.zeros(n * rows, cout, device="cuda")
    L().call("vm_conv_fwd", p(xp), p(wf), p(dev(b)), n, l, cin, cout, vm, p(z), p(ss), p(sq), stream())
    dx = torch.empty(n, l, cin, dtype=tdt, device="cuda")
    L().call("vm_conv_dgrad", p(dup), p(wd), n, l, cin, cout, vm, p(dx), stream())
    sel = _window_slices(n)
    pl_, pr_ = O.same_padding(3)
    for i in sel:
        xi = xq[i:i + 1].double().cpu()
        ref = _conv_ref(xi, w, b).numpy()
        zz = z[i:i + 1].float().cpu().numpy()
        assert rel_err(zz, ref) < TOL[dt], i
        assert rel_err(ss.view(n, rows, cout)[i].sum(0).cpu().numpy(), zz.astype(np.float64).sum((0, 1))) < 1e-5
        assert rel_err(sq.view(n, rows, cout)[i].sum(0).cpu().numpy(), (zz.astype(np.float64) ** 2).sum((0, 1))) < 1e-5
        dui = duq[i:i + 1].double().cpu()
        xr = xi.clone().requires_grad_(True)
        y = torch.nn.functional.conv1d(torch.nn.functional.pad(xr.transpose(1, 2), (pl_, pr_)), w.permute(2, 1, 0)).transpose(1, 2)
        gx, = torch.autograd.grad((y * dui).sum(), [xr])
        assert rel_err(dx[i:i + 1].float().cpu().numpy(), gx.numpy()) < TOL[dt], i
    # the statistics of the whole launch against the stored z (every window)
    zd = z.double()
    assert rel_err(ss.view(n, rows, cout).sum(1).cpu().numpy(), zd.sum(1).cpu().numpy()) < 1e-5
    assert rel_err(sq.view(n, rows, cout).sum(1).cpu().numpy(), (zd * zd).sum(1).cpu().numpy()) < 1e-5
    del zd
    # wgrad (a): linearity against the oracle
    ws = torch.empty(L().query("vm_conv_wgrad_workspace_bytes", n, l, cin, cout) // 4 + 16, device="cuda")
    gwd = torch.empty(3, cin, cout, device="cuda")
    dus = torch.zeros_like(dup)
    dus[sel] = dup[sel]
    L().call("vm_conv_wgrad", p(xp), p(dus), n, l, cin, cout, vm, p(ws), p(gwd), stream())
    xs = xq[sel].double().cpu()
    wr = w.clone().requires_grad_(True)
    y = torch.nn.functional.conv1d(torch.nn.functional.pad(xs.transpose(1, 2), (pl_, pr_)), wr.permute(2, 1, 0)).transpose(1, 2)
    gw, = torch.autograd.grad((y * duq[sel].double().cpu()).sum(), [wr])
    assert rel_err(gwd.cpu().numpy(), gw.numpy()) < (5e-5 if dt == "f32s" else 2e-5)
    # wgrad (b): dense du, float64 restatement of the sum on the device (torch.matmul, not a kernel of this repo)
    L().call("vm_conv_wgrad", p(xp), p(dup), n, l, cin, cout, vm, p(ws), p(gwd), stream())
    ref = torch.empty(3, cin, cout, dtype=torch.float64, device="cuda")
    dd = duq.double().reshape(n * l, cout)
    for k in range(3):
        ref[k] = xp[:, k:k + l].double().reshape(n * l, cin).t() @ dd
    assert rel_err(gwd.cpu().numpy(), ref.cpu().numpy()) < (2e-5 if dt == "f32" else 1e-4)


@pytest.mark.parametrize("i16", [False, True])
def test_crop_decimate_whiten_vs_oracle(i16):
    """vm_crop_decimate_whiten (device-side crop of voicemap/librispeech.py:103-137 + utils.py:22-34, 88-101) compared
    DIRECTLY with O.preprocess_instances on numpy-cropped windows (not via the host-crop kernel)."""
    r = rng(13)
    n, wpt, raw_len, ds = 6, 3, 4801, 4
    total = 60000
    audio = r.normal(0, 0.05, total) + 0.01 * np.sin(np.arange(total) / 900.0)
    if i16:
        a16 = np.clip(np.round(audio * 32768), -32768, 32767).astype(np.int16)
        audio = a16.astype(np.float64) / 32768.0
        ad = dev(a16, torch.int16)
    else:
        audio = audio.astype(np.float32)
        ad = dev(audio)
    offs = r.integers(0, total - raw_len, n).astype(np.int64)
    offs[0], offs[-1] = 0, total - raw_len
    l0 = (raw_len + ds - 1) // ds
    out = torch.zeros(n, l0 + 31, device="cuda")
    ws = torch.empty(L().query("vm_decimate_whiten_workspace_bytes", n) // 8, dtype=torch.float64, device="cuda")
    L().call("vm_crop_decimate_whiten", p(ad), int(i16), p(dev(offs, torch.int64)), n, raw_len, ds, 1, 0.038021, wpt, p(out),
             p(ws), stream())
    win = np.stack([audio[o:o + raw_len] for o in offs]).astype(np.float64)
    pre = O.preprocess_instances(ds)
    ref = np.concatenate([pre(win[t:t + wpt, :, None]) for t in range(0, n, wpt)])[:, :, 0]
    o = out.cpu().numpy()
    assert np.all(o[:, :15] == 0) and np.all(o[:, 15 + l0:] == 0)
    assert max_err(o[:, 15:15 + l0], ref) < 1e-7


# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,l,cin,cout,padded_a", [(3, 254, 128, 256, True), (2, 255, 128, 64, False), (2, 1000, 256, 384, True),
                                                   (5, 509, 384, 512, True), (2, 2000, 128, 256, False), (1, 130, 128, 32, True)])
@pytest.mark.parametrize("dt16", ["bf16", "f16"])
def test_conv_dgrad_bnred(n, l, cin, cout, padded_a, dt16):
    """vm_conv_dgrad_bnred: dx bit-identical to vm_conv_dgrad, and the partial rows sum to sum_t dx and sum_t dx * A taken over
    the stored (bf16) dx in float64.  Lengths either side of the 254-row tile edge, several channel tiles, both layouts of A
    with garbage in the halo rows (they must not be read into the sums)."""
    vm, tdt = DTYPES[dt16]
    if not L().query("vm_conv_dgrad_bnred_supported", n, l, cin, cout, vm):
        pytest.skip("shape not served by the 256 x 128 kernel under the current tuning")
    g = torch.Generator(device="cuda").manual_seed(l + cin)
    r = rng(9)
    w = quant(r.normal(0, 0.1, (3, cin, cout)), dt16)
    wf = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda")
    wd = torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
    L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
    dup = torch.zeros(n, l + 2, cout, dtype=tdt, device="cuda")
    dup[:, 1:l + 1] = torch.randn(n, l, cout, device="cuda", generator=g).to(tdt)
    a = torch.randn(n, l, cin, device="cuda", generator=g).add_(0.5).to(tdt)
    if padded_a:
        ap = torch.full((n, l + 2, cin), 1e30 if dt16 == "bf16" else 6e4, dtype=tdt, device="cuda")
        ap[:, 1:l + 1] = a
    else:
        ap = a.contiguous()
    dx0 = torch.empty(n, l, cin, dtype=tdt, device="cuda")
    L().call("vm_conv_dgrad", p(dup), p(wd), n, l, cin, cout, vm, p(dx0), stream())
    rows = L().query("vm_conv_dgrad_bnred_rows", l)
    s0 = torch.full((n * rows, cin), float("nan"), device="cuda")
    s1 = torch.full((n * rows, cin), float("nan"), device="cuda")
    dx = torch.empty(n, l, cin, dtype=tdt, device="cuda")
    L().call("vm_conv_dgrad_bnred", p(dup), p(wd), n, l, cin, cout, vm, p(dx), p(ap), int(padded_a), p(s0), p(s1), None, stream())
    assert torch.equal(dx, dx0)
    if L().query("vm_pack_nt_weights_supported", cin, cout, vm):
        # packed weights (conv_nt3_kernel for 128 / 256 / 384 / 512 channels on the K side): bit-identical dx and partial rows
        wdp = torch.empty_like(wd)
        L().call("vm_pack_nt_weights", p(wd), 1, cin, cout, vm, p(wdp), stream())
        dx3, s03, s13 = torch.empty_like(dx), torch.full_like(s0, float("nan")), torch.full_like(s1, float("nan"))
        L().call("vm_conv_dgrad_bnred", p(dup), p(wd), n, l, cin, cout, vm, p(dx3), p(ap), int(padded_a), p(s03), p(s13), p(wdp), stream())
        assert torch.equal(dx3, dx) and torch.equal(s03, s0) and torch.equal(s13, s1)
    d64, a64 = dx.double(), a.double()
    ref0, ref1 = d64.sum(1).cpu().numpy(), (d64 * a64).sum(1).cpu().numpy()
    got0, got1 = s0.view(n, rows, cin).double().sum(1).cpu().numpy(), s1.view(n, rows, cin).double().sum(1).cpu().numpy()
    assert np.isfinite(got0).all() and np.isfinite(got1).all()
    scale0 = float(d64.abs().sum(1).max())
    scale1 = float((d64 * a64).abs().sum(1).max())
    assert np.abs(got0 - ref0).max() < 2e-6 * scale0
    assert np.abs(got1 - ref1).max() < 2e-6 * scale1


@pytest.mark.parametrize("towers,rows,ac", [(1, 128, 32), (2, 256, 128), (1, 384, 256), (1, 512, 384)])
@pytest.mark.parametrize("dt16", ["bf16", "f16"])
def test_pack_nt_weights_is_the_fragment_permutation(towers, rows, ac, dt16):
    """vm_pack_nt_weights: (towers, rows, 3 * a_c) -> [tower][rows / 64][K tile = 3 * chunk + tap][32-row half j][16-channel half ks]
    [lane = 32 * kh + r][8 values], element (row, tap * a_c + c) with row = 64 b + 32 j + r, c = 32 chunk + 16 ks + 8 kh + e."""
    vm, tdt = DTYPES[dt16]
    assert L().query("vm_pack_nt_weights_supported", rows, ac, vm) == 1
    assert L().query("vm_pack_nt_weights_supported", rows + 64, ac, vm) == 0 and L().query("vm_pack_nt_weights_supported", rows, ac + 8, vm) == 0
    assert L().query("vm_pack_nt_weights_supported", rows, ac, DTYPES["f32"][0]) == 0
    src = torch.arange(towers * rows * 3 * ac, device="cuda").remainder(2039).to(tdt).view(towers, rows, 3, ac)   # exact in both types
    out = torch.empty(towers * rows * 3 * ac, dtype=tdt, device="cuda")
    L().call("vm_pack_nt_weights", p(src), towers, rows, ac, vm, p(out), stream())
    want = (src.view(towers, rows // 64, 2, 32, 3, ac // 32, 2, 2, 8)      # t, b, j, r, tap, chunk, ks, kh, e
            .permute(0, 1, 5, 4, 2, 6, 7, 3, 8).contiguous().view(-1))    # t, b, chunk, tap, j, ks, kh, r, e
    assert torch.equal(out, want)
    with pytest.raises(RuntimeError):
        L().call("vm_pack_nt_weights", p(src), towers, rows, ac + 8, vm, p(out), stream())


def test_conv_dgrad_bnred_refuses_unserved_shapes():
    vm, _ = DTYPES["bf16"]
    assert L().query("vm_conv_dgrad_bnred_supported", 2, 300, 136, 64, vm) == 0  # c_in not a multiple of 128
    assert L().query("vm_conv_dgrad_bnred_supported", 2, 300, 128, 64, DTYPES["f32"][0]) == 0
    assert L().query("vm_conv_dgrad_bnred_supported", 2, 260, 128, 64, vm) == 0  # a 256-row tile would be half padding
    assert L().query("vm_conv_dgrad_bnred_supported", 2, 500, 128, 64, vm) == 1
    d = torch.zeros(16, device="cuda")
    with pytest.raises(RuntimeError):
        L().call("vm_conv_dgrad_bnred", p(d), p(d), 2, 300, 136, 64, vm, p(d), p(d), 1, p(d), p(d), None, stream())


@pytest.mark.parametrize("n,wpt,l,cin,cout,pool,use_drop", [(4, 2, 508, 128, 256, 2, True), (2, 1, 1016, 256, 128, 4, False),
                                                            (6, 3, 500, 128, 64, 1, True)])
@pytest.mark.parametrize("dt16", ["bf16", "f16"])
def test_bn_bwd_from_sums_equals_pooled_reduce(n, wpt, l, cin, cout, pool, use_drop, dt16):
    """dgrad + fused sums + vm_bn_bwd_from_sums against dgrad + vm_bn_pool_bwd_reduce_pooled: the same (sum dy, sum dy*zhat) per
    window up to fp32 summation order, including a channel with scale == 0 (re-derived from z) and dropped channels."""
    vm, tdt = DTYPES[dt16]
    lq = l // pool
    if not L().query("vm_conv_dgrad_bnred_supported", n, lq, cin, cout, vm):
        pytest.skip("shape not served under the current tuning")
    r = rng(43)
    g = torch.Generator(device="cuda").manual_seed(3)
    towers = n // wpt
    z = quant(np.maximum(r.normal(0.3, 1.0, (n, l, cin)), 0.0), dt16).to("cuda", tdt).contiguous()
    sc = r.normal(1.0, 0.3, (towers, cin)) * np.where(r.random((towers, cin)) < 0.3, -1, 1)
    sc[:, 3] = 0.0
    scale, shift = dev(sc), dev(r.normal(0, 0.3, (towers, cin)))
    mean, invstd = dev(r.normal(0.4, 0.1, (towers, cin))), dev(r.uniform(0.5, 2.0, (towers, cin)))
    drop = dev((r.random((n, cin)) > 0.25) / 0.75) if use_drop else None
    act = torch.zeros(n, lq + 2, cin, dtype=tdt, device="cuda")
    L().call("vm_bn_drop_pool_fwd", p(z), p(scale), p(shift), p(drop), n, wpt, l, cin, pool, vm, p(act), stream())
    w = quant(r.normal(0, 0.1, (3, cin, cout)), dt16)
    wf = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda")
    wd = torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
    L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
    dup = torch.zeros(n, lq + 2, cout, dtype=tdt, device="cuda")
    dup[:, 1:lq + 1] = torch.randn(n, lq, cout, device="cuda", generator=g).to(tdt)
    rows2 = L().query("vm_conv_dgrad_bnred_rows", lq)
    s0, s1 = (torch.empty(n * rows2, cin, device="cuda") for _ in range(2))
    dp = torch.empty(n, lq, cin, dtype=tdt, device="cuda")
    L().call("vm_conv_dgrad_bnred", p(dup), p(wd), n, lq, cin, cout, vm, p(dp), p(act), 1, p(s0), p(s1), None, stream())
    rows = L().query("vm_bn_part_rows")
    pa0, pb0, pa1, pb1 = (torch.zeros(n * rows, cin, device="cuda") for _ in range(4))
    L().call("vm_bn_pool_bwd_reduce_pooled", p(z), p(act), p(dp), p(scale), p(shift), p(mean), p(invstd), p(drop), n, wpt, l, cin,
             pool, vm, p(pa0), p(pb0), stream())
    L().call("vm_bn_bwd_from_sums", p(s0), p(s1), rows2, p(z), p(dp), p(scale), p(shift), p(mean), p(invstd), p(drop), n, wpt, l, cin,
             pool, vm, 1, p(pa1), p(pb1), stream())
    a0, b0 = (t.view(n, rows, cin).double().sum(1).cpu().numpy() for t in (pa0, pb0))
    a1, b1 = (t.view(n, rows, cin).double().sum(1).cpu().numpy() for t in (pa1, pb1))
    # the terms cancel, so errors are measured against what was added up: sum |dp| * |ext| * drop * invstd
    mag = float(dp.double().abs().sum(1).max()) * 4.0 * 2.0 * (1 / 0.75 if use_drop else 1.0)
    assert np.abs(a1 - a0).max() < 2e-6 * mag
    # the reference kernel sends the whole 8-channel vector of the scale == 0 channel through z (no storage rounding of act):
    # those channels agree to the bf16 level only, every other channel to fp32 summation order, the scale == 0 channel itself
    # (re-derived from z in both) again to summation order
    vec = np.zeros(cin, bool)
    vec[0:8] = True
    assert np.abs(b1 - b0)[:, ~vec].max() < 2e-5 * mag
    assert np.abs(b1 - b0)[:, 3].max() < 2e-5 * mag
    assert rel_err(b1[:, vec], b0[:, vec]) < 1.5e-2
    # the fused form (sums -> per-window map -> fp64 column reduction in one launch, then the finalize): the same c1 / c2 and BatchNorm
    # parameter gradients as vm_bn_bwd_from_sums + vm_bn_bwd_finalize, up to the summation order (fp64 here, fp32 per window there)
    f32 = dict(dtype=torch.float32, device="cuda")
    crws = torch.empty(L().query("vm_colreduce_workspace_bytes", towers, cin) // 8, dtype=torch.float64, device="cuda")
    outs = []
    for fused in (False, True):
        c1, c2, gg, gb = torch.empty(towers, cin, **f32), torch.empty(towers, cin, **f32), torch.empty(cin, **f32), torch.empty(cin, **f32)
        if fused:
            L().call("vm_bn_bwd_from_sums_finalize", p(s0), p(s1), rows2, p(z), p(dp), p(scale), p(shift), p(mean), p(invstd), p(drop), n,
                     wpt, l, cin, pool, vm, 1, float(wpt * l), p(c1), p(c2), p(gg), p(gb), p(crws), stream())
        else:
            L().call("vm_bn_bwd_finalize", p(pa1), p(pb1), n, wpt, cin, float(wpt * l), p(c1), p(c2), p(gg), p(gb), p(crws), stream())
        outs.append([t.double().cpu().numpy() for t in (c1, c2, gg, gb)])
    for u, v, scale_ in zip(outs[0], outs[1], (mag / (wpt * l), mag / (wpt * l), mag, mag)):
        assert np.abs(u - v).max() < 1e-4 * scale_


def test_prep_conv_weights_batch_equals_single():
    import ctypes
    vm, tdt = DTYPES["bf16"]
    r = rng(12)
    shapes = [(128, 256), (256, 384), (8, 24)]
    ws = [dev(r.normal(0, 0.1, (3, ci, co))) for ci, co in shapes]
    single = []
    for w, (ci, co) in zip(ws, shapes):
        wf, wd = torch.zeros(co * 3 * ci, dtype=tdt, device="cuda"), torch.zeros(ci * 3 * co, dtype=tdt, device="cuda")
        L().call("vm_prep_conv_weights", p(w), ci, co, vm, p(wf), p(wd), stream())
        single.append((wf, wd))
    wfs = [torch.zeros_like(a) for a, _ in single]
    wds = [torch.zeros_like(b) for _, b in single]
    wts = [torch.zeros(a.numel(), dtype=torch.float32, device="cuda") for a, _ in single]
    vp, ci_t = ctypes.c_void_p * 3, ctypes.c_int * 3
    L().call("vm_prep_conv_weights_batch", 3, vp(*[p(w) for w in ws]), ci_t(*[s[0] for s in shapes]), ci_t(*[s[1] for s in shapes]), vm,
             vp(*[p(t) for t in wfs]), vp(*[p(t) for t in wds]), vp(p(wts[0]), None, p(wts[2])), stream())
    for (a, b), a2, b2 in zip(single, wfs, wds):
        assert torch.equal(a, a2) and torch.equal(b, b2)
    # wt: the fp32 kernel itself in wf's layout (c_out, 3 * c_in); a NULL entry is skipped
    for k in (0, 2):
        ci, co = shapes[k]
        assert torch.equal(wts[k].view(co, 3, ci), ws[k].permute(2, 0, 1).contiguous())
    assert not wts[1].any()
    L().call("vm_prep_conv_weights_batch", 3, vp(*[p(w) for w in ws]), ci_t(*[s[0] for s in shapes]), ci_t(*[s[1] for s in shapes]), vm,
             vp(*[p(t) for t in wfs]), vp(*[p(t) for t in wds]), None, stream())
    with pytest.raises(RuntimeError):
        L().call("vm_prep_conv_weights_batch", 9, vp(), ci_t(), ci_t(), vm, vp(), vp(), None, stream())


@pytest.mark.parametrize("n,l,cin,cout", [(3, 254, 128, 256), (2, 256, 32, 128), (2, 1000, 256, 384), (5, 750, 384, 512), (4, 3000, 128, 256),
                                          (1, 508, 64, 128)])
@pytest.mark.parametrize("dt16", ["bf16", "f16"])
def test_conv_fwd_pool_equals_two_kernel_inference_path(n, l, cin, cout, dt16):
    """vm_conv_fwd_pool (conv + ReLU + BatchNorm affine + MaxPool1D(2) in the GEMM epilogue, inference mode) is bit-identical to
    vm_conv_fwd followed by vm_bn_drop_pool_fwd, leaves the halo rows of the pooled tensor alone, and both agree with the oracle."""
    vm, tdt = DTYPES[dt16]
    assert L().query("vm_conv_fwd_pool_supported", n, l, cin, cout, vm) == 1
    r = rng(21)
    x = quant(r.normal(0, 1.0, (n, l, cin)), dt16)
    w = quant(r.normal(0, 0.1, (3, cin, cout)), dt16)
    b = torch.tensor(r.normal(0, 0.3, (cout,)).astype(np.float32), dtype=torch.float64)
    scale = dev(r.normal(1.0, 0.3, (1, cout)) * np.where(r.random((1, cout)) < 0.3, -1, 1))
    shift = dev(r.normal(0, 0.3, (1, cout)))
    wf = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda")
    wd = torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
    L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
    xp = padded(x, tdt)
    z = torch.empty(n, l, cout, dtype=tdt, device="cuda")
    L().call("vm_conv_fwd", p(xp), p(wf), p(dev(b)), n, l, cin, cout, vm, p(z), None, None, stream())
    lq = l // 2
    a0 = torch.zeros(n, lq + 2, cout, dtype=tdt, device="cuda")
    L().call("vm_bn_drop_pool_fwd", p(z), p(scale), p(shift), None, n, n, l, cout, 2, vm, p(a0), stream())
    a1 = torch.full((n, lq + 2, cout), 7.0, dtype=tdt, device="cuda")
    L().call("vm_conv_fwd_pool", p(xp), p(wf), p(dev(b)), p(scale), p(shift), n, l, cin, cout, vm, p(a1), None, stream())
    if L().query("vm_pack_nt_weights_supported", cout, cin, vm):   # packed weights: the same bits
        wfp = torch.empty_like(wf)
        L().call("vm_pack_nt_weights", p(wf), 1, cout, cin, vm, p(wfp), stream())
        a3 = torch.full((n, lq + 2, cout), 7.0, dtype=tdt, device="cuda")
        L().call("vm_conv_fwd_pool", p(xp), p(wf), p(dev(b)), p(scale), p(shift), n, l, cin, cout, vm, p(a3), p(wfp), stream())
        assert torch.equal(a3, a1)
    if 2 * ((l + 253) // 254) == (l + 127) // 128:
        # vm_conv_fwd runs the same kernel (conv_nt2r_kernel: the same K walk, the same accumulation order) -- every cfg-A layer
        assert torch.equal(a1[:, 1:lq + 1], a0[:, 1:lq + 1])
    else:
        # (l = 256, 508) vm_conv_fwd falls back to the 128 x 128 kernel, which walks K (tap, chunk) instead of (chunk, tap): z can
        # differ in the last fp32 bit before it is rounded to storage
        assert rel_err(a1[:, 1:lq + 1].float().cpu().numpy(), a0[:, 1:lq + 1].float().cpu().numpy()) < TOL[dt16]
    assert (a1[:, 0] == 7.0).all() and (a1[:, lq + 1] == 7.0).all()
    ref = _conv_ref(x, w, b)
    y = ref * scale.double().cpu()[0] + shift.double().cpu()[0]
    pooled = torch.maximum(y[:, 0:2 * lq:2], y[:, 1:2 * lq:2]).numpy()
    assert rel_err(a1[:, 1:lq + 1].float().cpu().numpy(), pooled) < 2e-2


def test_conv_fwd_pool_refuses_unserved_shapes():
    vm, _ = DTYPES["bf16"]
    assert L().query("vm_conv_fwd_pool_supported", 2, 301, 128, 128, vm) == 0   # odd length
    assert L().query("vm_conv_fwd_pool_supported", 2, 300, 128, 136, vm) == 0   # c_out not a multiple of 128
    assert L().query("vm_conv_fwd_pool_supported", 2, 300, 128, 128, DTYPES["f32"][0]) == 0
    d = torch.zeros(16, device="cuda")
    with pytest.raises(RuntimeError):
        L().call("vm_conv_fwd_pool", p(d), p(d), p(d), p(d), p(d), 2, 301, 128, 128, vm, p(d), None, stream())


@pytest.mark.parametrize("n,l,cin,cout", [(3, 508, 128, 256), (2, 254, 32, 128), (2, 1016, 256, 384), (4, 3000, 128, 256), (4, 1500, 256, 384)])
@pytest.mark.parametrize("dt16", ["bf16", "f16"])
def test_conv_fwd_e_pool_extreme(n, l, cin, cout, dt16):
    """vm_conv_fwd_e: z and the statistics bit-identical to vm_conv_fwd, e = the pair maximum where gamma >= 0 and the pair minimum where
    gamma < 0 of the stored z, and vm_bn_drop_pool_fwd(e, pool 1) == vm_bn_drop_pool_fwd(z, pool 2) bit for bit (negative scales and
    dropped channels included)."""
    vm, tdt = DTYPES[dt16]
    assert L().query("vm_conv_fwd_e_supported", n, l, cin, cout, vm) == 1
    r = rng(23)
    x = quant(r.normal(0, 1.0, (n, l, cin)), dt16)
    w = quant(r.normal(0, 0.1, (3, cin, cout)), dt16)
    b = dev(r.normal(0, 0.3, (cout,)))
    gamma = r.normal(1.0, 0.3, (cout,)) * np.where(r.random(cout) < 0.4, -1, 1)
    gamma[5] = 0.0
    wf = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda")
    wd = torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
    L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
    xp = padded(x, tdt)
    rows = L().query("vm_conv_stat_rows", l)
    z0, z1 = (torch.empty(n, l, cout, dtype=tdt, device="cuda") for _ in range(2))
    s0, q0, s1, q1 = (torch.zeros(n * rows, cout, device="cuda") for _ in range(4))
    lq = l // 2
    e = torch.full((n, lq, cout), -5.0, dtype=tdt, device="cuda")
    L().call("vm_conv_fwd", p(xp), p(wf), p(b), n, l, cin, cout, vm, p(z0), p(s0), p(q0), stream())
    L().call("vm_conv_fwd_e", p(xp), p(wf), p(b), p(dev(gamma)), n, l, cin, cout, vm, p(z1), p(s1), p(q1), p(e), stream())
    assert torch.equal(z0, z1) and torch.equal(s0, s1) and torch.equal(q0, q1)
    zz = z0.float()
    hi, lo = torch.maximum(zz[:, 0::2], zz[:, 1::2]), torch.minimum(zz[:, 0::2], zz[:, 1::2])
    want = torch.where(torch.tensor(gamma < 0, device="cuda")[None, None, :], lo, hi)
    assert torch.equal(e.float(), want)
    # the pass over e gives the pooled output of the pass over z (per-tower scale with the sign of gamma, dropout mask)
    invstd = r.uniform(0.5, 2.0, (1, cout))
    scale, shift = dev(gamma[None, :] * invstd), dev(r.normal(0, 0.3, (1, cout)))
    drop = dev((r.random((n, cout)) > 0.25) / 0.75)
    a0 = torch.zeros(n, lq + 2, cout, dtype=tdt, device="cuda")
    a1 = torch.zeros_like(a0)
    L().call("vm_bn_drop_pool_fwd", p(z0), p(scale), p(shift), p(drop), n, n, l, cout, 2, vm, p(a0), stream())
    L().call("vm_bn_drop_pool_fwd", p(e), p(scale), p(shift), p(drop), n, n, lq, cout, 1, vm, p(a1), stream())
    assert torch.equal(a0, a1)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("n,l,cin,cout", [(40, 37, 96, 64), (7, 149, 192, 96), (64, 5, 24, 128), (3, 300, 32, 32)])
def test_conv_fwd_flat_equals_per_window_forward(dt, n, l, cin, cout):
    """vm_conv_fwd_flat -- many short windows, each with its own zero halo rows, run as one sequence on full 128-row tiles with the
    halo positions dropped by the epilogue -- gives vm_conv_fwd's z bit for bit (the same products in the same order per output) and
    the same BatchNorm sums over all windows (one partial row per tile of the concatenation instead of per window)."""
    vm, tdt = DTYPES[dt]
    r = rng(n + l)
    x = quant(r.normal(0, 1, (n, l, cin)), dt).numpy()
    w = r.normal(0, 0.3 / np.sqrt(3 * cin), (3, cin, cout)).astype(np.float32)
    bias = r.normal(0, 0.1, cout).astype(np.float32)
    xin = padded(x, tdt)
    wf, wd = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda"), torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
    L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
    rows = L().query("vm_conv_stat_rows", l)
    z1 = torch.empty(n, l, cout, dtype=tdt, device="cuda")
    s1, q1 = torch.empty(n * rows, cout, device="cuda"), torch.empty(n * rows, cout, device="cuda")
    L().call("vm_conv_fwd", p(xin), p(wf), p(dev(bias)), n, l, cin, cout, vm, p(z1), p(s1), p(q1), stream())
    frows = L().query("vm_conv_flat_stat_rows", n, l)
    assert frows == -(-(n * (l + 2) - 2) // 128)
    z2 = torch.full((n, l, cout), 7.0, dtype=tdt, device="cuda")
    s2, q2 = torch.empty(frows, cout, device="cuda"), torch.empty(frows, cout, device="cuda")
    L().call("vm_conv_fwd_flat", p(xin), p(wf), p(dev(bias)), n, l, cin, cout, vm, p(z2), p(s2), p(q2), stream())
    torch.cuda.synchronize()
    assert torch.equal(z1, z2)
    assert np.allclose(s1.sum(0).cpu().numpy(), s2.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    assert np.allclose(q1.sum(0).cpu().numpy(), q2.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # inference form: no statistics
    z3 = torch.empty_like(z2)
    L().call("vm_conv_fwd_flat", p(xin), p(wf), p(dev(bias)), n, l, cin, cout, vm, p(z3), None, None, stream())
    torch.cuda.synchronize()
    assert torch.equal(z1, z3)


@pytest.mark.parametrize("head,loss,pairs,c,e,parts", [("uniform_euclidean", "contrastive", 6, 64, 32, True), ("weighted_l1", "bce", 5, 96, 128, True),
                                                       ("uniform_euclidean", "bce", 128, 512, 64, True), ("weighted_l1", "contrastive", 3, 40, 24, False)])
def test_tail_fwd_bwd_equals_the_six_launches_and_the_oracle(head, loss, pairs, c, e, parts):
    """vm_tail_fwd_bwd + vm_tail_param_grads (voicemap/models.py:37-39,55-69, utils.py:77-85) against the launches they
    replace -- gmax segment finish, vm_dense_fwd, vm_siamese_head_loss (+ reduce), vm_dense_bwd -- bit for bit, and against the float64
    oracle (GlobalMaxPool1D values given -> Dense -> head -> loss and every gradient)."""
    from voicemap_amd.engine import HEADS, LOSSES
    from tests.gpu_util import grad_close
    r = rng(21)
    n, seg, gs = 2 * pairs, L().query("vm_bn_part_rows"), 4096.0
    # segment partials: per (window, segment, channel) a value and a position; some segments empty (position 0x7fffffff, value -inf)
    pv = r.normal(0, 1, (n, seg, c)).astype(np.float32)
    pi = r.integers(0, 300, (n, seg, c)).astype(np.int32)
    empty = r.random((n, seg, c)) < 0.2
    empty[:, 0, :] = False
    pv[empty], pi[empty] = -np.inf, 0x7fffffff
    tie = r.random((n, c)) < 0.1                       # equal maxima in two segments: the smaller position wins
    pv[:, 1, :][tie] = pv[:, 0, :][tie]
    empty[:, 1, :][tie] = False
    pi[:, 1, :][tie] = pi[:, 0, :][tie] + 7
    best = pv.max(axis=1)
    want_idx = np.where(pv == best[:, None, :], pi, 0x7fffffff).min(axis=1)
    dw = (r.normal(0, 0.4, (c, e)) / np.sqrt(c)).astype(np.float32)       # embeddings ~N(0, 0.4): the sigmoid stays off its clip
    db = r.normal(0, 0.1, (e,)).astype(np.float32)
    hw = r.normal(0.5, 0.3, (1, 1) if head == "uniform_euclidean" else (e, 1)).astype(np.float32)
    hb = r.normal(-0.5, 0.1, (1,)).astype(np.float32)
    if head == "weighted_l1":
        hw /= np.sqrt(e / 16.0)
    y = (r.random(pairs) > 0.5).astype(np.float32)
    d_dw, d_db, d_hw, d_hb, d_y = dev(dw), dev(db), dev(hw), dev(hb), dev(y)
    f = lambda *shape: torch.empty(*shape, device="cuda")
    # --- fused
    gmax, gidx = f(n, c), torch.empty(n, c, dtype=torch.int32, device="cuda")
    if parts:
        d_pv, d_pi = dev(pv), dev(pi, torch.int32)
        a_pv, a_pi = p(d_pv), p(d_pi)
    else:
        gmax.copy_(torch.as_tensor(best))
        a_pv = a_pi = None
    emb, pred, demb, dgmax, ws = f(n, e), f(pairs), f(n, e), f(n, c), f(4 * pairs)
    assert L().query("vm_tail_fwd_bwd_supported", c, e) == 1 and L().query("vm_tail_fwd_bwd_supported", 2048, e) == 0
    L().call("vm_tail_fwd_bwd", a_pv, a_pi, seg, p(gmax), p(gidx), p(d_dw), p(d_db), p(d_hw), p(d_hb), p(d_y), pairs, c, e, HEADS[head],
             LOSSES[loss], gs, p(emb), p(pred), p(demb), p(dgmax), p(ws), stream())
    la, g_dw, g_db, g_hw, g_hb = f(2), f(c, e), f(e), f(hw.size), f(1)
    L().call("vm_tail_param_grads", p(gmax), p(demb), p(emb), p(ws), pairs, c, e, HEADS[head], p(la), p(g_dw), p(g_db), p(g_hw), p(g_hb), stream())
    if parts:
        assert np.array_equal(gmax.cpu().numpy(), best) and np.array_equal(gidx.cpu().numpy(), want_idx)
    # --- the launches it replaces
    emb2, pred2, demb2, dgmax2, ws2 = f(n, e), f(pairs), f(n, e), f(n, c), f(4 * pairs)
    la2, g_dw2, g_db2, g_hw2, g_hb2 = f(2), f(c, e), f(e), f(hw.size), f(1)
    L().call("vm_dense_fwd", p(gmax), p(d_dw), p(d_db), n, c, e, p(emb2), stream())
    L().call("vm_siamese_head_loss", p(emb2), p(d_hw), p(d_hb), p(d_y), pairs, e, HEADS[head], LOSSES[loss], gs, p(pred2), p(la2), p(demb2),
             p(g_hw2), p(g_hb2), p(ws2), stream())
    L().call("vm_dense_bwd", p(gmax), p(d_dw), p(demb2), n, c, e, p(g_dw2), p(g_db2), p(dgmax2), stream())
    for a, b_, nm in ((emb, emb2, "emb"), (pred, pred2, "pred"), (demb, demb2, "demb"), (dgmax, dgmax2, "dgmax"), (la, la2, "loss_acc"),
                      (g_dw, g_dw2, "grad dense w"), (g_db, g_db2, "grad dense b"), (g_hw, g_hw2, "grad head w"), (g_hb, g_hb2, "grad head b")):
        assert torch.equal(a, b_), nm
    # --- the oracle
    gt = torch.tensor(best, dtype=torch.float64, requires_grad=True)
    prm = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in
           (("dense.kernel", dw), ("dense.bias", db), ("head.kernel", hw), ("head.bias", hb))}
    et = gt @ prm["dense.kernel"] + prm["dense.bias"]
    pr = O.siamese_head(prm, et[:pairs], et[pairs:], head)
    yt = torch.tensor(y, dtype=torch.float64)[:, None]
    lo = O.contrastive_loss(yt, pr) if loss == "contrastive" else O.binary_crossentropy(yt, pr)
    gg, gdw, gdb, ghw, ghb = torch.autograd.grad(lo, [gt, prm["dense.kernel"], prm["dense.bias"], prm["head.kernel"], prm["head.bias"]])
    assert rel_err(emb.cpu().numpy(), et.detach().numpy()) < 1e-5 and rel_err(pred.cpu().numpy(), pr.detach().numpy()[:, 0]) < 1e-5
    assert abs(la[0].item() - lo.item()) < 2e-5 * max(1.0, abs(lo.item())) and abs(la[1].item() - O.binary_accuracy(yt, pr).item()) < 1e-6
    assert rel_err(dgmax.cpu().numpy() / gs, gg.numpy()) < 1e-4 and rel_err(g_dw.cpu().numpy() / gs, gdw.numpy()) < 1e-4
    assert grad_close(g_db.cpu().numpy() / gs, gdb.numpy(), 1e-4, atol=1e-6)     # the twin towers' contributions cancel: ~0
    assert rel_err(g_hw.cpu().numpy() / gs, ghw.numpy().ravel()) < 1e-4 and rel_err(g_hb.cpu().numpy() / gs, ghb.numpy()) < 1e-4
