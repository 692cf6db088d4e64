"""-m gpu parity of the folded-BatchNorm training path (vm_fold_bn_weights, vm_conv_fwd_fold, vm_du_tower_sums, vm_conv_wgrad_fold and the
engine mode that strings them together: ``HipEncoderEngine.fold_affine``).

What is folded: models.py:20-35 of the reference puts BatchNormalization -> SpatialDropout1D -> MaxPool1D between two Conv1D layers.
With dropout rate 0 the pooled BatchNorm output is y = scale[c] * e + shift[c], e = the pool-window extreme of the conv output, so the
next conv can run on e with W * scale as weights and the shift as per-tap constants -- except in the SAME padding, where y (not e) is
0.  The tests compare every kernel with that definition evaluated in float64 on the same (storage-rounded) operands, and the whole
training step with the float64 oracle and with the un-folded engine path."""
import numpy as np
import pytest
import torch

from oracle import voicemap_oracle as O
from tests.gpu_util import DTYPES, L, dev, grad_close, p, padded, quant, rel_err, report, stream

pytestmark = pytest.mark.gpu

DT16 = ["f16", "bf16"]


def _fold_weights(w, s, h, dtype, bias=None, packed=False):
    """w (3, c_in, c_out) Keras kernel, s / h (towers, c_in) or (c_in,) -> wf (towers, c_out, 3 * c_in), hb (towers, 4, c_out): rows
    0..2 the per-tap constants, row 3 = bias + their sum (``packed``: also the fragment-order copy vm_fold_bn_weights writes itself)."""
    vm, tdt = DTYPES[dtype]
    cin, cout = w.shape[1], w.shape[2]
    s, h = np.atleast_2d(s), np.atleast_2d(h)
    towers = s.shape[0]
    bias = np.zeros(cout, np.float32) if bias is None else bias
    wt = np.ascontiguousarray(w.transpose(2, 0, 1).reshape(cout, 3 * cin))   # the `wt` output of vm_prep_conv_weights_batch
    wf = torch.empty(towers, cout, 3 * cin, dtype=tdt, device="cuda")
    wfp = torch.empty(towers, cout, 3 * cin, dtype=tdt, device="cuda") if packed else None
    hb = torch.empty(towers, 4, cout, dtype=torch.float32, device="cuda")
    L().call("vm_fold_bn_weights", p(dev(wt)), p(dev(s)), p(dev(h)), p(dev(bias)), towers, cin, cout, vm, p(wf), p(wfp), p(hb), None, stream())
    torch.cuda.synchronize()
    return (wf, hb, wfp) if packed else (wf, hb)


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("cin,cout", [(128, 256), (40, 24)])
def test_fold_bn_weights(dtype, cin, cout):
    r = np.random.default_rng(cin + cout)
    w = r.normal(0, 0.1, (3, cin, cout)).astype(np.float32)
    s = r.normal(0, 2.0, (2, cin)).astype(np.float32)
    h = r.normal(0, 1.0, (2, cin)).astype(np.float32)
    bias = r.normal(0, 0.3, cout).astype(np.float32)
    wf, hb = _fold_weights(w, s, h, dtype, bias)
    for t in range(2):
        # the fp32 product rounded to the storage type, or -- where the compiler picks a mixed-precision fma (v_fma_mixlo_f16) -- the
        # exact product rounded once; the two differ only on double-rounding ties
        want = quant((w * s[t][None, :, None]).astype(np.float32), dtype).numpy()
        want1 = quant(w.astype(np.float64) * s[t][None, :, None].astype(np.float64), dtype).numpy()
        got = wf[t].to(torch.float64).cpu().numpy().reshape(cout, 3, cin).transpose(1, 2, 0)  # (c_out, 3 * c_in) -> (3, c_in, c_out)
        # (half: a product in the subnormal range, < 6.1e-5, may also come out flushed or one subnormal step away)
        ok = (got == want) | (got == want1) | ((np.abs(want) < 6.2e-5) & (np.abs(got - want) < 6.2e-5))
        ulp = (2.0 ** -10 if dtype == "f16" else 2.0 ** -7) * np.abs(want)
        assert ok.mean() > 0.999 and (np.abs(got - want) <= ulp + 6.2e-5).all(), (got[~ok][:8], want[~ok][:8], want1[~ok][:8])
        hb_ref = np.einsum("kio,i->ko", w.astype(np.float64), h[t].astype(np.float64))
        hbt = hb[t].cpu().numpy()
        assert np.abs(hbt[:3] - hb_ref).max() < 1e-5 * max(1.0, np.abs(hb_ref).max())
        assert np.array_equal(hbt[3], bias + ((hbt[0] + hbt[1]) + hbt[2]))      # the accumulators' start value, in the kernel's own order
    if L().query("vm_pack_nt_weights_supported", cout, cin, DTYPES[dtype][0]):
        # the fragment-order copy written by the fold kernel itself == vm_pack_nt_weights of its row-major output
        wf2, hb2, wfp = _fold_weights(w, s, h, dtype, bias, packed=True)
        want = torch.empty_like(wf2)
        L().call("vm_pack_nt_weights", p(wf2), 2, cout, cin, DTYPES[dtype][0], p(want), stream())
        assert torch.equal(wf2, wf) and torch.equal(hb2, hb) and torch.equal(wfp.view(-1), want.view(-1))


def _fold_case(r, n, Lw, cin, cout, dtype):
    e = np.abs(r.normal(0, 1.0, (n, Lw, cin)))                 # a pool extreme of relu(conv) is >= 0
    e = quant(e, dtype).numpy()
    w = r.normal(0, 0.6 / np.sqrt(3 * cin), (3, cin, cout)).astype(np.float32)
    bias = r.normal(0, 0.1, cout).astype(np.float32)
    s = (r.normal(1.0, 0.3, cin) * np.where(r.random(cin) < 0.2, -1, 1)).astype(np.float32)
    h = r.normal(0.0, 0.7, cin).astype(np.float32)             # large on purpose: a missing edge correction would be a gross error
    gamma = (r.normal(1.0, 0.2, cout) * np.where(r.random(cout) < 0.25, -1, 1)).astype(np.float32)
    return e, w, bias, s, h, gamma


def _conv_same(y, w):
    """y (n, L, c_in) float64, w (3, c_in, c_out) -> (n, L, c_out): Conv1D(3, padding='same') of the reference (cross-correlation)."""
    n, Lw, _ = y.shape
    yp = np.zeros((n, Lw + 2, y.shape[2]))
    yp[:, 1:Lw + 1] = y
    return sum(np.einsum("nli,io->nlo", yp[:, k:k + Lw], w[k]) for k in range(3))


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("n,Lw,cin,cout,with_e", [(3, 508, 128, 256, True), (2, 254, 256, 128, False), (2, 1016, 64, 128, True),
                                                  (1, 300, 32, 128, True)])
def test_conv_fwd_fold_matches_definition(dtype, n, Lw, cin, cout, with_e):
    vm, tdt = DTYPES[dtype]
    if not L().query("vm_conv_fwd_fold_supported", n, Lw, cin, cout, vm, int(with_e)):
        pytest.skip("shape not served by conv_nt2r_kernel")
    r = np.random.default_rng(Lw + cin)
    e, w, bias, s, h, gamma = _fold_case(r, n, Lw, cin, cout, dtype)
    towers = 2 if n % 2 == 0 else 1                            # an even window count runs as two towers with their own affines
    wpt = n // towers
    s = np.stack([s, (s * r.normal(1.0, 0.2, cin)).astype(np.float32)][:towers])
    h = np.stack([h, (h + r.normal(0.0, 0.3, cin)).astype(np.float32)][:towers])
    wf, hb = _fold_weights(w, s, h, dtype, bias)
    rows = L().query("vm_conv_stat_rows", Lw)
    z = torch.empty(n, Lw, cout, dtype=tdt, device="cuda")
    ssum = torch.empty(n * rows, cout, dtype=torch.float32, device="cuda")
    ssq = torch.empty_like(ssum)
    eo = torch.full((n, Lw // 2 + 2, cout), 7.0, dtype=tdt, device="cuda") if with_e else None
    L().call("vm_conv_fwd_fold", p(padded(e, tdt)), p(wf), p(dev(bias)), p(hb), p(dev(gamma)) if with_e else None, n, wpt, Lw, cin, cout,
             vm, p(z), p(ssum), p(ssq), p(eo), None, None, None, stream())
    torch.cuda.synchronize()
    if L().query("vm_pack_nt_weights_supported", cout, cin, vm):
        # round 4: the same launch with the weights in fragment order (conv_nt3_kernel where the channel count has a written-out K
        # loop: L2 -> registers, no weight stage in LDS, one barrier per channel chunk) -- the same products in the same order, so
        # every output is bit-identical to the staged kernel's
        wfp = torch.empty_like(wf)
        L().call("vm_pack_nt_weights", p(wf), towers, cout, cin, vm, p(wfp), stream())
        z3, ssum3, ssq3 = torch.empty_like(z), torch.empty_like(ssum), torch.empty_like(ssq)
        eo3 = torch.full_like(eo, 7.0) if with_e else None
        L().call("vm_conv_fwd_fold", p(padded(e, tdt)), p(wf), p(dev(bias)), p(hb), p(dev(gamma)) if with_e else None, n, wpt, Lw, cin, cout,
                 vm, p(z3), p(ssum3), p(ssq3), p(eo3), None, p(wfp), None, stream())
        torch.cuda.synchronize()
        assert torch.equal(z3, z) and torch.equal(ssum3, ssum) and torch.equal(ssq3, ssq) and (not with_e or torch.equal(eo3, eo))
    # the definition, with the weights the kernel multiplies by (W * scale rounded to the storage type) and exact shift terms
    zr = np.empty((n, Lw, cout))
    for t in range(towers):
        wq = quant((w * s[t][None, :, None]).astype(np.float32), dtype).numpy()
        ones = np.ones((wpt, Lw, cin))
        zr[t * wpt:(t + 1) * wpt] = np.maximum(_conv_same(e[t * wpt:(t + 1) * wpt], wq)
                                               + _conv_same(ones * h[t][None, None, :].astype(np.float64), w.astype(np.float64)) + bias, 0.0)
    zg = z.to(torch.float64).cpu().numpy()
    tol = 2e-3 if dtype == "f16" else 1.2e-2
    report("conv_fwd_fold[%s]" % dtype, "rel_err[n%d L%d %d->%d]" % (n, Lw, cin, cout), rel_err(zg, zr))
    assert rel_err(zg, zr) < tol
    for pos in (0, Lw - 1):   # the positions whose padding tap must NOT see the shift
        assert rel_err(zg[:, pos], zr[:, pos]) < 2 * tol, pos
    # statistics = sums of the STORED values; the extreme = pair max / min of the stored values by sign(gamma), padded layout
    assert np.allclose(ssum.cpu().numpy().reshape(n, rows, cout).sum(1), zg.sum(1), rtol=1e-4, atol=1e-3)
    assert np.allclose(ssq.cpu().numpy().reshape(n, rows, cout).sum(1), (zg * zg).sum(1), rtol=1e-4, atol=1e-3)
    if with_e:
        pairs = zg.reshape(n, Lw // 2, 2, cout)
        want = np.where(gamma[None, None, :] >= 0, pairs.max(2), pairs.min(2))
        eg = eo.to(torch.float64).cpu().numpy()
        assert np.array_equal(eg[:, 1:-1], want)
        assert (eg[:, 0] == 7.0).all() and (eg[:, -1] == 7.0).all()   # halo rows are the caller's
        # the pair form: z is not written; (e, o) hold it -- o = the other element, sign bit = "the extreme is the second element"
        e2, o2 = torch.zeros_like(eo), torch.empty(n, Lw // 2, cout, dtype=tdt, device="cuda")
        ssum2, ssq2 = torch.empty_like(ssum), torch.empty_like(ssq)
        L().call("vm_conv_fwd_fold", p(padded(e, tdt)), p(wf), p(dev(bias)), p(hb), p(dev(gamma)), n, wpt, Lw, cin, cout, vm, None,
                 p(ssum2), p(ssq2), p(e2), p(o2), None, None, stream())
        torch.cuda.synchronize()
        assert torch.equal(e2[:, 1:-1], eo[:, 1:-1]) and torch.equal(ssum2, ssum) and torch.equal(ssq2, ssq)
        ob = o2.view(torch.int16).cpu().numpy().view(np.uint16)
        second = (ob >> 15).astype(bool)
        oth = torch.from_numpy((ob & 0x7fff).view(np.int16)).view(tdt).to(torch.float64).numpy()
        zb = pairs.copy()                                   # rebuild the pairs and compare with the z of the first launch, bit for bit
        zb[:, :, 0] = np.where(second, oth, want)
        zb[:, :, 1] = np.where(second, want, oth)
        assert np.array_equal(zb, pairs)
        first_wins = (pairs[:, :, 0] == pairs[:, :, 1])
        assert not second[first_wins].any()                 # ties: the first element is the extreme


@pytest.mark.parametrize("dtype", DT16)
def test_bn_pool_bwd_apply_pairs_equals_apply_on_z(dtype):
    """vm_bn_pool_bwd_apply_pairs on (e, o) == vm_bn_pool_bwd_apply on the z they encode: same du, same partial column sums."""
    vm, tdt = DTYPES[dtype]
    r = np.random.default_rng(11)
    n, wpt, Lw, c = 4, 2, 60, 136
    z = quant(np.maximum(r.normal(0, 1, (n, Lw, c)), 0.0), dtype)   # post-ReLU: >= 0, with exact zeros and (rare) ties
    z[:, 10:12] = z[:, 10:11]                                        # a tie in every channel
    dp = quant(r.normal(0, 1, (n, Lw // 2, c)), dtype)
    scale = (r.normal(1.0, 0.3, (2, c)) * np.where(r.random((2, c)) < 0.3, -1, 1)).astype(np.float32)
    shift, mean = r.normal(0, 0.5, (2, c)).astype(np.float32), r.normal(0.4, 0.1, (2, c)).astype(np.float32)
    invstd = r.uniform(0.5, 2.0, (2, c)).astype(np.float32)
    c1, c2 = r.normal(0, 0.01, (2, c)).astype(np.float32), r.normal(0, 0.01, (2, c)).astype(np.float32)
    zt = z.to("cuda", tdt)
    pr = zt.view(n, Lw // 2, 2, c)
    pos = torch.tensor(scale >= 0, device="cuda").repeat_interleave(wpt, 0)[:, None, :]       # (n, 1, c): a maximum is pooled
    second = torch.where(pos, pr[:, :, 1] > pr[:, :, 0], pr[:, :, 1] < pr[:, :, 0])
    ext = torch.where(second, pr[:, :, 1], pr[:, :, 0])
    oth = torch.where(second, pr[:, :, 0], pr[:, :, 1])
    ep = torch.zeros(n, Lw // 2 + 2, c, dtype=tdt, device="cuda")
    ep[:, 1:-1] = ext
    o = (oth.contiguous().view(torch.int16) | (second.to(torch.int16) << 15)).view(tdt).contiguous()
    prow = L().query("vm_bn_part_rows")
    outs = []
    for pairs in (False, True):
        du = torch.zeros(n, Lw + 2, c, dtype=tdt, device="cuda")
        pdu = torch.empty(n * prow, c, dtype=torch.float32, device="cuda")
        common = (p(dev(dp, tdt)), p(dev(scale)), p(dev(shift)), p(dev(mean)), p(dev(invstd)), None, p(dev(c1)), p(dev(c2)), n, wpt, Lw, c)
        if pairs:
            L().call("vm_bn_pool_bwd_apply_pairs", p(ep), p(o), *common, vm, p(du), p(pdu), None, stream())
        else:
            L().call("vm_bn_pool_bwd_apply", p(zt.contiguous()), *common, 2, vm, p(du), p(pdu), stream())
        torch.cuda.synchronize()
        outs.append((du, pdu))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].abs().sum() > 0


@pytest.mark.parametrize("dtype", DT16)
def test_du_tower_sums(dtype):
    vm, tdt = DTYPES[dtype]
    r = np.random.default_rng(5)
    n, wpt, Lw, c = 6, 3, 40, 72
    prow = L().query("vm_bn_part_rows")
    pdu = r.normal(0, 1, (n * prow, c)).astype(np.float32)
    du = quant(r.normal(0, 1, (n, Lw, c)), dtype).numpy()
    gb = torch.empty(c, dtype=torch.float32, device="cuda")
    ds = torch.empty(2, 3, c, dtype=torch.float32, device="cuda")
    ws = torch.empty(L().query("vm_colreduce_workspace_bytes", 2, c) // 8, dtype=torch.float64, device="cuda")
    L().call("vm_du_tower_sums", p(dev(pdu)), p(padded(du, tdt)), n, wpt, Lw, c, vm, p(gb), p(ds), p(ws), stream())
    torch.cuda.synchronize()
    cs = pdu.astype(np.float64).reshape(2, wpt * prow, c).sum(1)
    want = np.stack([cs - du.reshape(2, wpt, Lw, c)[:, :, 0].sum(1), cs, cs - du.reshape(2, wpt, Lw, c)[:, :, -1].sum(1)], 1)
    assert np.abs(ds.cpu().numpy() - want).max() < 1e-4
    assert np.abs(gb.cpu().numpy() - cs.sum(0)).max() < 1e-4


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("n,wpt,Lw,cin,cout", [(6, 3, 254, 128, 128), (4, 4, 100, 64, 192), (10, 5, 61, 40, 24)])
def test_conv_wgrad_fold_matches_definition(dtype, n, wpt, Lw, cin, cout):
    vm, tdt = DTYPES[dtype]
    r = np.random.default_rng(n * Lw)
    towers = n // wpt
    e = quant(np.abs(r.normal(0, 1, (n, Lw, cin))), dtype).numpy()
    du = quant(r.normal(0, 1, (n, Lw, cout)), dtype).numpy()
    s = r.normal(1.0, 0.5, (towers, cin)).astype(np.float32)
    h = r.normal(0.0, 0.7, (towers, cin)).astype(np.float32)
    # D_t[k] from the definition (vm_du_tower_sums is tested above)
    dut = du.reshape(towers, wpt, Lw, cout)
    cs = dut.sum((1, 2))
    dsum = np.stack([cs - dut[:, :, 0].sum(1), cs, cs - dut[:, :, -1].sum(1)], 1).astype(np.float32)
    ws = torch.empty(L().query("vm_conv_wgrad_fold_workspace_bytes", n, wpt, Lw, cin, cout) // 4 + 16, dtype=torch.float32, device="cuda")
    gw = torch.empty(3, cin, cout, dtype=torch.float32, device="cuda")
    L().call("vm_conv_wgrad_fold", p(padded(e, tdt)), p(padded(du, tdt)), n, wpt, Lw, cin, cout, vm, p(dev(s)), p(dev(h)), p(dev(dsum)),
             p(ws), p(gw), stream())
    torch.cuda.synchronize()
    # definition: the layer's input is y = s_t * e + h_t inside the window, 0 in the padding
    y = e.reshape(towers, wpt, Lw, cin) * s[:, None, None, :].astype(np.float64) + h[:, None, None, :].astype(np.float64)
    yp = np.zeros((n, Lw + 2, cin))
    yp[:, 1:Lw + 1] = y.reshape(n, Lw, cin)
    want = np.stack([np.einsum("nli,nlo->io", yp[:, k:k + Lw], du) for k in range(3)])
    err = rel_err(gw.cpu().numpy(), want)
    report("conv_wgrad_fold[%s]" % dtype, "rel_err[n%d L%d %d->%d]" % (n, Lw, cin, cout), err)
    assert err < 2e-5


def _fold_arch_case(seed, pairs, l0, f=128, e=64):
    """An encoder whose k = 3 layers ARE served by conv_nt2r_kernel at a small window (channels 128..512, lengths 1016 / 508 / 254)."""
    arch = O.EncoderArch.baseline(f, e, dropout=0.0)
    p_ = O.init_params(arch, head="uniform_euclidean", seed=seed)
    r = np.random.default_rng(seed)
    for i in range(1, 5):
        c = p_[f"bn{i}.gamma"].shape[0]
        p_[f"bn{i}.gamma"] = torch.tensor(r.normal(1.0, 0.2, c) * np.where(r.random(c) < 0.15, -1, 1))
        p_[f"bn{i}.beta"] = torch.tensor(r.normal(0.0, 0.2, c))
        p_[f"conv{i}.bias"] = torch.tensor(r.normal(0.0, 0.05, c))
    mk = lambda: O.whiten(r.normal(0, 0.05, (pairs, l0, 1)) + r.uniform(-0.01, 0.01, (pairs, 1, 1))).astype(np.float32).astype(np.float64)
    x1, x2 = mk(), mk()
    y = np.concatenate([np.zeros(pairs // 2), np.ones(pairs - pairs // 2)])[:, None]
    return arch, p_, x1, x2, y


def _run(arch, p_, x1, x2, y, dtype, fold, split=True, loss="contrastive", pairs=True, center=True):
    from voicemap_amd.engine import HipEncoderEngine
    eng = HipEncoderEngine(arch.blocks, arch.embedding_dimension, dropout=0.0, head="uniform_euclidean", dtype=dtype)
    eng.set_params({k: v.numpy() for k, v in p_.items()})
    eng.fold_affine = fold
    eng.fold_pairs = pairs
    if not center:
        eng.center_blocks = ()
    eng.split_towers = split
    pl = eng.siamese_train_step(x1, x2, y, loss=loss, drop_masks=None)
    torch.cuda.synchronize()
    return eng, pl


# (the suite has a wall-clock budget: both towers in one launch -- split False -- is what test_folded_classifier_step_one_tower and the
# data-parallel tests run; the split only changes which slab a tower's windows go to)
@pytest.mark.parametrize("split,dtype", [(True, d) for d in DT16])
def test_folded_train_step_matches_oracle_and_unfolded_path(dtype, split):
    """One train_on_batch with and without the fold against the float64 oracle (train_siamese.py:52-71 on models.py:6-60): the folded
    embeddings must stay as close as the pass they replace (measured over six seeds at this size: 5-10 % further -- the weight
    rounding now scales with |scale * e| instead of |y| -- against one storage rounding of y less; at cfg-A's size 6.9e-4 against
    7.4e-4 in half, 5.5e-3 against 5.9e-3 in bf16, tests/test_gpu_fullsize_oracle.py).  Gradients of 16-bit storage are dominated
    by max-pool re-routing (tests/test_gpu_e2e.py docstring), a discrete effect of which any change of the rounding pattern draws a
    new sample: bounded loosely against the oracle here (tools/probe/fold_seeds.py prints both paths over seeds: equal on average)."""
    arch, p_, x1, x2, y = _fold_arch_case(3, 4, 4064)
    ref = O.siamese_train_step(arch, p_, O.AdamState(), torch.tensor(x1), torch.tensor(x2), torch.tensor(y))
    e_ref = np.concatenate([ref["e1"].numpy(), ref["e2"].numpy()])
    eng_f, pl_f = _run(arch, p_, x1, x2, y, dtype, True, split)
    assert pl_f["fold_now"], "the folded path did not run on a shape it is built for"
    eng_u, pl_u = _run(arch, p_, x1, x2, y, dtype, False, split)
    assert not pl_u["fold_now"]
    tag = "fold_step[%s-%s]" % (dtype, "split" if split else "serial")
    err_f, err_u = rel_err(pl_f["emb"].cpu().numpy(), e_ref), rel_err(pl_u["emb"].cpu().numpy(), e_ref)
    report(tag, "emb_rel_err_vs_fp64_folded", err_f)
    report(tag, "emb_rel_err_vs_fp64_unfolded", err_u)
    tol = 4e-3 if dtype == "f16" else 3e-2
    assert err_f < tol and err_f < 1.25 * err_u
    lf = pl_f["loss_acc"][0].item()
    assert abs(lf - ref["loss"].item()) < tol * max(1.0, abs(ref["loss"].item()))
    gf, gu = eng_f.get_grads(), eng_u.get_grads()
    gtol = 0.3 if dtype == "f16" else 0.9
    for k, g in ref["grads"].items():
        report(tag, "grad_rel_err_vs_fp64_folded[%s]" % k, rel_err(gf[k], g.numpy()))
        report(tag, "grad_rel_err_vs_fp64_unfolded[%s]" % k, rel_err(gu[k], g.numpy()))
        assert grad_close(gf[k], g.numpy(), gtol, atol=1e-5), k
    pf = eng_f.get_params()
    for k, v in ref["params"].items():
        if "moving" in k:
            assert rel_err(pf[k], v.numpy()) < (3e-3 if dtype == "f16" else 2e-2), k
    assert eng_f.skipped_steps() == 0
    # (extreme, other element) instead of (z, extreme) is a re-encoding of the same values: the whole step is bit-identical -- with
    # the centred tile of f16 storage (round 5, pairs form only) switched off; with it the embeddings move by the storage rounding
    # it removes and stay as close to the oracle
    eng_z, pl_z = _run(arch, p_, x1, x2, y, dtype, True, split, pairs=False)
    eng_p, pl_p = _run(arch, p_, x1, x2, y, dtype, True, split, pairs=True, center=False)
    assert pl_p[1]["pairs_now"] and not pl_z[1]["pairs_now"] and not pl_p[1]["ctr_now"]
    assert torch.equal(pl_z["emb"], pl_p["emb"]) and torch.equal(eng_z.G, eng_p.G)
    assert bool(pl_f[1]["ctr_now"]) == (dtype == "f16")
    if dtype == "f16":
        err_p = rel_err(pl_p["emb"].cpu().numpy(), e_ref)
        report(tag, "emb_rel_err_vs_fp64_folded_uncentred", err_p)
        assert err_f < 1.15 * err_p


def test_folded_path_falls_back_with_dropout_masks_and_is_deterministic():
    """Dropout masks (SpatialDropout1D scales per window and channel) switch the fold off for that call; two folded steps from the same
    state give bit-identical gradients (fixed summation orders throughout)."""
    from voicemap_amd.engine import HipEncoderEngine
    arch, p_, x1, x2, y = _fold_arch_case(4, 2, 4064)
    eng = HipEncoderEngine(arch.blocks, arch.embedding_dimension, dropout=0.0, head="uniform_euclidean", dtype="f16")
    eng.set_params({k: v.numpy() for k, v in p_.items()})
    masks = [torch.ones(4, c, device="cuda") for (_, c, _) in arch.blocks]
    pl = eng.siamese_train_step(x1, x2, y, drop_masks=masks, apply_update=False)
    assert not pl["fold_now"]
    g_masked = eng.G.clone()
    pl = eng.siamese_train_step(x1, x2, y, drop_masks=None, apply_update=False)
    assert pl["fold_now"]
    g1 = eng.G.clone()
    eng.siamese_train_step(x1, x2, y, drop_masks=None, apply_update=False)
    assert torch.equal(g1, eng.G)
    # all-ones masks are the identity: the two paths compute the same step up to the storage roundings they do not share
    assert rel_err(g1.cpu().numpy(), g_masked.cpu().numpy()) < 0.2


def test_folded_classifier_step_one_tower():
    """experiments/train_classifier.py's step (one encoder call = one tower, Dense(num_classes, softmax) + categorical CE) through the
    folded forward at channel counts the kernels serve: probabilities, loss and gradients against the float64 oracle."""
    from voicemap_amd.engine import HipEncoderEngine
    nc, n, l0 = 24, 6, 4064
    arch = O.EncoderArch.baseline(128, 64, dropout=0.0)
    p_ = O.init_params(arch, head="classifier", num_classes=nc, seed=5)
    r = np.random.default_rng(5)
    x = O.whiten(r.normal(0, 0.05, (n, l0, 1))).astype(np.float32).astype(np.float64)
    labels = r.integers(0, nc, n)
    eng = HipEncoderEngine(arch.blocks, arch.embedding_dimension, dropout=0.0, head="classifier", dtype="f16", num_classes=nc)
    eng.set_params({k: v.numpy() for k, v in p_.items()})
    pl = eng.classifier_train_step(x, labels, drop_masks=None, apply_update=False)
    torch.cuda.synchronize()
    assert pl["fold_now"] and pl[1]["pairs_now"]
    oh = torch.nn.functional.one_hot(torch.tensor(labels), nc).double()
    ref = O.classifier_train_step(arch, p_, O.AdamState(), torch.tensor(x), oh)
    assert rel_err(pl["prob"].cpu().numpy(), ref["prob"].numpy()) < 4e-3
    assert abs(pl["loss_acc"][0].item() - ref["loss"].item()) < 4e-3 * max(1.0, abs(ref["loss"].item()))
    grads = eng.get_grads()
    for k, g in ref["grads"].items():
        report("fold_classifier[f16]", "grad_rel_err_vs_fp64[%s]" % k, rel_err(grads[k], g.numpy()))
        assert grad_close(grads[k], g.numpy(), 0.3, atol=1e-5), k


@pytest.mark.parametrize("n,Lw,cin,cout", [(4, 508, 128, 256), (2, 1016, 64, 128)])
def test_conv_fwd_fold_centred_tile(n, Lw, cin, cout):
    """Round 5, f16 storage: vm_fold_bn_weights(ctr_out) -> vm_conv_fwd_fold(e_center) computes and rounds the tile CENTRED on the
    per-channel pedestal ctr = f16(max(start, 0)).  Against the un-centred launch on the same operands and the float64 definition:
    e_c + ctr is z's extreme to half an ulp of the CENTRED value (closer to the definition wherever the pedestal dominates), o is the
    un-centred other element with the same position flags, the statistics finalize (tile_center) to the same mean / variance, the
    *_adj constants carry the offset, and the BatchNorm-backward apply pass (e_center) sees the same z."""
    dtype, (vm, tdt) = "f16", DTYPES["f16"]
    if not L().query("vm_conv_fwd_fold_supported", n, Lw, cin, cout, vm, 1):
        pytest.skip("shape not served")
    r = np.random.default_rng(cin + 7)
    e, w, bias, s, h, gamma = _fold_case(r, n, Lw, cin, cout, dtype)
    bias = (bias + np.where(r.random(cout) < 0.5, r.uniform(1.0, 4.0, cout), 0.0)).astype(np.float32)   # half the channels on a pedestal
    towers, wpt = 2, n // 2
    s = np.stack([s, (s * r.normal(1.0, 0.2, cin)).astype(np.float32)])
    h = np.stack([h * 0.1, (h * 0.1 + r.normal(0.0, 0.05, cin)).astype(np.float32)])
    wt = np.ascontiguousarray(np.transpose(w, (2, 0, 1)).reshape(cout, 3 * cin))
    rows = L().query("vm_conv_stat_rows", Lw)

    def launch(centre):
        wf = torch.empty(towers, cout, 3 * cin, dtype=tdt, device="cuda")
        hb = torch.empty(towers, 4, cout, dtype=torch.float32, device="cuda")
        ctr = torch.zeros(towers, cout, dtype=torch.float32, device="cuda")
        L().call("vm_fold_bn_weights", p(dev(wt)), p(dev(s)), p(dev(h)), p(dev(bias)), towers, cin, cout, vm, p(wf), None, p(hb),
                 p(ctr) if centre else None, stream())
        eo = torch.zeros(n, Lw // 2 + 2, cout, dtype=tdt, device="cuda")
        oo = torch.empty(n, Lw // 2, cout, dtype=tdt, device="cuda")
        ssum = torch.empty(n * rows, cout, dtype=torch.float32, device="cuda")
        ssq = torch.empty_like(ssum)
        L().call("vm_conv_fwd_fold", p(padded(e, tdt)), p(wf), p(dev(bias)), p(hb), p(dev(gamma)), n, wpt, Lw, cin, cout, vm, None,
                 p(ssum), p(ssq), p(eo), p(oo), None, p(ctr) if centre else None, stream())
        torch.cuda.synchronize()
        return hb, ctr, eo, oo, ssum, ssq
    hb0, _, e0, o0, ss0, sq0 = launch(False)
    hb1, ctr, e1, o1, ss1, sq1 = launch(True)
    start = hb0[:, 3].cpu().numpy()
    c = ctr.cpu().numpy()
    assert np.array_equal(c, np.maximum(start, 0).astype(np.float16).astype(np.float32)) and (c > 1.0).sum() > cout // 2
    assert np.allclose(hb1[:, 3].cpu().numpy(), start - c, rtol=0, atol=1e-6) and torch.equal(hb1[:, :3], hb0[:, :3])
    cw = np.repeat(c, wpt, axis=0)[:, None, :]                                   # per window
    g_e0, g_e1 = e0[:, 1:-1].double().cpu().numpy(), e1[:, 1:-1].double().cpu().numpy() + cw
    # the float64 definition of z's pool extreme
    zr = np.empty((n, Lw, cout))
    for t in range(towers):
        wq = quant((w * s[t][None, :, None]).astype(np.float32), dtype).numpy()
        zr[t * wpt:(t + 1) * wpt] = np.maximum(_conv_same(e[t * wpt:(t + 1) * wpt], wq)
                                               + _conv_same(np.ones((wpt, Lw, cin)) * h[t][None, None, :].astype(np.float64), w.astype(np.float64)) + bias, 0.0)
    pr = zr.reshape(n, Lw // 2, 2, cout)
    want = np.where(gamma[None, None, :] >= 0, pr.max(2), pr.min(2))
    ped = np.broadcast_to(cw > 1.0, want.shape)
    err0, err1 = np.abs(g_e0 - want), np.abs(g_e1 - want)
    report("conv_fwd_fold_centred", "extreme_rms_err_pedestal_channels_uncentred[%d->%d]" % (cin, cout), float(np.sqrt((err0[ped] ** 2).mean())))
    report("conv_fwd_fold_centred", "extreme_rms_err_pedestal_channels_centred[%d->%d]" % (cin, cout), float(np.sqrt((err1[ped] ** 2).mean())))
    assert np.sqrt((err1[ped] ** 2).mean()) < 0.6 * np.sqrt((err0[ped] ** 2).mean())       # the point of it (measured: ~0.3)
    assert np.sqrt((err1[~ped] ** 2).mean()) < 1.1 * np.sqrt((err0[~ped] ** 2).mean()) + 1e-7
    assert np.abs(g_e1 - g_e0).max() <= 2.0 ** -10 * max(1.0, np.abs(g_e0).max())          # and the same value to f16's spacing
    # relu's clip: an extreme that is exactly 0 un-centred is exactly -ctr centred
    assert np.array_equal(g_e1[g_e0 == 0.0], np.zeros((g_e0 == 0.0).sum()))
    # o: the un-centred other element (one more rounding: <= 1 ulp apart), flags equal wherever the pair is not a near-tie
    b0, b1 = o0.view(torch.int16).cpu().numpy().view(np.uint16), o1.view(torch.int16).cpu().numpy().view(np.uint16)
    v0 = torch.from_numpy((b0 & 0x7fff).view(np.int16)).view(tdt).double().numpy()
    v1 = torch.from_numpy((b1 & 0x7fff).view(np.int16)).view(tdt).double().numpy()
    assert np.abs(v1 - v0).max() <= 2.0 ** -10 * max(1.0, np.abs(v0).max())
    neartie = np.abs(pr[:, :, 0] - pr[:, :, 1]) < 2.0 ** -9 * np.maximum(np.abs(pr[:, :, 0]), 1.0)
    assert np.array_equal((b0 >> 15)[~neartie], (b1 >> 15)[~neartie])
    # (a z within half a storage step of relu's clip rounds to the clip itself when it is centred: -ctr, i.e. z = 0)
    mism = (v1 == 0.0) != (v0 == 0.0)
    assert mism.mean() < 1e-3 and (np.abs(v0)[mism] <= 2.0 ** -10 * np.broadcast_to(cw, v0.shape)[mism] + 1e-7).all()
    # statistics -> vm_bn_finalize(tile_center): the same mean / variance; shift_adj / mean_adj carry ctr
    f32 = dict(dtype=torch.float32, device="cuda")
    gd, btd = dev(gamma), dev(r.normal(0, 0.2, cout).astype(np.float32))
    crws = torch.empty(L().query("vm_colreduce_workspace_bytes", towers, cout) // 8, dtype=torch.float64, device="cuda")
    out = []
    for ss, sq, tc in ((ss0, sq0, None), (ss1, sq1, ctr)):
        mean, invstd, scale, shift, sha, mea = (torch.zeros(towers, cout, **f32) for _ in range(6))
        L().call("vm_bn_finalize", p(ss), p(sq), wpt * rows, towers, cout, float(wpt * Lw), p(gd), p(btd), 1e-3, 0.99, 1, None, None,
                 p(mean), p(invstd), p(scale), p(shift), p(crws), None, 0.0, None, p(sha) if tc is not None else None,
                 p(mea) if tc is not None else None, p(tc), stream())
        torch.cuda.synchronize()
        out.append((mean, invstd, scale, shift, sha, mea))
    (m0, i0, s0, h0, _, _), (m1, i1, s1, h1, sha, mea) = out
    std = (1.0 / i0).cpu().numpy()
    # (the un-centred sums carry the rounding of pedestal-sized values: ~2^-11 ctr / sqrt(3 n) on the mean)
    assert (np.abs((m1 - m0).cpu().numpy()) < 2e-4 * std + 1e-4 * (1.0 + c)).all() and rel_err(i1.cpu().numpy(), i0.cpu().numpy()) < 1e-3
    assert ((sha - (h1 + s1 * ctr)).abs() <= 2.4e-7 * (h1.abs() + (s1 * ctr).abs()) + 1e-7).all()
    assert torch.allclose(mea, m1 - ctr, rtol=1e-6, atol=1e-6)
    # apply pass: (e_c, o, e_center) == (e, o) of the un-centred launch up to the storage spacing of du
    dp = quant(r.normal(0, 1, (n, Lw // 2, cout)), dtype).to("cuda", tdt)
    c1, c2 = dev(r.normal(0, 0.01, (towers, cout)).astype(np.float32)), dev(r.normal(0, 0.01, (towers, cout)).astype(np.float32))
    prow = L().query("vm_bn_part_rows")
    dus = []
    for ee, oo, tc in ((e0, o0, None), (e1, o1, ctr)):
        du = torch.zeros(n, Lw + 2, cout, dtype=tdt, device="cuda")
        pdu = torch.empty(n * prow, cout, **f32)
        L().call("vm_bn_pool_bwd_apply_pairs", p(ee), p(oo), p(dp), p(s0), p(h0), p(m0), p(i0), None, p(c1), p(c2), n, wpt, Lw, cout, vm,
                 p(du), p(pdu), p(tc), stream())
        torch.cuda.synchronize()
        dus.append(du.double().cpu().numpy())
    # (pairs whose elements are within a storage step of each other may route dp to the other position: the centred comparison is
    # the finer one; everywhere else the two agree to the storage spacing of du)
    d0, d1 = (d[:, 1:-1].reshape(n, Lw // 2, 2, cout) for d in dus)
    near_k = np.abs(g_e0 - v0) <= 2.0 ** -9 * np.maximum(np.abs(g_e0), 1.0)       # by the KERNEL's values (they are 1e-3 from the definition's)
    clip_k = (np.minimum(g_e0, v0) <= 2.0 ** -9 * cw) & (cw > 0)                  # within a centred storage step of relu's clip
    keep = np.broadcast_to(~(neartie | near_k | clip_k)[:, :, None, :], d0.shape)
    bad = np.abs(d1 - d0) > 2e-3 * np.abs(d0).max()
    assert rel_err(d1[keep], d0[keep]) < 2e-3, (rel_err(d1[keep], d0[keep]), float((bad & keep).mean()), float(keep.mean()),
                                                 [(float(a), float(b)) for a, b in zip(d0[bad & keep][:6], d1[bad & keep][:6])])
    assert ((d1 == 0.0) != (d0 == 0.0))[keep].mean() < 1e-3


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("with_drop", [False, True])
def test_last_block_pair_form_equals_the_z_form(dtype, with_drop):
    """Round 6: the last block in pair form (voicemap/models.py:31-37 and their backward).  On the (e, o) that encode a z:
    vm_bn_drop_pool_gmax_partials_e(e) leaves the partial (value, position) rows vm_bn_drop_pool_gmax_partials(z) leaves;
    vm_bn_bwd_gmax_finalize_e(e) the c1 / c2 / grad_gamma / grad_beta of vm_bn_bwd_gmax_finalize(z); vm_bn_pool_bwd_apply_pairs_gmax(e, o)
    the du and partial column sums of vm_bn_pool_bwd_apply_gmax(z) -- bit for bit."""
    vm, tdt = DTYPES[dtype]
    r = np.random.default_rng(23)
    n, wpt, Lw, c = 6, 3, 100, 136
    lq = Lw // 2
    z = quant(np.maximum(r.normal(0.1, 1, (n, Lw, c)), 0.0), dtype)   # post-ReLU: >= 0, with exact zeros
    z[:, 10:12] = z[:, 10:11]                                          # a tie inside a pair, in every channel
    z[:, 40:42] = z[:, 20:22]                                          # ... and two pairs with the same extreme
    scale = (r.normal(1.0, 0.3, (2, c)) * np.where(r.random((2, c)) < 0.3, -1, 1)).astype(np.float32)
    shift, mean = r.normal(0, 0.5, (2, c)).astype(np.float32), r.normal(0.4, 0.1, (2, c)).astype(np.float32)
    invstd = r.uniform(0.5, 2.0, (2, c)).astype(np.float32)
    c1h, c2h = r.normal(0, 0.01, (2, c)).astype(np.float32), r.normal(0, 0.01, (2, c)).astype(np.float32)
    drop = dev(((r.random((n, c)) > 0.2) / 0.8).astype(np.float32)) if with_drop else None
    zt = z.to("cuda", tdt).contiguous()
    pr = zt.view(n, lq, 2, c)
    pos = torch.tensor(scale >= 0, device="cuda").repeat_interleave(wpt, 0)[:, None, :]
    second = torch.where(pos, pr[:, :, 1] > pr[:, :, 0], pr[:, :, 1] < pr[:, :, 0])
    ext = torch.where(second, pr[:, :, 1], pr[:, :, 0])
    oth = torch.where(second, pr[:, :, 0], pr[:, :, 1])
    ep = torch.zeros(n, lq + 2, c, dtype=tdt, device="cuda")
    ep[:, 1:-1] = ext
    ep[:, 0] = 7.0      # halo rows must never be read (a finite marker that would win every maximum)
    ep[:, -1] = 7.0
    o = (oth.contiguous().view(torch.int16) | (second.to(torch.int16) << 15)).view(tdt).contiguous()
    sc, sh, mu, isd = dev(scale), dev(shift), dev(mean), dev(invstd)
    rows = L().query("vm_bn_part_rows")
    f32 = dict(dtype=torch.float32, device="cuda")
    # ---- forward: GlobalMaxPool1D partials
    pv = [torch.full((n * rows, c), float("nan"), **f32) for _ in range(2)]
    pi = [torch.full((n * rows, c), -7, dtype=torch.int32, device="cuda") for _ in range(2)]
    L().call("vm_bn_drop_pool_gmax_partials", p(zt), p(sc), p(sh), p(drop), n, wpt, Lw, c, 2, vm, p(pv[0]), p(pi[0]), stream())
    L().call("vm_bn_drop_pool_gmax_partials_e", p(ep), p(sc), p(sh), p(drop), n, wpt, lq, c, vm, p(pv[1]), p(pi[1]), stream())
    torch.cuda.synchronize()
    assert torch.equal(pv[0], pv[1]) and torch.equal(pi[0], pi[1])
    assert int(pi[1][pi[1] != 0x7fffffff].max()) < lq    # (0x7fffffff: the "empty" partial rows of a window with fewer segments)
    # ---- backward: sparse sums + finalize, then the apply pass
    dg = dev(r.normal(0, 1, (n, c)))
    gi = r.integers(0, lq, (n, c))
    gi[0, :3] = -1
    gidx = dev(gi, torch.int32)
    outs = []
    for pairs in (False, True):
        fin = [torch.full((2, c), float("nan"), **f32), torch.full((2, c), float("nan"), **f32), torch.full((c,), float("nan"), **f32),
               torch.full((c,), float("nan"), **f32)]
        du = torch.zeros(n, Lw + 2, c, dtype=tdt, device="cuda")
        pdu = torch.empty(n * rows, c, **f32)
        tail = (p(dg), p(gidx), p(sc), p(sh), p(mu), p(isd), p(drop))
        if pairs:
            L().call("vm_bn_bwd_gmax_finalize_e", p(ep), *tail, n, wpt, lq, c, vm, float(wpt * Lw), p(fin[0]), p(fin[1]), p(fin[2]), p(fin[3]), stream())
            L().call("vm_bn_pool_bwd_apply_pairs_gmax", p(ep), p(o), *tail, p(dev(c1h)), p(dev(c2h)), n, wpt, Lw, c, vm, p(du), p(pdu), stream())
        else:
            L().call("vm_bn_bwd_gmax_finalize", p(zt), *tail, n, wpt, Lw, c, 2, vm, float(wpt * Lw), p(fin[0]), p(fin[1]), p(fin[2]), p(fin[3]), stream())
            L().call("vm_bn_pool_bwd_apply_gmax", p(zt), *tail, p(dev(c1h)), p(dev(c2h)), n, wpt, Lw, c, 2, vm, p(du), p(pdu), stream())
        torch.cuda.synchronize()
        outs.append(fin + [du, pdu])
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(b.float()).all() and torch.equal(a, b)
    assert outs[1][4].abs().sum() > 0
