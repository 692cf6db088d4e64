import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpu_util import *
from voicemap_amd import _lib
Lb = _lib.lib()
rng = lambda seed: np.random.default_rng(seed)
for dt in ("f32", "bf16"):
    vm, tdt = DTYPES[dt]
    r = rng(31)
    n, wpt, l, c, pool = 4, 2, 750, 64, 2
    towers = n // wpt
    z = quant(np.maximum(np.round(r.normal(0.2, 1.0, (n, l, c)) * 2) / 2, 0.0), dt).to("cuda", tdt).contiguous()
    scale = dev(np.round(r.normal(1.0, 0.3, (towers, c)) * 4) / 4 * np.where(r.random((towers, c)) < 0.3, -1, 1))
    shift = dev(np.round(r.normal(0, 0.3, (towers, c)) * 4) / 4)
    drop = dev((r.random((n, c)) > 0.2) / 0.8)
    lq = l // pool
    act = torch.zeros(n, lq + 2, c, dtype=tdt, device="cuda")
    Lb.call("vm_bn_drop_pool_fwd", p(z), p(scale), p(shift), p(drop), n, wpt, l, c, pool, vm, p(act), stream())
    # torch reference of act
    zz = z.float().view(n, lq, pool, c)
    sc = scale.repeat_interleave(wpt, 0)[:, None, None, :]; sh = shift.repeat_interleave(wpt, 0)[:, None, None, :]
    y = ((zz * sc + sh) * drop[:, None, None, :]).max(2).values.to(tdt).float()
    print(dt, "fwd vs torch", (act[:, 1:-1].float() - y).abs().max().item())
    g0, i0 = torch.empty(n, c, device="cuda"), torch.empty(n, c, dtype=torch.int32, device="cuda")
    Lb.call("vm_global_maxpool_fwd", p(act), n, lq, c, vm, p(g0), p(i0), stream())
    g1, i1 = torch.empty_like(g0), torch.empty_like(i0)
    ws = torch.empty(Lb.query("vm_bn_drop_pool_gmax_workspace_bytes", n, c) // 4, device="cuda")
    Lb.call("vm_bn_drop_pool_gmax_fwd", p(z), p(scale), p(shift), p(drop), n, wpt, l, c, pool, vm, p(g1), p(i1), p(ws), stream())
    print(dt, "gmax two-pass vs torch", (g0 - y.max(1).values).abs().max().item(), "fused vs torch", (g1 - y.max(1).values).abs().max().item())
    bad = (g0 != g1).nonzero()
    print(dt, "mismatch count", len(bad), bad[:8].tolist())
