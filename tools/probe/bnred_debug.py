import numpy as np, torch
from tests.gpu_util import DTYPES, L, dev, p, quant, stream
vm, tdt = DTYPES["bf16"]
l, cin, cout = 254, 128, 64
n = l
r = np.random.default_rng(1)
w = quant(r.normal(0, 0.1, (3, cin, cout)), "bf16")
wf = torch.empty(cout * 3 * cin, dtype=tdt, device="cuda"); wd = torch.empty(cin * 3 * cout, dtype=tdt, device="cuda")
L().call("vm_prep_conv_weights", p(dev(w)), cin, cout, vm, p(wf), p(wd), stream())
g = torch.Generator(device="cuda").manual_seed(1)
dup = torch.zeros(n, l + 2, cout, dtype=tdt, device="cuda")
dup[:, 1:l + 1] = torch.randn(n, l, cout, device="cuda", generator=g).to(tdt)
a = torch.zeros(n, l, cin, dtype=tdt, device="cuda")
for i in range(n):
    a[i, i, :] = (torch.arange(cin, device='cuda') % 61 + 1).to(tdt)
rows = L().query("vm_conv_dgrad_bnred_rows", l)
s0 = torch.zeros(n * rows, cin, device="cuda"); s1 = torch.zeros(n * rows, cin, device="cuda")
dx = torch.empty(n, l, cin, dtype=tdt, device="cuda")
L().call("vm_conv_dgrad_bnred", p(dup), p(wd), n, l, cin, cout, vm, p(dx), p(a), 0, p(s0), p(s1), stream())
torch.cuda.synchronize()
S1 = s1.view(n, rows, cin).sum(1)
S0 = s0.view(n, rows, cin).sum(1)
ref1 = torch.stack([dx[i, i].float() * a[i, i].float() for i in range(n)])
ref0 = dx.float().sum(1)
e1 = (S1 - ref1).abs()
e0 = (S0 - ref0).abs()
print("S0 max err", e0.max().item(), "S1 max err", e1.max().item())
bad = (e1 > 1e-3).nonzero()
print("bad count", len(bad), "of", n * cin)
print("bad rows", sorted(set(bad[:, 0].tolist()))[:80])
print("bad chans", sorted(set(bad[:, 1].tolist()))[:130])
# does S1[i] match another row of dx?
for i in sorted(set(bad[:, 0].tolist()))[:6]:
    d = (dx[i].float() - S1[i][None, :]).abs().sum(1)
    print(i, "closest row", int(d.argmin()), float(d.min()), " s1row h0/h1 nonzero:", (s1.view(n, rows, cin)[i].abs().sum(1) > 0).tolist())

# full random A
a2 = (torch.randn(n, l, cin, device="cuda", generator=g) + 0.5).to(tdt)
L().call("vm_conv_dgrad_bnred", p(dup), p(wd), n, l, cin, cout, vm, p(dx), p(a2), 0, p(s0), p(s1), stream())
torch.cuda.synchronize()
S1 = s1.view(n, rows, cin).sum(1).double()
ref = (dx.double() * a2.double()).sum(1)
e = (S1 - ref).abs()
print("random A: max err", e.max().item(), "mean", e.mean().item(), "ref scale", ref.abs().mean().item())
print("per-window max err (first 10)", e.max(1).values[:10].tolist())
print("per-chan max err (first 16)", e.max(0).values[:16].tolist())
# half rows
sh = s1.view(n, rows, cin).double()
refh0 = (dx.double() * a2.double())[:, :128].sum(1); refh1 = (dx.double() * a2.double())[:, 128:].sum(1)
print("half0 err", (sh[:, 0] - refh0).abs().max().item(), "half1 err", (sh[:, 1] - refh1).abs().max().item())
