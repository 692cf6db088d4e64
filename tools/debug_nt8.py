"""Debug aid: compare the 256x256 phase-interleaved NT kernel with the 128x128 kernels on one shape and print where they
differ (row block x column block map).  python tools/debug_nt8.py n l cin cout [blocks] [order]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from voicemap_amd import _lib

n, l, cin, cout = [int(x) for x in sys.argv[1:5]]
blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 256
order = int(sys.argv[6]) if len(sys.argv) > 6 else 1
lib = _lib.lib()
p = lambda t: t.data_ptr()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
x = torch.randn(n, l, cin, generator=g)
w = torch.randn(3, cin, cout, generator=g) * 0.1
b = torch.randn(cout, generator=g) * 0.3
xp = torch.zeros(n, l + 2, cin, dtype=torch.bfloat16)
xp[:, 1:-1] = x.bfloat16()
xp = xp.cuda()
wd_ = w.cuda()
wf = torch.empty(cout * 3 * cin, dtype=torch.bfloat16, device="cuda")
wd = torch.empty(cin * 3 * cout, dtype=torch.bfloat16, device="cuda")
lib.call("vm_prep_conv_weights", p(wd_), cin, cout, 1, p(wf), p(wd), s)
bb = b.cuda()
rows = lib.query("vm_conv_stat_rows", l)
res = {}
for p8 in (0, 1):
    lib.call("vm_set_tuning", b"nt_p8", p8)
    lib.call("vm_set_tuning", b"nt_p8_blocks", blocks)
    lib.call("vm_set_tuning", b"nt_order", order)
    z = torch.full((n, l, cout), 7.0, dtype=torch.bfloat16, device="cuda")
    ss = torch.zeros(n * rows, cout, device="cuda")
    sq = torch.zeros(n * rows, cout, device="cuda")
    lib.call("vm_conv_fwd", p(xp), p(wf), p(bb), n, l, cin, cout, 1, p(z), p(ss), p(sq), s)
    torch.cuda.synchronize()
    res[p8] = (z.float().cpu().numpy(), ss.cpu().numpy(), sq.cpu().numpy())
z0, z1 = res[0][0], res[1][0]
bad = np.abs(z0 - z1) > 1e-2 * (1 + np.abs(z0))
print("fwd: mismatching elements", bad.sum(), "of", bad.size)
if bad.any():
    for w_ in range(n):
        bw = bad[w_]
        if not bw.any():
            continue
        rb = (l + 31) // 32
        m = np.zeros((rb, cout // 32), dtype=int)
        for i in range(rb):
            for j in range(cout // 32):
                m[i, j] = bw[i * 32:(i + 1) * 32, j * 32:(j + 1) * 32].sum()
        print("window", w_)
        print(m)
        break
print("stat sum max diff", np.abs(res[0][1] - res[1][1]).max(), "sq", np.abs(res[0][2] - res[1][2]).max())
