"""CPU emulation of config 4 (log-mel + 2-D CNN, 298 x 64, filters 32) in f16 storage, one storage site at a time: which rounding moves the
embeddings by how much (float64 restatement with torch.float16 round trips at the image, the GEMM weights, the conv outputs z and the
pooled activations).  Round 6: all sites 1.23e-3 (the HIP path measured 1.27e-3); the image alone 7.7e-4, block 1's z 5.0e-4 -- both now
enter on two planes of the storage type (vm_stft_logmel_f16s_split, vm_conv2d_first_fwd_split, vm_bn_pool2d_stack_fwd_split): 7.3e-4 measured.
Runs on the CPU in ~10 s:  python tools/probe/config4_storage_sites.py"""
import os, sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import voicemap_oracle as O
torch.set_num_threads(8)
def rng(s): return np.random.default_rng(s)
def _clips(n, raw_len, seed):
    r = rng(seed)
    t = np.arange(raw_len) / 16000.0
    f0 = r.uniform(100, 400, (n, 1))
    x = 0.1 * np.sin(2 * np.pi * f0 * t[None, :]) * (0.5 + 0.5 * np.sin(2 * np.pi * 3.0 * t[None, :] + r.uniform(0, 6, (n, 1)))) \
        + 0.02 * r.normal(0, 1, (n, raw_len))
    return x.astype(np.float32)
def r16(x): return x.to(torch.float16).to(x.dtype)
def fwd(arch, p, feats, sites, center=None):
    h = feats[:, None, :, :]
    if 'in' in sites:
        if center is not None and 'in' in center:
            m = h.mean(); h = r16(h - m) + m
        else: h = r16(h)
    for i, c in enumerate(arch.channels):
        w = p[f"conv{i+1}.kernel"].permute(3, 2, 0, 1)
        if 'w' in sites or ('w%d'%i) in sites: w = r16(w)
        z = torch.relu(F.conv2d(h, w, p[f"conv{i+1}.bias"], padding=1))
        if 'z' in sites or ('z%d'%i) in sites:
            if center is not None and ('z%d'%i) in center:
                m = z.mean(dim=(0,2,3),keepdim=True); z = r16(z-m)+m
            else: z = r16(z)
        gam, bet = p[f"bn{i+1}.gamma"], p[f"bn{i+1}.beta"]
        mean = z.mean(dim=(0, 2, 3)); var = z.var(dim=(0, 2, 3), unbiased=False)
        y = (z - mean[None, :, None, None]) * torch.rsqrt(var + arch.bn_eps)[None, :, None, None] * gam[None, :, None, None] + bet[None, :, None, None]
        h = F.max_pool2d(y, 2, 2)
        if i < 3 and ('a' in sites or ('a%d'%i) in sites): h = r16(h)
    g = h.amax(dim=(2, 3))
    return g @ p["dense.kernel"] + p["dense.bias"]
pairs, raw_len, F_, E = 8, 48000, 32, 64
arch = O.Encoder2dArch(F_, E, dropout=0.0)
pr = O.init_params2d(arch, head="uniform_euclidean", seed=5)
x1, x2 = _clips(pairs, raw_len, 6), _clips(pairs, raw_len, 7)
f1 = torch.tensor(O.logmel_features(x1.astype(np.float64)))
f2 = torch.tensor(O.logmel_features(x2.astype(np.float64)))
print('feat stats', f1.mean().item(), f1.std().item(), f1.min().item(), f1.max().item())
def emb(sites, center=None):
    return torch.cat([fwd(arch, pr, f1, sites, center), fwd(arch, pr, f2, sites, center)])
ref = emb(())
def rel(a): return (torch.linalg.norm(a-ref)/torch.linalg.norm(ref)).item()
for s in [('in',), ('w',), ('z',), ('a',), ('w0',),('w1',),('w2',),('w3',),('z0',),('z1',),('z2',),('z3',),('a0',),('a1',),('a2',), ('in','w','z','a')]:
    print(s, '%.3e' % rel(emb(s)))
print('in centered', '%.3e' % rel(emb(('in',), ('in',))))
for k in range(4):
    print('z%d centered'%k, '%.3e' % rel(emb(('z%d'%k,), ('z%d'%k,))))
print('all, in centered', '%.3e' % rel(emb(('in','w','z','a'), ('in',))))
print('all, in+z centered', '%.3e' % rel(emb(('in','w','z','a'), ('in','z0','z1','z2','z3'))))
