import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import voicemap_oracle as O
from tests.test_gpu_spectro import _clips
from tests.gpu_util import rel_err
from voicemap_amd.spectro_engine import HipSpectrogramEncoderEngine
pairs, raw_len, F_, E = 2, 48000, 32, 64
arch = O.Encoder2dArch(F_, E, dropout=0.0)
pr = O.init_params2d(arch, head="uniform_euclidean", seed=5)
x1, x2 = _clips(pairs, raw_len, 6), _clips(pairs, raw_len, 7)
y = np.concatenate([np.zeros(pairs // 2), np.ones(pairs - pairs // 2)])[:, None]
f1, f2 = torch.tensor(O.logmel_features(x1.astype(np.float64))), torch.tensor(O.logmel_features(x2.astype(np.float64)))
ref = O.siamese2d_train_step(arch, pr, O.AdamState(), f1, f2, torch.tensor(y))
print("ref loss", float(ref["loss"]), "pred", ref["pred"].numpy().ravel())
for scale in (4096.0, 2.0**16, 2.0**20, 2.0**24):
    eng = HipSpectrogramEncoderEngine(F_, E, dropout=0.0, head="uniform_euclidean", dtype="f16")
    eng.set_params({k: v.numpy() for k, v in pr.items()})
    eng.loss_scale = scale
    pl = eng.siamese_train_step(x1, x2, y, loss="contrastive", drop_masks=None, apply_update=False)
    torch.cuda.synchronize()
    g = eng.get_grads()
    print(scale, {k: round(rel_err(g[k], ref["grads"][k].numpy()), 3) for k in ("conv1.kernel", "conv2.kernel", "conv4.kernel", "dense.kernel")}, "finite", all(np.isfinite(v).all() for v in g.values()))
