"""Folded vs un-folded training step against the float64 oracle over several seeds (gradient error is dominated by max-pool
re-routing, a discrete effect: one seed says little).  python tools/probe/fold_seeds.py [dtype] [pairs] [l0]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from oracle import voicemap_oracle as O
from tests.test_gpu_fold import _fold_arch_case, _run
from tests.gpu_util import rel_err

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
l0 = int(sys.argv[3]) if len(sys.argv) > 3 else 4064
keys = ["conv1.kernel", "bn1.gamma", "conv2.kernel", "bn2.beta", "conv3.kernel", "conv4.kernel", "bn4.gamma", "dense.kernel"]
for seed in range(6):
    arch, p_, x1, x2, y = _fold_arch_case(seed, pairs, l0)
    ref = O.siamese_train_step(arch, p_, O.AdamState(), torch.tensor(x1), torch.tensor(x2), torch.tensor(y))
    e_ref = np.concatenate([ref["e1"].numpy(), ref["e2"].numpy()])
    out = []
    for fold in (True, False):
        eng, pl = _run(arch, p_, x1, x2, y, dtype, fold)
        g = eng.get_grads()
        out.append((rel_err(pl["emb"].cpu().numpy(), e_ref), [rel_err(g[k], ref["grads"][k].numpy()) for k in keys]))
    print("seed %d %s emb fold %.2e plain %.2e | grads fold/plain: %s" % (
        seed, dtype, out[0][0], out[1][0], " ".join("%s %.2f/%.2f" % (k.replace(".kernel", ".k"), a, b) for k, a, b in zip(keys, out[0][1], out[1][1]))))
