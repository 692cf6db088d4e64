import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import voicemap_oracle as O
from voicemap_amd import keras_like as K, models, utils
from voicemap_amd.librispeech import SyntheticSpeechDataset
valid = SyntheticSpeechDataset(num_speakers=14, files_per_speaker=8, seconds=0.5, stochastic=False, seed=3)
enc = models.get_baseline_convolutional_encoder(16, 32, dropout=0.0, dtype="f32")
net = models.build_siamese_net(enc, (2000, 1))
net.compile(loss=utils.contrastive_loss, optimizer=K.Adam(clipnorm=1.))
eng = net._ensure_engine()
r = np.random.default_rng(0)
eng.set_params({f"bn{i}.moving_mean": r.normal(0.05, 0.02, c) for i, (_, c, _) in enumerate(eng.blocks, 1)})
eng.set_params({f"bn{i}.moving_variance": r.uniform(0.01, 0.1, c) for i, (_, c, _) in enumerate(eng.blocks, 1)})
bp = utils.BatchPreProcessor("siamese", utils.preprocess_instances(4))
arch = O.EncoderArch(blocks=eng.blocks, embedding_dimension=32, dropout=0.0)
p = {k: torch.tensor(v, dtype=torch.float64) for k, v in eng.get_params().items()}
pre = O.preprocess_instances(4)
np.random.seed(11)
tasks = [valid.build_n_shot_task(5, 1) for _ in range(4)]
k = 5
in1 = np.concatenate([np.stack([q[0]] * k) for q, s in tasks])[:, :, None]
in2 = np.concatenate([s[0] for q, s in tasks])[:, :, None]
([a, b], _) = bp(([in1, in2], []))
pred = utils._siamese_predict_towers(eng, a, b, tower=k).reshape(len(tasks), k)
pl = eng.plan(2 * len(tasks) * k, eng.last_infer_l0, False)
emb = pl["emb"].cpu().numpy()
for t, (q, s) in enumerate(tasks):
    i1 = pre(np.stack([q[0]] * k)[:, :, None]); i2 = pre(s[0][:, :, None])
    pr, e1, e2 = O.siamese_forward(arch, p, torch.tensor(i1), torch.tensor(i2), False)
    print(t, "gpu", pred[t].round(5), "oracle", pr[:, 0].numpy().round(5))
    print("   emb1 err", np.abs(emb[t*k:(t+1)*k] - e1.numpy()).max(), "emb2 err", np.abs(emb[len(tasks)*k + t*k: len(tasks)*k + (t+1)*k] - e2.numpy()).max())
# single-task call
for t, (q, s) in enumerate(tasks[:2]):
    i1 = np.stack([q[0]] * k)[:, :, None]; i2 = s[0][:, :, None]
    ([a1, b1], _) = bp(([i1, i2], []))
    print("single", t, utils._siamese_predict_towers(eng, a1, b1, tower=k).round(5))
print("---- full")
np.random.seed(11)
got = utils.n_shot_task_evaluation(net, valid, bp, 12, 1, 5, network_type="siamese")
np.random.seed(11)
tasks = [valid.build_n_shot_task(5, 1) for _ in range(12)]
in1 = np.concatenate([np.stack([q[0]] * k) for q, s in tasks])[:, :, None]
in2 = np.concatenate([s[0] for q, s in tasks])[:, :, None]
([a, b], _) = bp(([in1, in2], []))
pred = utils._siamese_predict_towers(eng, a, b, tower=k).reshape(len(tasks), k)
print("got", got, "argmins", pred.argmin(1))
want = []
for t, (q, s) in enumerate(tasks):
    i1 = pre(np.stack([q[0]] * k)[:, :, None]); i2 = pre(s[0][:, :, None])
    pr, e1, e2 = O.siamese_forward(arch, p, torch.tensor(i1), torch.tensor(i2), False)
    want.append(int(pr[:, 0].argmin()))
    if want[-1] != pred[t].argmin():
        print(t, pred[t], pr[:, 0].numpy())
print("want argmins", want)
