"""A TRAINED cfg-A state for the f16 guard test (tests/test_gpu_f16_guard.py): 1 200 Adam steps of this repository's own
siamese training loop (the pipeline of experiments/train_siamese.py --device-data: BCE head, Adam(clipnorm 1), 64 pairs per step) on
64 synthetic speakers, in the parity-grade fp32-storage mode, then every trainable tensor AND the BatchNorm moving statistics as
float16-representable values (both the oracle and the engine load exactly these numbers).  Needs a GPU:
    gpurun -- 'PYTHONPATH=$PWD python tests/golden/make_trained_state.py gpurun_out/trained_cfgA_state.npz'
The fixture is data produced by this repository (not by the reference, which cannot run here: SURVEY 8c)."""
import os, sys, tempfile
import numpy as np, torch
from voicemap_amd import models as VM, shards as VS, utils as VU
from voicemap_amd.keras_like import Adam
from voicemap_amd.librispeech import SyntheticSpeechDataset

out = sys.argv[1]
F, E, steps, bs = 128, 64, int(os.environ.get("STEPS", "1200")), 64
bp = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
with tempfile.TemporaryDirectory() as td:
    VS.write_shards(SyntheticSpeechDataset(num_speakers=64, files_per_speaker=8, seconds=3, seed=0), os.path.join(td, "train"))
    train = VS.ShardedSpeechDataset(os.path.join(td, "train"), 3, stochastic=True)
    train.to_device("cuda")
    torch.manual_seed(1)
    np.random.seed(1)
    enc = VM.get_baseline_convolutional_encoder(F, E, dropout=0.0, dtype="f32s")
    net = VM.build_siamese_net(enc, (train.fragment_length // 4, 1))
    net.compile(loss="binary_crossentropy", optimizer=Adam(clipnorm=1.0), metrics=["accuracy"])
    gen = (bp(b) for b in train.yield_verification_batches_device(bs))
    hist = []
    for i in range(steps):
        x, yb = next(gen)
        o = net.train_on_batch(x, yb)
        hist.append([float(o[0]), float(o[1])])
    torch.cuda.synchronize()
    eng = net._ensure_engine()
    params = eng.get_params()
    h = np.array(hist)
    print("loss/acc first 100 steps %.4f / %.3f, last 100 steps %.4f / %.3f" % (h[:100, 0].mean(), h[:100, 1].mean(), h[-100:, 0].mean(), h[-100:, 1].mean()))
    np.savez_compressed(out, **{k: np.asarray(v, dtype=np.float32).astype(np.float16) for k, v in params.items()},
                        __meta__=np.array([steps, bs, h[:100, 0].mean(), h[-100:, 0].mean(), h[-100:, 1].mean()], dtype=np.float64))
    print("wrote", out, os.path.getsize(out), "bytes;", {k: tuple(np.shape(v)) for k, v in params.items()})
