"""fit_generator wall time per step, cfg-A siamese net, 64 pairs per batch from a device-resident synthetic corpus (the pipeline of
experiments/train_siamese.py --device-data): per-batch loss read-back against the deferred one (models.py: defer_batch_logs).
PYTHONPATH=$PWD python tools/probe/fit_loop_rate.py"""
import os, tempfile, time
import numpy as np, torch
from voicemap_amd import models as VM, shards as VS, utils as VU
from voicemap_amd.keras_like import Adam
from voicemap_amd.librispeech import SyntheticSpeechDataset
bp = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
with tempfile.TemporaryDirectory() as td:
    VS.write_shards(SyntheticSpeechDataset(num_speakers=64, files_per_speaker=8, seconds=3, seed=0), td)
    for bs in (64, 128):
        for defer in (False, True, False, True):
            train = VS.ShardedSpeechDataset(td, 3, stochastic=True)
            train.to_device("cuda")
            torch.manual_seed(1); np.random.seed(1)
            net = VM.build_siamese_net(VM.get_baseline_convolutional_encoder(128, 64, dropout=0.0, dtype="f16"), (train.fragment_length // 4, 1))
            net.compile(loss="binary_crossentropy", optimizer=Adam(clipnorm=1.0), metrics=["accuracy"])
            net.defer_batch_logs = defer
            gen = (bp(b) for b in train.yield_verification_batches_device(bs))
            net.fit_generator(generator=gen, steps_per_epoch=20, epochs=1, workers=0, verbose=0)   # warm-up (plans, first launches)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h = net.fit_generator(generator=gen, steps_per_epoch=300, epochs=1, workers=0, verbose=0)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            # the same net's bare device step on the last batch's shape (no generator, no wrappers): what the GPU needs
            eng = net._ensure_engine()
            (x1, x2), yb = next(gen)
            o1, o2 = x1.raw.offsets, x2.raw.offsets
            yd = torch.as_tensor(np.asarray(yb, dtype=np.float32).reshape(-1)).to("cuda")
            for _ in range(10):
                eng.siamese_train_step_from_offsets(x1.raw.audio, o1, o2, yd, x1.raw.length, loss="bce")
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(200):
                eng.siamese_train_step_from_offsets(x1.raw.audio, o1, o2, yd, x1.raw.length, loss="bce")
            torch.cuda.synchronize(); bare = (time.perf_counter() - t1) / 200 * 1e3
            print("batch %3d pairs  deferred logs %-5s  %.3f ms per step   epoch loss %.4f   bare device step %.3f ms" % (bs, defer, dt / 300 * 1e3, h.history["loss"][0], bare))
