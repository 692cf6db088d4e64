import cProfile, pstats, tempfile, numpy as np, torch, time
from voicemap_amd import models as VM, shards as VS, utils as VU
from voicemap_amd.keras_like import Adam
from voicemap_amd.librispeech import SyntheticSpeechDataset
bp = VU.BatchPreProcessor("siamese", VU.preprocess_instances(4))
with tempfile.TemporaryDirectory() as td:
    VS.write_shards(SyntheticSpeechDataset(num_speakers=64, files_per_speaker=8, seconds=3, seed=0), td)
    train = VS.ShardedSpeechDataset(td, 3, stochastic=True); train.to_device("cuda")
    torch.manual_seed(1); np.random.seed(1)
    net = VM.build_siamese_net(VM.get_baseline_convolutional_encoder(128, 64, dropout=0.0, dtype="f16"), (train.fragment_length // 4, 1))
    net.compile(loss="binary_crossentropy", optimizer=Adam(clipnorm=1.0), metrics=["accuracy"])
    gen = (bp(b) for b in train.yield_verification_batches_device(64))
    net.fit_generator(generator=gen, steps_per_epoch=20, epochs=1, workers=0, verbose=0)
    t0=time.perf_counter()
    for _ in range(200): next(gen)
    print("generator alone: %.3f ms per batch" % ((time.perf_counter()-t0)/200*1e3))
    pr = cProfile.Profile(); pr.enable()
    net.fit_generator(generator=gen, steps_per_epoch=200, epochs=1, workers=0, verbose=0)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
