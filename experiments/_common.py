"""Shared pieces of the experiment drivers (MI355X versions of the reference's experiments/*.py harness):
argument parsing, dataset construction (LibriSpeech on disk or the synthetic stand-in) and the callback list that
experiments/train_siamese.py:73-93 of the reference wires up, in the order it relies on."""
import argparse
import os

import numpy as np

from config import LIBRISPEECH_SAMPLING_RATE, PATH
from voicemap_amd.keras_like import CSVLogger, ModelCheckpoint, ReduceLROnPlateau
from voicemap_amd.librispeech import LibriSpeechDataset, SyntheticSpeechDataset
from voicemap_amd.utils import NShotEvaluationCallback


def base_parser(description, **defaults):
    d = dict(n_seconds=3.0, downsampling=4, batchsize=64, filters=128, embedding_dimension=64, dropout=0.0, epochs=50,
             steps=500, validation_steps=100, tasks=500, n_shot=1, k_way=5, pad=True)
    d.update(defaults)
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--n-seconds", type=float, default=d["n_seconds"])
    p.add_argument("--downsampling", type=int, default=d["downsampling"])
    p.add_argument("--batchsize", type=int, default=d["batchsize"])
    p.add_argument("--filters", type=int, default=d["filters"])
    p.add_argument("--embedding-dimension", type=int, default=d["embedding_dimension"])
    p.add_argument("--dropout", type=float, default=d["dropout"])
    p.add_argument("--epochs", type=int, default=d["epochs"])
    p.add_argument("--steps-per-epoch", type=int, default=d["steps"], help="evaluate_every_n_batches of the reference")
    p.add_argument("--validation-steps", type=int, default=d["validation_steps"])
    p.add_argument("--num-evaluation-tasks", type=int, default=d["tasks"])
    p.add_argument("--n-shot", type=int, default=d["n_shot"])
    p.add_argument("--k-way", type=int, default=d["k_way"])
    p.add_argument("--no-pad", dest="pad", action="store_false", default=d["pad"])
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32", "f32s"])
    p.add_argument("--workers", type=int, default=min(8, os.cpu_count() or 1))
    p.add_argument("--synthetic", action="store_true", help="generated speakers instead of LibriSpeech on disk")
    p.add_argument("--device-data", metavar="DIR", default="",
                   help="decode the TRAINING set once into int16 shards under DIR (reused if present), keep them resident in HBM and "
                        "crop/decimate/whiten on the GPU; the host only draws pair offsets (voicemap_amd/shards.py)")
    p.add_argument("--shard-speakers", action="store_true",
                   help="with --device-data under torchrun: every rank keeps only 1/world of the speakers resident and draws its "
                        "different-speaker pairs among them (a different negative distribution from the reference's whole-corpus "
                        "sampling: opt-in; the default keeps the whole corpus on every rank)")
    p.add_argument("--sync-bn", action="store_true",
                   help="under torchrun: BatchNormalization over the GLOBAL batch (one small all-reduce per BatchNorm and direction; "
                        "N ranks x B pairs then train exactly like one device with N x B pairs).  Off by default: the reference's "
                        "BatchNormalization is per process")
    p.add_argument("--training-set", nargs="+", default=["train-clean-100", "train-clean-360"])
    p.add_argument("--validation-set", default="dev-clean")
    return p


def datasets(a, pad):
    if a.synthetic:
        train = SyntheticSpeechDataset(num_speakers=64, files_per_speaker=8, seconds=a.n_seconds, pad=pad, seed=0)
        valid = SyntheticSpeechDataset(num_speakers=72, files_per_speaker=6, seconds=a.n_seconds, stochastic=False, pad=pad,
                                       seed=1)
    else:
        train = LibriSpeechDataset(a.training_set, a.n_seconds, pad=pad)
        valid = LibriSpeechDataset(a.validation_set, a.n_seconds, stochastic=False, pad=pad)
    return train, valid


def device_resident(a, train):
    """--device-data: the training set as a ShardedSpeechDataset whose audio lives on the GPU (no padding on this path)."""
    from voicemap_amd import shards
    if not os.path.exists(os.path.join(a.device_data, "index.csv")):
        shards.write_shards(train, a.device_data)
    from voicemap_amd import parallel
    rank, world = parallel.rank_world()
    # default: the whole corpus on every rank, pairs sampled over all speakers as the reference does (librispeech.py:139-177);
    # --shard-speakers: 1 / world of the speakers per rank (shards.ShardedSpeechDataset speaker_shard) -- opt-in, ADVICE r3
    shard = (rank, world) if (world > 1 and getattr(a, "shard_speakers", False)) else None
    ds = shards.ShardedSpeechDataset(a.device_data, a.n_seconds, stochastic=True, pad=False, speaker_shard=shard)
    ds.to_device("cuda")
    return ds


def apply_sync_bn(a, model):
    """--sync-bn: switch the model's engine to SyncBN (engine.sync_bn; waveform encoders only)."""
    if getattr(a, "sync_bn", False):
        eng = model._ensure_engine()
        if type(eng).__name__ != "HipEncoderEngine":   # the log-mel / 2-D engine has its own BatchNorm passes
            raise SystemExit("--sync-bn: waveform encoders only (%s)" % type(eng).__name__)
        eng.sync_bn = True


def input_length(a):
    return int(LIBRISPEECH_SAMPLING_RATE * a.n_seconds / a.downsampling)


def standard_callbacks(a, valid, preprocessor, mode, param_str, plateau_patience=10):
    """n-shot metric FIRST (it creates logs['val_{n}-shot_acc']), then the callbacks that monitor it."""
    key = "val_{}-shot_acc".format(a.n_shot)
    return [NShotEvaluationCallback(a.num_evaluation_tasks, a.n_shot, a.k_way, valid, preprocessor=preprocessor, mode=mode),
            CSVLogger(PATH + "/logs/{}.csv".format(param_str)),
            # Keras HDF5 for the reference's architectures; the log-mel variant (not a Keras model of the reference) saves as .npz
            ModelCheckpoint(PATH + "/models/{}.{}".format(param_str, "npz" if getattr(a, "frontend", "waveform") == "logmel" else "hdf5"),
                            monitor=key, mode="max", save_best_only=True, verbose=True),
            ReduceLROnPlateau(monitor=key, mode="max", verbose=1, patience=plateau_patience)]


def seed_everything(seed=0, rank=0):
    """np.random drives the pair / task sampling (a different stream per data-parallel rank: every rank draws its own
    batches); torch's global generator seeds the weight initialisers and, through the engine's own generator, the
    SpatialDropout1D masks (HipEncoderEngine.init_params) -- the same on every rank, and rank 0's state is broadcast anyway."""
    import torch
    np.random.seed(seed + 100003 * rank)
    torch.manual_seed(seed)


def setup(seed=0):
    """First call of every script: join the torchrun process group if there is one (one process per GPU; `python -m
    torch.distributed.run --nproc-per-node N -m experiments.train_siamese ...`), bind this process to its GPU, seed.
    Returns (rank, world)."""
    import torch
    from voicemap_amd import parallel
    rank, world, local = parallel.init_distributed()
    if torch.cuda.is_available():
        if os.environ.get("VOICEMAP_DIST_BACKEND") == "gloo":   # rehearsal with more ranks than GPUs (voicemap_amd/parallel.py)
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    seed_everything(seed, rank)
    return rank, world
