"""Pre-decoded int16 shards + device-side cropping.

The reference decodes a whole FLAC file for every 3 s window it samples (voicemap/librispeech.py:103-137) and builds pair
batches with pandas merges -- fine for Keras on one GPU of 2018, three to four orders of magnitude short of what the HIP
training step consumes (~190 k audio-seconds per second per GPU).  This module keeps the reference's dataset API and
changes where the bytes live:

* ``write_shards(dataset, out_dir)`` decodes every file of a ``LibriSpeechDataset`` (or anything with its ``df`` /
  ``_load``) ONCE into flat little-endian int16 shard files plus ``index.csv`` (the reference's index columns + ``shard``,
  ``offset``); train-clean-360 is ~41 GB this way and fits the HBM of one MI355X seven times over.
* ``ShardedSpeechDataset`` serves the same ``__getitem__`` / pair / task API from memory-mapped shards (no decode), and
  ``to_device()`` uploads the shards once; ``build_verification_batch_offsets`` then returns only START OFFSETS (two int64
  vectors + labels) chosen exactly like ``build_verification_batch`` chooses files and fragments, and the crop, the
  decimation and the whitening run on the GPU (``vm_crop_decimate_whiten``).
"""
from __future__ import annotations

import json
import os

import numpy as np
import pandas as pd

from config import LIBRISPEECH_SAMPLING_RATE
from .librispeech import LibriSpeechDataset, sex_to_label  # noqa: F401  (re-exported for symmetry)

INT16_SCALE = 32768.0


def to_int16(x: np.ndarray) -> np.ndarray:
    """float waveform in [-1, 1) -> int16 PCM (what a FLAC/WAV of the corpus holds)."""
    return np.clip(np.round(np.asarray(x, dtype=np.float64) * INT16_SCALE), -32768, 32767).astype("<i2")


def write_shards(dataset, out_dir: str, shard_samples: int = 1 << 27) -> pd.DataFrame:
    """Decode every file of ``dataset`` once and append it to flat int16 shards of at most ``shard_samples`` samples
    (a file never straddles shards).  Returns the index (also written to ``out_dir/index.csv``)."""
    os.makedirs(out_dir, exist_ok=True)
    rows, shard_no, fill, fh = [], 0, 0, None

    def open_shard(k):
        return open(os.path.join(out_dir, "shard_%05d.i16" % k), "wb")
    fh = open_shard(shard_no)
    for idx in range(len(dataset)):
        pcm = to_int16(dataset._load(idx))
        if fill > 0 and fill + len(pcm) > shard_samples:
            fh.close()
            shard_no, fill = shard_no + 1, 0
            fh = open_shard(shard_no)
        fh.write(pcm.tobytes())
        row = dataset.df.loc[idx].to_dict()
        row.pop('id', None)  # the dataset id is positional; 'id' in an index file is the speaker id (librispeech.py:70-78)
        row.update(shard=shard_no, offset=fill, length=len(pcm), seconds=len(pcm) * 1.0 / LIBRISPEECH_SAMPLING_RATE)
        rows.append(row)
        fill += len(pcm)
    fh.close()
    df = pd.DataFrame(rows).rename(columns={"speaker_id": "id", "speaker_minutes": "minutes"})
    df.to_csv(os.path.join(out_dir, "index.csv"), index=False)
    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump({"sampling_rate": LIBRISPEECH_SAMPLING_RATE, "dtype": "int16-le", "shards": shard_no + 1,
                   "files": len(rows)}, f)
    return df


class DeviceWindows:
    """A batch of windows that exists only as start offsets into a device-resident recording buffer; quacks like the
    (n, T, 1) array the reference's generators yield (``shape``, ``ndim``, ``len``, ``np.asarray``) so that it flows through
    ``BatchPreProcessor`` / ``preprocess_instances`` / ``fit_generator`` unchanged.  The models crop + decimate + whiten it on
    the GPU (``vm_crop_decimate_whiten``); ``np.asarray`` materialises the float windows on the host (tests, oracle)."""

    def __init__(self, audio, offsets, length: int):
        self.audio = audio
        # the offsets stay on the HOST until somebody needs them on the device: the training step uploads both towers' offsets and
        # the labels in ONE asynchronous copy from pinned memory (engine.siamese_train_step_from_offsets); a torch ``.to(device)`` of
        # a pageable array here would block the host until the device has drained its queue -- once per tower per step
        self.offsets_host = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64).reshape(-1))
        self._offsets_dev = None
        self.length = int(length)

    ndim = 3

    @property
    def offsets(self):
        """The start offsets as an int64 tensor on the audio's device (uploaded on first use)."""
        if self._offsets_dev is None:
            import torch
            self._offsets_dev = torch.as_tensor(self.offsets_host).to(self.audio.device)
        return self._offsets_dev

    @property
    def shape(self):
        return (int(self.offsets_host.size), self.length, 1)

    def __len__(self):
        return int(self.offsets_host.size)

    def gather(self):
        """(n, T) windows on the device, in the buffer's dtype."""
        import torch
        idx = self.offsets[:, None] + torch.arange(self.length, device=self.audio.device)[None, :]
        return self.audio[idx]

    def __array__(self, dtype=None, copy=None):
        import torch
        w = self.gather()
        x = (w.to(torch.float64) / INT16_SCALE if w.dtype == torch.int16 else w.to(torch.float64)).cpu().numpy()[:, :, None]
        return x.astype(dtype) if dtype is not None else x


class ShardedSpeechDataset(LibriSpeechDataset):
    """``LibriSpeechDataset`` API over the shards written by ``write_shards`` (same constructor semantics for ``seconds``,
    ``label``, ``stochastic``, ``pad``)."""

    def __init__(self, shard_dir, seconds, label='speaker', stochastic=True, pad=False, speaker_shard=None):
        """``speaker_shard = (rank, world)``: keep only the speakers whose position in the sorted speaker list is ``rank`` modulo
        ``world`` -- the data-parallel form of the resident corpus: every rank draws its pairs among its own 1 / world of the
        speakers and ``to_device`` uploads only their recordings (train-clean-100 + 360: ~52 GB as int16 -> 6.5 GB per rank of 8
        instead of 52 GB on each).  Same-speaker pairs are unaffected; different-speaker pairs are drawn within the rank's speakers
        (1172 / 8 = 146 of them) -- a DIFFERENT training distribution from the reference's whole-corpus negatives
        (librispeech.py:159-177), which is why it is an explicit opt-in (``--shard-speakers`` of the experiment scripts) and the
        default keeps the whole corpus on every rank (52 GB of 288 GB).  A shard needs at least two speakers."""
        assert label in ('sex', 'speaker'), 'Label type must be one of (\'sex\', \'speaker\')'
        self.subset = shard_dir
        self.fragment_seconds = seconds
        self.fragment_length = int(seconds * LIBRISPEECH_SAMPLING_RATE)
        self.stochastic, self.pad, self.label = stochastic, pad, label
        self.shard_dir = shard_dir
        df = pd.read_csv(os.path.join(shard_dir, "index.csv"))
        self.speaker_shard = None
        if speaker_shard is not None and int(speaker_shard[1]) > 1:
            rank, world = int(speaker_shard[0]), int(speaker_shard[1])
            speakers = np.sort(df['id'].unique())
            mine = set(speakers[rank::world].tolist())
            if len(mine) < 2:
                raise ValueError('speaker_shard %r leaves %d speaker(s) on this rank: different-speaker pairs need two'
                                 % ((rank, world), len(mine)))
            df = df[df['id'].isin(mine)].reset_index(drop=True)
            self.speaker_shard = (rank, world)
        self._finalise(df)
        n_shards = int(self.df['shard'].max()) + 1 if len(self.df) else 0
        self._maps = [np.memmap(os.path.join(shard_dir, "shard_%05d.i16" % k), dtype="<i2", mode="r") for k in range(n_shards)]
        base = np.concatenate([[0], np.cumsum([len(m) for m in self._maps])]) if n_shards else np.zeros(1, dtype=np.int64)
        # start of every file in the concatenation of all shards (the layout of the device buffer)
        self.global_offset = (base[self.df['shard'].values] + self.df['offset'].values).astype(np.int64)
        self.file_length = self.df['length'].values.astype(np.int64)
        if self.speaker_shard is not None:
            # the device buffer of a speaker shard holds only this rank's recordings, back to back in index order: the offsets are
            # those of THAT layout from the start (host reads go through the shard / offset columns, never through global_offset)
            self.global_offset = np.concatenate([[0], np.cumsum(self.file_length)[:-1]]).astype(np.int64) if len(self.df) \
                else np.zeros(0, dtype=np.int64)
        self.device_audio = None

    def _pcm(self, index):
        r = self.df.iloc[index]
        return self._maps[int(r['shard'])][int(r['offset']):int(r['offset']) + int(r['length'])]

    def _load(self, index):
        return np.asarray(self._pcm(index), dtype=np.float64) / INT16_SCALE

    # ---- device path ---------------------------------------------------------------------------------------
    def to_device(self, device="cuda"):
        """Upload all shards once (int16, back to back); returns the 1-D device tensor."""
        import torch
        if self.device_audio is None:
            if self.speaker_shard is None:
                parts = [torch.from_numpy(np.array(m, dtype=np.int16, copy=True)) for m in self._maps]
            else:
                # only this rank's recordings, back to back in index order (the layout global_offset describes since __init__)
                parts = [torch.from_numpy(np.array(self._pcm(i), dtype=np.int16, copy=True)) for i in range(len(self.df))]
            self.device_audio = torch.cat(parts).to(device) if parts else torch.zeros(0, dtype=torch.int16, device=device)
        return self.device_audio

    def window_starts(self, indices) -> np.ndarray:
        """Global start sample of one fragment per file id, chosen like ``__getitem__`` (random if ``stochastic``, else the
        beginning).  Padding is a host-path feature: the device path needs files at least one fragment long."""
        indices = np.asarray(indices, dtype=np.int64)
        lengths = self.file_length[indices]
        if np.any(lengths < self.fragment_length):
            raise ValueError('the device path cannot pad: a file is shorter than the fragment length')
        if self.stochastic:
            span = np.maximum(lengths - self.fragment_length, 1)
            start = np.random.randint(0, span).astype(np.int64)  # one draw per file, same stream as a loop of scalar draws
        else:
            start = np.zeros(len(indices), dtype=np.int64)
        return self.global_offset[indices] + start

    def build_verification_batch_offsets(self, batchsize):
        """(offsets_1, offsets_2, outputs): the pairs of ``build_verification_batch`` (batchsize//2 same-speaker pairs, then
        batchsize//2 different-speaker pairs; outputs (batchsize, 1) zeros then ones) as start offsets into the device buffer."""
        half = batchsize // 2
        # same np.random consumption order as build_verification_batch (reference librispeech.py:179-189)
        alike = self.get_alike_pairs(half)
        l_a = self.window_starts([i for i, _ in alike])
        r_a = self.window_starts([j for _, j in alike])
        differing = self.get_differing_pairs(half)
        l_d = self.window_starts([i for i, _ in differing])
        r_d = self.window_starts([j for _, j in differing])
        outputs = np.append(np.zeros(half), np.ones(half))[:, np.newaxis]
        return np.concatenate([l_a, l_d]), np.concatenate([r_a, r_d]), outputs

    def build_verification_batch_device(self, batchsize):
        """``build_verification_batch`` with the two inputs as ``DeviceWindows`` (needs ``to_device()`` first)."""
        assert self.device_audio is not None, 'call to_device() first'
        o1, o2, outputs = self.build_verification_batch_offsets(batchsize)
        T = self.fragment_length
        return [DeviceWindows(self.device_audio, o1, T), DeviceWindows(self.device_audio, o2, T)], outputs

    def yield_verification_batches_device(self, batchsize):
        while True:
            yield self.build_verification_batch_device(batchsize)

    def build_n_shot_task_offsets(self, k, n=1):
        """The task of ``build_n_shot_task`` (librispeech.py:204-240 of the reference) as start offsets into the device
        buffer: ((query_offset, query_label), (support_offsets (k*n,), support_labels (k*n,))), support laid out
        [class_1]*n + ... + [class_k]*n with class_1 = the query's speaker.  Same draws in the same order as the host method
        (query file, its fragment, n other files of that speaker, k-1 other speakers, n files each, their fragments)."""
        if k >= self.unique_speakers:
            raise ValueError('k must be smaller than the number of unique speakers in this dataset!')
        if k <= 1:
            raise ValueError('k must be greater than or equal to one!')
        query_index = int(self._weighted(1, self._len)[0])
        q_off = self.window_starts([query_index])[0]
        support_index = self._n_shot_support(query_index, k, n)
        s_off = self.window_starts(support_index)
        return (q_off, self._label(query_index)), (s_off, np.array([self._label(i) for i in support_index]))

    def build_n_shot_tasks_device(self, num_tasks, k, n=1):
        """``num_tasks`` tasks at once: (queries, supports, query_labels, support_labels) with queries a ``DeviceWindows`` of
        num_tasks windows and supports one of num_tasks*k*n windows (task-major, then the layout above)."""
        assert self.device_audio is not None, 'call to_device() first'
        q, s, ql, sl = [], [], [], []
        for _ in range(num_tasks):
            (qo, qlab), (so, slab) = self.build_n_shot_task_offsets(k, n)
            q.append(qo)
            s.append(so)
            ql.append(qlab)
            sl.append(slab)
        T = self.fragment_length
        return (DeviceWindows(self.device_audio, np.array(q, dtype=np.int64), T),
                DeviceWindows(self.device_audio, np.concatenate(s) if s else np.zeros(0, np.int64), T),
                np.array(ql), np.array(sl))
