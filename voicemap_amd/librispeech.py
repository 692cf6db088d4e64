"""Drop-in mirror of ``voicemap/librispeech.py``: the LibriSpeech index + window / verification-pair / n-shot-task
sampling API with the same method names, arguments, return layouts (channels-last ``(B, T, 1)`` inputs, ``(B, 1)``
labels, label 0 = same speaker) and error behaviour.

    LibriSpeechDataset(subsets, seconds, label='speaker', stochastic=True, pad=False, cache=True)   librispeech.py:15
      __getitem__ :103   __len__ :139   num_classes :142   get_alike_pairs :145   get_differing_pairs :157
      build_verification_batch :169   yield_verification_batches :198   build_n_shot_task :204   index_subset :243

This is host-side code (the GPU never sees file IO).  Audio decoding uses ``soundfile`` when it is installed (the
reference's only decoder); ``.wav`` (stdlib) and ``.npy`` files are also accepted so that the API can be exercised
without libsndfile.  ``SyntheticSpeechDataset`` provides the same API over generated speakers for tests, the smoke
run and benchmarks (no LibriSpeech on the GPU box).
"""
from __future__ import annotations

import os
import functools
import wave

import numpy as np
import pandas as pd

from .keras_like import Sequence

try:  # optional, like in the reference environment (PySoundFile==0.9.0.post1)
    import soundfile as sf
except Exception:  # pragma: no cover - not installed in the build container
    sf = None

try:
    from config import PATH, LIBRISPEECH_SAMPLING_RATE
except Exception:  # pragma: no cover
    PATH = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    LIBRISPEECH_SAMPLING_RATE = 16000

sex_to_label = {'M': False, 'F': True}
label_to_sex = {False: 'M', True: 'F'}


def read_audio(path):
    """-> (float64 samples in [-1, 1), sample rate).  FLAC/OGG/... through soundfile, .wav through the stdlib, .npy raw."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npy':
        return np.load(path).astype(np.float64), LIBRISPEECH_SAMPLING_RATE
    if ext == '.wav' and sf is None:
        with wave.open(path, 'rb') as w:
            assert w.getsampwidth() == 2 and w.getnchannels() == 1, 'only 16-bit mono wav without soundfile'
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2')
            return pcm.astype(np.float64) / 32768.0, w.getframerate()
    if sf is None:
        raise ImportError('soundfile (libsndfile) is needed to decode %s' % path)
    return sf.read(path)


def audio_length(path):
    ext = os.path.splitext(path)[1].lower()
    if sf is not None and ext not in ('.npy',):
        return sf.info(path).frames
    return len(read_audio(path)[0])


class LibriSpeechDataset(Sequence):
    """A ``keras.utils.Sequence``-like dataset: ``dataset[i]`` -> (raw audio fragment of ``seconds`` s, label).

    # Arguments (librispeech.py:21-31)
        subsets: LibriSpeech subset name or list of names.
        seconds: fragment length; shorter files are dropped unless ``pad``.
        label: 'speaker' or 'sex'.
        stochastic: random fragment of each file (True) or its beginning (False).
        pad: zero-pad short files to the fragment length (random split of the padding if stochastic).
        cache: reuse / write ``data/{subset}.index.csv``.
    """

    def __init__(self, subsets, seconds, label='speaker', stochastic=True, pad=False, cache=True):
        assert label in ('sex', 'speaker'), 'Label type must be one of (\'sex\', \'speaker\')'
        self.subset = subsets
        self.fragment_seconds = seconds
        self.fragment_length = int(seconds * LIBRISPEECH_SAMPLING_RATE)
        self.stochastic = stochastic
        self.pad = pad
        self.label = label
        print('Initialising LibriSpeechDataset with minimum length = {}s and subsets = {}'.format(seconds, subsets))
        if isinstance(subsets, str):
            subsets = [subsets]
        self._finalise(self._build_index(list(subsets), cache))
        print('Finished indexing data. {} usable files found.'.format(len(self)))

    # ---- index ---------------------------------------------------------------------------------------------
    def _build_index(self, subsets, cache):
        cached, found = [], {s: False for s in subsets}
        if cache:
            for s in subsets:
                p = PATH + '/data/{}.index.csv'.format(s)
                if os.path.exists(p):
                    cached.append(pd.read_csv(p))
                    found[s] = True
        if all(found.values()) and cache:
            df = pd.concat(cached)
        else:
            speakers = self.read_speakers_table(PATH + '/data/LibriSpeech/SPEAKERS.TXT')
            audio_files = []
            for subset, ok in found.items():
                if not ok:
                    audio_files += self.index_subset(subset)
            df = pd.concat(cached + [pd.merge(speakers, pd.DataFrame(audio_files))])
        for s in subsets:  # librispeech.py:81-82
            df[df['subset'] == s].to_csv(PATH + '/data/{}.index.csv'.format(s), index=False)
        return df

    @staticmethod
    def read_speakers_table(path):
        """SPEAKERS.TXT parse rule of librispeech.py:61-67: skip 11 rows, '|'-separated, malformed lines dropped,
        column names stripped of ';' and blanks, string fields stripped."""
        df = pd.read_csv(path, skiprows=11, delimiter='|', on_bad_lines='skip')
        df.columns = [col.strip().replace(';', '').lower() for col in df.columns]
        return df.assign(sex=df['sex'].apply(lambda x: x.strip()), subset=df['subset'].apply(lambda x: x.strip()),
                         name=df['name'].apply(lambda x: x.strip()))

    def _finalise(self, df):
        if not self.pad:
            df = df[df['seconds'] > self.fragment_seconds]
        self.unique_speakers = len(df['id'].unique())
        df = df.rename(columns={'id': 'speaker_id', 'minutes': 'speaker_minutes'})
        df = df.reset_index(drop=True)
        self.df = df.assign(id=df.index.values)
        d = self.df.to_dict()
        self.datasetid_to_filepath = d['filepath']
        self.datasetid_to_speaker_id = d['speaker_id']
        self.datasetid_to_sex = d['sex']
        self._build_sampling_index()

    def _build_sampling_index(self):
        """Arrays behind the numpy restatement of the pair / task sampling below.  The reference does these draws with
        ``DataFrame.sample`` / ``pd.merge`` per batch (librispeech.py:145-240), which costs 6-9 ms for one 128-pair batch on a
        4000-file index -- twice the GPU's step time.  pandas' ``sample`` is ``np.random.choice(len, size, replace=False,
        p=weights/weights.sum())`` on the global RandomState, an inner ``merge`` keeps the left rows' order and, inside a key,
        the right rows' order: so the same ``np.random`` calls in the same order on plain arrays give the SAME pairs and tasks
        (tests/test_host_api.py compares the two under one seed; the pandas formulation lives in tests/pandas_sampling.py)."""
        spk = self.df['speaker_id'].values
        self._len = self.df['length'].values.astype(np.float64)
        self._uniq, self._code = np.unique(spk, return_inverse=True)          # speaker codes 0..S-1 (sorted ids)
        order = np.argsort(self._code, kind='stable')                         # files grouped by speaker, df order inside
        self._cnt = np.bincount(self._code, minlength=len(self._uniq))
        self._start = np.concatenate([[0], np.cumsum(self._cnt)[:-1]])
        self._files = order
        first = np.full(len(self._uniq), len(spk), dtype=np.int64)
        np.minimum.at(first, self._code, np.arange(len(spk)))
        self._appear = np.argsort(first, kind='stable')                       # speaker codes in order of first appearance

    _cdf_cache = None   # (the weights array, its length, p, cdf): the all-files draws repeat with the same weights every batch / task

    def _weighted(self, n, weights):
        """``DataFrame.sample(n, weights=...)`` on row positions = ``np.random.choice(len, n, replace=False, p=w / w.sum())``.

        This is ``choice``'s own algorithm for weighted sampling without replacement, restated so that (a) its argument validation
        (half a dozen passes over p per call) is skipped and (b) the first-round cdf of the all-files draws -- an O(files)
        normalisation and cumsum per batch and per n-shot task inside ``choice``, most of the host time on an index of
        train-clean-360's size -- is cached: rounds of ``random_sample(n - found)`` -> ``cdf.searchsorted(side='right')`` -> first
        occurrences kept, with the weights of the found rows zeroed and the cdf rebuilt before another round.  Same arithmetic, same
        use of the random stream: the same rows for the same seed (tests/test_host_api.py checks it against ``choice``)."""
        # np.random.choice's own refusals (pandas' sample raises the same ValueErrors through it): an empty population, a sample
        # larger than the population, fewer rows with positive weight than requested -- e.g. an n-shot task on a speaker with
        # fewer than n other files, or differing pairs when one speaker owns the whole index
        if len(weights) == 0:
            raise ValueError("a must be non-empty")
        if n > len(weights):
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        if weights is self._len:
            c = self._cdf_cache
            if c is None or c[0] is not weights or c[1] != len(weights):
                p = weights / weights.sum()
                cdf = np.cumsum(p)
                cdf /= cdf[-1]
                c = self._cdf_cache = (weights, len(weights), p, cdf, int(np.count_nonzero(weights > 0)))
            p, cdf, npos = c[2], c[3], c[4]
        else:
            npos = int(np.count_nonzero(weights > 0))
            if npos >= n:
                p = weights / weights.sum()
                cdf = np.cumsum(p)
                cdf /= cdf[-1]
        if npos < n:
            raise ValueError("Fewer non-zero entries in p than size")
        if n == 1:
            return cdf.searchsorted(np.random.random_sample(1), side='right')
        found = np.zeros(n, dtype=np.int64)
        n_uniq, pw = 0, None
        while n_uniq < n:
            x = np.random.random_sample(n - n_uniq)
            if n_uniq > 0:
                if pw is None:
                    pw = p.copy()
                pw[found[:n_uniq]] = 0
                cdf = np.cumsum(pw)
                cdf /= cdf[-1]
            new = cdf.searchsorted(x, side='right')
            _, first = np.unique(new, return_index=True)
            first.sort()
            new = new.take(first)
            found[n_uniq:n_uniq + new.size] = new
            n_uniq += new.size
        return found

    # ---- Sequence ------------------------------------------------------------------------------------------
    def _load(self, index):
        return read_audio(self.datasetid_to_filepath[index])[0]

    def __getitem__(self, index):
        instance = self._load(index)
        if self.stochastic:
            start = np.random.randint(0, max(len(instance) - self.fragment_length, 1))
        else:
            start = 0
        instance = instance[start:start + self.fragment_length]
        if self.pad and len(instance) < self.fragment_length:
            missing = self.fragment_length - len(instance)
            if self.stochastic:
                before = np.random.randint(0, missing)
                instance = np.pad(instance, (before, missing - before), 'constant')
            else:
                instance = np.pad(instance, (0, missing), 'constant')
        return instance, self._label(index)

    def _label(self, index):
        if self.label == 'sex':
            return sex_to_label[self.datasetid_to_sex[index]]
        elif self.label == 'speaker':
            return self.datasetid_to_speaker_id[index]
        raise ValueError('Label type must be one of (\'sex\', \'speaker\')')

    def __len__(self):
        return len(self.df)

    def num_classes(self):
        return len(self.df['speaker_id'].unique())

    # ---- verification pairs (librispeech.py:145-202) -----------------------------------------------------------
    def get_alike_pairs(self, num_pairs):
        """List of (id, id) pairs from the same speaker: 2*num_pairs anchors drawn with probability ~ file length, joined
        with every file of the same speaker (a file may pair with itself), num_pairs rows of the join kept."""
        anchors = self._weighted(num_pairs * 2, self._len)
        cnt = self._cnt[self._code[anchors]]                  # rows each anchor contributes to the join
        ends = np.cumsum(cnt)
        rows = np.random.choice(int(ends[-1]), size=num_pairs, replace=False)   # .sample(num_pairs) of the joined frame
        a = np.searchsorted(ends, rows, side='right')         # which anchor a joined row belongs to
        k = rows - (ends[a] - cnt[a])                         # ... and which file of that anchor's speaker (df order)
        left = anchors[a]
        right = self._files[self._start[self._code[left]] + k]
        return list(zip(left, right))

    def get_differing_pairs(self, num_pairs):
        """List of (id, id) pairs from different speakers: num_pairs files ~ length, then num_pairs files ~ length from
        the speakers NOT in the first draw."""
        first = self._weighted(num_pairs, self._len)
        taken = np.zeros(len(self._uniq), dtype=bool)
        taken[self._code[first]] = True
        others = np.flatnonzero(~taken[self._code])            # files of the speakers not in the first draw, df order
        rest = others[self._weighted(num_pairs, self._len[others])]
        return list(zip(first, rest))

    def build_verification_batch(self, batchsize):
        """([input_1, input_2], outputs): batchsize//2 same-speaker pairs then batchsize//2 different-speaker pairs;
        inputs (batchsize, T, 1) float, outputs (batchsize, 1) = zeros (same) then ones (different)."""
        half = batchsize // 2
        # np.random is consumed in the reference's order (librispeech.py:179-189): alike pairs, their input_1 fragments, their
        # input_2 fragments, THEN the differing pairs and their fragments
        alike = self.get_alike_pairs(half)
        left = [self[i][0] for i, _ in alike]
        right = [self[j][0] for _, j in alike]
        differing = self.get_differing_pairs(half)
        left += [self[i][0] for i, _ in differing]
        right += [self[j][0] for _, j in differing]
        input_1 = np.stack(left)[:, :, np.newaxis]
        input_2 = np.stack(right)[:, :, np.newaxis]
        outputs = np.append(np.zeros(half), np.ones(half))[:, np.newaxis]
        return [input_1, input_2], outputs

    def yield_verification_batches(self, batchsize):
        while True:
            yield self.build_verification_batch(batchsize)

    # ---- n-shot tasks (librispeech.py:204-240) -------------------------------------------------------------------
    def build_n_shot_task(self, k, n=1):
        """(query_sample, support_set_samples): query_sample = (audio, label); support_set_samples = (audio (k*n, T),
        labels (k*n,)) laid out [class_1]*n + ... + [class_k]*n with class_1 = the query's speaker (never the query
        file itself); the other k-1 speakers are drawn uniformly without replacement."""
        if k >= self.unique_speakers:
            raise ValueError('k must be smaller than the number of unique speakers in this dataset!')
        if k <= 1:
            raise ValueError('k must be greater than or equal to one!')
        query_index = int(self._weighted(1, self._len)[0])
        query_sample = self[query_index]
        support_index = self._n_shot_support(query_index, k, n)
        samples = [self[i] for i in support_index]
        return query_sample, (np.stack([s[0] for s in samples]), np.stack([s[1] for s in samples]))

    def _n_shot_support(self, query_index, k, n):
        """File ids of the support set of ``build_n_shot_task`` for a drawn query: n other files of its speaker, then n files
        of each of k-1 other speakers (drawn uniformly without replacement, in order of first appearance like ``unique()``)."""
        q = self._code[query_index]
        mine = self._files[self._start[q]:self._start[q] + self._cnt[q]]
        mine = mine[mine != query_index]
        parts = [mine[self._weighted(n, self._len[mine])]]
        pool = self._appear[self._appear != q]
        for sp in np.random.choice(self._uniq[pool], k - 1, replace=False):
            c = np.searchsorted(self._uniq, sp)
            files = self._files[self._start[c]:self._start[c] + self._cnt[c]]
            parts.append(files[self._weighted(n, self._len[files])])
        return np.concatenate(parts)

    @staticmethod
    def index_subset(subset):
        """Walk ``data/LibriSpeech/{subset}/<speaker>/<chapter>/*.flac`` and record speaker id, path and length."""
        audio_files = []
        print('Indexing {}...'.format(subset))
        root_dir = PATH + '/data/LibriSpeech/{}/'.format(subset)
        for root, _, files in os.walk(root_dir):
            flacs = [f for f in files if f.endswith('.flac')]
            if not flacs:
                continue
            librispeech_id = int(root.split('/')[-2])
            for f in flacs:
                path = os.path.join(root, f)
                frames = audio_length(path)
                audio_files.append({'id': librispeech_id, 'filepath': path, 'length': frames,
                                    'seconds': frames * 1. / LIBRISPEECH_SAMPLING_RATE})
        return audio_files


@functools.lru_cache(maxsize=512)
def _synthetic_recording(sid, fno, length, f0, harm, noise):
    """One generated recording: a deterministic function of its arguments, so the ~5 ms of numpy per call are spent once per process
    (the evaluation paths walk a corpus several times).  Read-only; _load hands out copies."""
    rng = np.random.default_rng(sid * 1000 + fno)
    t = np.arange(length) / LIBRISPEECH_SAMPLING_RATE
    vib = 1.0 + 0.01 * np.sin(2 * np.pi * rng.uniform(3, 7) * t + rng.uniform(0, 6.28))
    phase = 2 * np.pi * f0 * np.cumsum(vib) / LIBRISPEECH_SAMPLING_RATE
    x = sum(h * np.sin((i + 1) * phase + rng.uniform(0, 6.28)) for i, h in enumerate(harm))
    env = 0.5 + 0.5 * np.sin(2 * np.pi * rng.uniform(1.5, 4.0) * t + rng.uniform(0, 6.28)) ** 2
    out = 0.05 * env * x / np.sqrt(len(harm)) + noise * rng.standard_normal(length)
    out.setflags(write=False)
    return out


class SyntheticSpeechDataset(LibriSpeechDataset):
    """Same API over generated "speakers" (no audio files): every speaker has a fundamental frequency, a few harmonic
    weights and a noise colour; every file is a deterministic function of (speaker, file number).  Used by the tests
    (the reference's sampling invariants, restated), the smoke run and the benchmarks."""

    def __init__(self, num_speakers=12, files_per_speaker=6, seconds=3, label='speaker', stochastic=True, pad=False,
                 min_file_seconds=3.2, max_file_seconds=6.0, seed=0, subset='synthetic'):
        assert label in ('sex', 'speaker')
        self.subset = subset
        self.fragment_seconds = seconds
        self.fragment_length = int(seconds * LIBRISPEECH_SAMPLING_RATE)
        self.stochastic, self.pad, self.label = stochastic, pad, label
        rng = np.random.default_rng(seed)
        rows = []
        self._voice = {}
        for s in range(num_speakers):
            sid = 1000 + 7 * s
            self._voice[sid] = (float(rng.uniform(90, 260)), rng.uniform(0.2, 1.0, 5), float(rng.uniform(0.002, 0.02)))
            for f in range(files_per_speaker):
                secs = float(rng.uniform(min_file_seconds, max_file_seconds))
                length = int(secs * LIBRISPEECH_SAMPLING_RATE)
                rows.append({'id': sid, 'sex': 'F' if s % 2 else 'M', 'subset': subset, 'minutes': 1.0, 'name': 'speaker %d' % s,
                             'filepath': 'synthetic://%d/%d' % (sid, f), 'length': length,
                             'seconds': length * 1. / LIBRISPEECH_SAMPLING_RATE})
        self._finalise(pd.DataFrame(rows))

    def _load(self, index):
        sid = self.datasetid_to_speaker_id[index]
        path = self.datasetid_to_filepath[index]
        fno = int(path.rsplit('/', 1)[1])
        length = int(self.df.loc[index, 'length'])
        f0, harm, noise = self._voice[sid]
        return _synthetic_recording(sid, fno, length, f0, tuple(float(h) for h in harm), noise).copy()   # (the cached array is read-only)
