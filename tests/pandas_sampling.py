"""The reference's pandas formulation of the pair / task sampling (voicemap/librispeech.py:145-240), restated in py3 /
pandas-2 syntax as TEST infrastructure: the product (voicemap_amd/librispeech.py) does the same draws on plain arrays and
tests/test_host_api.py checks that both consume np.random identically -- same files, same fragments, same order."""
import numpy as np
import pandas as pd


def get_alike_pairs(ds, num_pairs):
    """librispeech.py:145-155"""
    anchors = ds.df.sample(num_pairs * 2, weights='length')
    joined = pd.merge(anchors, ds.df, on='speaker_id').sample(num_pairs)
    return list(zip(joined['id_x'].values, joined['id_y'].values))


def get_differing_pairs(ds, num_pairs):
    """librispeech.py:157-167"""
    first = ds.df.sample(num_pairs, weights='length')
    rest = ds.df[~ds.df['speaker_id'].isin(first['speaker_id'])].sample(num_pairs, weights='length')
    return list(zip(first['id'].values, rest['id'].values))


def build_verification_batch(ds, batchsize):
    """librispeech.py:169-196, statement by statement (the fragments of the alike pairs are cut BEFORE the differing pairs
    are drawn)."""
    half = batchsize // 2
    alike = get_alike_pairs(ds, half)
    input_1_alike = np.stack([ds[i][0] for i in list(zip(*alike))[0]])
    input_2_alike = np.stack([ds[i][0] for i in list(zip(*alike))[1]])
    differing = get_differing_pairs(ds, half)
    input_1_different = np.stack([ds[i][0] for i in list(zip(*differing))[0]])
    input_2_different = np.stack([ds[i][0] for i in list(zip(*differing))[1]])
    input_1 = np.vstack([input_1_alike, input_1_different])[:, :, np.newaxis]
    input_2 = np.vstack([input_2_alike, input_2_different])[:, :, np.newaxis]
    outputs = np.append(np.zeros(half), np.ones(half))[:, np.newaxis]
    return [input_1, input_2], outputs


def build_n_shot_task(ds, k, n=1):
    """librispeech.py:204-240"""
    query = ds.df.sample(1, weights='length')
    query_index = query.index.values[0]
    query_sample = ds[query_index]
    same_speaker = ds.df['speaker_id'] == query['speaker_id'].values[0]
    correct = ds.df[same_speaker & (ds.df.index != query_index)].sample(n, weights='length')
    others = np.random.choice(ds.df[~same_speaker]['speaker_id'].unique(), k - 1, replace=False)
    parts = [correct]
    for speaker in others:
        parts.append(ds.df[~same_speaker & (ds.df['speaker_id'] == speaker)].sample(n, weights='length'))
    support_index = pd.concat(parts).index.values
    samples = [ds[i] for i in support_index]
    return query_sample, (np.stack([s[0] for s in samples]), np.stack([s[1] for s in samples]))
