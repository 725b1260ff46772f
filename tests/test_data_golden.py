"""data/dataset.py against fixtures recorded from the imported reference
(tools/gen_golden.py:gen_data -> tests/golden/data.npz): collate results, the bucket sampler's batches under a seeded
np.random, SpeechDataset crops / masks / extra features; plus the rank-sharded sampler and the ragged host batch."""
import os
import numpy as np
import pandas as pd
import pytest
import torch

from pytorch_sound_amd.data import dataset as D
from pytorch_sound_amd.data.meta import MetaFrame, MetaType

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'data.npz'))


def _batch():
    return [[(int(G['collate/in/%d/%d' % (i, j)]) if j == 2 else G['collate/in/%d/%d' % (i, j)]) for j in range(5)] for i in range(5)]


def test_pad_collate_matches_reference():
    res = D.SpeechDataLoader.pad_collate_fn(_batch())
    assert len(res) == 5
    for j, x in enumerate(res):
        want = G['collate/out/%d' % j]
        assert isinstance(x, torch.Tensor) and x.dtype == torch.from_numpy(want).dtype
        assert x.shape == want.shape and np.array_equal(x.numpy(), want)             # bit-exact: copies and zeros only
    one = D.SpeechDataLoader.pad_collate_fn(_batch()[:1])
    for j, x in enumerate(one):
        assert np.array_equal(x.numpy(), G['collate/one/%d' % j])
    assert D.SpeechDataLoader.pad_collate_fn([None]) is None
    # python floats / lists are left as lists, mixed types are refused (dataset.py:208-215)
    assert D.SpeechDataLoader.pad_collate_fn([[0.5], [1.5]]) == [[0.5, 1.5]]
    with pytest.raises(AssertionError):
        D.SpeechDataLoader.pad_collate_fn([[1], [1.5]])
    with pytest.raises(ValueError):
        D.SpeechDataLoader.pad_collate_fn([[np.zeros((1, 1, 1, 2))], [np.zeros((1, 1, 1, 3))]])


class _Src:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


@pytest.mark.parametrize('tag,skip', [('all', False), ('skip', True)])
def test_bucket_sampler_matches_reference(tag, skip):
    np.random.seed(7)
    smp = D.BucketRandomBatchSampler(_Src(1000), n_buckets=5, batch_size=16, skip_last_bucket=skip)
    got = np.asarray(list(smp), np.int64)
    assert np.array_equal(got, G['sampler/%s/batches' % tag])                        # same RNG consumption, same batches
    assert len(smp) == int(G['sampler/%s/len' % tag]) and smp.bucket_size == int(G['sampler/%s/bucket_size' % tag])
    with pytest.raises(AssertionError):
        D.BucketRandomBatchSampler(_Src(80), n_buckets=5, batch_size=16)


def test_bucket_sampler_rank_sharding():
    """world 4: every rank sees the same bucket sequence, the slices are disjoint, their union is the world-1 sampler's
    batch of 4x the size, and every batch stays inside one bucket."""
    per_rank = []
    for r in range(4):
        np.random.seed(3)
        per_rank.append(list(D.BucketRandomBatchSampler(_Src(2000), 5, 8, rank=r, world_size=4)))
    np.random.seed(3)
    whole = list(D.BucketRandomBatchSampler(_Src(2000), 5, 32))
    assert len(per_rank[0]) == len(whole) > 0
    bsize = D.BucketRandomBatchSampler(_Src(2000), 5, 32).bucket_size
    for step, glob in enumerate(whole):
        parts = [per_rank[r][step] for r in range(4)]
        assert all(len(p) == 8 for p in parts)
        assert sorted(sum(parts, [])) == sorted(glob)
        assert len({i // bsize for i in glob}) == 1
    with pytest.raises(ValueError):
        D.BucketRandomBatchSampler(_Src(2000), 5, 8, rank=4, world_size=4)


class _Meta(MetaFrame):
    sr = 22050

    def __init__(self, rows):
        self._meta = pd.DataFrame(rows)

    @property
    def columns(self):
        return [(MetaType.AUDIO, 'mix'), (MetaType.AUDIO, 'voice'), (MetaType.SCALAR, 'speaker'), (MetaType.META, 'note')]

    @property
    def meta(self):
        return self._meta

    def make_meta(self):
        pass


def _meta(tmp_path):
    rows = []
    for i in range(4):
        a, b = str(tmp_path / ('a%d.npy' % i)), str(tmp_path / ('b%d.npy' % i))
        np.save(a, G['dataset/mix/%d' % i])
        np.save(b, G['dataset/voice/%d' % i])
        rows.append({'mix': a, 'voice': b, 'speaker': i % 2, 'note': 'x'})
    return _Meta(rows)


@pytest.mark.parametrize('tag,kw', [('crop', dict(fix_len=2048, audio_mask=True)),
                                    ('shuffle', dict(fix_len=1024, fix_shuffle=True,
                                                     extra_features=[('mix', lambda x: np.abs(x).astype(np.float32))])),
                                    ('whole', dict())])
def test_speech_dataset_matches_reference(tmp_path, tag, kw):
    meta = _meta(tmp_path)
    assert meta.process_columns == meta.columns[:3] and meta.column_names == ['mix', 'voice', 'speaker', 'note']
    np.random.seed(11)
    ds = D.SpeechDataset(meta, **kw)
    assert len(ds) == 4
    for i in range(4):
        item = ds[i]
        assert len(item) == int(G['dataset/%s/nfields' % tag])
        for j, x in enumerate(item):
            assert np.array_equal(np.asarray(x), G['dataset/%s/%d/%d' % (tag, i, j)]), (tag, i, j)
    with pytest.raises(AssertionError):
        D.SpeechDataset(meta, extra_features=[('nope', abs)])
    assert [c[1] for c in D.SpeechDataset(meta, skip_audio=True).cols] == ['speaker']


def test_wav_loading_and_rate_check(tmp_path):
    from scipy.io import wavfile
    x = (np.sin(np.arange(2205) * 0.05) * 20000).astype(np.int16)
    p = str(tmp_path / 'c.wav')
    wavfile.write(p, 22050, np.stack([x, x], axis=1))
    meta = _meta(tmp_path)
    ds = D.SpeechDataset(meta)
    w = ds.load_audio(p)
    assert w.dtype == np.float32 and w.shape == (2205,) and np.allclose(w, x / 32768.0)
    wavfile.write(p, 16000, x)
    with pytest.raises(AssertionError):
        ds.load_audio(p)
    with pytest.raises(NotImplementedError):
        ds.load_audio('clip.flac')


def test_loader_end_to_end_and_ragged_batch(tmp_path):
    meta = _meta(tmp_path)
    np.random.seed(5)
    loader = D.SpeechDataLoader(D.SpeechDataset(meta, audio_mask=True), batch_size=4, num_workers=0, pin_memory=False)
    mix, voice, spk, mask = next(iter(loader))
    assert mix.shape == (4, 4000) and voice.shape == (4, 4000) and spk.dtype == torch.int64 and mask.shape == (4, 4000)
    lens = [3000, 2500, 4000, 2048]
    for n, l in enumerate(lens):
        assert np.array_equal(mix[n, :l].numpy(), G['dataset/mix/%d' % n]) and float(mix[n, l:].abs().sum()) == 0.0
        assert float(mask[n, :l].min()) == 1.0 and float(mask[n, l:].abs().sum()) == 0.0
    # the ragged collate keeps the same clips back to back; its host restatement of the padded layout == pad_collate_fn
    rl = D.SpeechDataLoader(D.SpeechDataset(meta), batch_size=4, num_workers=0, pin_memory=False, collate_fn=D.ragged_collate_fn)
    rmix, rvoice, rspk = next(iter(rl))
    assert isinstance(rmix, D.RaggedBatch) and rmix.lens.tolist() == lens and rmix.flat.numel() == sum(lens)
    assert torch.equal(rmix.padded(), mix) and torch.equal(rvoice.padded(), voice) and torch.equal(rspk, spk)
