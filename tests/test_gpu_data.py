"""Device side of the data path: psnd_pad_collate against the host padding (bit-exact: copies and zeros only) incl.
ragged / unaligned sizes, and DevicePrefetcher over a SpeechDataLoader (side-stream copy, event hand-over)."""
import numpy as np
import pytest
import torch

from pytorch_sound_amd.data import dataset as D

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.mark.parametrize('lens,multiple', [([700, 1000, 333, 1000, 512], 1), ([1, 4097, 2, 4096], 1), ([5, 3], 1),
                                           ([44100, 330750, 170000], 256), ([1023] * 7, 4)])
def test_pad_collate_kernel_bit_exact(lens, multiple):
    rs = np.random.RandomState(len(lens))
    clips = [rs.randn(n).astype(np.float32) for n in lens]
    rb = D.RaggedBatch(clips)
    out, mask = rb.to_device(DEV, want_mask=True, multiple=multiple)
    Tmax = (max(lens) + multiple - 1) // multiple * multiple
    assert out.shape == (len(lens), Tmax) and mask.shape == out.shape
    want = np.zeros((len(lens), Tmax), np.float32)
    wmask = np.zeros_like(want)
    for n, c in enumerate(clips):
        want[n, :len(c)] = c
        wmask[n, :len(c)] = 1
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(mask.cpu().numpy(), wmask)
    assert np.array_equal(rb.to_device(DEV, multiple=multiple).cpu().numpy(), want)
    assert np.array_equal(rb.padded().numpy(), want[:, :max(lens)])


def test_pad_collate_rejects_bad_arguments():
    from pytorch_sound_amd import _lib
    assert _lib.lib().psnd_pad_collate(None, None, None, 1, 8, None, None, None) == -1


class _Ragged(torch.utils.data.Dataset):
    def __init__(self, n):
        rs = np.random.RandomState(1)
        self.items = [[rs.randn(int(rs.randint(2000, 9000))).astype(np.float32), int(i)] for i in range(n)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_device_prefetcher_matches_host_collate():
    ds = _Ragged(40)
    host = torch.utils.data.DataLoader(ds, batch_size=8, collate_fn=D.SpeechDataLoader.pad_collate_fn)
    dev = D.DevicePrefetcher(torch.utils.data.DataLoader(ds, batch_size=8, collate_fn=D.ragged_collate_fn), DEV, want_mask=True)
    assert len(dev) == 5
    n = 0
    for (hw, hi), (dw, di, dm) in zip(host, dev):
        assert dw.device.type == 'cuda' and torch.equal(dw.cpu(), hw) and torch.equal(di.cpu(), hi)
        lens = [len(ds.items[int(i)][0]) for i in hi]
        assert dm.sum(1).cpu().tolist() == [float(l) for l in lens]
        # consume on the compute stream like a training step would
        assert float((dw * dm).abs().sum()) == pytest.approx(float(hw.abs().sum()), rel=1e-5)
        n += 1
    assert n == 5
