#!/usr/bin/env python
"""BASELINE config 4 end to end on ONE GPU: variable-length clips (2-15 s at 22.05 kHz, sorted, 5 length buckets), batch 32 drawn
inside a bucket (BucketRandomBatchSampler), ragged pinned batch -> side-stream copy -> psnd_pad_collate (DevicePrefetcher, Tmax
rounded to 4096 samples so that step shapes repeat), 80-mel log-mel features (psnd_logmel_fwd), 1x1 projection ->
PositionalEncoding -> MultiHeadAttention(256, 4) -> PointwiseFeedForward with the padding mask, masked L1, Adam; steps replayed
as hipGraphs per batch shape (Trainer.graph_steps, bounded cache).  Prints ms/step and audio-s/s (real, unpadded seconds)."""
import os
import sys
import tempfile
import time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sound_amd.data import dataset as D  # noqa: E402
from pytorch_sound_amd.models.modules import MultiHeadAttention, PointwiseFeedForward, PositionalEncoding, _conv1x1  # noqa: E402
from pytorch_sound_amd.models.transforms import LogMelSpectrogram, SpectrogramMasker  # noqa: E402
from pytorch_sound_amd.trainer import Trainer, LogType  # noqa: E402
from pytorch_sound_amd import optim as poptim  # noqa: E402

dev = torch.device('cuda:0')
SR, HOP, NB = 22050, 256, 32
MULT = int(os.environ.get('MULT', 22016))     # Tmax granularity (a multiple of the hop: 86 frames ~ 1 s): step shapes repeat -> few graphs


class Clips(torch.utils.data.Dataset):
    def __init__(self, n=1200, seed=0):
        rs = np.random.RandomState(seed)
        lens = np.sort((rs.uniform(2.0, 15.0, n) * SR).astype(np.int64))          # sorted by length, like the meta frames
        self.items = [(0.07 * rs.randn(int(l))).astype(np.float32) for l in lens]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return [self.items[i]]


class Net(torch.nn.Module):
    def __init__(self, C=256, H=4):
        super().__init__()
        self.inp = torch.nn.Conv1d(80, C, 1)
        self.pe = PositionalEncoding(C, 2048)
        self.mha = MultiHeadAttention(C, H, 0.0)
        self.ffn = PointwiseFeedForward(C, 0.0)
        self.out = torch.nn.Conv1d(C, 80, 1)

    def forward(self, mel, pad_mask):
        x = self.pe(_conv1x1(self.inp, mel))
        x, _ = self.mha(x, pad_mask)
        return _conv1x1(self.out, self.ffn(x))


def main():
    torch.manual_seed(0)
    np.random.seed(0)
    ds = Clips()
    loader = D.SpeechDataLoader(ds, batch_size=NB, num_workers=0, n_buckets=5, is_bucket=True, pin_memory=False,
                                collate_fn=D.ragged_collate_fn)
    feats = LogMelSpectrogram(SR, 80, 1024, 1024, HOP, -80.0, 20.0, 0.0, 8000.0).to(dev)
    masker = SpectrogramMasker(1024, HOP)
    model = Net().to(dev)

    class Step(Trainer):
        def prepare(self, wav, wav_mask):
            with torch.no_grad():
                mel = feats(wav)                                   # (N, 80, F)
                valid = masker(wav_mask)[:, :mel.shape[2]]          # (N, F) 1 = frame holds signal
            return mel, valid

        def forward(self, mel, valid, is_logging=False):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled='--amp' in sys.argv):   # bf16 projection and attention operands
                y = self.model(mel, valid < 0.5)
            from pytorch_sound_amd import kernels as K
            loss = K.masked_l1_loss(y.float(), mel, valid)             # psnd_masked_l1_* instead of abs / mul / sum / sum / div
            return loss, {'loss': (loss, LogType.SCALAR)}

    def batches():
        while True:
            np.random.seed(int(time.time() * 1e3) % (1 << 31))
            for b in D.DevicePrefetcher(loader, dev, want_mask=True, multiple=MULT):
                yield b

    opt = poptim.Adam(model.parameters(), lr=1e-4)
    tr = Step(model, opt, batches(), batches(), max_step=10 ** 9, valid_max_step=1, save_interval=10 ** 9, log_interval=10 ** 9,
              save_dir=tempfile.mkdtemp(prefix='psnd_c4_'), seed=1)
    tr.graph_steps, tr.graph_warmup, tr.graph_cache_size = '--eager' not in sys.argv, 1, 48
    model.train()
    secs = []
    orig = tr.prepare

    def counting(wav, wav_mask):
        secs.append(wav_mask.sum() / SR)
        return orig(wav, wav_mask)
    tr.prepare = counting
    s = 0
    for _ in range(int(os.environ.get('WARM', 150))):                                             # warm-up: captures one graph per batch shape seen
        s += 1
        tr.step = s
        tr.train(s)
    torch.cuda.synchronize()
    secs.clear()
    t0 = time.perf_counter()
    steps = 60
    for _ in range(steps):
        s += 1
        tr.step = s
        tr.train(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio = float(torch.stack(secs).sum())
    print('config 4 (%s): %.2f ms/step, %.1f k audio-s/s (unpadded); %d step graphs cached' % (
        'hipGraph' if tr.graph_steps else 'eager', dt / steps * 1e3, audio / dt / 1e3,
        len([1 for v in getattr(tr, '_graphs', {}).values() if 'graph' in v])))


if __name__ == '__main__':
    main()
