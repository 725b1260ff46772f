"""cProfile of the EAGER drop-in step (bench._dropin_leg's Step): where the host spends its ~2 ms per step.  argv: steps"""
import cProfile, pstats, io, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import bench
from pytorch_sound.models import build_model
from pytorch_sound.models.transforms import LogMelSpectrogram
from pytorch_sound.trainer import Trainer, LogType
from pytorch_sound_amd.models import separator  # noqa
from pytorch_sound_amd import optim as poptim
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda:0')
T, N = int(bench.SR * bench.CLIP_SECONDS), bench.BATCH_PER_GPU
fe = LogMelSpectrogram(bench.SR, bench.N_MEL, bench.N_FFT, bench.N_FFT, bench.HOP, -50, 30, bench.FMIN, bench.FMAX).to(dev)


class Step(Trainer):
    def forward(self, both, is_logging=False):
        n = both.shape[0] // 2
        noisy, clean = both[:n], both[n:]
        with torch.no_grad():
            mag_ref, _ = fe.stft.transform(clean)
            mel_ref = fe(clean)
        mag_mix, _ = fe.stft.transform(noisy)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            est = self.model(mag_mix)
        est = est.float()
        mel_est = torch.log(torch.matmul(fe.mel_filter, est) + 1e-6).clamp(fe.min_db, fe.max_db)
        loss = F.l1_loss(est, mag_ref) + 0.5 * F.l1_loss(mel_est, mel_ref)
        return loss, {'loss': (loss, LogType.SCALAR)}


torch.manual_seed(1234)
model = build_model('conv_separator_voicebank').to(dev)
pool = [bench.synth_batch(1234 + 1000 * i, N, T, dev) for i in range(4)]
tr = Step(model, poptim.Adam(model.parameters(), lr=2e-4, betas=(0.8, 0.99)), pool, pool, max_step=10 ** 9, valid_max_step=1,
          save_interval=10 ** 9, log_interval=10 ** 9, save_dir=tempfile.mkdtemp(prefix='psnd_hp_'), seed=1234)
model.train()
s = 0
for _ in range(20):
    s += 1; tr.step = s; tr.train(s)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    s += 1; tr.step = s; tr.train(s)
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats('tottime')
st.print_stats(45)
txt = out.getvalue()
print('\n'.join(l[:170] for l in txt.splitlines()[:70]))
out2 = io.StringIO(); pstats.Stats(pr, stream=out2).sort_stats('cumulative').print_stats(40)
print('\n'.join(l[:170] for l in out2.getvalue().splitlines()[5:55]))
