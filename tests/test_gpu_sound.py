"""models/sound.py on the GPU: psnd_preemphasis_* and the multi_stft_loss path (psnd_stft_fwd/bwd +
psnd_stft_loss_partial/final/bwd) through the drop-in functions, against
  * the imported reference's values and gradients (tests/golden/sound.npz),
  * the float64 oracle (oracle/sound.py) on fresh seeded inputs incl. the config-3 shape (16 x 8192),
  * size-independent properties at a large batch (loss(x, x) = 0 exactly; scaling law of the spectral convergence).
Tolerances (fp32 kernels vs f64 oracle): loss terms 2e-5 relative, waveform gradient 2e-4 of its max (1e-2 with the default eps, see below)."""
import ctypes
import numpy as np
import pytest
import torch

from conftest import seeded_wav
from oracle import sound as osnd

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PARAMS = [(1024, 600, 120), (2048, 1200, 240), (512, 240, 50)]


def test_preemphasis_golden_and_oracle(golden):
    from pytorch_sound_amd.models.sound import PreEmphasis
    g = golden('sound')
    pe = PreEmphasis(0.97).to(DEV)
    x = torch.from_numpy(g['preemph/x']).to(DEV).requires_grad_(True)
    y = pe(x)
    (y * torch.from_numpy(g['preemph/gy']).to(DEV)).sum().backward()
    assert np.abs(y.detach().cpu().numpy() - g['preemph/y']).max() < 1e-6
    assert np.abs(x.grad.cpu().numpy() - g['preemph/gx']).max() < 2e-6
    # ragged sizes: T not a multiple of the 1024-sample workgroup span, T = 2 (smallest the reflect pad allows)
    for N, T in ((5, 1025), (1, 2), (3, 4099)):
        xs = np.random.RandomState(T).randn(N, 1, T).astype(np.float32)
        gy = np.random.RandomState(T + 1).randn(N, 1, T).astype(np.float32)
        xt = torch.from_numpy(xs).to(DEV).requires_grad_(True)
        yt = pe(xt)
        (yt * torch.from_numpy(gy).to(DEV)).sum().backward()
        assert np.abs(yt.detach().cpu().numpy() - osnd.pre_emphasis(xs)).max() < 1e-6
        assert np.abs(xt.grad.cpu().numpy() - osnd.pre_emphasis_bwd(gy)).max() < 2e-6
    with pytest.raises(Exception):
        pe(torch.zeros(1, 1, 1, device=DEV))                      # reflect padding needs T >= 2


@pytest.mark.parametrize('case,params,eps', [('msl', PARAMS, 1e-5), ('msl1', [(256, 200, 64)], 1e-3)])
def test_multi_stft_loss_golden(golden, case, params, eps):
    from pytorch_sound_amd.models.sound import multi_stft_loss
    g = golden('sound')
    pred = torch.from_numpy(g[case + '/pred']).to(DEV).requires_grad_(True)
    target = torch.from_numpy(g[case + '/target']).to(DEV)
    loss, sc, mag = multi_stft_loss(pred, target, params, eps)
    loss.backward()
    got = np.asarray([float(loss), float(sc), float(mag)])
    assert np.allclose(got, g[case + '/loss'], rtol=2e-5), (got, g[case + '/loss'])
    ref = g[case + '/gpred']
    assert np.abs(pred.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()       # reference gradient is fp32


# The log-magnitude term's gradient sign(.)/(|X| + eps) * X/|X| is ill-conditioned where |X| falls to the fp32 noise floor
# of the transform (|X| ~ 1e-5 against a frame energy of O(1): 1 % error in |X| and in the direction, amplified by
# 1/(|X| + eps) ~ 1e5): with the reference's default eps = 1e-5 fp32 and f64 agree to 1e-2 of the gradient's max (the
# reference's own fp32 gradient sits at the same distance, see the golden test); with eps = 1e-2 the bar is 2e-4.
@pytest.mark.parametrize('N,T,eps,gtol', [(16, 8192, 1e-5, 1e-2), (16, 8192, 1e-2, 2e-4), (2, 3001, 1e-2, 2e-4)])
def test_multi_stft_loss_vs_oracle(N, T, eps, gtol):
    from pytorch_sound_amd.models.sound import multi_stft_loss
    t = seeded_wav(600 + N, N, T)
    p = (0.8 * t + 0.05 * seeded_wav(700 + N, N, T)).astype(np.float32)
    pred = torch.from_numpy(p).to(DEV).requires_grad_(True)
    target = torch.from_numpy(t).to(DEV).requires_grad_(True)
    loss, sc, mag = multi_stft_loss(pred, target, PARAMS, eps)
    # gradient of a weighted combination of the three outputs: exercises every upstream-gradient slot
    (0.5 * loss + 2.0 * sc - 0.25 * mag).backward()
    # the kernels use each module's own fp32 hann window, centre-padded - hand the same taps to the oracle
    from pytorch_sound_amd.models.sound import build_stft_functions
    from pytorch_sound_amd.models.transforms import centre_pad
    wins = [centre_pad(f.window.numpy().astype(np.float64), f.n_fft) for f in build_stft_functions(*PARAMS)]
    want = osnd.multi_stft_loss(p, t, PARAMS, eps, windows=wins)
    got = [float(loss), float(sc), float(mag)]
    assert np.allclose(got, want, rtol=2e-5), (got, want)
    # oracle gradient of the same combination: loss = (sc+mag)/L summed -> coefficients on sc_i and mag_i
    L = len(PARAMS)
    from oracle import features as fe
    gp = np.zeros_like(p, dtype=np.float64)
    gt = np.zeros_like(p, dtype=np.float64)
    for (n_fft, win, hop), w in zip(PARAMS, wins):
        pm = osnd.stft_mag_torchaudio_f64(p, n_fft, win, hop, w)
        tm = osnd.stft_mag_torchaudio_f64(t, n_fft, win, hop, w)
        a, b = osnd.stft_loss_terms_bwd(pm, tm, (0.5 + 2.0) / L, (0.5 - 0.25) / L, eps)
        gp += fe.stft_mag_bwd_f64(a, p, n_fft, hop, framing=fe.CENTER, window=w)
        gt += fe.stft_mag_bwd_f64(b, t, n_fft, hop, framing=fe.CENTER, window=w)
    assert np.abs(pred.grad.cpu().numpy() - gp).max() <= gtol * np.abs(gp).max()
    assert np.abs(target.grad.cpu().numpy() - gt).max() <= gtol * np.abs(gt).max()


@pytest.mark.parametrize('N,T,eps,gtol', [(16, 8192, 1e-2, 2e-4), (3, 20000, 1e-2, 2e-4), (2, 3001, 1e-2, 2e-4), (5, 8192, 1e-5, 1e-2)])
def test_multi_stft_loss_fused_gradient(N, T, eps, gtol, monkeypatch):
    """Training case (only the prediction needs a gradient): psnd_stft_bwd_msl forms d loss / d |X| inside the adjoint STFT.  Against
    the f64 oracle and against the two-launch path (psnd_stft_loss_bwd + psnd_stft_bwd) on the same inputs."""
    from pytorch_sound_amd import kernels as K
    from pytorch_sound_amd.models.sound import multi_stft_loss, build_stft_functions
    from pytorch_sound_amd.models.transforms import centre_pad
    from oracle import features as fe
    for n_fft, _, hop in PARAMS:
        assert K.lib().psnd_stft_bwd_msl_supported(n_fft, hop) == 1
    t = seeded_wav(900 + N, N, T)
    p = (0.8 * t + 0.05 * seeded_wav(950 + N, N, T)).astype(np.float32)
    target = torch.from_numpy(t).to(DEV)
    grads = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('PSND_MSL_FUSED', fused)
        pred = torch.from_numpy(p).to(DEV).requires_grad_(True)
        loss, sc, mag = multi_stft_loss(pred, target, PARAMS, eps)
        (0.5 * loss + 2.0 * sc - 0.25 * mag).backward()
        grads[fused] = pred.grad.cpu().numpy()
        grads['v' + fused] = [float(loss.detach()), float(sc.detach()), float(mag.detach())]
    wins = [centre_pad(f.window.numpy().astype(np.float64), f.n_fft) for f in build_stft_functions(*PARAMS)]
    L = len(PARAMS)
    gp = np.zeros_like(p, dtype=np.float64)
    for (n_fft, win, hop), w in zip(PARAMS, wins):
        pm = osnd.stft_mag_torchaudio_f64(p, n_fft, win, hop, w)
        tm = osnd.stft_mag_torchaudio_f64(t, n_fft, win, hop, w)
        a, _ = osnd.stft_loss_terms_bwd(pm, tm, (0.5 + 2.0) / L, (0.5 - 0.25) / L, eps)
        gp += fe.stft_mag_bwd_f64(a, p, n_fft, hop, framing=fe.CENTER, window=w)
    assert np.abs(grads['1'] - gp).max() <= gtol * np.abs(gp).max()
    assert np.abs(grads['1'] - grads['0']).max() <= gtol * np.abs(gp).max()
    # the forward sums left by psnd_stft_fwd_msl against psnd_stft_loss_partial's, and against the oracle
    want = osnd.multi_stft_loss(p, t, PARAMS, eps, windows=wins)
    assert np.allclose(grads['v1'], grads['v0'], rtol=1e-5), (grads['v1'], grads['v0'])
    assert np.allclose(grads['v1'], want, rtol=2e-5), (grads['v1'], want)


def test_multi_stft_loss_properties_large():
    """N = 256 clips x 16384 samples (3 resolutions: 0.5 GB of magnitudes): loss(x, x) has sc = mag = 0 exactly;
    pred = a * target gives sc = |1 - a| for every clip and mag = |log a| up to the eps inside the logs."""
    from pytorch_sound_amd.models.sound import multi_stft_loss
    x = torch.from_numpy(seeded_wav(800, 256, 16384)).to(DEV)
    loss, sc, mag = multi_stft_loss(x, x.clone(), PARAMS)
    assert float(loss) == 0.0 and float(sc) == 0.0 and float(mag) == 0.0
    a = 0.5
    loss, sc, mag = multi_stft_loss(a * x, x, PARAMS)
    assert float(sc) == pytest.approx(1 - a, rel=1e-5)
    assert float(mag) == pytest.approx(abs(np.log(a)), rel=2e-3)         # eps = 1e-5 against magnitudes of O(1e-2 .. 10)
    assert float(loss) == pytest.approx(float(sc) + float(mag), rel=1e-6)


def test_loss_kernels_reject_bad_arguments():
    from pytorch_sound_amd import _lib
    from pytorch_sound_amd.models.sound import multi_stft_loss
    lib = _lib.lib()
    assert lib.psnd_stft_loss_blocks(0) == 0 and lib.psnd_stft_loss_blocks(8192) == 1 and lib.psnd_stft_loss_blocks(8193) == 2
    assert lib.psnd_stft_loss_partial(None, None, 1, 10, ctypes.c_float(1e-5), None, None) == -1
    with pytest.raises(_lib.PsndError):
        multi_stft_loss(torch.zeros(2, 4096, device=DEV), torch.zeros(2, 4000, device=DEV), PARAMS)
    with pytest.raises(_lib.PsndError):
        multi_stft_loss(torch.zeros(2, 4096, device=DEV), torch.zeros(2, 4096, device=DEV), [(1000, 600, 120)])   # n_fft not 2^k


def test_l1_loss_sum_matches_torch():
    """psnd_l1_loss_sum_fwd / psnd_l1_loss_bwd_w: w1 * l1(a, b) + w2 * l1(c, d) as one node == the same expression on F.l1_loss in
    float64 (value 1e-6 relative, gradients 1e-6 relative; a target that needs no gradient gets none)"""
    from pytorch_sound_amd import kernels as K
    torch.manual_seed(5)
    a = torch.randn(3, 513, 173, device=DEV, requires_grad=True)
    b = torch.randn(3, 513, 173, device=DEV)
    c = torch.randn(3, 80, 173, device=DEV, requires_grad=True)
    d = torch.randn(3, 80, 173, device=DEV, requires_grad=True)
    out = K.l1_loss_sum([(a, b), (c, d)], (1.0, 0.5))
    (2.0 * out).backward()
    a64, c64, d64 = (t.detach().double().requires_grad_(True) for t in (a, c, d))
    ref = torch.nn.functional.l1_loss(a64, b.double()) + 0.5 * torch.nn.functional.l1_loss(c64, d64)
    (2.0 * ref).backward()
    assert abs(float(out) - float(ref)) <= 1e-6 * abs(float(ref))
    for g, r in ((a.grad, a64.grad), (c.grad, c64.grad), (d.grad, d64.grad)):
        assert torch.allclose(g.double(), r, rtol=1e-6, atol=0)
    assert b.grad is None
    with pytest.raises(Exception):
        K.l1_loss_sum([(a, b)] * 5, (1.0,) * 5)


@pytest.mark.parametrize('shape', [(32, 513, 173), (3, 7), (1, 16385), (5,)])
def test_l1_loss_matches_torch(shape):
    """psnd_l1_loss_fwd/bwd == F.l1_loss (mean) in float64 and its autograd (sign(a - b) / n, 0 at ties)"""
    from pytorch_sound_amd import kernels as K
    torch.manual_seed(len(shape))
    a = torch.randn(*shape, device=DEV, requires_grad=True)
    b = torch.randn(*shape, device=DEV, requires_grad=True)
    with torch.no_grad():
        b.view(-1)[0] = a.view(-1)[0]                     # a tie: gradient 0 there, as torch.sign gives
    out = K.l1_loss(a, b)
    (3.0 * out).backward()
    a64, b64 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    ref = torch.nn.functional.l1_loss(a64, b64)
    (3.0 * ref).backward()
    assert abs(float(out) - float(ref)) <= 1e-6 * abs(float(ref))
    assert torch.allclose(a.grad.double(), a64.grad, rtol=1e-6, atol=0) and torch.allclose(b.grad.double(), b64.grad, rtol=1e-6, atol=0)
    with pytest.raises(Exception):
        K.l1_loss(a, torch.zeros(2, device=DEV))


def test_masked_l1_loss_matches_torch_f64():
    """kernels.masked_l1_loss (psnd_masked_l1_fwd / _bwd) against the padded-batch recipe's formulation in float64: value to 1e-6, both
    gradients to 1e-6 of their largest entry; an all-ones weight equals F.l1_loss"""
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    for N, C, T in [(32, 80, 1292), (3, 5, 7), (1, 1, 4097)]:
        a = torch.randn(N, C, T, device=dev, requires_grad=True)
        b = torch.randn(N, C, T, device=dev, requires_grad=True)
        lens = torch.randint(1, T + 1, (N,))
        w = (torch.arange(T)[None, :] < lens[:, None]).float().to(dev)
        loss = K.masked_l1_loss(a, b, w)
        (loss * 1.7).backward()
        a64, b64 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        ref = ((a64 - b64).abs() * w.double().unsqueeze(1)).sum() / (w.double().sum() * C)
        (ref * 1.7).backward()
        assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
        assert float((a.grad.double() - a64.grad).abs().max()) <= 1e-6 * float(a64.grad.abs().max())
        assert float((b.grad.double() - b64.grad).abs().max()) <= 1e-6 * float(b64.grad.abs().max())
        ones = torch.ones(N, T, device=dev)
        assert abs(float(K.masked_l1_loss(a.detach(), b.detach(), ones)) - float(torch.nn.functional.l1_loss(a64, b64))) <= 1e-6
