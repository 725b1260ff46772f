"""The bin-fastest (N, F, K) path around the conv stack (psnd_nfk.hip, the NFK operand paths of psnd_mel.hip): every kernel against
its frame-fastest (N, K, F) twin on the transposed tensors - bit-exact where the arithmetic is the same (layout change, mask head,
mel adjoint), to fp32 summation order where it is not (the mel forward walks its band in 16-bin groups) - against the float64 oracle
for the mel product, and the whole fused loss of the config-2 recipe in both layouts (value, est, every parameter gradient)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import features as ofe

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def relf(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('N,K,F_,HP', [(3, 513, 173, 25), (2, 513, 61, 5), (1, 257, 33, 1), (2, 80, 40, 3), (1, 16, 9, 0)])
def test_to_cl_nfk_equals_to_cl(N, K, F_, HP):
    from pytorch_sound_amd import cl
    torch.manual_seed(K + F_)
    x = (torch.rand(N, K, F_, device=DEV) * 5)
    x[0, :, 0] = 0
    x[0, 3, 1] = 1e-9                                  # log1p of a tiny value: the corrected formula
    shape = cl.CLShape(N, F_, HP)
    for preop in (0, 1):
        a = cl.ToCL.apply(x, shape, preop)
        b = cl.to_cl_nfk(x.transpose(1, 2).contiguous(), shape, preop)
        assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16))
    with pytest.raises(Exception):
        cl.to_cl_nfk(x.transpose(1, 2).contiguous().requires_grad_(True), shape, 1)


@pytest.mark.parametrize('sr,n_fft,M,fmin,fmax,F_', [(22050, 1024, 80, 0, 8000, 173), (22050, 1024, 80, 0, None, 32), (16000, 512, 40, 50, 7000, 61),
                                                     (44100, 4096, 128, 0, None, 70), (22050, 1024, 80, 0, 8000, 3)])
def test_mel_nfk_vs_oracle_and_nkf(sr, n_fft, M, fmin, fmax, F_):
    from pytorch_sound_amd import kernels as K
    W = ofe.mel_filterbank(sr, n_fft, M, fmin, fmax)
    g = np.random.RandomState(F_)
    mag = np.abs(g.randn(3, n_fft // 2 + 1, F_)).astype(np.float32) * 3
    mag[0, :, 0] = 0
    mag[1] *= 3.0e4
    lo, hi = ofe.db_to_ln(-50), ofe.db_to_ln(30)
    plan = K.mel_plan(W).to(DEV)
    m_nkf = torch.from_numpy(mag).to(DEV)
    m_nfk = m_nkf.transpose(1, 2).contiguous()
    out, lin = K.mel_forward_nfk(m_nfk, plan, M, K.LOG_E, 1e-6, None, lo, hi, want_lin=True)
    out0, lin0 = K.mel_forward(m_nkf, plan, M, K.LOG_E, 1e-6, None, lo, hi, want_lin=True)
    lin64 = np.matmul(W.astype(np.float64), mag.astype(np.float64))
    ref = np.clip(np.log(lin64 + 1e-6), lo, hi)
    assert np.abs(lin.cpu().numpy() - lin64).max() <= 2e-6 * np.abs(lin64).max()
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5
    assert relf(lin, lin0) <= 1e-6
    # the adjoint writes (N, F, K): same arithmetic as the (N, K, F) one, bit for bit
    gout = torch.from_numpy(g.randn(*out.shape).astype(np.float32)).to(DEV)
    g_nfk = K.mel_backward_nfk(gout, lin0, plan, n_fft // 2 + 1, K.LOG_E, 1e-6, None, lo, hi)
    g_nkf = K.mel_backward(gout, lin0, plan, n_fft // 2 + 1, K.LOG_E, 1e-6, None, lo, hi)
    assert torch.equal(g_nfk.transpose(1, 2), g_nkf)
    # autograd node
    mm = m_nfk.clone().requires_grad_(True)
    (K.MelLogNfk.apply(mm, plan, M, K.LOG_E, 1e-6, None, lo, hi) * gout).sum().backward()
    y = np.log(lin64 + 1e-6)
    dl = np.where((y < lo) | (y > hi), 0.0, 1.0 / (lin64 + 1e-6))
    gref = np.einsum('mk,nmf->nfk', W.astype(np.float64), gout.cpu().numpy() * dl)
    assert np.abs(mm.grad.cpu().numpy() - gref).max() <= 3e-6 * np.abs(gref).max()


def test_mel_nfk_dense_matrix_last_group():
    """a dense (non-banded) projection whose band reaches the last, partial 16-bin group (K = 513: one bin) - the element-load path -
    and K values around the group size"""
    from pytorch_sound_amd import kernels as K
    g = np.random.RandomState(1)
    for Kb, M, F_ in [(513, 24, 37), (17, 5, 20), (16, 16, 64), (31, 3, 5), (2049, 20, 19)]:
        W = g.randn(M, Kb).astype(np.float32)
        mag = g.randn(2, F_, Kb).astype(np.float32)
        plan = K.mel_plan(W).to(DEV)
        out, _ = K.mel_forward_nfk(torch.from_numpy(mag).to(DEV), plan, M, K.LOG_NONE)
        ref = np.einsum('mk,nfk->nmf', W.astype(np.float64), mag.astype(np.float64))
        assert np.abs(out.cpu().numpy() - ref).max() <= 4e-6 * np.abs(ref).max(), (Kb, M, F_)
        # NaN-filled neighbours must not leak in: the frame behind the last one of a clip, the bins past K
        buf = torch.full((2 * F_ * Kb + 64,), float('nan'), device=DEV)
        buf[:2 * F_ * Kb] = torch.from_numpy(mag).to(DEV).flatten()
        out2, _ = K.mel_forward_nfk(buf[:2 * F_ * Kb].view(2, F_, Kb), plan, M, K.LOG_NONE)
        assert torch.equal(out, out2)


@pytest.mark.parametrize('N,T,channels', [(32, 173, 256), (3, 61, 64)])
def test_fused_spectral_l1_loss_nfk_equals_nkf(N, T, channels):
    """ConvSeparator.spectral_l1_loss(layout='nfk') (cl.to_cl_nfk + cl.MaskHeadSpectralL1NFK) against layout='nkf' on the transposed
    tensors: est bit for bit, the value to 1e-6 (fp32 summation order of the partial sums and of the mel forward), the parameter
    gradients to 2e-3 of each tensor's largest entry (the conv stack in between is the same launches on the same bits up to the mel
    forward's summation order) - and the value against plain torch in float64."""
    from pytorch_sound_amd import kernels as K
    from pytorch_sound_amd.models import build_model
    from pytorch_sound_amd.models import separator  # noqa: F401
    from pytorch_sound_amd.models.transforms import LogMelSpectrogram
    torch.manual_seed(7 + T)
    model = build_model('conv_separator_voicebank', {'channels': channels, 'num_blocks': 2}).to(DEV)
    fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0.0, 8000.0).to(DEV)
    mag = torch.rand(N, 513, T, device=DEV) * 4
    mag_ref = torch.rand(N, 513, T, device=DEV) * 4
    mel_ref = K.MelLog.apply(mag_ref, fe._mel_plan(), 80, K.LOG_E, 1e-6, None, fe.min_db, fe.max_db)
    mel_ref2 = K.mel_forward_nfk(mag_ref.transpose(1, 2).contiguous(), fe._mel_plan(), 80, K.LOG_E, 1e-6, None, fe.min_db, fe.max_db)[0]
    assert float((mel_ref - mel_ref2).abs().max()) <= 2e-5

    def grads():
        return {k: p.grad.clone() for k, p in model.named_parameters()}

    model.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss_a, est_a = model.spectral_l1_loss(mag, mag_ref, mel_ref, fe._mel_plan(), 80, 1.0, 0.5, 1e-6, fe.min_db, fe.max_db)
    loss_a.backward()
    ga = grads()
    model.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss_b, est_b = model.spectral_l1_loss(mag.transpose(1, 2).contiguous(), mag_ref.transpose(1, 2).contiguous(), mel_ref, fe._mel_plan(), 80,
                                               1.0, 0.5, 1e-6, fe.min_db, fe.max_db, layout='nfk')
    loss_b.backward()
    gb = grads()
    assert tuple(est_b.shape) == (N, T, 513) and not est_b.requires_grad
    assert float((est_b.transpose(1, 2) - est_a).abs().max()) <= 2e-7 * float(est_a.abs().max())
    mel_est = K.MelLog.apply(est_a, fe._mel_plan(), 80, K.LOG_E, 1e-6, None, fe.min_db, fe.max_db)
    ref = float(F.l1_loss(est_a.double(), mag_ref.double()) + 0.5 * F.l1_loss(mel_est.double(), mel_ref.double()))
    assert abs(float(loss_b) - ref) <= 2e-6 * abs(ref), (float(loss_b), ref)
    assert abs(float(loss_b) - float(loss_a)) <= 1e-6 * abs(ref)
    for k in ga:
        assert relf(gb[k], ga[k]) <= 2e-3, (k, relf(gb[k], ga[k]))


def test_mask_head_nfk_kernels_vs_float64():
    """psnd_mask_head_l1_fwd_nfk / _bwd_nfk through the C ABI against the float64 formulas (est = sigmoid(y) mag, partial sums of
    |est - ref|, gy = (gest + c g sign(est - ref)) mag s (1 - s) rounded to bf16), halo rows and padded channels written as zeros."""
    from pytorch_sound_amd import cl
    from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
    torch.manual_seed(3)
    N, F_, Kb, HP = 3, 29, 513, 4
    shape = cl.CLShape(N, F_, HP)
    Cp = cl.round_up(Kb, cl.ALIGN_C)
    y = torch.randn(N, shape.Lp, Cp, device=DEV).bfloat16()
    mag = torch.rand(N, F_, Kb, device=DEV) * 3
    ref = torch.rand(N, F_, Kb, device=DEV) * 3
    est = torch.full_like(mag, float('nan'))
    nb = int(lib().psnd_mask_head_l1_blocks_nfk(N, F_, Kb))
    part = torch.full((nb,), float('nan'), dtype=torch.float64, device=DEV)
    check(lib().psnd_mask_head_l1_fwd_nfk(ptr(y), ptr(mag), ptr(ref), N, Kb, F_, shape.Lp, HP, Cp, ptr(est), ptr(part), stream_ptr(DEV)), 'fwd')
    yy = y[:, HP:HP + F_, :Kb].double()
    e64 = torch.sigmoid(yy) * mag.double()
    assert float((est.double() - e64).abs().max()) <= 2e-6 * float(e64.abs().max())        # __expf: 2 ulp
    assert abs(float(part.sum()) - float((est.double() - ref.double()).abs().sum())) <= 1e-7 * float(e64.abs().sum())      # 8 terms per thread in fp32, the rest in double
    gest = torch.randn_like(mag)
    g = torch.tensor([0.7], device=DEV)
    gy = torch.full_like(y, float('nan'))
    check(lib().psnd_mask_head_l1_bwd_nfk(ptr(gest), ptr(mag), ptr(y), ptr(est), ptr(ref), ptr(g), 0.25, N, Kb, F_, shape.Lp, HP, Cp, ptr(gy),
                                          stream_ptr(DEV)), 'bwd')
    s = torch.sigmoid(yy)
    want = (gest.double() + 0.25 * 0.7 * torch.sign(est.double() - ref.double())) * mag.double() * s * (1 - s)
    got = gy.float()
    assert torch.isfinite(got).all()
    assert float((got[:, HP:HP + F_, :Kb].double() - want).abs().max()) <= 2 ** -8 * float(want.abs().max())      # one bf16 rounding
    assert float(got[:, :HP].abs().max()) == 0 and float(got[:, HP + F_:].abs().max()) == 0 and float(got[:, :, Kb:].abs().max()) == 0
