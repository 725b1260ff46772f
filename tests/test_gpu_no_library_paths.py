"""A tensor that lives on the GPU never takes a library (rocBLAS / MIOpen / ATen softmax / rocFFT) path in the model-level
modules (policy; VERDICT r02 "silent library paths"): other floating dtypes are CAST to the kernels' fp32 and the result
cast back, geometries the kernels do not cover RAISE.  Each test also checks that the kernels really ran, by comparing the
non-fp32 call with the fp32 call on the same (rounded) operands - equal up to the output rounding - and by running under
`forbid_library_ops`, which makes the ATen ops the old fallbacks used raise for HIP tensors."""
from argparse import Namespace
from contextlib import contextmanager

import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode

pytestmark = pytest.mark.gpu

FORBIDDEN = ('aten.bmm', 'aten.baddbmm', 'aten.mm.', 'aten.addmm', 'aten._softmax', 'aten.native_group_norm', 'aten.convolution',
             'aten.miopen', 'aten.cudnn', 'aten._fft', 'aten.stft', 'aten.istft')


class _Forbid(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(name.startswith(f) for f in FORBIDDEN):
            flat = torch.utils._pytree.tree_leaves((args, kwargs or {}))
            if any(isinstance(a, torch.Tensor) and a.is_cuda for a in flat):
                raise AssertionError('library op %s reached with a HIP tensor' % name)
        return func(*args, **(kwargs or {}))


@contextmanager
def forbid_library_ops():
    with _Forbid():
        yield


def _dev():
    assert torch.cuda.is_available(), 'GPU test run without a GPU'
    return torch.device('cuda:0')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float64])
def test_transformer_block_casts_instead_of_falling_back(dtype):
    from pytorch_sound_amd.models.modules import MultiHeadAttention, PointwiseFeedForward, PositionalEncoding
    dev = _dev()
    torch.manual_seed(0)
    mha, ffn, pe = MultiHeadAttention(64, 4, 0.0).to(dev), PointwiseFeedForward(64, 0.0).to(dev), PositionalEncoding(64, 256).to(dev)
    x32 = torch.randn(3, 64, 50, device=dev).to(dtype).float()          # operands representable in `dtype`
    mask = torch.zeros(3, 50, dtype=torch.bool, device=dev)
    mask[1, 40:] = True
    with forbid_library_ops():
        y32, att32 = mha(pe(x32), mask)
        z32 = ffn(y32)
        y, att = mha(pe(x32).to(dtype), mask)
        z = ffn(y32.to(dtype))
    assert y.dtype == dtype and att.dtype == dtype and z.dtype == dtype
    tol = {torch.bfloat16: 2e-2, torch.float16: 2e-3, torch.float64: 1e-6}[dtype]
    assert (y.float() - y32).abs().max().item() <= tol * y32.abs().max().item() + tol
    assert (z.float() - ffn(y32.to(dtype).float())).abs().max().item() <= tol * z32.abs().max().item() + tol
    # with gradient
    xg = x32.clone().to(dtype).requires_grad_(True)
    with forbid_library_ops():
        out, _ = mha(xg, mask)
        ffn(out).float().sum().backward()
    assert xg.grad is not None and xg.grad.dtype == dtype and torch.isfinite(xg.grad.float()).all()


@pytest.mark.parametrize('C,H,T', [(256, 2, 50), (256, 2, 300), (384, 3, 77), (192, 2, 40)])
@pytest.mark.parametrize('bf16', [False, True])
def test_attention_head_dimension_128(C, H, T, bf16):
    """round 5 (VERDICT r04 missing 3): head dimensions 65 .. 128 (MultiHeadAttention(256, 2): reference modules.py:24-27 takes any
    hidden_dim / num_head) run on the HDP = 128 instances of psnd_mha_*: output, attention tensor, input and parameter gradients against
    the float64 torch formulation of the same module; masked batch, `att` in the loss.  fp32: 3e-5 / 2e-4 of max; bf16 operands under
    autocast: 3e-2."""
    import copy
    import pytorch_sound_amd.models.modules as M
    from pytorch_sound_amd.models.modules import MultiHeadAttention
    dev = _dev()
    torch.manual_seed(C + T)
    N = 3
    mha = MultiHeadAttention(C, H, 0.0).to(dev)
    lens = torch.tensor([T, max(T // 2, 3), max(T // 3, 2)])
    mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    x0 = (0.1 * torch.randn(N, C, T, device=dev)) * (~mask).unsqueeze(1)
    gy = torch.randn(N, C, T, device=dev)
    gatt = torch.randn(H * N, T, T, device=dev) * 0.1

    def run(m, dt, amp):
        m.zero_grad()
        x = x0.detach().clone().to(dt).requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            y, att = m(x, mask)
        ((y.to(dt) * gy.to(dt)).sum() + (att.to(dt) * gatt.to(dt)).sum()).backward()
        return y.detach().double(), att.detach().double(), x.grad.double(), {k: p.grad.double() for k, p in m.named_parameters()}

    with forbid_library_ops():
        y, att, gx, gp = run(mha, torch.float32, bf16)
    ref = copy.deepcopy(mha).double()
    keep = M._hip_ok
    M._hip_ok = lambda t: False                       # the torch formulation (bmm / softmax) in float64
    try:
        yr, attr, gxr, gpr = run(ref, torch.float64, False)
    finally:
        M._hip_ok = keep
    to, tg = (3e-2, 3e-2) if bf16 else (3e-5, 2e-4)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(y, yr) <= to and rel(att, attr) <= to, (rel(y, yr), rel(att, attr))
    assert rel(gx, gxr) <= tg, rel(gx, gxr)
    worst = max(rel(gp[k], gpr[k]) for k in gp)
    assert worst <= tg, worst


def test_attention_head_dimension_beyond_the_kernel_raises():
    from pytorch_sound_amd._lib import PsndError
    from pytorch_sound_amd.models.modules import MultiHeadAttention
    dev = _dev()
    x = torch.randn(2, 512, 20, device=dev)
    with pytest.raises(PsndError):
        MultiHeadAttention(512, 2, 0.0).to(dev)(x)                     # head dimension 256 > 128
    y, att = MultiHeadAttention(256, 4, 0.0).to(dev)(x[:, :256])        # 64: on psnd_mha_*
    assert tuple(y.shape) == (2, 256, 20) and tuple(att.shape) == (8, 20, 20)


def _gen(rates, ksz):
    from pytorch_sound_amd.models.vocoders.hifi_gan import Generator
    torch.manual_seed(5)
    h = Namespace(resblock='2', upsample_rates=rates, upsample_kernel_sizes=ksz, upsample_initial_channel=64,
                  resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1, 2]])
    return Generator(h).cuda()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_generator_casts_instead_of_falling_back(dtype):
    g = _gen([4, 2], [8, 4])
    x = torch.randn(2, 80, 16, device='cuda').to(dtype)
    with forbid_library_ops():
        y = g(x)
        y32 = g(x.float())
    assert y.dtype == dtype and tuple(y.shape) == (2, 1, 128)
    assert (y.float() - y32).abs().max().item() <= 1e-2                  # tanh output in [-1, 1]: one rounding of `dtype`


@pytest.mark.parametrize('rates,ksz', [([3, 2], [6, 4]), ([3, 5], [7, 11]), ([4, 2], [7, 4]), ([2, 2], [4, 6])])
def test_generator_upsamplers_beyond_the_polyphase_kernel_run_on_the_conv_kernels(rates, ksz):
    """round 5 (VERDICT r04 missing 3): ConvTranspose1d with an odd stride or k != 2 * stride (reference hifi_gan.py:107-110 takes any)
    used to raise on a HIP tensor; it now runs as a convolution over zero-spread rows on the CL conv kernels (cl.conv_transpose_cl) -
    the reference's output length (T - 1) * s - 2 * p + k per stage, values and gradients against the fp32 torch formulation of the same
    module (use_cl = False) at the tolerance of the bf16 conv stack, no library convolution on the way."""
    g = _gen(rates, ksz)
    x = torch.randn(2, 80, 12, device='cuda')
    w = torch.randn(1, device='cuda')
    xk = x.clone().requires_grad_(True)
    with forbid_library_ops():
        y = g(xk)
        (y * w).sum().backward()
    gk = {n: p.grad.clone() for n, p in g.named_parameters()}
    g.zero_grad(set_to_none=True)
    g.use_cl = False
    xt = x.clone().requires_grad_(True)
    yt = g(xt)
    (yt * w).sum().backward()
    assert y.shape == yt.shape
    T = 12
    for r, k in zip(rates, ksz):
        T = (T - 1) * r - 2 * ((k - r) // 2) + k
    assert y.shape[-1] == T
    assert (y - yt).abs().max().item() <= 3e-2                           # tanh output in [-1, 1], bf16 operands
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    assert rel(xk.grad, xt.grad) <= 0.1
    worst = max(rel(gk[n], p.grad) for n, p in g.named_parameters())
    assert worst <= 0.15, worst


def test_generator_input_rank_is_checked():
    with pytest.raises(RuntimeError):
        _gen([4, 2], [8, 4])(torch.randn(80, 16, device='cuda'))         # not (N, 80, T)


def test_pqmf_and_preemphasis_have_no_library_path():
    from pytorch_sound_amd._lib import PsndError
    from pytorch_sound_amd.models.transforms import PQMF, MelToMFCC
    from pytorch_sound_amd.models.sound import PreEmphasis
    dev = _dev()
    torch.manual_seed(1)
    pq = PQMF(4, 62).to(dev)
    x = torch.randn(2, 1, 4096, device=dev)
    with forbid_library_ops():
        a32 = pq.analysis(x)
        a16 = pq.analysis(x.bfloat16())
        s16 = pq.synthesis(a32.bfloat16())
        s32 = pq.synthesis(a32.bfloat16().float())
        p16 = PreEmphasis(0.97).to(dev)(x.half())
        p32 = PreEmphasis(0.97).to(dev)(x.half().float())
        m16 = MelToMFCC(13, 80).to(dev)(torch.randn(2, 80, 30, device=dev).bfloat16())
    assert a16.dtype == torch.bfloat16 and s16.dtype == torch.bfloat16 and p16.dtype == torch.float16 and m16.dtype == torch.bfloat16
    assert (a16.float() - pq.analysis(x.bfloat16().float())).abs().max().item() <= 2e-2 * a32.abs().max().item()
    assert (s16.float() - s32).abs().max().item() <= 2e-2 * s32.abs().max().item()
    assert (p16.float() - p32).abs().max().item() <= 2e-3 * p32.abs().max().item()
    with pytest.raises(PsndError):
        PQMF(32, 62).to(dev).analysis(x)                                  # 32 bands: beyond psnd_pqmf_*
    with pytest.raises(RuntimeError):
        pq.analysis(torch.randn(2, 2, 64, device=dev))                    # the reference's conv1d would refuse two channels too
    with pytest.raises(RuntimeError):
        PreEmphasis(0.97).to(dev)(torch.randn(2, 2, 64, device=dev))


def test_hot_path_flows_reach_no_library_op():
    """The flows of SURVEY 8 end to end under the dispatch-mode guard, forward AND backward: log-mel front end, STFT transform / inverse,
    multi_stft_loss, the config-2 separator with its fused loss, a HiFi-GAN generator, the fused optimizer step, the padding-mask helper
    - none of them may reach bmm / mm / convolution / softmax / group_norm / FFT ops with a HIP tensor."""
    from pytorch_sound_amd import optim as poptim
    from pytorch_sound_amd.models import build_model
    from pytorch_sound_amd.models import separator  # noqa: F401
    from pytorch_sound_amd.models.sound import multi_stft_loss
    from pytorch_sound_amd.models.transforms import LogMelSpectrogram, STFT, SpectrogramMasker
    dev = _dev()
    torch.manual_seed(11)
    wav = (0.1 * torch.randn(4, 16384, device=dev)).requires_grad_(True)
    with forbid_library_ops():
        fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50.0, 30.0, 0.0, 8000.0).to(dev)
        fe(wav).sum().backward()
        st = STFT(1024, 256).to(dev)
        mag, ph = st.transform(wav)
        st.inverse(mag, ph).pow(2).sum().backward()
        a, b, c = multi_stft_loss(wav, 0.1 * torch.randn(4, 16384, device=dev), [(1024, 600, 120), (2048, 1200, 240), (512, 240, 50)])
        (a + b + c).backward()
        SpectrogramMasker(1024, 256)(torch.ones(4, 16384, device=dev))
        model = build_model('conv_separator_voicebank', {'channels': 256, 'num_blocks': 1}).to(dev)
        opt = poptim.Adam(model.parameters(), lr=1e-4)
        m = torch.rand(4, 513, 65, device=dev)
        loss, _ = model.spectral_l1_loss(m, torch.rand_like(m), torch.randn(4, 80, 65, device=dev), fe._mel_plan(), 80, 1.0, 0.5, 1e-6,
                                         fe.min_db, fe.max_db)
        loss.backward()
        opt.step()
        g = _gen([4, 2], [8, 4])
        g(torch.randn(2, 80, 16, device=dev)).abs().mean().backward()
    assert wav.grad is not None and torch.isfinite(wav.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in g.parameters())


def test_config3_training_step_reaches_no_library_math():
    """the WHOLE config-3 training step (mel -> hifi_gan_v1-shaped generator -> mel of the output -> L1 -> backward ->
    fused Adam) under the guard, with the step's former library launches added to the list: tanh / tanh_backward (now inside the output
    layout change), leaky_relu, tensor adds / means on the bf16 CL buffers, abs / sign of the L1.  What remains library work in this step
    is memory initialisation (zero fills of halo rows / slabs) and the four bias-gradient column sums of the upsamplers (off the
    critical path, on the parameter branch)."""
    from pytorch_sound_amd import kernels as K, optim as poptim
    from pytorch_sound_amd.interface.hifi_gan import MelSpectrogram
    from pytorch_sound_amd.models.vocoders.hifi_gan import Generator
    dev = _dev()
    torch.manual_seed(2)
    h = Namespace(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=64,
                  resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    gen = Generator(h).to(dev).train()
    mel = MelSpectrogram().to(dev)
    opt = poptim.Adam(gen.parameters(), lr=2e-4, betas=(0.8, 0.99))
    wav = (0.07 * torch.randn(2, 8192, device=dev)).clamp(-1, 1)
    extra = ('aten.tanh', 'aten.leaky_relu', 'aten.abs', 'aten.sign', 'aten.sgn', 'aten.mean', 'aten.l1_loss', 'aten.add.Tensor', 'aten.add_.Tensor',
             'aten.mul.Tensor', 'aten.div.Tensor')

    class _ForbidMore(_Forbid):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if any(name.startswith(f) for f in extra):
                flat = torch.utils._pytree.tree_leaves((args, kwargs or {}))
                if any(isinstance(a, torch.Tensor) and a.is_cuda and a.numel() > 4096 for a in flat):       # (scalars / tiny host-side bookkeeping aside)
                    raise AssertionError('library op %s reached with a HIP tensor of %s' % (name, [tuple(a.shape) for a in flat if isinstance(a, torch.Tensor)]))
            return super().__torch_dispatch__(func, types, args, kwargs)

    losses = []
    with _ForbidMore():
        for _ in range(2):
            with torch.no_grad():
                m = mel(wav)
            y = gen(m).squeeze(1)
            loss = K.l1_loss(mel(y), m)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    assert all(l == l and l > 0 for l in losses)


def test_config4_training_step_reaches_no_library_math():
    """the whole training step of the BASELINE configs[3] block (1x1 -> PositionalEncoding -> MultiHeadAttention ->
    PointwiseFeedForward -> 1x1, padding mask, masked L1, Adam) under the guard config 3 has - no library GEMM / conv, and no library
    elementwise or reduction pass over an activation-sized HIP tensor either: the gradient of a block's input, which reaches it along
    the residual AND through the first projection, is summed in that projection's input-gradient GEMM (kernels.ResidualLink), not by
    autograd's `add`."""
    from pytorch_sound_amd.models import modules as M
    from pytorch_sound_amd import kernels as K, optim as poptim
    dev = _dev()
    torch.manual_seed(0)
    C, H, N, T = 64, 4, 3, 200

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inp, self.out = torch.nn.Conv1d(80, C, 1), torch.nn.Conv1d(C, 80, 1)
            self.pe, self.mha, self.ffn = M.PositionalEncoding(C, 512), M.MultiHeadAttention(C, H, 0.0), M.PointwiseFeedForward(C, 0.0)

        def forward(self, mel_, pad):
            x = self.pe(M._conv1x1(self.inp, mel_))
            x, _ = self.mha(x, pad)
            return M._conv1x1(self.out, self.ffn(x))

    net = Net().to(dev).train()
    net.mha.return_att = False
    opt = poptim.Adam(net.parameters(), lr=1e-4)
    lens = torch.tensor([200, 150, 90])
    valid = (torch.arange(T)[None, :] < lens[:, None]).float().to(dev)
    pad = valid < 0.5
    mel_ = torch.randn(N, 80, T, device=dev)
    extra = ('aten.add.Tensor', 'aten.add_.Tensor', 'aten.mul.Tensor', 'aten.div.Tensor', 'aten.sum', 'aten.mean', 'aten.abs', 'aten.sign', 'aten.sgn',
             'aten.relu', 'aten.threshold_backward', 'aten.native_group_norm', 'aten._softmax', 'aten.l1_loss', 'aten.where', 'aten.masked_fill')

    class _ForbidMore(_Forbid):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if any(name.startswith(f) for f in extra):
                flat = torch.utils._pytree.tree_leaves((args, kwargs or {}))
                if any(isinstance(a, torch.Tensor) and a.is_cuda and a.numel() > 4096 for a in flat):
                    raise AssertionError('library op %s reached with a HIP tensor of %s' % (name, [tuple(a.shape) for a in flat if isinstance(a, torch.Tensor)]))
            return super().__torch_dispatch__(func, types, args, kwargs)

    losses = []
    with _ForbidMore():
        for _ in range(2):
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = net(mel_, pad)
            loss = K.masked_l1_loss(y.float(), mel_, valid)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    assert all(l == l and l > 0 for l in losses)


def test_residual_link_gradient_equals_autograd_accumulation():
    """the input gradient of MultiHeadAttention / PointwiseFeedForward with the residual's share folded into the projection's GEMM
    (psnd_linear1x1_bwd_acc) against the same modules with the link off (autograd adds the two): equal to fp32 rounding of one add"""
    from pytorch_sound_amd.models import modules as M
    from pytorch_sound_amd import kernels as K
    dev = _dev()
    torch.manual_seed(1)
    mha, ffn = M.MultiHeadAttention(64, 4, 0.0).to(dev), M.PointwiseFeedForward(64, 0.0).to(dev)
    x0 = torch.randn(3, 64, 120, device=dev)
    w = torch.randn(3, 64, 120, device=dev)

    def run(link_on):
        orig = K.ResidualLink.offer
        if not link_on:
            K.ResidualLink.offer = lambda self, g: False
        try:
            x = x0.clone().requires_grad_(True)
            y, _ = mha(x, None)
            z = ffn(y)
            (z * w).sum().backward()
            grads = [x.grad.clone()] + [p.grad.clone() for p in list(mha.parameters()) + list(ffn.parameters())]
            for p in list(mha.parameters()) + list(ffn.parameters()):
                p.grad = None
            return grads
        finally:
            K.ResidualLink.offer = orig

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 2e-6 * float(v.abs().max()) + 1e-7


@pytest.mark.parametrize('bf16', [False, True])
@pytest.mark.parametrize('shape', [(3, 64, 120), (2, 96, 173), (1, 32, 7)])
def test_relu_link_gradients_are_the_unlinked_ones(shape, bf16, monkeypatch):
    """Conv1d -> ReLU -> Conv1d of PointwiseFeedForward (modules.py:93-95): the ReLU's backward in the epilogue of the second projection's
    input-gradient GEMM (psnd_linear1x1_bwd_ex, gx_mask) against each GEMM of the first projection masking its operand itself.  On the
    pair of projections alone the numbers are THE SAME (a masked element is an exact zero either way), with and without a gradient wanted
    for the input, and so is the whole block; with the torch formulation in fp32 it agrees at the GEMM's tolerance."""
    from pytorch_sound_amd.models import modules as M
    from pytorch_sound_amd import kernels as K
    import torch.nn.functional as F
    dev = _dev()
    torch.manual_seed(2)
    n, c, t = shape
    ffn = M.PointwiseFeedForward(c, 0.0).to(dev)
    x0, w = torch.randn(n, c, t, device=dev), torch.randn(n, c, t, device=dev)
    seen = []
    orig = K.ReluLink.take

    def take(self):
        seen.append(self.masked)
        return orig(self)
    monkeypatch.setattr(K.ReluLink, 'take', take)
    params = list(ffn.ff.parameters())

    def pair(on, need_x):
        monkeypatch.setattr(K, 'RELU_LINKS', on)
        x = x0.clone().requires_grad_(need_x)
        link = K.ReluLink()
        h = K.Linear1x1.apply(x, ffn.ff[0].weight, ffn.ff[0].bias, True, bf16, None, link)
        y = K.Linear1x1.apply(h, ffn.ff[2].weight, ffn.ff[2].bias, False, bf16, None, link)
        (y * w).sum().backward()
        g = ([x.grad.clone()] if need_x else []) + [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        return g

    for need_x in (True, False):
        del seen[:]
        a = pair(True, need_x)
        assert seen == [True], 'the second projection did not mask the gradient it produced'
        b = pair(False, need_x)
        assert len(a) == len(b) == 4 + int(need_x)
        for u, v in zip(a, b):
            assert torch.equal(u, v)

    monkeypatch.setattr(K, 'HIDDEN_BF16', False)          # (the bf16 hidden tensor has its own test below)

    def block(on):
        monkeypatch.setattr(K, 'RELU_LINKS', on)
        x = x0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            z = ffn(x)
        (z.float() * w).sum().backward()
        g = [x.grad.clone()] + [p.grad.clone() for p in ffn.parameters()]
        for p in ffn.parameters():
            p.grad = None
        return g

    del seen[:]
    a, b = block(True), block(False)
    assert seen == [True]
    for u, v in zip(a, b):
        assert torch.equal(u, v)                 # (the GroupNorm behind the pair adds its sums up in a fixed order since round 6: the whole block is the same bits)
    if not bf16:
        x = x0.clone().requires_grad_(True)
        h = ffn.ff[2](F.relu(ffn.ff[0](x)))
        z = F.relu(F.group_norm(h + x, 1, ffn.layernorm.weight, ffn.layernorm.bias, ffn.layernorm.eps))
        (z * w).sum().backward()
        ref = [x.grad] + [p.grad for p in ffn.parameters()]
        for u, v in zip(a, ref):
            assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-6


@pytest.mark.parametrize('shape', [(3, 64, 120), (2, 96, 173), (1, 32, 7), (2, 256, 431)])
def test_ffn_hidden_tensor_stored_as_bf16(shape, monkeypatch):
    """PointwiseFeedForward under torch.autocast(bfloat16): the tensor between the two projections (and its gradient) STORED as bf16
    (psnd_linear1x1_fwd_ex / _bwd_ex, io_h) against the same block with that tensor in fp32.  The products round their operands to bf16 when
    they load them, so the block's output, the input gradient and the weight gradients are THE SAME bits;
    only the first projection's bias gradient sums rounded values (bf16 rounding of each term).  Odd T (173: 2-byte aligned rows), a
    single short clip and channel counts off the tile size included; eval mode (no autograd) takes the bf16 tensor too."""
    from pytorch_sound_amd.models import modules as M
    from pytorch_sound_amd import kernels as K
    dev = _dev()
    torch.manual_seed(3)
    n, c, t = shape
    ffn = M.PointwiseFeedForward(c, 0.0).to(dev)
    x0, w = torch.randn(n, c, t, device=dev), torch.randn(n, c, t, device=dev)
    seen = []
    orig = K.Linear1x1.forward

    def fwd(ctx, x, *a):
        seen.append(x.dtype)
        return orig(ctx, x, *a)
    monkeypatch.setattr(K.Linear1x1, 'forward', staticmethod(fwd))

    def block(hidden_h):
        monkeypatch.setattr(K, 'HIDDEN_BF16', hidden_h)
        del seen[:]
        x = x0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            z = ffn(x)
        assert seen == [torch.float32, torch.bfloat16 if hidden_h else torch.float32]
        (z.float() * w).sum().backward()
        g = [z.detach().float().clone(), x.grad.clone()] + [p.grad.clone() for p in ffn.parameters()]
        for p in ffn.parameters():
            p.grad = None
        return g

    a, b = block(True), block(False)
    names = ['out', 'gx'] + [k for k, _ in ffn.named_parameters()]
    for k, u, v in zip(names, a, b):
        if k == 'ff.0.bias':
            assert float((u - v).abs().max()) <= 4e-3 * float(v.abs().max()) + 1e-7, k
        else:
            assert torch.equal(u, v), k          # the same products, a reproducible GroupNorm: the same bits
    monkeypatch.setattr(K, 'HIDDEN_BF16', True)
    del seen[:]
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        z = ffn(x0)
    assert seen == [torch.float32, torch.bfloat16]
    assert float((z.float() - a[0]).abs().max()) <= 2e-6 * float(a[0].abs().max())
