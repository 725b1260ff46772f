"""Transformer blocks on the GPU: psnd_groupnorm1_* and psnd_softmax_keys_* inside MultiHeadAttention /
PointwiseFeedForward against the imported reference's outputs and gradients (tests/golden/modules.npz) and against
the torch formulation at BASELINE config-4-like sizes.  fp32 throughout: tolerance 2e-5 of max (outputs),
1e-4 (gradients; library GEMM reassociation + fast exp)."""
import pytest
import torch

from test_modules_golden import sd_from, close

pytestmark = pytest.mark.gpu


def test_mha_ffn_golden_on_gpu(golden):
    from pytorch_sound_amd.models.modules import MultiHeadAttention, PointwiseFeedForward
    dev = torch.device('cuda:0')
    g = golden('modules')
    for tag in ('nomask', 'mask'):
        mha = MultiHeadAttention(16, 4, 0.0)
        mha.load_state_dict(sd_from(g, 'mha/sd/'))
        mha.to(dev)
        x = torch.from_numpy(g['mha/x']).to(dev).requires_grad_(True)
        mask = torch.from_numpy(g['mha/mask']).to(dev) if tag == 'mask' else None
        y, att = mha(x, mask)
        (y * torch.from_numpy(g['mha/g']).to(dev)).sum().backward()
        assert close(y, g['mha/%s/y' % tag]) and close(att, g['mha/%s/att' % tag])
        assert close(x.grad, g['mha/%s/gx' % tag], 1e-4)
        for k, p in mha.named_parameters():
            assert close(p.grad, g['mha/%s/g/%s' % (tag, k)], 1e-4), k
    ffn = PointwiseFeedForward(16, 0.0)
    ffn.load_state_dict(sd_from(g, 'ffn/sd/'))
    ffn.to(dev)
    x = torch.from_numpy(g['mha/x']).to(dev).requires_grad_(True)
    y = ffn(x)
    (y * torch.from_numpy(g['mha/g']).to(dev)).sum().backward()
    assert close(y, g['ffn/y']) and close(x.grad, g['ffn/gx'], 1e-4)
    for k, p in ffn.named_parameters():
        assert close(p.grad, g['ffn/g/' + k], 1e-4), k


@pytest.mark.parametrize('N,C,T,relu,with_res', [(3, 256, 700, False, True), (2, 64, 1292, True, True), (4, 16, 33, False, False)])
def test_groupnorm1_vs_torch(N, C, T, relu, with_res):
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(N + C)
    x = (torch.randn(N, C, T, device=dev) * 2 + 0.7).requires_grad_(True)
    r = torch.randn(N, C, T, device=dev).requires_grad_(True) if with_res else None
    gn = torch.nn.GroupNorm(1, C).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C) * 0.3 + 1)
        gn.bias.copy_(torch.randn(C) * 0.2)
    g = torch.randn(N, C, T, device=dev)
    ref = gn((x + r).double() if with_res else x.double()) if False else None
    xs = x.double() + (r.double() if with_res else 0)
    gd = torch.nn.GroupNorm(1, C).to(dev).double()
    gd.load_state_dict({k: v.double() for k, v in gn.state_dict().items()})
    xr = x.detach().double().requires_grad_(True)
    rr = r.detach().double().requires_grad_(True) if with_res else None
    yr = gd(xr + rr if with_res else xr)
    if relu:
        yr = torch.relu(yr)
    (yr * g.double()).sum().backward()
    y = K.GroupNorm1.apply(x, r, gn.weight, gn.bias, gn.eps, relu)
    (y * g).sum().backward()
    torch.cuda.synchronize()
    tol = lambda a, b, rt: float((a.double() - b).abs().max()) <= rt * float(b.abs().max())  # noqa: E731
    assert tol(y, yr, 3e-6)
    assert tol(x.grad, xr.grad, 2e-5)
    if with_res:
        assert tol(r.grad, rr.grad, 2e-5)
    assert tol(gn.weight.grad, gd.weight.grad, 2e-5) and tol(gn.bias.grad, gd.bias.grad, 2e-5)


def test_softmax_keys_large_with_mask():
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    B, T = 8, 1292
    s = torch.randn(B, T, T, device=dev) * 3
    lens = [1292, 1000, 700, 173, 1292, 64, 900, 500]
    mask = torch.zeros(B, T, dtype=torch.bool, device=dev)
    for i, L in enumerate(lens):
        mask[i, L:] = True
    ga = torch.randn(B, T, T, device=dev)
    sd = s.double().requires_grad_(True)
    sc = (sd / 8.0).masked_fill(mask.unsqueeze(2), -float('inf'))
    ref = torch.softmax(sc, 1).masked_fill(mask.unsqueeze(1), 0.0)
    (ref * ga.double()).sum().backward()
    sx = s.clone().requires_grad_(True)
    att = K.SoftmaxKeys.apply(sx, mask.to(torch.uint8), 1.0 / 8.0)
    (att * ga).sum().backward()
    torch.cuda.synchronize()
    assert float((att.double() - ref).abs().max()) <= 2e-6
    assert float((sx.grad.double() - sd.grad).abs().max()) <= 2e-6 * max(1.0, float(sd.grad.abs().max()))
    a = att[3]
    assert float(a[173:, :].abs().max()) == 0 and float(a[:, 173:].abs().max()) == 0
    assert torch.allclose(a[:, :173].sum(0), torch.ones(173, device=dev), atol=1e-5)


@pytest.mark.parametrize('T,N', [(173, 8), (1292, 4)])
def test_config4_block_vs_float64(T, N):
    """BASELINE config 4 at module level: MultiHeadAttention(256, 4) + PointwiseFeedForward(256) on bucket-style padded
    batches (modules.py:32-79, 108-116) against the SAME modules evaluated in float64 with the torch formulation:
    outputs, the returned attention tensor, the input gradient and every parameter gradient.
    Tolerance: 3e-5 of max on outputs, 2e-4 of max on gradients - or twice the error of the same formulation in plain fp32
    torch ops against float64 where fp32 itself is the limit (T = 1292: the softmax backward over 1292 keys)."""
    import copy
    from pytorch_sound_amd.models.modules import MultiHeadAttention, PointwiseFeedForward, PositionalEncoding
    dev = torch.device('cuda:0')
    torch.manual_seed(T)
    C, H = 256, 4
    pe = PositionalEncoding(C, 2048).to(dev)
    mha = MultiHeadAttention(C, H, 0.0).to(dev)
    ffn = PointwiseFeedForward(C, 0.0).to(dev)
    with torch.no_grad():
        for m in (mha, ffn):
            m.layernorm.weight.copy_(1 + 0.2 * torch.randn(C))
            m.layernorm.bias.copy_(0.1 * torch.randn(C))
    lens = torch.linspace(T, max(T // 3, 8), N).long()             # a length bucket padded to its longest clip
    mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    x0 = (0.1 * torch.randn(N, C, T, device=dev)) * (~mask).unsqueeze(1)   # keeps the logits within +-20: a conditioned softmax
    gy = torch.randn(N, C, T, device=dev)
    gatt = torch.randn(H * N, T, T, device=dev) * 0.1

    def run(mods, dt, att_in_loss):
        pe_, mha_, ffn_ = mods
        x = x0.detach().clone().to(dt).requires_grad_(True)
        h, att = mha_(pe_(x), mask)
        y = ffn_(h)
        loss = (y * gy.to(dt)).sum()
        if att_in_loss:
            loss = loss + (att * gatt.to(dt)).sum()
        loss.backward()
        grads = {('mha.' + k): p.grad for k, p in mha_.named_parameters()}
        grads.update({('ffn.' + k): p.grad for k, p in ffn_.named_parameters()})
        return y.detach(), att.detach(), x.grad, grads

    ref_mods = tuple(copy.deepcopy(m).double() for m in (pe, mha, ffn))
    for att_in_loss in (False, True):
        for m in (mha, ffn) + ref_mods[1:]:
            m.zero_grad()
        y, att, gx, gp = run((pe, mha, ffn), torch.float32, att_in_loss)
        yr, attr, gxr, gpr = run(ref_mods, torch.float64, att_in_loss)
        # yardstick: the SAME formulation in plain fp32 torch ops (library GEMMs, torch softmax / group_norm) against float64 -
        # what fp32 arithmetic itself costs at this size (the softmax backward cancels: att * (g - sum att g))
        import pytorch_sound_amd.models.modules as M
        keep_hip_ok = M._hip_ok
        M._hip_ok = lambda t: False                   # the torch formulation evaluated on the GPU: this test's fp32 yardstick
        try:
            for m in (mha, ffn):
                m.zero_grad()
            y32, att32, gx32, gp32 = run((pe, mha, ffn), torch.float32, att_in_loss)
            y32, att32, gx32, gp32 = y32.clone(), att32.clone(), gx32.clone(), {k: v.clone() for k, v in gp32.items()}
        finally:
            M._hip_ok = keep_hip_ok
        err = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())   # noqa: E731
        ok = lambda a, a32, b, rt: err(a, b) <= max(rt, 2.0 * err(a32, b))        # noqa: E731
        print('T=%d att_in_loss=%s: y %.1e (torch fp32 %.1e)  gx %.1e (%.1e)  worst param grad %.1e (%.1e)' % (
            T, att_in_loss, err(y, yr), err(y32, yr), err(gx, gxr), err(gx32, gxr),
            max(err(gp[k], gpr[k]) for k in gp), max(err(gp32[k], gpr[k]) for k in gp)))
        assert ok(y, y32, yr, 3e-5), err(y, yr)
        assert float((att.double() - attr).abs().max()) <= 1e-5      # probabilities in [0, 1]
        assert ok(gx, gx32, gxr, 2e-4), (err(gx, gxr), err(gx32, gxr))
        for k in gp:
            assert ok(gp[k], gp32[k], gpr[k], 2e-4), (k, err(gp[k], gpr[k]), err(gp32[k], gpr[k]))
        # padded keys get no weight, padded queries are zeroed (modules.py:69-76)
        a = att.view(H, N, T, T)[:, -1]
        L = int(lens[-1])
        assert float(a[:, L:, :].abs().max()) == 0 and float(a[:, :, L:].abs().max()) == 0


@pytest.mark.parametrize('N,Cin,Cout,T,relu', [(3, 80, 256, 173, False), (2, 256, 768, 1292, False), (4, 256, 1024, 431, True),
                                               (2, 1024, 256, 100, False), (1, 16, 48, 10, True), (5, 33, 65, 131, True),
                                               (7, 20, 24, 5, False), (2, 8, 8, 3, False), (3, 64, 128, 129, False), (40, 16, 16, 7, True)])
@pytest.mark.parametrize('bf16', [False, True])
def test_linear1x1_exact(N, Cin, Cout, T, relu, bf16):
    """psnd_linear1x1_fwd / _bwd (the 1x1 Conv1d projections, modules.py:21-22, 93-95), ragged sizes and unaligned rows included,
    against float64: exact-fp32 form to 2e-6 of max; bf16-operand form against float64 arithmetic on the bf16-ROUNDED operands
    to 1e-5 of max (fp32 accumulation), and within 2e-2 of the unrounded result."""
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(N * 1000 + T)
    x = torch.randn(N, Cin, T, device=dev, requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, device=dev) / Cin ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, device=dev, requires_grad=True)
    g = torch.randn(N, Cout, T, device=dev)
    y = K.Linear1x1.apply(x, w, b, relu, bf16)
    (y * g).sum().backward()
    rnd = (lambda t: t.to(torch.bfloat16).double()) if bf16 else (lambda t: t.double())

    def ref(round_fn):
        xd, wd, bd = x.detach().double(), w.detach().double().squeeze(-1), b.detach().double()
        yd = torch.einsum('oc,nct->not', round_fn(wd), round_fn(xd)) + bd.view(1, -1, 1)
        gd = g.double()
        if relu:
            gd = gd * (yd > 0)
            yd = yd.clamp_min(0)
        gx = torch.einsum('oc,not->nct', round_fn(wd), round_fn(gd))
        gw = torch.einsum('not,nct->oc', round_fn(gd), round_fn(xd))
        return yd, gx, gw, gd.sum((0, 2))

    yd, gx, gw, gb = ref(rnd)
    tol = 1e-5 if bf16 else 2e-6
    close = lambda a, d, rt: float((a.double() - d).abs().max()) <= rt * float(d.abs().max())   # noqa: E731
    if relu:            # the kernel masks by ITS y > 0: compare away from the kink (|y| tiny flips with the last bit)
        keep = (yd.abs() > 1e-4) | (yd == 0)
        assert float(((y.double() - yd) * keep).abs().max()) <= tol * float(yd.abs().max())
    else:
        assert close(y, yd, tol)
        assert close(x.grad, gx, tol) and close(w.grad.squeeze(-1), gw, tol * 4)
    assert close(b.grad, gb, 1e-4 if relu else tol * 4)
    if bf16:
        y32, gx32, gw32, _ = ref(lambda t: t.double())
        assert close(y, y32, 2e-2) and (relu or (close(x.grad, gx32, 2e-2) and close(w.grad.squeeze(-1), gw32, 2e-2)))


def test_block_under_autocast_uses_bf16_products():
    """torch.autocast(bfloat16) around the modules: the 1x1 projections AND the attention products take bf16 operands (fp32 accumulation,
    scores, statistics and activations) - outputs and gradients within bf16 tolerance (3e-2 relative Frobenius) of the fp32 run"""
    from pytorch_sound_amd.models.modules import MultiHeadAttention, PointwiseFeedForward
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    mha, ffn = MultiHeadAttention(256, 4, 0.0).to(dev), PointwiseFeedForward(256, 0.0).to(dev)
    x0 = 0.3 * torch.randn(4, 256, 300, device=dev)
    mask = (torch.arange(300)[None, :] >= torch.tensor([300, 250, 200, 120])[:, None]).to(dev)
    g = torch.randn(4, 256, 300, device=dev)

    def run(ac):
        for m in (mha, ffn):
            m.zero_grad()
        x = x0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=ac):
            h, att = mha(x, mask)
            y = ffn(h)
        assert y.dtype == torch.float32
        (y.float() * g).sum().backward()
        return y.detach().float(), x.grad.clone(), torch.cat([p.grad.flatten() for m in (mha, ffn) for p in m.parameters()])

    a, b = run(True), run(False)
    for u, v in zip(a, b):           # relative Frobenius: single elements behind a sharp softmax move more than the average
        assert float((u - v).norm() / v.norm()) <= 3e-2, float((u - v).norm() / v.norm())
    assert float((a[0] - b[0]).abs().max()) > 0          # the bf16 path really ran


def _attention_float64(kvq, mask, H, gout, gatt, round_operands):
    """scale_dot_att over folded heads (modules.py:38-48, 61-79) in float64 torch ops with autograd; round_operands: K, Q, V and the
    probabilities pass through bf16 before the two products - what the bf16 kernels compute, up to fp32 accumulation order."""
    N, C3, T = kvq.shape
    C = C3 // 3
    d = C // H
    x = kvq.double().requires_grad_(True)

    class _R(torch.autograd.Function):          # straight-through rounding: gradients as if exact
        @staticmethod
        def forward(ctx, t):
            return t.float().bfloat16().double()

        @staticmethod
        def backward(ctx, g):
            return g

    r = _R.apply if round_operands else (lambda t: t)
    k, v, q = (t.view(N, H, d, T).transpose(0, 1).reshape(H * N, d, T) for t in x.chunk(3, 1))
    s = torch.einsum('bdk,bdq->bkq', r(k), r(q)) / (d ** 0.5)
    if mask is not None:
        m = mask.bool().repeat(H, 1)
        s = s.masked_fill(m[:, :, None], float('-inf'))
    att = torch.softmax(s, 1)
    if mask is not None:
        att = att.masked_fill(m[:, None, :], 0.0)
    out = torch.einsum('bdk,bkq->bdq', r(v), r(att))
    out = out.view(H, N, d, T).transpose(0, 1).reshape(N, C, T)
    loss = (out * gout.double()).sum()
    if gatt is not None:
        loss = loss + (att * gatt.double()).sum()
    loss.backward()
    return out.detach(), att.detach(), x.grad


@pytest.mark.parametrize('N,H,C,T,masked,with_gatt', [(2, 4, 256, 173, True, False), (3, 4, 64, 50, True, True), (2, 4, 256, 1292, True, False),
                                                     (2, 2, 96, 77, False, True), (1, 4, 128, 33, False, False), (2, 4, 256, 431, True, True)])
def test_attention_bf16_operands(N, H, C, T, masked, with_gatt):
    """psnd_mha_fwd / _bwd with bf16 = 1 (head dimensions 64, 16, 48, 32; ragged T; padding masks; with and without a gradient into
    the returned attention tensor):
      * forward against float64 with the SAME operand rounding: 2e-3 of the largest output (fp32 accumulation + the probabilities
        that sit on a bf16 rounding boundary in fp32 but not in float64); the returned probabilities are not rounded: 2e-6 absolute
        beyond what rounding K and Q moves them;
      * backward against exact float64: relative Frobenius 2e-2 (bf16 operands: 2^-9 per element, random signs), the bound the
        projections under autocast are held to;
      * padded keys / queries get exactly zero gradient, as in the fp32 kernels."""
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(N * 1000 + T)
    kvq = torch.randn(N, 3 * C, T, device=dev)
    mask = None
    if masked:
        lens = torch.linspace(T, max(T // 3, 4), N).long()
        mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    gout = torch.randn(N, C, T, device=dev)
    gatt = 0.3 * torch.randn(H * N, T, T, device=dev) if with_gatt else None
    x = kvq.clone().requires_grad_(True)
    out, att = K.AttentionKVQ.apply(x, None if mask is None else mask.to(torch.uint8), H, True, True)
    loss = (out * gout).sum()
    if gatt is not None:
        loss = loss + (att * gatt).sum()
    loss.backward()
    o_r, a_r, _ = _attention_float64(kvq, mask, H, gout, gatt, True)
    o_x, a_x, g_x = _attention_float64(kvq, mask, H, gout, gatt, False)
    assert float((out.double() - o_r).abs().max()) <= 2e-3 * float(o_r.abs().max())
    assert float((att.double() - a_r).abs().max()) <= 2e-6 + 1e-4 * float(a_r.max())
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())           # noqa: E731
    assert rel(out, o_x) <= 2e-2, rel(out, o_x)
    assert rel(x.grad, g_x) <= 2e-2, rel(x.grad, g_x)
    # the fp32 kernels on the same input: the bf16 path is a different computation, and close to it
    x2 = kvq.clone().requires_grad_(True)
    out2, att2 = K.AttentionKVQ.apply(x2, None if mask is None else mask.to(torch.uint8), H, True, False)
    assert float((out2 - out).abs().max()) > 0 and rel(out2, o_x) <= 1e-5
    if masked:
        g = x.grad.view(N, 3, C, T)
        L = int(lens[-1])
        assert float(g[-1, :, :, L:].abs().max()) == 0


def test_scale_dot_att_static_method_on_hip_tensors():
    """MultiHeadAttention.scale_dot_att is a public static method in the reference (modules.py:61-79): called directly on HIP tensors it
    runs on psnd_mha_* too (one head per batch entry) - no library bmm / softmax (dispatch-mode check) - and equals the torch
    formulation in float64, masks included; an uncovered head dimension raises."""
    from pytorch_sound_amd.models.modules import MultiHeadAttention
    from pytorch_sound_amd._lib import PsndError
    from test_gpu_no_library_paths import forbid_library_ops
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    B, d, T = 6, 48, 77
    k, v, q = (torch.randn(B, d, T, device=dev) for _ in range(3))
    lens = torch.randint(T // 2, T + 1, (B,))
    mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    for m in (None, mask):
        with forbid_library_ops():
            x, att = MultiHeadAttention.scale_dot_att(k, v, q, m)
        xr, attr = MultiHeadAttention.scale_dot_att(k.double().cpu(), v.double().cpu(), q.double().cpu(), None if m is None else m.cpu())
        assert float((x.double().cpu() - xr).abs().max()) <= 2e-5 * float(xr.abs().max())
        assert float((att.double().cpu() - attr).abs().max()) <= 1e-5
    # round 5: head dimensions up to 128 run on the HDP = 128 instances; beyond that the call raises
    k9, v9, q9 = (torch.randn(2, 96, 8, device=dev) for _ in range(3))
    x9, a9 = MultiHeadAttention.scale_dot_att(k9, v9, q9, None)
    x9r, a9r = MultiHeadAttention.scale_dot_att(k9.double().cpu(), v9.double().cpu(), q9.double().cpu(), None)
    assert float((x9.double().cpu() - x9r).abs().max()) <= 2e-5 * float(x9r.abs().max()) and float((a9.double().cpu() - a9r).abs().max()) <= 1e-5
    with pytest.raises(PsndError):
        MultiHeadAttention.scale_dot_att(torch.randn(2, 160, 8, device=dev), torch.randn(2, 160, 8, device=dev), torch.randn(2, 160, 8, device=dev), None)


def test_positional_encoding_kernel_matches_the_torch_formulation():
    """PositionalEncoding on a HIP tensor (psnd_posenc, modules.py:143-145) against x * sqrt(C) + pe[..., :T] in torch, bit for bit, and
    its gradient g * sqrt(C)"""
    from pytorch_sound_amd.models import modules as M
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    # (1000, 1292 frames: the four-frames-per-thread kernel; 66 x 512 x 1028: more 16-byte pieces than its capped grid covers in one trip)
    for N, C, T in ((3, 16, 10), (2, 256, 173), (1, 30, 1000), (32, 256, 1292), (66, 512, 1028)):
        pe = M.PositionalEncoding(C, 1200 if T <= 1200 else 2048).to(dev)
        x = torch.randn(N, C, T, device=dev, requires_grad=True)
        y = pe(x)
        ref = x.detach() * (C ** 0.5) + pe.pe[..., :T]
        assert torch.equal(y.detach(), ref)
        g = torch.randn_like(ref)
        y.backward(g)
        assert torch.equal(x.grad, g * (C ** 0.5))


@pytest.mark.parametrize('N,H,C,T,masked', [(2, 4, 256, 173, True), (3, 4, 64, 50, True), (2, 4, 256, 1292, True), (2, 2, 96, 77, False),
                                            (1, 4, 128, 33, False)])
def test_attention_bf16_single_pass_forward(N, H, C, T, masked):
    """psnd_mha_fwd, bf16 operands, WITHOUT the returned attention tensor: the one-pass form (running column maximum, rescaled
    accumulator - attn_fwd_bf16_kernel<HDP, false>) against float64 with rounded operands (4e-3 of the largest output), against the two-pass instance (want_att = True), and its gradients (the backward recomputes the
    probabilities from the statistics this pass writes) against exact float64; padded queries are exactly zero."""
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(N * 100 + T)
    kvq = torch.randn(N, 3 * C, T, device=dev)
    mask = None
    if masked:
        lens = torch.linspace(T, max(T // 3, 4), N).long()
        mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    m8 = None if mask is None else mask.to(torch.uint8)
    gout = torch.randn(N, C, T, device=dev)
    x = kvq.clone().requires_grad_(True)
    out, att = K.AttentionKVQ.apply(x, m8, H, False, True)
    assert att is None or att.numel() == 0
    (out * gout).sum().backward()
    o_r, _, _ = _attention_float64(kvq, mask, H, gout, None, True)
    o_x, _, g_x = _attention_float64(kvq, mask, H, gout, None, False)
    # 4e-3 of the largest output: the one-pass form rounds exp(s - running maximum) to bf16 (2^-9 each, random signs over the keys), the
    # emulation rounds the final probabilities - the same size of error, not the same values
    assert float((out.double() - o_r).abs().max()) <= 4e-3 * float(o_r.abs().max())
    x2 = kvq.clone().requires_grad_(True)
    out2, _ = K.AttentionKVQ.apply(x2, m8, H, True, True)
    assert float((out - out2).abs().max()) <= 4e-3 * float(o_r.abs().max())
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())           # noqa: E731
    assert rel(out, o_x) <= 2e-2 and rel(x.grad, g_x) <= 2e-2, (rel(out, o_x), rel(x.grad, g_x))
    if masked:
        L = int(lens[-1])
        assert float(out[-1, :, L:].abs().max()) == 0
        assert float(x.grad.view(N, 3, C, T)[-1, :, :, L:].abs().max()) == 0


@pytest.mark.parametrize('N,H,C,T,masked', [(2, 4, 256, 173, True), (3, 4, 64, 50, True), (2, 2, 96, 77, False), (1, 4, 256, 700, True)])
def test_attention_fp32_single_pass_forward(N, H, C, T, masked):
    """psnd_mha_fwd with exact fp32 products and no returned attention tensor (attn_fwd_kernel<HDP, false>, the one-pass form) against
    float64 and against the two-pass instance; gradients against float64."""
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(N * 10 + T)
    kvq = torch.randn(N, 3 * C, T, device=dev)
    mask = None
    if masked:
        lens = torch.linspace(T, max(T // 3, 4), N).long()
        mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    m8 = None if mask is None else mask.to(torch.uint8)
    gout = torch.randn(N, C, T, device=dev)
    x = kvq.clone().requires_grad_(True)
    out, _ = K.AttentionKVQ.apply(x, m8, H, False, False)
    (out * gout).sum().backward()
    o_x, _, g_x = _attention_float64(kvq, mask, H, gout, None, False)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())           # noqa: E731
    assert rel(out, o_x) <= 1e-5 and rel(x.grad, g_x) <= 1e-4, (rel(out, o_x), rel(x.grad, g_x))
    out2, _ = K.AttentionKVQ.apply(kvq.clone(), m8, H, True, False)
    assert float((out - out2).abs().max()) <= 1e-5 * float(o_x.abs().max())
    if masked and int(lens[-1]) < T:
        assert float(out[-1, :, int(lens[-1]):].abs().max()) == 0


@pytest.mark.parametrize('N,H,C,T,masked', [(2, 4, 256, 173, True), (3, 4, 64, 50, True), (2, 4, 256, 1292, True), (2, 2, 96, 77, False),
                                             (1, 2, 256, 300, True), (1, 4, 128, 31, False)])
def test_attention_kvq_stored_as_bf16(N, H, C, T, masked):
    """psnd_mha_fwd / _bwd with bf16 = 2 (kvq and gout read, out and gkvq written as bf16 tensors - what the projections' epilogues store
    and take under autocast) against bf16 = 1 on the SAME values held in fp32: the kernels round their operands to bf16 when they load
    them, so out is the fp32 result stored as bf16 (bit-equal after rounding).  Head dimensions 16-128, odd T (2-byte aligned rows), a
    short single tile."""
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda')
    torch.manual_seed(5)
    kvq_h = torch.randn(N, 3 * C, T, device=dev).to(torch.bfloat16)
    kvq_f = kvq_h.float()
    mask = None
    if masked:
        mask = torch.zeros(N, T, dtype=torch.uint8, device=dev)
        for n in range(N):
            mask[n, T - 1 - 3 * n - (T // 5):] = 1
    w = torch.randn(N, C, T, device=dev)

    def run(kvq):
        kvq = kvq.clone().requires_grad_(True)
        out, _ = K.AttentionKVQ.apply(kvq, mask, H, False, True)
        (out * w).sum().backward()
        return out.detach(), kvq.grad

    oh, gh = run(kvq_h)
    of, gf = run(kvq_f)
    assert oh.dtype == torch.bfloat16 and gh.dtype == torch.bfloat16 and of.dtype == gf.dtype == torch.float32
    assert torch.equal(oh, of.to(torch.bfloat16))          # the same accumulators, stored as bf16
    # the gradient: out and gout reach the backward as bf16 (the products round them anyway; only the softmax's column term delta =
    # sum out * gout sees the rounding) and gkvq is stored as bf16 - one bf16 rounding (2^-9 relative) on each
    assert float((gh.float() - gf).norm()) <= 6e-3 * float(gf.norm())
    assert float((gh.float() - gf).abs().max()) <= 2e-2 * float(gf.abs().max())


def test_mha_module_kvq_bf16_under_autocast(monkeypatch):
    """MultiHeadAttention under autocast with return_att = False: the projection hands kvq over as a bf16 tensor (kvq_bf16) - output and
    every gradient against the module with kvq kept in fp32: the weight gradients of the projection see the bf16-rounded gkvq (4e-3)"""
    from pytorch_sound_amd.models import modules as M
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda')
    torch.manual_seed(6)
    mha = M.MultiHeadAttention(128, 4, 0.0).to(dev)
    mha.return_att = False
    x0, w = torch.randn(3, 128, 211, device=dev), torch.randn(3, 128, 211, device=dev)
    pad = torch.zeros(3, 211, dtype=torch.bool, device=dev)
    pad[1, 180:] = True
    seen = []
    orig = K.AttentionKVQ.forward

    def fwd(ctx, kvq, *a):
        seen.append(kvq.dtype)
        return orig(ctx, kvq, *a)
    monkeypatch.setattr(K.AttentionKVQ, 'forward', staticmethod(fwd))

    def run(h):
        mha.kvq_bf16 = h
        x = x0.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y, att = mha(x, pad)
        assert att is None
        (y.float() * w).sum().backward()
        g = [y.detach().float().clone(), x.grad.clone()] + [p.grad.clone() for p in mha.parameters()]
        for p in mha.parameters():
            p.grad = None
        return g

    a, b = run(True), run(False)
    assert seen == [torch.bfloat16, torch.float32]
    assert float((a[0] - b[0]).abs().max()) <= 2e-6 * float(b[0].abs().max())
    for u, v in zip(a[1:], b[1:]):
        assert float((u - v).norm()) <= 4e-3 * float(v.norm())


def test_groupnorm1_is_reproducible_bit_for_bit():
    """psnd_groupnorm1_fwd / _bwd keep one pair of sums per row and add them up in a fixed order (no atomics since round 6): two runs on the
    same data give the same bits - output, statistics, input gradient and the parameter gradients (which add the rows of all samples up)"""
    from pytorch_sound_amd import kernels as K
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    for N, C, T, relu in ((32, 256, 1292, True), (3, 48, 173, False), (5, 300, 64, True)):
        x, res, w = (torch.randn(N, C, T, device=dev) for _ in range(3))
        gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
        runs = []
        for _ in range(3):
            xx, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
            g, b = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            y = K.GroupNorm1.apply(xx, rr, g, b, 1e-5, relu)
            (y * w).sum().backward()
            runs.append([y.detach().clone(), xx.grad.clone(), rr.grad.clone(), g.grad.clone(), b.grad.clone()])
        for other in runs[1:]:
            for u, v in zip(runs[0], other):
                assert torch.equal(u, v)


@pytest.mark.parametrize('autocast', [False, True])
def test_transformer_block_gradients_are_reproducible(autocast):
    """1x1 projection -> PositionalEncoding -> MultiHeadAttention -> PointwiseFeedForward -> 1x1 projection (the config-4 block), forward and
    backward twice on the same data: every kernel of it adds up in a fixed order (slab sums, row sums, the GroupNorm's row pairs since round
    6, no atomics) - loss and all gradients are the same bits, with the parameter side on its own stream or not"""
    from pytorch_sound_amd.models import modules as M
    dev = torch.device('cuda:0')
    torch.manual_seed(11)
    C, H, T, N = 128, 4, 333, 5

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inp, self.pe = torch.nn.Conv1d(80, C, 1), M.PositionalEncoding(C, 512)
            self.mha, self.ffn, self.out = M.MultiHeadAttention(C, H, 0.0), M.PointwiseFeedForward(C, 0.0), torch.nn.Conv1d(C, 80, 1)
            self.mha.return_att = False

        def forward(self, x, pad):
            y, _ = self.mha(self.pe(M._conv1x1(self.inp, x)), pad)
            return M._conv1x1(self.out, self.ffn(y))

    net = Net().to(dev)
    x, w = torch.randn(N, 80, T, device=dev), torch.randn(N, 80, T, device=dev)
    pad = torch.zeros(N, T, dtype=torch.bool, device=dev)
    pad[2, 300:] = True
    runs = []
    for _ in range(3):
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            y = net(x, pad)
        loss = (y.float() * w).sum()
        loss.backward()
        torch.cuda.synchronize()
        runs.append([loss.detach().clone()] + [p.grad.clone() for p in net.parameters()])
        for p in net.parameters():
            p.grad = None
    for other in runs[1:]:
        for u, v in zip(runs[0], other):
            assert torch.equal(u, v)
