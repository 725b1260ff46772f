"""debug: one residual pair through psnd_conv1d_cl_pair against two psnd_conv1d_cl launches; where do they differ?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_sound_amd import cl
from pytorch_sound_amd.models.vocoders.hifi_gan import ResBlock1
dev = torch.device('cuda:0')
relf = lambda a, b: float((a - b).norm() / b.norm())
for C, k, d, N, T in [(128, 7, 1, 2, 300), (128, 7, 3, 2, 300), (128, 7, 5, 2, 300), (256, 11, 5, 2, 131), (64, 11, 5, 2, 333), (128, 3, 5, 2, 300)]:
    torch.manual_seed(1)
    blk = ResBlock1(None, C, k, (d,)).to(dev)
    with torch.no_grad():
        for c in list(blk.convs1) + list(blk.convs2):
            c.weight_g.copy_(0.7 + 0.6 * torch.rand_like(c.weight_g))
    x = torch.randn(N, C, T, device=dev)
    gy = torch.randn(N, C, T, device=dev)
    shape = cl.CLShape(N, T, 25)
    outs, grads, mids = {}, {}, {}
    for pair in (1, 0, 2):
        os.environ['PSND_CL_PAIR'] = str(min(pair, 1))
        blk.zero_grad()
        xc = x.clone().requires_grad_(True)
        xr = cl.ToCL.apply(xc, shape, 0)
        y, ya = cl.resblock1_cl(blk, xr, cl.MeanActCL.apply(0.1, xr), shape, want_raw=True)
        out = cl.FromCL.apply(y, C, T, shape)
        node = y.grad_fn
        mids[pair] = [t.detach().float().clone() for t in node.saved_tensors if t.dtype == torch.bfloat16 and t.dim() == 3]
        (out * gy).sum().backward()
        outs[pair] = out.detach().clone()
        grads[pair] = {n_: p.grad.clone() for n_, p in blk.named_parameters()}
        grads[pair]['x'] = xc.grad.clone()
    print('C', C, 'k', k, 'd', d, 'T', T, 'out relf', relf(outs[1], outs[0]), ' repeat', relf(outs[2], outs[1]))
    print('   grads pair vs conv:', {n_: round(relf(grads[1][n_], grads[0][n_]), 5) for n_ in grads[0]})
    print('   grads pair vs pair:', {n_: round(relf(grads[2][n_], grads[1][n_]), 5) for n_ in grads[0]})
    for i, (a, b) in enumerate(zip(mids[1], mids[0])):
        if a.shape == b.shape:
            dd = (a - b).abs()
            sgn = ((a > 0) != (b > 0)).sum().item()
            print('   saved[%d]' % i, tuple(a.shape), 'max diff', float(dd.max()), 'sign flips', sgn, 'of', a.numel(), 'rows with diff > 0.05:', (dd.amax(dim=2) > 0.05).sum().item())
