import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pytorch_sound_amd import cl
from pytorch_sound_amd.models.vocoders.hifi_gan import WNConv1d
dev = torch.device('cuda:0')
for (Cin, Cout, k, dil, L, N) in [(128, 128, 11, 5, 64, 2), (128, 128, 11, 5, 200, 1), (64, 64, 3, 1, 50, 2)]:
    torch.manual_seed(Cin + k)
    pad = (k * dil - dil) // 2
    conv = WNConv1d(Cin, Cout, k, dil, pad, init_std=0.05).to(dev)
    bf = lambda t: t.to(torch.bfloat16).double()
    x = torch.randn(N, Cin, L, device=dev); gy = torch.randn(N, Cout, L, device=dev)
    shape = cl.CLShape(N, L, pad + 2)
    xc = x.clone().requires_grad_(True)
    yb, yab = cl.fused_conv(cl.ToCL.apply(xc, shape, 0), conv, shape, None, False, True, 0.1)
    out = cl.FromCL.apply(yab, Cout, L, shape)
    (out * gy).sum().backward()
    w32 = (conv.weight_v.detach() * (conv.weight_g.detach() / conv.weight_v.detach().flatten(1).norm(dim=1).view(-1, 1, 1)))
    wq, xq = bf(w32), bf(x)
    g = bf(gy.to(torch.bfloat16).float() * torch.where(out.detach() > 0, 1.0, 0.1).float())
    gx = torch.nn.grad.conv1d_input(xq.shape, wq, g, 1, pad, dil)
    d = (xc.grad.double() - gx).abs()
    lim = 1.01 * 2.0 ** -8 * gx.abs() + 1e-4 * float(gx.abs().max())
    bad = (d > lim).nonzero()
    print((Cin, Cout, k, dil, L, N), 'violations', bad.shape[0], 'of', d.numel(), 'max d', float(d.max()), 'max gx', float(gx.abs().max()))
    for b in bad[:10].tolist():
        n, c, t = b
        print('   at', b, 'got', float(xc.grad[n, c, t]), 'want', float(gx[n, c, t]), 'diff/ulp', float(d[n, c, t] / (2.0 ** -8 * gx[n, c, t].abs())))
