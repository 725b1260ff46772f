import torch, torch.nn.functional as F, sys
sys.path.insert(0,'/root/repo')
from pytorch_sound_amd import kernels as K
for (N,Cin,Cout,k,dil,pad,T,slope) in [(9,1024,16,8,1,3,24,0.1),(2,1024,16,8,1,3,24,0.1),(9,1024,16,8,1,3,24,1.0),(9,256,16,8,1,3,24,0.1)]:
    torch.manual_seed(0)
    x = torch.randn(N, Cin, T); w = torch.randn(Cout, Cin, k) * 0.2; b = torch.randn(Cout)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    yd = F.conv1d(F.leaky_relu(xd, slope) if slope != 1.0 else xd, wd, bd, 1, pad, dil)
    gy = torch.randn_like(yd); (yd * gy).sum().backward()
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = K.conv1d_f32(xg, wg, bg, pad, dil, slope); (y * gy.float().cuda()).sum().backward()
    r = lambda a, b_: float((a.detach().double().cpu() - b_).abs().max() / b_.abs().max())
    print((N,Cin,Cout,k,slope), 'y %.2e gx %.2e gw %.2e gb %.2e' % (r(y, yd.detach()), r(xg.grad, xd.grad), r(wg.grad, wd.grad), r(bg.grad, bd.grad)))
