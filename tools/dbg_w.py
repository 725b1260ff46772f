"""debug harness of psnd_stft_w.hip (build the library with -DPSND_W_DEBUG: tools/build_variant.sh wdbg -DPSND_W_DEBUG): dumps the
registers of wave 3 (frame 3) of the first tile after every stage and compares them with a numpy restatement of the algorithm."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
dev = torch.device('cuda:0')
dbg = torch.zeros(5 * 64 * 32 * 2, device=dev)
os.environ['PSND_W_DBG_PTR'] = str(dbg.data_ptr())
from pytorch_sound_amd import kernels as K
n, h, T = 4096, 1024, 9000
rs = np.random.RandomState(1)
x = rs.randn(1, T).astype(np.float32)
w = (0.5 - 0.5*np.cos(2*np.pi*np.arange(n)/n)).astype(np.float32)
plan = K.stft_plan(n, w).to(dev)
out = K.stft_forward(torch.from_numpy(x).to(dev), n, h, plan, 0, 0.0)['mag'].cpu().numpy()[0]
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(5, 64, 32, 2)
d = d[..., 0] + 1j * d[..., 1]
xp = np.pad(x[0].astype(np.float64), (n//2, n//2), mode='reflect')
fr = xp[3*h:3*h+n]
z = 0.5*w[0::2]*fr[0::2] + 1j*0.5*w[1::2]*fr[1::2]
def W(N, e): return np.exp(-2j*np.pi*np.asarray(e)/N)
def br(v, b=5): return int(format(v, '0%db' % b)[::-1], 2)
lanes = [(l & 31, l >> 5) for l in range(64)]
s0 = np.zeros((64, 32), complex)
for l, (lam, g) in enumerate(lanes):
    cL = W(2048, lam) * (-1j)**g
    for a in range(16):
        nn = lam + 32*(a + 16*g)
        lo, hi = z[nn], z[nn+1024]
        s0[l, a] = lo + hi; s0[l, 16+a] = (lo - hi)*W(64, a)*cL
s1 = np.zeros((64, 32), complex)
for l, (lam, hh) in enumerate(lanes):
    if hh == 0: s1[l, :16] = s0[lam, :16]; s1[l, 16:] = s0[lam+32, :16]
    else: s1[l, :16] = s0[lam, 16:]; s1[l, 16:] = s0[lam+32, 16:]
Y = np.fft.fft(s1, axis=1)
for l in range(64): Y[l] *= W(1024, (l & 31)*np.arange(32))
s2 = np.zeros((64, 32), complex)
for l in range(64):
    for s in range(32): s2[l, s] = Y[l, br(s)]
s3 = np.zeros((64, 32), complex)
for l, (m, hh) in enumerate(lanes):
    s3[l] = [Y[l2 + 32*hh][m] for l2 in range(32)]
Z2 = np.fft.fft(s3, axis=1)
s4 = np.zeros((64, 32), complex)
for l in range(64):
    for s in range(32): s4[l, s] = Z2[l, br(s)]
for st, ref in enumerate([s0, s1, s2, s3, s4]):
    e = np.abs(d[st] - ref)
    print('stage', st, 'max err', e.max(), 'ref max', np.abs(ref).max(), 'worst lane/slot', np.unravel_index(e.argmax(), e.shape))
    if e.max() > 1e-3:
        bad = np.argwhere(e > 1e-3)
        print('  bad count', len(bad), 'first', bad[:8].tolist())
        prev = [s0, s0, s1, s2, s3][st]
        for l, i in [(0, 0), (0, 16), (1, 3), (32, 0), (32, 16), (40, 20)]:
            v = d[st][l, i]
            hit = np.argwhere(np.abs(prev - v) < 1e-5)
            hit2 = np.argwhere(np.abs(ref - v) < 1e-5)
            print('  got[%d,%d] = %s; equals previous-stage entries %s; equals expected-stage entries %s' % (l, i, v, hit[:4].tolist(), hit2[:4].tolist()))
        break
