import torch, time
dev = torch.device('cuda:0')
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for mb in (363, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    t = timeit(lambda: x.fill_(1.0)); print('fill %d MB: %.1f us  %.0f GB/s write' % (mb, t*1e6, n*4/t/1e9))
    t = timeit(lambda: x.zero_()); print('zero %d MB: %.1f us  %.0f GB/s write' % (mb, t*1e6, n*4/t/1e9))
    t = timeit(lambda: y.copy_(x)); print('copy %d MB: %.1f us  %.0f GB/s r+w' % (mb, t*1e6, 2*n*4/t/1e9))
    t = timeit(lambda: x.sum()); print('sum  %d MB: %.1f us  %.0f GB/s read' % (mb, t*1e6, n*4/t/1e9))
# strided-row write: emulate (K=513 rows) x 64-byte chunks per tile
a = torch.empty(1024, 513, 173, device=dev)
src = torch.randn(1024, 513, 16, device=dev)
def tilewrite():
    for j in range(0, 160, 16):
        a[:, :, j:j+16] = src
t = timeit(tilewrite, iters=5); print('tile-strided write 10x(1024x513x16): %.1f us -> %.0f GB/s' % (t*1e6, 10*src.numel()*4/t/1e9))
