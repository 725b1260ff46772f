import torch
dev = torch.device('cuda:0')
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for F in (173, 176, 1292):
    N = 1024 if F < 1000 else 128
    a = torch.empty(N, 513, F, device=dev)
    for ch in (16, 32, 64, 128, F):
        src = torch.randn(N, 513, ch, device=dev)
        nt = F // ch
        def w():
            for j in range(nt):
                a[:, :, j*ch:(j+1)*ch] = src
        t = timeit(w)
        print('F=%d chunk=%d frames (%d B): %.0f GB/s' % (F, ch, ch*4, nt*src.numel()*4/t/1e9), flush=True)
