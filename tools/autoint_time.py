# times the AutoInt layer kernels in isolation (HIP events): python tools/autoint_time.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeptables_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda', 0)
B, F, D, H = 8192, 26, 32, 4
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, F, D, generator=g) * 0.5).to(dev)
Ws = [(torch.randn(D, D, generator=g) * 0.2).to(dev) for _ in range(4)]
bs = [(torch.randn(D, generator=g) * 0.1).to(dev) for _ in range(4)]
a = torch.empty_like(x); go = torch.randn_like(x)
dX = torch.empty_like(x); dY = torch.empty(B * F, 4 * D, device=dev)
W = [ptr(t) for t in Ws]; Bv = [ptr(t) for t in bs]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd = lambda: check(lib().dt_autoint_fwd(ptr(x), *W, *Bv, B, F, D, H, 0.0, 0, ptr(a), None, stream_ptr()), 'f')
def bwd(dy, dx):
    return lambda: check(lib().dt_autoint_bwd(ptr(x), *W, *Bv, ptr(a), ptr(go), B, F, D, H, 0.0, 0, ptr(dy) if dy is not None else None,
                                             ptr(dx) if dx is not None else None, stream_ptr()), 'b')
print(f'fwd                 {timeit(fwd):8.1f} us')
print(f'bwd (none)          {timeit(bwd(None, None)):8.1f} us')
print(f'bwd dX              {timeit(bwd(None, dX)):8.1f} us')
print(f'bwd dX + dY         {timeit(bwd(dY, dX)):8.1f} us')
