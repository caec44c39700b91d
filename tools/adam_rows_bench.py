"""Micro-benchmark of dt_adam_rows_step (csrc/optim.hip) on Criteo-shaped lookups: python tools/adam_rows_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeptables_amd._lib import lib, ptr, stream_ptr, check

dev = torch.device('cuda')
B, F, D, V = 8192, 26, 16, 1_000_000
table = torch.zeros(F * V, D, device=dev)
m, v = torch.zeros_like(table), torch.zeros_like(table)
state = torch.zeros(68, dtype=torch.int32, device=dev)
check(lib().dt_adam_state_init(ptr(state), 1e-3, 0.9, 0.999, 0, stream_ptr()), 'init')
n = B * F
SLOT_MULT = int(os.environ.get('SLOT_MULT', '1'))
slots = torch.zeros(lib().dt_adam_rows_slots(n) * SLOT_MULT, dtype=torch.int64, device=dev)
mark = torch.empty(n, dtype=torch.int32, device=dev)
vals = torch.randn(n, D, device=dev)
off = (torch.arange(F, device=dev) * V)[None, :]


def run(rows, fields, tag):
    def once():
        check(lib().dt_adam_rows_step(ptr(table), ptr(m), ptr(v), ptr(rows), ptr(vals), n, D, fields, ptr(slots),
                                      slots.numel(), ptr(mark), 0.0, 0.9, 0.999, 1e-7, ptr(state), None, None, None, None, 0, 1, 1e-3, stream_ptr()), 'x')
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        once()
    e1.record()
    torch.cuda.synchronize()
    print(f'{tag:40s} fields={fields:2d}  {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us / call (dedupe + update)')


g = torch.Generator(device='cpu').manual_seed(0)
uni = (torch.randint(0, V, (B, F), generator=g).to(dev) + off).reshape(-1)
perm = (torch.stack([torch.randperm(V, generator=g)[:B] for _ in range(F)], 1).to(dev) + off).reshape(-1)
skip = torch.full((n,), -1, dtype=torch.int64, device=dev)
hot = (torch.randint(0, 64, (B, F), generator=g).to(dev) + off).reshape(-1)
for rows, tag in ((uni, 'uniform ids (~33 dups/field)'), (perm, 'unique ids'), (skip, 'all skipped (-1)'),
                  (hot, '64 hot ids per field')):
    run(rows, F, tag)
    run(rows, 0, tag)
