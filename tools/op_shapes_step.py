"""which autograd node / python line issues the small torch ops of one eager train step (zeros, add_, cat, copy_): shapes +
the chain of enclosing profiler events.  python tools/op_shapes_step.py [model]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from deeptables_amd.models import deepnets
name = sys.argv[1] if len(sys.argv) > 1 else 'xDeepFM'
dev = torch.device('cuda', 0)
dm = bench.build_model(getattr(deepnets, name), dev, None, 32 if name == 'AutoInt' else 16, bench.MODEL_PARAMS.get(name))
batches = bench.make_batches(8192, dev, 1)
dm.model.train()
for i in range(4):
    dm.train_step([batches[i][0], batches[i][1]], batches[i][2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    b = batches[0]
    dm.train_step([b[0], b[1]], b[2])
    torch.cuda.synchronize()
WANT = ('aten::zeros', 'aten::zeros_like', 'aten::add_', 'aten::add', 'aten::cat', 'aten::copy_', 'aten::zero_', 'aten::sum',
        'aten::mul', 'aten::contiguous', 'aten::clone')
for e in prof.events():
    if e.name not in WANT:
        continue
    chain, p = [], e.cpu_parent
    while p is not None:
        chain.append(p.name[:48])
        p = p.cpu_parent
    if chain and chain[0] in WANT:          # the inner op of a composite already listed
        continue
    stack = [s for s in (e.stack or []) if 'deeptables_amd' in s or 'bench.py' in s][:2]
    print(f'{e.name:16s} dev {e.device_time_total:7.1f} us  {str(e.input_shapes)[:70]:70s} <- {" <- ".join(chain[:3])}  {stack}')
