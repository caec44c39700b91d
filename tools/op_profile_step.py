"""torch.profiler view of one eager train step: which autograd nodes / ops launch the small fill / add kernels.
python tools/op_profile_step.py [model]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from deeptables_amd.models import deepnets
name = sys.argv[1] if len(sys.argv) > 1 else 'AutoInt'
dev = torch.device('cuda', 0)
dm = bench.build_model(getattr(deepnets, name), dev, None, 32 if name == 'AutoInt' else 16, bench.MODEL_PARAMS.get(name))
batches = bench.make_batches(8192, dev, 1)
dm.model.train()
for i in range(5):
    dm.train_step([batches[i][0], batches[i][1]], batches[i][2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(5):
        b = batches[i]
        dm.train_step([b[0], b[1]], b[2])
    torch.cuda.synchronize()
rows = [(e.key, e.count / 5, e.device_time_total / 5) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[2])
for k, c, t in rows[:40]:
    print(f'{t:9.1f} us/step  x{c:6.1f}  {k[:80]}')
