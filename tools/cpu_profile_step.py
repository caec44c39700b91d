"""cProfile of eager train steps (host-side overhead): python tools/cpu_profile_step.py [model]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeptables_amd.models import deepnets
name = sys.argv[1] if len(sys.argv) > 1 else 'DCN'
dev = torch.device('cuda', 0)
dm = bench.build_model(getattr(deepnets, name), dev)
batches = bench.make_batches(8192, dev, 1)
dm.model.train()
for i in range(10):
    dm.train_step([batches[i][0], batches[i][1]], batches[i][2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    b = batches[i % 50]
    dm.train_step([b[0], b[1]], b[2])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
print('\n'.join(s.getvalue().split('\n')[:32]))
