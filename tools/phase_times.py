# Phase timing of the fused DeepFM kernels from s_memtime stamps (run with DT_DEEPFM_STAMPS=1 on the GPU box).
import os, sys
os.environ['DT_DEEPFM_STAMPS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from deeptables_amd.models import deepnets
from deeptables_amd._lib import lib
dev = torch.device('cuda', 0)
dm = bench.build_model(deepnets.DeepFM, dev)
batches = bench.make_batches(8192, dev, 1)
dm.model.train()
for i in range(5):
    dm.forward_backward([batches[i][0], batches[i][1]], batches[i][2])
torch.cuda.synchronize()
plan = dm.fused_plan()
ws = plan._bufs[8192]['ws']
off = lib().dt_deepfm_stamps_offset_floats(8192, plan.F, plan.D, plan.Nd)
tiles = 256
raw = ws[off: off + 2 * tiles * 8 * 2].cpu().numpy().view(np.uint64).reshape(2, tiles, 8).astype(np.float64)
names = [['start', 'staged', 'gemm1', 'h1 stored', 'gemm2+h2', 'end'], ['start', 'prologue', 'dH1', 'end(dXn)']]
for k, kn in enumerate(['k_mlp_fwd', 'k_mlp_bwd']):
    st = raw[k][:, :len(names[k])]
    d = np.diff(st, axis=1)
    print(kn, 'phase cycles (mean over blocks, s_memtime ticks @100MHz => x10 ns):')
    for j in range(d.shape[1]):
        print(f'   {names[k][j]:>10s} -> {names[k][j+1]:<10s} mean {d[:, j].mean():9.0f}  min {d[:, j].min():9.0f}  max {d[:, j].max():9.0f}')
    print('   total', (st[:, -1] - st[:, 0]).mean(), ' span over blocks', st[:, -1].max() - st[:, 0].min())
