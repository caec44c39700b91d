import torch, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from deeptables_amd.models import deepnets
from oracle import headline
dev = torch.device('cuda', 0)
dm = bench.build_model(deepnets.DCN, dev, None, bench.D, bench.MODEL_PARAMS.get('DCN'))
bench.N_BATCHES = 1
batches = bench.make_batches(8192, dev, seed=1234, dist_kind="uniform")
idx, dense, y = batches[0]
orig = headline._rel
def rel2(a, b):
    r = orig(a, b)
    print('  rel', tuple(b.shape), f'{r:.3e}', 'max|ref|', f'{b.abs().max().item():.3e}')
    return r
headline._rel = rel2
import torch
with torch.no_grad():
    dm.model.layers_by_name['dcn_dense_1'].bias.add_(float(os.environ.get('SHIFT', '0')))
res = headline.check_train_step(dm, (idx, dense, y), adam=False)
print({k: v for k, v in res.items()})
