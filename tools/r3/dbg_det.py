import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import __graft_entry__ as ge
ge.build()
from deeptables_amd.models import layers as L
L.DENSE_GRAD_MAX_ELEMS = 0
import test_fused_gpu as T
dev = torch.device('cuda', 0)
for (vocab, B, F, D) in [(5000, 1000, 26, 16), (3000, 300, 7, 32), (200000, 4096, 26, 16)]:
    dm, cats = T.build(F, 13, D, vocab=vocab)
    idx, dense, y = T.batch(cats, 13, B, seed=5)
    ins = [idx.to(torch.int32).to(dev), dense.to(dev)]; yy = y.to(dev)
    dm.model.train()
    plan = dm.fused_plan()
    outs = []
    for rep in range(6):
        dm.forward_backward(ins, yy)
        torch.cuda.synchronize()
        buf = plan._bufs[B]
        ws = buf['ws']
        outs.append((plan.accum.clone(), buf['grad_rows'].clone(), buf['rows'].clone(), buf['logit'].clone()))
    for rep in range(1, 6):
        a, b = outs[0], outs[rep]
        n_flat = plan.off['dwlin'] + F + 13
        da = (a[0][:n_flat] - b[0][:n_flat]).abs()
        k = int(da.argmax())
        names = sorted(plan.off.items(), key=lambda kv: kv[1])
        where = [n for n, o in names if o <= k][-1]
        print(vocab, B, F, D, 'rep', rep, 'accum max diff', da.max().item(), 'at', k, where, '| grad_rows diff', (a[1] - b[1]).abs().max().item(),
              '| rows equal', bool(torch.equal(a[2], b[2])), '| logit diff', (a[3] - b[3]).abs().max().item())
