import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import __graft_entry__ as ge
ge.build()
from deeptables_amd.models import layers as L
L.DENSE_GRAD_MAX_ELEMS = 0
import test_fused_gpu as T
from oracle import headline
dev = torch.device('cuda', 0)
vocab, B, F, D = 5000, 1000, 26, 16
for steps in (1, 2):
    dm, cats = T.build(F, 13, D, vocab=vocab)
    idx, dense, y = T.batch(cats, 13, B, seed=5)
    b = (idx.to(torch.int32).to(dev), dense.to(dev), y.to(dev))
    # per-parameter errors
    emb = dm.model.layers_by_name['emb_categorical_vars_all']
    opt = dm.optimizer
    names = [n for n, _ in headline.dense_parameters(dm)]
    import types
    orig = headline.check_rows_in_step
    res = orig(dm, b, steps=steps)
    print('steps', steps, {k: v for k, v in res.items()})
# detailed: run path A and B manually with 1 step and compare each dense param
dm, cats = T.build(F, 13, D, vocab=vocab)
idx, dense, y = T.batch(cats, 13, B, seed=5)
ins = [idx.to(torch.int32).to(dev), dense.to(dev)]; yy = y.to(dev)
dm2, _ = T.build(F, 13, D, vocab=vocab)
dm.model.train(); dm2.model.train()
for st in range(3):
    dm.forward_backward(ins, yy); dm.optimizer.step()
    dm2._forward_backward(ins, yy, apply_rows=True); dm2.optimizer.step()
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(headline.dense_parameters(dm), headline.dense_parameters(dm2)):
        e = (p.detach() - q.detach()).abs().max().item()
        if e > 1e-7:
            print('step', st, n, tuple(p.shape), 'max abs diff', e, 'max |p|', p.detach().abs().max().item())
    t1 = dm.model.layers_by_name['emb_categorical_vars_all'].tables['d16']; t2 = dm2.model.layers_by_name['emb_categorical_vars_all'].tables['d16']
    print('step', st, 'table diff', (t1.detach() - t2.detach()).abs().max().item(), 't', dm.optimizer.t, dm2.optimizer.t)
