# Two identical small dense-gradient DeepFM models stepped in lockstep on the same batches: where do they first differ?
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_compiled_gpu as T
dev = torch.device('cuda', 0)
df, y = T._frame(64 * 23 + 17)
a, b = T._model('DeepFM'), T._model('DeepFM')
a.model.train(); b.model.train()
cats = [c for c in df.columns if c.startswith('C')]
conts = [c for c in df.columns if c.startswith('I')]
idx_all = torch.tensor(df[cats].values, dtype=torch.int32, device=dev)
dense_all = torch.tensor(df[conts].values, dtype=torch.float32, device=dev)
y_all = torch.tensor(y, dtype=torch.float32, device=dev).reshape(-1, 1)
g = torch.Generator().manual_seed(1)
first = None
for step in range(400):
    sel = torch.randperm(idx_all.shape[0], generator=g)[:64].to(dev)
    idx, dense, yy = idx_all[sel].contiguous(), dense_all[sel].contiguous(), y_all[sel].contiguous()
    outs = []
    for m in (a, b):
        m.train_step([idx, dense], yy)
        torch.cuda.synchronize()
        plan = m.fused_plan()
        buf = plan._bufs[64]
        outs.append({'grad_rows': buf['grad_rows'].clone(), 'rows': buf['rows'].clone(), 'accum': plan.accum.clone(),
                     'ws': buf['ws'].clone(), 'table': plan.emb.tables['d16'].detach().clone(),
                     'logit': buf['logit'].clone()})
    d = {k: float((outs[0][k].double() - outs[1][k].double()).abs().max()) for k in outs[0]}
    bad = {k: v for k, v in d.items() if (v > 1e-5 and k != 'ws')}
    if bad:
        print('step', step, 'differs:', bad, flush=True)
        if first is None:
            first = step
            ws0, ws1 = outs[0]['ws'], outs[1]['ws']
            nz = (ws0 != ws1).nonzero().flatten()
            print('   ws words differing:', nz.numel(), 'first', nz[:8].tolist(), 'last', nz[-8:].tolist(), 'of', ws0.numel())
            gr = ((outs[0]['grad_rows'] - outs[1]['grad_rows']).abs() > 1e-6).any(-1).nonzero()
            print('   grad_rows lookups differing > 1e-6:', gr.shape[0], gr[:6].tolist())
            td = (outs[0]['table'] - outs[1]['table'])
            rows = (td.abs() > 1e-5).any(-1).nonzero().flatten()
            print('   table rows differing:', rows.numel(), rows[:10].tolist(), 'max', float(td.abs().max()), 'signs', int((td > 1e-5).sum()), int((td < -1e-5).sum()))
            looked = set((idx.long() + torch.tensor([sum(T.V + i for i in range(f)) for f in range(T.F)], device=dev)).flatten().tolist())
            print('   rows looked up this step:', len(looked), 'of the differing rows looked up now:', sum(1 for r in rows.tolist() if r in looked))
            ac = (outs[0]['accum'] - outs[1]['accum']).abs()
            print('   accum max diff', float(ac.max()), 'logit diff', d['logit'])
        if step > (first or 0) + 3:
            break
print('done; first divergence at step', first)
