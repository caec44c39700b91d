import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import __graft_entry__ as ge
ge.build()
from deeptables_amd.models import layers as L
L.DENSE_GRAD_MAX_ELEMS = 0
import test_fused_gpu as T
from oracle import headline
dev = torch.device('cuda', 0)
NS = int(os.environ.get('NSTEPS', '1'))
for (vocab, B, F, D) in [(200000, 4096, 26, 16), (3000, 300, 7, 32)]:
    dm, cats = T.build(F, 13, D, vocab=vocab)
    idx, dense, y = T.batch(cats, 13, B, seed=5)
    ins = [idx.to(torch.int32).to(dev), dense.to(dev)]; yy = y.to(dev)
    dm.model.train()
    emb = dm.model.layers_by_name['emb_categorical_vars_all']
    key = f'd{D}'; table = emb.tables[key]; opt = dm.optimizer; plan = dm.fused_plan()
    slots = opt._st(table, rows=True)
    dps = [p for _, p in headline.dense_parameters(dm)]
    names = [n for n, _ in headline.dense_parameters(dm)]
    def snap():
        return {'table': table.detach().clone(), 'mv': slots['mv'].clone(), 'dense': [p.detach().clone() for p in dps],
                'dm': [opt._st(p)['m'].clone() for p in dps], 'dv': [opt._st(p)['v'].clone() for p in dps],
                'bn': [b.detach().clone() for b in dm.model.buffers()], 't': opt.t}
    def restore(s):
        with torch.no_grad():
            table.data.copy_(s['table']); slots['mv'].copy_(s['mv'])
            for p, v, m_, v_ in zip(dps, s['dense'], s['dm'], s['dv']):
                p.data.copy_(v); opt._st(p)['m'].copy_(m_); opt._st(p)['v'].copy_(v_)
            for b, v in zip(dm.model.buffers(), s['bn']): b.copy_(v)
        opt.t = s['t']
    s0 = snap()
    restore(s0)
    for _ in range(NS):
        dm.forward_backward(ins, yy); opt.step()
    torch.cuda.synchronize()
    ref = snap()
    rows_b = plan._bufs[B]['rows'].reshape(-1).clone()
    for rep in range(12):
        restore(s0)
        for _ in range(NS):
            dm._forward_backward(ins, yy, apply_rows=True); opt.step()
        torch.cuda.synchronize()
        b = snap()
        dt = (ref['table'] - b['table']).abs().amax(1)
        bad = torch.nonzero(dt > 1e-7).reshape(-1)
        msg = f'{vocab},{B},{F},{D} rep {rep}: rows off {bad.numel()}'
        if bad.numel():
            da = (ref['table'][bad] - s0['table'][bad]); db = (b['table'][bad] - s0['table'][bad])
            ratio = (db.reshape(-1) / da.reshape(-1).clamp_min(1e-30).where(da.reshape(-1).abs() > 1e-6, torch.ones_like(da.reshape(-1))))
            sel = da.reshape(-1).abs() > 1e-6
            r = (db.reshape(-1)[sel] / da.reshape(-1)[sel])
            single = torch.isin(bad, rows_b[rows_b >= 0])
            msg += f' single {int(single.sum())} ratio(B step / A step) median {r.median().item():.4f} min {r.min().item():.4f} max {r.max().item():.4f}; first bad rows {bad[:6].tolist()}'
        dd = [(n, f'{(p - q).abs().max().item():.2e}') for n, p, q in zip(names, ref['dense'], b['dense']) if (p - q).abs().max().item() > 1e-7]
        print(msg, 'dense off', dd, 't', b['t'])
