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
for (vocab, B, F, D) in [(3000, 300, 7, 32), (5000, 1000, 26, 16), (200000, 4096, 26, 16)]:
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
    res = {}
    for name, ar in (('A1', False), ('A2', False), ('B1', True), ('B2', True), ('A3', False), ('B3', True)):
        restore(s0)
        for _ in range(2):
            dm._forward_backward(ins, yy, apply_rows=ar); opt.step()
        torch.cuda.synchronize()
        res[name] = snap()
    rows_b = plan._bufs[B]['rows'].reshape(-1)
    def cmp(x, y_):
        a, b = res[x], res[y_]
        dt = (a['table'] - b['table']).abs().amax(1)
        nbad = int((dt > 1e-7).sum())
        bad_rows = torch.nonzero(dt > 1e-7).reshape(-1)
        single = torch.isin(bad_rows, rows_b[rows_b >= 0])
        dd = [(n, (p - q).abs().max().item()) for n, p, q in zip(names, a['dense'], b['dense']) if (p - q).abs().max().item() > 1e-7]
        print(f'{vocab},{B},{F},{D} {x} vs {y_}: table max {dt.max().item():.3e} rows off {nbad} (of them single-lookup rows {int(single.sum())}) '
              f'mv max {(a["mv"] - b["mv"]).abs().max().item():.3e} dense off {dd} t {a["t"]} {b["t"]}')
    for pair in (('A1', 'A2'), ('A1', 'A3'), ('B1', 'B2'), ('B1', 'B3'), ('A1', 'B1'), ('A2', 'B2')):
        cmp(*pair)
