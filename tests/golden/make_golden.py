# -*- coding:utf-8 -*-
"""Generates the known-answer fixtures the reference never had: seeded inputs and
weights, float64 oracle outputs and gradients, per layer and for the four BASELINE.json model
graphs.  Run from the repo root:  python tests/golden/make_golden.py

The reference itself cannot be executed here (TensorFlow/Keras/hypernets are not installable —
`import tensorflow` fails), so these vectors come from the oracle (oracle/reference_layers.py, a
literal restatement of deeptables/models/layers.py + deepmodel.py) after it has been cross-checked
against the independent closed forms (oracle/closed_form.py).  PARITY UNPINNED by reference tests.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_layers as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
F64 = torch.float64


def gen(seed):
    return torch.Generator().manual_seed(seed)


def f32(t):
    """round inputs to float32 so the GPU sees exactly the numbers the oracle used"""
    return t.float().double()


def layer_cases():
    out = {}
    for tag, (B, F, D) in {'tiny': (5, 4, 3), 'medium': (64, 26, 16)}.items():
        g = gen(1234 + B)
        x = f32(torch.randn(B, F, D, generator=g, dtype=F64) * 0.5).requires_grad_(True)
        xs = [x[:, i:i + 1] for i in range(F)]
        P = F * (F - 1) // 2
        # FM
        up = f32(torch.randn(B, 1, generator=g, dtype=F64))
        y = R.fm(x)
        (gx,) = torch.autograd.grad((y * up).sum(), x)
        out[f'fm_{tag}'] = dict(x=x, up=up, y=y, gx=gx)
        # InnerProduct
        up = f32(torch.randn(B, P, generator=g, dtype=F64))
        y = R.inner_product(xs)
        (gx,) = torch.autograd.grad((y * up).sum(), x)
        out[f'ip_{tag}'] = dict(x=x, up=up, y=y, gx=gx)
        # OuterProduct x3
        for kt, shape in (('mat', (D, P, D)), ('vec', (P, D)), ('num', (P, 1))):
            k = f32(torch.randn(shape, generator=g, dtype=F64) * 0.3).requires_grad_(True)
            y = R.outer_product(xs, k, kt)
            gx, gk = torch.autograd.grad((y * up).sum(), [x, k])
            out[f'op_{kt}_{tag}'] = dict(x=x, k=k, up=up, y=y, gx=gx, gk=gk)
        # Cross
        C, L = (7, 2) if tag == 'tiny' else (429, 6)
        xc = f32(torch.randn(B, C, generator=g, dtype=F64) * 0.5).requires_grad_(True)
        w = f32(torch.randn(L, C, generator=g, dtype=F64) / np.sqrt(C)).requires_grad_(True)
        b = f32(torch.randn(L, C, generator=g, dtype=F64) * 0.1).requires_grad_(True)
        up = f32(torch.randn(B, C, generator=g, dtype=F64))
        y = R.cross(xc, [w[i].unsqueeze(1) for i in range(L)], [b[i].unsqueeze(1) for i in range(L)])
        gx, gw, gb = torch.autograd.grad((y * up).sum(), [xc, w, b])
        out[f'cross_{tag}'] = dict(x=xc, w=w, b=b, up=up, y=y, gx=gx, gw=gw, gb=gb)
        # CIN (full layer stack, direct=False) -> result before exFM Dense
        cls = (6, 4, 5) if tag == 'tiny' else (32, 32, 16)
        fn, filters = [F], []
        for ls in cls:
            filters.append(f32(torch.randn(1, fn[-1] * F, ls, generator=g, dtype=F64) *
                               np.sqrt(2.0 / (fn[-1] * F))).requires_grad_(True))
            fn.append(ls // 2)
        y = R.cin(x, filters, None, cls, 'relu', False, return_hidden=True)
        up = f32(torch.randn(y.shape, generator=g, dtype=F64))
        grads = torch.autograd.grad((y * up).sum(), [x] + filters)
        d = dict(x=x, up=up, y=y, gx=grads[0], cls=torch.tensor(cls))
        for i, (fl, gf) in enumerate(zip(filters, grads[1:])):
            d[f'f{i}'] = fl
            d[f'gf{i}'] = gf
        out[f'cin_{tag}'] = d
        # MultiheadAttention core via the full layer (H heads)
        Dm, H = (8, 2) if tag == 'tiny' else (32, 4)
        xm = f32(torch.randn(B, F, Dm, generator=g, dtype=F64) * 0.5).requires_grad_(True)
        wts = {}
        for kq in 'QKVR':
            wts[kq] = (f32(torch.randn(Dm, Dm, generator=g, dtype=F64) * np.sqrt(2.0 / Dm)),
                       f32(torch.randn(Dm, generator=g, dtype=F64) * 0.1))
        wts['bn'] = (f32(1 + 0.1 * torch.randn(Dm, generator=g, dtype=F64)),
                     f32(0.1 * torch.randn(Dm, generator=g, dtype=F64)))
        y = R.multihead_attention(xm, wts, H, True, training=True)
        up = f32(torch.randn(y.shape, generator=g, dtype=F64))
        (gx,) = torch.autograd.grad((y * up).sum(), xm)
        d = dict(x=xm, up=up, y=y, gx=gx, H=torch.tensor(H))
        for kq in 'QKVR':
            d[f'k{kq}'], d[f'b{kq}'] = wts[kq]
        d['gamma'], d['beta'] = wts['bn']
        out[f'mha_{tag}'] = d
    # embedding gather incl. float ids with fractional parts (truncation) — bit-exact expectation
    g = gen(77)
    vocabs = [11, 7, 13, 5]
    tables = [(torch.rand(v, 8, generator=g) * 0.1 - 0.05) for v in vocabs]
    idx = torch.stack([torch.randint(0, v, (32,), generator=g) for v in vocabs], 1).float()
    idx[::3] += 0.75          # 3.75 -> 3 under keras.ops.cast(int32)
    emb = torch.cat(R.multi_column_embedding(idx, tables), dim=1)
    d = dict(idx=idx, emb=emb)
    for i, t in enumerate(tables):
        d[f't{i}'] = t
    out['gather'] = d
    return out


def f3_cases():
    """SURVEY §8 f3 layers: AFM, BilinearInteraction, SENET, FGCNN, focal / GHM-C losses."""
    out = {}
    for tag, (B, F, D, H) in {'tiny': (5, 4, 3, 2), 'medium': (48, 26, 16, 16)}.items():
        g = gen(4321 + B)
        P = F * (F - 1) // 2
        x = f32(torch.randn(B, F, D, generator=g, dtype=F64) * 0.6).requires_grad_(True)
        xs = [x[:, i:i + 1] for i in range(F)]
        # AFM (out_kernel = Dense(1, no bias))
        Wa = f32(torch.randn(D, H, generator=g, dtype=F64) * 0.4).requires_grad_(True)
        ba = f32(torch.randn(H, generator=g, dtype=F64) * 0.2).requires_grad_(True)
        pv = f32(torch.randn(H, 1, generator=g, dtype=F64)).requires_grad_(True)
        wo = f32(torch.randn(D, 1, generator=g, dtype=F64)).requires_grad_(True)
        up = f32(torch.randn(B, 1, generator=g, dtype=F64))
        y = R.afm(xs, Wa, ba, pv, wo, 'relu')
        gx, gWa, gba, gpv, gwo = torch.autograd.grad((y * up).sum(), [x, Wa, ba, pv, wo])
        out[f'afm_{tag}'] = dict(x=x, Wa=Wa, ba=ba, pv=pv, wo=wo, up=up, y=y, gx=gx, gWa=gWa, gba=gba, gpv=gpv,
                                 gwo=gwo)
        # BilinearInteraction x3
        Bb = min(B, 12)                                              # [B,P,D] outputs: keep the fixture small
        xb = x[:Bb].detach().clone().requires_grad_(True)
        up = f32(torch.randn(Bb, P, D, generator=g, dtype=F64))
        for bt, nW in (('field_interaction', P), ('field_each', F - 1), ('field_all', 1)):
            W = f32(torch.randn(nW, D, D, generator=g, dtype=F64) * 0.3).requires_grad_(True)
            y = R.bilinear_interaction(xb, [W[i] for i in range(nW)], bt)
            gx, gW = torch.autograd.grad((y * up).sum(), [xb, W])
            out[f'bilinear_{bt}_{tag}'] = dict(x=xb, W=W, up=up, y=y, gx=gx, gW=gW)
        # SENET mean / max
        Rn = max(F // 3, 1)
        k1 = f32(torch.randn(F, Rn, generator=g, dtype=F64) * 0.5).requires_grad_(True)
        b1 = f32(torch.rand(Rn, generator=g, dtype=F64) * 0.5).requires_grad_(True)
        k2 = f32(torch.randn(Rn, F, generator=g, dtype=F64) * 0.5).requires_grad_(True)
        b2 = f32(torch.rand(F, generator=g, dtype=F64) * 0.5).requires_grad_(True)
        up = f32(torch.randn(B, F, D, generator=g, dtype=F64))
        for op in ('mean', 'max'):
            y = R.senet(x, (k1, b1), (k2, b2), op)
            gx, gk1, gb1, gk2, gb2 = torch.autograd.grad((y * up).sum(), [x, k1, b1, k2, b2])
            out[f'senet_{op}_{tag}'] = dict(x=x, k1=k1, b1=b1, k2=k2, b2=b2, up=up, y=y, gx=gx, gk1=gk1, gb1=gb1,
                                            gk2=gk2, gb2=gb2)
    # FGCNN: the default first block (F=26, D=4, 14 filters, height 7, pool 2, 2 new filters) and an odd-sized one
    for tag, (B, F, D, C, filters, h, pool, nf) in {'default': (8, 26, 4, 1, 14, 7, 2, 2),
                                                    'odd': (6, 9, 4, 3, 5, 4, 3, 1)}.items():
        g = gen(99 + F)
        x = f32(torch.randn(B, F, D, C, generator=g, dtype=F64)).requires_grad_(True)
        ck = f32(torch.randn(h, 1, C, filters, generator=g, dtype=F64) * 0.3).requires_grad_(True)
        cb = f32(torch.randn(filters, generator=g, dtype=F64) * 0.1).requires_grad_(True)
        Fp = -(-F // pool)
        dk = f32(torch.randn(Fp * D * filters, F * D * nf, generator=g, dtype=F64) * 0.05).requires_grad_(True)
        db = f32(torch.randn(F * D * nf, generator=g, dtype=F64) * 0.1).requires_grad_(True)
        pooled, newf = R.fgcnn(x, ck, cb, dk, db, pool, nf)
        up_p = f32(torch.randn(pooled.shape, generator=g, dtype=F64))
        up_n = f32(torch.randn(newf.shape, generator=g, dtype=F64))
        gx, gck, gcb, gdk = torch.autograd.grad((pooled * up_p).sum() + (newf * up_n).sum(), [x, ck, cb, dk])
        out[f'fgcnn_{tag}'] = dict(x=x, ck=ck, cb=cb, dk=dk, db=db, up_p=up_p, up_n=up_n, pooled=pooled, newf=newf,
                                   gx=gx, gck=gck, gcb=gcb, gdk=gdk, meta=np.array([pool, nf]))
    # losses
    g = gen(2024)
    B, C = 200, 5
    yb = (torch.rand(B, 1, generator=g) < 0.3).double()
    pb = f32(torch.rand(B, 1, generator=g, dtype=F64).clamp(1e-3, 1 - 1e-3))
    yc = torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=g), C).double()
    pc = f32(torch.softmax(torch.randn(B, C, generator=g, dtype=F64), -1))
    z1 = f32(torch.randn(B, 1, generator=g, dtype=F64) * 2)
    z2 = f32(torch.randn(B, 1, generator=g, dtype=F64) * 2)
    l1, acc = R.ghmc_loss(z1, yb, torch.zeros(10, dtype=F64))
    l2, acc = R.ghmc_loss(z2, yb, acc)
    out['losses'] = dict(yb=yb, pb=pb, yc=yc, pc=pc, z1=z1, z2=z2,
                         binary_focal=R.binary_focal_loss(yb, pb), categorical_focal=R.categorical_focal_loss(yc, pc),
                         ghmc1=l1, ghmc2=l2, ghmc_acc=acc)
    return out


MODEL_CONFIGS = {
    'DeepFM': dict(nets=['linear', 'fm_nets', 'dnn_nets'], D=16),
    'xDeepFM': dict(nets=['linear', 'cin_nets', 'dnn_nets'], D=16,
                    cin_params={'cross_layer_size': (128, 128, 128), 'activation': 'relu', 'use_residual': False,
                                'use_bias': False, 'direct': False, 'reduce_D': False}),
    'AutoInt': dict(nets=['autoint_nets'], D=32,
                    autoint_params={'num_attention': 3, 'num_heads': 4, 'dropout_rate': 0, 'use_residual': True}),
    'DCN': dict(nets=['dcn_nets'], D=16, cross_params={'num_cross_layer': 6}),
}


def model_weights(name, g, F=26, Nd=13, vocab=40):
    spec = MODEL_CONFIGS[name]
    D = spec['D']
    C = F * D + Nd
    w = {'emb_categorical_vars_all': [f32((torch.rand(vocab + i, D, generator=g, dtype=F64) * 0.1 - 0.05))
                                      for i in range(F)]}
    ru = lambda shape, fi, fo=None: f32(R.glorot_uniform(shape, g, fi, fo if fo else shape[-1], dtype=F64))
    he = lambda shape, fi: f32(R.he_uniform(shape, g, fi, dtype=F64))
    nets = spec['nets']
    if any(n in nets for n in ('dnn_nets', 'dcn_nets')):
        w['bn_concat_emb_dense'] = (f32(1 + 0.1 * torch.randn(C, generator=g, dtype=F64)),
                                    f32(0.1 * torch.randn(C, generator=g, dtype=F64)))
    mlp = lambda: [(he((C, 128), C), f32(0.05 * torch.randn(128, generator=g, dtype=F64))),
                   (he((128, 64), 128), f32(0.05 * torch.randn(64, generator=g, dtype=F64)))]
    width = {}
    if 'linear' in nets:
        w['linear_logit'] = ru((F + Nd, 1), F + Nd)
        width['linear'] = 1
    if 'fm_nets' in nets:
        width['fm_nets'] = 1
    if 'dnn_nets' in nets:
        w['dnn'] = mlp()
        width['dnn_nets'] = 64
    if 'cin_nets' in nets:
        cls = spec['cin_params']['cross_layer_size']
        fn, filters = [F], []
        for ls in cls:
            filters.append(he((1, fn[-1] * F, ls), fn[-1] * F))
            fn.append(ls // 2)
        w['cin_filters'] = filters
        res = sum(l // 2 if i != len(cls) - 1 else l for i, l in enumerate(cls))
        w['cin_exFM_out'] = (ru((res, 1), res), f32(0.05 * torch.randn(1, generator=g, dtype=F64)))
        width['cin_nets'] = 1
    if 'dcn_nets' in nets:
        L = spec['cross_params']['num_cross_layer']
        w['dcn_cross_kernels'] = [ru((C, 1), C) for _ in range(L)]
        w['dcn_cross_bias'] = [f32(0.05 * torch.randn(C, 1, generator=g, dtype=F64)) for _ in range(L)]
        w['dcn_dnn'] = mlp()
        width['dcn_nets'] = C + 64
    if 'autoint_nets' in nets:
        layers = []
        for _ in range(spec['autoint_params']['num_attention']):
            lw = {k: (he((D, D), D), f32(0.05 * torch.randn(D, generator=g, dtype=F64))) for k in 'QKVR'}
            lw['bn'] = (f32(1 + 0.1 * torch.randn(D, generator=g, dtype=F64)),
                        f32(0.1 * torch.randn(D, generator=g, dtype=F64)))
            layers.append(lw)
        w['autoint_layers'] = layers
        width['autoint_nets'] = F * D
    if len(nets) > 1:
        for n in nets:
            if width[n] > 1:
                w[f'dense_logit_{n}'] = ru((width[n], 1), width[n])
        last = 1
    else:
        last = width[nets[0]]
    w['task_output'] = (ru((last, 1), last), f32(0.05 * torch.randn(1, generator=g, dtype=F64)))
    return w


def flat_items(prefix, o, out):
    if torch.is_tensor(o):
        out[prefix] = o.detach().numpy()
    elif isinstance(o, dict):
        for k, v in o.items():
            flat_items(f'{prefix}.{k}', v, out)
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o):
            flat_items(f'{prefix}.{i}', v, out)


def model_cases():
    out = {}
    for name, spec in MODEL_CONFIGS.items():
        g = gen(hash(name) % 1000 + 4321 if False else {'DeepFM': 1, 'xDeepFM': 2, 'AutoInt': 3, 'DCN': 4}[name])
        F, Nd, B = 26, 13, 48
        w = model_weights(name, g)
        idx = torch.stack([torch.randint(0, t.shape[0], (B,), generator=g) for t in w['emb_categorical_vars_all']],
                          1).float()
        dense = f32(torch.randn(B, Nd, generator=g, dtype=F64))
        y = (torch.rand(B, 1, generator=g) < 0.3).double()
        cfg = {'cin_params': spec.get('cin_params'), 'autoint_params': spec.get('autoint_params')}
        # gradient wrt the task_output kernel and the first embedding table as known answers
        w['task_output'][0].requires_grad_(True)
        w['emb_categorical_vars_all'][0].requires_grad_(True)
        logit, prob = R.model_forward(w, idx, dense, spec['nets'], cfg, training=True)
        loss = R.binary_crossentropy_from_logits(logit, y)
        g_out, g_emb0 = torch.autograd.grad(loss, [w['task_output'][0], w['emb_categorical_vars_all'][0]])
        d = {'idx': idx.numpy(), 'dense': dense.numpy(), 'y': y.numpy(), 'logit': logit.detach().numpy(),
             'loss': loss.detach().numpy(), 'g_task_output_kernel': g_out.numpy(), 'g_emb0': g_emb0.numpy()}
        flat_items('w', w, d)
        out[name] = d
    return out


EXPECTED = ('y', 'gx', 'gk', 'gw', 'gb', 'gf0', 'gf1', 'gf2', 'emb', 'gWa', 'gba', 'gpv', 'gwo', 'gW', 'gk1', 'gb1', 'gk2',
            'gb2', 'pooled', 'newf', 'gck', 'gcb', 'gdk', 'binary_focal', 'categorical_focal', 'ghmc1', 'ghmc2',
            'ghmc_acc')


def save(path, d):
    """inputs are float32-representable -> stored as float32; expected values stay float64"""
    arrs = {}
    for k, v in d.items():
        a = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        if a.dtype == np.float64 and k not in EXPECTED:
            a = a.astype(np.float32)
        arrs[k] = a
    np.savez_compressed(path, **arrs)


def main():
    if '--f3' in sys.argv:          # only the f3 fixtures (leaves the earlier files byte-identical)
        for name, case in f3_cases().items():
            save(os.path.join(OUT, f'layer_{name}.npz'), case)
        print('f3 golden fixtures written to', OUT)
        return
    for name, case in list(layer_cases().items()) + list(f3_cases().items()):
        save(os.path.join(OUT, f'layer_{name}.npz'), case)
    for name, case in model_cases().items():
        # inputs / weights are float32-representable: store them as float32 to keep the fixtures small
        small = {}
        for k, v in case.items():
            a = np.asarray(v)
            small[k] = a.astype(np.float32) if (k.startswith('w.') or k in ('dense', 'idx', 'y')) else a
        np.savez_compressed(os.path.join(OUT, f'model_{name}.npz'), **small)
    print('golden fixtures written to', OUT)


if __name__ == '__main__':
    main()
