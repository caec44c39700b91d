# -*- coding:utf-8 -*-
"""CPU ORACLE (test infrastructure) — reads the weights out of a deeptables_amd DeepModel and
evaluates the oracle's restatement of the reference graph (reference_layers.model_forward) on the
same inputs.  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import torch

from . import reference_layers as R


def _t(p, dtype):
    return p.detach().to('cpu', dtype).clone()


def oracle_weights(dm, dtype=torch.float64, requires_grad=False, tables=True):
    """deeptables_amd.models.deepmodel.DeepModel -> weights dict for R.model_forward.
    tables=False leaves 'emb_categorical_vars_all' empty (oracle/headline.py fills it with row-lookup stand-ins so
    the 26 x 1M-row benchmark tables are never copied in float64)."""
    m = dm.model
    L = m.layers_by_name
    w = {}

    def g(t):
        t = _t(t, dtype)
        return t.requires_grad_(True) if requires_grad else t

    emb = L.get('emb_categorical_vars_all')
    w['emb_categorical_vars_all'] = [g(e) for e in emb.embeddings] if (emb is not None and tables) else []
    vls = [l for l in L.values() if l.__class__.__name__ == 'VarLenColumnEmbedding']
    if vls:
        w['var_len_tables'] = [g(l.embeddings) for l in vls]
    bn = L.get('bn_concat_emb_dense')     # absent when no net consumes concat_emb_dense (e.g. AutoInt)
    if bn is not None:
        w['bn_concat_emb_dense'] = (g(bn.gamma), g(bn.beta), _t(bn.moving_mean, dtype),
                                    _t(bn.moving_variance, dtype))
    if 'linear_logit' in L:
        w['linear_logit'] = g(L['linear_logit'].kernel)

    def dnn(prefix):
        out, i = [], 1
        while f'{prefix}_dense_{i}' in L:
            d = L[f'{prefix}_dense_{i}']
            cell = (g(d.kernel), None if d.bias is None else g(d.bias))
            bn_i = L.get(f'{prefix}_bn_{i}')                           # batch_norm cell (deepnets.py:420-421)
            if bn_i is not None:
                cell += ((g(bn_i.gamma), g(bn_i.beta), _t(bn_i.moving_mean, dtype), _t(bn_i.moving_variance, dtype)),)
            out.append(cell)
            i += 1
        return out

    for prefix, key in (('dnn', 'dnn'), ('dcn', 'dcn_dnn'), ('opnn', 'opnn'), ('ipnn', 'ipnn'), ('pnn', 'pnn'),
                        ('cross_dnn', 'cross_dnn'), ('fibi_dnn', 'fibi_dnn'), ('fgcnn_dnn', 'fgcnn_dnn'),
                        ('fgcnn_ipnn', 'fgcnn_ipnn')):
        if f'{prefix}_dense_1' in L:
            w[key] = dnn(prefix)
    cross_keys = {'dcn_cross_layer': 'dcn_cross', 'cross_layer': 'cross', 'cross_dnn_layer': 'cross_dnn'}
    for name, layer in L.items():
        if name.startswith('dense_logit_'):
            w[name] = g(layer.kernel)
        cls = layer.__class__.__name__
        if cls == 'CIN':
            w['cin_filters'] = [g(f) for f in layer.f_]
            w['cin_bias'] = [g(b) for b in layer.bias] if layer.use_bias else None
            w['cin_exFM_out'] = (g(layer.exFM_out.kernel), g(layer.exFM_out.bias))
        elif cls == 'Cross' and name in cross_keys:
            w[cross_keys[name] + '_kernels'] = [g(k) for k in layer.kernels]
            w[cross_keys[name] + '_bias'] = [g(b) for b in layer.bias]
        elif cls == 'OuterProduct' and name in ('outer_product_layer', 'pnn_outer_product_layer'):
            w['opnn_kernel' if name == 'outer_product_layer' else 'pnn_kernel'] = g(layer.kernel)
        elif cls == 'AFM':
            da = layer.dense_attention
            w.setdefault('afm', []).append({
                'att_kernel': g(da.kernel), 'att_bias': None if da.bias is None else g(da.bias),
                'projection_h': g(layer.attention_p), 'out_kernel': g(layer.dense_out.kernel),
                'activation': layer.activation_function})
        elif cls == 'SENET':
            w.setdefault('senet', []).append({
                'att1': (g(layer.dense_att1.kernel), g(layer.dense_att1.bias)),
                'att2': (g(layer.dense_att2.kernel), g(layer.dense_att2.bias))})
        elif cls == 'BilinearInteraction':
            Wg = g(layer.W)
            key = 'senet' if name.startswith('senet_bilinear') else 'embedding'
            w.setdefault('bilinear', {})[key] = Wg          # [nW,D,D], slices in creation order
        elif cls == 'FGCNN':
            w.setdefault('fgcnn', []).append({
                'conv_kernel': g(layer.conv_kernel), 'conv_bias': g(layer.conv_bias),
                'dense_kernel': g(layer.dense_output.kernel), 'dense_bias': g(layer.dense_output.bias)})
        elif cls == 'MultiheadAttention':
            w.setdefault('autoint_layers', []).append({
                'Q': (g(layer.dense_Q.kernel), g(layer.dense_Q.bias)),
                'K': (g(layer.dense_K.kernel), g(layer.dense_K.bias)),
                'V': (g(layer.dense_V.kernel), g(layer.dense_V.bias)),
                'R': (g(layer.dense_residual.kernel), g(layer.dense_residual.bias)),
                'bn': (g(layer.batch_normalize.gamma), g(layer.batch_normalize.beta),
                       _t(layer.batch_normalize.moving_mean, dtype), _t(layer.batch_normalize.moving_variance, dtype)),
            })
    out = L['task_output']
    w['task_output'] = (g(out.kernel), None if out.bias is None else g(out.bias))
    return w


def oracle_config(dm):
    c = dm.config
    return {'cin_params': c.cin_params, 'autoint_params': c.autoint_params, 'fibinet_params': c.fibinet_params,
            'fgcnn_params': c.fgcnn_params, 'pnn_params': c.pnn_params,
            'dnn_activation': c.dnn_params.get('activation', 'relu'), 'stacking_op': c.stacking_op, 'task': dm.task}


def oracle_forward(dm, cat, dense, dtype=torch.float64, training=True, weights=None, var_len=None):
    """-> (logit [B,units], activation) of the oracle for the model's current weights.  The nets run in the order of
    the model's layers (dm.config.nets is the order DeepModel._build_model iterated)."""
    w = weights if weights is not None else oracle_weights(dm, dtype)
    cat_f = None if cat is None else cat.detach().cpu().to(torch.float32)     # reference contract: float32 ids
    dn = None if dense is None else dense.detach().cpu().to(dtype)
    vl = None if not var_len else [v.detach().cpu().to(torch.float32) for v in var_len]
    return R.model_forward(w, cat_f, dn, dm.config.nets, oracle_config(dm), training=training, var_len_idx=vl)


def param_pairs(dm, w):
    """Walks an oracle weights dict `w` (or a dict of the same structure holding gradients) along the DeepModel's layers ->
    [(parameter of the model, entry of w)], embedding tables excluded (they are one packed parameter per dimension:
    `emb.set_embeddings` / `emb.tables`).  Entries of `w` the model has no layer for are skipped."""
    L = dm.model.layers_by_name
    out = []

    def dense(layer, kb):
        out.append((layer.kernel, kb[0]))
        if layer.bias is not None and len(kb) > 1 and kb[1] is not None:
            out.append((layer.bias, kb[1]))

    vls = [l for l in L.values() if l.__class__.__name__ == 'VarLenColumnEmbedding']
    for layer, table in zip(vls, w.get('var_len_tables', [])):
        out.append((layer.embeddings, table))
    if 'bn_concat_emb_dense' in L:
        bn = L['bn_concat_emb_dense']
        out.append((bn.gamma, w['bn_concat_emb_dense'][0]))
        out.append((bn.beta, w['bn_concat_emb_dense'][1]))
    if 'linear_logit' in L:
        out.append((L['linear_logit'].kernel, w['linear_logit']))
    for prefix, key in (('dnn', 'dnn'), ('dcn', 'dcn_dnn'), ('opnn', 'opnn'), ('ipnn', 'ipnn'), ('pnn', 'pnn'),
                        ('cross_dnn', 'cross_dnn'), ('fibi_dnn', 'fibi_dnn'), ('fgcnn_dnn', 'fgcnn_dnn'),
                        ('fgcnn_ipnn', 'fgcnn_ipnn')):
        i = 1
        while f'{prefix}_dense_{i}' in L and key in w:
            cell = w[key][i - 1]
            dense(L[f'{prefix}_dense_{i}'], cell)
            if len(cell) > 2 and cell[2] is not None:                      # batch_norm cell (deepnets.py:420-421)
                bn = L[f'{prefix}_bn_{i}']
                out.append((bn.gamma, cell[2][0]))
                out.append((bn.beta, cell[2][1]))
            i += 1
    seen = {'att': 0, 'afm': 0, 'senet': 0, 'fgcnn': 0}
    cross_keys = {'dcn_cross_layer': 'dcn_cross', 'cross_layer': 'cross', 'cross_dnn_layer': 'cross_dnn'}
    for name, layer in L.items():
        cls = layer.__class__.__name__
        if name.startswith('dense_logit_'):
            out.append((layer.kernel, w[name]))
        elif cls == 'CIN':
            out.extend(zip(layer.f_, w['cin_filters']))
            if layer.use_bias and w.get('cin_bias') is not None:
                out.extend(zip(layer.bias, w['cin_bias']))
            dense(layer.exFM_out, w['cin_exFM_out'])
        elif cls == 'Cross' and name in cross_keys:
            key = cross_keys[name]
            out.append((layer.kernel_stack, torch.stack([torch.as_tensor(k).reshape(-1) for k in w[key + '_kernels']])))
            out.append((layer.bias_stack, torch.stack([torch.as_tensor(b).reshape(-1) for b in w[key + '_bias']])))
        elif cls == 'OuterProduct':
            key = {'outer_product_layer': 'opnn_kernel', 'pnn_outer_product_layer': 'pnn_kernel'}.get(name)
            if key in w:
                out.append((layer.kernel, w[key]))
        elif cls == 'AFM':
            a = w['afm'][seen['afm']]
            seen['afm'] += 1
            dense(layer.dense_attention, (a['att_kernel'], a.get('att_bias')))
            out.append((layer.attention_p, a['projection_h']))
            out.append((layer.dense_out.kernel, a['out_kernel']))
        elif cls == 'SENET':
            se = w['senet'][seen['senet']]
            seen['senet'] += 1
            dense(layer.dense_att1, se['att1'])
            dense(layer.dense_att2, se['att2'])
        elif cls == 'BilinearInteraction':
            Wv = w['bilinear']['senet' if name.startswith('senet_bilinear') else 'embedding']
            out.append((layer.W, torch.stack([torch.as_tensor(t) for t in Wv]) if isinstance(Wv, (list, tuple)) else Wv))
        elif cls == 'FGCNN':
            fw = w['fgcnn'][seen['fgcnn']]
            seen['fgcnn'] += 1
            out.append((layer.conv_kernel, fw['conv_kernel']))
            out.append((layer.conv_bias, fw['conv_bias']))
            dense(layer.dense_output, (fw['dense_kernel'], fw['dense_bias']))
        elif cls == 'MultiheadAttention':
            lw = w['autoint_layers'][seen['att']]
            seen['att'] += 1
            for dn, key in ((layer.dense_Q, 'Q'), (layer.dense_K, 'K'), (layer.dense_V, 'V'),
                            (layer.dense_residual, 'R')):
                if dn is not None and key in lw:
                    dense(dn, lw[key])
            out.append((layer.batch_normalize.gamma, lw['bn'][0]))
            out.append((layer.batch_normalize.beta, lw['bn'][1]))
    dense(L['task_output'], w['task_output'])
    return out


def load_weights(dm, w):
    """Inverse of oracle_weights: copy an oracle weights dict (numpy/torch, any float dtype) into the DeepModel.
    Every entry of `w` that names a layer of the model is written; a layer whose weights `w` does not hold keeps its
    initial values (callers that need completeness compare oracle_weights(dm) with `w` afterwards)."""
    import numpy as np
    emb = dm.model.layers_by_name.get('emb_categorical_vars_all')
    if emb is not None:
        emb.set_embeddings(w['emb_categorical_vars_all'])
    for param, value in param_pairs(dm, w):
        v = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
        with torch.no_grad():
            param.copy_(torch.as_tensor(v, dtype=torch.float32).reshape(param.shape))


def model_from_reference_fixture(static, tensors, device):
    """A deeptables_amd DeepModel equal to a whole-model fixture of tests/golden/make_reference_golden.py
    (`reference_code_model_*.npz`): the same ModelConfig / column descriptors the reference's DeepModel.__build_model was
    given there, the fixture's weights copied in.  -> (DeepModel, ids [B,F] float32, dense [B,Nd] float32 or None)."""
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    cfg, b = static['config'], static['config']['build']
    w = tensors['weights']
    conf = ModelConfig(nets=list(static['nets']), fixed_embedding_dim=True, embeddings_output_dim=b['embeddings_output_dim'],
                       embedding_dropout=0, dense_dropout=0, stacking_op=cfg['stacking_op'],
                       output_use_bias=b['output_use_bias'],
                       dnn_params=dict(b['dnn_params'], hidden_units=tuple(tuple(h) for h in b['dnn_params']['hidden_units'])),
                       cross_params=b['cross_params'], afm_params=b['afm_params'], cin_params=cfg['cin_params'],
                       autoint_params=cfg['autoint_params'], fibinet_params=cfg['fibinet_params'],
                       fgcnn_params={k: tuple(v) for k, v in cfg['fgcnn_params'].items()}, pnn_params=cfg['pnn_params'])
    cats = [CategoricalColumn(f'C{i}', int(t.shape[0]), int(t.shape[1])) for i, t in enumerate(w['emb_categorical_vars_all'])]
    dense = tensors['dense']
    conts = [] if dense is None else [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(dense.shape[1])])]
    vl_cols = []
    for name, vocab, max_len in b.get('var_len', []):
        from deeptables_amd.models.metainfo import VarLenCategoricalColumn
        col = VarLenCategoricalColumn(name, int(vocab), b['embeddings_output_dim'])
        col.max_elements_length = int(max_len)
        vl_cols.append(col)
    dm = DeepModel(cfg['task'], b['num_classes'], conf, cats, conts, var_categorical_len_columns=vl_cols or None)
    dm.build(device)
    # build() keeps net order as ModelConfig returns it; the fixture's order is the one its weights were created in
    load_weights(dm, w)
    return dm, tensors['cat_idx'].to(torch.float32), None if dense is None else dense.to(torch.float32)
