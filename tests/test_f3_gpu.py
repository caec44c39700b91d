# -*- coding:utf-8 -*-
"""SURVEY §8 f3 layers on the GPU — AFM, SENET, BilinearInteraction, FGCNN, VarLenColumnEmbedding, the focal /
GHM-C losses, and the nets built from them (afm_nets, fibi_*, fgcnn_*) — against the oracle's op-for-op
restatement (oracle/reference_layers.py).  Tolerance: 1e-4 absolute on outputs/logits (north_star),
2e-4 relative on gradients (fp32 kernels vs float64 oracle)."""
import itertools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    b = b.double()
    return (a.detach().double().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def maxerr(a, b):
    return (a.detach().double().cpu() - b.double()).abs().max().item()


# ---------------------------------------------------------------------------------------------
# kernels through the C-ABI wrappers
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,F,D,H,act', [(64, 5, 8, 16, 'relu'), (300, 26, 16, 16, 'relu'), (33, 7, 12, 4, 'linear'),
                                         (17, 3, 64, 40, 'relu'), (1, 2, 4, 1, 'relu'), (40, 6, 8, 12, 'tanh'),
                                         (40, 6, 8, 12, 'sigmoid'), (40, 6, 8, 12, 'elu'), (40, 6, 8, 12, 'selu'),
                                         (40, 6, 8, 12, 'softplus'), (40, 6, 8, 12, 'softsign'),
                                         (40, 6, 8, 12, 'exponential')])
def test_afm_pool(dev, B, F, D, H, act):
    from deeptables_amd import ops
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(B + F)
    x = torch.randn(B, F, D, generator=g) * 0.7
    Wa = torch.randn(D, H, generator=g) * 0.4
    ba = torch.randn(H, generator=g) * 0.2
    pv = torch.randn(H, 1, generator=g)
    go = torch.randn(B, D, generator=g)
    xd, Wd, bd, pd = [t.double().requires_grad_(True) for t in (x, Wa, ba, pv)]
    eye = torch.eye(D, dtype=torch.float64)
    ref = R.afm([xd[:, i:i + 1] for i in range(F)], Wd, bd, pd, eye, act)       # out_kernel = I -> pooled vector
    ref.backward(go.double())
    xg, Wg, bg, pg = [t.to(dev).requires_grad_(True) for t in (x, Wa, ba, pv)]
    out = ops.afm_pool(xg, Wg, bg, pg, act)
    out.backward(go.to(dev))
    assert maxerr(out, ref) < 1e-4
    for a, b, n in ((xg, xd, 'x'), (Wg, Wd, 'Wa'), (bg, bd, 'ba'), (pg, pd, 'pv')):
        if b.grad.abs().max() < 1e-9:        # linear attention: the bias shifts every logit alike -> exactly 0
            assert a.grad.abs().max().item() < 1e-5, n
        else:
            assert rel(a.grad, b.grad) < 2e-4, n


def test_afm_pool_no_bias_and_limits(dev):
    from deeptables_amd import ops, _lib
    x = torch.randn(8, 4, 8, device=dev)
    Wa = torch.randn(8, 16, device=dev)
    pv = torch.randn(16, 1, device=dev)
    out = ops.afm_pool(x, Wa, None, pv)
    assert out.shape == (8, 8) and torch.isfinite(out).all()
    with pytest.raises(_lib.DtHipError):
        ops.afm_pool(x, torch.randn(8, 100, device=dev), None, torch.randn(100, 1, device=dev))


@pytest.mark.parametrize('btype', ['field_interaction', 'field_each', 'field_all'])
@pytest.mark.parametrize('B,F,D', [(130, 6, 8), (64, 26, 16), (5, 2, 3), (70, 4, 33), (1000, 7, 16), (13, 3, 16)])
def test_bilinear(dev, btype, B, F, D):
    from deeptables_amd import ops
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(B * 7 + F)
    P = F * (F - 1) // 2
    nW = {'field_interaction': P, 'field_each': F - 1, 'field_all': 1}[btype]
    x = torch.randn(B, F, D, generator=g)
    W = torch.randn(nW, D, D, generator=g) * 0.3
    go = torch.randn(B, P, D, generator=g)
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    ref = R.bilinear_interaction(xd, [Wd[i] for i in range(nW)], btype)
    ref.backward(go.double())
    xg, Wg = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
    out = ops.bilinear_interaction(xg, Wg, btype)
    out.backward(go.to(dev))
    assert out.shape == (B, P, D)
    assert maxerr(out, ref) < 1e-4
    assert rel(xg.grad, xd.grad) < 2e-4
    assert rel(Wg.grad, Wd.grad) < 2e-4


@pytest.mark.parametrize('pooling_op', ['mean', 'max'])
def test_senet_layer(dev, pooling_op):
    from deeptables_amd import functional
    from deeptables_amd.models import layers
    from oracle import reference_layers as R
    functional.set_seed(4)
    B, F, D = 96, 26, 16
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, F, D, generator=g)
    go = torch.randn(B, F, D, generator=g)
    layer = layers.SENET(pooling_op=pooling_op, reduction_ratio=3)
    xg = x.to(dev).requires_grad_(True)
    layer.build((None, F, D))
    layer.to(dev)
    with torch.no_grad():     # he_uniform + relu can zero a whole gate: shift biases so the test exercises it
        layer.dense_att1.bias.add_(0.3)
        layer.dense_att2.bias.add_(0.2)
    out = layer(xg)
    out.backward(go.to(dev))
    xd = x.double().requires_grad_(True)
    a1 = (layer.dense_att1.kernel.detach().cpu().double().requires_grad_(True),
          layer.dense_att1.bias.detach().cpu().double().requires_grad_(True))
    a2 = (layer.dense_att2.kernel.detach().cpu().double().requires_grad_(True),
          layer.dense_att2.bias.detach().cpu().double().requires_grad_(True))
    ref = R.senet(xd, a1, a2, pooling_op)
    ref.backward(go.double())
    assert layer.reduction_num == 8
    assert maxerr(out, ref) < 1e-4
    assert rel(xg.grad, xd.grad) < 2e-4
    assert rel(layer.dense_att1.kernel.grad, a1[0].grad) < 2e-4
    assert rel(layer.dense_att2.bias.grad, a2[1].grad) < 2e-4


@pytest.mark.parametrize('F,D,C,filters,h,pool,nf', [(26, 16, 1, 14, 7, 2, 2), (13, 8, 14, 16, 7, 2, 2),
                                                     (9, 4, 1, 3, 4, 3, 1), (5, 6, 2, 4, 3, 2, 3)])
def test_fgcnn_layer(dev, F, D, C, filters, h, pool, nf):
    from deeptables_amd import functional
    from deeptables_amd.models import layers
    from oracle import reference_layers as R
    functional.set_seed(8)
    B = 32
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, F, D, C, generator=g)
    layer = layers.FGCNN(filters=filters, kernel_height=h, new_filters=nf, pool_height=pool)
    layer.build((None, F, D, C))
    layer.to(dev)
    with torch.no_grad():
        layer.conv_bias.add_(torch.randn(filters, generator=g).to(dev) * 0.1)
    xg = x.to(dev).requires_grad_(True)
    pooled, newf = layer(xg)
    gp = torch.randn(pooled.shape, generator=g)
    gn = torch.randn(newf.shape, generator=g)
    (pooled * gp.to(dev)).sum().add((newf * gn.to(dev)).sum()).backward()
    xd = x.double().requires_grad_(True)
    ws = [t.detach().cpu().double().requires_grad_(True) for t in
          (layer.conv_kernel, layer.conv_bias, layer.dense_output.kernel, layer.dense_output.bias)]
    rp, rn = R.fgcnn(xd, ws[0], ws[1], ws[2], ws[3], pool, nf)
    ((rp * gp.double()).sum() + (rn * gn.double()).sum()).backward()
    assert tuple(pooled.shape) == tuple(rp.shape) == (B, -(-F // pool), D, filters)
    assert tuple(newf.shape) == (B, F * nf, D)
    assert maxerr(pooled, rp) < 1e-4 and maxerr(newf, rn) < 1e-4
    assert rel(xg.grad, xd.grad) < 2e-4
    assert rel(layer.conv_kernel.grad, ws[0].grad) < 2e-4
    assert rel(layer.dense_output.kernel.grad, ws[2].grad) < 2e-4


def test_var_len_embedding(dev):
    from deeptables_amd import functional
    from deeptables_amd.models import layers
    from oracle import reference_layers as R
    functional.set_seed(5)
    B, L, V, D = 40, 6, 30, 8
    layer = layers.VarLenColumnEmbedding(emb_vocab_size=V, emb_output_dim=D, embeddings_initializer='uniform',
                                         embeddings_regularizer=None, activity_regularizer=None)
    layer.build((None, L))
    layer.to(dev)
    idx = torch.randint(0, V, (B, L))
    out = layer(idx.float().to(dev))
    ref = R.var_len_embedding(idx.float(), layer.embeddings.detach().cpu())
    assert out.shape == (B, 1, L * D)
    assert torch.equal(out.cpu(), ref)                        # a gather is bit-exact
    go = torch.randn(B, 1, L * D)
    out.backward(go.to(dev))
    t = layer.embeddings.detach().cpu().double().requires_grad_(True)
    R.var_len_embedding(idx.float(), t).backward(go.double())
    assert rel(layer.embeddings.grad, t.grad) < 1e-5


def test_losses(dev):
    from deeptables_amd.models import layers
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(0)
    B = 500
    y = (torch.rand(B, 1, generator=g) < 0.3).float()
    p = torch.rand(B, 1, generator=g).clamp(1e-4, 1 - 1e-4)
    pg = p.to(dev).requires_grad_(True)
    pd = p.double().requires_grad_(True)
    l = layers.BinaryFocalLoss(gamma=2., alpha=.25)(y.to(dev), pg)
    r = R.binary_focal_loss(y.double(), pd)
    l.backward(); r.backward()
    assert abs(l.item() - r.item()) < 1e-6 and rel(pg.grad, pd.grad) < 1e-4
    C = 5
    yc = torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=g), C).float()
    pc = torch.softmax(torch.randn(B, C, generator=g), -1)
    l = layers.CategoricalFocalLoss()(yc.to(dev), pc.to(dev))
    r = R.categorical_focal_loss(yc.double(), pc.double()).mean()
    assert abs(l.item() - r.item()) < 1e-6
    ghm = layers.GHMCLoss(bins=10, momentum=0.75)
    acc = torch.zeros(10, dtype=torch.float64)
    for step in range(3):                                  # the histogram state carries across calls
        z = torch.randn(B, 1, generator=g) * 2
        l = ghm.calc(z.to(dev), y.to(dev))
        r, acc = R.ghmc_loss(z.double(), y.double(), acc)
        assert abs(l.item() - r.item()) < 1e-5, step
    ghm0 = layers.GHMCLoss(bins=7, momentum=0)
    z = torch.randn(B, 1, generator=g)
    r, _ = R.ghmc_loss(z.double(), y.double(), None, bins=7, momentum=0)
    assert abs(ghm0.calc(z.to(dev), y.to(dev)).item() - r.item()) < 1e-5


# ---------------------------------------------------------------------------------------------
# whole models built from the f3 nets
# ---------------------------------------------------------------------------------------------
NETS = {
    'afm': dict(nets=['afm_nets'], afm_params={'hidden_factor': 8, 'dropout_rate': 0}),
    'afm_default': dict(nets=['linear', 'afm_nets']),
    'fibi': dict(nets=['fibi_dnn_nets']),
    'fibi_each_max': dict(nets=['linear', 'fibi_dnn_nets'],
                          fibinet_params={'senet_pooling_op': 'max', 'senet_reduction_ratio': 2,
                                          'bilinear_type': 'field_each'}),
    'fibi_all': dict(nets=['fibi_dnn_nets', 'dnn_nets'],
                     fibinet_params={'senet_pooling_op': 'mean', 'senet_reduction_ratio': 3,
                                     'bilinear_type': 'field_all'}),
    'fgcnn_dnn': dict(nets=['fgcnn_dnn_nets']),
    'fgcnn_ipnn': dict(nets=['linear', 'fgcnn_ipnn_nets'],
                       fgcnn_params={'fg_filters': (6,), 'fg_heights': (3,), 'fg_pool_heights': (2,),
                                     'fg_new_feat_filters': (2,)}),
    'fgcnn_fm': dict(nets=['fgcnn_fm_nets', 'dnn_nets']),
    'fgcnn_afm': dict(nets=['fgcnn_afm_nets']),
    'fgcnn_cin': dict(nets=['fgcnn_cin_nets'],
                      cin_params={'cross_layer_size': (16, 16), 'activation': 'relu', 'use_residual': False,
                                  'use_bias': False, 'direct': False, 'reduce_D': False}),
}


def build(name, F=9, Nd=4, vocab=30, D=8, seed=3):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    functional.set_seed(seed)
    conf = ModelConfig(fixed_embedding_dim=True, embeddings_output_dim=D, embedding_dropout=0, dense_dropout=0,
                       metrics=['AUC'], **NETS[name])
    cats = [CategoricalColumn(f'C{i}', vocab + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])] if Nd else []
    dm = DeepModel('binary', 2, conf, cats, conts)
    dm.build()
    return dm, cats


def _flat_grads(w, out):
    if torch.is_tensor(w):
        if w.requires_grad and w.grad is not None:
            out.append(w.grad)
    elif isinstance(w, dict):
        for k in sorted(w):
            _flat_grads(w[k], out)
    elif isinstance(w, (list, tuple)):
        for v in w:
            _flat_grads(v, out)
    return out


@pytest.mark.parametrize('name', list(NETS))
def test_f3_models_match_oracle(dev, name):
    from oracle import bridge, reference_layers as R
    dm, cats = build(name)
    B = 96
    g = torch.Generator().manual_seed(5)
    idx = torch.stack([torch.randint(0, c.vocabulary_size, (B,), generator=g) for c in cats], 1)
    dense = torch.randn(B, 4, generator=g)
    y = (torch.rand(B, 1, generator=g) < 0.3).float()
    with torch.no_grad():
        for n, p in dm.model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g).to(p.device) * 0.1)
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_logit, _ = bridge.oracle_forward(dm, idx, dense, training=True, weights=w)
    R.binary_crossentropy_from_logits(ref_logit, y.double()).backward()
    dm.model.train()
    dm.optimizer.zero_grad()
    logit = dm.model([idx.int().to(dev), dense.to(dev)])
    assert maxerr(logit, ref_logit) < 1e-4, name
    dm._loss(logit, y.to(dev)).backward()
    w2 = bridge.oracle_weights(dm, requires_grad=False)       # same structure, our tensors
    # walk the two weight trees in lockstep and compare every gradient the oracle produced
    L = dm.model.layers_by_name
    emb = L['emb_categorical_vars_all']
    ref_table_grad = torch.cat([t.grad for t in w['emb_categorical_vars_all']], 0)
    assert rel(emb.tables['d8'].grad, ref_table_grad) < 3e-4, name
    assert rel(L['task_output'].kernel.grad, w['task_output'][0].grad) < 3e-4
    checked = 0
    for lname, layer in L.items():
        cls = layer.__class__.__name__
        if cls == 'AFM':
            a = w['afm'][0]
            assert rel(layer.dense_attention.kernel.grad, a['att_kernel'].grad) < 3e-4
            assert rel(layer.attention_p.grad, a['projection_h'].grad) < 3e-4
            assert rel(layer.dense_out.kernel.grad, a['out_kernel'].grad) < 3e-4
            checked += 1
        elif cls == 'SENET':
            assert rel(layer.dense_att1.kernel.grad, w['senet'][0]['att1'][0].grad) < 3e-4
            checked += 1
        elif cls == 'BilinearInteraction':
            key = 'senet' if lname.startswith('senet_bilinear') else 'embedding'
            assert rel(layer.W.grad, w['bilinear'][key].grad) < 3e-4, lname
            checked += 1
        elif cls == 'FGCNN':
            i = [l for l in L.values() if l.__class__.__name__ == 'FGCNN'].index(layer)
            assert rel(layer.conv_kernel.grad, w['fgcnn'][i]['conv_kernel'].grad) < 3e-4
            assert rel(layer.dense_output.kernel.grad, w['fgcnn'][i]['dense_kernel'].grad) < 3e-4
            checked += 1
    assert checked > 0
    del w2


def test_f3_train_step_and_focal_loss(dev):
    """A FiBiNet model trains through DeepModel.train_step with a BinaryFocalLoss custom loss."""
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel, layers
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    functional.set_seed(1)
    conf = ModelConfig(nets=['fibi_dnn_nets'], fixed_embedding_dim=True, embeddings_output_dim=4, embedding_dropout=0,
                       loss=layers.BinaryFocalLoss(), metrics=['AUC'])
    cats = [CategoricalColumn(f'C{i}', 20, 4) for i in range(5)]
    conts = [ContinuousColumn('input_continuous_all', ['a', 'b'])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    dm.build()
    g = torch.Generator().manual_seed(2)
    idx = torch.randint(0, 20, (256, 5), generator=g).int().to(dev)
    dense = torch.randn(256, 2, generator=g).to(dev)
    y = ((idx[:, 0] % 2) == 0).float().reshape(-1, 1)
    dm.model.train()
    losses = [float(dm.train_step([idx, dense], y)[0]) for _ in range(30)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_var_len_columns_in_model(dev):
    """VarLenCategoricalColumn inputs: model input order (cat, var-len, continuous) as deepmodel.py:310, the
    var-len embedding joins the embeddings list, and the model trains on a frame holding padded id lists."""
    import pandas as pd
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn, VarLenCategoricalColumn
    from oracle import reference_layers as R
    functional.set_seed(2)
    conf = ModelConfig(nets=['dnn_nets'], fixed_embedding_dim=True, embeddings_output_dim=4, embedding_dropout=0,
                       metrics=['AUC'], earlystopping_patience=0)
    cats = [CategoricalColumn(f'C{i}', 10, 4) for i in range(3)]
    conts = [ContinuousColumn('input_continuous_all', ['a', 'b'])]
    vl = VarLenCategoricalColumn('genres', 12, 4)
    vl.max_elements_length = 5
    dm = DeepModel('binary', 2, conf, cats, conts, var_categorical_len_columns=[vl])
    dm.build()
    L = dm.model.layers_by_name
    assert 'emb_genres' in L and [t.name for t in dm.model.inputs][1] == 'genres'
    g = torch.Generator().manual_seed(0)
    n = 300
    idx = torch.randint(0, 10, (n, 3), generator=g)
    seq = torch.randint(0, 12, (n, 5), generator=g)
    dense = torch.randn(n, 2, generator=g)
    emb = L['emb_categorical_vars_all']
    df = pd.DataFrame({'C0': idx[:, 0], 'C1': idx[:, 1], 'C2': idx[:, 2], 'a': dense[:, 0], 'b': dense[:, 1]})
    df['genres'] = list(seq.numpy())
    y = (seq[:, 0] % 2 == 0).float().numpy()
    hist = dm.fit(df, y, batch_size=64, epochs=8, verbose=0)
    assert hist.history['loss'][-1] < hist.history['loss'][0]
    # what the DNN sees: [cat embeddings | var-len embedding] flattened, bit-exact gathers
    got = dm.apply(df, output_layers=['flatten_embeddings'], batch_size=100)
    assert np.array_equal(np.asarray(got), torch.cat(
        [torch.cat([e.detach().cpu()[idx[:, i]] for i, e in enumerate(emb.embeddings)], 1),
         R.var_len_embedding(seq.float(), L['emb_genres'].embeddings.detach().cpu()).reshape(n, -1)], 1).numpy())
    assert dm.predict(df).shape[0] == n
