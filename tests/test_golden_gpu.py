# -*- coding:utf-8 -*-
"""GPU: the HIP path (through the C-ABI) against the COMMITTED golden fixtures (tests/golden/*.npz),
so parity does not depend on re-running the oracle on the GPU box.  fp32 tolerance 1e-4 (relative to
the tensor scale); gather bit-exact; model logits within 1e-4 absolute (north_star)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4


def load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name)).items()}


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def dv(a, dev, grad=False):
    t = torch.as_tensor(np.asarray(a, dtype=np.float32)).to(dev)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
def test_fm(dev, tag):
    from deeptables_amd import ops
    g = load(f'layer_fm_{tag}.npz')
    x = dv(g['x'], dev, True)
    y = ops.fm(x)
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL and rel(x.grad, g['gx']) < TOL


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
def test_products(dev, tag):
    from deeptables_amd import ops
    g = load(f'layer_ip_{tag}.npz')
    x = dv(g['x'], dev, True)
    y = ops.inner_product(x)
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL and rel(x.grad, g['gx']) < TOL
    for kt in ('mat', 'vec', 'num'):
        g = load(f'layer_op_{kt}_{tag}.npz')
        x, k = dv(g['x'], dev, True), dv(g['k'], dev, True)
        y = ops.outer_product(x, k, kt)
        (y * dv(g['up'], dev)).sum().backward()
        assert rel(y, g['y']) < TOL, kt
        assert rel(x.grad, g['gx']) < TOL and rel(k.grad, g['gk']) < TOL, kt


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
def test_cross(dev, tag):
    from deeptables_amd import ops
    g = load(f'layer_cross_{tag}.npz')
    x, w, b = dv(g['x'], dev, True), dv(g['w'], dev, True), dv(g['b'], dev, True)
    y = ops.cross(x, w, b)
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL and rel(x.grad, g['gx']) < TOL
    assert rel(w.grad, g['gw']) < TOL and rel(b.grad, g['gb']) < TOL


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
def test_cin_stack(dev, tag):
    from deeptables_amd import ops
    g = load(f'layer_cin_{tag}.npz')
    cls = [int(v) for v in g['cls']]
    x = dv(g['x'], dev, True)
    filters = [dv(g[f'f{i}'][0], dev, True) for i in range(len(cls))]
    hidden, outs = x, []
    for i, L in enumerate(cls):
        cur = ops.cin_layer(x, hidden, filters[i], None, 'relu')
        if i != len(cls) - 1:
            hidden, dc = cur[:, :L // 2], cur[:, L // 2:]
        else:
            dc = cur
        outs.append(dc)
    y = torch.cat(outs, 1).sum(-1)
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL and rel(x.grad, g['gx']) < TOL
    for i in range(len(cls)):
        assert rel(filters[i].grad, g[f'gf{i}'][0]) < TOL, f'filter {i}'


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
def test_mha_layer(dev, tag):
    from deeptables_amd.models import layers
    g = load(f'layer_mha_{tag}.npz')
    H = int(g['H'])
    x = dv(g['x'], dev, True)
    layer = layers.MultiheadAttention({'num_heads': H, 'dropout_rate': 0, 'use_residual': True})
    layer.build(tuple(x.shape))
    layer.to(dev)
    with torch.no_grad():
        for dn, k in ((layer.dense_Q, 'Q'), (layer.dense_K, 'K'), (layer.dense_V, 'V'), (layer.dense_residual, 'R')):
            dn.kernel.copy_(dv(g[f'k{k}'], dev))
            dn.bias.copy_(dv(g[f'b{k}'], dev))
        layer.batch_normalize.gamma.copy_(dv(g['gamma'], dev))
        layer.batch_normalize.beta.copy_(dv(g['beta'], dev))
    layer.train()
    y = layer(x)
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL and rel(x.grad, g['gx']) < 5e-4


def test_gather_bit_exact(dev):
    from deeptables_amd import ops
    g = load('layer_gather.npz')
    tables = [g[f't{i}'] for i in range(4)]
    vocabs = [t.shape[0] for t in tables]
    packed = torch.as_tensor(np.concatenate(tables, 0)).to(dev)
    offs = torch.as_tensor(np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)).to(dev)
    voc = torch.tensor(vocabs, dtype=torch.int32, device=dev)
    emb, _ = ops.embedding_lookup(torch.as_tensor(g['idx']).to(dev), packed, offs, voc)
    assert np.array_equal(emb.cpu().numpy(), g['emb'])
    emb2, _, _, _, _ = ops.embed_fm_linear(torch.as_tensor(g['idx']).to(dev), packed, offs, voc, None)
    assert np.array_equal(emb2.cpu().numpy(), g['emb'])


@pytest.mark.parametrize('name', ['DeepFM', 'xDeepFM', 'AutoInt', 'DCN'])
def test_model_logits_and_grads(dev, name):
    sys.path.insert(0, GOLD)
    import make_golden as MG
    from oracle import bridge
    from tests.test_oracle import unflatten
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    g = load(f'model_{name}.npz')
    w = unflatten(g, torch.float32)
    spec = dict(MG.MODEL_CONFIGS[name])
    D = spec.pop('D')
    conf = ModelConfig(fixed_embedding_dim=True, embeddings_output_dim=D, embedding_dropout=0, **spec)
    cats = [CategoricalColumn(f'C{i}', t.shape[0], D) for i, t in enumerate(w['emb_categorical_vars_all'])]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(13)])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    dm.build(dev)
    bridge.load_weights(dm, w)
    dm.model.train()
    dm.optimizer.zero_grad()
    logit = dm.model([torch.as_tensor(g['idx']).to(dev), torch.as_tensor(g['dense']).to(dev)])
    assert np.abs(logit.detach().cpu().numpy() - g['logit']).max() < 1e-4, name
    loss = dm._loss(logit, torch.as_tensor(g['y']).to(dev))
    assert abs(float(loss) - float(g['loss'])) < 1e-5
    loss.backward()
    assert rel(dm.model.layers_by_name['task_output'].kernel.grad, g['g_task_output_kernel']) < 2e-4
    emb = dm.model.layers_by_name['emb_categorical_vars_all']
    gt = emb.tables[f'd{D}'].grad[:cats[0].vocabulary_size]
    assert rel(gt, g['g_emb0']) < 2e-4


# ---------------------------------------------------------------------------------------------
# SURVEY §8 f3 layers against their committed fixtures
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag', ['tiny', 'medium'])
def test_afm_golden(dev, tag):
    from deeptables_amd import ops
    g = load(f'layer_afm_{tag}.npz')
    x, Wa, ba, pv, wo = [dv(g[k], dev, True) for k in ('x', 'Wa', 'ba', 'pv', 'wo')]
    y = ops.afm_pool(x, Wa, ba, pv, 'relu') @ wo
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL
    for t, k in ((x, 'gx'), (Wa, 'gWa'), (ba, 'gba'), (pv, 'gpv'), (wo, 'gwo')):
        assert rel(t.grad, g[k]) < 2 * TOL, k


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
@pytest.mark.parametrize('bt', ['field_interaction', 'field_each', 'field_all'])
def test_bilinear_golden(dev, tag, bt):
    from deeptables_amd import ops
    g = load(f'layer_bilinear_{bt}_{tag}.npz')
    x, W = dv(g['x'], dev, True), dv(g['W'], dev, True)
    y = ops.bilinear_interaction(x, W, bt)
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL and rel(x.grad, g['gx']) < TOL and rel(W.grad, g['gW']) < TOL


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
@pytest.mark.parametrize('op', ['mean', 'max'])
def test_senet_golden(dev, tag, op):
    from deeptables_amd import ops
    g = load(f'layer_senet_{op}_{tag}.npz')
    x, k1, b1, k2, b2 = [dv(g[k], dev, True) for k in ('x', 'k1', 'b1', 'k2', 'b2')]
    z = ops.field_pool(x, op)
    a = torch.relu(torch.relu(z @ k1 + b1) @ k2 + b2)
    y = ops.field_scale(x, a)
    (y * dv(g['up'], dev)).sum().backward()
    assert rel(y, g['y']) < TOL
    for t, k in ((x, 'gx'), (k1, 'gk1'), (b1, 'gb1'), (k2, 'gk2'), (b2, 'gb2')):
        assert rel(t.grad, g[k]) < 2 * TOL, k


@pytest.mark.parametrize('tag', ['default', 'odd'])
def test_fgcnn_golden(dev, tag):
    from deeptables_amd.models import layers
    g = load(f'layer_fgcnn_{tag}.npz')
    pool, nf = [int(v) for v in g['meta']]
    h, _, C, filters = g['ck'].shape
    B, F, D, _ = g['x'].shape
    layer = layers.FGCNN(filters=filters, kernel_height=h, new_filters=nf, pool_height=pool)
    layer.build((None, F, D, C))
    layer.to(dev)
    with torch.no_grad():
        layer.conv_kernel.copy_(dv(g['ck'], dev))
        layer.conv_bias.copy_(dv(g['cb'], dev))
        layer.dense_output.kernel.copy_(dv(g['dk'], dev))
        layer.dense_output.bias.copy_(dv(g['db'], dev))
    x = dv(g['x'], dev, True)
    pooled, newf = layer(x)
    ((pooled * dv(g['up_p'], dev)).sum() + (newf * dv(g['up_n'], dev)).sum()).backward()
    assert rel(pooled, g['pooled']) < TOL and rel(newf, g['newf']) < TOL
    assert rel(x.grad, g['gx']) < 2 * TOL and rel(layer.conv_kernel.grad, g['gck']) < 2 * TOL
    assert rel(layer.conv_bias.grad, g['gcb']) < 2 * TOL and rel(layer.dense_output.kernel.grad, g['gdk']) < 2 * TOL


def test_losses_golden(dev):
    from deeptables_amd.models import layers
    g = load('layer_losses.npz')
    bf = layers.BinaryFocalLoss()(dv(g['yb'], dev), dv(g['pb'], dev))
    assert abs(bf.item() - float(g['binary_focal'])) < 1e-6
    cf = layers.CategoricalFocalLoss(reduction='none')(dv(g['yc'], dev), dv(g['pc'], dev))
    assert rel(cf, g['categorical_focal']) < TOL
    ghm = layers.GHMCLoss(bins=10, momentum=0.75)
    l1 = ghm.calc(dv(g['z1'], dev), dv(g['yb'], dev))
    l2 = ghm.calc(dv(g['z2'], dev), dv(g['yb'], dev))
    assert abs(l1.item() - float(g['ghmc1'])) < 1e-5 and abs(l2.item() - float(g['ghmc2'])) < 1e-5
    assert rel(ghm.acc_sum, g['ghmc_acc']) < 1e-6
