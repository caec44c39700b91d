# -*- coding:utf-8 -*-
"""CPU checks of the oracle's f3 restatements (AFM, SENET, BilinearInteraction, FGCNN, VarLen embedding, losses)
against independent closed forms written with different primitives (numpy einsum / torch conv2d)."""
import itertools

import numpy as np
import torch

from oracle import reference_layers as R


def test_afm_closed_form():
    rng = np.random.default_rng(0)
    B, F, D, H = 7, 5, 6, 4
    x = rng.normal(size=(B, F, D))
    Wa, ba, pv, wo = rng.normal(size=(D, H)), rng.normal(size=H), rng.normal(size=(H, 1)), rng.normal(size=(D, 1))
    pairs = list(itertools.combinations(range(F), 2))
    bi = np.stack([x[:, i] * x[:, j] for i, j in pairs], 1)                  # [B,P,D]
    a = np.maximum(bi @ Wa + ba, 0)
    lg = (a @ pv)[..., 0]
    sc = np.exp(lg - lg.max(1, keepdims=True))
    sc /= sc.sum(1, keepdims=True)
    want = np.einsum('bp,bpd->bd', sc, bi) @ wo
    xs = [torch.tensor(x[:, i:i + 1]) for i in range(F)]
    got = R.afm(xs, torch.tensor(Wa), torch.tensor(ba), torch.tensor(pv), torch.tensor(wo))
    assert np.allclose(got.numpy(), want, atol=1e-12)


def test_bilinear_closed_form():
    rng = np.random.default_rng(1)
    B, F, D = 4, 5, 3
    x = rng.normal(size=(B, F, D))
    pairs = list(itertools.combinations(range(F), 2))
    for btype, nW in (('field_interaction', len(pairs)), ('field_each', F - 1), ('field_all', 1)):
        W = rng.normal(size=(nW, D, D))
        want = np.stack([np.einsum('bd,de->be', x[:, i], W[p if btype == 'field_interaction' else
                                                           (i if btype == 'field_each' else 0)]) * x[:, j]
                         for p, (i, j) in enumerate(pairs)], 1)
        got = R.bilinear_interaction(torch.tensor(x), [torch.tensor(W[k]) for k in range(nW)], btype)
        assert got.shape == (B, len(pairs), D)
        assert np.allclose(got.numpy(), want, atol=1e-12), btype


def test_senet_closed_form():
    rng = np.random.default_rng(2)
    B, F, D, R_ = 5, 6, 4, 2
    x = rng.normal(size=(B, F, D))
    k1, b1, k2, b2 = rng.normal(size=(F, R_)), rng.normal(size=R_), rng.normal(size=(R_, F)), rng.normal(size=F)
    for op in ('mean', 'max'):
        z = x.mean(-1) if op == 'mean' else x.max(-1)
        a = np.maximum(np.maximum(z @ k1 + b1, 0) @ k2 + b2, 0)
        want = x * a[:, :, None]
        got = R.senet(torch.tensor(x), (torch.tensor(k1), torch.tensor(b1)), (torch.tensor(k2), torch.tensor(b2)), op)
        assert np.allclose(got.numpy(), want, atol=1e-12)


def test_fgcnn_against_conv2d():
    """Independent route: torch conv2d/max_pool2d on NCHW with explicit TF-'same' padding."""
    g = torch.Generator().manual_seed(3)
    for F, D, C, filters, h, pool, nf in ((26, 16, 1, 14, 7, 2, 2), (9, 4, 3, 5, 4, 3, 1), (5, 6, 2, 4, 2, 2, 3)):
        B = 3
        x = torch.randn(B, F, D, C, generator=g, dtype=torch.float64)
        k = torch.randn(h, 1, C, filters, generator=g, dtype=torch.float64)
        b = torch.randn(filters, generator=g, dtype=torch.float64)
        Fp = -(-F // pool)
        dk = torch.randn(Fp * D * filters, F * D * nf, generator=g, dtype=torch.float64) * 0.1
        db = torch.randn(F * D * nf, generator=g, dtype=torch.float64)
        pooled, newf = R.fgcnn(x, k, b, dk, db, pool, nf)
        tot = h - 1
        xc = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (0, 0, tot // 2, tot - tot // 2))
        conv = torch.tanh(torch.nn.functional.conv2d(xc, k.permute(3, 2, 0, 1), b))
        ptot = max((Fp - 1) * pool + pool - F, 0)
        cp = torch.nn.functional.pad(conv, (0, 0, ptot // 2, ptot - ptot // 2), value=float('-inf'))
        want_pool = torch.nn.functional.max_pool2d(cp, (pool, 1), (pool, 1)).permute(0, 2, 3, 1)
        assert pooled.shape == (B, Fp, D, filters)
        assert torch.allclose(pooled, want_pool, atol=1e-12)
        want_new = torch.tanh(want_pool.reshape(B, -1) @ dk + db).reshape(B, F * nf, D)
        assert torch.allclose(newf, want_new, atol=1e-12)


def test_same_padding_rule():
    # TF SAME: out = ceil(n/s); pad_total = max((out-1)*s + k - n, 0); extra pad goes at the end
    assert R._same_pad_1d(26, 7, 1) == (26, 3, 3)
    assert R._same_pad_1d(26, 4, 1) == (26, 1, 2)
    assert R._same_pad_1d(13, 2, 2) == (7, 0, 1)
    assert R._same_pad_1d(26, 2, 2) == (13, 0, 0)
    assert R._same_pad_1d(10, 3, 3) == (4, 1, 1)


def test_var_len_embedding():
    table = torch.arange(20, dtype=torch.float64).reshape(5, 4)
    idx = torch.tensor([[0., 4., 2.], [1., 1., 3.]])
    out = R.var_len_embedding(idx, table)
    assert out.shape == (2, 1, 12)
    assert torch.equal(out[0, 0, 4:8], table[4]) and torch.equal(out[1, 0, 8:], table[3])


def test_losses_closed_form():
    rng = np.random.default_rng(4)
    B = 50
    y = (rng.random((B, 1)) < 0.4).astype(np.float64)
    p = rng.random((B, 1)).clip(1e-3, 1 - 1e-3)
    a, gm = .25, 2.
    want = -(a * (1 - p[y == 1]) ** gm * np.log(p[y == 1])).sum() / B - ((1 - a) * p[y == 0] ** gm * np.log(1 - p[y == 0])).sum() / B
    got = R.binary_focal_loss(torch.tensor(y), torch.tensor(p))
    assert abs(got.item() - want) < 1e-12
    C = 4
    yc = np.eye(C)[rng.integers(0, C, B)]
    pc = rng.random((B, C))
    pn = pc / pc.sum(1, keepdims=True)
    want = (a * (1 - pn) ** gm * (-yc * np.log(pn))).sum(1)
    got = R.categorical_focal_loss(torch.tensor(yc), torch.tensor(pc))
    assert np.allclose(got.numpy(), want, atol=1e-12)
    # GHM-C, momentum 0: weight of a sample = tot / (count of its bin) / (number of non-empty bins)
    z = rng.normal(size=(B, 1)) * 2
    gl = np.abs(1 / (1 + np.exp(-z)) - y)
    bins = 10
    bidx = np.minimum((gl * bins).astype(int), bins - 1)
    counts = np.bincount(bidx.ravel(), minlength=bins)
    wgt = B / counts[bidx] / (counts > 0).sum()
    bce = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    want = (bce * wgt).sum() / B
    got, _ = R.ghmc_loss(torch.tensor(z), torch.tensor(y), None, bins=bins, momentum=0)
    assert abs(got.item() - want) < 1e-9
    # momentum > 0 carries the histogram
    acc = torch.zeros(bins, dtype=torch.float64)
    l1, acc = R.ghmc_loss(torch.tensor(z), torch.tensor(y), acc, bins=bins, momentum=0.75)
    assert np.allclose(acc.numpy(), 0.25 * counts)
    l2, acc2 = R.ghmc_loss(torch.tensor(z), torch.tensor(y), acc, bins=bins, momentum=0.75)
    assert np.allclose(acc2.numpy(), np.where(counts > 0, 0.75 * 0.25 * counts + 0.25 * counts, 0))
    assert l2.item() < l1.item()


# ---------------------------------------------------------------------------------------------
# committed golden fixtures (tests/golden/make_golden.py --f3) re-derived by the oracle
# ---------------------------------------------------------------------------------------------
import os

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return {k: torch.as_tensor(np.asarray(v, dtype=np.float64)) for k, v in np.load(os.path.join(GOLD, name)).items()}


@pytest.mark.parametrize('tag', ['tiny', 'medium'])
def test_f3_golden(tag):
    g = _load(f'layer_afm_{tag}.npz')
    F = g['x'].shape[1]
    y = R.afm([g['x'][:, i:i + 1] for i in range(F)], g['Wa'], g['ba'], g['pv'], g['wo'], 'relu')
    assert torch.allclose(y, g['y'], atol=1e-12)
    for bt in ('field_interaction', 'field_each', 'field_all'):
        g = _load(f'layer_bilinear_{bt}_{tag}.npz')
        y = R.bilinear_interaction(g['x'], [g['W'][i] for i in range(g['W'].shape[0])], bt)
        assert torch.allclose(y, g['y'], atol=1e-12), bt
    for op in ('mean', 'max'):
        g = _load(f'layer_senet_{op}_{tag}.npz')
        y = R.senet(g['x'], (g['k1'], g['b1']), (g['k2'], g['b2']), op)
        assert torch.allclose(y, g['y'], atol=1e-12), op


def test_f3_golden_fgcnn_and_losses():
    for tag in ('default', 'odd'):
        g = _load(f'layer_fgcnn_{tag}.npz')
        pool, nf = [int(v) for v in g['meta']]
        pooled, newf = R.fgcnn(g['x'], g['ck'], g['cb'], g['dk'], g['db'], pool, nf)
        assert torch.allclose(pooled, g['pooled'], atol=1e-12) and torch.allclose(newf, g['newf'], atol=1e-12)
    g = _load('layer_losses.npz')
    assert abs(R.binary_focal_loss(g['yb'], g['pb']) - g['binary_focal']) < 1e-12
    assert torch.allclose(R.categorical_focal_loss(g['yc'], g['pc']), g['categorical_focal'], atol=1e-12)
    l1, acc = R.ghmc_loss(g['z1'], g['yb'], torch.zeros(10, dtype=torch.float64))
    l2, acc = R.ghmc_loss(g['z2'], g['yb'], acc)
    assert abs(l1 - g['ghmc1']) < 1e-12 and abs(l2 - g['ghmc2']) < 1e-12 and torch.allclose(acc, g['ghmc_acc'])
