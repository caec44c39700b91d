# -*- coding:utf-8 -*-
"""Edge cases through the C-ABI wrappers: empty and single-row batches, ids on the vocabulary boundary, ragged sizes
that straddle every tile width, the BASELINE full batch (size-independent properties: linearity, permutation
equivariance, gather idempotence), bad arguments reported as errors (never a silent fallback)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_empty_batches_are_no_ops(dev):
    from deeptables_amd import ops
    F, D = 5, 8
    x = torch.zeros(0, F, D, device=dev, requires_grad=True)
    assert ops.fm(x).shape == (0, 1)
    assert ops.inner_product(x).shape == (0, 10)
    assert ops.bilinear_interaction(x, torch.zeros(10, D, D, device=dev)).shape == (0, 10, D)
    assert ops.afm_pool(x, torch.zeros(D, 4, device=dev), None, torch.zeros(4, 1, device=dev)).shape == (0, D)
    assert ops.field_pool(x).shape == (0, F)
    table = torch.randn(20, D, device=dev)
    off = torch.arange(F, device=dev) * 4
    voc = torch.full((F,), 4, dtype=torch.int32, device=dev)
    emb, rows = ops.embedding_lookup(torch.zeros(0, F, dtype=torch.int32, device=dev), table, off, voc)
    assert emb.shape == (0, F, D) and rows.shape == (0, F)
    ops.fm(x).sum().backward()
    assert x.grad.shape == (0, F, D)


@pytest.mark.parametrize('B', [1, 2, 15, 16, 17, 31, 33, 63, 65, 255, 257])
def test_ragged_batches_straddle_every_tile(dev, B):
    """B around the 16/32/64-row tiles of the MFMA / wave-per-row kernels: same result as row-by-row evaluation."""
    from deeptables_amd import ops
    g = torch.Generator().manual_seed(B)
    F, D = 7, 16
    P = F * (F - 1) // 2
    x = torch.randn(B, F, D, generator=g).to(dev)
    W = (torch.randn(P, D, D, generator=g) * 0.3).to(dev)
    K = (torch.randn(D, P, D, generator=g) * 0.3).to(dev)
    full_b = ops.bilinear_interaction(x, W)
    full_o = ops.outer_product(x, K, 'mat')
    full_f = ops.fm(x)
    last = x[B - 1:B].contiguous()
    assert torch.allclose(ops.bilinear_interaction(last, W), full_b[B - 1:], atol=1e-5)
    assert torch.allclose(ops.outer_product(last, K, 'mat'), full_o[B - 1:], atol=1e-5)
    assert torch.allclose(ops.fm(last), full_f[B - 1:], atol=1e-5)
    W1 = (torch.randn(F * D, 24, generator=g) * 0.1).to(dev)
    y = ops.dense(x.reshape(B, -1), W1, None, 'relu')
    assert torch.allclose(y[B - 1:], ops.dense(last.reshape(1, -1), W1, None, 'relu'), atol=1e-5)


def test_boundary_ids_and_float_truncation(dev):
    """ids 0 and vocab-1 hit the first/last row of each field's range; vocab and -1 are out of range; float ids
    truncate toward zero like keras.ops.cast (3.999 -> 3, -0.5 -> 0)."""
    from deeptables_amd import ops
    vocabs = [3, 5, 2]
    D = 4
    table = torch.arange(sum(vocabs) * D, dtype=torch.float32, device=dev).reshape(-1, D)
    off = torch.tensor([0, 3, 8], device=dev)
    voc = torch.tensor(vocabs, dtype=torch.int32, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    idx = torch.tensor([[0, 0, 0], [2, 4, 1], [3, 5, 2], [-1, -1, -1]], dtype=torch.int32, device=dev)
    emb, rows = ops.embedding_lookup(idx, table, off, voc, oob=oob)
    assert rows.tolist() == [[0, 3, 8], [2, 7, 9], [-1, -1, -1], [-1, -1, -1]]
    assert torch.equal(emb[1, 1], table[7]) and float(emb[2:].abs().sum()) == 0.0 and int(oob) == 6
    fidx = torch.tensor([[2.999, 4.5, 1.0], [-0.5, 0.25, 0.999]], device=dev)
    _, rows_f = ops.embedding_lookup(fidx, table, off, voc)
    assert rows_f.tolist() == [[2, 7, 9], [0, 3, 8]]


def test_full_baseline_batch_properties(dev):
    """BASELINE size (B=8192, F=26, D=16): FM is permutation-invariant over fields and even in x, InnerProduct is
    bilinear (scaling one field scales its pairs), a gather of a gather's rows is idempotent."""
    from deeptables_amd import ops
    g = torch.Generator().manual_seed(0)
    B, F, D = 8192, 26, 16
    x = (torch.randn(B, F, D, generator=g) * 0.3).to(dev)
    fm = ops.fm(x)
    perm = torch.randperm(F, generator=g).to(dev)
    assert torch.allclose(ops.fm(x[:, perm].contiguous()), fm, atol=2e-4)
    assert torch.allclose(ops.fm(-x), fm, atol=1e-6)
    ip = ops.inner_product(x)
    x2 = x.clone()
    x2[:, 0] *= 2.0
    ip2 = ops.inner_product(x2)
    assert torch.allclose(ip2[:, :F - 1], 2.0 * ip[:, :F - 1], atol=1e-4) and torch.allclose(ip2[:, F - 1:], ip[:, F - 1:])
    vocab = 1000
    table = torch.randn(F * vocab, D, generator=g).to(dev)
    off = (torch.arange(F) * vocab).to(dev)
    voc = torch.full((F,), vocab, dtype=torch.int32, device=dev)
    idx = torch.randint(0, vocab, (B, F), generator=g, dtype=torch.int32).to(dev)
    emb, rows = ops.embedding_lookup(idx, table, off, voc)
    assert torch.equal(emb.reshape(-1, D), table[rows.reshape(-1)])            # bit-exact
    emb2, rows2 = ops.embedding_lookup((rows - off[None, :]).int(), table, off, voc)
    assert torch.equal(emb2, emb) and torch.equal(rows2, rows)


def test_bad_arguments_raise(dev):
    from deeptables_amd import ops, _lib
    x = torch.randn(4, 3, 8, device=dev)
    with pytest.raises(ValueError):
        ops.bilinear_interaction(x, torch.zeros(2, 8, 8, device=dev))          # wrong number of matrices
    with pytest.raises(_lib.DtHipError):
        ops.afm_pool(x, torch.randn(8, 65, device=dev), None, torch.randn(65, 1, device=dev))   # H > 64
    with pytest.raises((_lib.DtHipError, RuntimeError)):
        ops.fm(torch.randn(4, 3, 8))                                           # CPU tensor: no fallback
    with pytest.raises(_lib.DtHipError):
        _lib.check(_lib.lib().dt_adam_rows_step(None, None, None, None, None, 5, 16, 0, None, 0, None, 0.0, 0.9,
                                                0.999, 1e-7, None, None, None, None, None, 0, 0, 1e-3, None), 'x')
    assert b'null pointer' in _lib.lib().dt_last_error() or 'null' in str(_lib.lib().dt_last_error())
