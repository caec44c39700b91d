# -*- coding:utf-8 -*-
"""Parity of every HIP kernel (through the C-ABI, via deeptables_amd.ops) against the CPU oracle.

Tolerances (BASELINE.json north_star): embedding gather bit-exact; fp32 layer outputs and
gradients within 1e-4 relative to the tensor's scale (oracle evaluated in float64).
"""
import numpy as np
import pytest
import torch

from oracle import reference_layers as R

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = max(b.abs().max().item(), 1e-6)
    return (a - b).abs().max().item() / scale


def gen(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale)


def packed_tables(vocabs, D, g):
    tables = [(torch.rand(v, D, generator=g, dtype=torch.float64) * 0.1 - 0.05).float() for v in vocabs]
    offs = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    return tables, torch.cat(tables, 0), torch.from_numpy(offs), torch.tensor(vocabs, dtype=torch.int32)


@pytest.mark.parametrize('B,F,D,kind', [(5, 4, 3, 'f32'), (64, 26, 16, 'f32'), (64, 26, 16, 'i32'),
                                        (33, 7, 8, 'i32'), (17, 3, 10, 'f32'), (1, 1, 4, 'i32'),
                                        (257, 26, 32, 'f32')])
def test_embedding_gather_bit_exact(dev, B, F, D, kind):
    from deeptables_amd import ops
    g = gen(B * 100 + F)
    vocabs = [int(v) for v in torch.randint(3, 50, (F,), generator=g)]
    tables, packed, offs, voc = packed_tables(vocabs, D, g)
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], 1)
    idx_in = idx.float() if kind == 'f32' else idx.int()
    ref = torch.cat(R.multi_column_embedding(idx.float(), tables), dim=1)     # [B,F,D]
    emb, rows = ops.embedding_lookup(idx_in.to(dev), packed.to(dev), offs.to(dev), voc.to(dev))
    assert torch.equal(emb.cpu(), ref), 'gather must be bit-exact'
    assert torch.equal(rows.cpu(), idx + offs[None, :])


def test_embedding_oob_reads_zero_and_counts(dev):
    from deeptables_amd import ops
    g = gen(7)
    vocabs = [5, 6, 7]
    tables, packed, offs, voc = packed_tables(vocabs, 4, g)
    idx = torch.tensor([[0., 5., 2.], [4., 9., -1.], [1., 1.9, 6.]])   # 1.9 truncates to 1
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    emb, rows = ops.embedding_lookup(idx.to(dev), packed.to(dev), offs.to(dev), voc.to(dev), oob=oob)
    emb = emb.cpu()
    assert int(oob.item()) == 2
    assert torch.equal(emb[1, 1], torch.zeros(4)) and torch.equal(emb[1, 2], torch.zeros(4))
    assert torch.equal(emb[2, 1], tables[1][1])
    assert rows.cpu()[1, 1].item() == -1


def test_embedding_dense_grad_matches_oracle(dev):
    from deeptables_amd import ops
    g = gen(11)
    vocabs = [4, 9, 6]
    D = 8
    tables, packed, offs, voc = packed_tables(vocabs, D, g)
    B = 40
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], 1)
    up = rnd((B, 3, D), g).float()
    tref = [t.clone().double().requires_grad_(True) for t in tables]
    ref = torch.cat(R.multi_column_embedding(idx.float(), tref), dim=1)
    (ref * up.double()).sum().backward()
    gref = torch.cat([t.grad for t in tref], 0)
    p = packed.to(dev).requires_grad_(True)
    emb, _ = ops.embedding_lookup(idx.float().to(dev), p, offs.to(dev), voc.to(dev), dense_grad=True)
    (emb * up.to(dev)).sum().backward()
    assert rel_err(p.grad, gref) < TOL


@pytest.mark.parametrize('B,F,D', [(5, 4, 3), (64, 26, 16), (130, 26, 32), (9, 2, 4), (31, 39, 8), (3, 5, 10)])
def test_fm(dev, B, F, D):
    from deeptables_amd import ops
    g = gen(B + F + D)
    x = rnd((B, F, D), g)
    up = rnd((B, 1), g)
    xr = x.clone().requires_grad_(True)
    ref = R.fm(xr)
    (ref * up).sum().backward()
    xd = x.float().to(dev).requires_grad_(True)
    out = ops.fm(xd)
    (out * up.float().to(dev)).sum().backward()
    assert out.shape == (B, 1)
    assert rel_err(out, ref) < TOL
    assert rel_err(xd.grad, xr.grad) < TOL


@pytest.mark.parametrize('B,F,D,Nd,kind', [(64, 26, 16, 13, 'f32'), (37, 5, 8, 0, 'i32'), (16, 4, 6, 3, 'f32')])
def test_fused_embed_fm_linear(dev, B, F, D, Nd, kind):
    from deeptables_amd import ops
    g = gen(B * 7 + F)
    vocabs = [int(v) for v in torch.randint(3, 40, (F,), generator=g)]
    tables, packed, offs, voc = packed_tables(vocabs, D, g)
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], 1)
    dense = rnd((B, Nd), g).float() if Nd else None
    tref = [t.clone().double().requires_grad_(True) for t in tables]
    embs = R.multi_column_embedding(idx.float(), tref)
    E = torch.cat(embs, dim=1)
    fm_ref = R.fm(E)
    fsum_ref = E.sum(-1)
    flat = E.reshape(B, -1)
    concat_ref = torch.cat([flat, dense.double()], -1) if Nd else flat
    u1, u2, u3, u4 = rnd((B, F, D), g), rnd(concat_ref.shape, g), rnd((B, F), g), rnd((B, 1), g)
    ((E * u1).sum() + (concat_ref * u2).sum() + (fsum_ref * u3).sum() + (fm_ref * u4).sum()).backward()
    gref = torch.cat([t.grad for t in tref], 0)

    p = packed.to(dev).requires_grad_(True)
    idx_in = (idx.float() if kind == 'f32' else idx.int()).to(dev)
    emb, concat, fsum, fmo, rows = ops.embed_fm_linear(idx_in, p, offs.to(dev), voc.to(dev),
                                                       None if dense is None else dense.to(dev),
                                                       dense_grad=True)
    assert torch.equal(emb.detach().cpu(), E.detach().float())
    assert torch.equal(concat.detach().cpu(), concat_ref.detach().float())
    assert rel_err(fsum, fsum_ref) < TOL and rel_err(fmo, fm_ref) < TOL
    f32 = lambda t: t.float().to(dev)
    ((emb * f32(u1)).sum() + (concat * f32(u2)).sum() + (fsum * f32(u3)).sum() + (fmo * f32(u4)).sum()).backward()
    assert rel_err(p.grad, gref) < TOL


@pytest.mark.parametrize('N,C', [(64, 7), (8192, 429), (1000, 32), (3, 5), (26 * 50, 32)])
def test_batchnorm_train(dev, N, C):
    from deeptables_amd import ops
    g = gen(N + C)
    x = rnd((N, C), g) * 2.0 + 3.0      # non-zero mean: exercises the shifted-variance path
    gamma, beta = rnd((C,), g), rnd((C,), g)
    up = rnd((N, C), g)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    mm0, mv0 = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    ref, nm, nv = R.keras_batchnorm(xr, gr, br, mm0, mv0, training=True)
    (ref * up).sum().backward()
    xd = x.float().to(dev).requires_grad_(True)
    gd = gamma.float().to(dev).requires_grad_(True)
    bd = beta.float().to(dev).requires_grad_(True)
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y = ops.batchnorm_train(xd, gd, bd, mm, mv, 1e-3, 0.99)
    (y * up.float().to(dev)).sum().backward()
    assert rel_err(y, ref) < TOL
    assert rel_err(mm, nm) < TOL and rel_err(mv, nv) < TOL
    assert rel_err(xd.grad, xr.grad) < 5e-4
    assert rel_err(gd.grad, gr.grad) < TOL and rel_err(bd.grad, br.grad) < TOL
    yi = ops.batchnorm_infer(xd.detach(), gd.detach(), bd.detach(), mm, mv, 1e-3)
    refi, _, _ = R.keras_batchnorm(x, gamma, beta, nm, nv, training=False)
    assert rel_err(yi, refi) < TOL


@pytest.mark.parametrize('B,C,L', [(5, 7, 2), (64, 429, 6), (300, 429, 4), (17, 64, 1), (9, 130, 3), (4, 1000, 2)])
def test_cross(dev, B, C, L):
    from deeptables_amd import ops
    g = gen(B + C + L)
    x = rnd((B, C), g, 0.5)
    w = rnd((L, C), g, 1.0 / np.sqrt(C))
    b = rnd((L, C), g, 0.1)
    up = rnd((B, C), g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = R.cross(xr, [wr[i].unsqueeze(1) for i in range(L)], [br[i].unsqueeze(1) for i in range(L)])
    (ref * up).sum().backward()
    xd, wd, bd = (t.float().to(dev).requires_grad_(True) for t in (x, w, b))
    out = ops.cross(xd, wd, bd)
    (out * up.float().to(dev)).sum().backward()
    assert rel_err(out, ref) < TOL
    assert rel_err(xd.grad, xr.grad) < TOL
    assert rel_err(wd.grad, wr.grad) < TOL and rel_err(bd.grad, br.grad) < TOL


@pytest.mark.parametrize('B,F,D', [(5, 4, 3), (64, 26, 16), (70, 5, 8), (3, 2, 4)])
def test_inner_product(dev, B, F, D):
    from deeptables_amd import ops
    g = gen(B + F + D)
    x = rnd((B, F, D), g)
    P = F * (F - 1) // 2
    up = rnd((B, P), g)
    xr = x.clone().requires_grad_(True)
    ref = R.inner_product([xr[:, i:i + 1] for i in range(F)])
    (ref * up).sum().backward()
    xd = x.float().to(dev).requires_grad_(True)
    out = ops.inner_product(xd)
    (out * up.float().to(dev)).sum().backward()
    assert rel_err(out, ref) < TOL and rel_err(xd.grad, xr.grad) < TOL


@pytest.mark.parametrize('kt', ['mat', 'vec', 'num'])
@pytest.mark.parametrize('B,F,D', [(5, 4, 3), (64, 26, 16), (130, 6, 8), (1000, 7, 16), (13, 3, 16)])
def test_outer_product(dev, B, F, D, kt):
    from deeptables_amd import ops
    g = gen(B + F + D)
    x = rnd((B, F, D), g)
    P = F * (F - 1) // 2
    kshape = {'mat': (D, P, D), 'vec': (P, D), 'num': (P, 1)}[kt]
    k = rnd(kshape, g, 0.3)
    up = rnd((B, P), g)
    xr, kr = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    ref = R.outer_product([xr[:, i:i + 1] for i in range(F)], kr, kt)
    (ref * up).sum().backward()
    xd, kd = x.float().to(dev).requires_grad_(True), k.float().to(dev).requires_grad_(True)
    out = ops.outer_product(xd, kd, kt)
    (out * up.float().to(dev)).sum().backward()
    assert rel_err(out, ref) < TOL
    assert rel_err(xd.grad, xr.grad) < TOL and rel_err(kd.grad, kr.grad) < TOL


@pytest.mark.parametrize('B,F0,Hk,L,D,bias,act', [(5, 4, 4, 6, 3, False, 'relu'), (64, 26, 26, 128, 16, False, 'relu'),
                                                 (40, 26, 64, 128, 16, True, 'relu'), (9, 5, 7, 33, 8, True, 'linear'),
                                                 (20, 6, 100, 200, 4, False, 'relu'), (16, 3, 2, 10, 10, False, 'relu'),
                                                 (12, 5, 6, 40, 8, True, 'tanh'), (12, 5, 6, 40, 8, True, 'sigmoid'),
                                                 (12, 5, 6, 40, 8, False, 'elu'), (12, 5, 6, 40, 8, True, 'selu'),
                                                 (12, 5, 6, 40, 8, True, 'softplus'), (12, 5, 6, 40, 8, False, 'softsign'),
                                                 (12, 5, 6, 40, 8, True, 'exponential')])
def test_cin_layer(dev, B, F0, Hk, L, D, bias, act):
    from deeptables_amd import ops
    from oracle import closed_form as C
    g = gen(B + F0 + Hk + L)
    x0, xk = rnd((B, F0, D), g, 0.5), rnd((B, Hk, D), g, 0.5)
    W = rnd((F0 * Hk, L), g, 1.0 / np.sqrt(F0 * Hk))
    bv = rnd((L,), g, 0.1) if bias else None
    up = rnd((B, L, D), g)
    x0r, xkr, Wr = (t.clone().requires_grad_(True) for t in (x0, xk, W))
    bvr = bv.clone().requires_grad_(True) if bias else None
    y = torch.einsum('bid,bjd,ijl->bld', x0r, xkr, Wr.reshape(F0, Hk, L))
    if bias:
        y = y + bvr[None, :, None]
    from oracle import reference_layers as R
    ref = R._activation(act)(y)        # keras `Activation(name)`, layers.py:709
    if act in ('relu', 'linear'):
        np.testing.assert_allclose(ref.detach().numpy(),
                                   C.cin_layer(x0.numpy(), xk.numpy(), W.numpy(), None if bv is None else bv.numpy(),
                                               act == 'relu'), atol=1e-10)
    (ref * up).sum().backward()
    x0d, xkd, Wd = (t.float().to(dev).requires_grad_(True) for t in (x0, xk, W))
    bd = bv.float().to(dev).requires_grad_(True) if bias else None
    out = ops.cin_layer(x0d, xkd, Wd, bd, act)
    (out * up.float().to(dev)).sum().backward()
    assert rel_err(out, ref) < TOL
    assert rel_err(x0d.grad, x0r.grad) < TOL and rel_err(xkd.grad, xkr.grad) < TOL
    assert rel_err(Wd.grad, Wr.grad) < TOL
    if bias:
        assert rel_err(bd.grad, bvr.grad) < TOL


def test_cin_layer_strided_hidden(dev):
    """direct=False feeds the [:, :L/2] channel slice of the previous output (layers.py:715)."""
    from deeptables_amd import ops
    g = gen(5)
    B, F0, L, D = 12, 5, 8, 4
    x0 = rnd((B, F0, D), g).float().to(dev)
    prev = rnd((B, L, D), g).float().to(dev)
    W = rnd((F0 * (L // 2), 6), g, 0.3).float().to(dev)
    a = ops.cin_layer(x0, prev[:, :L // 2], W, None, 'relu')
    b = ops.cin_layer(x0, prev[:, :L // 2].contiguous(), W, None, 'relu')
    assert torch.equal(a, b)


@pytest.mark.parametrize('B,F,D,H', [(5, 4, 8, 2), (64, 26, 32, 4), (33, 26, 16, 1), (7, 3, 6, 2), (9, 5, 10, 1), (2, 40, 32, 4)])
def test_mha_core(dev, B, F, D, H):
    from deeptables_amd import ops
    from oracle import closed_form as C
    g = gen(B + F + D + H)
    q, k, v = (torch.relu(rnd((B, F, D), g)) for _ in range(3))
    up = rnd((B, F, D), g)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    hs = D // H
    Q_ = torch.cat(torch.split(qr, hs, dim=2), dim=0)
    K_ = torch.cat(torch.split(kr, hs, dim=2), dim=0)
    V_ = torch.cat(torch.split(vr, hs, dim=2), dim=0)
    wts = torch.softmax(torch.matmul(Q_, K_.transpose(1, 2)) / (hs ** 0.5), dim=-1)
    ref = torch.cat(torch.split(torch.matmul(wts, V_), B, dim=0), dim=2)
    np.testing.assert_allclose(ref.detach().numpy(), C.mha_core(q.numpy(), k.numpy(), v.numpy(), H), atol=1e-10)
    (ref * up).sum().backward()
    qd, kd, vd = (t.float().to(dev).requires_grad_(True) for t in (q, k, v))
    out = ops.mha_core(qd, kd, vd, H)
    (out * up.float().to(dev)).sum().backward()
    assert rel_err(out, ref) < TOL
    assert rel_err(qd.grad, qr.grad) < TOL and rel_err(kd.grad, kr.grad) < TOL and rel_err(vd.grad, vr.grad) < TOL


def test_cpu_tensor_is_rejected_loudly():
    from deeptables_amd import ops
    from deeptables_amd._lib import DtHipError
    with pytest.raises(DtHipError):
        ops.fm(torch.zeros(2, 3, 4))


@pytest.mark.parametrize('N,K,M,act,bias', [(64, 429, 128, 'relu', True), (300, 128, 64, 'relu', True), (8192, 64, 1, None, False),
                                            (1000, 39, 1, None, True), (26 * 37, 32, 32, 'relu', True), (50, 7, 5, None, True),
                                            (129, 493, 1, 'relu', True), (70, 33, 40, 'relu', False), (5, 3, 2, None, True)])
def test_dense(dev, N, K, M, act, bias):
    from deeptables_amd import ops
    g = gen(N + K + M)
    x = rnd((N, K), g)
    W = rnd((K, M), g, 1.0 / np.sqrt(K))
    b = rnd((M,), g, 0.3) if bias else None
    up = rnd((N, M), g)
    xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    ref = xr @ Wr + (br if bias else 0)
    if act == 'relu':
        ref = torch.relu(ref)
    (ref * up).sum().backward()
    xd, Wd = x.float().to(dev).requires_grad_(True), W.float().to(dev).requires_grad_(True)
    bd = b.float().to(dev).requires_grad_(True) if bias else None
    out = ops.dense(xd, Wd, bd, act)
    (out * up.float().to(dev)).sum().backward()
    assert rel_err(out, ref) < TOL
    assert rel_err(xd.grad, xr.grad) < TOL and rel_err(Wd.grad, Wr.grad) < TOL
    if bias:
        assert rel_err(bd.grad, br.grad) < TOL


@pytest.mark.parametrize('B,L,D,half', [(37, 128, 16, 64), (5, 10, 8, 0), (300, 7, 4, 3), (1, 2, 32, 1), (2049, 128, 16, 0)])
def test_cin_split_pool_matches_torch(dev, B, L, D, half):
    """ops.cin_split_pool (dt_cin_pool / dt_cin_pool_bwd): the `direct=False` bookkeeping of CIN.call (layers.py:713-721, :726)
    — hidden half as a view, the rest pooled over D; the backward assembles the layer's gradient in one pass — against the
    same thing written with torch slicing / sum"""
    from deeptables_amd import ops
    g = torch.Generator().manual_seed(B + L)
    y = torch.randn(B, L, D, generator=g).to(dev)
    gh = torch.randn(B, half, D, generator=g).to(dev)
    gp = torch.randn(B, L - half, generator=g).to(dev)
    y1 = y.clone().requires_grad_(True)
    h1, p1 = ops.cin_split_pool(y1, half)
    y2 = y.clone().requires_grad_(True)
    h2, p2 = y2[:, :half], y2[:, half:].sum(-1)
    assert torch.equal(h1, h2)
    assert (p1 - p2).abs().max().item() <= 1e-5 * max(1.0, p2.abs().max().item())
    if half:
        (h1 * gh).sum().backward(retain_graph=True)
        (h2 * gh).sum().backward(retain_graph=True)
    (p1 * gp).sum().backward()
    (p2 * gp).sum().backward()
    assert torch.allclose(y1.grad, y2.grad, rtol=0, atol=1e-6)
    # only one of the two outputs used: the other half of the gradient is zero
    y3 = y.clone().requires_grad_(True)
    ops.cin_split_pool(y3, half)[1].sum().backward()
    assert torch.equal(y3.grad[:, :half], torch.zeros_like(y3.grad[:, :half]))
    assert torch.equal(y3.grad[:, half:], torch.ones_like(y3.grad[:, half:]))
