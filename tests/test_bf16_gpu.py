# -*- coding:utf-8 -*-
"""GPU: the opt-in bf16-MFMA mode of the CIN layer (csrc/cin_bf16.hip; north_star "logits within ... 1e-2 bf16"):
forward and every gradient within 1e-2 (relative to the tensor's scale) of the float64 restatement of
CIN.call's per-layer op chain (layers.py:689-710), and xDeepFM logits within 1e-2 of the float64 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-2          # north_star tolerance of the bf16 mode


def rel(a, b):
    b = b.detach().double().cpu()
    return (a.detach().double().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize('B,F0,Hk,L,D,bias,act', [(64, 26, 26, 128, 16, False, 'relu'), (40, 26, 64, 128, 16, True, 'relu'),
                                                 (9, 5, 7, 33, 8, True, 'linear'), (20, 6, 100, 200, 4, False, 'relu'),
                                                 (16, 3, 2, 10, 10, False, 'tanh'), (130, 26, 64, 128, 16, False, 'relu')])
def test_cin_layer_bf16_within_1e2_of_float64(dev, B, F0, Hk, L, D, bias, act):
    from deeptables_amd import ops
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(B + F0 + Hk + L)
    x0 = torch.randn(B, F0, D, generator=g, dtype=torch.float64) * 0.5
    xk = torch.randn(B, Hk, D, generator=g, dtype=torch.float64) * 0.5
    W = torch.randn(F0 * Hk, L, generator=g, dtype=torch.float64) / np.sqrt(F0 * Hk)
    bv = torch.randn(L, generator=g, dtype=torch.float64) * 0.1 if bias else None
    up = torch.randn(B, L, D, generator=g, dtype=torch.float64)
    x0r, xkr, Wr = (t.clone().requires_grad_(True) for t in (x0, xk, W))
    bvr = bv.clone().requires_grad_(True) if bias else None
    yref = torch.einsum('bid,bjd,ijl->bld', x0r, xkr, Wr.reshape(F0, Hk, L))
    if bias:
        yref = yref + bvr[None, :, None]
    ref = R._activation(act)(yref)
    x0d, xkd, Wd = (t.float().to(dev).requires_grad_(True) for t in (x0, xk, W))
    bd = bv.float().to(dev).requires_grad_(True) if bias else None
    out = ops.cin_layer(x0d, xkd, Wd, bd, act, mfma_dtype='bf16')
    (out * up.float().to(dev)).sum().backward()
    assert rel(out, ref) < TOL
    if act == 'relu':
        # a pre-activation within the bf16 error of zero may land on the other side of the kink: the gradient is that
        # of the function the kernel evaluated, so the reference takes the relu mask from the kernel's own output
        mask = (out.detach().double().cpu() > 0).double()
        (yref * mask * up).sum().backward()
        assert float((mask != (yref.detach() > 0).double()).double().mean()) < 0.02      # and few elements differ
    else:
        (ref * up).sum().backward()
    assert rel(x0d.grad, x0r.grad) < TOL and rel(xkd.grad, xkr.grad) < TOL
    assert rel(Wd.grad, Wr.grad) < TOL
    if bias:
        assert rel(bd.grad, bvr.grad) < TOL
    # and it really is a different (lower-precision) path than the default exact-fp32 one
    out32 = ops.cin_layer(x0d.detach(), xkd.detach(), Wd.detach(), None if bd is None else bd.detach(), act)
    assert rel(out32, ref) < 1e-4


def test_xdeepfm_bf16_logits_within_1e2(dev):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel, deepnets
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    from oracle import bridge
    F, Nd, D, B = 26, 13, 16, 128
    functional.set_seed(2)
    conf = ModelConfig(nets=deepnets.xDeepFM, fixed_embedding_dim=True, embeddings_output_dim=D, embedding_dropout=0,
                       metrics=['AUC'],
                       cin_params={'cross_layer_size': (128, 128, 128), 'activation': 'relu', 'use_residual': False,
                                   'use_bias': False, 'direct': False, 'reduce_D': False, 'mfma_dtype': 'bf16'})
    cats = [CategoricalColumn(f'C{i}', 50 + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    dm.build()
    g = torch.Generator().manual_seed(1)
    idx = torch.stack([torch.randint(0, c.vocabulary_size, (B,), generator=g) for c in cats], 1)
    dense = torch.randn(B, Nd, generator=g)
    y = (torch.rand(B, 1, generator=g) < 0.25).float()
    ref_logit, _ = bridge.oracle_forward(dm, idx, dense, training=True)
    dm.model.train()
    loss, logit = dm.forward_backward([idx.int().to(dev), dense.to(dev)], y.to(dev))
    torch.cuda.synchronize()
    err = (logit.double().cpu() - ref_logit).abs().max().item()
    assert err < TOL, err
    assert torch.isfinite(loss).all()


def test_xdeepfm_reference_code_fixture_in_bf16_mode(dev):
    """The reference's OWN xDeepFM graph at D = 16 (tests/golden/reference_code_model_xdeepfm_d16.npz: deepmodel.py /
    deepnets.py:69-81 / layers.py:638-734 imported unmodified, float64) replayed through the drop-in API with
    cin_params['mfma_dtype'] = 'bf16': the bf16-MFMA CIN kernels against the reference code itself, not the oracle —
    logit and output within north_star's 1e-2; the float32 default of the same fixture holds 1e-4
    (tests/test_reference_models_gpu.py)."""
    import copy
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_reference_code import GOLDEN as GOLDEN_DIR, load_model_fixture
    from oracle import bridge          # only its weight loader
    from oracle.reference_layers import _leaves, _map_leaves
    meta, tensors, want = load_model_fixture(os.path.join(GOLDEN_DIR, 'reference_code_model_xdeepfm_d16.npz'))
    static = copy.deepcopy(meta['static'])
    static['config']['cin_params'] = dict(static['config']['cin_params'], mfma_dtype='bf16')
    dm, ids, dense = bridge.model_from_reference_fixture(static, tensors, dev)
    assert dm.config.cin_params['mfma_dtype'] == 'bf16'
    dm.model.train()
    logit = dm.model([ids.to(dev), dense.to(dev)])
    got = torch.cat([logit, dm._activate(logit)], -1).detach().double().cpu()
    err = (got - want).abs().max().item()
    assert err < TOL * max(1.0, want.abs().max().item()), err
    assert err > 1e-7          # and it is not the exact-fp32 path answering
    # the train step's gradients in that mode against autograd through the reference's graph: relative L2 (a relu unit
    # whose pre-activation lies within the bf16 error of zero takes the other derivative: rank-1 terms, see the layer test)
    gmeta, gt, gwant = load_model_fixture(os.path.join(GOLDEN_DIR, 'reference_code_modelgrad_xdeepfm_d16.npz'))
    gstatic = copy.deepcopy(gmeta['static'])
    gstatic['config']['cin_params'] = dict(gstatic['config']['cin_params'], mfma_dtype='bf16')
    from deeptables_amd.models import layers as dl
    keep, dl.DENSE_GRAD_MAX_ELEMS = dl.DENSE_GRAD_MAX_ELEMS, 1 << 22
    try:
        dm, ids, dense = bridge.model_from_reference_fixture(gstatic, gt, dev)
        dm.model.train()
        dm.optimizer.zero_grad()
        dm.forward_backward([ids.to(dev), dense.to(dev)], gt['y'].to(torch.float32).to(dev))
        torch.cuda.synchronize()
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = keep
    leaves = _leaves(gt['weights'])
    pieces = iter(torch.split(gwant, [t.numel() for t in leaves]))
    grads = _map_leaves(gt['weights'], lambda t: next(pieces).reshape(t.shape))
    num = den = 0.0
    for p, g in bridge.param_pairs(dm, grads):
        g = torch.as_tensor(g).double()
        num += (p.grad.detach().double().cpu().reshape(g.shape) - g).pow(2).sum().item()
        den += g.pow(2).sum().item()
    assert (num / den) ** 0.5 < 0.1, (num / den) ** 0.5
