# -*- coding:utf-8 -*-
"""GPU: the split-bf16 ("bf16 x 3") Dense tower of the fused DeepFM step (csrc/tower_x3.h, DT_STEP_TOWER_X3, selected by
`dnn_params['mfma_dtype'] = 'bf16x3'`): Dense128 / Dense64 / dH1 / dXn of deepnets.py:401-427 on v_mfma_f32_16x16x32_bf16
with every fp32 operand split into two bf16 halves.  Held to the SAME bars as the exact-fp32 kernels: logits within 1e-4 of
the float64 oracle, every gradient within 2e-4 of its tensor's largest entry (the mode's own rounding is ~2^-17 per
product: the measured figures are printed by bench.py's parity leg and recorded in DESIGN.md)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

X3 = {'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu', 'mfma_dtype': 'bf16x3'}


def _rel(a, b):
    b = b.detach().double().cpu()
    return (a.detach().double().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize('B,F,Nd,D,idt', [(256, 26, 13, 16, 'int32'), (100, 26, 13, 16, 'float32'), (37, 5, 3, 8, 'int32'),
                                          (64, 7, 0, 4, 'int32'), (513, 16, 2, 32, 'int32'), (1000, 26, 13, 16, 'int32')])
def test_x3_step_matches_oracle(dev, B, F, Nd, D, idt):
    import tests.test_fused_gpu as T
    from oracle import bridge, reference_layers as R
    from deeptables_amd import _lib
    dm, cats = T.build(F, Nd, D, vocab=30, dnn_params=dict(X3))
    plan = dm.fused_plan()
    assert plan is not None and plan.tower_flag == _lib.DT_STEP_TOWER_X3
    idx, dense, y = T.batch(cats, Nd, B)
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_logit, _ = bridge.oracle_forward(dm, idx, dense, training=True, weights=w)
    ref_loss = R.binary_crossentropy_from_logits(ref_logit, y.double())
    ref_loss.backward()
    dm.model.train()
    ins = [idx.to(getattr(torch, idt)).to(dev)] + ([dense.to(dev)] if Nd else [])
    loss, logit = dm.forward_backward(ins, y.to(dev))
    torch.cuda.synchronize()
    assert (logit.double().cpu() - ref_logit).abs().max().item() < 1e-4
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    L = dm.model.layers_by_name
    pairs = [(L['task_output'].kernel.grad, w['task_output'][0].grad),
             (L['dense_logit_dnn_nets'].kernel.grad, w['dense_logit_dnn_nets'].grad),
             (L['dnn_dense_2'].kernel.grad, w['dnn'][1][0].grad), (L['dnn_dense_2'].bias.grad, w['dnn'][1][1].grad),
             (L['dnn_dense_1'].kernel.grad, w['dnn'][0][0].grad), (L['dnn_dense_1'].bias.grad, w['dnn'][0][1].grad),
             (L['bn_concat_emb_dense'].gamma.grad, w['bn_concat_emb_dense'][0].grad),
             (L['bn_concat_emb_dense'].beta.grad, w['bn_concat_emb_dense'][1].grad),
             (L['linear_logit'].kernel.grad, w['linear_logit'].grad),
             (L['task_output'].bias.grad, w['task_output'][1].grad)]
    for i, (a, b) in enumerate(pairs):
        assert _rel(a, b) < 2e-4, f'dense grad {i}: {_rel(a, b)}'
    table = L['emb_categorical_vars_all'].tables[f'd{D}']
    ref_tg = torch.cat([t.grad for t in w['emb_categorical_vars_all']], 0)
    assert _rel(table.grad, ref_tg) < 2e-4


def test_x3_takes_narrow_towers_and_the_regression_task(dev):
    """zero-padded slabs (H1 = 100, H2 = 40) and the MSE loss block through the split-bf16 kernel"""
    import tests.test_fused_gpu as T
    from oracle import bridge, reference_layers as R
    dm, cats = T.build(26, 13, 16, vocab=30, task='regression',
                       dnn_params={'hidden_units': ((100, 0, False), (40, 0, False)), 'activation': 'relu',
                                   'mfma_dtype': 'bf16x3'})
    assert dm.fused_plan() is not None
    idx, dense, y = T.batch(cats, 13, 300)
    y = torch.randn(300, 1)
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_logit, _ = bridge.oracle_forward(dm, idx, dense, training=True, weights=w)
    ((ref_logit - y.double()) ** 2).mean().backward()
    dm.model.train()
    loss, logit = dm.forward_backward([idx.int().to(dev), dense.to(dev)], y.to(dev))
    assert (logit.double().cpu() - ref_logit).abs().max().item() < 1e-4
    L = dm.model.layers_by_name
    assert _rel(L['dnn_dense_1'].kernel.grad, w['dnn'][0][0].grad) < 2e-4
    assert _rel(L['dnn_dense_2'].kernel.grad, w['dnn'][1][0].grad) < 2e-4
    assert _rel(L['bn_concat_emb_dense'].gamma.grad, w['bn_concat_emb_dense'][0].grad) < 2e-4


@pytest.mark.parametrize('dist', ['uniform', 'zipf'])
def test_x3_headline_config_matches_oracle(dev, dist):
    """the benchmarked configuration (B = 8192, 26 x 1 M rows, in-step dedupe, Keras Adam) in split-bf16 mode against the
    float64 oracle, at the exact-fp32 bars of tests/test_headline_gpu.py"""
    import bench
    from oracle import headline
    from deeptables_amd.models import deepnets
    import tests.test_headline_gpu as H
    dm = bench.build_model(deepnets.DeepFM, dev, None, bench.D, {'dnn_params': dict(X3)})
    bench.N_BATCHES, keep = 2, bench.N_BATCHES
    try:
        batches = bench.make_batches(8192, dev, seed=1234, dist_kind=dist)
    finally:
        bench.N_BATCHES = keep
    res = headline.check_train_step(dm, batches[0])
    print('x3', dist, {k: res[k] for k in ('max_abs_logit_err', 'max_abs_logit', 'dense_grad_rel_err', 'rows_grad_rel_err',
                                           'adam_rows_rel_err', 'adam_dense_rel_err')})
    H._check(res)
    # ... and the timed path (optimizer inside the step's launches) against the oracle's Adam, two steps
    rv = headline.check_in_step_vs_oracle(dm, batches)
    assert rv['ok'], str(sorted(rv.items()))


def test_x3_rows_in_step_equals_the_separate_optimizer_step(dev):
    import bench
    from oracle import headline
    from deeptables_amd.models import deepnets
    dm = bench.build_model(deepnets.DeepFM, dev, None, bench.D, {'dnn_params': dict(X3)})
    bench.N_BATCHES, keep = 1, bench.N_BATCHES
    try:
        b = bench.make_batches(8192, dev, seed=4321, dist_kind='zipf')[0]
    finally:
        bench.N_BATCHES = keep
    res = headline.check_rows_in_step(dm, b)
    assert headline.rows_in_step_ok(res), str(sorted(res.items()))


# ---- the tower's plain-bf16 mode (north_star "1e-2 bf16"; dnn_params['mfma_dtype'] = 'bf16', DT_STEP_TOWER_BF16) -----------
@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
def test_bf16_tower_headline_config_holds_the_bf16_bar(dev, net):
    """B = 8192, 26 x 1 M rows: one bf16 product per operand pair in the tile kernel; logits within north_star's 1e-2 (of
    max(1, max |logit|)), gradients by relative L2 error < 1e-1 (the relu decisions are taken on 8-bit inputs too:
    oracle/headline.verdict, bf16='tower').  The gather stays bit-exact and the in-step optimizer agrees with the separate
    optimizer step of the same mode."""
    import bench
    from oracle import headline
    from deeptables_amd import _lib
    from deeptables_amd.models import deepnets
    params = dict(bench.MODEL_PARAMS.get(net) or {})
    params['dnn_params'] = {'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu', 'mfma_dtype': 'bf16'}
    dm = bench.build_model(getattr(deepnets, net), dev, None, bench.D, params)
    assert dm.fused_plan().tower_flag == _lib.DT_STEP_TOWER_BF16
    bench.N_BATCHES, keep = 1, bench.N_BATCHES
    try:
        b = bench.make_batches(8192, dev, seed=1234, dist_kind='uniform')[0]
        b2 = bench.make_batches(8192, dev, seed=4321, dist_kind='zipf')[0]
    finally:
        bench.N_BATCHES = keep
    res = headline.check_train_step(dm, b)
    print('bf16 tower', net, {k: res[k] for k in ('max_abs_logit_err', 'max_abs_logit', 'dense_grad_rel_err',
                                                  'dense_grad_l2_rel_err', 'rows_grad_rel_err', 'rows_grad_l2_rel_err')})
    good, rule = headline.verdict(res, bf16='tower')
    assert good, (rule, sorted(res.items()))
    assert res['max_abs_logit_err'] > 1e-6        # (the mode is really on: fp32-class arithmetic lands below this)
    ri = headline.check_rows_in_step(dm, b2)
    assert headline.rows_in_step_ok(ri), str(sorted(ri.items()))


# ---- CIN on split-bf16 matrix cores (csrc/cin_bf16.hip with NP parts; cin_params['mfma_dtype'] = 'bf16x3') -------------------
@pytest.mark.parametrize('B,F0,Hk,L,D,bias,act', [(5, 4, 4, 6, 3, False, 'relu'), (64, 26, 26, 128, 16, False, 'relu'),
                                                 (40, 26, 64, 128, 16, True, 'relu'), (9, 5, 7, 33, 8, True, 'linear'),
                                                 (20, 6, 100, 200, 4, False, 'relu'), (12, 5, 6, 40, 8, True, 'tanh'),
                                                 (300, 26, 64, 128, 16, False, 'relu'),
                                                 # B D >= 32768: the 256-row (eight-wave) forward / dgrad blocks and the
                                                 # wide wgrad blocks of large batches, Hk <= 32 and Hk <= 64 forms
                                                 (2100, 26, 64, 128, 16, False, 'relu'), (2060, 26, 26, 128, 16, True, 'relu')])
def test_cin_layer_split_bf16_holds_the_fp32_bar(dev, B, F0, Hk, L, D, bias, act):
    """CIN.call (layers.py:689-710) with three-part operands / six products in the forward, two parts / three products in
    the backward: outputs and all four gradients within 1e-4 of the float64 restatement — the bar of the exact-fp32 kernels
    (tests/test_kernels_gpu.py::test_cin_layer)"""
    import numpy as np
    from deeptables_amd import ops
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(B + F0 + Hk + L)

    def rnd(shape, scale=1.0):
        return torch.randn(shape, generator=g, dtype=torch.float64) * scale
    x0, xk = rnd((B, F0, D), 0.5), rnd((B, Hk, D), 0.5)
    W = rnd((F0 * Hk, L), 1.0 / np.sqrt(F0 * Hk))
    bv = rnd((L,), 0.1) if bias else None
    up = rnd((B, L, D))
    x0r, xkr, Wr = (t.clone().requires_grad_(True) for t in (x0, xk, W))
    bvr = bv.clone().requires_grad_(True) if bias else None
    y = torch.einsum('bid,bjd,ijl->bld', x0r, xkr, Wr.reshape(F0, Hk, L))
    if bias:
        y = y + bvr[None, :, None]
    if act == 'relu':
        # a unit within fp32 rounding of its kink may land on the other side of it (millions of units in the large cases:
        # it happens); no gradient is sent through those, so the comparison is about arithmetic, not about the kink
        up = torch.where(y.detach().abs() < 1e-5, torch.zeros_like(up), up)
    ref = R._activation(act)(y)
    (ref * up).sum().backward()
    x0d, xkd, Wd = (t.float().to(dev).requires_grad_(True) for t in (x0, xk, W))
    bd = bv.float().to(dev).requires_grad_(True) if bias else None
    out = ops.cin_layer(x0d, xkd, Wd, bd, act, 'bf16x3')
    (out * up.float().to(dev)).sum().backward()
    assert _rel(out, ref) < 1e-4, _rel(out, ref)
    assert _rel(x0d.grad, x0r.grad) < 1e-4 and _rel(xkd.grad, xkr.grad) < 1e-4
    assert _rel(Wd.grad, Wr.grad) < 1e-4
    if bias:
        assert _rel(bd.grad, bvr.grad) < 1e-4


def test_xdeepfm_config_split_bf16_matches_oracle(dev):
    """bench.py --model xDeepFM --cin bf16x3 at the size it is timed (B = 8192, CIN 3 x 128): the same figures and the same
    verdict rule as the exact-fp32 path (tests/test_headline_gpu.py::test_xdeepfm_config_matches_oracle)"""
    import bench
    from oracle import headline
    from deeptables_amd.models import deepnets
    import tests.test_headline_gpu as H
    params = dict(bench.MODEL_PARAMS['xDeepFM'])
    params['cin_params'] = dict(params['cin_params'], mfma_dtype='bf16x3')
    dm = bench.build_model(deepnets.xDeepFM, dev, None, bench.D, params)
    bench.N_BATCHES, keep = 1, bench.N_BATCHES
    try:
        batches = bench.make_batches(8192, dev, seed=1234, dist_kind='uniform')
    finally:
        bench.N_BATCHES = keep
    res = headline.check_train_step(dm, batches[0])
    print('cin x3', {k: res[k] for k in ('max_abs_logit_err', 'max_abs_logit', 'dense_grad_rel_err', 'dense_grad_l2_rel_err',
                                         'rows_grad_rel_err', 'rows_grad_l2_rel_err', 'relu_units_near_kink')})
    H._check_layer_path(res, n_dense=15)
