# -*- coding:utf-8 -*-
"""CPU: host-side logic added in round 2 that needs no kernel — the per-lookup view of a segmented sparse gradient
(ops.SparseRowGrad.expanded), Keras' sample / class weighting of the loss (training.weighted_loss), the feed carrying
per-row weights through a shuffle (training.TableBatches), balanced class weights (DeepTable.get_class_weight)."""
import os

import numpy as np
import torch


def test_segmented_sparse_grad_expands_to_one_entry_per_lookup():
    from deeptables_amd.ops import SparseRowGrad
    ids = torch.tensor([5, 9, 5, 2, 9, 9, 7, -1, 2], dtype=torch.int64)
    vals = torch.arange(9 * 4, dtype=torch.float32).reshape(9, 4)
    # rows 5 (lookups 0, 2), 9 (1, 4, 5) and 2 (3, 8) are looked up several times -> segments in 2 regions of capacity 3
    regions, cap = 2, 3
    nseg = torch.tensor([2, 1], dtype=torch.int32)
    seg_row = torch.tensor([5, 2, 0, 9, 0, 0], dtype=torch.int64)
    seg_off = torch.tensor([0, 2, 0, 4, 0, 0], dtype=torch.int32)
    seg_cnt = torch.tensor([2, 2, 0, 3, 0, 0], dtype=torch.int32)
    seg_list = torch.tensor([0, 2, 3, 8, 1, 4, 5, 0, 0], dtype=torch.int32)
    rows = ids.clone()
    rows[[0, 2, 3, 8, 1, 4, 5]] = -1
    sg = SparseRowGrad(rows, vals, fields=-1, segments=(nseg, seg_row, seg_off, seg_cnt, seg_list, regions, cap))
    got_rows, got_vals = sg.expanded()
    assert torch.equal(got_rows, ids) and got_vals is vals
    assert torch.equal(rows[[0, 2]], torch.tensor([-1, -1]))        # the stored rows are untouched
    plain = SparseRowGrad(ids, vals)
    assert plain.expanded()[0] is ids
    empty = SparseRowGrad(ids, vals, segments=(torch.zeros(2, dtype=torch.int32), seg_row, seg_off, seg_cnt, seg_list, 2, 3))
    assert torch.equal(empty.expanded()[0], ids)


def test_weighted_loss_follows_keras_sample_weight_semantics():
    from deeptables_amd import training
    g = torch.Generator().manual_seed(0)
    z = torch.randn(17, 1, generator=g, dtype=torch.float64)
    y = (torch.rand(17, 1, generator=g) < 0.4).double()
    w = torch.rand(17, generator=g, dtype=torch.float64) * 3
    p = torch.sigmoid(z)
    per = -(y * torch.log(p) + (1 - y) * torch.log(1 - p)).reshape(-1)
    assert abs(float(training.weighted_loss('binary_crossentropy', z, y, w)) - float((per * w).sum() / 17)) < 1e-12
    # all-ones weights = the unweighted mean
    assert abs(float(training.weighted_loss('binary_crossentropy', z, y, torch.ones(17, dtype=torch.float64))) -
               float(training.bce_from_logits(z, y))) < 1e-12
    zc = torch.randn(9, 4, generator=g, dtype=torch.float64)
    yc = torch.eye(4, dtype=torch.float64)[torch.randint(0, 4, (9,), generator=g)]
    wc = torch.rand(9, generator=g, dtype=torch.float64)
    per_c = -(torch.log_softmax(zc, -1) * yc).sum(-1)
    assert abs(float(training.weighted_loss('categorical_crossentropy', zc, yc, wc)) - float((per_c * wc).sum() / 9)) < 1e-12
    yr = torch.randn(17, 1, generator=g, dtype=torch.float64)
    assert abs(float(training.weighted_loss('mse', z, yr, w)) - float((((z - yr) ** 2).reshape(-1) * w).sum() / 17)) < 1e-12


def test_feed_carries_row_weights_through_the_shuffle():
    from deeptables_amd import training
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    n = 50
    X = {'cat': np.arange(n * 2).reshape(n, 2) % 7, 'input_continuous_all': np.arange(n, dtype=np.float32).reshape(n, 1)}

    class XF:
        def __init__(self, d):
            self.d = d

        def __len__(self):
            return n

        def __getitem__(self, k):
            return self.d[k]
    y = (np.arange(n) % 3 == 0).astype(np.float32)
    w = np.arange(n, dtype=np.float32) + 0.5                        # weight i+0.5 belongs to the row whose dense value is i
    cats = [CategoricalColumn('C0', 7, 4), CategoricalColumn('C1', 7, 4)]
    conts = [ContinuousColumn('input_continuous_all', ['I0'])]
    tb = training.TableBatches(XF(X), y, cats, conts, 'cpu', 'binary', 2, sample_weight=w)
    assert tb.weighted
    seen = 0
    for ins, yb in tb.iterate(16, True, drop_remainder=False):
        dense = ins[-1].reshape(-1)
        assert yb.shape[1] == 2
        assert torch.allclose(yb[:, -1], dense + 0.5)               # the weight travelled with its row
        assert torch.allclose(yb[:, 0], (dense.long() % 3 == 0).float())
        seen += len(dense)
    assert seen == n
    assert not training.TableBatches(XF(X), y, cats, conts, 'cpu', 'binary', 2).weighted


def test_balanced_class_weights():
    from deeptables_amd.models.deeptable import DeepTable

    class Stub:
        classes_ = [0, 1, 2]
    y = np.array([0] * 6 + [1] * 3 + [2] * 1)
    cw = DeepTable.get_class_weight(Stub(), y)
    assert cw == {0: 10 / (3 * 6), 1: 10 / (3 * 3), 2: 10 / (3 * 1)}      # sklearn 'balanced': n / (classes * count)


def _cpu_model(nets, hidden, **extra):
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    conf = ModelConfig(nets=nets, fixed_embedding_dim=True, embeddings_output_dim=8, embedding_dropout=0,
                       dnn_params={'hidden_units': hidden, 'activation': 'relu'}, **extra)
    dm = DeepModel('binary', 2, conf, [CategoricalColumn(f'C{i}', 20 + i, 8) for i in range(6)],
                   [ContinuousColumn('input_continuous_all', ['a', 'b', 'c'])])
    dm.build('cpu')
    return dm


def test_fused_plan_eligibility_by_tower_shape():
    """which dnn_params the whole-step plans take (fused._tower_widths): two relu cells without dropout / batch norm whose
    widths fit the compiled 128 x 64 tile; everything else falls back to the per-layer kernels (plan None).  Host logic
    only: the plan is constructed on CPU, nothing is launched."""
    from deeptables_amd.fused import FusedDCN, FusedDeepFM, _tower_widths
    assert _tower_widths({'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu'}) == (128, 64)
    assert _tower_widths({'hidden_units': [[100, 0, False], [40, 0, False]]}) == (100, 40)
    for bad in ({'hidden_units': ((129, 0, False), (64, 0, False))}, {'hidden_units': ((128, 0, False), (65, 0, False))},
                {'hidden_units': ((64, 0, True), (32, 0, False))}, {'hidden_units': ((64, 0.2, False), (32, 0, False))},
                {'hidden_units': ((64, 0, False),)}, {'hidden_units': ((64, 0, False), (32, 0, False), (16, 0, False))},
                {'hidden_units': ((64, 0, False), (32, 0, False)), 'activation': 'tanh'},
                {'hidden_units': ((64, 0, False), (32, 0, False)), 'custom_dnn_fn': lambda x, p, c: x}, {'hidden_units': ()}):
        assert _tower_widths(bad) is None, bad
    assert isinstance(_cpu_model(['linear', 'fm_nets', 'dnn_nets'], ((100, 0, False), (40, 0, False))).fused_plan(), FusedDeepFM)
    assert isinstance(_cpu_model(['dcn_nets'], ((32, 0, False), (64, 0, False)), cross_params={'num_cross_layer': 3}).fused_plan(),
                      FusedDCN)
    assert _cpu_model(['linear', 'fm_nets', 'dnn_nets'], ((256, 0, False), (64, 0, False))).fused_plan() is None
    assert _cpu_model(['linear', 'fm_nets', 'dnn_nets'], ((64, 0, False), (1, 0, False))).fused_plan() is None    # no dense_logit layer
    assert _cpu_model(['linear', 'dnn_nets'], ((128, 0, False), (64, 0, False))).fused_plan() is None             # not DeepFM
    assert _cpu_model(['linear', 'fm_nets', 'dnn_nets'], ((128, 0, False), (64, 0, False)), stacking_op='concat').fused_plan() is None


def test_narrow_tower_parameters_are_views_of_zero_padded_slabs():
    """the plan moves W1 / b1 / W2 / b2 / w3 into [C,128] / [128] / [128,64] / [64] / [64] slabs of its flat parameter
    buffer: values kept, pads zero, the gradient views and the Adam moments laid out the same way"""
    dm = _cpu_model(['linear', 'fm_nets', 'dnn_nets'], ((100, 0, False), (40, 0, False)))
    L = dm.model.layers_by_name
    before = {n: p.detach().clone() for n, p in dm.model.named_parameters()}
    plan = dm.fused_plan()
    for n, p in dm.model.named_parameters():
        assert torch.equal(p.detach(), before[n]), n                      # moving the storage keeps every value
    C = 6 * 8 + 3
    d1, d2, dl = L['dnn_dense_1'], L['dnn_dense_2'], L['dense_logit_dnn_nets']
    assert tuple(d1.kernel.shape) == (C, 100) and d1.kernel.stride() == (128, 1)
    assert tuple(d2.kernel.shape) == (100, 40) and d2.kernel.stride() == (64, 1)
    o, flat = plan.off, plan.flat_params
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    for p, g in plan.grad_views:
        assert lo <= p.data_ptr() < hi and tuple(p.shape) == tuple(g.shape) and p.stride() == g.stride()
        assert (p.data_ptr() - lo) == (g.data_ptr() - plan.accum.data_ptr())          # same offset in both buffers
        st = dm.optimizer.state[id(p)]
        assert tuple(st['m'].shape) == tuple(p.shape) and st['m'].stride() == p.stride()
    W1 = flat[o['dW1']:o['dW1'] + C * 128].view(C, 128)
    W2 = flat[o['dW2']:o['dW2'] + 128 * 64].view(128, 64)
    assert torch.equal(W1[:, :100], d1.kernel.detach()) and W1[:, 100:].abs().sum() == 0
    assert torch.equal(W2[:100, :40], d2.kernel.detach()) and W2[100:].abs().sum() == 0 and W2[:, 40:].abs().sum() == 0
    assert flat[o['db1'] + 100:o['db1'] + 128].abs().sum() == 0 and flat[o['db2'] + 40:o['db2'] + 64].abs().sum() == 0
    assert torch.equal(flat[o['dw3']:o['dw3'] + 40], dl.kernel.detach().reshape(-1)) and flat[o['dw3'] + 40:o['dw3'] + 64].abs().sum() == 0
    with torch.no_grad():                         # writes through the parameter land in the slab (the kernels read the slab)
        d1.kernel[3, 7] = 42.0
    assert W1[3, 7].item() == 42.0


def test_segmented_sparse_grad_expands_for_random_segment_layouts():
    """property: whatever way duplicated rows are split into (region, segment) records — regions of different fill, segments
    in any order inside a region, members in any order inside a segment — `expanded()` restores the row id of every lookup"""
    from hypothesis import given, settings, strategies as st
    from deeptables_amd.ops import SparseRowGrad

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 40), st.integers(1, 4), st.integers(1, 6), st.integers(0, 2 ** 31 - 1))
    def check(n, regions, cap, seed):
        rng = np.random.RandomState(seed)
        rows = rng.randint(0, max(2, n // 2), size=n).astype(np.int64)          # many duplicates
        rows[rng.rand(n) < 0.1] = -1                                            # out-of-range lookups stay -1
        reported = rows.copy()
        nseg = np.zeros(regions, np.int32)
        seg_row = np.zeros(regions * cap, np.int64)
        seg_off = np.zeros(regions * cap, np.int32)
        seg_cnt = np.zeros(regions * cap, np.int32)
        seg_list = np.zeros((regions, n), np.int32)
        fill = np.zeros(regions, np.int64)
        for r in rng.permutation(np.unique(rows[rows >= 0])):
            members = np.flatnonzero(rows == r)
            if len(members) < 2:
                continue
            reg = int(rng.randint(regions))
            if nseg[reg] == cap:                                                # region full: the row stays per-lookup
                continue                                                        # (the kernels size cap so this cannot happen)
            k = reg * cap + nseg[reg]
            seg_row[k], seg_off[k], seg_cnt[k] = r, reg * n + fill[reg], len(members)
            seg_list[reg, fill[reg]:fill[reg] + len(members)] = rng.permutation(members)
            fill[reg] += len(members)
            nseg[reg] += 1
            reported[members] = -1
        T = torch.as_tensor
        sg = SparseRowGrad(T(reported), torch.zeros(n, 2), fields=-1,
                           segments=(T(nseg), T(seg_row), T(seg_off), T(seg_cnt), T(seg_list.reshape(-1)), regions, cap))
        got, _ = sg.expanded()
        assert np.array_equal(got.numpy(), rows)
        assert np.array_equal(sg.rows.numpy(), reported)                       # the stored gradient is not modified
    check()


def _run_bench(extra_args, env_extra, timeout=300):
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([_sys.executable, os.path.join(root, 'bench.py')] + extra_args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher must run TWO ranks (re-exec under torch.distributed.run),
    not one rank labelled 2 (reference shape: deeptables/tests/models/run_dt.py:35-44).  The probe mode joins a gloo group
    and leaves before any device is touched, so this runs on CPU."""
    import json
    r = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '0'], {'DT_BENCH_SPAWN_PROBE': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert lines, r.stdout + r.stderr
    assert json.loads(lines[-1]) == {'spawned_ranks': 2, 'world_size': 2}


def test_bench_gpus_flag_must_match_the_launcher():
    """inside a launcher (WORLD_SIZE set) a different --gpus is an error, not a mislabelled line"""
    r = _run_bench(['--gpus', '4'], {'DT_BENCH_SPAWN_PROBE': '1', 'WORLD_SIZE': '2', 'RANK': '0'}, timeout=120)
    assert r.returncode != 0
    assert 'WORLD_SIZE=2' in (r.stderr + r.stdout)


def test_model_peephole_defers_the_interacting_layers_normalisation_only_for_consumers_that_take_it():
    """functional.Model.__init__: an AutoInt interacting layer hands over its output with the BatchNormalization pending when
    that output has ONE consumer which applies it on load — the next interacting layer, or Flatten -> linear Dense(1)
    (deepnets.py:219-224, deepmodel.py:131-143) — and never otherwise (a second consumer, a model output, a wider Dense)."""
    import torch
    from deeptables_amd import functional as K
    from deeptables_amd.models import layers

    def stack(tail):
        inp = K.Input(shape=(6, 32), name='x')
        a = layers.MultiheadAttention(params={'num_heads': 4}, name='att_a')(inp)
        b = layers.MultiheadAttention(params={'num_heads': 4}, name='att_b')(a)
        return inp, a, b, tail(a, b)

    def deferred(model):
        return sorted(n.layer.name for n in model.nodes if id(n) in model._defer_norm)

    inp, a, b, out = stack(lambda a, b: K.Dense(1, name='out')(K.Flatten(name='fl')(b)))
    assert deferred(K.Model(inputs=[inp], outputs=out)) == ['att_a', 'att_b']
    inp, a, b, out = stack(lambda a, b: K.Dense(1, activation='relu', name='out')(K.Flatten(name='fl')(b)))
    assert deferred(K.Model(inputs=[inp], outputs=out)) == ['att_a']                       # a relu unit is not the linear head
    inp, a, b, out = stack(lambda a, b: K.Dense(3, name='out')(K.Flatten(name='fl')(b)))
    assert deferred(K.Model(inputs=[inp], outputs=out)) == ['att_a']                       # neither is a wider Dense
    inp, a, b, out = stack(lambda a, b: K.Add(name='add')([K.Dense(1, name='o1')(K.Flatten(name='f1')(b)),
                                                          K.Dense(1, name='o2')(K.Flatten(name='f2')(a))]))
    assert deferred(K.Model(inputs=[inp], outputs=out)) == ['att_b']                       # att_a's output has two consumers
    inp, a, b, out = stack(lambda a, b: b)
    assert deferred(K.Model(inputs=[inp], outputs=out)) == ['att_a']                       # a model output stays normalised
