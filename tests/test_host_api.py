# -*- coding:utf-8 -*-
"""CPU: host-side logic of the drop-in API (no kernel launches) and the C-ABI library surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C-ABI ------------------------------------------------------------------------------------
def header_functions():
    text = open(os.path.join(ROOT, 'include', 'dt_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dt_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_declared_symbol():
    from deeptables_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build the library first: python __graft_entry__.py'
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(handle, n), f'{n} declared in include/dt_hip.h but not exported'
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    h = _lib.lib()
    assert h.dt_version() == 1 and h.dt_build_arch() == b'gfx950'


def test_argument_validation_without_a_gpu():
    """bad sizes are rejected before any launch -> exercisable on CPU"""
    from deeptables_amd import _lib
    h = _lib.lib()
    assert h.dt_fm_fwd(None, -1, 3, 4, None, None) == -1
    assert b'dt_fm_fwd' in h.dt_last_error()
    assert h.dt_fm_fwd(None, 0, 3, 4, None, None) == 0            # empty batch is a no-op
    assert h.dt_embedding_fwd(None, 7, None, None, None, 4, 2, 8, None, None, None, None) == -1
    assert h.dt_mha_core_fwd(None, None, None, 4, 3, 10, 3, 10, 0.0, 0, None, None, None) == -1    # D % H != 0
    assert h.dt_bn_workspace_bytes(8192, 429) > 0 and h.dt_cross_workspace_bytes(8192, 429, 6) > 0


def test_ops_reject_cpu_tensors_loudly():
    from deeptables_amd import ops
    from deeptables_amd._lib import DtHipError
    with pytest.raises(DtHipError):
        ops.fm(torch.zeros(2, 3, 4))
    with pytest.raises(DtHipError):
        ops.cross(torch.zeros(2, 3), torch.zeros(1, 3), torch.zeros(1, 3))


def test_product_path_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import deeptables_amd, deeptables_amd.models, deeptables_amd.ops, "
            "deeptables_amd.parallel, deeptables_amd.training; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.check_call([sys.executable, '-c', code])


# ---- ModelConfig / metainfo / deepnets --------------------------------------------------------
def test_model_config_defaults_and_fields():
    from deeptables_amd.models import ModelConfig, deepnets
    c = ModelConfig()
    assert len(c._fields) == 45
    assert c.nets == ['dnn_nets'] and c.embeddings_output_dim == 4 and c.embedding_dropout == 0.3
    assert c.dnn_params['hidden_units'] == ((128, 0, False), (64, 0, False))
    assert c.cross_params == {'num_cross_layer': 4} and c.stacking_op == 'add'
    assert c.cin_params['cross_layer_size'] == (128, 128) and c.autoint_params['num_heads'] == 1
    c2 = ModelConfig(nets=deepnets.DeepFM, metrics=['AUC'])
    assert c2.nets == ['linear', 'fm_nets', 'dnn_nets'] and c2.first_metric_name == 'AUC'
    assert hash(c2) == hash('conf-1')
    with pytest.raises(TypeError):
        ModelConfig(no_such_field=1)
    with pytest.raises(ValueError):
        ModelConfig(var_len_categorical_columns=[('a', '|')])


def test_nets_registry_and_signature_check():
    from deeptables_amd.models import deepnets

    def custom_net(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
        return None

    assert deepnets.get('fm_nets') is deepnets.fm_nets
    assert deepnets.get(custom_net) is custom_net and deepnets.custom_nets['custom_net'] is custom_net
    assert deepnets.get_nets(['linear', custom_net, 'linear']) == ['linear', 'custom_net']
    with pytest.raises(ValueError):
        deepnets.register_nets(lambda a, b: None)
    with pytest.raises(TypeError):
        deepnets.get(3)
    with pytest.raises(ValueError):
        deepnets.get(None)


def test_metainfo():
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn, VarLenCategoricalColumn
    c = CategoricalColumn('x', 10000, 0)
    assert c.embeddings_output_dim == 10 and c.input_name == 'cat_x' and c.dtype == 'int32'
    assert ContinuousColumn('n', ['a', 'b', 'c']).input_dim == 3
    assert VarLenCategoricalColumn('v', 5).sep == '|'


def test_ignore_case_dict():
    from deeptables_amd.models.deepmodel import IgnoreCaseDict
    d = IgnoreCaseDict({'AUC': 0.5, 'loss': 1.0})
    assert d['auc'] == 0.5 and d['Auc'] == 0.5 and 'LOSS' in d
    d['Accuracy'] = 0.9
    assert d['accuracy'] == 0.9
    with pytest.raises(KeyError):
        d[1]
    with pytest.raises(KeyError):
        IgnoreCaseDict({1: 2})


# ---- symbolic graph construction (no compute) -------------------------------------------------
def _build_graph(nets, F=6, Nd=3, D=8, **kw):
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    conf = ModelConfig(nets=nets, embeddings_output_dim=D, embedding_dropout=0, **kw)
    cats = [CategoricalColumn(f'C{i}', 10 + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])] if Nd else []
    dm = DeepModel('binary', 2, conf, cats, conts)
    model = dm._build_model('binary', 2, conf.nets, cats, conts, conf)
    return dm, model


def test_deepfm_graph_has_reference_layer_names_and_shapes():
    dm, model = _build_graph(['linear', 'fm_nets', 'dnn_nets'])
    names = list(model.layers_by_name)
    for n in ['emb_categorical_vars_all', 'concat_embeddings_axis_0', 'flatten_embeddings', 'concat_embedding_dense',
              'bn_concat_emb_dense', 'concat_linear_embedding', 'concat_linear_emb_dense', 'linear_logit',
              'concat_fm_embedding', 'fm_layer', 'dnn_dense_1', 'dnn_activation_1', 'dnn_dense_2',
              'dense_logit_dnn_nets', 'add_logits', 'task_output']:
        assert n in names, n
    L = model.layers_by_name
    assert tuple(L['linear_logit'].kernel.shape) == (6 + 3, 1) and L['linear_logit'].bias is None
    assert tuple(L['dnn_dense_1'].kernel.shape) == (6 * 8 + 3, 128)
    assert tuple(L['bn_concat_emb_dense'].gamma.shape) == (51,)
    assert L['bn_concat_emb_dense'].epsilon == 1e-3 and L['bn_concat_emb_dense'].momentum == 0.99
    assert [tuple(e.shape) for e in L['emb_categorical_vars_all'].embeddings] == [(10 + i, 8) for i in range(6)]
    assert model.input_names == ['input_categorical_vars_all', 'input_continuous_all']
    assert 'linear' in str(dm.model_desc) and 'fm: input_shape (None, 6, 8), output_shape (None, 1)' in str(dm.model_desc)


def test_embedding_init_is_keras_uniform_and_dense_glorot():
    dm, model = _build_graph(['linear', 'dnn_nets'])
    L = model.layers_by_name
    t = L['emb_categorical_vars_all'].tables['d8']
    assert float(t.abs().max()) <= 0.05 and float(t.std()) > 0.02
    k = L['dnn_dense_1'].kernel                     # he_uniform: limit sqrt(6/fan_in)
    assert float(k.abs().max()) <= np.sqrt(6.0 / 51) + 1e-6
    k = L['task_output'].kernel                     # glorot_uniform
    assert float(k.abs().max()) <= np.sqrt(6.0 / 2) + 1e-6


def test_other_graphs_build():
    _, m = _build_graph(['linear', 'cin_nets', 'dnn_nets'],
                        cin_params={'cross_layer_size': (8, 8, 4), 'direct': False})
    cin = [l for l in m.layers if l.__class__.__name__ == 'CIN'][0]
    assert [tuple(f.shape) for f in cin.f_] == [(1, 36, 8), (1, 24, 8), (1, 24, 4)]
    assert tuple(cin.exFM_out.kernel.shape) == (4 + 4 + 4, 1)
    _, m = _build_graph(['dcn_nets'], cross_params={'num_cross_layer': 6})
    assert len(m.layers_by_name['dcn_cross_layer'].kernels) == 6
    assert tuple(m.layers_by_name['task_output'].kernel.shape) == (51 + 64, 1)
    _, m = _build_graph(['autoint_nets'], autoint_params={'num_attention': 3, 'num_heads': 2, 'dropout_rate': 0,
                                                          'use_residual': True})
    assert sum(l.__class__.__name__ == 'MultiheadAttention' for l in m.layers) == 3
    assert 'bn_concat_emb_dense' not in m.layers_by_name      # pruned: no net consumes it
    _, m = _build_graph(['pnn_nets'])
    assert tuple(m.layers_by_name['pnn_outer_product_layer'].kernel.shape) == (8, 15, 8)
    _, m = _build_graph(['linear', 'fm_nets'], stacking_op='concat')
    assert 'concat_logits' in m.layers_by_name


def test_error_behaviour_matches_reference():
    from deeptables_amd.models import layers
    from deeptables_amd.functional import Input
    with pytest.raises(ValueError):
        layers.FM()(Input(shape=(8,)))                                       # needs 3-D
    with pytest.raises(ValueError):
        layers.Cross({'num_cross_layer': 2})(Input(shape=(4, 8)))           # needs 2-D
    with pytest.raises(ValueError):
        layers.OuterProduct({'outer_product_kernel_type': 'bad'})
    with pytest.raises(ValueError):
        layers.CIN({'cross_layer_size': ()})
    with pytest.raises(ValueError):
        layers.CIN({'cross_layer_size': (7, 4)})(Input(shape=(4, 8)))       # odd, not last, direct=False
    with pytest.raises(ValueError):
        layers.MultiColumnEmbedding([3, 4], [2])
    with pytest.raises(ValueError):
        _build_graph(['linear'], F=0, Nd=0)                                  # 'No input layer exists.'
    with pytest.raises(ValueError):
        _build_graph(['linear', 'fm_nets'], stacking_op='mul')
    assert set(layers.dt_custom_objects) >= {'MultiColumnEmbedding', 'FM', 'CIN', 'MultiheadAttention', 'Cross',
                                             'InnerProduct', 'OuterProduct'}


def test_single_categorical_column_edge_case():
    """nets_test.py:166-189: one categorical column — fm/ipnn opt out or degrade gracefully."""
    _, m = _build_graph(['linear', 'fm_nets', 'dnn_nets', 'ipnn_nets'], F=1, Nd=2)
    assert 'inner_product_layer' not in m.layers_by_name     # ipnn returns None with < 2 embeddings
    assert 'fm_layer' in m.layers_by_name


def test_preprocessor_metadata():
    import pandas as pd
    from deeptables_amd.models import ModelConfig
    from deeptables_amd.models.deeptable import SimplePreprocessor
    df = pd.DataFrame({'a': ['x', 'y', 'x', None], 'b': [1.0, np.nan, 3.0, 4.0], 'c': [True, False, True, True]})
    pp = SimplePreprocessor(ModelConfig())
    X, y = pp.fit_transform(df, ['p', 'n', 'p', 'n'])
    assert pp.task_ == 'binary' and pp.labels_ == ['n', 'p']
    assert [c.name for c in pp.categorical_columns] == ['a', 'c']
    assert pp.categorical_columns[0].vocabulary_size == 2 + 2          # nunique + 2 (preprocessor.py:333)
    assert pp.continuous_columns[0].column_names == ['b'] and X['b'].isna().sum() == 0
    Xt = pp.transform_X(pd.DataFrame({'a': ['zzz'], 'b': [1.0], 'c': [False]}))
    assert int(Xt['a'][0]) == 0                                        # unseen value -> 0
    assert list(y) == [1.0, 0.0, 1.0, 0.0]


def test_dense_relu_peephole_cpu():
    """Model's execution plan folds Dense -> Activation('relu') when the Dense output has no other consumer and is
    not a model output; results are identical to running the two layers, and proxies asking for the Dense output
    keep the pre-activation values (pure functional layers: runs on CPU through the vendor-GEMM branch)."""
    import torch
    from deeptables_amd import functional as Fn
    Fn.set_seed(0)
    inp = Fn.Input(shape=(6,), name='x')
    d1 = Fn.Dense(8, name='d1')
    a1 = Fn.Activation('relu', name='a1')
    d2 = Fn.Dense(4, name='d2')
    a2 = Fn.Activation('relu', name='a2')
    h = a1(d1(inp))
    t = d2(h)
    out = Fn.Concatenate(name='cc')([a2(t), t])          # d2's output has two consumers: not fused
    m = Fn.Model(inputs=[inp], outputs=out)
    assert len(m._fused_relu) == 1 and len(m._passthrough) == 1
    x = torch.randn(5, 6)
    y = m([x])
    h_ref = torch.relu(x @ d1.kernel + d1.bias)
    t_ref = h_ref @ d2.kernel + d2.bias
    assert torch.allclose(y, torch.cat([torch.relu(t_ref), t_ref], -1), atol=1e-6)
    pre = Fn.Model(inputs=[inp], outputs=d1.output)      # the Dense output itself requested: never fused
    assert not pre._fused_relu and torch.allclose(pre([x]), x @ d1.kernel + d1.bias, atol=1e-6)
    assert (pre([x]) < 0).any()


def test_split_cols_backward_completes_the_shared_buffer():
    """ops.split_cols: generic gradients are concatenated; gradients that already are column blocks of one buffer
    (mha_core's layout) are completed in place and that buffer is returned (pure torch: runs on CPU)."""
    import torch
    from deeptables_amd.ops import split_cols
    D = 4
    want = torch.cat([torch.full((3, 5, D), float(k)) for k in (1, 2, 3, 4)], -1)
    y = torch.randn(3, 5, 4 * D, requires_grad=True)
    a, b, c, d = split_cols(y, D)
    assert a.shape == (3, 5, D) and torch.equal(c, y[..., 2 * D:3 * D])
    (a.sum() * 1 + b.sum() * 2 + c.sum() * 3 + d.sum() * 4).backward()
    assert torch.equal(y.grad, want)
    y.grad = None
    a, b, c, d = split_cols(y, D)
    G = torch.zeros(3, 5, 4 * D)
    G[..., 0:D], G[..., D:2 * D], G[..., 2 * D:3 * D] = 1, 2, 3
    torch.autograd.backward([a, b, c, d], [G[..., 0:D], G[..., D:2 * D], G[..., 2 * D:3 * D], torch.full((3, 5, D), 4.)])
    assert torch.equal(y.grad, want) and y.grad.data_ptr() == G.data_ptr()


def test_committed_traffic_json_names_the_sources_it_was_measured_on():
    """profiles/deepfm_traffic.json (the PMC passes behind bench.py's roofline.traffic) carries the hash of csrc/ + include/
    at measurement time; bench.py reports it only while that equals the running sources.  A stale stamp is not an error
    (any kernel or header edit makes it stale until the two PMC passes are collected again) but it should be visible."""
    import json
    import warnings
    import __graft_entry__ as ge
    path = os.path.join(ROOT, 'profiles', 'deepfm_traffic.json')
    j = json.load(open(path))
    assert {'source_hash', 'bytes_per_step_corrected', 'algorithmic_bytes_per_step', 'per_kernel_KB'} <= set(j)
    assert len(j['source_hash']) == 16
    if j['source_hash'] != ge.source_hash():
        warnings.warn(f"profiles/deepfm_traffic.json was measured on sources {j['source_hash']}, the tree is at "
                      f"{ge.source_hash()}: bench.py will print roofline.traffic = null until tools_pmc.sh is re-run")


def test_roofline_fraction_follows_from_the_committed_evidence():
    """every roofline number reproducible — the committed bench line's `roofline.frac` is recomputed from its
    own fields and the formula of SURVEY §8(d), its per-step time is consistent with the committed rocprofv3 kernel stats,
    and `roofline.traffic` is the committed traffic json's figure."""
    import csv
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.loads(open(os.path.join(root, 'profiles', 'r06_bench_line.json')).read())
    rf = line['roofline']
    F, ND, D, B = 26, 13, 16, 8192
    n_dense = (F * D + ND) * 128 + 128 + 128 * 64 + 64 + 64 + 1 + 1 + 2 * (F * D + ND) + (F + ND)
    bpr = 4 * F + 4 * ND + 8 + 12 * F * D + 12.0 * n_dense / B + 6 * 4 * F * D       # fwd+bwd 5,250 + row Adam 9,984
    assert abs(bpr - rf['algorithmic_bytes_per_row']) < 1.0
    achieved = B * bpr / (rf['launch_us'] * 1e-6) / 1e9
    assert abs(achieved - rf['achieved']) / rf['achieved'] < 1e-6
    assert abs(rf['frac'] - achieved / 8000.0) < 1e-9 and rf['peak'] == 8000.0 and rf['bound'] == 'hbm'
    # the step time against the kernel-trace summary of the same command: the kernels of one step sum to the step within
    # -15 % (kernel boundaries + the replay's fixed cost are in the step, not in the sum) / +5 % (under the tracer every
    # dispatch is timed on its own: with ten steps per replay the untraced step has almost no idle time left to absorb that)
    rows = list(csv.DictReader(open(os.path.join(root, 'profiles', 'r06_deepfm_kernel_stats.csv'))))
    stats = {r['Name'].split('(')[0].replace('void ', ''): (float(r['AverageNs']) / 1e3, int(r['Calls'])) for r in rows}
    # (per replay, not per step: the compiled loop's feed gather + its cursor advance)
    step_kernels = {k: v for k, v in stats.items() if k.startswith('dt::k_') and 'state_init' not in k and 'feed' not in k}
    # round 5: FOUR launches per step in a chained execution — kernel A, the tile kernel, the weight-gradient /
    # row launch, the finishing launch — and the prep launch once per execution (its average weighted by its launches per step)
    steps = max(c for _, c in step_kernels.values())
    per_step = [k for k, (_, c) in step_kernels.items() if c == steps]
    assert len(per_step) == 4 and len(step_kernels) == 5, sorted(stats)
    # (round 6's command — 200 timed steps x 3 regions after the capture's eager steps and the warm-up replay, twenty per
    # replay: 622 step launches, `k_prep` once per replay + once per eager step = 33)
    assert any('k_prep' in k and c * 4 <= steps for k, (_, c) in step_kernels.items())
    total = sum(us * c / steps for us, c in step_kernels.values())
    assert 0.85 * rf['launch_us'] <= total <= 1.05 * rf['launch_us'], (total, rf['launch_us'])
    tj = json.load(open(os.path.join(root, 'profiles', 'deepfm_traffic.json')))
    if rf['traffic'] is not None:
        assert rf['traffic'] == tj['bytes_per_step_corrected']
    assert tj['algorithmic_bytes_per_step'] == int(B * (4 * F + 4 * ND + 8 + 12 * F * D) + 12 * n_dense) + B * 6 * 4 * F * D
    cal = tj['calibration']['kernels']
    assert 0.9 < cal['k_gather<2>']['fetch_factor'] < 1.1 and 0.45 < cal['k_copy 28 MB (16 B/lane stream)']['fetch_factor'] < 0.55


def test_parity_verdict_uses_the_bf16_bar_only_in_bf16_mode():
    """oracle/headline.verdict: the figures the bf16 CIN mode measured at bench size (gpurun_out/r03_line_xdeepfm_bf16,
    round 3) pass north_star's bf16 bar (logits 1e-2) and fail the fp32 one; a wrong gather fails both"""
    from oracle import headline
    res = {'gather_bit_exact': True, 'rows_identical': True, 'max_abs_logit_err': 4.357e-05, 'max_abs_logit': 6.856,
           'dense_grad_rel_err': 0.004838, 'dense_grad_l2_rel_err': 0.005716, 'rows_grad_rel_err': 0.0008991,
           'rows_grad_l2_rel_err': 0.0001644, 'relu_units_near_kink': 45}
    assert headline.verdict(res, bf16=True)[0] is True
    assert headline.verdict(res)[0] is False
    assert headline.verdict(dict(res, gather_bit_exact=False), bf16=True)[0] is False
    assert headline.verdict(dict(res, max_abs_logit_err=0.2), bf16=True)[0] is False


def test_row_weights_of_the_fused_steps_are_validated():
    """fused._row_weights: Keras' sample_weight x class_weight vector as the step's loss block reads it — float32, flat,
    contiguous, one entry per row; a wrong length raises instead of reading past the end"""
    from deeptables_amd.fused import _row_weights
    y = torch.zeros(6, 1)
    assert _row_weights(None, y) is None
    w = _row_weights(torch.arange(12, dtype=torch.float64)[::2].reshape(6, 1), y)
    assert w.dtype == torch.float32 and w.shape == (6,) and w.is_contiguous()
    assert torch.equal(w, torch.tensor([0., 2., 4., 6., 8., 10.]))
    with pytest.raises(ValueError, match='sample_weight has 5 entries for 6 rows'):
        _row_weights(torch.ones(5), y)


def test_unit_gradient_is_recognised_by_storage_not_by_value():
    """training.unit_grad: the cached 1.0 handed to loss.backward(); a loss function skips the multiplication only for THAT
    tensor — another tensor holding 1.0 (or a scaled loss's gradient) is multiplied as usual"""
    from deeptables_amd import training
    u = training.unit_grad(torch.device('cpu'))
    assert u.dim() == 0 and float(u) == 1.0 and training.unit_grad(torch.device('cpu')) is u
    assert training._is_unit_grad(u)
    assert not training._is_unit_grad(torch.ones(()))
    assert not training._is_unit_grad(torch.ones(1))


def test_header_constants_equal_the_binding_s():
    """every `#define DT_*` flag / size of include/dt_hip.h that deeptables_amd/_lib.py mirrors has the same value there"""
    import re
    from deeptables_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'dt_hip.h')).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r'^#define (DT_[A-Z0-9_]+) (0x[0-9a-fA-F]+|\d+)\s*$', text, re.M)}
    assert {'DT_STEP_TOWER_X3', 'DT_STEP_TOWER_BF16', 'DT_STEP_PREELECTED', 'DT_FEED_CURSOR_WORDS'} <= set(defs)
    mirrored = [n for n in defs if hasattr(_lib, n)]
    assert len(mirrored) >= 6, mirrored
    for n in mirrored:
        assert getattr(_lib, n) == defs[n], n
    # the step's phase bits do not overlap
    bits = [defs[n] for n in defs if n.startswith('DT_STEP_')]
    assert len(bits) == len(set(bits)) and all(b & (b - 1) == 0 for b in bits)


def test_tower_precision_modes_map_to_the_step_s_phase_bits(monkeypatch):
    """dnn_params['mfma_dtype'] / DT_AMD_TOWER_DTYPE -> fused._tower_mfma_flag: split-bf16 is the default, exact fp32 and
    plain bf16 are options, anything else is an error (no silent fallback to another precision)"""
    from deeptables_amd import _lib, fused
    monkeypatch.delenv('DT_AMD_TOWER_DTYPE', raising=False)
    assert fused._tower_mfma_flag({}) == _lib.DT_STEP_TOWER_X3
    assert fused._tower_mfma_flag({'mfma_dtype': 'f32'}) == 0
    assert fused._tower_mfma_flag({'mfma_dtype': 'bf16'}) == _lib.DT_STEP_TOWER_BF16
    monkeypatch.setenv('DT_AMD_TOWER_DTYPE', 'f32')
    assert fused._tower_mfma_flag({}) == 0
    assert fused._tower_mfma_flag({'mfma_dtype': 'bf16x3'}) == _lib.DT_STEP_TOWER_X3        # the config wins over the env
    with pytest.raises(ValueError):
        fused._tower_mfma_flag({'mfma_dtype': 'fp8'})


def test_parity_verdict_of_the_bf16_tower_mode():
    """oracle/headline.verdict(bf16='tower'): the figures `bench.py --tower bf16` measured (profiles/r04_bench_lines.jsonl)
    pass the tower mode's bars (logits 1e-2, gradient L2 1e-1), fail the bf16 CIN mode's (L2 2e-2) and the fp32 ones"""
    from oracle import headline
    res = {'gather_bit_exact': True, 'rows_identical': True, 'max_abs_logit_err': 0.00882, 'max_abs_logit': 3.945,
           'dense_grad_rel_err': 0.01706, 'dense_grad_l2_rel_err': 0.03836, 'rows_grad_rel_err': 0.2346,
           'rows_grad_l2_rel_err': 0.06196, 'relu_units_near_kink': 2, 'relu_units': 1572864}
    assert headline.verdict(res, bf16='tower')[0] is True
    assert headline.verdict(res, bf16=True)[0] is False
    assert headline.verdict(res)[0] is False
    assert headline.verdict(dict(res, max_abs_logit_err=0.05), bf16='tower')[0] is False
    assert headline.verdict(dict(res, rows_grad_l2_rel_err=0.2), bf16='tower')[0] is False


def test_steps_per_execution_policy():
    """compiled.resolve_steps_per_execution: 'auto' compiles only graphs with a whole-step plan on a resident device feed
    with enough steps per epoch; explicit values are clamped to the epoch; 0 / 1 mean eager"""
    import types
    from deeptables_amd import compiled
    feed = types.SimpleNamespace(resident=True, device=types.SimpleNamespace(type='cuda'), weighted=False)
    host_feed = types.SimpleNamespace(resident=False, device=types.SimpleNamespace(type='cpu'), weighted=False)
    plan = types.SimpleNamespace(takes_sample_weight=True)
    dm = types.SimpleNamespace(config=types.SimpleNamespace(distribute_strategy=None), fused_plan=lambda: plan)
    no_plan = types.SimpleNamespace(config=types.SimpleNamespace(distribute_strategy=None), fused_plan=lambda: None)
    r = compiled.resolve_steps_per_execution
    assert r(dm, 'auto', feed, 8192, 50) == 20 and r(dm, 'auto', feed, 8192, 25) == 10
    assert r(dm, 'auto', feed, 8192, 12) == 5 and r(dm, 'auto', feed, 8192, 3) == 1
    assert r(no_plan, 'auto', feed, 8192, 50) == 1          # layer-by-layer graphs: on request only
    assert r(no_plan, 4, feed, 8192, 50) == 4 and r(dm, 100, feed, 8192, 23) == 23
    assert r(dm, 1, feed, 8192, 50) == 1 and r(dm, 0, feed, 8192, 50) == 1
    assert r(dm, 'auto', host_feed, 8192, 50) == 1 and r(dm, 10, host_feed, 8192, 50) == 1


def test_no_undefined_names_in_the_host_code():
    """Most of the host code only runs on a GPU box; a name that is never bound anywhere in its module (a missing import —
    round 4 shipped one to the GPU in compiled.py) is caught here, on the CPU, by walking the syntax trees"""
    import ast
    import builtins
    import glob
    files = glob.glob(os.path.join(ROOT, 'deeptables_amd', '**', '*.py'), recursive=True) + \
        [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')] + glob.glob(os.path.join(ROOT, 'oracle', '*.py'))
    known = set(dir(builtins)) | {'__file__', '__name__', '__doc__'}
    bad = []
    for f in files:
        tree = ast.parse(open(f).read())
        bound = set()
        for node in ast.walk(tree):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                bound.update((a.asname or a.name).split('.')[0] for a in node.names)
            elif isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                if not isinstance(node, ast.Lambda):
                    bound.add(node.name)
                a = node.args
                bound.update(x.arg for x in a.args + a.kwonlyargs + a.posonlyargs)
                bound.update(x.arg for x in (a.vararg, a.kwarg) if x is not None)
            elif isinstance(node, ast.ClassDef):
                bound.add(node.name)
            elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
                bound.add(node.id)
            elif isinstance(node, ast.ExceptHandler) and node.name:
                bound.add(node.name)
            elif isinstance(node, (ast.Global, ast.Nonlocal)):
                bound.update(node.names)
        for node in ast.walk(tree):
            if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in bound and node.id not in known:
                bad.append((os.path.relpath(f, ROOT), node.lineno, node.id))
    assert not bad, bad


def test_ctypes_signatures_follow_the_header_prototypes():
    """every prototype of include/dt_hip.h against deeptables_amd/_lib.SIGNATURES: the same number of parameters and the same
    kind (pointer / int / int64 / float) in every position — an ABI drift (a parameter added on one side only) shows up
    here, on the CPU, instead of as a fault on the GPU box"""
    import re
    from deeptables_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'dt_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    protos = re.findall(r'\b(dt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', text)
    assert len(protos) == len(_lib.SIGNATURES)

    def kind(decl):
        if '*' in decl:
            return 'ptr'
        words = decl.replace('const', ' ').split()
        base = ' '.join(words[:-1]) if len(words) > 1 else words[0]
        return {'int': 'int', 'unsigned': 'int', 'int32_t': 'int', 'int64_t': 'i64', 'float': 'f32'}[base]

    of_ctype = {ctypes.c_void_p: 'ptr', ctypes.c_char_p: 'ptr', ctypes.c_int: 'int', ctypes.c_uint: 'int',
                ctypes.c_int64: 'i64', ctypes.c_float: 'f32'}
    for name, params in protos:
        decls = [p.strip() for p in params.split(',') if p.strip() and p.strip() != 'void']
        assert [kind(d) for d in decls] == [of_ctype[a] for a in _lib.SIGNATURES[name][1]], name


def test_dedupe_layout_and_chain_queries_need_no_gpu():
    """Host-side answers of the C-ABI (no launch): the in-step dedupe's workspace layout for batches up to and beyond 8192 rows
    (round 5: any batch size — per-block segment regions up to 8192 rows, per-FIELD regions + cursors + partition bytes beyond)
    and which steps can be chained (`dt_deepfm_step_chains`)."""
    import ctypes
    from deeptables_amd import _lib
    lib = _lib.lib()
    F = 26

    def seg(B):
        out = (ctypes.c_int64 * 7)()
        assert lib.dt_deepfm_dedupe_segments(B, F, ctypes.cast(out, ctypes.c_void_p)) == 0
        return [int(v) for v in out]
    small, big = seg(8192), seg(32768)
    # B <= 8192: one region per election block (fields padded to 8 x B / 1024 hash partitions), 4096 segments each
    assert small[5] == 32 * 8 and small[6] == 4096
    # beyond: one region per field (padded to 8), a field's rows with >= 2 lookups are at most B / 2
    assert big[5] == 32 and big[6] == 32768 // 2
    for B, s in ((8192, small), (32768, big)):
        total = lib.dt_deepfm_dedupe_bytes(B, F)
        assert 0 < s[0] < s[1] < s[2] < s[3] < s[4] < total                      # nseg | seg_row | seg_off | seg_cnt | seg_list
        assert s[4] + 4 * s[5] * B <= total                                     # the member list: regions x B entries
        ov = lib.dt_deepfm_dedupe_overflow_offset(B, F)
        assert s[4] + 4 * s[5] * B <= ov < total and ov % 4 == 0
        assert lib.dt_deepfm_dedupe_slots(B, F) == B * F
    # the partition bytes only exist beyond 8192 rows
    assert lib.dt_deepfm_dedupe_bytes(32768, F) - lib.dt_deepfm_dedupe_overflow_offset(32768, F) >= 32 * 32768
    assert lib.dt_deepfm_dedupe_bytes(8192, F) - lib.dt_deepfm_dedupe_overflow_offset(8192, F) <= 64
    # chained steps: the in-step optimizer on the split-bf16 (or plain-bf16) tile kernel, batches the register-resident election takes
    x3, bf16 = _lib.DT_STEP_TOWER_X3, _lib.DT_STEP_TOWER_BF16
    assert lib.dt_deepfm_step_chains(8192, F, 16, 13, 2 | x3) == 1 and lib.dt_deepfm_step_chains(8192, F, 16, 13, 2 | bf16) == 1
    assert lib.dt_deepfm_step_chains(8192, F, 16, 13, 2) == 0                      # exact-fp32 tower
    assert lib.dt_deepfm_step_chains(8192, F, 16, 13, 1 | x3) == 0                  # forward only
    assert lib.dt_deepfm_step_chains(8192, F, 16, 13, 2 | x3 | _lib.DT_STEP_SKIP_FINISH) == 0
    assert lib.dt_deepfm_step_chains(8193, F, 16, 13, 2 | x3) == 0                  # beyond the register-resident election
    assert lib.dt_deepfm_step_chains(8192, F, 8, 13, 2 | x3) == 1 and lib.dt_deepfm_step_chains(8192, 200, 16, 13, 2 | x3) == 0
