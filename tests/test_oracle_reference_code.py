# -*- coding:utf-8 -*-
"""CPU: the oracle against outputs of the REFERENCE'S OWN LAYER CODE.

tests/golden/reference_code_*.npz were produced by tests/golden/make_reference_golden.py, which imports the reference's
layers.py, deepnets.py, config.py, metainfo.py and deepmodel.py unmodified (from /root/reference, in the build container)
on an in-process shim of the TensorFlow / Keras primitives they call, runs `build` / `call` of every layer class, every
net function and DeepModel.__build_model (whole models: the five BASELINE.json configurations, every preset, the
stacking / head variants) on seeded inputs and weights, and stores inputs, weights and outputs.  Here every fixture is
replayed through oracle/reference_layers.py: the restatement must reproduce the reference code's output to 1e-12 (float64).  This pins the oracle's op order, axes, splits and transposes to the
reference source; TensorFlow's float32 arithmetic inside a primitive is outside what any restatement controls."""
import glob
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, 'reference_code_*.npz')))


def _unpack(prefix, d):
    if prefix + '@none' in d:
        return None
    if prefix + '@keys' in d:
        return {k: _unpack(f'{prefix}/{k}', d) for k in json.loads(str(d[prefix + '@keys']))}
    if prefix + '@len' in d:
        return [_unpack(f'{prefix}#{k}', d) for k in range(int(d[prefix + '@len']))]
    a = d[prefix]
    return torch.as_tensor(a)


def test_fixture_set_is_complete():
    names = {os.path.basename(f)[len('reference_code_'):-4] for f in FIXTURES}
    assert {'fm', 'cross', 'inner_product', 'outer_product_mat', 'outer_product_vec', 'outer_product_num',
            'cin_split_bias', 'cin_direct_residual', 'cin_split_linear', 'cin_reduce_D_split', 'cin_reduce_D_direct',
            'mha_h2_res1', 'mha_h4_res0',
            'multi_column_embedding', 'afm', 'bilinear_field_all', 'bilinear_field_each', 'bilinear_field_interaction',
            'senet_mean', 'senet_max', 'net_linear', 'net_fm_nets', 'net_dnn_nets', 'net_dcn_nets', 'net_cin_nets',
            'net_autoint_nets',
            # the other layer types of layers.py
            'fgcnn', 'fgcnn_tall_kernel', 'var_len_column_embedding', 'binary_focal_loss', 'categorical_focal_loss',
            'ghmc_loss_step1', 'ghmc_loss_step2', 'ghmc_loss_no_momentum',
            # whole models through the reference's DeepModel.__build_model: the five BASELINE.json configurations ...
            'model_fm', 'model_deepfm', 'model_xdeepfm', 'model_autoint', 'model_dcn',
            # ... the other presets and net functions, and the stacking / head variants
            'model_widedeep', 'model_pnn', 'model_afm', 'model_fibinet', 'model_fgcnn', 'model_opnn_ipnn_vec',
            'model_cross_and_cross_dnn', 'model_fibi_nets_flattened', 'model_fibi_nets_alone', 'model_fgcnn_cin_fm',
            'model_fgcnn_afm_ipnn', 'model_deepfm_concat_nobias', 'model_deepfm_regression', 'model_dnn_multiclass',
            'model_deepfm_bn_tower', 'model_deepfm_no_dense', 'model_dnn_var_len',
            # ... and (round 3) the embedding widths the product's fast kernels take: autoint.hip (D = 16 / 32), cin_bf16.hip
            'model_autoint_d16', 'model_autoint_d32', 'model_xdeepfm_d16'} <= names
    # every whole model also has its loss gradients (autograd through the reference's graph)
    assert {n.replace('model_', 'modelgrad_', 1) for n in names if n.startswith('model_')} <= names


def test_every_reference_layer_class_and_net_function_has_a_fixture():
    """layers.py defines 15 layer / loss classes, deepnets.py 21 net functions: each is executed by at least
    one fixture (a net function either directly, net_*, or inside a whole model, model_*)"""
    metas = [json.loads(str(np.load(f)['meta'])) for f in FIXTURES]
    nets = set()
    for m in metas:
        nets.update(m['static'].get('nets', []))
    assert nets >= {'linear', 'cin_nets', 'fm_nets', 'afm_nets', 'opnn_nets', 'ipnn_nets', 'pnn_nets', 'dnn_nets',
                    'cross_nets', 'cross_dnn_nets', 'dcn_nets', 'autoint_nets', 'fgcnn_cin_nets', 'fgcnn_fm_nets',
                    'fgcnn_afm_nets', 'fgcnn_ipnn_nets', 'fgcnn_dnn_nets', 'fibi_nets', 'fibi_dnn_nets'}
    fns = {m['fn'] for m in metas}
    assert fns >= {'fm', 'cross', 'inner_product', 'outer_product', 'cin', '_mha_from_parts', 'multi_column_embedding', 'afm',
                   'bilinear_interaction', 'senet', '_fgcnn_from_parts', 'var_len_embedding', 'binary_focal_loss',
                   'categorical_focal_loss', '_ghmc_from_parts', '_model_from_parts'}


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(f)[len('reference_code_'):-4] for f in FIXTURES])
def test_oracle_reproduces_the_reference_layer_code(path):
    from oracle import reference_layers as R
    d = dict(np.load(path, allow_pickle=False))
    meta = json.loads(str(d['meta']))
    tensors = {k: _unpack('arg:' + k, d) for k in meta['args']}
    got = getattr(R, meta['fn'])(**tensors, **meta['static'])
    if meta['post'] == 'cat1':
        got = torch.cat(list(got), 1)
    want = torch.as_tensor(d['out'])
    assert tuple(got.shape) == tuple(want.shape)
    err = (got.double() - want.double()).abs().max().item()
    assert err <= 1e-12 * max(1.0, want.abs().max().item()), (meta['fn'], err)


MODEL_FIXTURES = [f for f in FIXTURES if os.path.basename(f).startswith('reference_code_model_')]
GRAD_FIXTURES = [f for f in FIXTURES if os.path.basename(f).startswith('reference_code_modelgrad_')]


def load_model_fixture(path):
    d = dict(np.load(path, allow_pickle=False))
    meta = json.loads(str(d['meta']))
    return meta, {k: _unpack('arg:' + k, d) for k in meta['args']}, torch.as_tensor(d['out'])


def _count(v):
    if v is None:
        return 0
    if isinstance(v, dict):
        return sum(_count(e) for e in v.values())
    if isinstance(v, (list, tuple)):
        return sum(_count(e) for e in v)
    return v.numel()


@pytest.mark.parametrize('path', MODEL_FIXTURES, ids=[os.path.basename(f)[len('reference_code_model_'):-4] for f in MODEL_FIXTURES])
def test_drop_in_model_has_the_reference_models_parameters(path):
    """The graph deeptables_amd builds for the same ModelConfig holds exactly the reference model's weights: same Keras
    layer names, same shapes, same count (the fixture's weights are those of the graph the reference's own
    DeepModel.__build_model created).  Weights in (float32), weights out, oracle on them == the reference code's output.
    Builds on CPU; nothing is computed by the product here (its forward needs the HIP library — the GPU twin of this
    test is tests/test_reference_models_gpu.py)."""
    from oracle import bridge
    meta, tensors, want = load_model_fixture(path)
    dm, ids, dense = bridge.model_from_reference_fixture(meta['static'], tensors, 'cpu')
    assert list(dm.config.nets) == meta['static']['nets']
    n_model = sum(p.numel() for p in dm.model.parameters())
    weights = dict(tensors['weights'])
    readers = {'dnn_nets', 'dcn_nets', 'cross_nets', 'cross_dnn_nets', 'opnn_nets', 'ipnn_nets', 'pnn_nets'}
    if 'bn_concat_emb_dense' not in dm.model.layers_by_name:
        # deepmodel.py:359 always CALLS this BatchNormalization; keras.Model keeps only layers on a path to the output,
        # so a model none of whose nets reads concat_emb_dense has no such weights (the eager shim created them anyway)
        assert not readers & set(meta['static']['nets'])
        weights.pop('bn_concat_emb_dense')
    assert n_model == _count(weights), 'the two graphs differ in their trainable weights'
    logit, act = bridge.oracle_forward(dm, ids, dense, training=True, var_len=tensors.get('var_len_idx'))
    got = torch.cat([logit, act], -1)
    assert tuple(got.shape) == tuple(want.shape)
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())     # float32 storage of the weights
    # every layer the reference names explicitly exists under that name (DeepModel.apply(output_layers=[...]) and Keras
    # weight files address layers by name); counters appended per process (concat_fgcnn_embedding_0) are ignored
    import re
    api = json.load(open(os.path.join(GOLDEN, 'reference_code_api.json')))
    tag = os.path.basename(path)[len('reference_code_model_'):-4]
    strip = lambda n: re.sub(r'_\d+$', '', n)
    mine = {strip(n) for n in dm.model.layers_by_name} | \
        {strip(m.name) for m in dm.model.modules() if isinstance(getattr(m, 'name', None), str)}      # + named sub-layers
    pruned = {'bn_concat_emb_dense', 'concat_embedding_dense', 'flatten_embeddings', 'concat_embeddings_axis'} \
        if 'bn_concat_emb_dense' not in dm.model.layers_by_name else set()     # not on a path to the output (see above)
    missing = {strip(n) for n in api['graph_layer_names'][tag]} - mine - pruned
    assert not missing, (tag, sorted(missing))


def test_drop_in_api_answers_like_the_reference():
    """ModelConfig (fields, order, every default), the column descriptors, the net presets, the set of net
    functions and the error behaviour of the plugin registry — recorded from the reference's own config.py / metainfo.py /
    deepnets.py (tests/golden/reference_code_api.json) and asked of deeptables_amd.models here."""
    from deeptables_amd.models import ModelConfig, deepnets
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn, VarLenCategoricalColumn
    import inspect
    api = json.load(open(os.path.join(GOLDEN, 'reference_code_api.json')))

    def jsonable(v):
        if isinstance(v, (list, tuple)):
            return [jsonable(e) for e in v]
        if isinstance(v, dict):
            return {str(k): jsonable(e) for k, e in v.items()}
        if callable(v):
            return getattr(v, '__name__', 'callable')
        return v if isinstance(v, (str, int, float, bool, type(None))) else repr(v)

    def outcome(fn):
        try:
            return {'ok': jsonable(fn())}
        except Exception as e:
            return {'raises': type(e).__name__}

    cfg0 = ModelConfig()
    assert list(cfg0._fields) == api['ModelConfig_fields']
    for k, want in api['ModelConfig_defaults'].items():
        assert jsonable(getattr(cfg0, k)) == want, k
    assert sorted(cfg0.nets) == api['ModelConfig_default_nets']
    for k, want in api['presets'].items():
        assert list(getattr(deepnets, k)) == want, k
    mine = sorted(n for n in dir(deepnets) if inspect.isfunction(getattr(deepnets, n)) and not n.startswith('_') and
                  getattr(deepnets, n).__module__ == deepnets.__name__ and
                  inspect.signature(getattr(deepnets, n)) == inspect.signature(deepnets.linear))
    assert mine == api['net_functions']
    assert jsonable(CategoricalColumn('c', 10)._asdict()) == api['CategoricalColumn']
    assert jsonable(VarLenCategoricalColumn('v', 10)._asdict()) == api['VarLenCategoricalColumn']
    assert jsonable(ContinuousColumn('x', ['a', 'b'])._asdict()) == api['ContinuousColumn']

    def good_net(embeddings, flatten_emb_layer, dense_layer, concat_emb_dense, config, model_desc):
        return None

    def bad_net(embeddings, dense_layer):
        return None

    N, C = deepnets, ModelConfig
    asked = {
        'get(None)': lambda: N.get(None),
        'get(123)': lambda: N.get(123),
        "get('dnn_nets')": lambda: N.get('dnn_nets'),
        'get(callable with the plugin signature)': lambda: N.get(good_net),
        'get(callable with another signature)': lambda: N.get(bad_net),
        "register_nets('not callable')": lambda: N.register_nets('x'),
        'register_nets(plugin)': lambda: N.register_nets(good_net),
        'get_nets(names + plugin), sorted': lambda: sorted(N.get_nets(['linear', good_net, 'linear'])),
        'ModelConfig(var_len item of length 2)': lambda: C(var_len_categorical_columns=[('g', '|')]),
        'ModelConfig(var_len column also excluded)': lambda: C(exclude_columns=['g'],
                                                               var_len_categorical_columns=[('g', '|', 'max')]),
        'ModelConfig(var_len column also categorical)': lambda: C(categorical_columns=['g'],
                                                                  var_len_categorical_columns=[('g', '|', 'max')]),
        'ModelConfig(var_len ok).var_len_categorical_columns':
            lambda: C(var_len_categorical_columns=[('g', '|', 'max')]).var_len_categorical_columns,
        'hash(ModelConfig) == hash(name)': lambda: hash(C(name='abc')) == hash('abc'),
    }
    assert set(asked) == set(api['behaviour'])
    for k, fn in asked.items():
        assert outcome(fn) == api['behaviour'][k], k

    # DeepModel.__build_model (deepmodel.py:259-317, 436-457): which configurations build, with how many output units, and
    # which raise (a net function returning None is dropped, :284; no net left / no input / unknown task or stacking -> ValueError)
    from deeptables_amd.models import DeepModel

    def build_outcome(nets, task='binary', num_classes=2, n_cat=3, n_dense=2, **conf):
        def go():
            config = ModelConfig(nets=nets, embedding_dropout=0, **conf)
            cats = [CategoricalColumn(f'C{i}', 5 + i, 4) for i in range(n_cat)]
            conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(n_dense)], input_dim=n_dense)] \
                if n_dense else []
            dm = DeepModel(task, num_classes, config, cats, conts)
            dm.build('cpu')
            return [4, int(dm.model.layers_by_name['task_output'].kernel.shape[1])]
        return outcome(go)
    built = {
        'DeepFM, binary': build_outcome(N.DeepFM),
        'DeepFM, multiclass 3': build_outcome(N.DeepFM, task='multiclass', num_classes=3),
        'DeepFM, multilabel 4': build_outcome(N.DeepFM, task='multilabel', num_classes=4),
        'DeepFM, regression': build_outcome(N.DeepFM, task='regression', num_classes=None),
        "stacking_op 'bogus'": build_outcome(N.DeepFM, stacking_op='bogus'),
        "task 'weird'": build_outcome(N.DeepFM, task='weird'),
        'multiclass without num_classes': build_outcome(N.DeepFM, task='multiclass', num_classes=None),
        'no inputs at all': build_outcome(['dnn_nets'], n_cat=0, n_dense=0),
        'fm_nets without categorical columns': build_outcome(['fm_nets'], n_cat=0),
        'afm_nets with one categorical column': build_outcome(['afm_nets'], n_cat=1),
        'pnn_nets with one categorical column, plus dnn_nets': build_outcome(['pnn_nets', 'dnn_nets'], n_cat=1),
        'dnn_nets on continuous inputs only': build_outcome(['dnn_nets'], n_cat=0),
        'dnn_nets on categorical inputs only': build_outcome(['dnn_nets'], n_dense=0),
        'linear on continuous inputs only': build_outcome(['linear'], n_cat=0),
    }
    def loss_outcome(task, num_classes):
        def go():
            dm = DeepModel(task, num_classes, ModelConfig(nets=['dnn_nets'], embedding_dropout=0), [CategoricalColumn('C0', 5, 4)], [])
            dm.build('cpu')
            return dm.model_desc.loss
        return outcome(go)
    for key, want in api['auto_loss'].items():                 # DeepModel.__compile_model, loss='auto'
        t, n = key.split('/')
        assert loss_outcome(t, None if n == 'None' else int(n)) == want, key
    assert set(built) == set(api['build'])
    for k, got in built.items():
        assert got == api['build'][k], (k, got, api['build'][k])


@pytest.mark.parametrize('tag', ['reduce_D_split', 'reduce_D_direct'])
def test_drop_in_cin_composes_reduce_D_filters_like_the_reference(tag):
    """CIN with reduce_D=True (layers.py:652-655, 696-701): the drop-in layer holds the same two low-rank factors per layer and
    hands its kernels their product laid out as the reference's conv1d filter.  CPU: parameter shapes and the composed
    filter against the oracle's (itself replayed against the reference code by the cin_reduce_D_* fixtures); the kernels
    that consume the filter are the reduce_D=False ones."""
    from oracle import reference_layers as R
    from deeptables_amd.models import layers as L
    d = dict(np.load(os.path.join(GOLDEN, f'reference_code_cin_{tag}.npz'), allow_pickle=False))
    meta = json.loads(str(d['meta']))
    x = _unpack('arg:x', d)
    factors = _unpack('arg:reduce_factors', d)
    cls = meta['static']['cross_layer_size']
    layer = L.CIN(params={'cross_layer_size': tuple(cls), 'activation': meta['static']['activation'], 'use_residual': False,
                          'use_bias': _unpack('arg:biases', d) is not None, 'direct': meta['static']['direct'], 'reduce_D': True})
    layer.build(tuple(x.shape))
    assert len(layer.f_) == 0 and len(layer.f0_) == len(cls) == len(layer.f__)
    F0 = x.shape[1]
    for k, (f0, f1) in enumerate(factors):
        assert tuple(layer.f0_[k].shape) == tuple(f0.shape) and tuple(layer.f__[k].shape) == tuple(f1.shape)
        with torch.no_grad():
            layer.f0_[k].copy_(f0.float())
            layer.f__[k].copy_(f1.float())
        want = R.cin_reduced_filter(f0, f1, cls[k], F0 * layer.field_nums[k])[0]
        got = layer._filter(k, cls[k])
        assert tuple(got.shape) == tuple(want.shape)
        assert (got.detach().double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


def _generator_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_reference_golden', os.path.join(GOLDEN, 'make_reference_golden.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)                    # defines functions only; the shim is installed by main()
    return gen


def test_drop_in_layers_keep_the_reference_layers_contract():
    """layers.py's own contract, recorded from the reference classes: every dimension / argument check raises ValueError
    (before anything is launched: the calls below run on CPU tensors), get_config() returns the constructor arguments,
    and dt_custom_objects names the classes load_model needs."""
    from deeptables_amd.models import layers as L
    api = json.load(open(os.path.join(GOLDEN, 'reference_code_api.json')))
    gen = _generator_module()

    def jsonable(v):
        if isinstance(v, (list, tuple)):
            return [jsonable(e) for e in v]
        if isinstance(v, dict):
            return {str(k): jsonable(e) for k, e in v.items()}
        return v if isinstance(v, (str, int, float, bool, type(None))) else repr(v)

    def outcome(fn):
        try:
            return {'ok': jsonable(fn())}
        except Exception as e:
            return {'raises': type(e).__name__}

    got = gen.layer_contract(L, outcome, lambda *shape: torch.zeros(*shape, dtype=torch.float32))
    assert set(got) == set(api['layer_errors'])
    for k, v in got.items():
        assert v == api['layer_errors'][k], (k, v)
    cfgs = gen.layer_configs(L, jsonable)
    for name, want in api['layer_get_config'].items():
        for key, value in want.items():                       # (a Keras base config adds name / trainable / dtype on top)
            assert cfgs[name].get(key) == value, (name, key, cfgs[name].get(key), value)
    assert set(api['dt_custom_objects']) <= set(L.dt_custom_objects)


@pytest.mark.skipif(not os.path.exists('/root/reference/deeptables/models/layers.py'),
                    reason='the reference tree exists only in the build container')
def test_generator_still_agrees_with_the_reference_tree(tmp_path, monkeypatch):
    """where the reference is present: re-run the generator (reference code on the shim vs oracle) into a scratch
    directory and require the committed fixtures to be what it writes"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_reference_golden', os.path.join(GOLDEN, 'make_reference_golden.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    monkeypatch.setattr(gen, 'HERE', str(tmp_path))
    saved = {k: v for k, v in __import__('sys').modules.items()}
    try:
        gen.main()
    finally:
        import sys
        for k in list(sys.modules):
            if k not in saved and (k.startswith('tensorflow') or k.startswith('keras') or k.startswith('hypernets') or k.startswith('deeptables.')
                                   or k == 'deeptables'):
                del sys.modules[k]
    assert json.load(open(os.path.join(GOLDEN, 'reference_code_api.json'))) == \
        json.load(open(os.path.join(str(tmp_path), 'reference_code_api.json')))
    for f in FIXTURES:
        a, b = dict(np.load(f)), dict(np.load(os.path.join(str(tmp_path), os.path.basename(f))))
        assert set(a) == set(b)
        for k in a:
            if k != 'meta':
                assert np.array_equal(a[k], b[k]), (os.path.basename(f), k)
