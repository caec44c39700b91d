# -*- coding:utf-8 -*-
"""CPU: the oracle against outputs of the REFERENCE'S OWN LAYER CODE.

tests/golden/reference_code_*.npz were produced by tests/golden/make_reference_golden.py, which imports the reference's
layers.py unmodified (from /root/reference, in the build container) on an in-process shim of the ~40 TensorFlow / Keras
primitives it calls, runs `build` / `call` of each hot-path layer on seeded inputs and weights, and stores inputs,
weights and outputs.  Here every fixture is replayed through oracle/reference_layers.py: the restatement must reproduce
the reference code's output to 1e-12 (float64).  This pins the oracle's op order, axes, splits and transposes to the
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
    if prefix + '@len' in d:
        return [_unpack(f'{prefix}#{k}', d) for k in range(int(d[prefix + '@len']))]
    a = d[prefix]
    return torch.as_tensor(a)


def test_fixture_set_is_complete():
    names = {os.path.basename(f)[len('reference_code_'):-4] for f in FIXTURES}
    assert {'fm', 'cross', 'inner_product', 'outer_product_mat', 'outer_product_vec', 'outer_product_num',
            'cin_split_bias', 'cin_direct_residual', 'cin_split_linear', 'mha_h2_res1', 'mha_h4_res0',
            'multi_column_embedding', 'afm', 'bilinear_field_all', 'bilinear_field_each', 'bilinear_field_interaction',
            'senet_mean', 'senet_max', 'net_linear', 'net_fm_nets', 'net_dnn_nets', 'net_dcn_nets', 'net_cin_nets',
            'net_autoint_nets'} <= names


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
            if k not in saved and (k.startswith('tensorflow') or k.startswith('keras') or k.startswith('deeptables.')
                                   or k == 'deeptables'):
                del sys.modules[k]
    for f in FIXTURES:
        a, b = dict(np.load(f)), dict(np.load(os.path.join(str(tmp_path), os.path.basename(f))))
        assert set(a) == set(b)
        for k in a:
            if k != 'meta':
                assert np.array_equal(a[k], b[k]), (os.path.basename(f), k)
