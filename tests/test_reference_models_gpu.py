# -*- coding:utf-8 -*-
"""GPU: the HIP path against outputs of the REFERENCE'S OWN CODE for whole models.

tests/golden/reference_code_model_*.npz hold, for 25 model configurations (the five of BASELINE.json, every preset of
deepnets.py, every net function, stacking add / concat, binary / regression / multiclass heads, a BatchNormalization
tower, no continuous inputs), the inputs, the weights and the output of the reference's DeepModel.__build_model graph as
the reference's own source computes it (tests/golden/make_reference_golden.py: deepmodel.py / deepnets.py / layers.py
imported unmodified on a shim of the TF / Keras primitives, float64).  Here the same ModelConfig is built through the
drop-in API, the fixture's weights are copied in, the forward runs on the GPU through libdt_hip.so, and the logit and the
output must match the reference code's within north_star's tolerance (1e-4, relative to the logit scale above 1).
No oracle arithmetic takes part: fixture -> HIP."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_oracle_reference_code import GRAD_FIXTURES, MODEL_FIXTURES, load_model_fixture  # noqa: E402

pytestmark = pytest.mark.gpu

def _inputs(ids, dense, tensors, dev, idx_dtype=torch.float32):
    """model inputs in the reference's order: categorical ids, the variable-length id blocks, the continuous block
    (deepmodel.py:310-311, 363-385)"""
    vl = [v.to(idx_dtype).to(dev) for v in (tensors.get('var_len_idx') or [])]
    return [ids.to(idx_dtype).to(dev)] + vl + ([] if dense is None else [dense.to(dev)])


@pytest.mark.parametrize('idx_dtype', ['float32', 'int32'])
@pytest.mark.parametrize('path', MODEL_FIXTURES, ids=[os.path.basename(f)[len('reference_code_model_'):-4] for f in MODEL_FIXTURES])
def test_hip_forward_matches_the_reference_codes_output(dev, path, idx_dtype):
    from oracle import bridge          # only its weight loader (no oracle arithmetic below)
    meta, tensors, want = load_model_fixture(path)
    dm, ids, dense = bridge.model_from_reference_fixture(meta['static'], tensors, dev)
    dm.model.train()                   # the fixtures are training-mode forwards (batch statistics in every BatchNormalization)
    inputs = _inputs(ids, dense, tensors, dev, getattr(torch, idx_dtype))
    logit = dm.model(inputs)
    out = dm._activate(logit)
    got = torch.cat([logit, out], -1).detach().double().cpu()
    assert tuple(got.shape) == tuple(want.shape)
    tol = 1e-4 * max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err < tol, f'{os.path.basename(path)}: |HIP - reference code| = {err:.3e} (tolerance {tol:.1e})'


@pytest.mark.parametrize('path', GRAD_FIXTURES, ids=[os.path.basename(f)[len('reference_code_modelgrad_'):-4] for f in GRAD_FIXTURES])
def test_hip_backward_matches_the_reference_codes_gradients(dev, path, monkeypatch):
    """d loss / d every weight: the train step's forward + loss + backward on the GPU (the fused DeepFM / DCN steps where
    the graph is one they take, the layer kernels elsewhere) against autograd through the reference's own graph
    (reference_code_modelgrad_*.npz: Keras loss formulas on the model output, float64)."""
    from oracle import bridge
    from oracle.reference_layers import _leaves, _map_leaves          # nest bookkeeping only
    from deeptables_amd.models import layers as dl
    monkeypatch.setattr(dl, 'DENSE_GRAD_MAX_ELEMS', 1 << 22)          # tiny vocabularies: the exact dense table gradient
    meta, tensors, want = load_model_fixture(path)
    dm, ids, dense = bridge.model_from_reference_fixture(meta['static'], tensors, dev)
    dm.model.train()
    inputs = _inputs(ids, dense, tensors, dev)
    y = tensors['y'].to(torch.float32).to(dev)
    dm.optimizer.zero_grad()
    loss, _ = dm.forward_backward(inputs, y)
    torch.cuda.synchronize()
    # the flat gradient vector back into the weights' nest, then onto the model's parameters
    leaves = _leaves(tensors['weights'])
    pieces = iter(torch.split(want, [t.numel() for t in leaves]))
    grads = _map_leaves(tensors['weights'], lambda t: next(pieces).reshape(t.shape))

    def rel(a, b):
        # error relative to the tensor's largest gradient; a gradient that is exactly zero in exact arithmetic (the concat
        # BatchNormalization's beta in front of a bias-free Dense -> BatchNormalization cell) only has float32 noise: 1e-7
        b = b.double()
        err = (a.detach().double().cpu().reshape(b.shape) - b).abs().max().item()
        return max(err - 1e-7, 0.0) / max(b.abs().max().item(), 1e-6)

    checked = 0
    for p, g in bridge.param_pairs(dm, grads):
        assert p.grad is not None, 'a parameter of the graph got no gradient'
        assert rel(p.grad, torch.as_tensor(g)) < 3e-4, f'{os.path.basename(path)}: parameter of shape {tuple(p.shape)}'
        checked += p.numel()
    emb = dm.model.layers_by_name['emb_categorical_vars_all']
    D = int(tensors['weights']['emb_categorical_vars_all'][0].shape[1])
    table = emb.tables[f'd{D}']
    assert table.grad is not None                      # small vocabularies: the exact dense gradient
    assert rel(table.grad, torch.cat(list(grads['emb_categorical_vars_all']), 0)) < 3e-4
    assert checked + table.numel() == sum(p.numel() for p in dm.model.parameters())


def test_the_fixture_set_covers_the_baseline_configurations():
    names = {os.path.basename(f)[len('reference_code_model_'):-4] for f in MODEL_FIXTURES}
    assert {'fm', 'deepfm', 'xdeepfm', 'autoint', 'dcn'} <= names
    for f in MODEL_FIXTURES:
        meta = json.loads(str(np.load(f)['meta']))
        assert meta['fn'] == '_model_from_parts'
