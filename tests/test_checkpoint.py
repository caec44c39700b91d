# -*- coding:utf-8 -*-
"""Checkpoint format: Keras weight names, safetensors round trip (weights + optimizer slots).
The naming / file-format checks run on CPU (the graph is assembled without touching the device); the DeepTable
save/load round trip and the SGD kernels need the GPU."""
import os

import numpy as np
import pytest
import torch


def _cpu_model(nets, F=4, Nd=3, D=8, **kw):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    functional.set_seed(7)
    conf = ModelConfig(nets=nets, fixed_embedding_dim=True, embeddings_output_dim=D, embedding_dropout=0,
                       dense_dropout=0, metrics=['AUC'], **kw)
    cats = [CategoricalColumn(f'C{i}', 11 + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    return dm, dm._build_model(dm.task, dm.num_classes, conf.nets, cats, conts, conf)


def test_keras_weight_names_cpu():
    from deeptables_amd import checkpoint
    _, m = _cpu_model(['linear', 'fm_nets', 'dnn_nets', 'cin_nets', 'dcn_nets', 'autoint_nets'])
    names = list(checkpoint.named_weights(m))
    # the reference's variables: layers.py:863-877 (embeddings_i), :423-426 (kernels_i/bias_i), :659 (f_i)
    for want in ['emb_categorical_vars_all/embeddings_0', 'emb_categorical_vars_all/embeddings_3',
                 'bn_concat_emb_dense/gamma', 'bn_concat_emb_dense/moving_variance', 'linear_logit/kernel',
                 'dnn_dense_1/kernel', 'dnn_dense_1/bias', 'dcn_cross_layer/kernels_0', 'dcn_cross_layer/bias_3',
                 'cin/f_0', 'cin/f_1', 'multihead_attention/dense_Q/kernel', 'multihead_attention/batch_normalize/gamma',
                 'task_output/kernel', 'task_output/bias']:
        assert want in names, want
    assert not any('row_offset' in n or 'oob' in n or 'tables' in n for n in names)
    w = checkpoint.named_weights(m)
    assert tuple(w['emb_categorical_vars_all/embeddings_2'].shape) == (13, 8)
    assert tuple(w['dcn_cross_layer/kernels_0'].shape) == (4 * 8 + 3, 1)
    # the per-column variables are views of ONE packed table
    base = m.layers_by_name['emb_categorical_vars_all'].tables['d8']
    assert w['emb_categorical_vars_all/embeddings_1'].data_ptr() == base.data_ptr() + 11 * 8 * 4


def test_f3_weight_names_cpu():
    from deeptables_amd import checkpoint
    _, m = _cpu_model(['afm_nets', 'fibi_dnn_nets', 'fgcnn_dnn_nets'],
                      fibinet_params={'senet_pooling_op': 'mean', 'senet_reduction_ratio': 3,
                                      'bilinear_type': 'field_interaction'})
    names = list(checkpoint.named_weights(m))
    senet = [n for n in names if n.startswith('senet_bilinear_layer_')]
    assert [n.split('/')[1] for n in senet] == ['bilinear_weight0_1', 'bilinear_weight0_2', 'bilinear_weight0_3',
                                                 'bilinear_weight1_2', 'bilinear_weight1_3', 'bilinear_weight2_3']
    assert 'afm_layer/projection_h' in names and 'afm_layer/dense_attention/kernel' in names
    assert 'fgcnn/conv2d/kernel' in names and 'fgcnn/conv2d/bias' in names


def test_safetensors_round_trip_cpu(tmp_path):
    from deeptables_amd import checkpoint
    from safetensors import safe_open
    _, m = _cpu_model(['linear', 'fm_nets', 'dnn_nets'])
    path = str(tmp_path / 'm.safetensors')
    written = checkpoint.save_model(m, path, metadata={'nets': ['linear', 'fm_nets', 'dnn_nets']})
    with safe_open(path, framework='pt') as f:
        assert f.metadata()['format'] == checkpoint.FORMAT
        assert set(f.keys()) == set(written)
        assert torch.equal(f.get_tensor('dnn_dense_1/kernel'), m.layers_by_name['dnn_dense_1'].kernel.detach())
    before = {k: v.detach().clone() for k, v in checkpoint.named_weights(m).items()}
    with torch.no_grad():
        for v in checkpoint.named_weights(m).values():
            v.add_(1.0)
    table_ptr = m.layers_by_name['emb_categorical_vars_all'].tables['d8'].data_ptr()
    checkpoint.load_model(m, path)
    for k, v in checkpoint.named_weights(m).items():
        assert torch.equal(v, before[k]), k
    assert m.layers_by_name['emb_categorical_vars_all'].tables['d8'].data_ptr() == table_ptr   # loaded in place
    # a file of a different graph is refused
    _, other = _cpu_model(['dcn_nets'])
    with pytest.raises(KeyError):
        checkpoint.load_model(other, path)
    with pytest.raises(ImportError):
        checkpoint.import_keras_h5(m, str(tmp_path / 'x.h5'))


def test_save_with_optimizer_refused_under_sharded_tables(tmp_path):
    """a rank of ShardedEmbeddingStrategy holds Adam moments only for its own fields: one rank's file must not pass for
    the whole optimizer state"""
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn

    class FakeSharded:
        sharded_embeddings, active, world_size, rank = True, True, 2, 0
    conf = ModelConfig(nets=['dnn_nets'], embeddings_output_dim=4, embedding_dropout=0, distribute_strategy=FakeSharded())
    dm = DeepModel('binary', 2, conf, [CategoricalColumn('C0', 10, 4)], [ContinuousColumn('input_continuous_all', ['a'])])
    dm.build('cpu')
    with pytest.raises(ValueError, match='ShardedEmbeddingStrategy'):
        dm.save(str(tmp_path / 'm.safetensors'), include_optimizer=True)
    dm.save(str(tmp_path / 'm.safetensors'))                       # weights only: fine


@pytest.mark.gpu
def test_deepmodel_save_load_with_optimizer(dev, tmp_path):
    """Train two steps, save (weights + Adam slots), load into a fresh model, one more step on both: identical."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_models_gpu import build, batch
    from deeptables_amd.models import DeepModel
    from deeptables_amd.models import layers as dl
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0            # exercise the row-sparse optimizer path too
    try:
        dm, cats = build('DeepFM', vocab=60, seed=21)
        idx, dense, y = batch(cats, 13, 128, dev)
        ins = [idx.int().to(dev), dense.to(dev)]
        dm.model.train()
        for _ in range(2):
            dm.train_step(ins, y.to(dev))
        path = str(tmp_path / 'dm.safetensors')
        dm.save(path, include_optimizer=True)
        dm2 = DeepModel('binary', 2, dm.config, dm.categorical_columns, dm.continuous_columns, model_file=path)
        assert dm2.optimizer.t == 2
        for a, b in zip(dm.model.state_dict().values(), dm2.model.state_dict().values()):
            assert torch.equal(a, b)
        dm2.model.train()
        l1, _ = dm.train_step(ins, y.to(dev))
        l2, _ = dm2.train_step(ins, y.to(dev))
        assert abs(float(l1) - float(l2)) < 1e-7
        t1 = dm.model.layers_by_name['emb_categorical_vars_all'].tables['d16']
        t2 = dm2.model.layers_by_name['emb_categorical_vars_all'].tables['d16']
        assert torch.allclose(t1, t2, atol=1e-7)
        assert torch.allclose(dm.model.layers_by_name['dnn_dense_1'].kernel,
                              dm2.model.layers_by_name['dnn_dense_1'].kernel, atol=1e-7)
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old


@pytest.mark.gpu
def test_sgd_kernels(dev):
    from deeptables_amd.training import SGD
    from deeptables_amd.ops import SparseRowGrad
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(300, generator=g)
    t0 = torch.randn(50, 8, generator=g)
    w, table = torch.nn.Parameter(w0.clone().to(dev)), torch.nn.Parameter(t0.clone().to(dev))

    class Emb:
        tables = {'d8': table}
        sparse_grads = {}
    opt = SGD([w, table], [Emb], learning_rate=0.1)
    wg = torch.randn(300, generator=g)
    rows = torch.randint(0, 50, (400,), generator=g)
    rows[::7] = -1
    vals = torch.randn(400, 8, generator=g)
    w.grad = wg.to(dev)
    Emb.sparse_grads = {'d8': [SparseRowGrad(rows.to(dev), vals.to(dev))]}
    opt.step()
    ref = t0.double().clone()
    ok = rows >= 0
    ref.index_add_(0, rows[ok], -0.1 * vals.double()[ok])
    assert torch.allclose(w.detach().cpu(), w0 - 0.1 * wg, atol=1e-6)
    assert (table.detach().cpu().double() - ref).abs().max().item() < 1e-5
