# -*- coding:utf-8 -*-
"""SURVEY §8 f1 — the input feed (training.TableBatches): int32 ids end to end; the streamed mode (pinned host table,
ring of pinned staging buffers, async H2D on a side stream) yields exactly the batches of the device-resident mode."""
import numpy as np
import pandas as pd
import pytest
import torch

from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn, VarLenCategoricalColumn
from deeptables_amd.training import TableBatches


def _frame(n=1003, seed=0):
    rng = np.random.default_rng(seed)
    df = pd.DataFrame({'a': rng.integers(0, 10, n), 'b': rng.integers(0, 5, n), 'x': rng.normal(size=n),
                       'z': rng.normal(size=n)})
    df['g'] = list(rng.integers(0, 6, (n, 3)))
    y = rng.integers(0, 2, n)
    cats = [CategoricalColumn('a', 12, 4), CategoricalColumn('b', 7, 4)]
    conts = [ContinuousColumn('input_continuous_all', ['x', 'z'])]
    vl = VarLenCategoricalColumn('g', 6, 4)
    vl.max_elements_length = 3
    return df, y, cats, conts, [vl]


def _collect(tb, batch, shuffle, drop, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return [([t.clone().cpu() for t in ins], yb.clone().cpu()) for ins, yb in tb.iterate(batch, shuffle, drop, generator=g)]


@pytest.mark.parametrize('shuffle', [False, True])
@pytest.mark.parametrize('drop', [False, True])
def test_streamed_equals_resident_cpu(shuffle, drop):
    df, y, cats, conts, vls = _frame()
    r = TableBatches(df, y, cats, conts, 'cpu', resident=True, var_len_categorical_columns=vls)
    s = TableBatches(df, y, cats, conts, 'cpu', resident=False, ring=3, var_len_categorical_columns=vls)
    A, B = _collect(r, 128, shuffle, drop, 5, 'cpu'), _collect(s, 128, shuffle, drop, 5, 'cpu')
    assert len(A) == len(B) == (7 if drop else 8)
    for (ia, ya), (ib, yb) in zip(A, B):
        assert len(ia) == 3 and ia[0].dtype == torch.int32 and ia[1].dtype == torch.int32     # cat, var-len, dense
        assert ia[1].shape[1] == 3 and ia[2].dtype == torch.float32
        assert all(torch.equal(p, q) for p, q in zip(ia, ib)) and torch.equal(ya, yb)
    if not shuffle:                                                  # order = frame order
        assert torch.equal(A[0][0][0][:, 0], torch.as_tensor(df['a'].values[:128]).int())
    rows = torch.cat([ins[0] for ins, _ in A])
    assert rows.shape[0] == (896 if drop else 1003)
    # auto mode: small tables are resident, a tiny budget forces streaming
    assert TableBatches(df, y, cats, conts, 'cpu').resident
    assert not TableBatches(df, y, cats, conts, 'cpu', max_resident_bytes=100).resident


@pytest.mark.gpu
def test_streamed_equals_resident_gpu_and_trains(dev):
    df, y, cats, conts, vls = _frame(n=5000, seed=3)
    r = TableBatches(df, y, cats, conts, dev, resident=True, var_len_categorical_columns=vls)
    s = TableBatches(df, y, cats, conts, dev, resident=False, ring=3, var_len_categorical_columns=vls)
    assert s.blocks[0].is_pinned() and not s.blocks[0].is_cuda and r.blocks[0].is_cuda
    for epoch in range(2):                                         # the ring is reused across epochs
        A, B = _collect(r, 512, True, False, 11 + epoch, dev), _collect(s, 512, True, False, 11 + epoch, dev)
        assert len(A) == len(B) == 10
        for (ia, ya), (ib, yb) in zip(A, B):
            assert all(torch.equal(p, q) for p, q in zip(ia, ib)) and torch.equal(ya, yb)
    # a model trained through the streamed feed follows the resident one step for step
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel
    losses = []
    for resident in (True, False):
        functional.set_seed(4)
        torch.manual_seed(0)
        np.random.seed(0)
        conf = ModelConfig(nets=['linear', 'fm_nets', 'dnn_nets'], fixed_embedding_dim=True, embeddings_output_dim=4,
                           embedding_dropout=0, metrics=['AUC'], earlystopping_patience=0)
        dm = DeepModel('binary', 2, conf, cats, conts)
        dm.feed_resident = resident
        hist = dm.fit(df, y.astype(np.float32), batch_size=256, epochs=2, verbose=0, validation_split=0, shuffle=False)
        losses.append(hist.history['loss'])
    assert np.allclose(losses[0], losses[1], atol=1e-6)
