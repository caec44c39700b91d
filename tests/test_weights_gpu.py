# -*- coding:utf-8 -*-
"""GPU: Keras fit's sample_weight / class_weight (reference deeptable.py:334-365 passes both to keras Model.fit; ADVICE r1:
they were silently ignored).  Weighted loss = sum_i w_i * loss_i / B, gradients scale with it, `fit` threads the
weights through the shuffled batches, `apply_class_weight` computes the balanced weights."""
import numpy as np
import pytest
import torch

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_fused_gpu import build, batch, rel      # noqa: E402

pytestmark = pytest.mark.gpu


def test_weighted_loss_and_gradients(dev):
    dm, cats = build(7, 3, 8, vocab=30)
    idx, dense, y = batch(cats, 3, 96)
    ins = [idx.int().to(dev), dense.to(dev)]
    w = torch.rand(96, generator=torch.Generator().manual_seed(1)) * 3
    dm.model.train()
    loss_w, logit = dm.forward_backward(ins, y.to(dev), w.to(dev))
    z, t = logit.double().cpu().reshape(-1), y.double().reshape(-1)
    per = torch.clamp(z, min=0) - z * t + torch.log1p(torch.exp(-z.abs()))
    assert abs(float(loss_w) - float((per * w.double()).sum() / 96)) < 1e-6
    L = dm.model.layers_by_name
    g_w = L['dnn_dense_1'].kernel.grad.clone()
    # weights identically 2: twice the unweighted gradient (the unweighted step runs on the fused plan)
    dm.forward_backward(ins, y.to(dev), torch.full((96,), 2.0, device=dev))
    g2 = L['dnn_dense_1'].kernel.grad.clone()
    dm.forward_backward(ins, y.to(dev))
    g1 = L['dnn_dense_1'].kernel.grad.clone()
    assert rel(g2, 2 * g1) < 2e-4
    assert rel(g_w, g1) > 1e-3          # and random weights really change it


def test_fit_with_class_and_sample_weights(dev):
    from deeptables_amd.models import ModelConfig, DeepModel, deepnets
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    rng = np.random.RandomState(0)
    n = 600
    X = {'cat': rng.randint(0, 20, size=(n, 4)), 'input_continuous_all': rng.randn(n, 2).astype(np.float32)}

    y = (rng.rand(n) < 0.1).astype(np.float32)                      # imbalanced
    conf = ModelConfig(nets=deepnets.DeepFM, fixed_embedding_dim=True, embeddings_output_dim=8, metrics=['AUC'])
    cats = [CategoricalColumn(f'C{i}', 20, 8) for i in range(4)]
    conts = [ContinuousColumn('input_continuous_all', ['I0', 'I1'])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    seen = []
    orig = dm.train_step

    def spy(ins, yb, wb=None):
        seen.append((yb.detach().cpu().reshape(-1).clone(), None if wb is None else wb.detach().cpu().clone()))
        return orig(ins, yb, wb)
    dm.train_step = spy
    Xf = {'cat': X['cat'], 'input_continuous_all': X['input_continuous_all']}

    class XF:                      # the minimal frame protocol TableBatches / fit use without pandas
        def __init__(self, d):
            self.d = d

        def __len__(self):
            return len(self.d['cat'])

        def __getitem__(self, k):
            if isinstance(k, str):
                return self.d[k]
            return XF({kk: v[k] for kk, v in self.d.items()})
    sw = np.linspace(0.5, 1.5, n).astype(np.float32)
    dm.fit(XF(Xf), y, batch_size=100, epochs=1, verbose=0, validation_split=0, class_weight={0: 1.0, 1: 9.0},
           sample_weight=sw, steps_per_execution=1)           # eager steps: the spy sees every train_step call
    assert seen and all(wb is not None for _, wb in seen)
    for yb, wb in seen:
        # every row's weight is its sample weight (0.5..1.5) times its class weight
        ratio = wb / torch.where(yb > 0.5, torch.tensor(9.0), torch.tensor(1.0))
        assert float(ratio.min()) >= 0.5 - 1e-6 and float(ratio.max()) <= 1.5 + 1e-6
