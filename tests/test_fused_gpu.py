# -*- coding:utf-8 -*-
"""GPU: the fused DeepFM train step (csrc/deepfm.hip, 6 launches) must produce the same logits, loss
and gradients as the oracle (1e-4) and as the layer-by-layer HIP path of this repo."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['bf16x3', 'f32'], autouse=True)
def tower_mode(request, monkeypatch):
    """every test of this file runs the tile kernel of the step both ways: split-bf16 matrix cores (k_tower_x3, the default)
    and exact-fp32 MFMA (k_mlp_fwd3) — the same bars for both"""
    monkeypatch.setenv('DT_AMD_TOWER_DTYPE', request.param)
    return request.param


def build(F, Nd, D, vocab, seed=3, use_bias=True, nets=None, task='binary', **extra):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel, deepnets
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    functional.set_seed(seed)
    conf = ModelConfig(nets=nets or deepnets.DeepFM, fixed_embedding_dim=True, embeddings_output_dim=D,
                       embedding_dropout=extra.pop('embedding_dropout', 0), metrics=['AUC'], output_use_bias=use_bias,
                       **extra)
    cats = [CategoricalColumn(f'C{i}', vocab + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])] if Nd else []
    dm = DeepModel(task, 2 if task == 'binary' else 1, conf, cats, conts)
    dm.build()
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for n, p in dm.model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g).to(p.device) * 0.1)
    return dm, cats


def batch(cats, Nd, B, seed=5):
    g = torch.Generator().manual_seed(seed)
    idx = torch.stack([torch.randint(0, c.vocabulary_size, (B,), generator=g) for c in cats], 1)
    dense = torch.randn(B, Nd, generator=g) * 1.5 + 0.5 if Nd else None
    y = (torch.rand(B, 1, generator=g) < 0.25).float()
    return idx, dense, y


def rel(a, b):
    b = b.detach().double().cpu()
    return (a.detach().double().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize('B,F,Nd,D,idt', [(256, 26, 13, 16, 'int32'), (100, 26, 13, 16, 'float32'), (37, 5, 3, 8, 'int32'),
                                          (64, 7, 0, 4, 'int32'), (513, 16, 2, 32, 'int32')])
def test_fused_deepfm_matches_oracle_and_generic_path(dev, B, F, Nd, D, idt):
    from oracle import bridge, reference_layers as R
    dm, cats = build(F, Nd, D, vocab=30)
    plan = dm.fused_plan()
    assert plan is not None, 'DeepFM graph should be eligible for the fused plan'
    idx, dense, y = batch(cats, Nd, B)
    # oracle
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_logit, _ = bridge.oracle_forward(dm, idx, dense, training=True, weights=w)
    ref_loss = R.binary_crossentropy_from_logits(ref_logit, y.double())
    ref_loss.backward()
    mm0 = dm.model.layers_by_name['bn_concat_emb_dense'].moving_mean.clone()
    # fused
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
        assert rel(a, b) < 2e-4, f'dense grad {i}: {rel(a, b)}'
    table = L['emb_categorical_vars_all'].tables[f'd{D}']
    ref_tg = torch.cat([t.grad for t in w['emb_categorical_vars_all']], 0)
    assert rel(table.grad, ref_tg) < 2e-4
    # BN moving statistics were updated exactly once
    bn = L['bn_concat_emb_dense']
    x = torch.cat([torch.cat(R.multi_column_embedding(idx.float(), [t.detach() for t in w['emb_categorical_vars_all']]),
                             1).reshape(B, -1)] + ([dense.double()] if Nd else []), -1)
    assert rel(bn.moving_mean, mm0.double().cpu() * 0.99 + x.mean(0) * 0.01) < 1e-4
    # generic layer-by-layer path gives the same thing
    import os
    fused_grads = [a.clone() for a, _ in pairs]
    dm._fused_plan = None
    loss2, logit2 = dm.forward_backward(ins, y.to(dev))
    assert (logit2 - logit).abs().max().item() < 1e-4
    for i, (a, _) in enumerate(pairs):
        pass
    assert rel(L['dnn_dense_1'].kernel.grad, fused_grads[4]) < 2e-4


@pytest.mark.parametrize('B,F,Nd,D,L,idt', [(256, 26, 13, 16, 6, 'int32'), (100, 26, 13, 16, 4, 'float32'),
                                            (37, 5, 3, 8, 2, 'int32'), (64, 7, 0, 4, 1, 'int32'), (513, 16, 2, 16, 8, 'int32')])
def test_fused_dcn_matches_oracle_and_generic_path(dev, B, F, Nd, D, L, idt):
    """nets ['dcn_nets'] (deepnets.py:194-207): the fused step with the Cross network inside the tile kernel"""
    from oracle import bridge, reference_layers as R
    from deeptables_amd.models import deepnets
    from deeptables_amd.fused import FusedDCN
    dm, cats = build(F, Nd, D, vocab=30, nets=deepnets.DCN, cross_params={'num_cross_layer': L},
                     dnn_params={'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu'})
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():       # cross biases start at zero: give them (and the kernels) some size
        cr = dm.model.layers_by_name['dcn_cross_layer']
        cr.bias_stack.add_(torch.randn(cr.bias_stack.shape, generator=g).to(cr.bias_stack.device) * 0.05)
    plan = dm.fused_plan()
    assert isinstance(plan, FusedDCN), 'the DCN graph should be eligible for the fused plan'
    idx, dense, y = batch(cats, Nd, B)
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_logit, _ = bridge.oracle_forward(dm, idx, dense, training=True, weights=w)
    ref_loss = R.binary_crossentropy_from_logits(ref_logit, y.double())
    ref_loss.backward()
    dm.model.train()
    ins = [idx.to(getattr(torch, idt)).to(dev)] + ([dense.to(dev)] if Nd else [])
    loss, logit = dm.forward_backward(ins, y.to(dev))
    torch.cuda.synchronize()
    assert (logit.double().cpu() - ref_logit).abs().max().item() < 1e-4 * max(1.0, ref_logit.abs().max().item())
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    Ly = dm.model.layers_by_name
    cr = Ly['dcn_cross_layer']
    pairs = [(Ly['task_output'].kernel.grad, w['task_output'][0].grad),
             (Ly['task_output'].kernel.grad, w['task_output'][0].grad),
             (Ly['dcn_dense_2'].kernel.grad, w['dcn_dnn'][1][0].grad), (Ly['dcn_dense_2'].bias.grad, w['dcn_dnn'][1][1].grad),
             (Ly['dcn_dense_1'].kernel.grad, w['dcn_dnn'][0][0].grad), (Ly['dcn_dense_1'].bias.grad, w['dcn_dnn'][0][1].grad),
             (Ly['bn_concat_emb_dense'].gamma.grad, w['bn_concat_emb_dense'][0].grad),
             (Ly['bn_concat_emb_dense'].beta.grad, w['bn_concat_emb_dense'][1].grad),
             (cr.kernel_stack.grad, torch.stack([k.grad.reshape(-1) for k in w['dcn_cross_kernels']], 0)),
             (cr.bias_stack.grad, torch.stack([b.grad.reshape(-1) for b in w['dcn_cross_bias']], 0)),
             (Ly['task_output'].bias.grad, w['task_output'][1].grad)]
    for i, (a, b) in enumerate(pairs):
        assert rel(a, b) < 2e-4, f'dense grad {i}: {rel(a, b)}'
    table = Ly['emb_categorical_vars_all'].tables[f'd{D}']
    ref_tg = torch.cat([t.grad for t in w['emb_categorical_vars_all']], 0)
    assert rel(table.grad, ref_tg) < 2e-4
    # the layer-by-layer path gives the same thing
    fused = [a.clone() for a, _ in pairs]
    dm._fused_plan = None
    loss2, logit2 = dm.forward_backward(ins, y.to(dev))
    # two fp32 evaluation orders of an L-layer cross network: 1e-4 of the logit scale
    assert (logit2 - logit).abs().max().item() < 1e-4 * max(1.0, logit.abs().max().item())
    assert rel(Ly['dcn_dense_1'].kernel.grad, fused[4]) < 2e-4
    assert rel(cr.kernel_stack.grad, fused[8]) < 2e-4
    # and a whole train step (Adam on the flat buffer incl. the cross parameters) runs
    del dm._fused_plan
    k0 = cr.kernel_stack.detach().clone()
    dm.train_step(ins, y.to(dev))
    assert (cr.kernel_stack.detach() - k0).abs().max().item() > 0


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
def test_fused_steps_take_the_regression_task(dev, net):
    """task 'regression' -> loss 'mse' on the linear task_output (deepmodel.py:126-141, 216): DT_STEP_LOSS_MSE"""
    from oracle import bridge
    from deeptables_amd.models import deepnets
    from deeptables_amd.fused import FusedDCN, FusedDeepFM
    extra = dict(cross_params={'num_cross_layer': 3}) if net == 'DCN' else {}
    dm, cats = build(9, 4, 8, vocab=30, nets=getattr(deepnets, net), task='regression',
                     dnn_params={'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu'}, **extra)
    assert dm.loss_name == 'mse'
    assert isinstance(dm.fused_plan(), FusedDCN if net == 'DCN' else FusedDeepFM)
    idx, dense, _ = batch(cats, 4, 200)
    y = torch.randn(200, 1, generator=torch.Generator().manual_seed(3)) * 2
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_out, _ = bridge.oracle_forward(dm, idx, dense, training=True, weights=w)
    ref_loss = ((ref_out - y.double()) ** 2).mean()
    ref_loss.backward()
    dm.model.train()
    ins = [idx.int().to(dev), dense.to(dev)]
    loss, out = dm.forward_backward(ins, y.to(dev))
    torch.cuda.synchronize()
    assert (out.double().cpu() - ref_out).abs().max().item() < 1e-4 * max(1.0, ref_out.abs().max().item())
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * max(1.0, float(ref_loss))
    Ly = dm.model.layers_by_name
    pre = 'dcn' if net == 'DCN' else 'dnn'
    key = 'dcn_dnn' if net == 'DCN' else 'dnn'
    assert rel(Ly[f'{pre}_dense_1'].kernel.grad, w[key][0][0].grad) < 2e-4
    assert rel(Ly['task_output'].kernel.grad, w['task_output'][0].grad) < 2e-4
    assert rel(Ly['bn_concat_emb_dense'].gamma.grad, w['bn_concat_emb_dense'][0].grad) < 2e-4
    table = Ly['emb_categorical_vars_all'].tables['d8']
    assert rel(table.grad, torch.cat([t.grad for t in w['emb_categorical_vars_all']], 0)) < 2e-4
    # same through the layer-by-layer path
    dm._fused_plan = None
    loss2, out2 = dm.forward_backward(ins, y.to(dev))
    assert abs(float(loss2) - float(loss)) < 1e-5 * max(1.0, float(loss))


def test_fused_sparse_gradient_rows(dev):
    """large-table mode: the fused step hands (rows, values) to the row-sparse optimizer"""
    from deeptables_amd.models import layers as dl
    old = dl.DENSE_GRAD_MAX_ELEMS
    try:
        dl.DENSE_GRAD_MAX_ELEMS = 0
        dm, cats = build(26, 13, 16, vocab=100)
        idx, dense, y = batch(cats, 13, 128)
        dm.model.train()
        ins = [idx.int().to(dev), dense.to(dev)]
        dm.forward_backward(ins, y.to(dev))
        emb = dm.model.layers_by_name['emb_categorical_vars_all']
        sg = emb.sparse_grads['d16'][0]
        offs = np.concatenate([[0], np.cumsum([c.vocabulary_size for c in cats])[:-1]])
        want_rows = idx.numpy() + offs[None, :]
        got_rows = sg.rows.cpu().numpy().reshape(128, 26)
        # the fused step dedupes: a row looked up once keeps its entry, rows looked up several times become segments
        # (all their lookups report -1)
        kept = got_rows >= 0
        assert np.array_equal(got_rows[kept], want_rows[kept]) and sg.fields == -1
        flat = want_rows.reshape(-1)
        uniq, counts = np.unique(flat, return_counts=True)
        assert sorted(got_rows[kept].tolist()) == sorted(uniq[counts == 1].tolist())
        nseg_r, srow, soff, scnt, slist, regions, cap = sg.segments
        valid = (torch.arange(cap, device=srow.device)[None, :] < nseg_r.long()[:, None]).reshape(-1)
        srow_v, soff_v, scnt_v = srow[valid].cpu(), soff[valid].cpu(), scnt[valid].cpu()
        assert len(srow_v) == int((counts > 1).sum()) and int(scnt_v.sum()) == int(counts[counts > 1].sum())
        assert sorted(srow_v.tolist()) == sorted(uniq[counts > 1].tolist())
        slist_c = slist.cpu().numpy()
        for s_ in range(len(srow_v)):        # every segment lists exactly the lookups of its row
            members = slist_c[int(soff_v[s_]):int(soff_v[s_]) + int(scnt_v[s_])]
            assert np.all(flat[members] == int(srow_v[s_])) and len(set(members.tolist())) == int(scnt_v[s_])
        assert np.array_equal(sg.expanded()[0].cpu().numpy().reshape(128, 26), want_rows)
        V = emb.tables['d16'].shape[0]

        def densify(g):
            d = torch.zeros(V, 16, device=dev)
            rows, values = g.expanded()
            ok = rows.reshape(-1) >= 0
            d.index_add_(0, rows.reshape(-1)[ok], values.reshape(-1, 16)[ok])
            return d
        dense_fused = densify(sg)
        # same per-row sums from the generic path (which keeps one entry per lookup)
        dm._fused_plan = None
        dm.forward_backward(ins, y.to(dev))
        sg2 = emb.sparse_grads['d16'][0]
        dense_generic = densify(sg2)
        assert (dense_generic - dense_fused).abs().max().item() < 1e-6 + 2e-4 * dense_generic.abs().max().item()
        # and a full train step runs through the sparse Adam
        del dm._fused_plan
        t0 = emb.tables['d16'].detach().clone()
        dm.train_step(ins, y.to(dev))
        assert (emb.tables['d16'].detach() - t0).abs().max().item() > 0
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old


def test_ineligible_graphs_fall_back_to_layer_kernels(dev):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    for hu in (((256, 0, False), (64, 0, False)),          # wider than the compiled 128 x 64 tile
               ((64, 0, True), (32, 0, False)),            # a BatchNormalization cell
               ((64, 0.1, False), (32, 0, False)),         # tower dropout
               ((64, 0, False), (1, 0, False)),            # a width-1 tower output gets no dense_logit_* layer (deepmodel.py:291)
               ((64, 0, False),), ((64, 0, False), (32, 0, False), (16, 0, False))):
        conf = ModelConfig(nets=['linear', 'fm_nets', 'dnn_nets'], embeddings_output_dim=8, embedding_dropout=0,
                           dnn_params={'hidden_units': hu, 'activation': 'relu'})
        dm = DeepModel('binary', 2, conf, [CategoricalColumn(f'C{i}', 10, 8) for i in range(4)],
                       [ContinuousColumn('input_continuous_all', ['a', 'b'])])
        dm.build()
        assert dm.fused_plan() is None, hu
    conf = ModelConfig(nets=['linear', 'fm_nets', 'dnn_nets'], embeddings_output_dim=8, embedding_dropout=0,
                       dnn_params={'hidden_units': ((64, 0, False), (32, 0, False)), 'activation': 'tanh'})
    dm = DeepModel('binary', 2, conf, [CategoricalColumn(f'C{i}', 10, 8) for i in range(4)],
                   [ContinuousColumn('input_continuous_all', ['a', 'b'])])
    dm.build()
    assert dm.fused_plan() is None


@pytest.mark.parametrize('net,H1,H2', [('DeepFM', 100, 40), ('DeepFM', 64, 32), ('DeepFM', 128, 17), ('DeepFM', 3, 2),
                                       ('DCN', 100, 40), ('DCN', 32, 64)])
def test_fused_steps_take_narrower_towers(dev, tmp_path, net, H1, H2):
    """hidden_units narrower than the compiled 128 x 64 tile (deepnets.py:401-427) run on the same kernels: the plan keeps
    W1 / b1 / W2 / b2 / w3 inside zero-padded slabs and the model's parameters are views of their leading blocks.
    Gradients vs the oracle, three Adam steps vs the layer-by-layer path, the pads stay exactly zero, and a checkpoint
    with optimizer slots round-trips through the strided views."""
    from oracle import bridge, reference_layers as R
    from deeptables_amd import checkpoint
    from deeptables_amd.models import deepnets
    from deeptables_amd.fused import FusedDCN, FusedDeepFM
    F, Nd, D, B = 11, 5, 8, 300
    extra = dict(cross_params={'num_cross_layer': 3}) if net == 'DCN' else {}
    hu = {'hidden_units': ((H1, 0, False), (H2, 0, False)), 'activation': 'relu'}
    dm, cats = build(F, Nd, D, vocab=30, nets=getattr(deepnets, net), dnn_params=hu, **extra)
    plan = dm.fused_plan()
    assert isinstance(plan, FusedDCN if net == 'DCN' else FusedDeepFM)
    pre, key = ('dcn', 'dcn_dnn') if net == 'DCN' else ('dnn', 'dnn')
    Ly = dm.model.layers_by_name
    d1, d2 = Ly[f'{pre}_dense_1'], Ly[f'{pre}_dense_2']
    assert tuple(d1.kernel.shape) == (F * D + Nd, H1) and tuple(d2.kernel.shape) == (H1, H2)
    idx, dense, y = batch(cats, Nd, B)
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_logit, _ = bridge.oracle_forward(dm, idx, dense, training=True, weights=w)
    ref_loss = R.binary_crossentropy_from_logits(ref_logit, y.double())
    ref_loss.backward()
    dm.model.train()
    ins = [idx.int().to(dev), dense.to(dev)]
    loss, logit = dm.forward_backward(ins, y.to(dev))
    torch.cuda.synchronize()
    assert (logit.double().cpu() - ref_logit).abs().max().item() < 1e-4 * max(1.0, ref_logit.abs().max().item())
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    for i, (a, b) in enumerate([(d1.kernel.grad, w[key][0][0].grad), (d1.bias.grad, w[key][0][1].grad),
                                (d2.kernel.grad, w[key][1][0].grad), (d2.bias.grad, w[key][1][1].grad),
                                (Ly['task_output'].kernel.grad, w['task_output'][0].grad),
                                (Ly['bn_concat_emb_dense'].gamma.grad, w['bn_concat_emb_dense'][0].grad)]):
        assert tuple(a.shape) == tuple(b.shape)
        assert rel(a, b) < 2e-4, f'grad {i}: {rel(a, b)}'
    if net == 'DeepFM':
        assert rel(Ly['dense_logit_dnn_nets'].kernel.grad, w['dense_logit_dnn_nets'].grad) < 2e-4
    table = Ly['emb_categorical_vars_all'].tables[f'd{D}']
    assert rel(table.grad, torch.cat([t.grad for t in w['emb_categorical_vars_all']], 0)) < 2e-4
    # three optimizer steps: fused (padded slabs, one flat Adam launch) vs a twin model on the layer-by-layer path
    twin, _ = build(F, Nd, D, vocab=30, nets=getattr(deepnets, net), dnn_params=hu, **extra)
    twin._fused_plan = None
    with torch.no_grad():
        for (n1, p1), (n2, p2) in zip(dm.model.named_parameters(), twin.model.named_parameters()):
            assert n1 == n2
            p2.copy_(p1)
        for (n1, b1), (n2, b2) in zip(dm.model.named_buffers(), twin.model.named_buffers()):
            assert n1 == n2
            b2.copy_(b1)                       # BatchNormalization's moving statistics (one update ahead by now)
    twin.model.train()
    for step in range(3):
        idx_s, dense_s, y_s = batch(cats, Nd, B, seed=20 + step)
        ins_s = [idx_s.int().to(dev), dense_s.to(dev)]
        l1, _ = dm.train_step(ins_s, y_s.to(dev))
        l2, _ = twin.train_step(ins_s, y_s.to(dev))
        assert abs(float(l1) - float(l2)) < 1e-5
    assert dm._fused_plan is plan and twin._fused_plan is None
    for (n1, p1), (_, p2) in zip(dm.model.named_parameters(), twin.model.named_parameters()):
        assert rel(p1, p2) < 5e-4, n1
    # the pads of the slabs: weights, gradients and Adam moments exactly zero
    C = F * D + Nd
    o = plan.off
    st = dm.optimizer._flat
    for flat in (plan.flat_params, plan.accum, st[2], st[3]):
        W1 = flat[o['dW1']:o['dW1'] + C * 128].view(C, 128)
        W2 = flat[o['dW2']:o['dW2'] + 128 * 64].view(128, 64)
        assert W1[:, H1:].abs().sum().item() == 0 and W2[H1:].abs().sum().item() == 0 and W2[:, H2:].abs().sum().item() == 0
        assert flat[o['db1'] + H1:o['db1'] + 128].abs().sum().item() == 0
        assert flat[o['db2'] + H2:o['db2'] + 64].abs().sum().item() == 0
    # checkpoint round trip through the strided parameter views, optimizer slots included
    path = str(tmp_path / 'narrow.safetensors')
    checkpoint.save_model(dm.model, path, optimizer=dm.optimizer)
    k1, m1 = d1.kernel.detach().clone(), dm.optimizer.state[id(d1.kernel)]['m'].clone()
    assert m1.abs().max().item() > 0
    with torch.no_grad():
        d1.kernel.zero_()
        dm.optimizer.state[id(d1.kernel)]['m'].zero_()
    checkpoint.load_model(dm.model, path, optimizer=dm.optimizer)
    assert torch.equal(d1.kernel.detach(), k1) and torch.equal(dm.optimizer.state[id(d1.kernel)]['m'], m1)
    # predict (layer-by-layer forward on the strided views) agrees with the step's logits
    dm.model.eval()
    twin.model.eval()
    assert (dm.model(ins) - twin.model(ins)).abs().max().item() < 1e-4


@pytest.mark.parametrize('overlap,collectives', [('1', False), ('0', False), ('0', True), ('1', True)])
def test_sharded_embedding_step_single_rank_equals_plain(dev, overlap, collectives, monkeypatch):
    """ShardedEmbeddingStrategy(force=True) on ONE rank drives the whole model-parallel-table step (ids exchange,
    owner gather, all-to-all, fused kernels on the received rows, gradient all-to-all, owner Adam) through RCCL with
    world size 1; after several train steps it must agree with the plain fused step.  overlap = 1 (opt-in): the gradient
    all-to-all starts behind the row-gradient launch and the step's last launch runs beside it (DT_STEP_SKIP_FINISH /
    DT_STEP_FINISH_ONLY); 0 (default): the step as one call, then the exchange."""
    import os
    monkeypatch.setenv('DT_AMD_SHARDED_OVERLAP', overlap)
    import socket
    import torch.distributed as dist
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel, layers as dl
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    from deeptables_amd.parallel import ShardedEmbeddingStrategy
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    st = ShardedEmbeddingStrategy.from_env('nccl')
    st.force = True
    # collectives=True: no world-size-1 shortcuts — ids all_gather_into_tensor, both all_to_all_single calls (with their split
    # lists) and the asynchronous dense all_reduce really go through RCCL, as they do at N > 1
    st.force_collectives = collectives
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0                      # sparse-gradient path (as with 1M-row tables)
    try:
        F, Nd, D, B, vocab = 26, 13, 16, 512, 300
        models = []
        for strategy in (None, st):
            functional.set_seed(12)
            conf = ModelConfig(nets=['linear', 'fm_nets', 'dnn_nets'], fixed_embedding_dim=True,
                               embeddings_output_dim=D, embedding_dropout=0, dense_dropout=0, metrics=['AUC'],
                               distribute_strategy=strategy)
            cats = [CategoricalColumn(f'C{i}', vocab + i, D) for i in range(F)]
            conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])]
            dm = DeepModel('binary', 2, conf, cats, conts)
            dm.build()
            dm.model.train()
            models.append(dm)
        g = torch.Generator().manual_seed(3)
        losses = [[], []]
        for step in range(4):
            idx = torch.stack([torch.randint(0, vocab + i, (B,), generator=g) for i in range(F)], 1)
            if step == 2:
                idx[::17, 3] = vocab + 500             # out-of-range ids: zero row, no update
            dense = torch.randn(B, Nd, generator=g)
            y = (torch.rand(B, 1, generator=g) < 0.3).float()
            for k, dm in enumerate(models):
                l, _ = dm.train_step([idx.int().to(dev), dense.to(dev)], y.to(dev))
                losses[k].append(float(l))
        assert models[1].model._dt_sharded_step is True and models[0].model._dt_sharded_step is False
        assert np.allclose(losses[0], losses[1], atol=1e-6), (losses)
        for (n0, p0), (n1, p1) in zip(models[0].model.named_parameters(), models[1].model.named_parameters()):
            assert (p0 - p1).abs().max().item() < 2e-6, n0
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old
        dist.destroy_process_group()


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
@pytest.mark.parametrize('uniform', [True, False])
def test_data_parallel_step_single_rank_through_rccl_equals_plain(dev, uniform, net):
    """DataParallelStrategy on ONE rank with force_dp + force_collectives: the replicated-table step of N > 1 — in-step dedupe
    kept, segments merged in place, the flat dense all-reduce and the sparse all-gathers really issued through RCCL (world
    size 1), the optimizer's global dedupe over the gathered entries — against the plain fused step, several train steps
    with duplicate and out-of-range ids.  uniform=False: the variable-count exchange (counts all-gather + padding).
    net = 'DCN': BASELINE.json configs[4] — DCN (6 cross layers) under data parallel (run_dt.py:35-44, deepmodel.py:88-103):
    the FusedDCN plan's flat gradient buffer (tower + cross kernels / biases) through the one all-reduce."""
    import os
    import socket
    import torch.distributed as dist
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel, layers as dl
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    from deeptables_amd.parallel import DataParallelStrategy
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    st = DataParallelStrategy.from_env('nccl')
    st.force_dp = True
    st.force_collectives = True
    st.assume_uniform_batches = uniform
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0
    try:
        F, Nd, D, B, vocab = 26, 13, 16, 512, 120            # vocab 120: most lookups are segment members
        models = []
        for strategy in (None, st):
            functional.set_seed(12)
            extra = {'nets': ['dcn_nets'], 'cross_params': {'num_cross_layer': 6}} if net == 'DCN' else \
                {'nets': ['linear', 'fm_nets', 'dnn_nets']}
            conf = ModelConfig(fixed_embedding_dim=True,
                               embeddings_output_dim=D, embedding_dropout=0, dense_dropout=0, metrics=['AUC'],
                               distribute_strategy=strategy, **extra)
            cats = [CategoricalColumn(f'C{i}', vocab + i, D) for i in range(F)]
            conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])]
            dm = DeepModel('binary', 2, conf, cats, conts)
            dm.build()
            dm.model.train()
            models.append(dm)
        assert type(models[1].fused_plan()).__name__ == ('FusedDCN' if net == 'DCN' else 'FusedDeepFM')
        g = torch.Generator().manual_seed(3)
        losses = [[], []]
        for step in range(4):
            idx = torch.stack([torch.randint(0, vocab + i, (B,), generator=g) for i in range(F)], 1)
            if step == 2:
                idx[::17, 3] = vocab + 500
            dense = torch.randn(B, Nd, generator=g)
            y = (torch.rand(B, 1, generator=g) < 0.3).float()
            for k, dm in enumerate(models):
                l, _ = dm.train_step([idx.int().to(dev), dense.to(dev)], y.to(dev))
                losses[k].append(float(l))
        assert np.allclose(losses[0], losses[1], atol=1e-6 if net == 'DeepFM' else 2e-5), losses
        for (n0, p0), (n1, p1) in zip(models[0].model.named_parameters(), models[1].model.named_parameters()):
            assert (p0 - p1).abs().max().item() < 2e-6, n0
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old
        dist.destroy_process_group()


def _emb_keep(seed, B, F, D, rate):
    """the keep-scale [B, F*D] of the fused step's embedding dropout for `seed` (csrc/deepfm.hip emb_drop_hash, numpy)"""
    b = np.arange(B, dtype=np.uint64)[:, None]
    col = np.arange(F * D, dtype=np.uint64)[None, :]
    M = np.uint64(0xFFFFFFFF)
    x = (np.uint64(seed) ^ ((b * np.uint64(0x9E3779B1)) & M) ^ ((col * np.uint64(0x85EBCA77)) & M)) & M
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    thr = np.uint64(int(rate * 4294967296.0))
    return (x >= thr).astype(np.float64) / (1.0 - rate)


def test_fused_step_with_embedding_dropout_matches_oracle_with_the_same_mask(dev):
    """ModelConfig.embedding_dropout = 0.3 is the REFERENCE DEFAULT (config.py:84): the graph stays on the fused plan;
    SpatialDropout1D on (B,1,D) (layers.py:878-880, :900-902) = element dropout with 1/(1-p) scaling"""
    from deeptables_amd import functional
    from deeptables_amd._lib import lib
    from deeptables_amd.models import ModelConfig, DeepModel, deepnets
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    from oracle import bridge, reference_layers as R
    F, Nd, D, B, rate = 26, 13, 16, 96, 0.3
    functional.set_seed(5)
    conf = ModelConfig(nets=deepnets.DeepFM, fixed_embedding_dim=True, embeddings_output_dim=D, embedding_dropout=rate,
                       metrics=['AUC'])
    cats = [CategoricalColumn(f'C{i}', 40 + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(Nd)])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    dm.build()
    plan = dm.fused_plan()
    assert plan is not None, 'the reference-default embedding_dropout must stay on the fused plan'
    idx, dense, y = batch(cats, Nd, B)
    seed = 0x1234567
    plan.drop_seed.fill_(seed)
    thr = int(0.5 * 4294967296.0)
    assert (lib().dt_deepfm_dropout_hash(seed, 3, 17) >= thr) == bool(_emb_keep(seed, 4, 2, 16, 0.5)[3, 17] > 0)
    keep = torch.from_numpy(_emb_keep(seed, B, F, D, rate)).reshape(B, F, D)
    assert abs(float((keep == 0).double().mean()) - rate) < 0.03

    class Masked:               # tables[f][ids] -> looked-up rows times this field's keep-scale
        def __init__(self, t, f):
            self.t, self.f = t, f

        def __getitem__(self, ids):
            return self.t[ids] * keep[:, self.f].reshape(B, 1, D)

    w = bridge.oracle_weights(dm, requires_grad=True)
    tabs = w['emb_categorical_vars_all']
    w['emb_categorical_vars_all'] = [Masked(t, f) for f, t in enumerate(tabs)]
    ref_logit, _ = R.model_forward(w, idx.float(), dense.double(), dm.config.nets, bridge.oracle_config(dm), training=True)
    ref_loss = R.binary_crossentropy_from_logits(ref_logit, y.double())
    ref_loss.backward()
    dm.model.train()
    loss, logit = dm.forward_backward([idx.int().to(dev), dense.to(dev)], y.to(dev))
    torch.cuda.synchronize()
    assert (logit.double().cpu() - ref_logit.detach()).abs().max().item() < 1e-4
    assert abs(float(loss) - float(ref_loss.detach())) < 1e-5
    L = dm.model.layers_by_name
    assert rel(L['dnn_dense_1'].kernel.grad, w['dnn'][0][0].grad) < 2e-4
    assert rel(L['linear_logit'].kernel.grad, w['linear_logit'].grad) < 2e-4
    table = L['emb_categorical_vars_all'].tables[f'd{D}']
    assert rel(table.grad, torch.cat([t.grad for t in tabs], 0)) < 2e-4
    # the step advanced the device seed: the next step draws another mask
    assert int(plan.drop_seed.item()) & 0xFFFFFFFF == (seed * 1664525 + 1013904223) & 0xFFFFFFFF


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
@pytest.mark.parametrize('task', ['binary', 'regression'])
def test_fused_weighted_step_with_dense_dropout_matches_oracle(dev, net, task):
    """Keras sample / class weights (loss = sum_b w_b l_b / B, reference deepmodel.py:114-129 hands both to
    keras fit) and ModelConfig.dense_dropout (Dropout on the continuous inputs, reference deepmodel.py:429-430) stay on the
    fused plan: the oracle, fed the continuous columns with the step's own keep mask, gives the same weighted loss, logits
    and gradients."""
    from deeptables_amd.models import deepnets
    from oracle import bridge, reference_layers as R
    F, Nd, D, B, rate = 9, 6, 8, 200, 0.25
    extra = dict(cross_params={'num_cross_layer': 2}) if net == 'DCN' else {}
    dm, cats = build(F, Nd, D, vocab=30, nets=getattr(deepnets, net), task=task, dense_dropout=rate, **extra)
    plan = dm.fused_plan()
    assert plan is not None and plan.takes_sample_weight and plan.dense_dropout == rate
    idx, dense, y = batch(cats, Nd, B)
    if task == 'regression':
        y = torch.randn(B, 1, generator=torch.Generator().manual_seed(2))
    wts = torch.rand(B, generator=torch.Generator().manual_seed(8)) * 2 + 0.1
    wts[::7] = 0                                                   # class_weight 0 rows take no part
    seed = 0x2468ACE
    plan.drop_seed.fill_(seed)
    assert Nd <= D                                                 # the hash columns of the continuous inputs: F*D + k
    keep = torch.from_numpy(_emb_keep(seed, B, F + 1, D, rate)[:, F * D:F * D + Nd])
    assert 0 < float((keep == 0).double().mean()) < 2 * rate
    w = bridge.oracle_weights(dm, requires_grad=True)
    ref_logit, _ = R.model_forward(w, idx.float(), dense.double() * keep, dm.config.nets, bridge.oracle_config(dm),
                                   training=True)
    z = ref_logit.reshape(B)
    t = y.double().reshape(B)
    per = (torch.clamp(z, min=0) - z * t + torch.log1p(torch.exp(-z.abs()))) if task == 'binary' else (z - t) ** 2
    ref_loss = (per * wts.double()).sum() / B
    ref_loss.backward()
    dm.model.train()
    loss, logit = dm.forward_backward([idx.int().to(dev), dense.to(dev)], y.to(dev), wts.to(dev))
    torch.cuda.synchronize()
    assert dm._step_used_plan
    assert (logit.double().cpu() - ref_logit.detach()).abs().max().item() < 1e-4
    assert abs(float(loss) - float(ref_loss.detach())) < 1e-5 * max(1.0, abs(float(ref_loss)))
    L = dm.model.layers_by_name
    pre = 'dcn' if net == 'DCN' else 'dnn'
    assert rel(L[pre + '_dense_1'].kernel.grad, w['dcn_dnn' if net == 'DCN' else 'dnn'][0][0].grad) < 2e-4
    table = L['emb_categorical_vars_all'].tables[f'd{D}']
    tabs = w['emb_categorical_vars_all']
    assert rel(table.grad, torch.cat([t_.grad for t_ in tabs], 0)) < 2e-4
    assert int(plan.drop_seed.item()) & 0xFFFFFFFF == (seed * 1664525 + 1013904223) & 0xFFFFFFFF


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
def test_fused_weighted_train_steps_equal_the_layer_by_layer_path(dev, net, tower_mode):
    """six weighted TRAIN steps (in-step optimizer included) on the fused plan against a twin without a plan"""
    from deeptables_amd.models import deepnets
    F, Nd, D, B = 26, 13, 16, 700
    extra = dict(cross_params={'num_cross_layer': 3}) if net == 'DCN' else {}
    dm, cats = build(F, Nd, D, vocab=300, nets=getattr(deepnets, net), **extra)
    twin, _ = build(F, Nd, D, vocab=300, nets=getattr(deepnets, net), **extra)
    twin._fused_plan = None
    with torch.no_grad():
        for (n1, p1), (n2, p2) in zip(dm.model.named_parameters(), twin.model.named_parameters()):
            p2.copy_(p1)
    dm.model.train(); twin.model.train()
    g = torch.Generator().manual_seed(5)
    for step in range(6):
        idx_s, dense_s, y_s = batch(cats, Nd, B, seed=70 + step)
        ins_s = [idx_s.int().to(dev), dense_s.to(dev)]
        wts = (torch.rand(B, generator=g) * 3).to(dev)
        l1, _ = dm.train_step(ins_s, y_s.to(dev), wts)
        l2, _ = twin.train_step(ins_s, y_s.to(dev), wts)
        assert dm._step_used_plan and not twin._step_used_plan
        assert abs(float(l1) - float(l2)) < 2e-5, step
    # Adam divides by sqrt(v): an entry whose gradient is ~0 moves by a good part of lr whichever way its last bits fall, and
    # the members of a row segment are summed in the election's order (not the same from run to run).  Six steps of lr =
    # 1e-3 on values of ~5e-2: 5e-4 of the largest entry is 0.4 % of the distance travelled; the split-bf16 backward (two
    # parts, 2^-17 per product) is held to 3e-3 (seen: 1.7e-3 on one table entry, DCN)
    tol = 5e-4 if tower_mode == 'f32' else 3e-3
    for (n1, p1), (_, p2) in zip(dm.model.named_parameters(), twin.model.named_parameters()):
        assert rel(p1, p2) < tol, n1


@pytest.mark.parametrize('net,H1,H2', [('DeepFM', 100, 40), ('DCN', 32, 64)])
def test_weighted_steps_after_a_narrow_tower_plan(dev, net, H1, H2, monkeypatch):
    """once a fused plan with a narrow tower exists, the tower parameters and their Adam moments are
    STRIDED views of zero-padded slabs.  A step with sample weights takes the layer-by-layer path and hands the optimizer
    ordinary autograd gradients: the update must land on the right elements (and keep the pads zero), not treat the views
    as contiguous.  Unweighted (fused) and weighted steps interleaved against a twin that never builds a plan.  (Since
    round 3 the DeepFM / DCN plans take the weights themselves — test_fused_weighted_step_* below; here the plan is told not
    to, which is the path any plan without `takes_sample_weight` goes.)"""
    from deeptables_amd import fused
    from deeptables_amd.models import deepnets
    monkeypatch.setattr(fused.FusedDeepFM, 'takes_sample_weight', False)
    F, Nd, D, B = 11, 5, 8, 300
    extra = dict(cross_params={'num_cross_layer': 3}) if net == 'DCN' else {}
    hu = {'hidden_units': ((H1, 0, False), (H2, 0, False)), 'activation': 'relu'}
    dm, cats = build(F, Nd, D, vocab=30, nets=getattr(deepnets, net), dnn_params=hu, **extra)
    twin, _ = build(F, Nd, D, vocab=30, nets=getattr(deepnets, net), dnn_params=hu, **extra)
    twin._fused_plan = None
    with torch.no_grad():
        for (n1, p1), (n2, p2) in zip(dm.model.named_parameters(), twin.model.named_parameters()):
            p2.copy_(p1)
    dm.model.train(); twin.model.train()
    g = torch.Generator().manual_seed(5)
    # a weighted step first: no plan yet, and none is built by it
    idx_s, dense_s, y_s = batch(cats, Nd, B, seed=40)
    ins_s = [idx_s.int().to(dev), dense_s.to(dev)]
    w0 = (torch.rand(B, generator=g) + 0.5).to(dev)
    l1, _ = dm.train_step(ins_s, y_s.to(dev), w0)
    l2, _ = twin.train_step(ins_s, y_s.to(dev), w0)
    assert not dm._step_used_plan, 'a weighted step ran on a plan that does not take weights'
    assert abs(float(l1) - float(l2)) < 1e-5
    for step in range(6):
        idx_s, dense_s, y_s = batch(cats, Nd, B, seed=50 + step)
        ins_s = [idx_s.int().to(dev), dense_s.to(dev)]
        wts = (torch.rand(B, generator=g) + 0.5).to(dev) if step % 2 else None      # fused, weighted, fused, ...
        l1, _ = dm.train_step(ins_s, y_s.to(dev), wts)
        l2, _ = twin.train_step(ins_s, y_s.to(dev), wts)
        assert abs(float(l1) - float(l2)) < 2e-5, step
    plan = dm._fused_plan
    assert plan is not None and twin._fused_plan is None
    assert not dm.model.layers_by_name[('dcn' if net == 'DCN' else 'dnn') + '_dense_1'].kernel.data.is_contiguous() or H1 == 128
    for (n1, p1), (_, p2) in zip(dm.model.named_parameters(), twin.model.named_parameters()):
        assert rel(p1, p2) < 5e-4, n1
    C = F * D + Nd
    o = plan.off
    st = dm.optimizer._flat
    for flat in (plan.flat_params, st[2], st[3]):
        W1 = flat[o['dW1']:o['dW1'] + C * 128].view(C, 128)
        W2 = flat[o['dW2']:o['dW2'] + 128 * 64].view(128, 64)
        assert W1[:, H1:].abs().sum().item() == 0 and W2[H1:].abs().sum().item() == 0 and W2[:, H2:].abs().sum().item() == 0
        assert flat[o['db1'] + H1:o['db1'] + 128].abs().sum().item() == 0
        assert flat[o['db2'] + H2:o['db2'] + 64].abs().sum().item() == 0


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
@pytest.mark.parametrize('vocab,B,F,D,drop', [(30, 256, 26, 16, 0.0), (5000, 1000, 26, 16, 0.0), (5000, 513, 26, 16, 0.3),
                                              (200000, 4096, 26, 16, 0.0), (3000, 300, 7, 32, 0.0), (900, 77, 5, 8, 0.0),
                                              # beyond 8192 rows (round 5): the election walks a field's lookups in chunks and
                                              # places its segments in per-field regions — rows looked up ~300 times each, ~4
                                              # times each, mostly once; a ragged chunk, D = 32
                                              (30, 9000, 26, 16, 0.0), (5000, 20000, 26, 16, 0.0), (200000, 16500, 7, 32, 0.0),
                                              # three rows per field at B = 60000: every row's ~20 K lookups sit in ONE hash
                                              # partition — more than an election block's 16384-entry lookup list holds (the
                                              # spill paths: BENCH b65536 with Zipf ids met them first and got them wrong)
                                              (3, 60000, 7, 16, 0.0)])
def test_rows_in_step_equals_the_separate_optimizer_step(dev, monkeypatch, vocab, B, F, D, drop, net):
    """DeepModel.train_step on the pipelined DeepFM step applies Keras Adam to the table rows looked up once inside the
    step (dt_deepfm_train_step_adam, k_wgrad_rows) and leaves only the segments to the optimizer launch: same tables,
    slots, dense parameters and step count as forward_backward + optimizer.step over every lookup's gradient row — from
    many duplicates (vocab 30: almost every lookup is a segment member) to almost none, ragged tiles, with dropout."""
    from deeptables_amd.models import layers as L
    from oracle import headline
    monkeypatch.setattr(L, 'DENSE_GRAD_MAX_ELEMS', 0)          # row-sparse ("lazy") update also on these small tables
    from deeptables_amd.models import deepnets
    extra = dict(nets=deepnets.DCN, cross_params={'num_cross_layer': 3 if D == 8 else 6}) if net == 'DCN' else {}
    dm, cats = build(F, 13, D, vocab=vocab, embedding_dropout=drop, **extra)
    assert type(dm.fused_plan()).__name__ == 'Fused' + net
    for seed in (5, 6):                                        # two batches: the second one starts from non-zero slots
        idx, dense, y = batch(cats, 13, B, seed=seed)
        # repeated: the first version of the finishing launch read gamma / beta while other blocks of the SAME launch updated
        # them — a race that showed up in one run out of a few (tools/r3/dbg_ab2.py tells which tensors moved)
        for rep in range(4):
            # (ONE step per comparison, repeated: after a first step the two paths' tables and dense weights differ by an ulp
            # here and there — the update rule runs in two kernels — and among the millions of relu units of a batch one now and
            # then sits within that ulp of its kink and takes the other derivative in one of the paths: a whole sample's gradient
            # term, 1e-5 .. 1e-3 of the slots' largest entry.  Rounds 3-4 compared after 2-3 steps and met this about once in
            # ~50 comparisons (B = 16500 DeepFM, B = 4096 DCN this round); a single step starts both paths from identical
            # state, so anything beyond the update rule's own rounding is a defect — a race shows up across the repetitions.)
            res = headline.check_rows_in_step(dm, (idx.to(torch.int32).to(dev), dense.to(dev), y.to(dev)), steps=1)
            assert headline.rows_in_step_ok(res), str((rep, sorted(res.items())))
        dm.train_step([idx.to(torch.int32).to(dev), dense.to(dev)], y.to(dev))   # move on (in-step path)


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
def test_in_step_optimizer_matches_oracle_adam_over_three_steps(dev, net):
    """small sizes (most lookups are segment members, every row warm from step 2 on): the step with the optimizer inside
    its launches against oracle.headline.check_in_step_vs_oracle — R.keras_adam_step on the oracle's own gradient and its
    own running slots"""
    from oracle import headline
    from deeptables_amd.models import deepnets, layers as dl
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0
    try:
        F, Nd, D, B = 26, 13, 16, 512
        extra = dict(cross_params={'num_cross_layer': 4}) if net == 'DCN' else {}
        dm, cats = build(F, Nd, D, vocab=300, nets=getattr(deepnets, net), **extra)
        batches = []
        for s in range(3):
            idx, dense, y = batch(cats, Nd, B, seed=20 + s)
            batches.append((idx.int().to(dev), dense.to(dev), y.to(dev)))
        res = headline.check_in_step_vs_oracle(dm, batches)
        assert res['ok'] and res['steps_counted'] == 3 and res['warm_rows'] > 1000, str(sorted(res.items()))
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
@pytest.mark.parametrize('vocab,B', [(40, 512), (3000, 1000), (200000, 4096)])
def test_chained_steps_equal_plain_steps(dev, monkeypatch, net, vocab, B, tower_mode):
    """Chained steps (include/dt_hip.h DT_STEP_PREPARED, csrc/deepfm.hip StepNext): step n, given step n + 1's ids, packs their
    rows (kernel A), runs their election on the weight-gradient launch's matrix waves and writes the tile kernel's bf16 weight
    layouts from the weights it has just updated; step n + 1 then runs WITHOUT its prep launch.  Four chained steps through
    `FusedDeepFM.run(next_ids=..., prepared=...)` (eager: no graph) must leave the tables, slots and dense parameters of
    four plain `train_step`s — with duplicate ids (vocab 40: almost every lookup is a segment member), out-of-range ids and a
    ragged last tile (B = 1000)."""
    from deeptables_amd.models import layers as L, deepnets
    monkeypatch.setattr(L, 'DENSE_GRAD_MAX_ELEMS', 0)
    extra = dict(nets=deepnets.DCN, cross_params={'num_cross_layer': 5}) if net == 'DCN' else {}
    F, Nd, D = 26, 13, 16
    plain, cats = build(F, Nd, D, vocab=vocab, **extra)
    chained, _ = build(F, Nd, D, vocab=vocab, **extra)
    plan = chained.fused_plan()
    assert type(plan).__name__ == 'Fused' + net
    if tower_mode == 'f32':          # the exact-fp32 tile kernel reads the prep launch's fp32 layouts: such steps are not chained
        assert not plan.can_chain(B)
        return
    assert plan.can_chain(B)
    steps = []
    for s in range(4):
        idx, dense, y = batch(cats, Nd, B, seed=70 + s)
        if s == 2:
            idx[::13, 4] = vocab + 999                       # out-of-range ids: zero row, no update
        steps.append((idx.to(torch.int32).to(dev), dense.to(dev), y.to(dev)))
    plain.model.train(); chained.model.train()
    for idx, dense, y in steps:
        plain.train_step([idx, dense], y)
    opt = chained.optimizer
    for s, (idx, dense, y) in enumerate(steps):
        nxt = (steps[s + 1][0], s + 1) if s + 1 < len(steps) else None
        opt.zero_grad(flat=False)
        plan.run(idx, dense, y, apply_rows=True, slot=s, next_ids=nxt, prepared=s >= 1)
        opt.step()
    torch.cuda.synchronize()
    assert plain.optimizer.t == chained.optimizer.t == 4
    for (n0, p0), (n1, p1) in zip(plain.model.named_parameters(), chained.model.named_parameters()):
        assert n0 == n1 and (p0 - p1).abs().max().item() <= 2e-6, (n0, (p0 - p1).abs().max().item())
    ta, tb = plain.optimizer._st(plain.fused_plan().emb.tables['d16'], rows=True), opt._st(plan.emb.tables['d16'], rows=True)
    assert (ta['m'] - tb['m']).abs().max().item() <= 1e-6 * max(1.0, ta['m'].abs().max().item())
    # a step that cannot be chained says so instead of running prepared on nothing
    from deeptables_amd import _lib
    assert _lib.lib().dt_deepfm_step_chains(B, F, D, Nd, 2 | _lib.DT_STEP_TOWER_X3) == 1
    assert _lib.lib().dt_deepfm_step_chains(B, F, D, Nd, 2) == 0                      # exact-fp32 tower: k_prep's layouts
    assert _lib.lib().dt_deepfm_step_chains(16384, F, D, Nd, 2 | _lib.DT_STEP_TOWER_X3) == 0
