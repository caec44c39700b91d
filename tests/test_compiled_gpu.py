# -*- coding:utf-8 -*-
"""GPU: `DeepModel.fit(steps_per_execution=k)` — k consecutive train steps per captured hipGraph over static input slots
(deeptables_amd/compiled.py; Keras' `model.compile(steps_per_execution=...)`, reference deepmodel.py:319-346 driven by
`model.fit`, deepmodel.py:114-129) — must train exactly like the eager step-by-step `fit`: same batches for the same
seed, same weights after several shuffled epochs, same loss curve and metrics."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

F, ND, D, V = 8, 3, 16, 40


def _frame(n, seed=0, task='binary'):
    rng = np.random.RandomState(seed)
    df = pd.DataFrame({f'C{i}': rng.randint(0, V + i, n) for i in range(F)})
    for j in range(ND):
        df[f'I{j}'] = rng.randn(n).astype(np.float32)
    y = (rng.rand(n) < 0.3).astype(np.float32) if task == 'binary' else rng.randn(n).astype(np.float32)
    return df, y


def _model(net, task='binary', seed=4, **extra):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel, deepnets
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    functional.set_seed(seed)
    conf = ModelConfig(nets=getattr(deepnets, net), fixed_embedding_dim=True, embeddings_output_dim=D, embedding_dropout=0,
                       metrics=['AUC'] if task == 'binary' else ['mse'], **extra)
    cats = [CategoricalColumn(f'C{i}', V + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(ND)])]
    dm = DeepModel(task, 2 if task == 'binary' else 1, conf, cats, conts)
    dm.build()
    return dm


def _fit(dm, df, y, spe, epochs=3, **kw):
    torch.manual_seed(123)                      # the epoch permutations come from the device generator
    torch.cuda.manual_seed_all(123)
    return dm.fit(df, y, batch_size=64, epochs=epochs, verbose=0, validation_split=0, shuffle=True,
                  steps_per_execution=spe, **kw)


def _same(a, b, tol=2e-6):
    for (n0, p0), (n1, p1) in zip(a.model.named_parameters(), b.model.named_parameters()):
        assert n0 == n1
        err = (p0 - p1).abs().max().item()
        assert err <= tol, (n0, err)


@pytest.mark.parametrize('sparse', [True, False])
@pytest.mark.parametrize('net,task', [('DeepFM', 'binary'), ('DCN', 'binary'), ('DeepFM', 'regression')])
def test_graphed_fit_trains_like_eager_fit(dev, net, task, sparse):
    from deeptables_amd.models import layers as dl
    old = dl.DENSE_GRAD_MAX_ELEMS
    if sparse:
        dl.DENSE_GRAD_MAX_ELEMS = 0             # row-sparse tables: the optimizer runs inside the fused step's launches
    try:
        df, y = _frame(64 * 23 + 17, task=task)     # 23 steps per epoch: two replays of 10 + three eager steps
        extra = dict(cross_params={'num_cross_layer': 3}) if net == 'DCN' else {}
        eager, graphed = _model(net, task, **extra), _model(net, task, **extra)
        # The dense-gradient path (sparse False) is compared after ONE epoch: its table gradient is a float-atomics scatter whose
        # order differs from run to run, Adam's normalisation amplifies those ulps to ~1e-6 of a weight within ~50 steps, and
        # from then on every few runs some relu unit sits inside that noise and the two fits take different derivatives for one
        # sample — a bimodal 1e-3 in the tables at a fixed step (tools/r6/dbg_lockstep2.py: two EAGER fits fork the same way,
        # 0-10 times in 16 runs depending on the seed, with the fp32 and with the split-bf16 weight-gradient kernels alike).
        # Twenty-three steps (two replays of ten + three eager steps) exercise the same loop logic below that horizon.
        epochs = 3 if sparse else 1
        h0 = _fit(eager, df, y, 1, epochs=epochs)
        h1 = _fit(graphed, df, y, 10, epochs=epochs)
        assert eager.compiled_loop is None
        loop = graphed.compiled_loop
        assert loop is not None and loop.graph is not None and loop.k == 10
        assert not loop.preelected                   # (opt-in, next test)
        # row-sparse tables: the steps of an execution are chained (step i runs step i + 1's election and weight layouts inside
        # its own launches: no prep launch from the second step on) — same weights as the eager steps, which prepare themselves
        assert loop.chained == sparse
        assert type(graphed.fused_plan()).__name__ == ('FusedDCN' if net == 'DCN' else 'FusedDeepFM')
        # small tables keep Keras' dense table gradient: its scatter adds the lookups' rows with float atomics, whose order
        # differs from run to run (ulps of a gradient; Adam's g / (sqrt(v) + eps) turns a few of them into 1e-5 of a weight
        # now and then) — the row-sparse path (segments, no atomics) is held to 2e-6
        tol = 2e-6 if sparse else 5e-5
        _same(eager, graphed, tol=tol)
        assert eager.optimizer.t == graphed.optimizer.t == epochs * 23
        assert np.allclose(h0.history['loss'], h1.history['loss'], atol=tol), (h0.history, h1.history)
        key = 'auc' if task == 'binary' else 'mse'
        assert np.allclose(h0.history[key], h1.history[key], atol=1e-5), (h0.history, h1.history)
        # predictions of the graphed model agree with the eager one's
        assert np.abs(eager.predict(df.iloc[:200]) - graphed.predict(df.iloc[:200])).max() < (1e-5 if sparse else 2e-4)
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old


def test_auto_steps_per_execution_compiles_fused_plans_only(dev):
    """the default ('auto'): a graph with a whole-step plan trains through the compiled loop, others step eagerly"""
    df, y = _frame(64 * 40)
    dm = _model('DeepFM')
    dm.fit(df, y, batch_size=64, epochs=1, verbose=0, validation_split=0)
    assert dm.compiled_loop is not None and dm.compiled_loop.graph is not None and dm.compiled_loop.k == 20
    other = _model('AFM')
    other.fit(df, y, batch_size=64, epochs=1, verbose=0, validation_split=0)
    assert other.compiled_loop is None
    short = _model('DeepFM')
    short.fit(df.iloc[:64 * 3], y[:64 * 3], batch_size=64, epochs=1, verbose=0, validation_split=0)
    assert short.compiled_loop is None              # three steps per epoch: nothing to replay


def test_graphed_fit_with_sample_weights_and_validation(dev):
    """Keras sample_weight x class_weight ride as the feed's last label column into the captured steps; validation runs
    between the replayed epochs"""
    from deeptables_amd.models import layers as dl
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0
    try:
        df, y = _frame(64 * 30)
        w = np.random.RandomState(5).rand(len(df)).astype(np.float32) + 0.5
        eager, graphed = _model('DeepFM'), _model('DeepFM')
        dv, yv = _frame(300, seed=9)
        h0 = _fit(eager, df, y, 1, sample_weight=w, class_weight={0: 1.0, 1: 2.0}, validation_data=(dv, yv))
        h1 = _fit(graphed, df, y, 5, sample_weight=w, class_weight={0: 1.0, 1: 2.0}, validation_data=(dv, yv))
        assert graphed.compiled_loop is not None and graphed.compiled_loop.k == 5
        _same(eager, graphed)
        assert np.allclose(h0.history['loss'], h1.history['loss'], atol=2e-6)
        assert np.allclose(h0.history['val_loss'], h1.history['val_loss'], atol=2e-6)
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old


def test_layer_by_layer_graph_is_capturable_on_request(dev):
    """an explicit steps_per_execution also captures graphs without a whole-step plan (the autograd path through the
    layer kernels): xDeepFM, four steps per replay"""
    df, y = _frame(64 * 12)
    cin = dict(cin_params={'cross_layer_size': (16, 16), 'activation': 'relu', 'use_residual': False, 'use_bias': False,
                           'direct': False, 'reduce_D': False})
    eager, graphed = _model('xDeepFM', **cin), _model('xDeepFM', **cin)
    h0 = _fit(eager, df, y, 1)
    h1 = _fit(graphed, df, y, 4)
    assert graphed.compiled_loop is not None and graphed.compiled_loop.graph is not None
    _same(eager, graphed, tol=5e-6)
    assert np.allclose(h0.history['loss'], h1.history['loss'], atol=5e-6)


def test_autoint_graph_with_pending_normalisations_is_capturable(dev):
    """the AutoInt graph — interacting layers whose BatchNormalization stays pending for the consumer (functional.Model's
    peephole: layer -> layer, layer -> Flatten -> Dense(1) with the rank-one gradient) — through the captured loop: same
    weights and loss curve as the eager fit, remainder steps included (12 steps per epoch, five per replay)"""
    att = dict(autoint_params={'num_attention': 2, 'num_heads': 2, 'dropout_rate': 0, 'use_residual': True})
    df, y = _frame(64 * 12)
    eager, graphed = _model('AutoInt', **att), _model('AutoInt', **att)
    assert len(graphed.model._defer_norm) == 2
    h0 = _fit(eager, df, y, 1, epochs=2)
    h1 = _fit(graphed, df, y, 5, epochs=2)
    assert graphed.compiled_loop is not None and graphed.compiled_loop.graph is not None
    _same(eager, graphed, tol=5e-6)
    assert np.allclose(h0.history['loss'], h1.history['loss'], atol=5e-6)


def test_compiled_loop_on_a_device_feed_reuses_its_graph(dev):
    """bench.py's use: `fit(feed)` on a ready device-resident TableBatches; the second call replays the first call's graph"""
    from deeptables_amd.training import TableBatches
    from deeptables_amd.models import layers as dl
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0
    try:
        g = torch.Generator().manual_seed(1)
        n = 64 * 20
        idx = torch.stack([torch.randint(0, V + i, (n,), generator=g) for i in range(F)], 1).int().to(dev)
        dense = torch.randn(n, ND, generator=g).to(dev)
        y = (torch.rand(n, 1, generator=g) < 0.3).float().to(dev)
        feed = TableBatches.from_device([idx, dense], ['cat', 'cont'], y, y_ndim=2)
        dm = _model('DeepFM')
        dm.fit(feed, batch_size=64, epochs=1, verbose=0, steps_per_execution=5)
        loop = dm.compiled_loop
        t1 = dm.optimizer.t
        dm.fit(feed, batch_size=64, epochs=2, verbose=0, steps_per_execution=5)
        assert dm.compiled_loop is loop and dm.optimizer.t == t1 + 40
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old


def test_preelected_steps_train_like_eager_steps(dev, monkeypatch):
    """DT_AMD_PREELECT=1: the ids-only half of steps 2..k of an execution (packed rows, the election of the rows looked up
    several times: dt_deepfm_preelect) runs ahead of them on a forked branch of the captured graph, the steps skip it
    (DT_STEP_PREELECTED).  Same weights as eager steps; vocab 40: most lookups are segment members"""
    from deeptables_amd.models import layers as dl
    monkeypatch.setenv('DT_AMD_PREELECT', '1')
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0
    try:
        for net in ('DeepFM', 'DCN'):
            df, y = _frame(64 * 23 + 17)
            extra = dict(cross_params={'num_cross_layer': 3}) if net == 'DCN' else {}
            eager, graphed = _model(net, **extra), _model(net, **extra)
            h0 = _fit(eager, df, y, 1)
            h1 = _fit(graphed, df, y, 10)
            assert graphed.compiled_loop.preelected and graphed.compiled_loop.k == 10
            _same(eager, graphed)
            assert np.allclose(h0.history['loss'], h1.history['loss'], atol=2e-6)
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old


@pytest.mark.parametrize('whole', [True, False])
@pytest.mark.parametrize('uniform', [False, True])
def test_compiled_data_parallel_fit_trains_like_the_eager_one(dev, monkeypatch, uniform, whole):
    """under data parallel the loop captures `optimizer.step()` into its own graph on the third step.  That is
    only valid while the exchanged sparse gradients keep their addresses — with a DEFAULT strategy (assume_uniform_batches
    False) `allgather_sparse` used to hand back freshly allocated tensors every step, so the replays updated the tables from
    whatever the capture step's (since recycled) buffers held.  World size 1 through RCCL (force_dp + force_collectives: the
    flat all-reduce and the sparse all-gathers are really issued), row-sparse tables; the compiled fit must leave the weights
    of the step-by-step fit under the same strategy.
    Round 6 (`whole`): the default structure captures k WHOLE steps — forward + backward, the RCCL exchange and the optimizer —
    into one hipGraph (compiled.CompiledTrainLoop._capture_dp_whole); DT_AMD_DP_GRAPH=0 keeps round 5's
    [graph] -> eager exchange -> [optimizer graph].  Both must train like the eager fit."""
    monkeypatch.setenv('DT_AMD_DP_GRAPH', '1' if whole else '0')
    import os
    import socket
    import torch.distributed as dist
    from deeptables_amd.models import layers as dl
    from deeptables_amd.parallel import DataParallelStrategy
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    old = dl.DENSE_GRAD_MAX_ELEMS
    dl.DENSE_GRAD_MAX_ELEMS = 0
    st_made = []
    try:
        df, y = _frame(64 * 12)

        def strategy():
            st = DataParallelStrategy.from_env('nccl') if not st_made else DataParallelStrategy(device=st_made[0].device)
            st.force_dp = True
            st.force_collectives = True
            st.assume_uniform_batches = uniform
            st_made.append(st)
            return st
        eager = _model('DeepFM', distribute_strategy=strategy())
        graphed = _model('DeepFM', distribute_strategy=strategy())
        _fit(eager, df, y, 1)
        _fit(graphed, df, y, 10)
        assert eager.compiled_loop is None
        loop = graphed.compiled_loop
        assert loop is not None and loop.dp and loop.graph is not None
        if whole:
            assert loop.dp_graph and loop.k == 10 and loop.opt_graph is None   # exchange + optimizer are inside the graph
        else:
            assert not loop.dp_graph and loop.k == 1
            assert loop.opt_graph is not None and not loop._opt_graph_refused  # the persistent exchange path was taken
        assert graphed.config.distribute_strategy.assume_uniform_batches is uniform   # (restored after every exchange)
        _same(eager, graphed, tol=2e-6)
        assert eager.optimizer.t == graphed.optimizer.t == 3 * 12
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = old
        if dist.is_initialized():
            dist.destroy_process_group()
