# -*- coding:utf-8 -*-
"""GPU, two PROCESSES on one device, gloo (collectives staged through the host): the whole DeepModel.train_step under
both data-parallel strategies — replicated tables + dense all-reduce + sparse all-gather (the reference's
MirroredStrategy shape, deepmodel.py:88-103, run_dt.py:42-44) and row-owned tables — must leave every rank with the
weights a single process gets by emulating the same semantics in sequence (per-replica BatchNormalization statistics,
mean of the replicas' dense gradients, sum of their row gradients / W, one Keras-Adam step)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

F, ND, D, V, B, W = 6, 3, 16, 50, 64, 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(strategy, net='DeepFM'):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel, deepnets
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    from deeptables_amd.models import layers as dl
    dl.DENSE_GRAD_MAX_ELEMS = 0                  # row-sparse table gradients, as at benchmark size
    functional.set_seed(11)
    # net = 'DCN': BASELINE.json configs[4] — DCN with 6 cross layers under data parallel (run_dt.py:35-44)
    extra = dict(nets=deepnets.DCN, cross_params={'num_cross_layer': 6}) if net == 'DCN' else dict(nets=deepnets.DeepFM)
    conf = ModelConfig(fixed_embedding_dim=True, embeddings_output_dim=D, embedding_dropout=0,
                       metrics=['AUC'], distribute_strategy=strategy, **extra)
    cats = [CategoricalColumn(f'C{i}', V + i, D) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(ND)])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    dm.build(torch.device('cuda', 0))
    return dm


def _batches(steps):
    g = torch.Generator().manual_seed(3)
    out = []
    for _ in range(steps):
        per_rank = []
        for _r in range(W):
            idx = torch.randint(0, V, (B, F), generator=g, dtype=torch.int32)
            idx[: B // 4] = idx[B // 4: B // 2]          # duplicates inside a rank and (below) across ranks
            dense = torch.randn(B, ND, generator=g)
            y = (torch.rand(B, 1, generator=g) < 0.3).float()
            per_rank.append((idx, dense, y))
        per_rank[1][0][:8] = per_rank[0][0][:8]
        out.append(per_rank)
    return out


def _weights(dm):
    return {n: p.detach().cpu().clone() for n, p in dm.model.named_parameters()}


def _worker(rank, port, kind, q, net='DeepFM'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(W), LOCAL_RANK='0')
    torch.cuda.set_device(0)
    from deeptables_amd.parallel import DataParallelStrategy, ShardedEmbeddingStrategy
    cls = ShardedEmbeddingStrategy if kind == 'sharded' else DataParallelStrategy
    st = cls.from_env('gloo')
    st.device = torch.device('cuda', 0)
    st.assume_uniform_batches = True
    dm = _model(st, net)
    st.broadcast_parameters(dm.model)
    dm.model.train()
    dev = st.device
    for per_rank in _batches(3):
        idx, dense, y = (t.to(dev) for t in per_rank[rank])
        dm.train_step([idx, dense], y)
    if kind == 'sharded':
        st.sync_tables(dm.model.layers_by_name['emb_categorical_vars_all'])
    torch.cuda.synchronize()
    q.put((rank, {k: v.numpy() for k, v in _weights(dm).items()}))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _emulate(net='DeepFM'):
    """the same three steps in ONE process: each replica's forward/backward in turn on the shared weights, gradients
    combined as the exchange does, one optimizer step"""
    from deeptables_amd.ops import SparseRowGrad
    dm = _model(None, net)
    dm.model.train()
    dev = torch.device('cuda', 0)
    emb = dm.model.layers_by_name['emb_categorical_vars_all']
    bn = dm.model.layers_by_name['bn_concat_emb_dense']
    dense_params = [p for n, p in dm.model.named_parameters() if 'tables' not in n]
    for per_rank in _batches(3):
        mm0, mv0 = bn.moving_mean.clone(), bn.moving_variance.clone()
        gsum = [torch.zeros_like(p) for p in dense_params]
        rows, vals = [], []
        for r in range(W):
            bn.moving_mean.copy_(mm0); bn.moving_variance.copy_(mv0)        # replica-local statistics (rank 0's are kept)
            idx, dense, y = (t.to(dev) for t in per_rank[r])
            dm.forward_backward([idx, dense], y)
            for a, p in zip(gsum, dense_params):
                a += p.grad
            for sg in emb.sparse_grads['d16']:
                r_, v_ = sg.expanded()            # rows looked up several times travel as segments in one process
                rows.append(r_.reshape(-1).clone())
                vals.append(v_.reshape(-1, D).clone() / W)
            if r == 0:
                mm_keep, mv_keep = bn.moving_mean.clone(), bn.moving_variance.clone()
        bn.moving_mean.copy_(mm_keep); bn.moving_variance.copy_(mv_keep)
        flat = getattr(dm.model, '_dt_flat_grad', None)
        for a, p in zip(gsum, dense_params):
            p.grad.copy_(a / W)
        emb.sparse_grads['d16'] = [SparseRowGrad(torch.cat(rows), torch.cat(vals))]
        dm.optimizer.step()
    torch.cuda.synchronize()
    return _weights(dm)


@pytest.mark.parametrize('net', ['DeepFM', 'DCN'])
@pytest.mark.parametrize('kind', ['replicated', 'sharded'])
def test_two_process_train_steps_match_single_process_emulation(dev, kind, net):
    from deeptables_amd.models import layers as dl
    keep = dl.DENSE_GRAD_MAX_ELEMS
    try:
        want = _emulate(net)              # (_model switches this process to row-sparse table gradients)
    finally:
        dl.DENSE_GRAD_MAX_ELEMS = keep
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, kind, q, net)) for r in range(W)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(W))
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in range(W):
        for name, w in want.items():
            g = torch.from_numpy(got[r][name])
            err = (g - w).abs().max().item()
            assert err <= 2e-6 + 1e-5 * w.abs().max().item(), (kind, net, r, name, err)
