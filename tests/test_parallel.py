# -*- coding:utf-8 -*-
"""CPU, world_size 2, gloo: the data-parallel exchange (flat dense all-reduce + sparse all-gather)
of deeptables_amd.parallel.DataParallelStrategy — the role tf.distribute.MirroredStrategy plays for
the reference (deepmodel.py:88-103).  Compute kernels are not involved: gradients are synthetic."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from deeptables_amd.parallel import DataParallelStrategy
    from deeptables_amd.ops import SparseRowGrad
    from deeptables_amd.models.layers import MultiColumnEmbedding
    st = DataParallelStrategy.from_env('gloo')
    assert st.world_size == world and st.rank == rank

    # --- sharding: AutoShardPolicy.DATA ---
    X = np.arange(10).reshape(10, 1)
    Xs, ys = st.shard(X, np.arange(10))
    assert list(Xs[:, 0]) == list(range(rank, 10, world)) and list(ys) == list(range(rank, 10, world))
    # uneven frames are truncated to the same length on every rank (a rank with one more step would hang in its
    # collectives): 11 rows over 2 ranks -> 5 each
    X11 = np.arange(11).reshape(11, 1)
    Xo, yo = st.shard(X11, np.arange(11))
    assert len(Xo) == 11 // world and list(Xo[:, 0]) == list(range(rank, 11, world))[:11 // world] and len(yo) == len(Xo)
    # the validation split is ONE partition of the frame on all ranks
    np.random.seed(100 + rank)                       # ranks deliberately disagree on their local RNG
    perm = st.shared_permutation(37)
    gathered = [None] * world
    dist.all_gather_object(gathered, perm.tolist())
    assert all(g == gathered[0] for g in gathered) and sorted(gathered[0]) == list(range(37))

    # --- parameters are broadcast from rank 0 ---
    lin = torch.nn.Linear(4, 3)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    st.broadcast_parameters(lin)
    assert torch.all(lin.weight == 1.0)

    # --- dense: one flat bucket, mean over ranks ---
    lin.weight.grad = torch.full_like(lin.weight, float(rank))        # 0 and 1 -> 0.5
    lin.bias.grad = torch.arange(3, dtype=torch.float32) * (rank + 1)  # x1 and x2 -> x1.5
    n = st.allreduce_dense(list(lin.parameters()))
    assert n == 15
    assert torch.allclose(lin.weight.grad, torch.full_like(lin.weight, 0.5))
    assert torch.allclose(lin.bias.grad, torch.arange(3, dtype=torch.float32) * 1.5)

    # --- sparse: ragged all-gather of (rows, values), values pre-divided by world size ---
    nloc = 3 + rank                                                    # ragged on purpose
    rows = torch.arange(nloc, dtype=torch.int64) + 10 * rank
    vals = torch.ones(nloc, 4) * (rank + 1)
    out = st.allgather_sparse(SparseRowGrad(rows, vals))
    valid = out.rows >= 0
    assert int(valid.sum()) == 3 + 4
    dense = torch.zeros(32, 4)
    dense.index_add_(0, out.rows[valid], out.values[valid])
    expect = torch.zeros(32, 4)
    expect[0:3] = 1.0 / world
    expect[10:14] = 2.0 / world
    assert torch.allclose(dense, expect)

    # --- uniform fast path (fixed batch per rank): no count exchange ---
    st.assume_uniform_batches = True
    out_u = st.allgather_sparse(SparseRowGrad(torch.tensor([rank, 5 + rank]), torch.ones(2, 4) * (rank + 1)))
    assert out_u.rows.tolist() == [0, 5, 1, 6]
    assert torch.allclose(out_u.values, torch.tensor([[0.5] * 4, [0.5] * 4, [1.0] * 4, [1.0] * 4]))
    st.assume_uniform_batches = False

    # --- exchange_gradients walks the model: dense params + every MultiColumnEmbedding's sparse grads ---
    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = MultiColumnEmbedding([5, 6], [4, 4])
            self.emb.build((None, 2))
            self.fc = torch.nn.Linear(2, 1)

    m = Tiny()
    m.fc.weight.grad = torch.full_like(m.fc.weight, float(rank))
    m.fc.bias.grad = torch.full_like(m.fc.bias, 2.0)
    m.emb.add_sparse_grad('d4', SparseRowGrad(torch.tensor([rank, 7]), torch.ones(2, 4)))
    st.exchange_gradients(m)
    assert torch.allclose(m.fc.weight.grad, torch.full_like(m.fc.weight, 0.5))
    g = m.emb.sparse_grads['d4'][0]
    assert sorted(g.rows.tolist()) == [0, 1, 7, 7] and torch.allclose(g.values, torch.full((4, 4), 0.5))
    assert m.emb.tables['d4'].grad is None          # the table itself is never densified / all-reduced

    # --- fused-step path: all dense gradients live in ONE flat buffer that is all-reduced in place ---
    m2 = Tiny()
    flat = torch.arange(3, dtype=torch.float32) * (rank + 1)        # [0,1,2]*1 and *2 -> mean *1.5
    m2._dt_flat_grad = flat
    m2.fc.weight.grad = flat[:2].view(1, 2)                         # views of the flat buffer (like fused.py)
    m2.fc.bias.grad = flat[2:3]
    m2.emb.tables['d4'].grad = torch.full_like(m2.emb.tables['d4'], float(rank))   # a small table's dense gradient
    m2.emb.add_sparse_grad('d4', SparseRowGrad(torch.tensor([3 + rank]), torch.ones(1, 4)))
    st.exchange_gradients(m2)
    assert torch.allclose(flat, torch.arange(3, dtype=torch.float32) * 1.5)
    assert torch.allclose(m2.fc.weight.grad, torch.tensor([[0.0, 1.5]]))
    assert torch.allclose(m2.emb.tables['d4'].grad, torch.full_like(m2.emb.tables['d4'], 0.5))
    assert sorted(m2.emb.sparse_grads['d4'][0].rows.tolist()) == [3, 4]
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, 'ok'))


def test_data_parallel_exchange_world2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, 'ok'), (1, 'ok')]


def test_world_size_one_is_a_no_op():
    port = _free_port()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    from deeptables_amd.parallel import DataParallelStrategy
    from deeptables_amd.ops import SparseRowGrad
    st = DataParallelStrategy.from_env('gloo')
    try:
        g = SparseRowGrad(torch.tensor([1, 2]), torch.ones(2, 3))
        assert st.allgather_sparse(g) is g
        lin = torch.nn.Linear(2, 2)
        lin.weight.grad = torch.ones_like(lin.weight)
        assert st.allreduce_dense(list(lin.parameters())) == 0
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# ShardedEmbeddingStrategy: field-owned table rows, ids all-gather + two all-to-alls
# ---------------------------------------------------------------------------------------------
def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from deeptables_amd.parallel import ShardedEmbeddingStrategy
    st = ShardedEmbeddingStrategy.from_env('gloo')
    F, B, D, V = 5, 4, 3, 7                     # 5 fields over 3 ranks -> 2, 2, 1
    bounds = st.field_bounds(F)
    assert bounds == [(0, 2), (2, 4), (4, 5)] and st.active
    g = torch.Generator().manual_seed(0)        # same on every rank: a replicated packed table [F*V, D]
    table = torch.randn(F * V, D, generator=g)
    row_offset = torch.arange(F) * V
    idx_all = torch.randint(0, V, (world, B, F), generator=g, dtype=torch.int32)
    idx = idx_all[rank]
    s, e = bounds[rank]
    # forward: ids -> owner gather -> all-to-all
    ids_own = st.gather_ids(idx, F)
    assert ids_own.shape == (world, e - s, B)
    assert torch.equal(ids_own, idx_all[:, :, s:e].permute(0, 2, 1))
    rows_own = ids_own.long() + row_offset[s:e].view(1, -1, 1)
    emb_own = table[rows_own]                                          # [W, Fo, B, D] (stand-in for the HIP gather)
    emb_T = st.forward_exchange(emb_own, F, B)
    want = table[(idx.long() + row_offset[None, :])].permute(1, 0, 2)  # [F,B,D] of MY minibatch, all fields
    assert torch.equal(emb_T, want)
    # backward: per-row gradients travel to the owner; summed there they equal the dense-table gradient of all ranks
    grad_all = torch.randn(world, F, B, D, generator=g)
    grad_own = st.backward_exchange(grad_all[rank].contiguous(), F, B)
    assert grad_own.shape == (world, e - s, B, D)
    assert torch.equal(grad_own, grad_all[:, s:e])
    # the asynchronous form (fused.FusedDeepFM._run_sharded's opt-in overlap): (out, handle or None), the same rows
    grad_own2, work = st.backward_exchange(grad_all[rank].contiguous(), F, B, async_op=True)
    if work is not None:
        work.wait()
    assert torch.equal(grad_own2, grad_own)
    dense = torch.zeros(F * V, D)
    dense.index_add_(0, rows_own.reshape(-1), grad_own.reshape(-1, D))
    ref = torch.zeros(F * V, D)
    for r in range(world):
        rows_r = (idx_all[r].long() + row_offset[None, :]).permute(1, 0)          # [F,B]
        ref.index_add_(0, rows_r.reshape(-1), grad_all[r].reshape(-1, D))
    lo, hi = row_offset[s].item(), row_offset[e - 1].item() + V
    assert torch.allclose(dense[lo:hi], ref[lo:hi], atol=1e-6)         # complete for the fields I own ...
    assert float(dense[:lo].abs().sum() + dense[hi:].abs().sum()) == 0.0   # ... and nothing else

    # sync_tables: each owner's rows win
    class Emb:
        groups = [(D, list(range(F)))]
        tables = {f'd{D}': table.clone() + (rank + 1) * 100.0}
    setattr(Emb, f'row_offset_d{D}', row_offset)
    setattr(Emb, f'vocab_d{D}', torch.full((F,), V))
    st.sync_tables(Emb)
    for r, (a, b) in enumerate(bounds):
        assert torch.allclose(Emb.tables[f'd{D}'][a * V:b * V], table[a * V:b * V] + (r + 1) * 100.0)

    # exchange_gradients in a sharded step: dense (flat) only, sparse gradients stay with the owner
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(2, 1)
    m = M()
    flat = torch.full((3,), float(rank))
    m._dt_flat_grad, m._dt_sharded_step = flat, True
    m.fc.weight.grad, m.fc.bias.grad = flat[:2].view(1, 2), flat[2:]
    st.exchange_gradients(m)
    assert torch.allclose(flat, torch.full((3,), 1.0))                 # mean of 0,1,2
    # with an optimizer the reduce is started asynchronously and finished by the optimizer's pre-dense hook

    class Opt:
        pre_dense_hook = None
    opt = Opt()
    flat.fill_(float(rank) * 2)
    st.exchange_gradients(m, opt)
    assert callable(opt.pre_dense_hook)
    opt.pre_dense_hook()
    assert torch.allclose(flat, torch.full((3,), 2.0))                 # mean of 0,2,4
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, 'ok'))


def test_sharded_embedding_routing_world3_gloo():
    world = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(0, 'ok'), (1, 'ok'), (2, 'ok')]
