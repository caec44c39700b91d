# -*- coding:utf-8 -*-
"""Keras-semantics Adam kernels (csrc/optim.hip) against the oracle's restatement of keras.optimizers.Adam:
dense step, row-sparse step (field-local LDS dedupe and global-hash dedupe, duplicate lookups, skipped rows),
the device-resident step counter, and replay of a captured step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _adam(dev, params, emb_layers=()):
    from deeptables_amd.training import KerasAdam
    return KerasAdam(params, emb_layers)


def test_dense_adam_matches_keras_formula(dev):
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(1000, 7, generator=g)
    p = torch.nn.Parameter(p0.clone().to(dev))
    opt = _adam(dev, [p])
    rp, rm, rv = p0.double(), torch.zeros(1000, 7, dtype=torch.float64), torch.zeros(1000, 7, dtype=torch.float64)
    for t in range(1, 6):
        grad = torch.randn(1000, 7, generator=g) * (0.1 if t % 2 else 3.0)
        p.grad = grad.to(dev)
        opt.step()
        rp, rm, rv = R.keras_adam_step(rp, grad.double(), rm, rv, t)
        assert opt.t == t
        assert (p.detach().cpu().double() - rp).abs().max().item() < 2e-6, t


class _FakeEmb:
    """Minimal stand-in for MultiColumnEmbedding as the optimizer sees it."""

    def __init__(self, table, n_fields):
        self.tables = {f'd{table.shape[1]}': table}
        self.groups = [(table.shape[1], list(range(n_fields)))]
        self.sparse_grads = {}


@pytest.mark.parametrize('D,F,B,vocab,use_fields', [(16, 26, 512, 40, True), (16, 26, 512, 40, False),
                                                    (8, 3, 8192, 50000, True), (6, 5, 300, 20, True),
                                                    (32, 4, 9000, 3000, True), (4, 1, 64, 5, True)])
def test_rows_adam_merges_duplicates(dev, D, F, B, vocab, use_fields):
    """Packed table of F fields x vocab rows; ids drawn with many duplicates (and some skipped rows, -1)."""
    from deeptables_amd.ops import SparseRowGrad
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(D * 100 + F)
    V = F * vocab
    t0 = torch.randn(V, D, generator=g) * 0.05
    table = torch.nn.Parameter(t0.clone().to(dev))
    emb = _FakeEmb(table, F if use_fields else 0)
    if not use_fields:
        emb.groups = [(D, [])]          # fields = 0 -> global-hash dedupe
    opt = _adam(dev, [table], [emb])
    rp = t0.double()
    rm, rv = torch.zeros_like(rp), torch.zeros_like(rp)
    for t in range(1, 4):
        ids = torch.randint(0, vocab, (B, F), generator=g)
        rows = ids + torch.arange(F) * vocab
        rows[torch.rand(B, F, generator=g) < 0.02] = -1                     # out-of-range lookups are skipped
        vals = torch.randn(B * F, D, generator=g)
        emb.sparse_grads = {f'd{D}': [SparseRowGrad(rows.reshape(-1).to(dev), vals.clone().to(dev))]}
        opt.step()
        dense = torch.zeros(V, D, dtype=torch.float64)
        ok = rows.reshape(-1) >= 0
        dense.index_add_(0, rows.reshape(-1)[ok], vals.double()[ok])
        touched = torch.zeros(V, dtype=torch.bool)
        touched[rows.reshape(-1)[ok]] = True
        np_, nm, nv = R.keras_adam_step(rp, dense, rm, rv, t)
        rp = torch.where(touched[:, None], np_, rp)                         # lazy: untouched rows keep p, m, v
        rm = torch.where(touched[:, None], nm, rm)
        rv = torch.where(touched[:, None], nv, rv)
        assert (table.detach().cpu().double() - rp).abs().max().item() < 2e-6, t
    st = opt.state[id(table)]
    assert (st['m'].cpu().double() - rm).abs().max().item() < 1e-6
    if not use_fields:
        assert int(st['slots'].abs().sum().item()) == 0                     # global hash left empty for the next step


@pytest.mark.parametrize('D,n,vocab,regions,hot', [(16, 4096, 500, 4, 0), (16, 20000, 3000, 256, 700), (32, 3000, 200, 3, 50),
                                                    (4, 500, 40, 1, 0), (64, 2000, 100, 8, 300)])
def test_rows_adam_takes_row_segments(dev, D, n, vocab, regions, hot):
    """dt_adam_rows_step_seg: rows looked up once arrive as (rows, values) entries, rows looked up several times as
    SEGMENTS (region e: nseg[e] x (row, offset, count) + a list of the `values` rows to sum; their `rows` entries are -1),
    the way the fused steps hand them over (csrc/deepfm.hip DedupeWs).  A hot row with `hot` lookups included."""
    from deeptables_amd.ops import SparseRowGrad
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(D + n)
    t0 = torch.randn(vocab, D, generator=g) * 0.05
    table = torch.nn.Parameter(t0.clone().to(dev))
    emb = _FakeEmb(table, 0)
    opt = _adam(dev, [table], [emb])
    rp = t0.double()
    rm, rv = torch.zeros_like(rp), torch.zeros_like(rp)
    cap = n // 2 + 1
    for t in range(1, 4):
        ids = torch.randint(0, vocab, (n,), generator=g)
        if hot:
            ids[torch.randperm(n, generator=g)[:hot]] = 7
        ids[torch.rand(n, generator=g) < 0.02] = -1
        vals = torch.randn(n, D, generator=g)
        # host-side segment construction: rows with >= 2 lookups, dealt to the regions round-robin
        rows = ids.clone()
        uniq, counts = torch.unique(ids[ids >= 0], return_counts=True)
        multi = uniq[counts > 1].tolist()
        nseg = torch.zeros(regions, dtype=torch.int32)
        seg_row = torch.zeros(regions * cap, dtype=torch.int64)
        seg_off = torch.zeros(regions * cap, dtype=torch.int32)
        seg_cnt = torch.zeros(regions * cap, dtype=torch.int32)
        seg_list = torch.zeros(n, dtype=torch.int32)
        cursor = 0
        for k, r in enumerate(multi):
            members = torch.nonzero(ids == r).reshape(-1)
            e = k % regions
            sidx = e * cap + int(nseg[e])
            seg_row[sidx], seg_off[sidx], seg_cnt[sidx] = r, cursor, len(members)
            seg_list[cursor:cursor + len(members)] = members.int()
            cursor += len(members)
            nseg[e] += 1
            rows[members] = -1
        seg = (nseg.to(dev), seg_row.to(dev), seg_off.to(dev), seg_cnt.to(dev), seg_list.to(dev), regions, cap)
        sg = SparseRowGrad(rows.to(dev), vals.clone().to(dev), fields=-1, segments=seg)
        assert torch.equal(sg.expanded()[0].cpu(), ids)                  # the per-lookup view restores every row id
        emb.sparse_grads = {f'd{D}': [sg]}
        opt.step()
        dense = torch.zeros(vocab, D, dtype=torch.float64)
        ok = ids >= 0
        dense.index_add_(0, ids[ok], vals.double()[ok])
        touched = torch.zeros(vocab, dtype=torch.bool)
        touched[ids[ok]] = True
        np_, nm, nv = R.keras_adam_step(rp, dense, rm, rv, t)
        rp = torch.where(touched[:, None], np_, rp)
        rm = torch.where(touched[:, None], nm, rm)
        rv = torch.where(touched[:, None], nv, rv)
        assert (table.detach().cpu().double() - rp).abs().max().item() < 2e-6, t


def test_two_gradient_pieces_and_dp_sized_input(dev):
    """Several SparseRowGrad pieces for one table (what the data-parallel all-gather hands over): rows repeat across
    pieces; more than 8192 lookups per field falls back to the global hash."""
    from deeptables_amd.ops import SparseRowGrad
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(5)
    D, F, B, vocab = 16, 4, 6000, 700
    V = F * vocab
    t0 = torch.randn(V, D, generator=g) * 0.05
    table = torch.nn.Parameter(t0.clone().to(dev))
    emb = _FakeEmb(table, F)
    opt = _adam(dev, [table], [emb])
    pieces, dense = [], torch.zeros(V, D, dtype=torch.float64)
    for _ in range(2):
        rows = (torch.randint(0, vocab, (B, F), generator=g) + torch.arange(F) * vocab).reshape(-1)
        vals = torch.randn(B * F, D, generator=g)
        dense.index_add_(0, rows, vals.double())
        pieces.append(SparseRowGrad(rows.to(dev), vals.to(dev)))
    emb.sparse_grads = {'d16': pieces}
    opt.step()
    rp, _, _ = R.keras_adam_step(t0.double(), dense, torch.zeros_like(dense), torch.zeros_like(dense), 1)
    touched = dense.abs().sum(1) > 0
    assert (table.detach().cpu().double() - torch.where(touched[:, None], rp, t0.double())).abs().max().item() < 2e-6


def test_captured_step_replays_with_advancing_bias_correction(dev):
    """The step (advance + dense + rows) captured once in a hipGraph and replayed: t and lr_t advance on the
    device, so replay k equals eager step k."""
    from deeptables_amd.ops import SparseRowGrad
    from oracle import reference_layers as R
    g = torch.Generator().manual_seed(1)
    D, F, B, vocab = 16, 3, 256, 1000
    t0 = torch.randn(F * vocab, D, generator=g) * 0.05
    w0 = torch.randn(50, generator=g)
    table = torch.nn.Parameter(t0.clone().to(dev))
    w = torch.nn.Parameter(w0.clone().to(dev))
    emb = _FakeEmb(table, F)
    opt = _adam(dev, [table, w], [emb])
    rows = (torch.randint(0, vocab, (B, F), generator=g) + torch.arange(F) * vocab).reshape(-1)
    vals = torch.randn(B * F, D, generator=g)
    wg = torch.randn(50, generator=g)
    rows_d, vals_src, vals_d, wg_d = rows.to(dev), vals.to(dev), torch.empty_like(vals).to(dev), wg.to(dev)

    def body():
        vals_d.copy_(vals_src)                      # the rows step merges duplicates into its input
        w.grad = wg_d
        emb.sparse_grads = {'d16': [SparseRowGrad(rows_d, vals_d)]}
        opt.step()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()                                      # warm-up (allocates state): step 1
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    import gc
    gc.collect()
    gc.freeze()                                     # (old garbage stays uncollected inside a stream capture: compiled.py)
    try:
        with torch.cuda.graph(graph):
            body()                                  # capture only (not executed)
    finally:
        gc.unfreeze()
    for _ in range(3):
        graph.replay()                              # steps 2, 3, 4
    torch.cuda.synchronize()
    assert opt.t == 4
    dense = torch.zeros(F * vocab, D, dtype=torch.float64)
    dense.index_add_(0, rows, vals.double())
    touched = dense.abs().sum(1) > 0
    rp, rm, rv = t0.double(), torch.zeros_like(dense), torch.zeros_like(dense)
    rw, rwm, rwv = w0.double(), torch.zeros(50, dtype=torch.float64), torch.zeros(50, dtype=torch.float64)
    for t in range(1, 5):
        np_, nm, nv = R.keras_adam_step(rp, dense, rm, rv, t)
        rp, rm, rv = [torch.where(touched[:, None], a, b) for a, b in ((np_, rp), (nm, rm), (nv, rv))]
        rw, rwm, rwv = R.keras_adam_step(rw, wg.double(), rwm, rwv, t)
    assert (w.detach().cpu().double() - rw).abs().max().item() < 2e-6
    assert (table.detach().cpu().double() - rp).abs().max().item() < 2e-6


def test_pre_dense_hook_orders_table_updates_first(dev):
    """With a pre-dense hook (an all-reduce pending on the dense gradients) the table update is launched first, the
    hook runs, then the dense update advances the state — results identical to the fused single launch."""
    from deeptables_amd.ops import SparseRowGrad
    g = torch.Generator().manual_seed(9)
    D, F, B, vocab = 16, 2, 128, 500
    t0 = torch.randn(F * vocab, D, generator=g) * 0.05
    w0 = torch.randn(300, generator=g)
    rows = (torch.randint(0, vocab, (B, F), generator=g) + torch.arange(F) * vocab).reshape(-1)
    vals = torch.randn(B * F, D, generator=g)
    wg = torch.randn(300, generator=g)
    out = []
    for use_hook in (False, True):
        table = torch.nn.Parameter(t0.clone().to(dev))
        w = torch.nn.Parameter(w0.clone().to(dev))
        emb = _FakeEmb(table, F)
        opt = _adam(dev, [table, w], [emb])
        called = []
        for step in range(2):
            w.grad = wg.to(dev) * (0.5 if use_hook else 1.0)
            emb.sparse_grads = {'d16': [SparseRowGrad(rows.to(dev), vals.clone().to(dev))]}
            if use_hook:
                def hook(w=w):
                    called.append(1)
                    w.grad.mul_(2.0)              # what the pending all-reduce would deliver
                opt.pre_dense_hook = hook
            opt.step()
        assert opt.t == 2 and (len(called) == 2) == use_hook and opt.pre_dense_hook is None
        out.append((table.detach().cpu().clone(), w.detach().cpu().clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.allclose(out[0][1], out[1][1], atol=1e-7)


@pytest.mark.parametrize('vocab,B', [(30, 256), (2000, 777), (100000, 4096)])
def test_rows_compact_packs_unique_rows_with_summed_gradients(dev, monkeypatch, vocab, B):
    """dt_rows_compact (the data-parallel exchange's sparse bucket): a fused step's gradient — entries for the rows looked
    up once + segments — becomes one entry per distinct row holding scale x the sum over its lookups, packed at the
    front; a bucket that is too small drops entries and says how many."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import test_fused_gpu as T
    from deeptables_amd import ops
    from deeptables_amd.models import layers as L
    from oracle import headline
    monkeypatch.setattr(L, 'DENSE_GRAD_MAX_ELEMS', 0)
    dm, cats = T.build(26, 13, 16, vocab=vocab)
    idx, dense, y = T.batch(cats, 13, B, seed=3)
    dm.model.train()
    dm.forward_backward([idx.to(torch.int32).to(dev), dense.to(dev)], y.to(dev))
    g = dm.model.layers_by_name['emb_categorical_vars_all'].sparse_grads['d16'][0]
    assert g.segments is not None
    rows_e, vals_e = g.expanded()
    u_ref, v_ref = headline.merge_rows(rows_e.reshape(-1).cpu(), vals_e.reshape(-1, 16).double().cpu())
    n = rows_e.numel()
    out, ctr = ops.compact_rows(g, n, scale=0.5)
    torch.cuda.synchronize()
    produced, dropped = int(ctr[0]), int(ctr[1])
    assert dropped == 0 and produced == u_ref.numel()
    r = out.rows.cpu()
    assert bool((r[:produced] >= 0).all()) and bool((r[produced:] == -1).all())
    order = torch.argsort(r[:produced])
    assert torch.equal(r[:produced][order], u_ref)
    got = out.values.double().cpu()[:produced][order]
    assert (got - 0.5 * v_ref).abs().max().item() <= 1e-6 * max(v_ref.abs().max().item(), 1e-30) + 1e-12
    # a bucket of half the distinct rows: that many entries kept, the rest counted as dropped
    cap = max(1, produced // 2)
    out2, ctr2 = ops.compact_rows(g, cap)
    torch.cuda.synchronize()
    assert int(ctr2[0]) == produced and int(ctr2[1]) == produced - cap
    assert bool((out2.rows >= 0).all())
    # the dropped count is a RUNNING total over the calls on one counter (a host check after many steps must still see the
    # overflow of any of them); the slot counter restarts with every call
    ops.compact_rows(g, cap, counter=ctr2)
    ops.compact_rows(g, n, counter=ctr2)
    torch.cuda.synchronize()
    assert int(ctr2[0]) == produced and int(ctr2[1]) == 2 * (produced - cap)
    # the in-place form (the exchange's default): segments summed into their first member's entry, holes elsewhere
    merged = ops.merge_segments_(g)
    torch.cuda.synchronize()
    rm = merged.rows.reshape(-1).cpu()
    keep = rm >= 0
    assert int(keep.sum()) == u_ref.numel() and merged.fields == -1
    order = torch.argsort(rm[keep])
    assert torch.equal(rm[keep][order], u_ref)
    gm = merged.values.reshape(-1, 16).double().cpu()[keep][order]
    assert (gm - v_ref).abs().max().item() <= 1e-6 * max(v_ref.abs().max().item(), 1e-30) + 1e-12
