# -*- coding:utf-8 -*-
"""Data-parallel leg of the hot path (SURVEY §8 a14 / e): one process per GPU, replicated model,
minibatches sharded by rank, gradients exchanged once per step over RCCL (torch.distributed
backend "nccl" on ROCm) — the role `tf.distribute.MirroredStrategy` plays for the reference
(deeptables/models/deepmodel.py:88-103; tests/models/run_dt.py:35-44, global batch = n_gpus x bs).

Exchange per step (sized for xGMI's point-to-point links, not for an NVSwitch):
  * dense gradients  -> ONE flat fp32 bucket, one all-reduce (DCN: 69,796 params = 279 KB —
    latency bound, so never more than one collective), averaged over ranks;
  * embedding gradients stay SPARSE: every rank all-gathers (rows int64, values fp32 [n,D]) —
    14.5 MB per GPU at B=8192,F=26,D=16 — and appends the peers' slices to its own sparse
    gradient list; the row-sparse optimizer merges duplicates.  The 1.66 GB tables are never
    densified or all-reduced.
BatchNormalization statistics stay per replica (Keras default under MirroredStrategy).
The same code runs on CPU tensors with the "gloo" backend (tests/test_parallel.py).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .ops import SparseRowGrad


class DataParallelStrategy:
    """Pass as `ModelConfig(distribute_strategy=...)`."""

    # the fused steps may keep their in-step row dedupe under data parallel: the exchange packs a rank's sparse gradient
    # into unique (row, summed gradient) entries first (`allgather_sparse`, ops.compact_rows)
    compacts_segments = True

    def __init__(self, device=None, process_group=None, assume_uniform_batches=False, sparse_bucket_ratio=1.0):
        # wire size of a rank's sparse bucket as a fraction of its lookups (B x F entries): 1.0 can never overflow; with
        # skewed ids (Criteo-like: ~0.5 of the lookups are distinct rows) a smaller bucket halves the all-gather.  Entries
        # that do not fit are DROPPED and counted: `check_sparse_overflow()` raises on the host (call it outside the step).
        self.sparse_bucket_ratio = float(sparse_bucket_ratio)
        self._overflow_counters = []
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised; use DataParallelStrategy.from_env()')
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        self.device = device
        # every rank contributes the same number of lookups per step (fixed batch, drop_remainder=True):
        # skips the count exchange and its host synchronisation
        self.assume_uniform_batches = assume_uniform_batches

    @classmethod
    def from_env(cls, backend=None):
        """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment."""
        use_cuda = torch.cuda.is_available()
        backend = backend or ('nccl' if use_cuda else 'gloo')
        local_rank = int(os.environ.get('LOCAL_RANK', 0))
        device = None
        if use_cuda and backend == 'nccl':
            torch.cuda.set_device(local_rank)
            device = torch.device('cuda', local_rank)
        if not dist.is_initialized():
            kwargs = {'device_id': device} if device is not None else {}
            dist.init_process_group(backend=backend, **kwargs)
        return cls(device=device)

    # -- collectives that also work for device tensors under the gloo backend (staged through the host): lets the
    #    whole train step be exercised with several processes on ONE GPU (tests/test_parallel_gpu.py) ------------
    def _stage(self, t):
        if not hasattr(self, '_host_staged'):
            self._host_staged = dist.get_backend(self.group) == 'gloo'
        return self._host_staged and t.is_cuda

    def _all_reduce(self, t, async_op=False):
        if self._stage(t):
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _all_gather_into(self, out, t):
        if self._stage(t):
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, t.detach().cpu().contiguous(), group=self.group)
            out.copy_(h)
            return
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)

    def _persistent(self, tag, shape, dtype, device):
        """a buffer that keeps its address from step to step (so the optimizer step that consumes the gathered
        gradients can be captured in a hipGraph and replayed)"""
        if not hasattr(self, '_bufs'):
            self._bufs = {}
        key = (tag, tuple(shape), dtype, str(device))
        b = self._bufs.get(key)
        if b is None:
            b = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[key] = b
        return b

    # -- data ------------------------------------------------------------------------------------
    def shard(self, X, y):
        """AutoShardPolicy.DATA: rank r keeps rows r, r+W, r+2W, ... (deepmodel.py:92-95), truncated to the SAME
        length on every rank (floor(n / W) rows): every train step issues collectives, so ranks must agree on the
        number of steps per epoch — tf.distribute gives the reference consistent step counts the same way."""
        per = len(X) // self.world_size
        idx = np.arange(self.rank, len(X), self.world_size)[:per]
        Xs = X.iloc[idx] if hasattr(X, 'iloc') else X[idx]
        return Xs, (None if y is None else np.asarray(y)[idx])

    def shared_permutation(self, n):
        """A permutation of range(n) that is identical on every rank (drawn on rank 0, broadcast): the train /
        validation split must be ONE partition of the frame, taken before the rows are dealt to the ranks."""
        perm = torch.from_numpy(np.random.permutation(n).astype(np.int64))
        if self.world_size > 1:
            if self.device is not None and getattr(self.device, 'type', 'cpu') == 'cuda':
                t = perm.to(self.device)
                dist.broadcast(t, src=0, group=self.group)
                perm = t.cpu()
            else:
                dist.broadcast(perm, src=0, group=self.group)
        return perm.numpy()

    # -- parameters --------------------------------------------------------------------------------
    def broadcast_parameters(self, model):
        if self.world_size == 1:
            return
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)

    # -- gradients ---------------------------------------------------------------------------------
    def allreduce_dense(self, params):
        """One flat bucket, one all-reduce, mean over ranks."""
        grads = [p.grad for p in params if p.grad is not None]
        if not grads or self.world_size == 1:
            return 0
        flat = torch.cat([g.reshape(-1) for g in grads])
        self._all_reduce(flat)
        flat.div_(self.world_size)
        o = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[o:o + n].view_as(g))
            o += n
        return flat.numel()

    def allgather_sparse(self, grad, tag=0):
        """SparseRowGrad -> SparseRowGrad of all ranks' rows (values pre-divided by world size).  With
        assume_uniform_batches the results live in persistent buffers (one set per `tag`)."""
        W = self.world_size
        if getattr(grad, 'fields', None) == -2:
            raise RuntimeError('a sparse gradient whose rows were already applied inside the train step '
                               '(_forward_backward(apply_rows=True)) cannot be exchanged: apply_rows belongs to the '
                               'single-process train_step only')
        if W == 1 and not getattr(self, 'force_dp', False):
            return grad
        if self.sparse_bucket_ratio < 1.0 and getattr(grad, 'segments', None) is None and \
                getattr(grad, 'fields', None) != -1 and not getattr(self, '_warned_ratio', False):
            import warnings
            warnings.warn('sparse_bucket_ratio < 1 is ignored for a sparse gradient without the in-step dedupe '
                          '(B > 8192, DT_AMD_FUSED_DEDUPE=0 or a layer-by-layer graph): per-lookup entries are gathered')
            self._warned_ratio = True
        if getattr(grad, 'segments', None) is not None and self.sparse_bucket_ratio >= 1.0:
            # a fused step's deduplicated gradient, bucket as large as the lookups: sum every segment into its first
            # member's entry IN PLACE (no packing, no slot counter — 18 K returning atomics on one word cost 200 us) and
            # gather the (rows, values) pair as it is: one entry per distinct row of the rank + holes (row -1) the
            # receivers skip, so each applies W x (distinct rows) updates instead of W x B x F
            from . import ops
            grad = ops.merge_segments_(grad)
            if W == 1 and not getattr(self, 'force_collectives', False):
                return grad
        elif getattr(grad, 'segments', None) is not None or getattr(grad, 'fields', None) == -1:
            # sparse_bucket_ratio < 1: pack unique (row, summed gradient / W) entries into a SMALLER fixed-size bucket
            # (persistent, so the step stays capturable) and gather that — fewer bytes on the wire, at the price of the
            # packing pass (~0.2 ms at 213 K lookups: its slot counter is one word)
            from . import ops
            n, D = grad.rows.numel(), grad.values.shape[-1]
            dev = grad.rows.device
            if W > 1 and not self.assume_uniform_batches:
                # the bucket's size must be the same on every rank (all_gather_into_tensor): derive it from the largest
                # local lookup count (one small all-reduce + a host read per step — the price of ragged batches)
                cdev = 'cpu' if dist.get_backend(self.group) == 'gloo' else dev
                nmax = torch.tensor([n], dtype=torch.int64, device=cdev)
                dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=self.group)
                n = int(nmax.item())
            cap = max(1, int(np.ceil(self.sparse_bucket_ratio * n)))
            b_rows = self._persistent(('b_rows', tag), (cap,), torch.int64, dev)
            b_vals = self._persistent(('b_vals', tag), (cap, D), torch.float32, dev)
            fresh = ('b_ctr', tag) not in {k[0] for k in getattr(self, '_bufs', {})}
            ctr = self._persistent(('b_ctr', tag), (2,), torch.int32, dev)
            if fresh:
                ctr.zero_()           # [1] is a running total of dropped entries (dt_rows_compact resets only [0])
            if not any(c is ctr for c in self._overflow_counters):
                self._overflow_counters.append(ctr)
            ops.compact_rows(grad, cap, 1.0 / W, b_rows, b_vals, ctr)
            if W == 1 and not getattr(self, 'force_collectives', False):
                return SparseRowGrad(b_rows, b_vals, fields=0)
            all_rows = self._persistent(('rows', tag), (W * cap,), torch.int64, dev)
            all_vals = self._persistent(('vals', tag), (W * cap, D), torch.float32, dev)
            self._all_gather_into(all_rows, b_rows)
            self._all_gather_into(all_vals, b_vals)
            return SparseRowGrad(all_rows, all_vals, fields=0)     # rows of different ranks may coincide: global dedupe
        if self.assume_uniform_batches:
            n, D = grad.rows.numel(), grad.values.shape[1]
            dev = grad.rows.device
            all_rows = self._persistent(('rows', tag), (W * n,), torch.int64, dev)
            all_vals = self._persistent(('vals', tag), (W * n, D), grad.values.dtype, dev)
            mine = self._persistent(('mine', tag), (n, D), grad.values.dtype, dev)
            torch.div(grad.values.reshape(n, D), W, out=mine)
            self._all_gather_into(all_rows, grad.rows.reshape(-1))
            self._all_gather_into(all_vals, mine)
            return SparseRowGrad(all_rows, all_vals)
        cdev = 'cpu' if dist.get_backend(self.group) == 'gloo' else grad.rows.device
        n = torch.tensor([grad.rows.numel()], dtype=torch.int64, device=cdev)
        counts = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(counts, n, group=self.group)
        counts = [int(c.item()) for c in counts]
        nmax = max(counts)
        D = grad.values.shape[1]
        rows = torch.full((nmax,), -1, dtype=torch.int64, device=grad.rows.device)
        vals = torch.zeros((nmax, D), dtype=grad.values.dtype, device=grad.values.device)
        rows[:counts[self.rank]] = grad.rows
        vals[:counts[self.rank]] = grad.values / W
        all_rows = torch.empty((W * nmax,), dtype=torch.int64, device=rows.device)
        all_vals = torch.empty((W * nmax, D), dtype=vals.dtype, device=vals.device)
        self._all_gather_into(all_rows, rows)
        self._all_gather_into(all_vals, vals)
        return SparseRowGrad(all_rows, all_vals)     # padded entries carry row -1 and are skipped

    def check_sparse_overflow(self):
        """host check of the sparse buckets (sparse_bucket_ratio < 1): raises if any step dropped entries"""
        for c in self._overflow_counters:
            dropped = int(c[1].item())
            if dropped:
                raise RuntimeError(f'sparse gradient bucket overflow: {dropped} unique rows did not fit '
                                   f'sparse_bucket_ratio={self.sparse_bucket_ratio}; raise it (1.0 never overflows)')

    def exchange_gradients(self, model, optimizer=None):
        from .models.layers import MultiColumnEmbedding
        if self.world_size == 1 and not getattr(self, 'force_dp', False):
            return
        flat = getattr(model, '_dt_flat_grad', None)
        work = None
        if self.world_size == 1 and not getattr(self, 'force_collectives', False):
            flat = None                 # force_dp at world size 1: only the sparse bucketing below does anything
        # force_collectives (tests, with force_dp): world size 1 still issues the flat all-reduce and the sparse all-gathers
        if flat is not None:
            # the fused train step already keeps every dense gradient in ONE contiguous buffer: all-reduce it in
            # place (async, overlapped with the sparse all-gathers below), no bucket copy in or out
            work = self._all_reduce(flat, async_op=True)
            base = flat.untyped_storage().data_ptr()
            rest = [p for p in model.parameters() if p.requires_grad and p.grad is not None and
                    p.grad.untyped_storage().data_ptr() != base]     # e.g. a small table's dense gradient
            if rest:
                self.allreduce_dense(rest)
        elif self.world_size > 1:
            self.allreduce_dense([p for p in model.parameters() if p.requires_grad])
        tag = 0
        for layer in model.modules():
            if isinstance(layer, MultiColumnEmbedding):
                for key, grads in list(layer.sparse_grads.items()):
                    out = []
                    for g in grads:
                        out.append(self.allgather_sparse(g, tag))
                        tag += 1
                    layer.sparse_grads[key] = out
        if work is not None:
            work.wait()
        if flat is not None:
            flat.div_(self.world_size)


class _StreamWork:
    """what an asynchronous exchange hands back when it ran as plain launches on a side stream"""

    def __init__(self, stream, device):
        self.stream, self.device = stream, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return True


class ShardedEmbeddingStrategy(DataParallelStrategy):
    """Dense layers data-parallel, embedding ROWS owned by one rank each (model-parallel tables).

    The replicated-table exchange above makes every rank receive and apply all W x B x F row gradients, so its step
    time grows with W.  Here field f of the packed table is owned by one rank (contiguous field
    ranges); per step
        ids       all-gather            W x B x F x 4 B          (0.85 MB per rank at B=8192, F=26)
        rows      owner gathers [W, F_own, B, D] from ITS fields of the table
        forward   all-to-all            -> every rank gets [F, B, D] for its own minibatch (13.6 MB)
        ...       the fused train step runs on the local minibatch, reading those rows as its "table"
        backward  all-to-all            row gradients [F, B, D] (already divided by W) back to the owners (13.6 MB)
        update    the owner applies the row-sparse Adam to its fields for all W minibatches
    and the dense gradients take the one flat all-reduce as before.  Every exchanged block is FIELD-MAJOR so that the
    per-owner pieces are contiguous — no packing copies on either side of the all-to-all; over xGMI each peer's piece
    rides its own link.  Arithmetic is unchanged: a row's gradient is the sum over all minibatches / W, exactly what
    the replicated exchange computes; only where rows live differs (each rank's copy of the table is current for the
    fields it owns; `sync_tables` makes every copy whole again, e.g. before predict / save).
    Used by the fused DeepFM plan (deeptables_amd/fused.py); other graphs fall back to the replicated exchange."""

    sharded_embeddings = True

    def __init__(self, *args, force=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.force = force                       # run the sharded path even with one rank (single-GPU validation)
        self._a2a_ok = None

    @property
    def active(self):
        return self.world_size > 1 or self.force

    def field_bounds(self, F):
        """contiguous field ranges, the first F % W owners get one more"""
        W = self.world_size
        base, extra = divmod(F, W)
        out, s = [], 0
        for r in range(W):
            n = base + (1 if r < extra else 0)
            out.append((s, s + n))
            s += n
        return out

    # -- collectives (all_to_all_single where the backend has it; gloo: all_gather + slice) -----------
    def _all_to_all(self, out, inp, out_splits, in_splits, async_op=False):
        """async_op: returns a handle whose .wait() orders the CURRENT stream behind the exchange (RCCL: the collective's own
        stream; world size 1: the copy on a side stream), or None when the exchange was synchronous"""
        W = self.world_size
        if W == 1 and not getattr(self, 'force_collectives', False):
            if async_op and out.is_cuda:
                side = self._side_stream(out.device)
                side.wait_stream(torch.cuda.current_stream(out.device))
                with torch.cuda.stream(side):
                    out.copy_(inp)
                return _StreamWork(side, out.device)
            out.copy_(inp)
            return None
        if self._a2a_ok is None:
            self._a2a_ok = dist.get_backend(self.group) != 'gloo'
        if self._a2a_ok:
            work = dist.all_to_all_single(out, inp, output_split_sizes=list(out_splits),
                                          input_split_sizes=list(in_splits), group=self.group, async_op=bool(async_op))
            return work if async_op else None
        # gloo has no all_to_all: everybody gathers everybody's (padded) input and cuts its piece out
        n = torch.tensor([inp.shape[0]], dtype=torch.int64)
        sizes = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(sizes, n, group=self.group)
        nmax = max(int(t.item()) for t in sizes)
        pad = torch.zeros((nmax,) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
        pad[:inp.shape[0]] = inp
        bufs = [torch.empty_like(pad) for _ in range(W)]
        dist.all_gather(bufs, pad, group=self.group)
        splits_of = [torch.zeros(W, dtype=torch.int64) for _ in range(W)]
        dist.all_gather(splits_of, torch.tensor(list(in_splits), dtype=torch.int64), group=self.group)
        o = 0
        for src in range(W):
            sp = [int(v) for v in splits_of[src]]
            start = sum(sp[:self.rank])
            k = sp[self.rank]
            out[o:o + k] = bufs[src][start:start + k]
            o += k

    def gather_all_ids(self, idx):
        """idx [B,F] int32 of the local minibatch -> [W, B, F] ids of every rank's minibatch."""
        W = self.world_size
        B, F = idx.shape
        if W == 1 and not getattr(self, 'force_collectives', False):
            return idx.reshape(1, B, F)
        all_idx = torch.empty((W * B, F), dtype=idx.dtype, device=idx.device)
        dist.all_gather_into_tensor(all_idx, idx.contiguous(), group=self.group)
        return all_idx.view(W, B, F)

    def gather_ids(self, idx, F):
        """idx [B,F] int32 of the local minibatch -> ids [W, F_own, B] int32 of the fields this rank owns."""
        W = self.world_size
        B = idx.shape[0]
        s, e = self.field_bounds(F)[self.rank]
        if W == 1 and not getattr(self, 'force_collectives', False):
            all_idx = idx.reshape(1, B, F)
        else:
            all_idx = torch.empty((W * B, F), dtype=idx.dtype, device=idx.device)
            dist.all_gather_into_tensor(all_idx, idx.contiguous(), group=self.group)
            all_idx = all_idx.view(W, B, F)
        return all_idx[:, :, s:e].permute(0, 2, 1).contiguous()

    def forward_exchange(self, emb_own, F, B, out=None):
        """emb_own [W, F_own, B, D] (rows gathered by the owner) -> [F, B, D] for the local minibatch."""
        D = emb_own.shape[-1]
        bounds = self.field_bounds(F)
        Fo = bounds[self.rank][1] - bounds[self.rank][0]
        if out is None:
            out = torch.empty((F * B, D), dtype=emb_own.dtype, device=emb_own.device)
        self._all_to_all(out.view(F * B, D), emb_own.reshape(-1, D), [(e - s) * B for s, e in bounds],
                         [Fo * B] * self.world_size)
        return out.view(F, B, D)

    def _side_stream(self, device):
        s = getattr(self, '_side', None)
        if s is None:
            s = self._side = torch.cuda.Stream(device=device)
        return s

    def backward_exchange(self, grad_T, F, B, out=None, async_op=False):
        """grad_T [F, B, D] (gradient of the local loss w.r.t. the received rows) -> [W, F_own, B, D] at the owner.
        async_op=True -> (out, handle or None): the caller issues more launches, then handle.wait()."""
        D = grad_T.shape[-1]
        bounds = self.field_bounds(F)
        Fo = bounds[self.rank][1] - bounds[self.rank][0]
        W = self.world_size
        if out is None:
            out = torch.empty((W * Fo * B, D), dtype=grad_T.dtype, device=grad_T.device)
        work = self._all_to_all(out.view(W * Fo * B, D), grad_T.reshape(F * B, D), [Fo * B] * W,
                                [(e - s) * B for s, e in bounds], async_op=async_op)
        return (out.view(W, Fo, B, D), work) if async_op else out.view(W, Fo, B, D)

    def sync_tables(self, emb_layer):
        """Make every rank's copy of the packed tables whole again: each owner broadcasts its fields' rows."""
        if self.world_size == 1:
            return
        for D, cols in emb_layer.groups:
            key = f'd{D}'
            table = emb_layer.tables[key]
            offs = getattr(emb_layer, f'row_offset_{key}').tolist()
            voc = getattr(emb_layer, f'vocab_{key}').tolist()
            for r, (s, e) in enumerate(self.field_bounds(len(cols))):
                if e > s:
                    lo, hi = offs[s], offs[e - 1] + voc[e - 1]
                    dist.broadcast(table.data[lo:hi], src=r, group=self.group)

    def exchange_gradients(self, model, optimizer=None):
        """Dense gradients only — the embedding gradients already travelled to their owners inside the step.
        With an optimizer that has a `pre_dense_hook`, the flat all-reduce is started asynchronously and waited for
        only right before the dense update, so it overlaps the owners' table updates.
        `force_collectives` (tests): world size 1 still issues every collective — the RCCL calls of the N > 1 step
        (all_gather_into_tensor, all_to_all_single with split sizes, the asynchronous all_reduce) on one rank."""
        if self.world_size == 1 and not getattr(self, 'force_collectives', False):
            return
        if getattr(model, '_dt_sharded_step', False):
            flat = getattr(model, '_dt_flat_grad', None)
            if flat is not None:
                W = self.world_size
                base = flat.untyped_storage().data_ptr()
                rest = [p for p in model.parameters() if p.requires_grad and p.grad is not None and
                        p.grad.untyped_storage().data_ptr() != base]
                if optimizer is not None and hasattr(optimizer, 'pre_dense_hook') and not rest:
                    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

                    def finish():
                        work.wait()
                        flat.div_(W)
                    optimizer.pre_dense_hook = finish
                    return
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                flat.div_(W)
                if rest:
                    self.allreduce_dense(rest)
                return
        super().exchange_gradients(model)
