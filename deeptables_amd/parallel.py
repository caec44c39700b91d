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

    def __init__(self, device=None, process_group=None, assume_uniform_batches=False):
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

    # -- data ------------------------------------------------------------------------------------
    def shard(self, X, y):
        """AutoShardPolicy.DATA: rank r keeps rows r, r+W, r+2W, ... (deepmodel.py:92-95)."""
        idx = np.arange(self.rank, len(X), self.world_size)
        Xs = X.iloc[idx] if hasattr(X, 'iloc') else X[idx]
        return Xs, (None if y is None else np.asarray(y)[idx])

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
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world_size)
        o = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[o:o + n].view_as(g))
            o += n
        return flat.numel()

    def allgather_sparse(self, grad):
        """SparseRowGrad -> SparseRowGrad of all ranks' rows (values pre-divided by world size)."""
        W = self.world_size
        if W == 1:
            return grad
        if self.assume_uniform_batches:
            all_rows = torch.empty((W * grad.rows.numel(),), dtype=torch.int64, device=grad.rows.device)
            all_vals = torch.empty((W * grad.values.shape[0], grad.values.shape[1]), dtype=grad.values.dtype,
                                   device=grad.values.device)
            dist.all_gather_into_tensor(all_rows, grad.rows.contiguous(), group=self.group)
            dist.all_gather_into_tensor(all_vals, (grad.values / W).contiguous(), group=self.group)
            return SparseRowGrad(all_rows, all_vals)
        n = torch.tensor([grad.rows.numel()], dtype=torch.int64, device=grad.rows.device)
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
        dist.all_gather_into_tensor(all_rows, rows, group=self.group)
        dist.all_gather_into_tensor(all_vals, vals, group=self.group)
        return SparseRowGrad(all_rows, all_vals)     # padded entries carry row -1 and are skipped

    def exchange_gradients(self, model):
        from .models.layers import MultiColumnEmbedding
        if self.world_size == 1:
            return
        flat = getattr(model, '_dt_flat_grad', None)
        work = None
        if flat is not None:
            # the fused train step already keeps every dense gradient in ONE contiguous buffer: all-reduce it in
            # place (async, overlapped with the sparse all-gathers below), no bucket copy in or out
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            base = flat.untyped_storage().data_ptr()
            rest = [p for p in model.parameters() if p.requires_grad and p.grad is not None and
                    p.grad.untyped_storage().data_ptr() != base]     # e.g. a small table's dense gradient
            if rest:
                self.allreduce_dense(rest)
        else:
            self.allreduce_dense([p for p in model.parameters() if p.requires_grad])
        for layer in model.modules():
            if isinstance(layer, MultiColumnEmbedding):
                for key, grads in list(layer.sparse_grads.items()):
                    layer.sparse_grads[key] = [self.allgather_sparse(g) for g in grads]
        if work is not None:
            work.wait()
            flat.div_(self.world_size)
