# -*- coding:utf-8 -*-
"""The compiled train loop: Keras' `model.compile(steps_per_execution=k)` (the knob of the reference's
`DeepModel.__compile_model`, deeptables/models/deepmodel.py:319-346, driven by `keras.Model.fit`,
deepmodel.py:114-129) on MI355X — `k` consecutive train steps captured ONCE into one hipGraph over static
input slots and replayed by `DeepModel.fit`.

Why: a fused DeepFM / DCN step is four or five launches of ~15-40 us; launched eagerly from Python the host cannot keep
up (~3.5 us per launch + the interpreter), and even a one-step graph pays its fixed replay cost (~10 us of idle
GPU between two replays) every step.  One graph of k steps keeps the launches of consecutive steps back to back.

Data path (inputs never leave HBM): the epoch's row order is a device vector (`set_order`); before a replay the
k*B indices of its steps are copied into the static `sel` vector (one stream-ordered copy), the graph's first
nodes gather those rows of the resident feed (`training.TableBatches`, resident mode) into the slots, step i of
the graph trains on slot rows [i*B, (i+1)*B).  Step counter, Adam's lr_t and the dropout seed live on the device
and advance inside the graph, so every replay is a new step.

Under a data-parallel strategy (`parallel.DataParallelStrategy`, world size > 1 or forced) RCCL collectives are
not captured: the step is [graph: gather + forward + backward] -> exchange_gradients (eager) -> [graph: the
optimizer launches]; with row-owned tables (`ShardedEmbeddingStrategy`) the step stays eager unless
`graph_segments=True`.  k is 1 in both cases.
"""
import ctypes
import os

import torch

from . import _lib, training
from ._lib import check, lib, ptr, stream_ptr


def _hip_graph_upload(graph):
    """hipGraphUpload of a captured torch graph onto the current stream (dt_graph_upload: through libdt_hip.so's HIP
    runtime, the one torch loaded — never a second copy dlopened by name): the first `replay()` then costs about what every
    later one costs (an un-uploaded 60-node graph pays ~60-200 us on its first launch).  Best effort: False when the handle
    is unavailable."""
    try:
        handle = graph.raw_cuda_graph_exec()
    except (AttributeError, RuntimeError):      # a torch build without the accessor / a graph that is not instantiated
        return False
    return lib().dt_graph_upload(ctypes.c_void_p(int(handle)), stream_ptr()) == 0


class _old_garbage_frozen:
    """Objects that exist when a stream capture starts are kept away from the cyclic collector until it ends (gc.freeze): a
    collection inside the capture otherwise runs the destructors of whatever old garbage it finds — an earlier loop's
    hipGraph, side streams, events of a model the caller dropped — and a destroy call of that kind inside a global-mode
    capture aborts the process.  The capture's own garbage is still collected as it appears.
    An application that froze objects itself (gc.freeze() at start-up) keeps its permanent generation untouched: unfreeze()
    would empty it and a re-freeze would pin everything alive at that moment — including cyclic garbage made since — for good
    (ADVICE r5).  For such a process the old garbage is collected once right before the capture and nothing is frozen."""

    def __enter__(self):
        import gc
        self.app_frozen = gc.get_freeze_count() > 0
        gc.collect()
        if not self.app_frozen:
            gc.freeze()
        return self

    def __exit__(self, *exc):
        import gc
        if not self.app_frozen:
            gc.unfreeze()
        return False


class CompiledTrainLoop:
    """`loop = CompiledTrainLoop(dm, feed, batch_size, steps_per_execution=10); loop.capture();
    loop.set_order(perm or None); loop.run(n_steps)`.

    feed: a resident `training.TableBatches`.  `run` executes exactly n_steps train steps from the current position of
    the order vector: whole executions (k steps per replay) and, when n_steps is not a multiple of k, the remaining
    steps eagerly through the same slots.  `on_execution(k_steps)` is called after every launch unit is enqueued (bench.py
    records a HIP event there); `collect=True` keeps per-step (loss, logits, y) on the device for the epoch's metrics."""

    def __init__(self, dm, feed, batch_size, steps_per_execution=10, with_optimizer=True, use_graph=True,
                 graph_segments=False, order_capacity=None):
        if not feed.resident:
            raise ValueError('CompiledTrainLoop needs a device-resident feed (training.TableBatches(resident=True))')
        self.dm, self.feed = dm, feed
        self._owner = None                # set by capture(): (model, optimizer, fused plan) the graph holds pointers into
        self.B = int(batch_size)
        self.with_optimizer = with_optimizer
        self.use_graph = use_graph
        self.graph_segments = graph_segments
        self.strategy = dm.config.distribute_strategy
        st = self.strategy
        self.dp = st is not None and (st.world_size > 1 or getattr(st, 'force_dp', False) or getattr(st, 'force', False))
        # Data parallel, round 6: the WHOLE step — forward + backward, the RCCL gradient exchange and the optimizer — of k
        # consecutive steps in one hipGraph (RCCL's collectives are stream-capturable; every rank captures the same sequence), so
        # that no Python runs between a step's launches.  Needs the "nccl" backend and fixed per-rank batches (persistent exchange
        # buffers); DT_AMD_DP_GRAPH=0 or a capture that fails falls back to the round-5 structure ([graph: fwd + bwd] -> eager
        # exchange -> [graph: optimizer], one step per unit).
        self.dp_graph = False
        self._dp_graph_wanted = bool(self.dp and use_graph and os.environ.get('DT_AMD_DP_GRAPH', '1') != '0' and
                                     self._nccl_backend(st))
        self.k = max(1, int(steps_per_execution)) if (not self.dp or self._dp_graph_wanted) else 1
        self.device = feed.device
        n = self.k * self.B
        # the epoch's row order lives in a persistent device vector and a device cursor walks it (dt_feed_gather advances it
        # behind every gather): a replay needs NO host-side feed work.  order_capacity: longest order set_order() will see
        self.order_cap = int(order_capacity or feed.n)
        self.order_buf = torch.arange(self.order_cap, dtype=torch.int64, device=self.device)
        self.order_len = min(self.order_cap, feed.n)
        self.cursor = torch.zeros(_lib.DT_FEED_CURSOR_WORDS, dtype=torch.int64, device=self.device)   # [position | dt_feed_gather's arrival tickets]
        self.sel = torch.zeros(n, dtype=torch.int64, device=self.device)     # (index_select fallback of odd feeds only)
        self.slots = [torch.empty((n,) + tuple(b.shape[1:]), dtype=b.dtype, device=self.device) for b in feed.blocks]
        self.slot_y = None if feed.y is None else \
            torch.empty((n,) + tuple(feed.y.shape[1:]), dtype=feed.y.dtype, device=self.device)
        self.pos = 0                # next row of the order to train on
        self.graph = None
        self.opt_graph = None       # data parallel: the optimizer launches, captured after the gather buffers exist
        self._dp_steps = 0
        self._opt_graph_sig = None       # addresses / shapes of the exchanged gradients the optimizer graph was captured on
        self._last_exchange_sig = None
        self._opt_graph_refused = False
        self._sig_objects = None         # (embedding layers, parameters) the signature walks
        self._sparse_refs = None
        self.logits = None          # [k, B, outputs] static: step i's logits
        self.losses = None          # [k] static (layer-by-layer path); fused plans: evaluated from the logits on demand
        # layer-by-layer path, single process: step i's (loss, logits) are kept BY REFERENCE — the tensors the step allocated
        # (inside a capture: in the graph's private pool, rewritten by every replay) — instead of two copy launches per step
        self._out_refs, self._graph_out_refs, self._outputs_by_ref = {}, None, False
        self.phase_events = None    # data parallel: [(e0, e1, e2, e3)] around fwd+bwd | exchange | optimizer (bench.py)
        self._first_events = None
        self._slots_per_step = False
        self.preelected = False
        self.chained = False
        self.uploaded = False
        srcs = list(feed.blocks) + ([] if self.slot_y is None else [feed.y])
        dsts = list(self.slots) + ([] if self.slot_y is None else [self.slot_y])
        nb = len(srcs)
        if nb > 8 or any((t.element_size() * (t.numel() // max(t.shape[0], 1))) % 4 for t in srcs):
            self._gather_args = False              # more blocks / odd row sizes than dt_feed_gather takes: index_select
        else:
            arr = ctypes.c_void_p * nb
            self._gather_args = (arr(*[t.data_ptr() for t in srcs]), arr(*[t.data_ptr() for t in dsts]),
                                 (ctypes.c_int * nb)(*[t.element_size() * (t.numel() // t.shape[0]) for t in srcs]), nb)

    @staticmethod
    def _nccl_backend(st):
        try:
            import torch.distributed as dist
            return dist.is_initialized() and dist.get_backend(st.group) == 'nccl'
        except Exception:
            return False

    def _dp_step(self, i):
        """one whole data-parallel train step on slot i: forward + backward, the gradient exchange, the optimizer step"""
        st, opt = self.strategy, self.dm.optimizer
        self._body(i)
        keep_uniform, st.assume_uniform_batches = getattr(st, 'assume_uniform_batches', False), True
        try:
            st.exchange_gradients(self.dm.model, opt if self.with_optimizer else None)
        finally:
            st.assume_uniform_batches = keep_uniform
        if self.with_optimizer:
            opt.step()          # (a pending asynchronous dense all-reduce is joined by the optimizer's pre_dense_hook)

    def _capture_dp_whole(self):
        """-> a graph of k whole data-parallel steps, or None (the caller falls back to the split structure)"""
        g = torch.cuda.CUDAGraph()
        try:
            with _old_garbage_frozen():
                with torch.cuda.graph(g):
                    self._gather()
                    for i in range(self.k):
                        self._dp_step(i)
            torch.cuda.synchronize()
            return g
        except Exception as e:          # a stack whose collectives cannot be captured: say so once, keep training
            import logging
            logging.getLogger('deeptables_amd').warning('data-parallel step: whole-step capture failed (%r); falling back to '
                                                        'graph | eager exchange | graph', e)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            return None

    def owned_by(self, dm):
        """True while `dm` still holds the model, optimizer and fused plan this loop (and its captured graph) was built on —
        `DeepModel.fit` rebuilds the loop otherwise (a rebuilt optimizer / plan has new buffers; the graph has the old ones')"""
        if self._owner is None:
            return True
        held = [r() if r is not None else None for r in self._owner]      # (weak references: a freed object's id can be reused)
        return held[0] is dm.model and held[1] is dm.optimizer and held[2] is getattr(dm, '_fused_plan', None)

    # -- the feed ---------------------------------------------------------------------------------------------
    def set_order(self, perm=None):
        """the row order the following `run` calls walk (a device int64 permutation of the feed's rows, or None = 0, 1, ..):
        copied into the persistent order vector, the device cursor back at its start (two stream-ordered launches)"""
        n = self.feed.n if perm is None else int(perm.numel())
        if n > self.order_cap:
            raise ValueError(f'set_order: {n} rows, the loop was built for at most {self.order_cap} (order_capacity)')
        if perm is None:
            torch.arange(n, out=self.order_buf[:n])
        else:
            self.order_buf[:n].copy_(perm)
        self.order_len = n
        self.cursor[:1].zero_()
        self.pos = 0

    def _select(self, steps):
        """host-side bookkeeping of the device cursor (the gather itself reads and advances it)"""
        n = steps * self.B
        if self.pos + n > self.rows_available():
            raise IndexError('CompiledTrainLoop.run past the end of the order: call set_order() for the next epoch')
        if self._gather_args is False:          # index_select fallback: the indices are copied out by the host-known position
            self.sel[:n].copy_(self.order_buf[self.pos:self.pos + n])
        self.pos += n

    def rows_available(self):
        return self.order_len

    def steps_left(self):
        return (self.rows_available() - self.pos) // self.B

    def _gather(self, steps=None):
        """slots[:, :n] <- the feed's rows sel[:n]: ONE launch for ids, continuous columns and labels (dt_feed_gather)"""
        n = (self.k if steps is None else steps) * self.B
        if self._gather_args is False:
            for blk, slot in zip(self.feed.blocks, self.slots):
                torch.index_select(blk, 0, self.sel[:n], out=slot[:n])
            if self.slot_y is not None:
                torch.index_select(self.feed.y, 0, self.sel[:n], out=self.slot_y[:n])
            return
        src, dst, rb, nb = self._gather_args
        check(lib().dt_feed_gather(ptr(self.order_buf), n, nb, ctypes.cast(src, ctypes.c_void_p),
                                   ctypes.cast(dst, ctypes.c_void_p), ctypes.cast(rb, ctypes.c_void_p), ptr(self.cursor),
                                   stream_ptr()), 'dt_feed_gather')

    def _step_inputs(self, i):
        s, e = i * self.B, (i + 1) * self.B
        ins = [sl[s:e] for sl in self.slots]
        yb = None if self.slot_y is None else self.slot_y[s:e]
        wb = None
        if self.feed.weighted:
            wb, yb = yb[:, -1].contiguous(), yb[:, :-1].contiguous()
            if self.feed.y_ndim == 1:
                yb = yb.reshape(-1)
        return ins, yb, wb

    # -- one train step on slot i (what the graph holds) -------------------------------------------------------------
    def _sharded(self):
        st = self.strategy
        return self.dp and getattr(st, 'sharded_embeddings', False) and st.active and self.dm.fused_plan() is not None

    @staticmethod
    def _id_slot(i, chained):
        """rows / segment buffers of captured step i.  Chained steps: step i consumes its slot and fills step i + 1's, and
        everything before step i has retired by then (stream order) — two buffers beside slot 0 alternate (0, 1, 2, 1, 2, ..)
        instead of one per captured step (27 MB each at batch 8192: 0.5 GB of never-touched memory in front of a 20-step
        execution's first replay).  The opt-in pre-election really runs every election ahead: one slot per step."""
        if not chained or i == 0:
            return i
        return 1 + ((i - 1) & 1)

    def _body(self, i, core_only=False, preelected=False, chained=False):
        dm = self.dm
        ins, yb, wb = self._step_inputs(i)
        if core_only:
            # row-owned tables, graph capture: only the launches between the collectives (the step's kernels)
            plan = dm.fused_plan()
            dm.optimizer.zero_grad(flat=False)
            loss, logit = plan.sharded_core(self.B, ins[1] if len(ins) > 1 else None, yb, self.strategy, wb)
            dm.model._dt_flat_grad = plan.accum
            return
        fused_opt = self.with_optimizer and not self.dp
        chain = {}
        if chained:
            # chained steps (fused.can_chain): step i prepares step i + 1 (its ids sit in the slots already: the execution's
            # gather ran first) and runs prepared by step i - 1; the first step of an execution prepares itself
            if i + 1 < self.k:
                chain['next_ids'] = (self.slots[0][(i + 1) * self.B:(i + 2) * self.B], self._id_slot(i + 1, True))
            chain['prepared'] = i >= 1
        loss, logit = dm._forward_backward(ins, yb, wb, apply_rows=fused_opt and self.strategy is None,
                                           logit_out=None if self.logits is None else self.logits[i],
                                           slot=self._id_slot(i, chained) if self._slots_per_step else 0,
                                           preelected=preelected, **chain)
        if fused_opt:
            dm.optimizer.step()             # single process: the optimizer step is part of the captured graph
        used_plan = getattr(dm, '_step_used_plan', False)
        if self.logits is None:             # first (eager) step: the static outputs of the k steps
            self.logits = torch.empty((self.k,) + tuple(logit.shape), dtype=logit.dtype, device=logit.device)
            self.losses = torch.zeros(self.k, dtype=torch.float32, device=logit.device)
        self._outputs_by_ref = not used_plan and not self.dp
        if self._outputs_by_ref:
            self._out_refs[i] = (loss.detach().reshape(()), logit.detach())
        else:
            if logit.data_ptr() != self.logits[i].data_ptr():
                self.logits[i].copy_(logit)     # row-owned tables / data parallel: the step wrote its own buffer
            if not used_plan:
                self.losses[i].copy_(loss.reshape(()))
        self._losses_from_logits = bool(used_plan)

    def capture(self, warm_steps=2, collect=None):
        """`warm_steps` eager steps on a side stream (lazy buffers, the optimizer's slot records) — they ARE train steps on
        the next rows of the current order: the caller counts them (-> their number) — then the capture."""
        dm = self.dm
        dm.model.train()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm_steps):
                self._eager_steps(1, collect)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self._dp_graph_wanted:
            g = self._capture_dp_whole()
            if self.strategy.world_size > 1:
                # every rank replays the same structure or none does: a rank whose capture failed takes the others with it to
                # the split structure (their k-step graphs would wait for collectives it never issues)
                import torch.distributed as dist
                ok = torch.tensor([1 if g is not None else 0], dtype=torch.int32, device=self.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.strategy.group)
                if int(ok.item()) == 0:
                    g = None
            if g is not None:
                self.graph, self.dp_graph = g, True
                import weakref
                self._owner = tuple(None if o is None else weakref.ref(o)
                                    for o in (dm.model, dm.optimizer, getattr(dm, '_fused_plan', None)))
                self.uploaded = _hip_graph_upload(g)
                torch.cuda.synchronize()
                return warm_steps
            self.k = 1              # the split structure runs one step per unit
        if not self.use_graph or (self._sharded() and not self.graph_segments):
            return warm_steps
        from .models.layers import MultiColumnEmbedding
        emb_layers = [l for l in dm.model.modules() if isinstance(l, MultiColumnEmbedding)]
        core_only = self._sharded()
        # the ids-only half of steps 2..k (packed rows of the lookups, the election of the rows looked up several times:
        # fused.preelect) depends on the gathered ids alone: it runs on a forked branch of the graph beside step 1, and the
        # steps behind it skip that work (every captured step has its own rows / segment buffers for this)
        plan = dm.fused_plan()
        pre = (self.k > 1 and not self.dp and not core_only and plan is not None and hasattr(plan, 'can_preelect') and
               plan.can_preelect(self.B) and self.feed.kinds[0] == 'cat' and
               self.slots[0].dtype in (torch.int32, torch.float32))
        # chained steps: the ids-only and weights-only work of step i + 1 inside step i's launches (no prep launch from the
        # second step of an execution on)
        chain = (not pre and self.k > 1 and not self.dp and not core_only and self.with_optimizer and plan is not None and
                 hasattr(plan, 'can_chain') and plan.can_chain(self.B) and self.feed.kinds[0] == 'cat' and
                 self.slots[0].dtype in (torch.int32, torch.float32) and not self.feed.weighted)
        self.chained = bool(chain)
        self._slots_per_step = bool(pre or chain)
        self.preelected = bool(pre)
        if pre or chain:                     # the steps' own rows / segment buffers exist before the capture starts
            for i in range(1, self.k):
                plan._slot_buffers(self.B, self._id_slot(i, chain))
        g = torch.cuda.CUDAGraph()
        # Objects that exist when the capture starts are kept away from the cyclic collector until it ends (gc.freeze): a
        # collection inside the capture otherwise runs the destructors of whatever old garbage it finds — an earlier
        # CompiledTrainLoop's hipGraph, side streams, events of a model the caller dropped — and a destroy call of that kind
        # inside a global-mode capture aborts the process (seen as an intermittent SIGABRT "Garbage-collecting" under pytest,
        # where earlier tests' loops are such garbage).  The capture's own garbage (autograd nodes of the layer-by-layer path)
        # is still collected as it appears: with the collector switched off altogether hipStreamEndCapture faulted on those
        # graphs (gpurun_out/r4c22).
        with _old_garbage_frozen():
            with torch.cuda.graph(g):
                if not core_only:
                    self._gather()
                if pre:
                    main = torch.cuda.current_stream()
                    side = torch.cuda.Stream()
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        for i in range(1, self.k):
                            plan.preelect(self.slots[0][i * self.B:(i + 1) * self.B], slot=i)
                for i in range(self.k):
                    if pre and i == 1:
                        torch.cuda.current_stream().wait_stream(side)
                    self._body(i, core_only=core_only, preelected=pre and i >= 1, chained=chain)
        self._slots_per_step = False        # eager steps (slot 0 buffers, their own election)
        self._graph_out_refs = dict(self._out_refs) if self._outputs_by_ref else None
        self.graph = g
        import weakref
        self._owner = tuple(None if o is None else weakref.ref(o)
                            for o in (dm.model, dm.optimizer, getattr(dm, '_fused_plan', None)))
        # python side effects (the sparse-gradient registration) are not replayed: keep the captured static
        # (rows, values) tensors and re-attach them after every replay (data parallel: the exchange reads them)
        self._sparse_refs = [(l, {key: list(v) for key, v in l.sparse_grads.items()}) for l in emb_layers]
        for layer in emb_layers:
            if not self.dp and self.with_optimizer:
                layer.sparse_grads.clear()
        torch.cuda.synchronize()
        self.uploaded = _hip_graph_upload(g)
        torch.cuda.synchronize()
        return warm_steps

    # -- running ---------------------------------------------------------------------------------------------------
    def _eager_steps(self, n, collect=None):
        """n steps through the slots without the graph (one at a time: slot 0)"""
        for _ in range(n):
            self._select(1)
            self._gather(1)
            self._run_unit(eager=True)
            if collect is not None:
                self._collect(collect, 1)

    def _run_unit(self, eager=False):
        """one launch unit: a replay of k steps (or one eager step on slot 0) + under data parallel the exchange and the
        optimizer"""
        ev = None
        if self.dp and self.phase_events is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        g = None if eager else self.graph
        if g is not None and self.dp_graph:
            g.replay()              # k whole steps: exchange and optimizer are inside
            return
        if self.dp_graph:           # an eager step of a loop whose graph holds whole steps (warm-up remainders)
            if ev:
                ev[1].record(); ev[2].record()
            self._dp_step(0)
            if ev:
                ev[3].record()
                self.phase_events.append(tuple(ev))
            return
        if g is not None and self._sharded():
            plan = self.dm.fused_plan()
            self._gather()
            plan.sharded_pre(self.slots[0][:self.B], self.strategy)
            g.replay()
            plan.sharded_post(self.B, self.strategy)
        elif g is not None:
            g.replay()
            if self._graph_out_refs is not None:
                self._out_refs = dict(self._graph_out_refs)     # (an eager remainder step may have replaced entry 0)
            if self.dp or not self.with_optimizer:
                for layer, refs in self._sparse_refs:
                    layer.sparse_grads = {key: list(v) for key, v in refs.items()}
        else:
            self._body(0)
        if self.dp:
            if ev:
                ev[1].record()
            opt = self.dm.optimizer
            # every step of this loop trains on exactly B rows per rank: the exchange may keep its results in persistent
            # buffers (no count exchange, no fresh allocations) — the precondition of a captured optimizer step
            st = self.strategy
            keep_uniform, st.assume_uniform_batches = getattr(st, 'assume_uniform_batches', False), True
            try:
                st.exchange_gradients(self.dm.model, opt if self.with_optimizer else None)
            finally:
                st.assume_uniform_batches = keep_uniform
            if ev:
                ev[2].record()
            # (only where an optimizer graph exists or may be captured: the eager row-owned step is host-bound, and walking the
            # model for the signature cost it ~10 us per step)
            graphed_opt = self.with_optimizer and self.use_graph and (not self._sharded() or self.graph_segments)
            sig = self._exchange_signature() if graphed_opt else None
            if self.opt_graph is not None and sig != self._opt_graph_sig:
                # the exchanged gradients moved (a strategy that does not keep them in place): the captured optimizer step
                # would read the capture step's addresses — drop it and stay eager
                self.opt_graph = None
                self._opt_graph_refused = True
            if self.with_optimizer:
                if self.opt_graph is not None:
                    hook, opt.pre_dense_hook = getattr(opt, 'pre_dense_hook', None), None
                    if hook is not None:
                        hook()          # the asynchronous dense all-reduce must have landed before the captured optimizer runs
                    self.opt_graph.replay()
                    for layer in getattr(opt, 'embedding_layers', []):
                        layer.sparse_grads.clear()
                elif self.use_graph and self._dp_steps >= 2 and (not self._sharded() or self.graph_segments) and \
                        not self._opt_graph_refused and sig == self._last_exchange_sig:
                    hook, opt.pre_dense_hook = getattr(opt, 'pre_dense_hook', None), None
                    if hook is not None:
                        hook()
                    torch.cuda.synchronize()
                    gopt = torch.cuda.CUDAGraph()
                    with _old_garbage_frozen():        # (old garbage is not collected inside a stream capture)
                        with torch.cuda.graph(gopt):
                            opt.step()
                    self.opt_graph = gopt
                    self._opt_graph_sig = sig
                    gopt.replay()
                else:
                    opt.step()          # first steps (slot buffers get allocated) and the eager row-owned step
            self._last_exchange_sig = sig
            self._dp_steps += 1
            if ev:
                ev[3].record()
                self.phase_events.append(tuple(ev))

    def _exchange_signature(self):
        """(address, shape) of every tensor the optimizer step reads after the exchange: the sparse (rows, values) pairs of
        the embedding layers and the dense gradients.  A captured optimizer step is only valid while these stay put."""
        if self._sig_objects is None:
            from .models.layers import MultiColumnEmbedding
            self._sig_objects = ([l for l in self.dm.model.modules() if isinstance(l, MultiColumnEmbedding)],
                                 list(self.dm.model.parameters()))
        layers, params = self._sig_objects
        sig = []
        for layer in layers:
            for key in sorted(layer.sparse_grads):
                for g in layer.sparse_grads[key]:
                    sig.append((key, g.rows.data_ptr(), tuple(g.rows.shape), g.values.data_ptr(), tuple(g.values.shape),
                                getattr(g, 'fields', None)))
        for p in params:
            if p.grad is not None:
                sig.append((p.grad.data_ptr(), tuple(p.grad.shape)))
        return tuple(sig)

    def run(self, n_steps, on_execution=None, collect=None):
        """exactly n_steps train steps.  collect: a dict with lists 'loss', 'logit', 'y' that receive per-execution device
        tensors ([k] losses, [k*B, ...] logits / labels) — the epoch's metrics read them once at its end."""
        done = 0
        while done < n_steps:
            if self.graph is not None and n_steps - done >= self.k:
                self._select(self.k)
                first = self._first_events is None
                if first:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self._run_unit()
                if first:
                    e1.record()
                    self._first_events = (e0, e1)
                k = self.k
            else:
                self._select(1)
                self._gather(1)
                self._run_unit(eager=True)
                k = 1
            done += k
            if collect is not None:
                self._collect(collect, k)
            if on_execution is not None:
                on_execution(k)
        return done

    def first_replay_us(self):
        """HIP-event time of the first replay of the captured graph (None: none yet)"""
        if self._first_events is None:
            return None
        torch.cuda.synchronize()
        return self._first_events[0].elapsed_time(self._first_events[1]) * 1e3

    def _collect(self, out, k):
        n = k * self.B
        by_ref = self._outputs_by_ref and all(j in self._out_refs for j in range(k))
        need_logits = getattr(self, '_losses_from_logits', False) or out.get('want_outputs', True)
        logits = None if not need_logits else \
            (torch.stack([self._out_refs[j][1] for j in range(k)]) if by_ref else self.logits[:k])
        yk = wk = None
        if self.slot_y is not None:
            yk = self.slot_y[:n]
            if self.feed.weighted:
                wk, yk = yk[:, -1], yk[:, :-1]
                if self.feed.y_ndim == 1:
                    yk = yk.reshape(-1)
        if getattr(self, '_losses_from_logits', False):
            # a fused plan evaluates the loss inside its kernels (one word of its gradient buffer, overwritten by the
            # next step): the epoch's loss curve is re-evaluated here from the k steps' logits, once per execution
            z = logits.reshape(k, self.B, -1)
            t = yk.reshape(k, self.B, -1).to(z.dtype)
            if self.dm.loss_name == 'binary_crossentropy':
                per = (torch.clamp(z, min=0) - z * t + torch.log1p(torch.exp(-z.abs()))).mean(-1)
            else:
                per = ((z - t) ** 2).mean(-1)
            if wk is not None:
                per = per * wk.reshape(k, self.B).to(z.dtype)
            out['loss'].append(per.mean(-1))
        else:
            out['loss'].append(torch.stack([self._out_refs[j][0] for j in range(k)]) if by_ref else self.losses[:k].clone())
        if out.get('want_outputs', True):
            out['logit'].append(logits.reshape((n,) + tuple(logits.shape[2:])).clone())
            out['y'].append(yk.clone())

    def phase_times(self):
        """mean microseconds of the three phases of a data-parallel step (HIP events on the launch stream; read after the
        timed region): what a scaling run needs to diagnose itself"""
        if not self.phase_events:
            return None
        torch.cuda.synchronize()
        n = len(self.phase_events)
        fb = sum(a.elapsed_time(b) for a, b, _, _ in self.phase_events) / n * 1e3
        ex = sum(b.elapsed_time(c) for _, b, c, _ in self.phase_events) / n * 1e3
        op = sum(c.elapsed_time(d) for _, _, c, d in self.phase_events) / n * 1e3
        return {'fwd_bwd_us': fb, 'exchange_us': ex, 'opt_us': op, 'steps': n}


def resolve_steps_per_execution(dm, requested, feed, batch_size, steps_per_epoch):
    """`fit(steps_per_execution=...)`: an int k > 1 asks for the compiled loop; 'auto' (the default, also
    `training.DEFAULT_STEPS_PER_EXECUTION`) compiles when the graph has a fused whole-step plan (known to be
    capturable: no host synchronisation inside the step), the feed is device resident and an epoch holds at least ten
    steps (5, 10 or 20 steps per execution, by the epoch's length); 1 / 0: eager steps."""
    if requested is None:
        requested = training.DEFAULT_STEPS_PER_EXECUTION
    if requested in (0, 1, False):
        return 1
    if not getattr(feed, 'resident', False) or feed.device.type != 'cuda':
        return 1
    st = dm.config.distribute_strategy
    if st is not None and st.world_size > 1:
        import torch.distributed as dist
        if dist.get_backend(st.group) != 'nccl':
            return 1                      # host-staged collectives (gloo tests): nothing to gain from captured halves
    if requested == 'auto':
        if dm.fused_plan() is None or (feed.weighted and not getattr(dm.fused_plan(), 'takes_sample_weight', False)):
            return 1
        # an epoch must hold at least two executions, and an execution at least five steps (below that the capture costs
        # more than it saves).  Twenty steps per execution where the epoch allows: the first step of an execution is the only
        # one that launches `k_prep`, and a replay boundary idles the GPU for a moment — 105.1 -> 104.4 us per step at
        # batch 8192 (tools/r5/call21.sh; 103.7 at forty, which an epoch of the reference's size rarely divides into)
        return 20 if steps_per_epoch >= 40 else 10 if steps_per_epoch >= 20 else 5 if steps_per_epoch >= 10 else 1
    return max(1, min(int(requested), int(steps_per_epoch)))
