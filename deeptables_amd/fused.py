# -*- coding:utf-8 -*-
"""Fused train-step plans: when the graph DeepModel assembled is one the library has a whole-step
kernel sequence for, `DeepModel.train_step` runs that instead of the layer-by-layer autograd
path.  Same weights, same gradients (tests/test_fused_gpu.py checks both against the oracle).

DeepFM (nets ['linear','fm_nets','dnn_nets'], deepnets.py:15) -> `dt_deepfm_train_step`
(csrc/deepfm.hip): four launches (five for a step that prepares itself) instead of ~60, the optimizer inside them, the
sparse gradient deduplicated in the step; consecutive steps of a captured execution are CHAINED (`can_chain`); under
`parallel.ShardedEmbeddingStrategy` the same kernels run on rows gathered by their owning ranks.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .ops import SparseRowGrad
from .utils import consts

_ACC_NAMES = ['dW1', 'dW2', 'db1', 'db2', 'dw3', 'dwo', 'dbo', 'loss', 'dgamma', 'dbeta', 'dwlin']


def _step_loss(dm):
    """loss flag of the fused steps: 0 = binary task with BinaryCrossentropy, DT_STEP_LOSS_MSE = regression task with
    'mse' (deepmodel.py:126-141), None = a task / loss the steps do not take"""
    ln = getattr(dm, 'loss_name', None)
    if dm.task == consts.TASK_BINARY and ln == 'binary_crossentropy':
        return 0
    if dm.task == consts.TASK_REGRESSION and ln in ('mse', 'mean_squared_error'):
        return _lib.DT_STEP_LOSS_MSE
    return None


TILE_H1, TILE_H2 = 128, 64         # the tower widths the step's kernels are compiled for (csrc/deepfm.hip kH1 / kH2)


def _tower_widths(dnn_params):
    """(H1, H2) when the tower is one the fused steps take — two Dense(bias) -> relu cells without dropout / batch norm
    whose widths fit the compiled 128 x 64 tile (deepnets.py:401-427) — else None.  Narrower towers run on the same
    kernels: the plan stores W1 / b1 / W2 / b2 / w3 inside zero-padded [C,128] / [128] / [128,64] / [64] / [64] slabs and
    the model's parameters are the leading blocks of those slabs (views).  A padded unit has zero weights and bias, so
    its activation, every gradient that touches it and its Adam moments stay exactly zero."""
    hu = tuple(tuple(h) for h in dnn_params.get('hidden_units', ()))
    if len(hu) != 2 or any(len(h) != 3 for h in hu):
        return None
    (h1, d1, bn1), (h2, d2, bn2) = hu
    if d1 or d2 or bn1 or bn2 or not (1 <= int(h1) <= TILE_H1 and 1 <= int(h2) <= TILE_H2):
        return None
    if dnn_params.get('activation', 'relu') != 'relu' or dnn_params.get('custom_dnn_fn') is not None:
        return None
    return int(h1), int(h2)


def _tower_mfma_flag(dnn_params):
    """`dnn_params['mfma_dtype']` (or DT_AMD_TOWER_DTYPE): 'bf16x3' (default) = the tile kernel's four GEMMs on split-bf16
    matrix cores (csrc/tower_x3.h, DT_STEP_TOWER_X3: three bf16 parts = all 24 mantissa bits in the forward, two parts in the
    backward, fp32 accumulate — measured at the exact kernels' parity bars), 'f32' = exact-fp32 MFMA,
    'bf16' = plain bf16 operands (1e-2 of the oracle: an opt-in precision mode) -> the `phases` bit of the fused step"""
    mode = dnn_params.get('mfma_dtype') or os.environ.get('DT_AMD_TOWER_DTYPE', 'bf16x3')
    if mode in ('f32', 'fp32', 'float32'):
        return 0
    if mode == 'bf16x3':
        return _lib.DT_STEP_TOWER_X3
    if mode in ('bf16', 'bfloat16'):         # north_star's "1e-2 bf16": one bf16 product per operand pair (opt-in)
        return _lib.DT_STEP_TOWER_BF16
    raise ValueError(f"dnn_params['mfma_dtype'] = {mode!r}: 'f32', 'bf16x3' or 'bf16'")


def _mirror_in_flat(flat_params, accum, grad_views):
    """Moves every parameter of `grad_views` [(param, view of accum)] to the same place of `flat_params` (same offset,
    shape and strides as its gradient view) -> members for KerasAdam.register_flat_group."""
    members = []
    for p, gview in grad_views:
        off = (gview.data_ptr() - accum.data_ptr()) // 4
        assert tuple(gview.shape) == tuple(p.shape), (tuple(gview.shape), tuple(p.shape))
        pview = torch.as_strided(flat_params, tuple(gview.shape), tuple(gview.stride()), off)
        with torch.no_grad():
            pview.copy_(p.data)
        p.data = pview
        members.append((p, off, p.numel(), (tuple(gview.shape), tuple(gview.stride()))))
    return members


def _dedupe_in_step(plan, B, backward):
    """The in-step dedupe hands the rows looked up several times to the optimizer as SEGMENTS: only for an optimizer that
    takes them, a single process (the data-parallel exchange gathers plain (rows, values)), and row-sparse tables."""
    if not (backward and plan.dedupe and B * plan.F < (1 << 23)):
        return False
    opt = getattr(plan.dm, 'optimizer', None)
    if not getattr(opt, 'supports_row_segments', False):
        return False
    st = plan.dm.config.distribute_strategy
    if st is not None and (int(getattr(st, 'world_size', 1)) > 1 or getattr(st, 'force_dp', False)):
        # data parallel: only with an exchange that packs the segments into unique (row, summed gradient) entries before
        # the all-gather (parallel.DataParallelStrategy.allgather_sparse); DT_AMD_DP_DEDUPE=0: per-lookup entries on the wire
        if not getattr(st, 'compacts_segments', False) or getattr(st, 'sharded_embeddings', False) or \
                os.environ.get('DT_AMD_DP_DEDUPE', '1') == '0':
            return False
    return not plan.emb.uses_dense_grad(plan.D)


def _rows_in_step(plan, B, backward, apply_rows):
    """The pipelined DeepFM step can apply the optimizer's row-sparse update to the rows looked up once INSIDE the step
    (dt_deepfm_train_step_adam): only when the caller promises `optimizer.step()` follows at once (DeepModel.train_step),
    the in-step dedupe is on (single process, row-sparse table), the optimizer is the library's KerasAdam with
    nothing between the gradient and its update (no pending all-reduce hook), and DT_AMD_ROWS_IN_STEP != 0."""
    if not (apply_rows and backward and plan.dm.model.training and _dedupe_in_step(plan, B, backward)):
        return None
    if os.environ.get('DT_AMD_ROWS_IN_STEP', '1') == '0':
        return None
    opt = getattr(plan.dm, 'optimizer', None)
    if not getattr(opt, 'supports_rows_in_step', False) or getattr(opt, 'pre_dense_hook', None) is not None:
        return None
    return opt


def _segments(buf, B, F):
    seg = buf.get('segments')
    if seg is None:
        offs = (ctypes.c_int64 * 7)()
        check(lib().dt_deepfm_dedupe_segments(B, F, ctypes.cast(offs, ctypes.c_void_p)), 'dt_deepfm_dedupe_segments')
        raw = buf['dedupe'].view(torch.uint8)
        E, cap = int(offs[5]), int(offs[6])
        seg = (raw[offs[0]:offs[0] + 4 * E].view(torch.int32), raw[offs[1]:offs[1] + 8 * E * cap].view(torch.int64),
               raw[offs[2]:offs[2] + 4 * E * cap].view(torch.int32), raw[offs[3]:offs[3] + 4 * E * cap].view(torch.int32),
               raw[offs[4]:offs[4] + 4 * E * B].view(torch.int32), E, cap)
        buf['segments'] = seg
    return seg


def fused_enabled():
    return os.environ.get('DT_AMD_FUSED', '1') != '0'


def _row_weights(sample_weight, y):
    """Keras fit's sample_weight x class_weight as the [B] fp32 vector the step's loss block reads (None: unweighted)."""
    if sample_weight is None:
        return None
    w = sample_weight.reshape(-1).to(device=y.device, dtype=torch.float32).contiguous()
    if w.shape[0] != y.shape[0]:
        raise ValueError(f'sample_weight has {w.shape[0]} entries for {y.shape[0]} rows')
    return w


class FusedDeepFM:
    """Whole-step executor for the DeepFM graph.  Holds the static workspace / gradient buffers."""

    NETS = {'linear', 'fm_nets', 'dnn_nets'}
    takes_sample_weight = True                # the loss block scales each row's loss / dlogit (csrc/deepfm.hip DcnArgs.sw)

    @classmethod
    def eligible(cls, dm):
        c = dm.config
        try:
            if set(c.nets) != cls.NETS or len(c.nets) != 3 or dm.var_len_categorical_columns:
                return False
            if _step_loss(dm) is None:
                return False
            if c.stacking_op != consts.STACKING_OP_ADD or not (0 <= float(c.dense_dropout or 0) < 1):
                return False
            if not (0 <= float(c.embedding_dropout or 0) < 1):
                return False
            if _tower_widths(c.dnn_params) is None:
                return False
            L = dm.model.layers_by_name
            need = ['emb_categorical_vars_all', 'bn_concat_emb_dense', 'linear_logit', 'dnn_dense_1', 'dnn_dense_2',
                    'dense_logit_dnn_nets', 'task_output', 'fm_layer']
            if any(n not in L for n in need):
                return False
            emb = L['emb_categorical_vars_all']
            if len(emb.groups) != 1 or len(dm.continuous_columns or []) > 1:
                return False
            D = emb.groups[0][0]
            F = len(emb.input_dims)
            Nd = sum(col.input_dim for col in (dm.continuous_columns or []))
            return bool(lib().dt_deepfm_supported(max(int(getattr(dm, '_batch_hint', 0) or 0), 1), F, D, Nd, 128, 64))
        except Exception:
            return False

    def __init__(self, dm):
        self.dm = dm
        L = dm.model.layers_by_name
        self.emb = L['emb_categorical_vars_all']
        self.bn = L['bn_concat_emb_dense']
        self.lin = L['linear_logit']
        self.d1, self.d2 = L['dnn_dense_1'], L['dnn_dense_2']
        self.dl = L['dense_logit_dnn_nets']
        self.out = L['task_output']
        self.D = self.emb.groups[0][0]
        self.F = len(self.emb.input_dims)
        self.Nd = sum(col.input_dim for col in (dm.continuous_columns or []))
        self.C = self.F * self.D + self.Nd
        self.key = f'd{self.D}'
        self.device = self.emb.tables[self.key].device
        n_acc = lib().dt_deepfm_accum_floats(self.F, self.D, self.Nd)
        offs = (ctypes.c_int64 * 11)()
        check(lib().dt_deepfm_accum_offsets(self.F, self.D, self.Nd, ctypes.cast(offs, ctypes.c_void_p)),
              'dt_deepfm_accum_offsets')
        self.off = dict(zip(_ACC_NAMES, [int(v) for v in offs]))
        self.accum = torch.zeros(n_acc, dtype=torch.float32, device=self.device)
        self._bufs = {}
        a, o, C = self.accum, self.off, self.C
        H1, H2 = _tower_widths(dm.config.dnn_params)       # <= the compiled tile: the slabs are zero padded
        self.grad_views = [
            (self.d1.kernel, a[o['dW1']:o['dW1'] + C * TILE_H1].view(C, TILE_H1)[:, :H1]),
            (self.d2.kernel, a[o['dW2']:o['dW2'] + TILE_H1 * TILE_H2].view(TILE_H1, TILE_H2)[:H1, :H2]),
            (self.d1.bias, a[o['db1']:o['db1'] + H1]),
            (self.d2.bias, a[o['db2']:o['db2'] + H2]),
            (self.dl.kernel, a[o['dw3']:o['dw3'] + H2].view(H2, 1)),
            (self.out.kernel, a[o['dwo']:o['dwo'] + 1].view(1, 1)),
            (self.bn.gamma, a[o['dgamma']:o['dgamma'] + C]),
            (self.bn.beta, a[o['dbeta']:o['dbeta'] + C]),
            (self.lin.kernel, a[o['dwlin']:o['dwlin'] + self.F + self.Nd].view(self.F + self.Nd, 1)),
        ]
        if self.out.bias is not None:
            self.grad_views.append((self.out.bias, a[o['dbo']:o['dbo'] + 1]))
        self.loss_view = a[o['loss']:o['loss'] + 1]
        # embedding_dropout (config.py:84): element dropout inside the step's kernels; the seed word lives on the device
        # and is advanced by the step (a captured graph draws a new mask at every replay)
        self.emb_dropout = float(dm.config.embedding_dropout or 0)
        # dense_dropout (config.py:83): Dropout on the continuous input columns, masked where kernel A packs them
        self.dense_dropout = float(dm.config.dense_dropout or 0) if self.Nd else 0.0
        seed = int(torch.randint(1, 2 ** 31 - 1, (1,)).item())
        self.drop_seed = torch.tensor([seed], dtype=torch.int32, device=self.device)
        # duplicate lookups are resolved inside the step (kernels A and G) unless DT_AMD_FUSED_DEDUPE=0
        self.dedupe = os.environ.get('DT_AMD_FUSED_DEDUPE', '1') != '0'
        self.tower_flag = _tower_mfma_flag(dm.config.dnn_params)
        # diagnostic: s_memtime phase stamps into the workspace (tools/phase_times.py sets DT_AMD_STEP_STAMPS=1)
        self.diag_flag = _lib.DT_STEP_STAMPS if os.environ.get('DT_AMD_STEP_STAMPS') == '1' else 0
        # Parameters mirror the gradient layout in one flat buffer, so the optimizer updates every dense layer of
        # the model with ONE launch over (flat_params, accum) instead of one launch per tensor.
        self.flat_params = torch.zeros_like(self.accum)
        members = _mirror_in_flat(self.flat_params, a, self.grad_views)
        n_flat = o['dwlin'] + self.F + self.Nd
        opt = getattr(dm, 'optimizer', None)
        if opt is not None and hasattr(opt, 'register_flat_group'):
            opt.register_flat_group(self.flat_params, self.accum, members, n_flat)
        dm.model._dt_flat_grad = self.accum     # lets DataParallelStrategy all-reduce the gradients in place

    def _buffers(self, B):
        b = self._bufs.get(B)
        if b is None:
            nbytes = lib().dt_deepfm_workspace_bytes(B, self.F, self.D, self.Nd)
            if nbytes < 0:
                raise _lib.DtHipError('fused DeepFM step: unsupported shape')
            dev = self.device
            b = {'ws': torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev),   # zero-filled once: the batch-sum accumulators
                 'logit': torch.empty((B, 1), dtype=torch.float32, device=dev),
                 'rows': torch.empty((B, self.F), dtype=torch.int64, device=dev),
                 'grad_rows': torch.empty((B, self.F, self.D), dtype=torch.float32, device=dev),
                 # scratch of the in-step dedupe: field-major rows + the segment arrays (csrc/deepfm.hip DedupeWs)
                 'dedupe': torch.zeros((lib().dt_deepfm_dedupe_bytes(B, self.F) + 7) // 8, dtype=torch.int64,
                                       device=dev),
                 'dedupe_slots': lib().dt_deepfm_dedupe_slots(B, self.F)}
            self._bufs[B] = b
        return b

    # -- per-slot id state: the compiled loop (compiled.CompiledTrainLoop) keeps k steps in one hipGraph and runs the ids-only
    #    work of steps 2..k ahead of them; every captured step then needs its OWN rows / segment buffers --------------------
    def _slot_buffers(self, B, slot):
        buf = self._buffers(B)
        if not slot:
            return buf
        slots = buf.setdefault('slots', {})
        sb = slots.get(slot)
        if sb is None:
            dev = self.device
            sb = {'rows': torch.empty((B, self.F), dtype=torch.int64, device=dev),
                  'dedupe': torch.zeros((lib().dt_deepfm_dedupe_bytes(B, self.F) + 7) // 8, dtype=torch.int64, device=dev),
                  'dedupe_slots': buf['dedupe_slots']}
            slots[slot] = sb
        return sb

    def check_dedupe(self):
        """Host check of the in-step dedupe (reads one word per batch size back: call it outside the step — `DeepModel.fit`
        does at the end of every epoch, bench.py after the timed region): raises when an election block of a batch beyond
        8192 rows found its 8192-slot table full, i.e. some duplicate lookups of that step were treated as distinct rows.
        Needs > 8192 distinct rows of ONE field hashing to ONE of its B / 1024 partitions: not reachable by chance."""
        for B, buf in self._bufs.items():
            if not isinstance(B, int) or 'dedupe' not in buf:
                continue
            off = int(lib().dt_deepfm_dedupe_overflow_offset(B, self.F))
            bufs = [buf] + list(buf.get('slots', {}).values())
            for b in bufs:
                n = int(b['dedupe'].view(torch.int32)[off // 4].item())
                if n:
                    raise _lib.DtHipError(f'in-step dedupe: {n} lookups found their election table full (batch {B})')

    def can_preelect(self, B):
        """the step's ids-only half can run ahead of it (dt_deepfm_preelect): in-step dedupe on, single process"""
        st = self.dm.config.distribute_strategy
        # OFF by default (DT_AMD_PREELECT=1 turns it on).  Measured (tools/r4/call7.sh): a forked branch of a captured hipGraph
        # does not run beside the main branch on this stack — the executor ran the nine elections first, on one queue, then
        # alternated queues per step with ~9.5 us joins: 131.9 us per step against 109.5 us with every step electing for
        # itself.  The entry point stays for a caller that owns a second stream outside a graph.
        return bool(st is None and _dedupe_in_step(self, B, True) and os.environ.get('DT_AMD_PREELECT', '0') == '1')

    def preelect(self, idx, slot):
        """rows + segments of batch `idx` into slot `slot`'s buffers, on the current stream; the step that follows is run
        with `run(..., slot=slot, preelected=True)`"""
        B = idx.shape[0]
        sb = self._slot_buffers(B, slot)
        idx = idx.contiguous()
        kind = _lib.DT_IDX_F32 if idx.dtype == torch.float32 else _lib.DT_IDX_I32
        if idx.dtype not in (torch.float32, torch.int32):
            raise ValueError('preelect: ids must be float32 or int32')
        check(lib().dt_deepfm_preelect(ptr(idx), kind, ptr(getattr(self.emb, f'row_offset_{self.key}')),
                                       ptr(getattr(self.emb, f'vocab_{self.key}')), B, self.F, ptr(sb['rows']),
                                       ptr(sb['dedupe']), sb['dedupe_slots'], stream_ptr()), 'dt_deepfm_preelect')

    # -- model-parallel tables (parallel.ShardedEmbeddingStrategy) --------------------------------------
    def _sharded_buffers(self, B, st):
        key = ('sharded', B)
        sb = self._bufs.get(key)
        if sb is None:
            dev, F, D, W = self.device, self.F, self.D, st.world_size
            s, e = st.field_bounds(F)[st.rank]
            Fo = e - s
            f_idx = torch.arange(F, device=dev, dtype=torch.int32)
            b_idx = torch.arange(B, device=dev, dtype=torch.int32)
            sb = {'Fo': Fo, 's': s, 'e': e,
                  # the received rows [F,B,D] are read by the fused step as a "table" of F*B rows: id(b,f) = f*B + b
                  'iota': (f_idx[None, :] * B + b_idx[:, None]).contiguous(),
                  'zero_off': torch.zeros(F, dtype=torch.int64, device=dev),
                  'fb_vocab': torch.full((F,), F * B, dtype=torch.int32, device=dev),
                  'emb_own': torch.empty((W * Fo * B, 1, D), dtype=torch.float32, device=dev),
                  'rows_own': torch.empty((W * Fo * B, 1), dtype=torch.int64, device=dev),
                  'emb_T': torch.empty((F * B, D), dtype=torch.float32, device=dev),
                  'grad_own': torch.empty((W * Fo * B, D), dtype=torch.float32, device=dev),
                  'rows_dummy': torch.empty((B, F), dtype=torch.int64, device=dev)}
            self._bufs[key] = sb
        return sb

    def _run_sharded(self, idx, dense, y, st, sample_weight=None):
        """One train step with the table rows owned per field by the ranks of `st` (see ShardedEmbeddingStrategy):
        ids all-gather -> owner gather -> all-to-all -> the same fused kernels on the local minibatch (reading the
        received rows) -> all-to-all of the row gradients -> the owner's sparse gradient.  Three pieces so that a caller
        can capture the kernels between the collectives into hipGraphs (bench.py): `sharded_pre` (collectives + the
        owner's gather, eager), `sharded_core` (the step's kernels: capturable), `sharded_post` (collective, eager)."""
        self.sharded_pre(idx, st)
        if os.environ.get('DT_AMD_SHARDED_OVERLAP', '0') != '1':
            out = self.sharded_core(idx.shape[0], dense, y, st, sample_weight)
            self.sharded_post(idx.shape[0], st)
            return out
        # opt-in (DT_AMD_SHARDED_OVERLAP=1): the row gradients are final one launch before the step is, so their all-to-all
        # can start there and the step's last launch (the dense gradients' last level, ~7 us) run beside it.  Measured at
        # world size 1 through RCCL (gpurun_out/r3c23): 214 us against 171 us for the plain order — the two stream
        # hand-overs cost more than the launch they hide — so the plain order is the default.
        out = self.sharded_core(idx.shape[0], dense, y, st, sample_weight, part=_lib.DT_STEP_SKIP_FINISH)
        work = self.sharded_post(idx.shape[0], st, async_op=True)
        self.sharded_core(idx.shape[0], dense, y, st, sample_weight, part=_lib.DT_STEP_FINISH_ONLY)
        if work is not None:
            work.wait()               # stream-ordered: the launches that follow (the owner's row update) see the received rows
        return out

    def sharded_pre(self, idx, st):
        B, F, D, W = idx.shape[0], self.F, self.D, st.world_size
        sb = self._sharded_buffers(B, st)
        if idx.dtype != torch.int32:
            idx = idx.to(torch.int32)             # float ids: truncation, as the gather's own cast
        table = self.emb.tables[self.key]
        row_offset = getattr(self.emb, f'row_offset_{self.key}')
        vocab = getattr(self.emb, f'vocab_{self.key}')
        s, e, Fo = sb['s'], sb['e'], sb['Fo']
        idx_all = st.gather_all_ids(idx.contiguous())                                 # [W, B, F] int32
        check(lib().dt_embedding_gather_owned(ptr(idx_all), _lib.DT_IDX_I32, ptr(table), ptr(row_offset), ptr(vocab),
                                              W, B, F, s, e, D, ptr(sb['emb_own']), ptr(sb['rows_own']),
                                              ptr(self.emb.oob_count) if self.emb.check_oob else None, stream_ptr()),
              'dt_embedding_gather_owned')
        st.forward_exchange(sb['emb_own'].view(W, Fo, B, D), F, B, out=sb['emb_T'])

    def sharded_core(self, B, dense, y, st, sample_weight=None, part=0):
        """the fused step's launches on the received rows (no collective inside: a hipGraph can hold them)"""
        F, D, W = self.F, self.D, st.world_size
        buf = self._buffers(B)
        sb = self._sharded_buffers(B, st)
        dense = None if dense is None else dense.contiguous()
        y = y.reshape(-1).contiguous()
        sw = _row_weights(sample_weight, y)
        training = self.dm.model.training
        check(lib().dt_deepfm_train_step(
            ptr(sb['iota']), _lib.DT_IDX_I32, ptr(sb['emb_T']), ptr(sb['zero_off']), ptr(sb['fb_vocab']), ptr(dense), ptr(y),
            B, F, D, self.Nd, ptr(self.lin.kernel), ptr(self.bn.gamma), ptr(self.bn.beta),
            ptr(self.bn.moving_mean) if training else None, ptr(self.bn.moving_variance) if training else None,
            float(self.bn.epsilon), float(self.bn.momentum), ptr(self.d1.kernel), ptr(self.d1.bias),
            ptr(self.d2.kernel), ptr(self.d2.bias), ptr(self.dl.kernel), ptr(self.out.kernel), ptr(self.out.bias),
            ptr(buf['logit']), ptr(sb['rows_dummy']), ptr(buf['grad_rows']), ptr(self.accum), ptr(buf['ws']),
            None, None, 0, 1.0 / W, 1, 2 | part | _step_loss(self.dm) | (0 if part == _lib.DT_STEP_FINISH_ONLY else self.tower_flag | self.diag_flag),
            self.emb_dropout if training else 0.0, ptr(self.drop_seed),
            self.dense_dropout if training else 0.0, ptr(sw), stream_ptr()),
            'dt_deepfm_train_step')
        for p, g in self.grad_views:
            p.grad = g
        self.dm.model._dt_sharded_step = True
        return self.loss_view, buf['logit']

    def sharded_post(self, B, st, async_op=False):
        # the loss is a mean over the LOCAL minibatch, the global objective the mean over W of them: the step already
        # wrote the row gradients field-major [F,B,D] and divided by W
        F, D = self.F, self.D
        buf = self._buffers(B)
        sb = self._sharded_buffers(B, st)
        if async_op:
            grad_own, work = st.backward_exchange(buf['grad_rows'].view(F, B, D), F, B, out=sb['grad_own'], async_op=True)
        else:
            grad_own, work = st.backward_exchange(buf['grad_rows'].view(F, B, D), F, B, out=sb['grad_own']), None
        self.emb.sparse_grads[self.key] = [SparseRowGrad(sb['rows_own'].view(-1), grad_own.view(-1, D), fields=0)]
        return work

    def _whole_in_step(self, opt):
        """the optimizer's flat dense group is this plan's: the step's last launch runs the whole optimizer step"""
        table = self.emb.tables[self.key]
        flat = getattr(opt, '_flat', None)
        return bool(flat is not None and flat[0] is self.flat_params and flat[1] is self.accum and
                    os.environ.get('DT_AMD_STEP_IN_STEP', '1') != '0' and
                    all(id(p) in flat[5] for p in opt.params if p is not table))

    def can_chain(self, B):
        """consecutive steps of one captured execution can be CHAINED (csrc/deepfm.hip StepNext): step i packs step i + 1's
        rows, runs its election on the weight-gradient launch's idle matrix waves and writes the tile kernel's weight layouts
        from the weights it has just updated — step i + 1 then has no prep launch (four launches).  Single process, in-step
        dedupe and optimizer, split-bf16 tower, B <= 8192; DT_AMD_CHAIN=0 turns it off."""
        if os.environ.get('DT_AMD_CHAIN', '1') == '0' or self.dm.config.distribute_strategy is not None:
            return False
        opt = _rows_in_step(self, B, True, True) if _dedupe_in_step(self, B, True) else None
        if opt is None or not self._whole_in_step(opt):
            return False
        phases = 2 | _step_loss(self.dm) | self.tower_flag
        return bool(lib().dt_deepfm_step_chains(B, self.F, self.D, self.Nd, phases))

    def run(self, idx, dense, y, backward=True, apply_rows=False, sample_weight=None, logit_out=None, slot=0,
            preelected=False, next_ids=None, prepared=False):
        """-> (loss [1] view, logit [B,1]).  slot / preelected: the compiled loop's per-step id buffers (`preelect`).
        next_ids / prepared (chained steps, `can_chain`): next_ids = (ids of the following step, its slot) — this step prepares
        it; prepared: this step was prepared by the one before it (slot's buffers hold its rows / segments).  logit_out: a caller-owned [B,1] fp32 buffer the step writes its logits to (the
        compiled loop keeps one per captured step) instead of the plan's own.  With backward=True fills `.grad` of every dense parameter
        (views of one static buffer) and registers the embedding table's sparse gradient.  apply_rows=True: the caller
        runs `optimizer.step()` right after this call, so the step may update the table rows looked up once itself
        (`_rows_in_step`); the registered sparse gradient then carries `fields = -2` (segments only)."""
        st = self.dm.config.distribute_strategy
        if backward and getattr(st, 'sharded_embeddings', False) and st.active and \
                not self.emb.uses_dense_grad(self.D):
            return self._run_sharded(idx, dense, y, st, sample_weight)
        self.dm.model._dt_sharded_step = False
        B = idx.shape[0]
        buf = self._buffers(B)
        idx = idx.contiguous()
        kind = _lib.DT_IDX_F32 if idx.dtype == torch.float32 else _lib.DT_IDX_I32
        if idx.dtype not in (torch.float32, torch.int32):
            idx = idx.to(torch.int32)
        dense = None if dense is None else dense.contiguous()
        y = y.reshape(-1).contiguous()
        sw = _row_weights(sample_weight, y)
        table = self.emb.tables[self.key]
        training = self.dm.model.training
        logit = buf['logit']
        if logit_out is not None:
            if logit_out.shape != logit.shape or logit_out.dtype != logit.dtype or not logit_out.is_contiguous():
                raise ValueError(f'logit_out must be a contiguous float32 {tuple(logit.shape)} tensor')
            logit = logit_out
        dedupe = _dedupe_in_step(self, B, backward)
        opt = _rows_in_step(self, B, backward, apply_rows)
        ids = self._slot_buffers(B, slot) if (slot and dedupe) else buf        # this step's rows / segment buffers
        pre = _lib.DT_STEP_PREELECTED if (preelected and dedupe and backward) else 0
        if preelected and not pre:
            raise _lib.DtHipError('a pre-elected step needs the in-step dedupe (backward, single process)')
        nxt = (None, None, None)
        if prepared or next_ids is not None:
            if opt is None or not dedupe:
                raise _lib.DtHipError('chained steps need the in-step dedupe and optimizer (can_chain)')
            if prepared:
                pre |= _lib.DT_STEP_PREPARED
            if next_ids is not None:
                nidx, nslot = next_ids
                if nidx.dtype != idx.dtype or nidx.shape != idx.shape or not nidx.is_contiguous() or not nslot or nslot == slot:
                    raise ValueError('next_ids: (contiguous ids like this step\'s, a slot of their own)')
                nb = self._slot_buffers(B, nslot)
                nxt = (ptr(nidx), ptr(nb['rows']), ptr(nb['dedupe']))
        head = (ptr(idx), kind, ptr(table), ptr(getattr(self.emb, f'row_offset_{self.key}')),
                ptr(getattr(self.emb, f'vocab_{self.key}')), ptr(dense), ptr(y), B, self.F, self.D, self.Nd,
                ptr(self.lin.kernel), ptr(self.bn.gamma), ptr(self.bn.beta),
                ptr(self.bn.moving_mean) if training else None, ptr(self.bn.moving_variance) if training else None,
                float(self.bn.epsilon), float(self.bn.momentum), ptr(self.d1.kernel), ptr(self.d1.bias),
                ptr(self.d2.kernel), ptr(self.d2.bias), ptr(self.dl.kernel), ptr(self.out.kernel), ptr(self.out.bias),
                ptr(logit), ptr(ids['rows']), ptr(buf['grad_rows']), ptr(self.accum), ptr(buf['ws']),
                ptr(self.emb.oob_count) if self.emb.check_oob else None,
                ptr(ids['dedupe']) if dedupe else None, buf['dedupe_slots'])
        whole = False
        if opt is not None:
            # the rows looked up once are updated where their gradient is formed (csrc/deepfm.hip k_wgrad_rows); when the
            # optimizer's flat dense group is this plan's, the step's last launch runs the rest of the optimizer step too
            # (dense elements, segments, the state's advance: k_finish_step) and `optimizer.step()` has nothing left to do
            slots = opt._st(table, rows=True)
            flat = getattr(opt, '_flat', None)
            whole = (flat is not None and flat[0] is self.flat_params and flat[1] is self.accum and
                     os.environ.get('DT_AMD_STEP_IN_STEP', '1') != '0' and
                     all(id(p) in flat[5] for p in opt.params if p is not table))
            dn = (ptr(flat[0]), ptr(flat[2]), ptr(flat[3]), int(flat[4]), float(opt.lr)) if whole else (None, None, None, 0, 0.0)
            check(lib().dt_deepfm_train_step_adam(
                *head, 2 | _step_loss(self.dm) | self.tower_flag | self.diag_flag | pre, self.emb_dropout if training else 0.0, ptr(self.drop_seed),
                self.dense_dropout if training else 0.0, ptr(sw), ptr(slots['m']), ptr(slots['v']), int(slots['m'].stride(0)), ptr(opt._state_tensor(table.device)), 0.0,
                opt.b1, opt.b2, opt.eps, *dn, *nxt, stream_ptr()), 'dt_deepfm_train_step_adam')
            if whole:
                opt.applied_in_step()
        else:
            check(lib().dt_deepfm_train_step(
                *head, 1.0, 0, (2 if backward else 1) | _step_loss(self.dm) | ((self.tower_flag | self.diag_flag) if backward else 0) | pre,
                self.emb_dropout if training else 0.0,
                ptr(self.drop_seed), self.dense_dropout if training else 0.0, ptr(sw), stream_ptr()), 'dt_deepfm_train_step')
        if backward:
            for p, g in self.grad_views:
                p.grad = g
            # with the in-step dedupe: rows looked up once keep their entry, the others travel as segments; fields = -2:
            # the entries of `rows` were applied inside the step, the optimizer only walks the segments
            self.emb.sparse_grads[self.key] = [SparseRowGrad(ids['rows'].view(-1), buf['grad_rows'].view(-1, self.D),
                                                             fields=(-2 if opt is not None else -1) if dedupe else None,
                                                             segments=_segments(ids, B, self.F) if dedupe else None)]
            if self.emb.uses_dense_grad(self.D):
                # small tables keep exact dense-Adam semantics: densify the row gradients
                g = torch.zeros_like(table)
                check(lib().dt_embedding_bwd_dense(ptr(ids['rows']), ptr(buf['grad_rows']), B * self.F, self.D,
                                                   ptr(g), stream_ptr()), 'dt_embedding_bwd_dense')
                table.grad = g
                self.emb.sparse_grads.pop(self.key, None)
        return self.loss_view, logit


_DCN_ACC_NAMES = ['dW1', 'dW2', 'db1', 'db2', 'dw3', 'dwo', 'dbo', 'loss', 'dgamma', 'dbeta', 'dcw', 'dcb']


class FusedDCN(FusedDeepFM):
    """Whole-step executor for the DCN graph (nets ['dcn_nets'], deepnets.py:194-207): the DeepFM kernel sequence with the
    Cross network (layers.py:428-436) running on the BN'd tile inside the tower kernel -> `dt_dcn_train_step`."""

    NETS = {'dcn_nets'}

    @classmethod
    def eligible(cls, dm):
        c = dm.config
        try:
            if list(c.nets) != ['dcn_nets'] or dm.var_len_categorical_columns:
                return False
            if _step_loss(dm) is None:
                return False
            if not (0 <= float(c.dense_dropout or 0) < 1) or not (0 <= float(c.embedding_dropout or 0) < 1):
                return False
            st = c.distribute_strategy
            if getattr(st, 'sharded_embeddings', False) and getattr(st, 'active', False):
                return False                      # row-owned tables: the layer-by-layer path
            if _tower_widths(c.dnn_params) is None:
                return False
            L = dm.model.layers_by_name
            # a single net: Concatenate([cross, dnn]) feeds task_output directly (deepmodel.py:286-301), no dense_logit_*
            need = ['emb_categorical_vars_all', 'bn_concat_emb_dense', 'dcn_cross_layer', 'dcn_dense_1', 'dcn_dense_2',
                    'task_output']
            if 'dense_logit_dcn_nets' in L:
                return False
            if any(n not in L for n in need):
                return False
            emb = L['emb_categorical_vars_all']
            if len(emb.groups) != 1 or len(dm.continuous_columns or []) > 1:
                return False
            D = emb.groups[0][0]
            F = len(emb.input_dims)
            Nd = sum(col.input_dim for col in (dm.continuous_columns or []))
            nl = int(L['dcn_cross_layer'].num_cross_layer)
            return bool(lib().dt_dcn_supported(max(int(getattr(dm, '_batch_hint', 0) or 0), 1), F, D, Nd, 128, 64, nl))
        except Exception:
            return False

    def __init__(self, dm):
        self.dm = dm
        L = dm.model.layers_by_name
        self.emb = L['emb_categorical_vars_all']
        self.bn = L['bn_concat_emb_dense']
        self.cross = L['dcn_cross_layer']
        self.nl = int(self.cross.num_cross_layer)
        self.d1, self.d2 = L['dcn_dense_1'], L['dcn_dense_2']
        self.out = L['task_output']        # kernel [C + H2, 1]: the step's w3; its w_out is the constant 1
        self.D = self.emb.groups[0][0]
        self.F = len(self.emb.input_dims)
        self.Nd = sum(col.input_dim for col in (dm.continuous_columns or []))
        self.C = self.F * self.D + self.Nd
        self.key = f'd{self.D}'
        self.device = self.emb.tables[self.key].device
        self.one = torch.ones(4, dtype=torch.float32, device=self.device)
        n_acc = lib().dt_dcn_accum_floats(self.F, self.D, self.Nd, self.nl)
        offs = (ctypes.c_int64 * 12)()
        check(lib().dt_dcn_accum_offsets(self.F, self.D, self.Nd, self.nl, ctypes.cast(offs, ctypes.c_void_p)),
              'dt_dcn_accum_offsets')
        self.off = dict(zip(_DCN_ACC_NAMES, [int(v) for v in offs]))
        self.accum = torch.zeros(n_acc, dtype=torch.float32, device=self.device)
        self._bufs = {}
        a, o, C, nl = self.accum, self.off, self.C, self.nl
        H1, H2 = _tower_widths(dm.config.dnn_params)
        self.grad_views = [
            (self.d1.kernel, a[o['dW1']:o['dW1'] + C * TILE_H1].view(C, TILE_H1)[:, :H1]),
            (self.d2.kernel, a[o['dW2']:o['dW2'] + TILE_H1 * TILE_H2].view(TILE_H1, TILE_H2)[:H1, :H2]),
            (self.d1.bias, a[o['db1']:o['db1'] + H1]),
            (self.d2.bias, a[o['db2']:o['db2'] + H2]),
            (self.out.kernel, a[o['dw3']:o['dw3'] + C + H2].view(C + H2, 1)),      # [w3c | w3d], w3d's pad at the end
            (self.bn.gamma, a[o['dgamma']:o['dgamma'] + C]),
            (self.bn.beta, a[o['dbeta']:o['dbeta'] + C]),
            (self.cross.kernel_stack, a[o['dcw']:o['dcw'] + nl * C].view(nl, C)),
            (self.cross.bias_stack, a[o['dcb']:o['dcb'] + nl * C].view(nl, C)),
        ]
        if self.out.bias is not None:
            self.grad_views.append((self.out.bias, a[o['dbo']:o['dbo'] + 1]))
        self.loss_view = a[o['loss']:o['loss'] + 1]
        self.emb_dropout = float(dm.config.embedding_dropout or 0)
        # dense_dropout (config.py:83): Dropout on the continuous input columns, masked where kernel A packs them
        self.dense_dropout = float(dm.config.dense_dropout or 0) if self.Nd else 0.0
        seed = int(torch.randint(1, 2 ** 31 - 1, (1,)).item())
        self.drop_seed = torch.tensor([seed], dtype=torch.int32, device=self.device)
        self.dedupe = os.environ.get('DT_AMD_FUSED_DEDUPE', '1') != '0'
        self.tower_flag = _tower_mfma_flag(dm.config.dnn_params)
        # diagnostic: s_memtime phase stamps into the workspace (tools/phase_times.py sets DT_AMD_STEP_STAMPS=1)
        self.diag_flag = _lib.DT_STEP_STAMPS if os.environ.get('DT_AMD_STEP_STAMPS') == '1' else 0
        # parameters mirror the gradient layout in one flat buffer (one optimizer launch, see FusedDeepFM); W1 / W2 precede
        # the [C + 64] output kernel, so they keep their 16-byte alignment whatever C is (the kernels read w3 with scalar loads)
        self.flat_params = torch.zeros_like(self.accum)
        members = _mirror_in_flat(self.flat_params, a, self.grad_views)
        n_flat = o['dcb'] + nl * C
        opt = getattr(dm, 'optimizer', None)
        if opt is not None and hasattr(opt, 'register_flat_group'):
            opt.register_flat_group(self.flat_params, self.accum, members, n_flat)
        dm.model._dt_flat_grad = self.accum

    def _buffers(self, B):
        b = self._bufs.get(B)
        if b is None:
            nbytes = lib().dt_dcn_workspace_bytes(B, self.F, self.D, self.Nd, self.nl)
            if nbytes < 0:
                raise _lib.DtHipError('fused DCN step: unsupported shape')
            dev = self.device
            b = {'ws': torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev),   # zero-filled once: the batch-sum accumulators
                 'logit': torch.empty((B, 1), dtype=torch.float32, device=dev),
                 'rows': torch.empty((B, self.F), dtype=torch.int64, device=dev),
                 'grad_rows': torch.empty((B, self.F, self.D), dtype=torch.float32, device=dev),
                 'dedupe': torch.zeros((lib().dt_deepfm_dedupe_bytes(B, self.F) + 7) // 8, dtype=torch.int64,
                                       device=dev),
                 'dedupe_slots': lib().dt_deepfm_dedupe_slots(B, self.F)}
            self._bufs[B] = b
        return b

    def run(self, idx, dense, y, backward=True, apply_rows=False, sample_weight=None, logit_out=None, slot=0,
            preelected=False, next_ids=None, prepared=False):
        self.dm.model._dt_sharded_step = False
        B = idx.shape[0]
        buf = self._buffers(B)
        idx = idx.contiguous()
        kind = _lib.DT_IDX_F32 if idx.dtype == torch.float32 else _lib.DT_IDX_I32
        if idx.dtype not in (torch.float32, torch.int32):
            idx = idx.to(torch.int32)
        dense = None if dense is None else dense.contiguous()
        y = y.reshape(-1).contiguous()
        sw = _row_weights(sample_weight, y)
        table = self.emb.tables[self.key]
        training = self.dm.model.training
        logit = buf['logit']
        if logit_out is not None:
            if logit_out.shape != logit.shape or logit_out.dtype != logit.dtype or not logit_out.is_contiguous():
                raise ValueError(f'logit_out must be a contiguous float32 {tuple(logit.shape)} tensor')
            logit = logit_out
        dedupe = _dedupe_in_step(self, B, backward)
        opt = _rows_in_step(self, B, backward, apply_rows)
        ids = self._slot_buffers(B, slot) if (slot and dedupe) else buf        # this step's rows / segment buffers
        pre = _lib.DT_STEP_PREELECTED if (preelected and dedupe and backward) else 0
        if preelected and not pre:
            raise _lib.DtHipError('a pre-elected step needs the in-step dedupe (backward, single process)')
        nxt = (None, None, None)
        if prepared or next_ids is not None:
            if opt is None or not dedupe:
                raise _lib.DtHipError('chained steps need the in-step dedupe and optimizer (can_chain)')
            if prepared:
                pre |= _lib.DT_STEP_PREPARED
            if next_ids is not None:
                nidx, nslot = next_ids
                if nidx.dtype != idx.dtype or nidx.shape != idx.shape or not nidx.is_contiguous() or not nslot or nslot == slot:
                    raise ValueError('next_ids: (contiguous ids like this step\'s, a slot of their own)')
                nb = self._slot_buffers(B, nslot)
                nxt = (ptr(nidx), ptr(nb['rows']), ptr(nb['dedupe']))
        head = (ptr(idx), kind, ptr(table), ptr(getattr(self.emb, f'row_offset_{self.key}')),
                ptr(getattr(self.emb, f'vocab_{self.key}')), ptr(dense), ptr(y), B, self.F, self.D, self.Nd,
                ptr(self.cross.kernel_stack), ptr(self.cross.bias_stack), self.nl, ptr(self.bn.gamma), ptr(self.bn.beta),
                ptr(self.bn.moving_mean) if training else None, ptr(self.bn.moving_variance) if training else None,
                float(self.bn.epsilon), float(self.bn.momentum), ptr(self.d1.kernel), ptr(self.d1.bias),
                ptr(self.d2.kernel), ptr(self.d2.bias), ptr(self.out.kernel), ptr(self.one), ptr(self.out.bias),
                ptr(logit), ptr(ids['rows']), ptr(buf['grad_rows']), ptr(self.accum), ptr(buf['ws']),
                ptr(self.emb.oob_count) if self.emb.check_oob else None,
                ptr(ids['dedupe']) if dedupe else None, buf['dedupe_slots'])
        if opt is not None:      # the whole optimizer step inside the train step (see FusedDeepFM.run)
            slots = opt._st(table, rows=True)
            flat = getattr(opt, '_flat', None)
            whole = (flat is not None and flat[0] is self.flat_params and flat[1] is self.accum and
                     os.environ.get('DT_AMD_STEP_IN_STEP', '1') != '0' and
                     all(id(p) in flat[5] for p in opt.params if p is not table))
            dn = (ptr(flat[0]), ptr(flat[2]), ptr(flat[3]), int(flat[4]), float(opt.lr)) if whole else (None, None, None, 0, 0.0)
            check(lib().dt_dcn_train_step_adam(
                *head, 2 | _step_loss(self.dm) | self.tower_flag | self.diag_flag | pre, self.emb_dropout if training else 0.0, ptr(self.drop_seed),
                self.dense_dropout if training else 0.0, ptr(sw), ptr(slots['m']), ptr(slots['v']), int(slots['m'].stride(0)), ptr(opt._state_tensor(table.device)), 0.0,
                opt.b1, opt.b2, opt.eps, *dn, *nxt, stream_ptr()), 'dt_dcn_train_step_adam')
            if whole:
                opt.applied_in_step()
        else:
            check(lib().dt_dcn_train_step(
                *head, (2 if backward else 1) | _step_loss(self.dm) | ((self.tower_flag | self.diag_flag) if backward else 0) | pre,
                self.emb_dropout if training else 0.0,
                ptr(self.drop_seed), self.dense_dropout if training else 0.0, ptr(sw), stream_ptr()), 'dt_dcn_train_step')
        if backward:
            for p, g in self.grad_views:
                p.grad = g
            self.emb.sparse_grads[self.key] = [SparseRowGrad(ids['rows'].view(-1), buf['grad_rows'].view(-1, self.D),
                                                             fields=(-2 if opt is not None else -1) if dedupe else None,
                                                             segments=_segments(ids, B, self.F) if dedupe else None)]
            if self.emb.uses_dense_grad(self.D):
                # small tables keep exact dense-Adam semantics: densify the row gradients
                g = torch.zeros_like(table)
                check(lib().dt_embedding_bwd_dense(ptr(ids['rows']), ptr(buf['grad_rows']), B * self.F, self.D,
                                                   ptr(g), stream_ptr()), 'dt_embedding_bwd_dense')
                table.grad = g
                self.emb.sparse_grads.pop(self.key, None)
        return self.loss_view, logit


def make_fused_plan(dm):
    if not fused_enabled() or dm.model is None:
        return None
    if FusedDeepFM.eligible(dm):
        return FusedDeepFM(dm)
    if FusedDCN.eligible(dm):
        return FusedDCN(dm)
    return None
