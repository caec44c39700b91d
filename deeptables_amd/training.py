# -*- coding:utf-8 -*-
"""Train-step glue around the layers hot path: losses, Keras-semantics Adam
(dense + row-sparse embedding update, HIP kernels), metrics, batching, and the data-parallel
gradient exchange hook.  This is what `keras.Model.fit` does for the reference
(deeptables/models/deepmodel.py:114-129, :319-346)."""
import math

import numpy as np
import torch

import os

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .utils import consts

# `DeepModel.fit(steps_per_execution=None)` falls back to this (Keras' `model.compile(steps_per_execution=...)`,
# deepmodel.py:319-346): 'auto' = k consecutive train steps per captured hipGraph (compiled.CompiledTrainLoop) when the
# graph has a fused whole-step plan and the feed is device resident, eager steps otherwise; an int forces k; 1 = eager
DEFAULT_STEPS_PER_EXECUTION = os.environ.get('DT_AMD_STEPS_PER_EXECUTION', 'auto')
if DEFAULT_STEPS_PER_EXECUTION != 'auto':
    DEFAULT_STEPS_PER_EXECUTION = int(DEFAULT_STEPS_PER_EXECUTION)


# ---------------------------------------------------------------------------------------------
# losses (keras.losses.* selected by DeepModel.__compile_model, deepmodel.py:324-338)
# ---------------------------------------------------------------------------------------------
_UNIT_GRAD = {}


def unit_grad(device):
    """the constant 1.0 `loss.backward()` would allocate and fill at every step: a cached 0-dim tensor per device.  A loss
    function that sees it as its incoming gradient (same storage) skips the multiplication by it."""
    key = str(device)
    if key not in _UNIT_GRAD:
        _UNIT_GRAD[key] = torch.ones((), dtype=torch.float32, device=device)
    return _UNIT_GRAD[key]


def _is_unit_grad(g):
    u = _UNIT_GRAD.get(str(g.device))
    return u is not None and g.dim() == 0 and g.data_ptr() == u.data_ptr()


class _BceFromLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, y):
        z, y = z.contiguous(), y.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        check(lib().dt_bce_logits(ptr(z), ptr(y), z.numel(), ptr(loss), ptr(dz), stream_ptr()), 'dt_bce_logits')
        ctx.save_for_backward(dz)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return (dz if _is_unit_grad(g) else dz * g), None


def bce_from_logits(logit, y):
    """BinaryCrossentropy on a sigmoid output evaluated from the logits (stable form); one HIP launch for the
    loss and its gradient on the GPU."""
    z = logit.reshape(y.shape[0], -1)
    y = y.reshape(z.shape).to(z.dtype)
    if z.is_cuda and z.dtype == torch.float32:
        return _BceFromLogits.apply(z, y)
    return (torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-z.abs()))).mean()


def weighted_loss(loss_name, logit, y, w):
    """Keras `sample_weight` / `class_weight` semantics (keras Model.fit -> compile loss with reduction
    SUM_OVER_BATCH_SIZE): per-sample loss times its weight, summed, divided by the batch size.  Element-wise torch on
    [B, outputs] — not on the hot path (the fused steps and the one-launch BCE are used only without weights)."""
    B = y.shape[0]
    z = logit.reshape(B, -1)
    t = y.reshape(B, -1).to(z.dtype)
    if loss_name == 'binary_crossentropy':
        per = (torch.clamp(z, min=0) - z * t + torch.log1p(torch.exp(-z.abs()))).mean(-1)
    elif loss_name == 'categorical_crossentropy':
        per = -(torch.log_softmax(z, dim=-1) * t).sum(-1)
    elif loss_name in ('mse', 'mean_squared_error'):
        per = ((z - t) ** 2).mean(-1)
    else:
        raise ValueError(f'sample / class weights with loss {loss_name!r}')
    return (per * w.reshape(B).to(z.dtype)).sum() / B


def categorical_ce_from_logits(logit, y_onehot):
    return -(torch.log_softmax(logit, dim=-1) * y_onehot).sum(-1).mean()


def mse(pred, y):
    return ((pred.reshape(y.shape[0], -1) - y.reshape(y.shape[0], -1).to(pred.dtype)) ** 2).mean()


# ---------------------------------------------------------------------------------------------
# Keras Adam
# ---------------------------------------------------------------------------------------------
class KerasAdam:
    """keras.optimizers.Adam(learning_rate=1e-3, beta_1=.9, beta_2=.999, epsilon=1e-7):
        lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  p -= lr_t*m/(sqrt(v)+eps).
    Dense parameters: one fused HIP launch each — or ONE launch for a whole registered flat group (the fused
    train-step plans keep parameters and gradients of all dense layers in two contiguous buffers).
    Packed embedding tables with a sparse gradient (MultiColumnEmbedding.sparse_grads): duplicate lookups merged
    through a per-step hash of the row ids, then a row-sparse ("lazy") update of only the rows touched
    this step — a documented deviation from Keras' dense semantics for tables too large to sweep every step.
    The step counter and lr_t live on the device (dt_adam_advance), so a captured hipGraph of the step replays
    with the right bias correction."""

    supports_row_segments = True      # takes SparseRowGrad.segments (dt_adam_rows_step_seg)
    supports_rows_in_step = True      # a fused step may apply the update of the rows looked up once itself (fields = -2)
    _name = 'Adam'

    def __init__(self, params, embedding_layers=(), learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.params = [p for p in params if p.requires_grad]
        self.state = {}
        self.embedding_layers = list(embedding_layers)
        self._dev_state = None        # device-resident step state (DT_ADAM_STATE_BYTES)
        # optional callable run right before the dense updates of a step (after the table updates were launched):
        # the data-parallel strategy uses it to wait for an all-reduce it started asynchronously
        self.pre_dense_hook = None
        self._applied_in_step = False  # a fused step ran this step's whole update itself (applied_in_step)
        self._flat = None             # (flat_param, flat_grad, m, v, n, {id(param): offset})
        self._flat_views = None       # [(param, grad view)] when gradients may be accumulated straight into flat_grad

    # -- step counter (device resident: int32 t_next, float lr_t, block-arrival counters) -----------------
    def _state_tensor(self, device):
        if self._dev_state is None:
            self._dev_state = torch.zeros(68, dtype=torch.int32, device=device)     # DT_ADAM_STATE_BYTES / 4
            check(lib().dt_adam_state_init(ptr(self._dev_state), self.lr, self.b1, self.b2, 0, stream_ptr()),
                  'dt_adam_state_init')
        return self._dev_state

    @property
    def t(self):
        """number of completed steps (keras `optimizer.iterations`)"""
        return 0 if self._dev_state is None else int(self._dev_state[0].item()) - 1

    @t.setter
    def t(self, value):
        if self._dev_state is None:
            if not self.params:
                return
            self._state_tensor(self.params[0].device)
        check(lib().dt_adam_state_init(ptr(self._dev_state), self.lr, self.b1, self.b2, int(value), stream_ptr()),
              'dt_adam_state_init')

    def _st(self, p, rows=False):
        """slots of parameter p.  rows=True (the row-sparse table update): m and v of a row side by side in ONE
        [V, 2, D] array (a 128-byte line at D = 16), so the update touches two random locations per row instead of
        three; `m` / `v` are its strided views.  The dense update needs contiguous slots and converts back."""
        s = self.state.get(id(p))
        if s is None:
            if rows and p.dim() == 2:
                mv = torch.zeros((p.shape[0], 2, p.shape[1]), dtype=p.dtype, device=p.device)
                s = {'m': mv[:, 0, :], 'v': mv[:, 1, :], 'mv': mv}
            else:
                s = {'m': torch.zeros_like(p), 'v': torch.zeros_like(p)}
            self.state[id(p)] = s
        elif not rows and 'mv' in s:
            s = {'m': s['m'].contiguous(), 'v': s['v'].contiguous()}
            self.state[id(p)] = s
        return s

    def register_flat_group(self, flat_param, flat_grad, members, n, grad_views=False):
        """members: [(param, offset, numel[, (shape, strides)])] whose .data are views of flat_param and whose fused-plan
        gradients are the matching views of flat_grad (contiguous unless shape / strides are given: a fused plan keeps
        narrow towers inside zero-padded slabs).  Their m/v become the same views of one flat m/v (existing state is kept).
        grad_views=True: `zero_grad()` zeroes flat_grad and hands its views out as `.grad` (generic autograd path)."""
        m = torch.zeros_like(flat_param)
        v = torch.zeros_like(flat_param)

        def view_of(flat, p, off, cnt, layout):
            if layout is None:
                return flat[off:off + cnt].view(p.shape)
            return torch.as_strided(flat, layout[0], layout[1], off)
        members = [(mb[0], mb[1], mb[2], mb[3] if len(mb) > 3 else None) for mb in members]
        for p, off, cnt, layout in members:
            old = self.state.get(id(p))
            mv, vv = view_of(m, p, off, cnt, layout), view_of(v, p, off, cnt, layout)
            if old is not None:
                mv.copy_(old['m'].reshape(p.shape))
                vv.copy_(old['v'].reshape(p.shape))
            self.state[id(p)] = {'m': mv, 'v': vv}
        self._flat = (flat_param, flat_grad, m, v, int(n), {id(p): off for p, off, _, _ in members})
        self._flat_views = [(p, view_of(flat_grad, p, off, cnt, layout)) for p, off, cnt, layout in members] \
            if grad_views else None
        for p, off, cnt, layout in members:
            p._dt_grad_view = None
        if grad_views:                       # same tensor objects in both places (`p.grad is p._dt_grad_view`)
            for p, view in self._flat_views:
                p._dt_grad_view = view

    def _flat_member_views(self):
        """{id(param): the parameter's view of the flat GRADIENT buffer} (same offset / shape / strides as its data)"""
        fp, fg = self._flat[0], self._flat[1]
        out = {}
        for p in self.params:
            if id(p) in self._flat[5]:
                off = (p.data.data_ptr() - fp.data_ptr()) // 4
                out[id(p)] = torch.as_strided(fg, tuple(p.data.shape), tuple(p.data.stride()), off)
        return out

    def _dense_launch(self, dense, sp, st, advance):
        """All dense tensors of the step in one launch (dt_adam_multi_step; chunks of 32 tensors)."""
        if not dense:
            return
        if len(dense) == 1:
            pp, gg, mm, vv, n = dense[0]
            check(lib().dt_adam_dense_step(ptr(pp), ptr(gg), ptr(mm), ptr(vv), n, 0.0, self.b1, self.b2, self.eps,
                                           sp, 1 if advance else 0, self.lr, st), 'dt_adam_dense_step')
            return
        import ctypes
        T = len(dense)
        arr = ctypes.c_void_p * T
        ps, gs, ms, vs = (arr(*[t[k].data_ptr() for t in dense]) for k in range(4))
        ns = (ctypes.c_int64 * T)(*[int(t[4]) for t in dense])
        check(lib().dt_adam_multi_step(T, ctypes.cast(ps, ctypes.c_void_p), ctypes.cast(gs, ctypes.c_void_p),
                                       ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(vs, ctypes.c_void_p),
                                       ctypes.cast(ns, ctypes.c_void_p), 0.0, self.b1, self.b2, self.eps, sp,
                                       1 if advance else 0, self.lr, st), 'dt_adam_multi_step')

    def zero_grad(self, flat=True):
        """flat=True: members of the registered flat group get their (zeroed) views of the flat gradient buffer as
        `.grad` — backward kernels and autograd accumulate straight into it; flat=False: the caller (a fused step)
        fills the flat buffer itself."""
        # _forward_backward(apply_rows=True) is a promise that step() follows at once: the rows looked up once were already
        # updated inside the step.  A gradient of that kind still pending here means the promise was broken (the segments and
        # the dense parameters never got their half of the update): fail loudly instead of training on half-applied steps
        if not self._applied_in_step:
            for layer in self.embedding_layers:
                for grads in layer.sparse_grads.values():
                    if any(getattr(g, 'fields', None) == -2 for g in grads):
                        raise RuntimeError('_forward_backward(apply_rows=True) was not followed by optimizer.step(): the '
                                           'table rows looked up once are updated, the segments and dense parameters are not')
        self._applied_in_step = False
        for p in self.params:
            p.grad = None
        if flat and self._flat is not None and self._flat_views is not None:
            self._flat[1].zero_()
            for p, view in self._flat_views:
                p.grad = view
        for layer in self.embedding_layers:
            layer.sparse_grads.clear()

    def applied_in_step(self):
        """A fused train step (fused.FusedDeepFM with dt_deepfm_train_step_adam) has applied THIS step's whole update —
        table rows, segments, every dense element — and advanced the device step state inside its own launches: the
        `step()` call that follows only drops the gradients."""
        self._applied_in_step = True

    def step(self):
        if not self.params:
            return
        if self._applied_in_step:
            self._applied_in_step = False
            for layer in self.embedding_layers:
                layer.sparse_grads.clear()
            return
        dev_state = self._state_tensor(self.params[0].device)
        st = stream_ptr()
        sp = ptr(dev_state)
        # every launch of the step reads lr_t from the device state; the LAST one advances it (no extra launch)
        dense = []                                        # (param, grad, m, v, n)
        flat_done = set()
        if self._flat is not None:
            fp, fg, fm, fv, n, members = self._flat
            base = fg.data_ptr()
            in_flat = [p for p in self.params if id(p) in members and p.grad is not None and
                       p.grad.data_ptr() == base + 4 * members[id(p)]]
            if len(in_flat) == len(members):       # every member's gradient is its flat view: one launch
                dense.append((fp, fg, fm, fv, n))
                flat_done = set(members)
            else:
                # gradients that are NOT the flat views (the layer-by-layer path after a fused plan re-homed the
                # parameters: a weighted step, DT_AMD_FUSED toggled, ...).  Members of a plan with a narrow tower are
                # STRIDED views of zero-padded slabs (fused._mirror_in_flat), so the per-tensor kernels — which take
                # (pointer, numel) — must not see them: scatter every member's gradient into its place of the flat
                # gradient buffer and update the whole group with the one flat launch.  Pads have zero gradient and
                # zero moments: they stay exactly zero.
                with_grad = [p for p in self.params if id(p) in members and p.grad is not None]
                if len(with_grad) == len(members):
                    views = self._flat_member_views()
                    if not in_flat:
                        fg.zero_()
                    flat_ids = {id(p) for p in in_flat}
                    for p in with_grad:
                        if id(p) not in flat_ids:
                            views[id(p)].copy_(p.grad.reshape(views[id(p)].shape))
                    dense.append((fp, fg, fm, fv, n))
                    flat_done = set(members)
        deferred = []                                     # (param, slots, contiguous copies) of strided tensors
        for p in self.params:
            if p.grad is None or id(p) in flat_done:
                continue
            s = self._st(p)
            if p.data.is_contiguous() and s['m'].is_contiguous() and s['v'].is_contiguous():
                dense.append((p.data, p.grad.contiguous(), s['m'], s['v'], p.numel()))
            else:
                # a strided parameter / slot on its own (only some members of a flat group carry a gradient): update
                # contiguous copies and write them back after the launches below
                pc, mc, vc = p.data.contiguous(), s['m'].contiguous(), s['v'].contiguous()
                dense.append((pc, p.grad.contiguous(), mc, vc, p.numel()))
                deferred.append((p, s, pc, mc, vc))
        sparse = []
        for layer in self.embedding_layers:
            for key, grads in layer.sparse_grads.items():
                sparse.append((layer, key, grads))
        # launch order: dense updates that cannot ride along, then the table updates; the last table update carries
        # one dense update (normally the model's flat buffer) in its trailing blocks and advances the state
        hook, self.pre_dense_hook = self.pre_dense_hook, None
        dense_after = hook is not None and bool(sparse)     # table updates first, overlapping the pending reduce
        tail = dense.pop(0) if (sparse and dense and not dense_after) else None
        if hook is not None and not dense_after:
            hook()
        if not dense_after:
            self._dense_launch(dense, sp, st, advance=not sparse)
        if not sparse and not dense:
            check(lib().dt_adam_advance(sp, self.lr, self.b1, self.b2, st), 'dt_adam_advance')
        for i, (layer, key, grads) in enumerate(sparse):
            table = layer.tables[key]
            s = self._st(table, rows=True)
            D = table.shape[1]
            if len(grads) == 1:
                rows, values = grads[0].rows, grads[0].values
            else:
                rows = torch.cat([g.rows.reshape(-1) for g in grads])
                values = torch.cat([g.values.reshape(-1, D) for g in grads])
            n = rows.numel()
            fields = len(dict(layer.groups)[D]) if hasattr(layer, 'groups') else 0
            hints = {getattr(g, 'fields', None) for g in grads}
            if hints != {None}:                           # an explicit layout promise overrides the layer default
                fields = hints.pop() if len(hints) == 1 else 0
                fields = 0 if fields is None else int(fields)
            if fields == -2 and len(grads) != 1:
                raise ValueError('a sparse gradient whose rows were applied inside the step cannot be concatenated')
            if fields == -1 and len(grads) != 1:
                fields = 0                                # distinct within each piece only
            values = values if values.is_contiguous() else values.contiguous()
            if fields in (-1, -2):
                slots = mark = None
                n_slots = 0
            else:
                n_slots = lib().dt_adam_rows_slots(n)
                if s.get('n_slots', 0) < n_slots or s['mark'].numel() < n:
                    s['slots'] = torch.zeros(n_slots, dtype=torch.int64, device=table.device)
                    s['mark'] = torch.empty(n, dtype=torch.int32, device=table.device)
                    s['n_slots'] = n_slots
                slots, mark, n_slots = s['slots'], s['mark'], s['n_slots']
            is_last = i == len(sparse) - 1 and not (dense_after and dense)
            tl = tail if (is_last and tail is not None) else (None, None, None, None, 0)
            seg = grads[0].segments if (len(grads) == 1 and getattr(grads[0], 'segments', None) is not None) else None
            if seg is None and any(getattr(g, 'segments', None) is not None for g in grads):
                raise ValueError('a segmented sparse gradient cannot be concatenated with other pieces')
            sg = [ptr(t) for t in seg[:5]] + [int(seg[5]), int(seg[6])] if seg is not None else [None] * 5 + [0, 0]
            check(lib().dt_adam_rows_step_seg(ptr(table.data), ptr(s['m']), ptr(s['v']), ptr(rows), ptr(values), n,
                                              D, fields, ptr(slots), n_slots, ptr(mark), 0.0,
                                              self.b1, self.b2, self.eps, sp, ptr(tl[0]), ptr(tl[1]), ptr(tl[2]),
                                              ptr(tl[3]), tl[4], 1 if is_last else 0, self.lr, *sg,
                                              int(s['m'].stride(0)), st),
                  'dt_adam_rows_step_seg')
        if dense_after:
            hook()
            self._dense_launch(dense, sp, st, advance=True)
        for p, s, pc, mc, vc in deferred:
            p.data.copy_(pc)
            s['m'].copy_(mc)
            s['v'].copy_(vc)
        for layer in self.embedding_layers:
            layer.sparse_grads.clear()


class SGD:
    _name = 'SGD'

    def __init__(self, params, embedding_layers=(), learning_rate=0.01):
        self.lr = learning_rate
        self.params = [p for p in params if p.requires_grad]
        self.embedding_layers = list(embedding_layers)

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        for layer in self.embedding_layers:
            layer.sparse_grads.clear()

    def step(self):
        st = stream_ptr()
        for p in self.params:
            if p.grad is not None:
                g = p.grad.contiguous()
                check(lib().dt_sgd_dense_step(ptr(p.data), ptr(g), p.numel(), self.lr, st), 'dt_sgd_dense_step')
        for layer in self.embedding_layers:
            for key, grads in layer.sparse_grads.items():
                table = layer.tables[key]
                for gr in grads:
                    vals = gr.values if gr.values.is_contiguous() else gr.values.contiguous()
                    check(lib().dt_sgd_rows_step(ptr(table.data), ptr(gr.rows), ptr(vals), gr.rows.numel(),
                                                 table.shape[1], self.lr, st), 'dt_sgd_rows_step')
            layer.sparse_grads.clear()


def make_optimizer(spec, params, embedding_layers):
    if spec == 'auto' or spec is None or (isinstance(spec, str) and spec.lower() == 'adam'):
        return KerasAdam(params, embedding_layers)
    if isinstance(spec, str) and spec.lower() == 'sgd':
        return SGD(params, embedding_layers)
    if callable(spec):
        return spec(params, embedding_layers)
    raise ValueError(f'Unsupported optimizer: {spec!r}')


def flatten_dense_parameters(model, optimizer, exclude=()):
    """Re-home every trainable dense parameter of `model` in ONE flat buffer (and its gradient in another), so that
    backward kernels accumulate in place (ops._grad_target), a data-parallel exchange is one all-reduce of the flat
    gradient and the optimizer is one launch.  `exclude`: parameters that stay on their own (embedding tables).
    Returns (flat_param, flat_grad) or None when the optimizer has no flat-group support."""
    if not hasattr(optimizer, 'register_flat_group'):
        return None
    skip = {id(p) for p in exclude}
    members, off = [], 0
    params = [p for p in model.parameters() if p.requires_grad and id(p) not in skip and p.dtype == torch.float32]
    if not params:
        return None
    for p in params:
        members.append((p, off, p.numel()))
        off += (p.numel() + 63) // 64 * 64            # 256-byte aligned starts (float4 / MFMA operand loads)
    device = params[0].device
    flat_param = torch.zeros(off, dtype=torch.float32, device=device)
    flat_grad = torch.zeros(off, dtype=torch.float32, device=device)
    with torch.no_grad():
        for p, o, n in members:
            flat_param[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat_param[o:o + n].view(p.shape)
    optimizer.register_flat_group(flat_param, flat_grad, members, off, grad_views=True)
    return flat_param, flat_grad


# ---------------------------------------------------------------------------------------------
# metrics (host side, outside any timed region)
# ---------------------------------------------------------------------------------------------
def metric_name(m):
    if isinstance(m, str):
        return m
    return getattr(m, 'name', None) or getattr(m, '__name__', str(m))


def compute_metric(m, y_true, y_prob, task):
    name = metric_name(m)
    key = name.lower()
    y_true = np.asarray(y_true)
    y_prob = np.asarray(y_prob)
    if callable(m) and not isinstance(m, str):
        return float(m(y_true, y_prob))
    if key in ('accuracy', 'acc'):
        if task == consts.TASK_MULTICLASS and y_prob.ndim == 2 and y_prob.shape[1] > 1:
            yt = y_true.argmax(-1) if y_true.ndim == 2 else y_true
            return float((y_prob.argmax(-1) == yt).mean())
        return float(((y_prob.reshape(-1) > 0.5).astype(np.int64) == y_true.reshape(-1).astype(np.int64)).mean())
    if key == 'auc':
        from sklearn.metrics import roc_auc_score
        try:
            if y_prob.ndim == 2 and y_prob.shape[1] > 1:
                return float(roc_auc_score(y_true, y_prob, multi_class='ovr'))
            return float(roc_auc_score(y_true.reshape(-1), y_prob.reshape(-1)))
        except ValueError:
            return float('nan')
    if key in ('mse', 'mean_squared_error'):
        return float(((y_prob.reshape(-1) - y_true.reshape(-1)) ** 2).mean())
    if key in ('rmse', 'rootmeansquarederror', 'root_mean_squared_error'):
        return float(np.sqrt(((y_prob.reshape(-1) - y_true.reshape(-1)) ** 2).mean()))
    if key in ('mae', 'mean_absolute_error'):
        return float(np.abs(y_prob.reshape(-1) - y_true.reshape(-1)).mean())
    raise ValueError(f'Unsupported metric: {name}')


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []

    def add(self, epoch, logs):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


# ---------------------------------------------------------------------------------------------
# data feed (replaces utils/dataset_generator.py:36-72 for in-memory frames)
# ---------------------------------------------------------------------------------------------
class TableBatches:
    """Input feed for in-memory frames (replaces `to_dataset`, utils/dataset_generator.py:36-72, which
    converts whole frames through `.tolist()` into tf.constants and feeds float32 ids).

    Ids are int32 end to end.  Two modes, same batches for the same seed:
      * resident (default when the table fits `max_resident_bytes`): the whole table lives in HBM, a shuffled epoch
        is one `randperm` on the device and every batch an index-select — no host work per step;
      * streamed: the table stays in PINNED host memory; batch k+1 is gathered on the host into a ring of pinned
        staging buffers and copied with an async H2D on a side stream while batch k trains; the compute stream
        waits on the copy's event only.
    Yields ([cat ids [B,F] int32, var-len id blocks..., dense blocks...], y)."""

    def __init__(self, X, y, categorical_columns, continuous_columns, device, task=None, num_classes=None,
                 cat_dtype=torch.int32, var_len_categorical_columns=None, resident=None,
                 max_resident_bytes=32 << 30, ring=3, sample_weight=None):
        self.n = len(X)
        self.weighted = sample_weight is not None      # per-row loss weights ride as the LAST column of y
        self.y_ndim = None
        self.device = torch.device(device)
        get = (lambda cols: X[cols].values) if hasattr(X, 'columns') else None
        host = []          # [(kind, host tensor)] in model input order: cat, var-len..., dense... (deepmodel.py:310)
        if categorical_columns:
            names = [c.name for c in categorical_columns]
            arr = get(names) if get else np.asarray(X['cat'])
            host.append(('cat', torch.as_tensor(np.ascontiguousarray(arr).astype(np.int64)).to(cat_dtype)))
        for c in var_len_categorical_columns or []:       # padded id lists, [N, max_elements_length]
            col = X[c.name]
            arr = np.array(col.tolist() if hasattr(col, 'tolist') else list(col))
            host.append(('var', torch.as_tensor(np.ascontiguousarray(arr).astype(np.int64)).to(cat_dtype)))
        for c in continuous_columns or []:
            arr = get(list(c.column_names)) if get else np.asarray(X[c.name])
            host.append(('cont', torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float32))))
        y_host = None
        if y is not None:
            y = np.asarray(y)
            if task == consts.TASK_MULTICLASS and y.ndim == 1:
                y = np.eye(num_classes, dtype=np.float32)[y.astype(np.int64)]
            y_host = torch.as_tensor(np.ascontiguousarray(y, dtype=np.float32))
            self.y_ndim = y_host.dim()                 # weights ride as an extra column: remember y's own rank
            if sample_weight is not None:
                w = torch.as_tensor(np.ascontiguousarray(np.asarray(sample_weight).reshape(-1), dtype=np.float32))
                if w.shape[0] != y_host.shape[0]:
                    raise ValueError(f'sample_weight has {w.shape[0]} entries for {y_host.shape[0]} rows')
                y_host = torch.cat([y_host.reshape(y_host.shape[0], -1), w[:, None]], 1).contiguous()
        nbytes = sum(t.numel() * t.element_size() for _, t in host) + (0 if y_host is None else y_host.numel() * 4)
        self.resident = (nbytes <= max_resident_bytes) if resident is None else bool(resident)
        self.kinds = [k for k, _ in host]
        if self.resident:
            self.blocks = [t.to(self.device) for _, t in host]
            self.y = None if y_host is None else y_host.to(self.device)
        else:
            pin = self.device.type == 'cuda'
            self.blocks = [t.pin_memory() if pin else t for _, t in host]
            self.y = None if y_host is None else (y_host.pin_memory() if pin else y_host)
            self._ring = max(2, int(ring))
            self._staging = None
            self._copy_stream = torch.cuda.Stream(self.device) if pin else None

    @classmethod
    def from_device(cls, blocks, kinds, y=None, y_ndim=None, weighted=False):
        """A resident feed over tensors that already live on the device (a synthetic table generated there, bench.py):
        blocks in model input order (cat ids int32 [N,F], var-len id blocks, continuous blocks float32), kinds the matching
        'cat' / 'var' / 'cont' tags, y float32 [N] or [N, outputs] (its last column the per-row weight when `weighted`)."""
        self = cls.__new__(cls)
        self.n = int(blocks[0].shape[0])
        self.weighted = bool(weighted)
        self.device = blocks[0].device
        self.kinds = list(kinds)
        self.blocks = [b.contiguous() for b in blocks]
        self.y = None if y is None else y.contiguous()
        self.y_ndim = y_ndim if y_ndim is not None else (None if y is None else y.dim())
        self.resident = True
        return self

    # kept for callers that look at the pieces
    @property
    def cat(self):
        return self.blocks[self.kinds.index('cat')] if 'cat' in self.kinds else None

    @property
    def var_lens(self):
        return [b for k, b in zip(self.kinds, self.blocks) if k == 'var']

    @property
    def conts(self):
        return [b for k, b in zip(self.kinds, self.blocks) if k == 'cont']

    def batch(self, sel):
        return [b[sel] for b in self.blocks], (None if self.y is None else self.y[sel])

    def _permutation(self, generator):
        """The same permutation in both modes: drawn on the device when there is one."""
        if self.device.type == 'cuda':
            return torch.randperm(self.n, device=self.device, generator=generator)
        return torch.randperm(self.n, generator=generator)

    def iterate(self, batch_size, shuffle, drop_remainder, generator=None):
        n = self.n
        stop = (n // batch_size) * batch_size if drop_remainder else n
        perm = self._permutation(generator) if shuffle else None
        if self.resident:
            for s in range(0, stop, batch_size):
                e = min(s + batch_size, n)
                yield self.batch(perm[s:e] if shuffle else slice(s, e))
            return
        yield from self._stream(batch_size, stop, None if perm is None else perm.cpu())

    # -- streamed mode ---------------------------------------------------------------------------------
    def _alloc_staging(self, batch_size):
        pin = self._copy_stream is not None
        srcs = self.blocks + ([] if self.y is None else [self.y])
        self._staging = []
        for _ in range(self._ring):
            hostbufs = [torch.empty((batch_size,) + tuple(t.shape[1:]), dtype=t.dtype, pin_memory=pin) for t in srcs]
            devbufs = [torch.empty_like(h, device=self.device) for h in hostbufs]
            ev = torch.cuda.Event() if pin else None
            self._staging.append((hostbufs, devbufs, ev, [None]))
        self._staging_batch = batch_size

    def _stage(self, slot, s, e, perm_host):
        """host gather into the slot's pinned buffers + async H2D on the copy stream; records the slot's event"""
        hostbufs, devbufs, ev, consumed = self._staging[slot]
        srcs = self.blocks + ([] if self.y is None else [self.y])
        k = e - s
        if consumed[0] is not None:
            consumed[0].synchronize()         # the step that used this slot's device buffers must be done
        for src, hb in zip(srcs, hostbufs):
            if perm_host is None:
                hb[:k].copy_(src[s:e])
            else:
                torch.index_select(src, 0, perm_host[s:e], out=hb[:k])
        if self._copy_stream is not None:
            with torch.cuda.stream(self._copy_stream):
                for hb, db in zip(hostbufs, devbufs):
                    db[:k].copy_(hb[:k], non_blocking=True)
                ev.record(self._copy_stream)
        else:
            for hb, db in zip(hostbufs, devbufs):
                db[:k].copy_(hb[:k])
        return k

    def _stream(self, batch_size, stop, perm_host):
        if self._staging is None or self._staging_batch != batch_size:
            self._alloc_staging(batch_size)
        starts = list(range(0, stop, batch_size))
        sizes = {}
        depth = self._ring - 1
        for j in range(min(depth, len(starts))):                       # prime the ring
            sizes[j] = self._stage(j % self._ring, starts[j], min(starts[j] + batch_size, self.n), perm_host)
        for i, s in enumerate(starts):
            slot = i % self._ring
            hostbufs, devbufs, ev, consumed = self._staging[slot]
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)  # compute waits for this batch's copy only
            k = sizes.pop(i)
            outs = [d[:k] for d in devbufs]
            nxt = i + depth
            if nxt < len(starts):                                      # overlap: stage a later batch now
                sizes[nxt] = self._stage(nxt % self._ring, starts[nxt], min(starts[nxt] + batch_size, self.n),
                                         perm_host)
            yb = outs.pop() if self.y is not None else None
            try:
                yield outs, yb
            finally:
                # mark where the consumer finished with the slot — also when it stops at this batch and closes the
                # generator (fit breaks out at steps_per_epoch): the next iterate() must not restage the slot against a
                # stale event while this step's kernels may still read the device buffers
                if ev is not None:
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.device))
                    consumed[0] = done
