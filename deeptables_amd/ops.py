# -*- coding:utf-8 -*-
"""torch.autograd.Function wrappers over the C-ABI kernels (include/dt_hip.h).

Each Function is the forward+backward of one reference `Layer.call` in
deeptables/models/layers.py; torch is only the carrier of device memory, streams and the
autograd tape.  All functions require CUDA(HIP) tensors — no CPU fallback.
"""
import os

import torch

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream_ptr


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _idx_kind(idx):
    if idx.dtype == torch.float32:
        return _lib.DT_IDX_F32
    if idx.dtype == torch.int32:
        return _lib.DT_IDX_I32
    raise TypeError(f'categorical ids must be float32 (reference contract) or int32, got {idx.dtype}')


# ------------------------------------------------------------------------------------------------
# MultiColumnEmbedding.call — deeptables/models/layers.py:889-904
# ------------------------------------------------------------------------------------------------
def _grad_target(p):
    """-> (buffer the backward kernel ACCUMULATES this parameter's gradient into, value `backward` returns for it).
    Parameters of a model whose dense weights were flattened (training.flatten_dense_parameters) carry a pre-zeroed
    view of the model's ONE flat gradient buffer: the kernel adds straight into it and autograd is told there is
    nothing left to accumulate (None) — no zero-fill, no AccumulateGrad add, and the optimizer updates the whole
    model with one launch.  Anything else gets a fresh zero tensor that is returned as the gradient."""
    view = getattr(p, '_dt_grad_view', None)
    g = p.grad if view is not None else None
    if g is not None and g.data_ptr() == view.data_ptr() and g.shape == view.shape:
        return view, None
    z = torch.zeros_like(p)
    return z, z


class SparseRowGrad:
    """The (indices, values) pair TF calls IndexedSlices: gradient of a packed embedding table."""

    __slots__ = ('rows', 'values', 'fields', 'segments')

    def __init__(self, rows, values, fields=None, segments=None):
        self.rows = rows        # int64 [n]   packed row ids, -1 = out-of-range lookup (or a member of a segment)
        self.values = values    # float32 [n, D]
        # layout promise for the optimizer's per-field dedupe: None = the owning layer's default ([.., F] lookups of
        # a packed table), 0 = no field structure (use the global hash), -1 = every row >= 0 appears once
        self.fields = fields
        # rows looked up several times, as the fused steps hand them over (csrc/deepfm.hip DedupeWs): device tensors
        # (nseg int32 [regions], seg_row int64, seg_off, seg_cnt, seg_list int32, regions, cap); the members' entries of
        # `rows` are -1 and the
        # optimizer's segment waves sum their `values` rows (dt_adam_rows_step_seg)
        self.segments = segments

    def expanded(self):
        """(rows, values) with the segment members' row ids restored — one entry per lookup again (host sync; for
        consumers that do not take segments, and for tests)."""
        if self.segments is None:
            return self.rows, self.values
        nseg, srow, soff, scnt, slist, regions, cap = self.segments
        rows = self.rows.reshape(-1).clone()
        valid = (torch.arange(cap, device=rows.device)[None, :] < nseg.long()[:, None]).reshape(-1)
        if bool(valid.any()):
            cnt, off, srow_v = scnt[:regions * cap][valid].long(), soff[:regions * cap][valid].long(), srow[:regions * cap][valid]
            start = torch.cumsum(cnt, 0) - cnt
            pos = torch.arange(int(cnt.sum().item()), device=rows.device) - torch.repeat_interleave(start, cnt)
            occ = slist[(torch.repeat_interleave(off, cnt) + pos)].long()
            rows[occ] = torch.repeat_interleave(srow_v, cnt)
        return rows, self.values


def merge_segments_(grad):
    """In place: the segments of a fused step's SparseRowGrad are summed into their first member's entry
    (dt_rows_merge_segments) -> SparseRowGrad over the same (rows, values) with one entry per distinct row of the step, the
    other members keeping row -1, in the original [.., fields] layout (fields = -1: rows are distinct)."""
    seg = grad.segments
    if seg is None:
        return grad
    D = grad.values.shape[-1]
    assert grad.values.is_contiguous() and grad.rows.is_contiguous()
    check(lib().dt_rows_merge_segments(ptr(grad.rows), ptr(grad.values), D, *[ptr(t) for t in seg[:5]], int(seg[5]),
                                       int(seg[6]), stream_ptr()), 'dt_rows_merge_segments')
    return SparseRowGrad(grad.rows, grad.values, fields=-1)


def compact_rows(grad, cap, scale=1.0, out_rows=None, out_vals=None, counter=None):
    """SparseRowGrad of a fused step (entries for the rows looked up once + segments) -> SparseRowGrad of UNIQUE
    (row, summed gradient * scale) entries packed at the front of [cap] buffers (row -1 = unused slot): the bucket a
    data-parallel rank puts on the wire (dt_rows_compact).  Returns (grad, counter): counter[0] = entries produced,
    counter[1] = entries that did not fit `cap` (a device tensor: read it outside the step)."""
    rows = grad.rows.reshape(-1)
    D = grad.values.shape[-1]
    values = grad.values.reshape(-1, D)
    values = values if values.is_contiguous() else values.contiguous()
    dev = rows.device
    if out_rows is None:
        out_rows = torch.empty((cap,), dtype=torch.int64, device=dev)
    if out_vals is None:
        out_vals = torch.empty((cap, D), dtype=torch.float32, device=dev)
    if counter is None:
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
    seg = grad.segments
    sg = [ptr(t) for t in seg[:5]] + [int(seg[5]), int(seg[6])] if seg is not None else [None] * 5 + [0, 0]
    check(lib().dt_rows_compact(ptr(rows), ptr(values), rows.numel(), D, *sg, float(scale), int(cap), ptr(out_rows),
                                ptr(out_vals), ptr(counter), stream_ptr()), 'dt_rows_compact')
    return SparseRowGrad(out_rows, out_vals, fields=0), counter


class _EmbeddingLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table, row_offset, vocab, holder, dense_grad, oob):
        require_cuda(idx, table)
        idx = idx.contiguous()
        B, F = idx.shape
        D = table.shape[1]
        out = torch.empty((B, F, D), dtype=torch.float32, device=table.device)
        rows = torch.empty((B, F), dtype=torch.int64, device=table.device)
        check(lib().dt_embedding_fwd(ptr(idx), _idx_kind(idx), ptr(table), ptr(row_offset), ptr(vocab),
                                     B, F, D, ptr(out), ptr(rows), ptr(oob), stream_ptr()),
              'dt_embedding_fwd')
        ctx.holder = holder
        ctx.dense_grad = dense_grad
        ctx.table_shape = table.shape
        ctx.save_for_backward(rows)
        ctx.mark_non_differentiable(rows)
        return out, rows

    @staticmethod
    def backward(ctx, g_out, _g_rows):
        (rows,) = ctx.saved_tensors
        g_out = _f32c(g_out)
        D = ctx.table_shape[1]
        g_table = None
        if ctx.dense_grad:
            g_table = torch.zeros(ctx.table_shape, dtype=torch.float32, device=g_out.device)
            check(lib().dt_embedding_bwd_dense(ptr(rows), ptr(g_out), rows.numel(), D, ptr(g_table),
                                               stream_ptr()), 'dt_embedding_bwd_dense')
        elif ctx.holder is not None:
            ctx.holder.add_sparse_grad(SparseRowGrad(rows.reshape(-1), g_out.reshape(-1, D)))
        return None, g_table, None, None, None, None, None


def embedding_lookup(idx, table, row_offset, vocab, holder=None, dense_grad=False, oob=None):
    """-> (emb [B,F,D], rows [B,F] int64).  With dense_grad=False the table gradient is handed to
    `holder.add_sparse_grad(SparseRowGrad)` instead of being densified."""
    # a dummy requires-grad path is needed so backward runs even though `table.grad` stays None
    return _EmbeddingLookup.apply(idx, table, row_offset, vocab, holder, dense_grad, oob)


# ------------------------------------------------------------------------------------------------
# FM.call — deeptables/models/layers.py:53-62
# ------------------------------------------------------------------------------------------------
class _FM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x)
        x = _f32c(x)
        B, F, D = x.shape
        out = torch.empty((B,), dtype=torch.float32, device=x.device)
        check(lib().dt_fm_fwd(ptr(x), B, F, D, ptr(out), stream_ptr()), 'dt_fm_fwd')
        ctx.save_for_backward(x)
        return out.view(B, 1)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, F, D = x.shape
        g = _f32c(g).view(B)
        gx = torch.empty_like(x)
        check(lib().dt_fm_bwd(ptr(x), ptr(g), B, F, D, ptr(gx), stream_ptr()), 'dt_fm_bwd')
        return gx


def fm(x):
    return _FM.apply(x)


# ------------------------------------------------------------------------------------------------
# fused: embedding gather + [flatten(emb), dense] concat + linear field-sum + FM
# layers.py:889-904 + deepmodel.py:269-274,348-353 + deepnets.py:49-51 + layers.py:53-62
# ------------------------------------------------------------------------------------------------
class _EmbedFmLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table, row_offset, vocab, dense, holder, dense_grad, oob, want_emb):
        require_cuda(idx, table)
        idx = idx.contiguous()
        B, F = idx.shape
        D = table.shape[1]
        Nd = 0 if dense is None else dense.shape[1]
        dev = table.device
        dense_c = None if dense is None else _f32c(dense)
        emb = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        concat = torch.empty((B, F * D + Nd), dtype=torch.float32, device=dev)
        fsum = torch.empty((B, F), dtype=torch.float32, device=dev)
        fmo = torch.empty((B,), dtype=torch.float32, device=dev)
        rows = torch.empty((B, F), dtype=torch.int64, device=dev)
        check(lib().dt_embed_fm_linear_fwd(ptr(idx), _idx_kind(idx), ptr(table), ptr(row_offset),
                                           ptr(vocab), ptr(dense_c), B, F, D, Nd, ptr(emb), ptr(concat),
                                           ptr(fsum), ptr(fmo), ptr(rows), ptr(oob), stream_ptr()),
              'dt_embed_fm_linear_fwd')
        ctx.holder, ctx.dense_grad, ctx.table_shape, ctx.Nd = holder, dense_grad, table.shape, Nd
        ctx.save_for_backward(emb, rows)
        ctx.mark_non_differentiable(rows)
        return emb, concat, fsum, fmo.view(B, 1), rows

    @staticmethod
    def backward(ctx, g_emb, g_concat, g_fsum, g_fm, _g_rows):
        emb, rows = ctx.saved_tensors
        B, F, D = emb.shape
        g_emb = None if g_emb is None else _f32c(g_emb)
        g_concat = None if g_concat is None else _f32c(g_concat)
        g_fsum = None if g_fsum is None else _f32c(g_fsum)
        g_fm = None if g_fm is None else _f32c(g_fm).view(B)
        grad_rows = torch.empty_like(emb)
        check(lib().dt_embed_fm_linear_bwd(ptr(emb), ptr(g_emb), ptr(g_concat), F * D + ctx.Nd,
                                           ptr(g_fsum), ptr(g_fm), B, F, D, ptr(grad_rows),
                                           stream_ptr()), 'dt_embed_fm_linear_bwd')
        g_table = None
        if ctx.dense_grad:
            g_table = torch.zeros(ctx.table_shape, dtype=torch.float32, device=emb.device)
            check(lib().dt_embedding_bwd_dense(ptr(rows), ptr(grad_rows), rows.numel(), D,
                                               ptr(g_table), stream_ptr()), 'dt_embedding_bwd_dense')
        elif ctx.holder is not None:
            ctx.holder.add_sparse_grad(SparseRowGrad(rows.reshape(-1), grad_rows.reshape(-1, D)))
        g_dense = None
        if ctx.Nd > 0 and g_concat is not None and ctx.needs_input_grad[4]:
            g_dense = g_concat[:, F * D:].contiguous()
        return None, g_table, None, None, g_dense, None, None, None, None


def embed_fm_linear(idx, table, row_offset, vocab, dense=None, holder=None, dense_grad=False, oob=None):
    """-> emb [B,F,D], concat [B,F*D+Nd], field_sum [B,F], fm [B,1], rows [B,F]"""
    return _EmbedFmLinear.apply(idx, table, row_offset, vocab, dense, holder, dense_grad, oob, True)


# ------------------------------------------------------------------------------------------------
# Keras BatchNormalization (deepmodel.py:359; layers.py:152)
# ------------------------------------------------------------------------------------------------
def _bn_ws(N, C, device):
    nbytes = lib().dt_bn_workspace_bytes(N, C)
    return torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=device)


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum):
        require_cuda(x)
        x2 = _f32c(x).view(-1, x.shape[-1])
        N, C = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = _bn_ws(N, C, x.device)
        check(lib().dt_bn_train_fwd(ptr(x2), N, C, ptr(gamma), ptr(beta), eps, momentum,
                                    ptr(moving_mean), ptr(moving_var), ptr(y), ptr(mean), ptr(rstd),
                                    ptr(ws), stream_ptr()), 'dt_bn_train_fwd')
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.has_affine = (gamma is not None, beta is not None)
        ctx.x_shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        N, C = x2.shape
        gy2 = _f32c(gy).view(N, C)
        gx = torch.empty_like(x2)
        ggamma = torch.empty((C,), dtype=torch.float32, device=x2.device)
        gbeta = torch.empty((C,), dtype=torch.float32, device=x2.device)
        ws = _bn_ws(N, C, x2.device)
        check(lib().dt_bn_train_bwd(ptr(x2), ptr(gy2), N, C, ptr(gamma), ptr(mean), ptr(rstd), ptr(gx),
                                    ptr(ggamma), ptr(gbeta), ptr(ws), stream_ptr()), 'dt_bn_train_bwd')
        return (gx.view(ctx.x_shape), ggamma if ctx.has_affine[0] else None,
                gbeta if ctx.has_affine[1] else None, None, None, None, None)


def batchnorm_train(x, gamma, beta, moving_mean, moving_var, eps=1e-3, momentum=0.99):
    return _BatchNormTrain.apply(x, gamma, beta, moving_mean, moving_var, float(eps), float(momentum))


def batchnorm_infer(x, gamma, beta, moving_mean, moving_var, eps=1e-3):
    require_cuda(x)
    x2 = _f32c(x).view(-1, x.shape[-1])
    N, C = x2.shape
    y = torch.empty_like(x2)
    check(lib().dt_bn_infer_fwd(ptr(x2), N, C, ptr(gamma), ptr(beta), float(eps), ptr(moving_mean),
                                ptr(moving_var), ptr(y), stream_ptr()), 'dt_bn_infer_fwd')
    return y.view(x.shape)


# ------------------------------------------------------------------------------------------------
# Cross.call — deeptables/models/layers.py:428-436
# ------------------------------------------------------------------------------------------------
class _Cross(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        require_cuda(x, w, b)
        x, w, b = _f32c(x), _f32c(w), _f32c(b)
        B, C = x.shape
        L = w.shape[0]
        out = torch.empty_like(x)
        save_s = torch.empty((B, max(L, 1)), dtype=torch.float32, device=x.device)
        check(lib().dt_cross_fwd(ptr(x), ptr(w), ptr(b), B, C, L, ptr(out), ptr(save_s), stream_ptr()),
              'dt_cross_fwd')
        ctx.save_for_backward(x, w, b, save_s)
        ctx.w_ref, ctx.b_ref = w, b
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, b, save_s = ctx.saved_tensors
        B, C = x.shape
        L = w.shape[0]
        g = _f32c(g)
        gx = torch.empty_like(x)
        gw, gw_ret = _grad_target(ctx.w_ref)
        gb, gb_ret = _grad_target(ctx.b_ref)
        nbytes = lib().dt_cross_workspace_bytes(B, C, L)
        ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)
        check(lib().dt_cross_bwd(ptr(x), ptr(w), ptr(b), ptr(save_s), ptr(g), B, C, L, ptr(gx), ptr(gw),
                                 ptr(gb), ptr(ws), stream_ptr()), 'dt_cross_bwd')
        return gx, gw_ret, gb_ret


def cross(x, w, b):
    """x [B,C]; w,b [L,C] (Keras kernels_l/bias_l of shape (C,1), stacked and squeezed)."""
    return _Cross.apply(x, w, b)


# ------------------------------------------------------------------------------------------------
# InnerProduct.call / OuterProduct.call — deeptables/models/layers.py:473-487, 543-581
# ------------------------------------------------------------------------------------------------
class _InnerProduct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x)
        x = _f32c(x)
        B, F, D = x.shape
        out = torch.empty((B, F * (F - 1) // 2), dtype=torch.float32, device=x.device)
        check(lib().dt_inner_product_fwd(ptr(x), B, F, D, ptr(out), stream_ptr()), 'dt_inner_product_fwd')
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, F, D = x.shape
        gx = torch.empty_like(x)
        check(lib().dt_inner_product_bwd(ptr(x), ptr(_f32c(g)), B, F, D, ptr(gx), stream_ptr()),
              'dt_inner_product_bwd')
        return gx


def inner_product(x):
    return _InnerProduct.apply(x)


class _OuterProduct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, kernel_type):
        require_cuda(x, kernel)
        x, kernel = _f32c(x), _f32c(kernel)
        B, F, D = x.shape
        out = torch.empty((B, F * (F - 1) // 2), dtype=torch.float32, device=x.device)
        check(lib().dt_outer_product_fwd(ptr(x), ptr(kernel), kernel_type, B, F, D, ptr(out),
                                         stream_ptr()), 'dt_outer_product_fwd')
        ctx.save_for_backward(x, kernel)
        ctx.kernel_type = kernel_type
        ctx.k_ref = kernel
        return out

    @staticmethod
    def backward(ctx, g):
        x, kernel = ctx.saved_tensors
        B, F, D = x.shape
        gx = torch.empty_like(x)
        gk, gk_ret = _grad_target(ctx.k_ref)
        check(lib().dt_outer_product_bwd(ptr(x), ptr(kernel), ctx.kernel_type, ptr(_f32c(g)), B, F, D,
                                         ptr(gx), ptr(gk), stream_ptr()), 'dt_outer_product_bwd')
        return gx, gk_ret, None


def outer_product(x, kernel, kernel_type='mat'):
    return _OuterProduct.apply(x, kernel, _lib.DT_OP_KERNEL[kernel_type])


# ------------------------------------------------------------------------------------------------
# one CIN layer — deeptables/models/layers.py:689-710
# ------------------------------------------------------------------------------------------------
class _CinLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, xk, W, bias, act, bf16):
        # bf16: False = exact fp32 MFMA, True / 'bf16' = plain bf16 operands (1e-2 mode), 'bf16x3' = split-bf16 (fp32 bars)
        require_cuda(x0, xk, W)
        x0, W = _f32c(x0), _f32c(W)
        if xk.dtype != torch.float32:
            xk = xk.float()
        # xk may be a channel-slice view [:, :H] of the previous layer's output (direct=False)
        if not (xk.stride(2) == 1 and xk.stride(1) == xk.shape[2]):
            xk = xk.contiguous()
        B, F0, D = x0.shape
        Hk = xk.shape[1]
        L = W.shape[1]
        y = torch.empty((B, L, D), dtype=torch.float32, device=x0.device)
        bias_c = None if bias is None else _f32c(bias)
        ctx.bf16 = bool(bf16)
        ctx.x3 = bf16 == 'bf16x3'
        if ctx.x3:       # split-bf16 mode (csrc/cin_bf16.hip, NP parts): the exact kernels' parity bars on the bf16 matrix cores
            ws = torch.empty((lib().dt_cin_bf16x3_workspace_bytes(F0, Hk, L) + 3) // 4, dtype=torch.float32, device=x0.device)
            check(lib().dt_cin_layer_fwd_bf16x3(ptr(x0), ptr(xk), ptr(W), ptr(bias_c), act, B, F0, Hk, L, D,
                                                F0 * D, xk.stride(0), ptr(y), ptr(ws), stream_ptr()), 'dt_cin_layer_fwd_bf16x3')
        elif ctx.bf16:   # opt-in bf16-MFMA mode (csrc/cin_bf16.hip): 1e-2 instead of 1e-4 against the float64 oracle
            ws = torch.empty((lib().dt_cin_bf16_workspace_bytes(F0, Hk, L) + 3) // 4, dtype=torch.float32, device=x0.device)
            check(lib().dt_cin_layer_fwd_bf16(ptr(x0), ptr(xk), ptr(W), ptr(bias_c), act, B, F0, Hk, L, D,
                                              F0 * D, xk.stride(0), ptr(y), ptr(ws), stream_ptr()), 'dt_cin_layer_fwd_bf16')
        else:
            check(lib().dt_cin_layer_fwd(ptr(x0), ptr(xk), ptr(W), ptr(bias_c), act, B, F0, Hk, L, D,
                                         F0 * D, xk.stride(0), ptr(y), stream_ptr()), 'dt_cin_layer_fwd')
        ctx.save_for_backward(x0, xk, W, y)
        ctx.act = act
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x0, xk, W, y = ctx.saved_tensors
        B, F0, D = x0.shape
        Hk = xk.shape[1]
        L = W.shape[1]
        gy = _f32c(gy)
        nws = 0 if ctx.bf16 or os.environ.get('DT_AMD_CIN_WGRAD_ATOMIC') == '1' else \
            lib().dt_cin_bwd_workspace_bytes(B, F0, Hk, L, D)
        alloc = torch.empty if nws > 0 else torch.zeros          # dt_cin_layer_bwd_ws overwrites the three gradients
        alloc_x = torch.empty if (nws > 0 or ctx.bf16) else torch.zeros     # ... and the bf16 backward grad_x0 / grad_xk
        gx0 = alloc_x(x0.shape, dtype=torch.float32, device=x0.device)
        gxk = alloc_x((B, Hk, D), dtype=torch.float32, device=x0.device)
        gW = alloc(W.shape, dtype=torch.float32, device=x0.device)
        gb = torch.zeros((L,), dtype=torch.float32, device=x0.device) if ctx.has_bias else None
        if ctx.x3:
            ws = torch.empty((lib().dt_cin_bf16x3_workspace_bytes(F0, Hk, L) + 3) // 4, dtype=torch.float32, device=x0.device)
            check(lib().dt_cin_layer_bwd_bf16x3(ptr(x0), ptr(xk), ptr(W), ptr(y), ptr(gy), ctx.act, B, F0, Hk, L, D,
                                                F0 * D, xk.stride(0), ptr(gx0), ptr(gxk), ptr(gW), ptr(gb), ptr(ws),
                                                stream_ptr()), 'dt_cin_layer_bwd_bf16x3')
        elif ctx.bf16:
            ws = torch.empty((lib().dt_cin_bf16_workspace_bytes(F0, Hk, L) + 3) // 4, dtype=torch.float32, device=x0.device)
            check(lib().dt_cin_layer_bwd_bf16(ptr(x0), ptr(xk), ptr(W), ptr(y), ptr(gy), ctx.act, B, F0, Hk, L, D,
                                              F0 * D, xk.stride(0), ptr(gx0), ptr(gxk), ptr(gW), ptr(gb), ptr(ws),
                                              stream_ptr()), 'dt_cin_layer_bwd_bf16')
        else:
            # the weight-gradient kernel's batch splits store partial [K][L] slabs in a workspace (no float atomics)
            if nws <= 0:
                check(lib().dt_cin_layer_bwd(ptr(x0), ptr(xk), ptr(W), ptr(y), ptr(gy), ctx.act, B, F0, Hk, L, D,
                                             F0 * D, xk.stride(0), ptr(gx0), ptr(gxk), ptr(gW), ptr(gb),
                                             stream_ptr()), 'dt_cin_layer_bwd')
            else:
                ws = torch.empty((nws + 3) // 4, dtype=torch.float32, device=x0.device)
                check(lib().dt_cin_layer_bwd_ws(ptr(x0), ptr(xk), ptr(W), ptr(y), ptr(gy), ctx.act, B, F0, Hk, L, D,
                                                F0 * D, xk.stride(0), ptr(gx0), ptr(gxk), ptr(gW), ptr(gb), ptr(ws),
                                                stream_ptr()), 'dt_cin_layer_bwd_ws')
        return gx0, gxk, gW, gb, None, None


class _CinSplitPool(torch.autograd.Function):
    """CIN `direct=False` bookkeeping of one layer output y [B,L,D] (layers.py:713-718, :720-721): the first `half` channels
    feed the next layer (a view of y), the others go to the result — of which only the sum over D is ever used
    (`tf.reduce_sum(result, -1)`), so they are pooled here.  Backward: the layer's incoming gradient [B,L,D] is assembled
    in ONE pass from the two pieces (hidden half copied, pooled half broadcast over D) — torch's slice / cat / sum backward
    built two zero-filled full-size tensors, added them and expanded the pooled gradient (6 launches per layer).
    One HIP launch each way (dt_cin_pool / dt_cin_pool_bwd) for contiguous float32 y with D % 4 == 0; other device layouts
    take the same steps as torch device ops.  CUDA(HIP) tensors only, like every function of this module."""

    @staticmethod
    def forward(ctx, y, half):
        require_cuda(y)
        ctx.shape, ctx.half = tuple(y.shape), int(half)
        hidden = y[:, :half]
        B, L, D = y.shape
        ctx.hip = bool(y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and D % 4 == 0 and B > 0)
        if ctx.hip:
            pooled = torch.empty((B, L - half), dtype=torch.float32, device=y.device)
            check(lib().dt_cin_pool(ptr(y), B, L, D, int(half), ptr(pooled), stream_ptr()), 'dt_cin_pool')
        else:
            pooled = y[:, half:].sum(-1)
        return hidden, pooled

    @staticmethod
    def backward(ctx, g_hidden, g_pooled):
        B, L, D = ctx.shape
        half = ctx.half
        gy = torch.empty(ctx.shape, dtype=torch.float32, device=(g_pooled if g_pooled is not None else g_hidden).device)
        if ctx.hip:
            gh = None if g_hidden is None else _f32c(g_hidden)
            gp = None if g_pooled is None else _f32c(g_pooled)
            check(lib().dt_cin_pool_bwd(ptr(gh), ptr(gp), B, L, D, half, ptr(gy), stream_ptr()), 'dt_cin_pool_bwd')
            return gy, None
        if g_hidden is None:
            gy[:, :half].zero_()
        else:
            gy[:, :half].copy_(g_hidden)
        if g_pooled is None:
            gy[:, half:].zero_()
        else:
            gy[:, half:].copy_(g_pooled.unsqueeze(-1).expand(B, L - half, D))
        return gy, None


def cin_split_pool(y, half):
    """y [B,L,D] -> (y[:, :half] (view), sum_D y[:, half:] [B, L-half])"""
    return _CinSplitPool.apply(y, int(half))


def cin_layer(x0, xk, W, bias=None, activation='relu', mfma_dtype='float32'):
    """x0 [B,F0,D], xk [B,Hk,D], W [F0*Hk, L] -> y [B,L,D] = act(conv1d(outer(x0,xk), W) + bias).
    mfma_dtype: 'float32' (exact fp32 MFMA), 'bf16x3' (split-bf16 operands on the bf16 matrix cores: three parts and six
    products in the forward, two parts and three products in the backward, fp32 accumulate — the exact kernels' bars) or 'bf16'
    (plain bf16 operands, fp32 accumulation: ~1e-2)."""
    act = _lib.act_code(activation, 'CIN')
    if mfma_dtype not in ('float32', 'fp32', 'f32', 'bf16', 'bfloat16', 'bf16x3'):
        raise ValueError(f'CIN mfma_dtype {mfma_dtype!r}: expected float32, bf16x3 or bf16')
    return _CinLayer.apply(x0, xk, W, bias, act, 'bf16x3' if mfma_dtype == 'bf16x3' else mfma_dtype in ('bf16', 'bfloat16'))


# ------------------------------------------------------------------------------------------------
# MultiheadAttention core — deeptables/models/layers.py:129-145
# ------------------------------------------------------------------------------------------------
def _row_stride(t, D):
    """Row stride (floats) if t [B,F,D] is float32 with unit inner stride and rows `ld` apart (a contiguous tensor or
    a column block of a wider [B,F,ld] buffer), else None."""
    if t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1:
        return None
    ld = t.stride(1)
    if ld < D or t.stride(0) != t.shape[1] * ld or (t.data_ptr() % 16) != 0:
        return None
    return ld


class _MhaCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, H, grad_cols, rate, seed):
        require_cuda(q, k, v)
        B, F, D = q.shape
        ctx.grad_cols = max(int(grad_cols), 3)
        lds = {_row_stride(t, D) for t in (q, k, v)}
        if len(lds) != 1 or None in lds:             # mixed layouts: make them contiguous
            q, k, v = _f32c(q), _f32c(k), _f32c(v)
            ld = D
        else:
            ld = lds.pop()
        out = torch.empty((B, F, D), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, H, F), dtype=torch.float32, device=q.device)
        check(lib().dt_mha_core_fwd(ptr(q), ptr(k), ptr(v), B, F, D, H, ld, float(rate), int(seed) & 0xFFFFFFFF,
                                    ptr(out), ptr(lse), stream_ptr()), 'dt_mha_core_fwd')
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.H, ctx.ld, ctx.drop = H, ld, (float(rate), int(seed) & 0xFFFFFFFF)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        B, F, D = out.shape
        g = _f32c(g)
        # gq | gk | gv side by side in a buffer `grad_cols` blocks wide: when q/k/v are column blocks of a fused
        # projection output, split_cols' backward completes this buffer in place instead of concatenating
        W = ctx.grad_cols * D
        G = torch.empty((B, F, W), dtype=torch.float32, device=out.device)
        gq, gk, gv = G[..., :D], G[..., D:2 * D], G[..., 2 * D:3 * D]
        check(lib().dt_mha_core_bwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(g), B, F, D, ctx.H, ctx.ld,
                                    W, ctx.drop[0], ctx.drop[1], ptr(gq), ptr(gk), ptr(gv), stream_ptr()),
              'dt_mha_core_bwd')
        return gq, gk, gv, None, None, None, None


def mha_core(q, k, v, num_heads, grad_cols=3, dropout_rate=0.0, seed=0):
    """grad_cols: width (in blocks of D) of the buffer the q/k/v gradients are laid out in (see split_cols).
    dropout_rate / seed: Dropout on the attention weights (layers.py:141), mask = autoint_dropout_keep(seed, ...)."""
    return _MhaCore.apply(q, k, v, int(num_heads), int(grad_cols), float(dropout_rate), int(seed))


# ------------------------------------------------------------------------------------------------
# AutoInt interacting layer without its BatchNormalization — MultiheadAttention.call layers.py:119-150, one forward and
# one backward launch on the fp32 matrix cores (csrc/autoint.hip); Q/K/V/residual never reach HBM
# ------------------------------------------------------------------------------------------------
def autoint_dropout_keep(seed, B, H, F, rate, device='cpu'):
    """[B,H,F,F] keep-scale of the attention-weight dropout (0 or 1/(1-rate)) for `seed`: the counter hash of
    csrc/autoint.hip restated with torch integer ops (tests rebuild the kernel's mask from it)."""
    M32 = 0xFFFFFFFF
    b = torch.arange(B, dtype=torch.int64, device=device).view(B, 1, 1, 1)
    h = torch.arange(H, dtype=torch.int64, device=device).view(1, H, 1, 1)
    i = torch.arange(F, dtype=torch.int64, device=device).view(1, 1, F, 1)
    j = torch.arange(F, dtype=torch.int64, device=device).view(1, 1, 1, F)
    x = (int(seed) & M32) ^ ((b * 0x9E3779B1) & M32) ^ (((h * 4096 + i * 64 + j) * 0x85EBCA77) & M32)
    x = x ^ (x >> 16); x = (x * 0x7FEB352D) & M32; x = x ^ (x >> 15); x = (x * 0x846CA68B) & M32; x = x ^ (x >> 16)
    thr = min(max(int(float(rate) * 4294967296.0), 1), 4294967295)
    return (x >= thr).to(torch.float32) / (1.0 - float(rate))


_AUTOINT_WS = {}
_AUTOINT_WS_RETIRED = []


def _autoint_ws(B, D, device, nbytes=None, tag='w'):
    """per-block partials of dt_autoint_bwd_w / dt_autoint_fwd_bn: ONE grow-only buffer per (kind, device, STREAM) — a
    partial buffer is written and consumed by two launches of the same call, so calls on one stream may share it; calls
    on different streams must not.  Keyed without the size: a ragged last batch or a validation batch size reuses the
    buffer (reallocated only when a LARGER one is needed), instead of pinning a new one per distinct B."""
    n = int(lib().dt_autoint_bwd_workspace_bytes(int(B), int(D))) if nbytes is None else int(nbytes)
    key = (tag, str(device), int(torch.cuda.current_stream(device).cuda_stream))
    buf = _AUTOINT_WS.get(key)
    if buf is None or buf.numel() * 4 < n:
        if buf is not None:
            # a captured hipGraph may hold the old buffer's address: it stays alive (growth is geometric, so only a few
            # are ever retired)
            _AUTOINT_WS_RETIRED.append(buf)
            n = max(n, int(buf.numel() * 4 * 1.5))
        buf = torch.empty((n + 3) // 4, dtype=torch.float32, device=device)
        _AUTOINT_WS[key] = buf
    return buf


class AutoIntBnLink:
    """What two stacked interacting layers share (deepnets.py:219-221: `output = MultiheadAttention(...)(output)`): the layer
    below leaves its pre-normalisation output, the batch statistics and a zeroed [2 D] double buffer here; the layer above, whose
    backward produces dX = the gradient w.r.t. the lower layer's BatchNormalization output, adds that normalisation's two
    backward sums into the buffer while dX leaves (csrc/autoint.hip AiPrev) and records dX's address.  The lower layer's
    backward then skips its own statistics pass — if the gradient it receives IS that dX (a second consumer of its output
    would make autograd hand it a sum — in another tensor, or added IN PLACE into dX, which moves dX's version counter: either
    way the pass runs as before)."""

    __slots__ = ('a', 'mean', 'rstd', 'sums', 'dx_ptr', 'dx_ver', 'lazy', 'gamma', 'beta', 'rank1', 'dirty')

    def __init__(self):
        self.a = self.mean = self.rstd = self.sums = None
        self.dx_ptr = self.dx_ver = None
        # lazy (autoint_layer(defer_bn=True)): the tensor handed to the layer above IS `a` — the normalisation is pending and
        # the consumer applies it while it loads its input (csrc/autoint.hip AiXn) with gamma / beta [D] (None = 1 / 0)
        self.lazy = False
        self.gamma = self.beta = None
        # rank1 = (gz [B], w [F D]) left by autoint_head's backward: the gradient w.r.t. the normalised output is gz w — the
        # tensor autograd hands over (address dx_ptr) is then an unwritten placeholder and must never be read
        self.rank1 = None
        # the sums are ADDED into by the consumer's backward and start at zero with every forward (the forward's second launch
        # zeroes them): a second backward over the same forward (retain_graph) finds them dirty and zeroes them itself
        self.dirty = False

    def sums_for_adding(self):
        if self.dirty:
            self.sums.zero_()
        self.dirty = True
        return ptr(self.sums)

    def xn_ptrs(self):
        return (ptr(self.mean), ptr(self.rstd), ptr(self.gamma), ptr(self.beta)) if self.lazy else (None,) * 4


class _AutoIntLayer(torch.autograd.Function):
    """forward(x, num_heads, dropout_rate, seed, bn, *wb): wb = Wq, Wk, Wv[, Wr], bq, bk, bv[, br] (the Keras variables,
    never concatenated).  bn = None -> returns a = relu(attention + residual); bn = (gamma, beta, moving_mean,
    moving_var, eps, momentum) -> returns BatchNormalization(a) with batch statistics (layers.py:151), and the BN
    backward is applied inside the layer's backward kernel while it reads the incoming gradient."""

    @staticmethod
    def forward(ctx, x, num_heads, dropout_rate, seed, bn, gamma, beta, mode, links, *wb):
        require_cuda(x, *wb)
        mode = int(mode)
        ctx.link_in, ctx.link_out, defer = links if links is not None else (None, None, False)
        x = _f32c(x)
        # the layer below handed over its UN-normalised output (AutoIntBnLink.lazy): normalised while this layer loads it
        xn = ctx.link_in.xn_ptrs() if ctx.link_in is not None else (None,) * 4
        wb = [_f32c(t) for t in wb]
        NP = len(wb) // 2
        Ws, bs = wb[:NP] + [None] * (4 - NP), wb[NP:] + [None] * (4 - NP)
        B, F, D = x.shape
        a = torch.empty_like(x)
        if bn is not None and B > 0 and os.environ.get('DT_AMD_AUTOINT_BN', 'fused') != 'separate':
            # attention + BatchNormalization in two launches: the statistics ride in the attention kernel's epilogue
            moving_mean, moving_var, eps, momentum = bn
            lk = ctx.link_out
            defer = bool(defer) and lk is not None and F <= 28 and os.environ.get('DT_AMD_AUTOINT_WGRAD', 'fused') != 'dense'
            y = None if defer else torch.empty_like(a)
            mean = torch.empty((D,), dtype=torch.float32, device=x.device)
            rstd = torch.empty((D,), dtype=torch.float32, device=x.device)
            n = int(lib().dt_autoint_fwd_bn_workspace_bytes(B, D))
            ws = _autoint_ws(0, D, x.device, nbytes=n, tag='bn')
            # (the link's double accumulators are zeroed by the call's second launch, not by a fill of their own)
            sums = torch.empty(2 * D, dtype=torch.float64, device=x.device) if lk is not None else None
            check(lib().dt_autoint_fwd_bn(ptr(x), *[ptr(t) for t in Ws], *[ptr(t) for t in bs], B, F, D, num_heads,
                                          float(dropout_rate), int(seed) & 0xFFFFFFFF, ptr(gamma), ptr(beta), float(eps),
                                          float(momentum), ptr(moving_mean), ptr(moving_var), ptr(a), ptr(y), ptr(mean),
                                          ptr(rstd), ptr(ws), *xn, ptr(sums), mode, stream_ptr()), 'dt_autoint_fwd_bn')
            ctx.cfg = (num_heads, NP, float(dropout_rate), int(seed) & 0xFFFFFFFF, True, mode)
            ctx.save_for_backward(x, a, *wb, mean, rstd, *([gamma] if gamma is not None else []))
            ctx.has_affine = (gamma is not None, beta is not None)
            if lk is not None:
                lk.a, lk.mean, lk.rstd = a, mean, rstd
                lk.sums = sums
                lk.dx_ptr = lk.rank1 = None
                lk.dirty = False
                lk.lazy, lk.gamma, lk.beta = defer, gamma, beta
            if defer:
                # the output IS a: its only consumer (the interacting layer above, by the caller's promise) normalises on load;
                # the gradient that comes back is the one w.r.t. the NORMALISED tensor, as before
                return a
            return y
        check(lib().dt_autoint_fwd(ptr(x), *[ptr(t) for t in Ws], *[ptr(t) for t in bs], B, F, D, num_heads,
                                   float(dropout_rate), int(seed) & 0xFFFFFFFF, ptr(a), None, *xn, mode, stream_ptr()),
              'dt_autoint_fwd')
        ctx.cfg = (num_heads, NP, float(dropout_rate), int(seed) & 0xFFFFFFFF, bn is not None, mode)
        if bn is None:
            ctx.save_for_backward(x, a, *wb)
            return a
        moving_mean, moving_var, eps, momentum = bn
        N = B * F
        y = torch.empty_like(a)
        mean = torch.empty((D,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((D,), dtype=torch.float32, device=x.device)
        ws = _bn_ws(N, D, x.device)
        check(lib().dt_bn_train_fwd(ptr(a), N, D, ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(moving_mean),
                                    ptr(moving_var), ptr(y), ptr(mean), ptr(rstd), ptr(ws), stream_ptr()),
              'dt_bn_train_fwd')
        ctx.save_for_backward(x, a, *wb, mean, rstd, *([gamma] if gamma is not None else []))
        ctx.has_affine = (gamma is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        H, NP, rate, seed, has_bn, mode = ctx.cfg
        saved = list(ctx.saved_tensors)
        x, a = saved[0], saved[1]
        wb = saved[2:2 + 2 * NP]
        Ws, bs = list(wb[:NP]) + [None] * (4 - NP), list(wb[NP:]) + [None] * (4 - NP)
        B, F, D = x.shape
        g = _f32c(g)
        gamma = mean = rstd = sums = ggamma = gbeta = sums64 = g1w = None
        fused_w = F <= 28 and os.environ.get('DT_AMD_AUTOINT_WGRAD', 'fused') != 'dense'
        if has_bn:
            mean, rstd = saved[2 + 2 * NP], saved[3 + 2 * NP]
            gamma = saved[4 + 2 * NP] if ctx.has_affine[0] else None
            lk = ctx.link_out
            linked = lk is not None and lk.dx_ptr is not None and lk.dx_ptr == g.data_ptr() and lk.dx_ver == g._version and \
                lk.sums is not None
            if lk is not None and lk.rank1 is not None:
                # autoint_head's backward: g is an unwritten placeholder, the gradient is gz[b] w[i D + c] (formed in the kernel)
                if not (linked and fused_w):
                    raise _lib.DtHipError('autoint_layer: the rank-one gradient of autoint_head did not reach its layer unchanged')
                g, g1w = lk.rank1
                lk.rank1 = None
            if linked:
                # the layer above formed this normalisation's backward sums while it wrote g (AutoIntBnLink): no pass over a, g
                if fused_w:
                    # ... and the kernel reads the doubles as they are; the gradients of beta | gamma (the same two sums as
                    # floats) come out of the call's reduction launch
                    sums64 = lk.sums
                    bn_grads = torch.empty((2 * D,), dtype=torch.float32, device=x.device)
                    gbeta, ggamma = bn_grads[:D], bn_grads[D:]
                else:
                    sums = lk.sums.to(torch.float32)
                    gbeta, ggamma = sums[:D], sums[D:]
                lk.dx_ptr = None
            else:
                sums = torch.empty((2 * D,), dtype=torch.float32, device=x.device)
                ggamma = torch.empty((D,), dtype=torch.float32, device=x.device)
                gbeta = torch.empty((D,), dtype=torch.float32, device=x.device)
                ws = _bn_ws(B * F, D, x.device)
                check(lib().dt_bn_train_bwd_stats(ptr(a), ptr(g), B * F, D, ptr(mean), ptr(rstd), ptr(sums), ptr(ggamma),
                                                  ptr(gbeta), ptr(ws), stream_ptr()), 'dt_bn_train_bwd_stats')
        need_x = ctx.needs_input_grad[0]
        gx = torch.empty_like(x) if need_x else None
        M = NP * D
        # the layer below: its BatchNormalization-backward sums ride in this backward's epilogue (AutoIntBnLink)
        li = ctx.link_in
        prev = (None, None, None, None)
        if li is not None and need_x and li.a is not None and li.sums is not None and tuple(li.a.shape) == tuple(x.shape):
            prev = (ptr(li.a), ptr(li.mean), ptr(li.rstd), li.sums_for_adding())
            li.dx_ptr, li.dx_ver = gx.data_ptr(), gx._version
        xn = li.xn_ptrs() if li is not None else (None,) * 4      # x IS a_prev, normalised on load (forward did the same)
        if xn[0] is not None and not fused_w:
            raise _lib.DtHipError('autoint_layer: a deferred BatchNormalization input needs the fused weight-gradient backward (F <= 28)')
        if fused_w:
            # the kernel / bias gradients are accumulated inside the layer's backward launch (csrc/autoint.hip WG): the
            # pre-activation gradients dY [B*F, NP*D] never reach HBM and no Dense weight-gradient launch follows
            gWs = torch.empty((NP, D, D), dtype=torch.float32, device=x.device)
            gbs = torch.empty((NP, D), dtype=torch.float32, device=x.device)
            wsw = _autoint_ws(B, D, x.device)
            check(lib().dt_autoint_bwd_w(ptr(x), *[ptr(t) for t in Ws], *[ptr(t) for t in bs], ptr(a), ptr(g), B, F, D, H,
                                         rate, seed, ptr(gamma), ptr(mean), ptr(rstd), ptr(sums), ptr(gx), ptr(gWs),
                                         ptr(gbs), ptr(wsw), *prev, *xn, ptr(sums64),
                                         ptr(bn_grads) if sums64 is not None else None, ptr(g1w), mode, stream_ptr()),
                  'dt_autoint_bwd_w')
            return (gx, None, None, None, None, ggamma if has_bn and ctx.has_affine[0] else None,
                    gbeta if has_bn and ctx.has_affine[1] else None, None, None, *[gWs[i] for i in range(NP)],
                    *[gbs[i] for i in range(NP)])
        dY = torch.empty((B * F, M), dtype=torch.float32, device=x.device)
        check(lib().dt_autoint_bwd(ptr(x), *[ptr(t) for t in Ws], *[ptr(t) for t in bs], ptr(a), ptr(g), B, F, D, H,
                                   rate, seed, ptr(gamma), ptr(mean), ptr(rstd), ptr(sums), ptr(dY), ptr(gx),
                                   *prev, *xn, mode, stream_ptr()), 'dt_autoint_bwd')
        # kernel / bias gradients = x^T dY, colsum(dY): batch reductions on the Dense weight-gradient kernel (its W / y
        # arguments are unused for a linear layer without grad_x)
        buf = torch.zeros(D * M + M, dtype=torch.float32, device=x.device)
        gWc, gbc = buf[:D * M].view(D, M), buf[D * M:]
        check(lib().dt_dense_bwd(ptr(x), ptr(dY), ptr(dY), ptr(dY), _lib.DT_ACT_LINEAR, B * F, D, M, None, ptr(gWc),
                                 ptr(gbc), None, stream_ptr()), 'dt_dense_bwd')
        gWs = gWc.view(D, NP, D).permute(1, 0, 2).contiguous()           # one copy: [NP][D][D], per-variable views
        gW = [gWs[i] for i in range(NP)]
        gb = [gbc[i * D:(i + 1) * D] for i in range(NP)]
        return (gx, None, None, None, None, ggamma if has_bn and ctx.has_affine[0] else None,
                gbeta if has_bn and ctx.has_affine[1] else None, None, None, *gW, *gb)


def autoint_supported(x, num_heads):
    return x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and \
        bool(lib().dt_autoint_supported(int(x.shape[1]), int(x.shape[2]), int(num_heads)))


def autoint_layer(x, kernels, biases, num_heads, dropout_rate=0.0, seed=0, batch_norm=None, mfma_dtype=None, link=True,
                  defer_bn=False):
    """a = relu(multi-head field attention(relu-projections of x) [+ relu residual projection]) — layers.py:123-150.
    kernels / biases: those of dense_Q, dense_K, dense_V[, dense_residual] (3 or 4 of each; [D,D] and [D]).
    batch_norm = (gamma, beta, moving_mean, moving_var, eps, momentum): also applies the layer's training-mode
    BatchNormalization (layers.py:151) and returns BN(a).
    mfma_dtype (autoint_params['mfma_dtype']): None / 'float32' — exact fp32 MFMA; 'bf16' — north_star's 1e-2 mode: the layer's
    projection-shaped products on the bf16 matrix cores (include/dt_hip.h DT_AI_BF16; embedding size 32 only).
    defer_bn=True (with batch_norm): the caller's promise that the ONLY consumer of the result is another autoint_layer call —
    the returned tensor then holds the un-normalised output, marked (`_dt_bn_link.lazy`), and that call normalises it while
    it loads its input: the normalised tensor never exists (models/deepnets.py autoint_nets sets it for all but the last
    layer of the stack).  autoint_materialize(y) turns such a tensor into the normalised one for any other consumer."""
    assert len(kernels) == len(biases) and len(kernels) in (3, 4)
    mode = autoint_mfma_mode(mfma_dtype, int(x.shape[-1]))
    lazy_in = getattr(x, '_dt_bn_link', None)
    if lazy_in is not None and lazy_in.lazy and (not link or os.environ.get('DT_AMD_AUTOINT_LINK', '1') == '0'):
        x = autoint_materialize(x)
    if batch_norm is None:
        li = getattr(x, '_dt_bn_link', None)
        return _AutoIntLayer.apply(x, int(num_heads), float(dropout_rate), int(seed), None, None, None, mode,
                                   (li, None, False) if li is not None and li.lazy else None, *kernels, *biases)
    gamma, beta, mm, mv, eps, momentum = batch_norm
    # stacked layers: the input's link (left by the layer below, when x IS its normalised output) and this layer's own
    link_in = getattr(x, '_dt_bn_link', None) if link and os.environ.get('DT_AMD_AUTOINT_LINK', '1') != '0' else None
    link_out = AutoIntBnLink() if link else None
    y = _AutoIntLayer.apply(x, int(num_heads), float(dropout_rate), int(seed), (mm, mv, float(eps), float(momentum)),
                            gamma, beta, mode, (link_in, link_out, bool(defer_bn) and link_out is not None),
                            *kernels, *biases)
    if link_out is not None and link_out.a is not None:
        y._dt_bn_link = link_out
    return y


class _AutoIntMaterialize(torch.autograd.Function):
    """a tensor whose BatchNormalization is pending (autoint_layer(defer_bn=True)) -> the normalised tensor.  The gradient
    passes through unchanged: what flows back into the deferring layer is the gradient w.r.t. the NORMALISED tensor."""

    @staticmethod
    def forward(ctx, a, mean, rstd, gamma, beta):
        y = (a - mean) * rstd
        if gamma is not None:
            y = y * gamma
        if beta is not None:
            y = y + beta
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None, None


def autoint_materialize(y):
    lk = getattr(y, '_dt_bn_link', None)
    if lk is None or not lk.lazy:
        return y
    D = int(lk.mean.numel())
    if y.shape[-1] != D:                                          # flattened [B, F D]: the channel is the fastest index
        return _AutoIntMaterialize.apply(y.reshape(-1, D), lk.mean, lk.rstd, lk.gamma, lk.beta).reshape(y.shape)
    return _AutoIntMaterialize.apply(y, lk.mean, lk.rstd, lk.gamma, lk.beta)


class _AutoIntHead(torch.autograd.Function):
    """BatchNormalization (pending) -> Flatten -> Dense(1): the head of the AutoInt graph (deepnets.py:222-224, deepmodel.py:
    131-143) on the top interacting layer's UN-normalised output (csrc/autoint.hip k_autoint_head_*).  Backward: the Dense
    kernel / bias gradients and the normalisation's two backward sums (into the link's double buffer); the gradient w.r.t. the
    normalised tensor is rank one — gz[b] w[k] — and is NOT written: the link carries (gz, w) to the layer's backward kernel,
    autograd carries an unwritten placeholder whose address the layer checks."""

    @staticmethod
    def forward(ctx, x, kernel, bias, lk):
        require_cuda(x, kernel)
        B, K = x.shape
        D = int(lk.mean.numel())
        z = torch.empty((B, 1), dtype=torch.float32, device=x.device)
        check(lib().dt_autoint_head_fwd(ptr(x), ptr(kernel), ptr(bias), *lk.xn_ptrs(), B, K, D, ptr(z), stream_ptr()),
              'dt_autoint_head_fwd')
        ctx.save_for_backward(x, kernel)
        ctx.lk, ctx.has_bias = lk, bias is not None
        return z

    @staticmethod
    def backward(ctx, gz):
        x, kernel = ctx.saved_tensors
        lk = ctx.lk
        B, K = x.shape
        D = int(lk.mean.numel())
        gz = _f32c(gz).reshape(-1)
        gW = torch.empty_like(kernel)
        gb = torch.empty((1,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        ws = _autoint_ws(0, D, x.device, nbytes=int(lib().dt_autoint_head_workspace_bytes(B, K)), tag='head')
        check(lib().dt_autoint_head_bwd(ptr(x), ptr(kernel), ptr(gz), *lk.xn_ptrs(), B, K, D, ptr(gW), ptr(gb),
                                        lk.sums_for_adding(), ptr(ws), stream_ptr()), 'dt_autoint_head_bwd')
        gx = torch.empty_like(x)                                  # placeholder: never written, never read (lk.rank1)
        lk.dx_ptr, lk.dx_ver, lk.rank1 = gx.data_ptr(), gx._version, (gz, kernel)
        return gx, gW, gb, None


def autoint_head_supported(x, kernel, lk):
    K = int(x.shape[-1]) if x.dim() == 2 else 0
    return bool(x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and lk is not None and lk.lazy and
                lk.sums is not None and lk.a is not None and x.data_ptr() == lk.a.data_ptr() and
                tuple(kernel.shape) == (K, 1) and 4 <= K <= 1024 and K % 4 == 0 and K % int(lk.mean.numel()) == 0 and
                int(lk.a.shape[1]) <= 28)


def autoint_head(x, kernel, bias):
    """x: the FLATTENED output of autoint_layer(defer_bn=True) (its `_dt_bn_link` carried along); kernel [K, 1], bias [1] / None
    of the Dense(1) that consumes it -> z [B, 1]."""
    return _AutoIntHead.apply(x, kernel, bias, x._dt_bn_link)


def autoint_mfma_mode(mfma_dtype, D):
    """autoint_params['mfma_dtype'] -> the C-ABI's mfma_mode; an unsupported request raises (no silent fp32)"""
    if mfma_dtype is None:
        # the library's default, like the Dense tower's (fused.py) and CIN's: split-bf16 where the kernels have it (embedding
        # size 32) — the exact kernels' parity bars at 0.92 x their step time on the AutoInt graph (tools/r6/call6.sh);
        # DT_AMD_AUTOINT_DTYPE overrides the default (a caller's explicit autoint_params['mfma_dtype'] always wins)
        mfma_dtype = os.environ.get('DT_AMD_AUTOINT_DTYPE') or ('bf16x2' if D == 32 else 'float32')
    if mfma_dtype in ('float32', 'f32', 'fp32'):
        return _lib.DT_AI_F32
    if mfma_dtype in ('bf16', 'bfloat16', 'bf16x2'):
        if D != 32:
            raise _lib.DtHipError(f"autoint_params['mfma_dtype'] = {mfma_dtype!r} needs an embedding size of 32 (got {D})")
        return _lib.DT_AI_BF16X2 if mfma_dtype == 'bf16x2' else _lib.DT_AI_BF16
    raise ValueError(f"autoint_params['mfma_dtype'] = {mfma_dtype!r}: 'float32', 'bf16x2' (split-bf16, fp32 bars) or 'bf16'")


class _SplitCols(torch.autograd.Function):
    """y [.., n*D] -> n column blocks [.., D] (views).  Backward: if the first incoming gradient already is column
    block 0 of a buffer laid out like y (what mha_core(grad_cols=n) hands back for q), the other blocks are copied
    into that buffer where they are not there already and the buffer IS the gradient — the full [.., n*D]
    concatenation is skipped."""

    @staticmethod
    def forward(ctx, y, D):
        y = y.contiguous()
        ctx.D, ctx.shape = D, tuple(y.shape)
        return tuple(y[..., i * D:(i + 1) * D] for i in range(y.shape[-1] // D))

    @staticmethod
    def backward(ctx, *grads):
        D, shape = ctx.D, ctx.shape
        want = torch.empty(shape, device='meta').stride()          # strides of y == strides of a column block
        g0 = grads[0]
        numel = 1
        for d in shape:
            numel *= d
        if (tuple(g0.shape) == shape[:-1] + (D,) and g0.stride() == want and g0.dtype == torch.float32 and
                (g0.storage_offset() + numel) * 4 <= g0.untyped_storage().nbytes()):
            G = torch.as_strided(g0, shape, want)                   # the whole buffer g0 is block 0 of
            for i in range(1, len(grads)):
                blk = G[..., i * D:(i + 1) * D]
                g = grads[i]
                if g.data_ptr() == blk.data_ptr() and g.stride() == want:
                    continue                                        # already in place (gk, gv)
                blk.copy_(g)
            return G, None
        return torch.cat(list(grads), dim=-1), None


def split_cols(y, D):
    return _SplitCols.apply(y, int(D))


# ------------------------------------------------------------------------------------------------
# Keras Dense — deepnets.py:401-427, deepmodel.py:291-292,455, layers.py:104-108
# ------------------------------------------------------------------------------------------------
class _Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, bias, act):
        require_cuda(x, W)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        W = _f32c(W)
        N, K = x2.shape
        M = W.shape[1]
        y = torch.empty((N, M), dtype=torch.float32, device=x.device)
        bias_c = None if bias is None else _f32c(bias)
        check(lib().dt_dense_fwd(ptr(x2), ptr(W), ptr(bias_c), act, N, K, M, ptr(y), stream_ptr()), 'dt_dense_fwd')
        ctx.save_for_backward(x2, W, y)
        ctx.act, ctx.has_bias, ctx.x_shape = act, bias is not None, x.shape
        ctx.W_ref, ctx.b_ref = W, bias_c
        return y.reshape(*x.shape[:-1], M)

    @staticmethod
    def backward(ctx, gy):
        x2, W, y = ctx.saved_tensors
        N, K = x2.shape
        M = W.shape[1]
        gy2 = _f32c(gy).reshape(N, M)
        need_x = ctx.needs_input_grad[0]
        gx = torch.empty_like(x2) if need_x else None
        if getattr(ctx.W_ref, '_dt_grad_view', None) is None:
            # kernel and bias gradients share ONE zero-filled buffer (one fill launch instead of two)
            buf = torch.zeros(K * M + (M if ctx.has_bias else 0), dtype=torch.float32, device=W.device)
            gW = gW_ret = buf[:K * M].view(K, M)
            gb = gb_ret = buf[K * M:] if ctx.has_bias else None
        else:
            gW, gW_ret = _grad_target(ctx.W_ref)
            gb, gb_ret = _grad_target(ctx.b_ref) if ctx.has_bias else (None, None)
        nbytes = lib().dt_dense_workspace_bytes(N, K, M)
        ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=W.device)
        check(lib().dt_dense_bwd(ptr(x2), ptr(W), ptr(y), ptr(gy2), ctx.act, N, K, M, ptr(gx), ptr(gW), ptr(gb),
                                 ptr(ws), stream_ptr()), 'dt_dense_bwd')
        return (gx.reshape(ctx.x_shape) if need_x else None), gW_ret, gb_ret, None


def dense_supported(x, W):
    if not x.is_cuda:
        return False
    n = 1
    for d in x.shape[:-1]:
        n *= int(d)
    return n > 0 and bool(lib().dt_dense_supported(n, int(x.shape[-1]), int(W.shape[1])))


def dense(x, W, bias=None, activation=None):
    """y = act(x @ W + bias) with act in {None/'linear', 'relu'} on the HIP Dense kernels."""
    act = {'relu': _lib.DT_ACT_RELU, 'linear': _lib.DT_ACT_LINEAR, None: _lib.DT_ACT_LINEAR}[activation]
    return _Dense.apply(x, W, bias, act)


# ------------------------------------------------------------------------------------------------
# AFM attention pooling — AFM.call layers.py:789-807
# ------------------------------------------------------------------------------------------------
class _AfmPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Wa, ba, pv, act):
        require_cuda(x, Wa)
        x, Wa, pv = _f32c(x), _f32c(Wa), _f32c(pv).reshape(-1)
        ba_c = None if ba is None else _f32c(ba)
        B, F, D = x.shape
        H = Wa.shape[1]
        out = torch.empty((B, D), dtype=torch.float32, device=x.device)
        score = torch.empty((B, F * (F - 1) // 2), dtype=torch.float32, device=x.device)
        check(lib().dt_afm_fwd(ptr(x), ptr(Wa), ptr(ba_c), ptr(pv), act, B, F, D, H, ptr(out), ptr(score),
                               stream_ptr()), 'dt_afm_fwd')
        ctx.save_for_backward(x, Wa, pv, score, *([] if ba_c is None else [ba_c]))
        ctx.act, ctx.pv_shape = act, pv.shape
        return out

    @staticmethod
    def backward(ctx, g):
        x, Wa, pv, score, *rest = ctx.saved_tensors
        ba = rest[0] if rest else None
        B, F, D = x.shape
        H = Wa.shape[1]
        gx = torch.empty_like(x)
        gWa = torch.zeros_like(Wa)
        gba = None if ba is None else torch.zeros_like(ba)
        gpv = torch.zeros_like(pv)
        check(lib().dt_afm_bwd(ptr(x), ptr(Wa), ptr(ba), ptr(pv), ptr(score), ptr(_f32c(g)), ctx.act, B, F, D, H,
                               ptr(gx), ptr(gWa), ptr(gba), ptr(gpv), stream_ptr()), 'dt_afm_bwd')
        return gx, gWa, gba, gpv, None


def afm_pool(x, Wa, ba, pv, activation='relu'):
    """x [B,F,D] -> attention-pooled pair interactions [B,D]; Wa [D,H], ba [H]|None, pv [H] or [H,1]."""
    act = _lib.act_code(activation, 'AFM')
    return _AfmPool.apply(x, Wa, ba, pv.reshape(-1), act)


# ------------------------------------------------------------------------------------------------
# BilinearInteraction — layers.py:363-377
# ------------------------------------------------------------------------------------------------
BILINEAR_TYPES = {'field_interaction': 0, 'field_each': 1, 'field_all': 2}


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, wtype):
        require_cuda(x, W)
        x, W = _f32c(x), _f32c(W)
        B, F, D = x.shape
        out = torch.empty((B, F * (F - 1) // 2, D), dtype=torch.float32, device=x.device)
        check(lib().dt_bilinear_fwd(ptr(x), ptr(W), wtype, B, F, D, ptr(out), stream_ptr()), 'dt_bilinear_fwd')
        ctx.save_for_backward(x, W)
        ctx.wtype = wtype
        ctx.W_ref = W
        return out

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        B, F, D = x.shape
        gx = torch.empty_like(x)
        gW, gW_ret = _grad_target(ctx.W_ref)
        check(lib().dt_bilinear_bwd(ptr(x), ptr(W), ptr(_f32c(g)), ctx.wtype, B, F, D, ptr(gx), ptr(gW),
                                    stream_ptr()), 'dt_bilinear_bwd')
        return gx, gW_ret, None


def bilinear_interaction(x, W, bilinear_type='field_interaction'):
    """x [B,F,D], W [nW,D,D] (nW = P | F-1 | 1) -> [B,P,D]."""
    F, D = x.shape[1], x.shape[2]
    wtype = BILINEAR_TYPES[bilinear_type]
    need = {0: F * (F - 1) // 2, 1: F - 1, 2: 1}[wtype]
    if tuple(W.shape) != (need, D, D):
        raise ValueError(f'bilinear_interaction: W must be [{need},{D},{D}] for {bilinear_type}, got {tuple(W.shape)}')
    return _Bilinear.apply(x, W, wtype)


# ------------------------------------------------------------------------------------------------
# SENET squeeze + re-weight — layers.py:291-302
# ------------------------------------------------------------------------------------------------
class _FieldPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, use_max):
        require_cuda(x)
        x = _f32c(x)
        B, F, D = x.shape
        z = torch.empty((B, F), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, F), dtype=torch.int32, device=x.device) if use_max else None
        check(lib().dt_field_pool_fwd(ptr(x), B, F, D, use_max, ptr(z), ptr(arg), stream_ptr()), 'dt_field_pool_fwd')
        ctx.shape, ctx.use_max = (B, F, D), use_max
        if use_max:
            ctx.save_for_backward(arg)
        return z

    @staticmethod
    def backward(ctx, gz):
        B, F, D = ctx.shape
        arg = ctx.saved_tensors[0] if ctx.use_max else None
        gz = _f32c(gz)
        gx = torch.empty((B, F, D), dtype=torch.float32, device=gz.device)
        check(lib().dt_field_pool_bwd(ptr(gz), ptr(arg), B, F, D, ctx.use_max, ptr(gx), stream_ptr()),
              'dt_field_pool_bwd')
        return gx, None


def field_pool(x, pooling_op='mean'):
    return _FieldPool.apply(x, 1 if pooling_op == 'max' else 0)


class _FieldScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a):
        require_cuda(x, a)
        x, a = _f32c(x), _f32c(a)
        B, F, D = x.shape
        out = torch.empty_like(x)
        check(lib().dt_field_scale_fwd(ptr(x), ptr(a), B, F, D, ptr(out), stream_ptr()), 'dt_field_scale_fwd')
        ctx.save_for_backward(x, a)
        return out

    @staticmethod
    def backward(ctx, g):
        x, a = ctx.saved_tensors
        B, F, D = x.shape
        gx = torch.empty_like(x)
        ga = torch.empty_like(a)
        check(lib().dt_field_scale_bwd(ptr(x), ptr(a), ptr(_f32c(g)), B, F, D, ptr(gx), ptr(ga), stream_ptr()),
              'dt_field_scale_bwd')
        return gx, ga


def field_scale(x, a):
    """x [B,F,D] * a [B,F,1|None]."""
    return _FieldScale.apply(x, a.reshape(x.shape[0], x.shape[1]))
