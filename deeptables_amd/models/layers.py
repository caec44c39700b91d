# -*- coding:utf-8 -*-
"""Drop-in for `deeptables.models.layers` (deeptables/models/layers.py): same class names,
constructor arguments, `build/call/get_config` protocol, error behaviour and custom-object
registry — but `call` dispatches to the MI355X HIP kernels through the C-ABI (deeptables_amd.ops)
instead of a chain of TF ops.  Weight names/shapes follow the Keras layer so checkpoints map 1:1.
"""
import itertools

import numpy as np
import os

import torch
from torch import nn

from .. import ops
from ..functional import (Layer, Dense, Dropout, BatchNormalization, Activation, Concatenate, Flatten,  # noqa: F401
                          Input, Lambda, Add, SpatialDropout1D, ReduceSum, initialize, get_activation)
from ..utils import consts  # noqa: F401


def _ndim_check(x, expected, what='inputs'):
    nd = x.dim() if torch.is_tensor(x) else len(x.shape)
    if nd != expected:
        raise ValueError(f'Wrong dimensions of {what}, expected {expected} but input {nd}.')


# =============================================================================================
# FM — deeptables/models/layers.py:27-62
# =============================================================================================
class FM(Layer):
    """Factorization Machine 2nd-order term.  Input (batch, field, emb) -> output (batch, 1)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {len(input_shape)}.')
        return (input_shape[0], 1)

    def call(self, x, **kwargs):
        _ndim_check(x, 3)
        return ops.fm(x)


# =============================================================================================
# MultiheadAttention — layers.py:65-158
# =============================================================================================
class MultiheadAttention(Layer):
    """AutoInt interacting layer: relu-Dense Q/K/V(/residual) projections, per-head field attention
    (HIP kernel), residual add, relu, BatchNormalization.  Output width = embedding_size
    (the reference docstring says embedding_size*num_head; its code keeps embedding_size)."""

    def __init__(self, params, **kwargs):
        self.params = params
        self.num_heads = params.get('num_heads', 1)
        self.dropout_rate = params.get('dropout_rate', 0)
        self.use_residual = params.get('use_residual', True)
        super().__init__(**kwargs)

    def build(self, input_shape):
        self.num_units = input_shape[-1]
        if self.num_units % self.num_heads != 0:
            raise ValueError(f'embedding size {self.num_units} is not divisible by num_heads {self.num_heads}')
        mk = lambda n: Dense(self.num_units, activation='relu', kernel_initializer='he_uniform',
                             name=f'{self.name}_dense_{n}')
        self.dense_Q, self.dense_K, self.dense_V = mk('q'), mk('k'), mk('v')
        self.dense_residual = mk('residual')
        self.dropout_weights = Dropout(rate=self.dropout_rate, name=f'{self.name}_dropout')
        self.batch_normalize = BatchNormalization(name=f'{self.name}_bn')
        for l in (self.dense_Q, self.dense_K, self.dense_V, self.dense_residual, self.batch_normalize):
            l.build(input_shape)
        self.built = True

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {len(input_shape)}.')
        return tuple(input_shape)

    # functional.Model's peephole: the trailing BatchNormalization may stay pending for a consumer that applies it on load
    can_defer_output_norm = True
    accepts_pending_norm = True

    def call(self, x, defer_bn=False, **kwargs):
        _ndim_check(x, 3)
        rate = float(self.dropout_rate) if self.training else 0.0
        projs = [self.dense_Q, self.dense_K, self.dense_V] + ([self.dense_residual] if self.use_residual else [])
        if not (self.training and ops.autoint_supported(x, self.num_heads)):
            x = ops.autoint_materialize(x)          # (a pending normalisation of the layer below: only the fused path takes it)
        if ops.autoint_supported(x, self.num_heads):
            # projections + attention + dropout + residual + relu in one launch per direction (csrc/autoint.hip); the
            # four Dense layers' variables are passed as they are (no concatenated copy, no gradient split)
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if rate > 0 else 0
            bnl = self.batch_normalize
            md = self.params.get('mfma_dtype')      # 'bf16': the layer's 1e-2 mode (ops.autoint_layer)
            if self.training:        # BN with batch statistics rides along: its backward is folded into the layer's
                return ops.autoint_layer(x, [p.kernel for p in projs], [p.bias for p in projs], self.num_heads, rate, seed,
                                         batch_norm=(bnl.gamma, bnl.beta, bnl.moving_mean, bnl.moving_variance,
                                                     bnl.epsilon, bnl.momentum), mfma_dtype=md,
                                         # (functional.Model's peephole: the output's only consumer — the next interacting
                                         # layer, or Flatten -> Dense(1) — normalises it while loading)
                                         defer_bn=bool(defer_bn))
            outputs = ops.autoint_layer(x, [p.kernel for p in projs], [p.bias for p in projs], self.num_heads, rate, seed,
                                        mfma_dtype=md)
            return bnl(outputs)
        # generic shapes: one [D, 4D] GEMM (relu and bias fused) for the four projections + the attention core kernel
        W_cat = torch.cat([p.kernel for p in projs], dim=1)
        b_cat = torch.cat([p.bias for p in projs], dim=0)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if rate > 0 else 0
        if x.is_cuda and ops.dense_supported(x, W_cat):
            y = ops.dense(x, W_cat, b_cat, 'relu')
            parts = list(ops.split_cols(y, self.num_units))  # column blocks of y: the attention kernel reads them in place
        else:
            parts = [p(x) for p in projs]
        q, k, v = parts[0], parts[1], parts[2]
        outputs = ops.mha_core(q, k, v, self.num_heads, grad_cols=len(parts), dropout_rate=rate, seed=seed)
        if self.use_residual:
            outputs = outputs + parts[3]
        outputs = torch.relu(outputs)
        return self.batch_normalize(outputs)

    def get_config(self):
        c = super().get_config()
        c['params'] = self.params
        return c


# =============================================================================================
# Cross — layers.py:385-441
# =============================================================================================
class Cross(Layer):
    def __init__(self, params, **kwargs):
        self.params = params
        self.num_cross_layer = params.get('num_cross_layer', 2)
        super().__init__(**kwargs)

    def build(self, input_shape):
        num_dims = input_shape[-1]
        L = self.num_cross_layer
        # the reference keeps L pairs of (num_dims, 1) variables kernels_i / bias_i (layers.py:423-426); here they are
        # the rows of two stacked parameters, which is the layout the depth-fused kernel reads (and lets its backward
        # accumulate straight into the model's flat gradient buffer).  `kernels` / `bias` expose the Keras shapes.
        self.kernel_stack = nn.Parameter(torch.stack([initialize((num_dims, 1), 'glorot_uniform').reshape(-1)
                                                      for _ in range(L)], 0) if L else torch.zeros(0, num_dims))
        self.bias_stack = nn.Parameter(torch.zeros(L, num_dims))
        self.built = True

    @property
    def kernels(self):
        return [self.kernel_stack[i].view(-1, 1) for i in range(self.num_cross_layer)]

    @property
    def bias(self):
        return [self.bias_stack[i].view(-1, 1) for i in range(self.num_cross_layer)]

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 2:
            raise ValueError(f'Wrong dimensions of x, expected 2 but input {len(input_shape)}.')
        return tuple(input_shape)

    def call(self, x, **kwargs):
        _ndim_check(x, 2, 'x')
        if self.num_cross_layer == 0:
            return x
        return ops.cross(x, self.kernel_stack, self.bias_stack)

    def get_config(self):
        c = super().get_config()
        c['params'] = self.params
        return c


# =============================================================================================
# InnerProduct / OuterProduct — layers.py:444-586
# =============================================================================================
def _stack_fields(xs):
    """list of F x [B,1,D] -> [B,F,D] (free when they are the views of one embedding block)."""
    from ..functional import _packed_view
    pv = _packed_view(list(xs), 1)
    return pv if pv is not None else torch.cat(list(xs), dim=1)


class InnerProduct(Layer):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def compute_output_shape(self, input_shape):
        n = len(input_shape)
        if len(input_shape[0]) != 3:
            raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {len(input_shape[0])}.')
        return (input_shape[0][0], n * (n - 1) // 2)

    def call(self, x, **kwargs):
        _ndim_check(x[0], 3)
        return ops.inner_product(_stack_fields(x))


class OuterProduct(Layer):
    def __init__(self, params, **kwargs):
        self.params = params
        self.kernel_type = params.get('outer_product_kernel_type', 'mat')
        if self.kernel_type not in ['mat', 'vec', 'num']:
            raise ValueError("kernel_type must be mat,vec or num")
        super().__init__(**kwargs)

    def build(self, input_shape):
        n = len(input_shape)
        num_pairs = n * (n - 1) // 2
        embed_size = input_shape[0][-1]
        shape = {'mat': (embed_size, num_pairs, embed_size), 'vec': (num_pairs, embed_size),
                 'num': (num_pairs, 1)}[self.kernel_type]
        self.kernel = self.add_weight('kernel', shape, 'glorot_uniform')   # add_weight default initializer
        self.built = True

    def compute_output_shape(self, input_shape):
        n = len(input_shape)
        if len(input_shape[0]) != 3:
            raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {len(input_shape[0])}.')
        return (input_shape[0][0], n * (n - 1) // 2)

    def call(self, x, **kwargs):
        _ndim_check(x[0], 3)
        return ops.outer_product(_stack_fields(x), self.kernel, self.kernel_type)

    def get_config(self):
        c = super().get_config()
        c['params'] = self.params
        return c


# =============================================================================================
# CIN — layers.py:589-739
# =============================================================================================
class CIN(Layer):
    def __init__(self, params, **kwargs):
        self.params = params
        self.cross_layer_size = params.get('cross_layer_size', (128, 128,))
        self.activation = params.get('activation', 'relu')
        # extension of cin_params (config.py:120-127): 'mfma_dtype' = 'bf16x3' (default: split-bf16 operands on the bf16 matrix
        # cores — three parts / six products forward, two parts / three products backward, fp32 accumulate — held to the exact
        # kernels' bars), 'float32' (exact fp32 MFMA), 'bf16' (plain bf16 operands: the 1e-2 mode)
        self.mfma_dtype = params.get('mfma_dtype', os.environ.get('DT_AMD_CIN_DTYPE', 'bf16x3'))
        self.use_residual = params.get('use_residual', False)
        self.use_bias = params.get('use_bias', False)
        self.direct = params.get('direct', False)
        self.reduce_D = params.get('reduce_D', False)
        if len(self.cross_layer_size) == 0:
            raise ValueError("cross_layer_size must be a list(tuple) of length greater than 1")
        super().__init__(**kwargs)

    def build(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(input_shape)))
        self.field_nums = [int(input_shape[1])]
        embed_dim = input_shape[-1]
        self.f_ = nn.ParameterList()
        self.f0_ = nn.ParameterList()
        self.f__ = nn.ParameterList()
        self.bias = nn.ParameterList()
        n = len(self.cross_layer_size)
        for i, layer_size in enumerate(self.cross_layer_size):
            if self.reduce_D:
                self.f0_.append(nn.Parameter(initialize((1, layer_size, self.field_nums[0], embed_dim), 'he_uniform')))
                self.f__.append(nn.Parameter(initialize((1, layer_size, embed_dim, self.field_nums[-1]), 'he_uniform')))
            else:
                self.f_.append(nn.Parameter(
                    initialize((1, self.field_nums[-1] * self.field_nums[0], layer_size), 'he_uniform')))
            if self.use_bias:
                self.bias.append(nn.Parameter(initialize((layer_size,), 'zeros')))
            if self.direct:
                self.field_nums.append(layer_size)
            else:
                if i != n - 1 and layer_size % 2 > 0:
                    raise ValueError(
                        "cross_layer_size must be even number except for the last layer when direct=True")
                self.field_nums.append(layer_size // 2)
        out_width = self._result_width()
        if self.use_residual:
            self.exFM_out0 = Dense(self.cross_layer_size[-1], activation=self.activation,
                                   kernel_initializer='he_uniform', name=f'{self.name}_exFM_out0')
            self.exFM_out0.build((None, out_width))
            self.exFM_out = Dense(1, activation=None, name=f'{self.name}_exFM_out')
            self.exFM_out.build((None, out_width + self.cross_layer_size[-1]))
        else:
            self.exFM_out = Dense(1, activation=None, name=f'{self.name}_exFM_out')
            self.exFM_out.build((None, out_width))
        self.built = True

    def _result_width(self):
        n = len(self.cross_layer_size)
        if self.direct:
            return int(sum(self.cross_layer_size))
        return int(sum(l // 2 if i != n - 1 else l for i, l in enumerate(self.cross_layer_size)))

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {len(input_shape)}.')
        return (input_shape[0], 1)

    def _filter(self, idx, layer_size):
        """[F0*Hk, L] filter of layer idx; reduce_D composes the low-rank factors (layers.py:697-702)."""
        if not self.reduce_D:
            f = self.f_[idx]
            return f.view(f.shape[1], f.shape[2])       # (a view: `f[0]`'s select_backward zero-fills + copies a full-size gradient)
        f_m = torch.matmul(self.f0_[idx], self.f__[idx])                       # [1,L,F0,Hk]
        f_o = f_m.reshape(1, layer_size, self.field_nums[0] * self.field_nums[idx])
        return f_o.permute(0, 2, 1)[0]

    def call(self, x, **kwargs):
        _ndim_check(x, 3)
        hidden = x
        final_result = []
        n = len(self.cross_layer_size)
        for idx, layer_size in enumerate(self.cross_layer_size):
            bias = self.bias[idx] if self.use_bias else None
            curr_out = ops.cin_layer(x, hidden, self._filter(idx, layer_size), bias, self.activation, self.mfma_dtype)  # [B,L,D]
            # only sum_D of the direct-connect channels is ever used (layers.py:720-721 `reduce_sum(result, -1)`): pool each
            # layer's share where it is produced instead of concatenating [B, sum L, D] first
            if self.direct:
                hidden, pooled = curr_out, torch.sum(curr_out, -1)
            elif idx != n - 1:
                hidden, pooled = ops.cin_split_pool(curr_out, layer_size // 2)      # next layer's input (view) | pooled rest
            else:
                hidden, pooled = None, ops.cin_split_pool(curr_out, 0)[1]            # the last layer: every channel is pooled
            final_result.append(pooled)
        result = torch.cat(final_result, dim=1) if len(final_result) > 1 else final_result[0]
        if self.use_residual:
            ex0 = self.exFM_out0(result)
            return self.exFM_out(torch.cat([ex0, result], dim=1))
        return self.exFM_out(result)

    def get_config(self):
        c = super().get_config()
        c['params'] = self.params
        return c


# =============================================================================================
# MultiColumnEmbedding — layers.py:815-922
# =============================================================================================
# tables up to this many floats get an exact dense gradient (TF densifies IndexedSlices for Keras Adam);
# larger tables keep the (rows, values) pair for the row-sparse optimizer / sparse all-gather.
DENSE_GRAD_MAX_ELEMS = 1 << 22


class MultiColumnEmbedding(Layer):
    """One embedding table per categorical column, all looked up from one [B,F] input.

    HBM layout: columns with the same output_dim share ONE packed table [sum vocab, D]
    (column f starts at row_offset[f]); `embeddings[i]` are views of it with the Keras weight
    names `embeddings_{i}`.  `call` returns the reference's list of F tensors [B,1,D_f]; they are
    views of one [B,F,D] block so the graph's three Concatenate layers over them cost nothing."""

    def __init__(self, input_dims, output_dims, dropout_rate=0., embeddings_initializer='uniform',
                 embeddings_regularizer=None, activity_regularizer=None, embeddings_constraint=None,
                 mask_zero=False, **kwargs):
        kwargs.pop('input_shape', None)
        kwargs.pop('autocast', None)
        super().__init__(**kwargs)
        if not isinstance(input_dims, (list, tuple)):
            raise ValueError(f'[input_dims] must be a list or tuple.')
        if not isinstance(output_dims, (list, tuple)):
            raise ValueError(f'[output_dims] must be a list or tuple.')
        if len(input_dims) != len(output_dims):
            raise ValueError(f'The length of [input_dims] and [output_dims] must be the same.')
        self.input_dims = [int(v) for v in input_dims]
        self.output_dims = [int(v) for v in output_dims]
        self.dropout_rate = dropout_rate
        self.embeddings_initializer = embeddings_initializer
        self.mask_zero = mask_zero
        self.sparse_grads = {}      # group key -> list[SparseRowGrad] (filled by backward)
        self.check_oob = False

    def build(self, input_shape):
        if input_shape[1] == 0:
            return
        if input_shape[1] != len(self.input_dims):
            raise ValueError('The inputs dimension on axis 1 must be the same as the length of [input_dims].')
        self.groups = []   # (D, [column indices])
        for D in sorted(set(self.output_dims)):
            self.groups.append((D, [i for i, d in enumerate(self.output_dims) if d == D]))
        self.tables = nn.ParameterDict()
        for D, cols in self.groups:
            vocabs = [self.input_dims[i] for i in cols]
            rows = int(sum(vocabs))
            # initialise column by column so that each `embeddings_i` is an independent Keras init
            t = torch.empty((rows, D), dtype=torch.float32)
            o = 0
            for v in vocabs:
                t[o:o + v] = initialize((v, D), self.embeddings_initializer)
                o += v
            self.tables[f'd{D}'] = nn.Parameter(t)
            offs = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
            self.register_buffer(f'row_offset_d{D}', torch.from_numpy(offs), persistent=False)
            self.register_buffer(f'vocab_d{D}', torch.tensor(vocabs, dtype=torch.int32), persistent=False)
            self.register_buffer(f'cols_d{D}', torch.tensor(cols, dtype=torch.int64), persistent=False)
        self.register_buffer('oob_count', torch.zeros(1, dtype=torch.int32), persistent=False)
        self.dropouts = nn.ModuleList(
            [SpatialDropout1D(self.dropout_rate) for _ in self.input_dims]) if self.dropout_rate > 0 else []
        self.built = True

    @property
    def embeddings(self):
        """Per-column [vocab_i, D_i] views in column order (Keras weights `embeddings_{i}`)."""
        out = [None] * len(self.input_dims)
        for D, cols in self.groups:
            t = self.tables[f'd{D}']
            o = 0
            for i in cols:
                out[i] = t[o:o + self.input_dims[i]]
                o += self.input_dims[i]
        return out

    def set_embeddings(self, arrays):
        with torch.no_grad():
            for dst, src in zip(self.embeddings, arrays):
                dst.copy_(torch.as_tensor(np.asarray(src), dtype=torch.float32))

    def get_weights_dict(self):
        return {f'embeddings_{i}': e.detach().cpu().numpy() for i, e in enumerate(self.embeddings)}

    def add_sparse_grad(self, key, grad):
        self.sparse_grads.setdefault(key, []).append(grad)

    def uses_dense_grad(self, D):
        return self.tables[f'd{D}'].numel() <= DENSE_GRAD_MAX_ELEMS

    def compute_output_shape(self, input_shape):
        if input_shape[1] == 0:
            return []
        return [(input_shape[0], 1, d) for d in self.output_dims]

    def compute_mask(self, inputs, mask=None):
        if not self.mask_zero:
            return None
        return inputs != 0

    def lookup_group(self, inputs, D, cols):
        key = f'd{D}'
        table = self.tables[key]
        idx = inputs
        if len(self.groups) > 1 or len(cols) != inputs.shape[1]:
            idx = inputs.index_select(1, getattr(self, f'cols_{key}'))
        holder = _GroupHolder(self, key)
        return ops.embedding_lookup(idx, table, getattr(self, f'row_offset_{key}'), getattr(self, f'vocab_{key}'),
                                    holder=holder, dense_grad=self.uses_dense_grad(D),
                                    oob=self.oob_count if self.check_oob else None)

    def call(self, inputs, **kwargs):
        if inputs.shape[1] == 0:
            return []
        if inputs.dtype not in (torch.float32, torch.int32):
            inputs = inputs.to(torch.int32)
        out = [None] * len(self.input_dims)
        for D, cols in self.groups:
            emb, _rows = self.lookup_group(inputs, D, cols)       # [B, len(cols), D]
            for k, i in enumerate(cols):
                v = emb[:, k:k + 1, :]
                if self.dropout_rate > 0:
                    v = self.dropouts[i](v)
                elif len(self.groups) == 1:
                    v._dt_pack = (emb, k)      # lets Concatenate return the packed block for free
                out[i] = v
        return out

    def get_config(self):
        c = super().get_config()
        c.update(input_dims=self.input_dims, output_dims=self.output_dims, dropout_rate=self.dropout_rate,
                 embeddings_initializer=self.embeddings_initializer, mask_zero=self.mask_zero)
        return c


class _GroupHolder:
    """Receives the sparse gradient of one packed table from the gather's backward."""

    def __init__(self, layer, key):
        self.layer, self.key = layer, key

    def add_sparse_grad(self, grad):
        self.layer.add_sparse_grad(self.key, grad)


# =============================================================================================
# AFM — layers.py:742-812
# =============================================================================================
class AFM(Layer):
    """Attentional FM: pair interactions pooled by a learnt softmax attention, then Dense(1, no bias).
    The pair loop, attention MLP, softmax and weighted sum run in one HIP kernel (ops.afm_pool)."""

    def __init__(self, params, **kwargs):
        self.params = params
        self.hidden_factor = params.get('hidden_factor', 16)
        self.dropout_rate = params.get('dropout_rate', 0)
        self.activation_function = params.get('activation', 'relu')
        self.kernel_regularizer = params.get('kernel_regularizer', None)
        super().__init__(**kwargs)

    def build(self, input_shape):
        if not isinstance(input_shape, list) or len(input_shape) < 2:
            raise ValueError('A `AttentionalFM` layer should be called on a list of at least 2 inputs')
        D = int(input_shape[0][-1])
        self.dense_attention = Dense(self.hidden_factor, activation=self.activation_function,
                                     kernel_initializer='glorot_normal', name='dense_afm_attention')
        self.dense_attention.build((None, D))
        self.dense_out = Dense(1, use_bias=False, name=f'{self.name}_dense_out')
        self.dense_out.build((None, D))
        self.attention_p = self.add_weight(shape=(self.hidden_factor, 1), name='projection_h')
        self.dropout = Dropout(self.dropout_rate)
        self.built = True

    def compute_output_shape(self, input_shape):
        return (input_shape[0][0], 1)

    def call(self, x, **kwargs):
        _ndim_check(x[0], 3)
        xs = _stack_fields(x)
        attention_out = ops.afm_pool(xs, self.dense_attention.kernel, self.dense_attention.bias, self.attention_p,
                                     self.activation_function)
        attention_out = self.dropout(attention_out)
        return self.dense_out(attention_out)

    def get_config(self):
        c = super().get_config()
        c['params'] = self.params
        return c


# =============================================================================================
# SENET — layers.py:245-308
# =============================================================================================
class SENET(Layer):
    def __init__(self, pooling_op='mean', reduction_ratio=3, **kwargs):
        self.pooling_op = pooling_op
        self.reduction_ratio = reduction_ratio
        super().__init__(**kwargs)

    def build(self, input_shape):
        self.field_num = int(input_shape[1])
        self.embedding_size = int(input_shape[-1])
        self.reduction_num = max(self.field_num // self.reduction_ratio, 1)
        self.dense_att1 = Dense(self.reduction_num, activation='relu', kernel_initializer='he_uniform',
                                name=f'{self.name}_att1')
        self.dense_att1.build((None, self.field_num))
        self.dense_att2 = Dense(self.field_num, activation='relu', kernel_initializer='he_uniform',
                                name=f'{self.name}_att2')
        self.dense_att2.build((None, self.reduction_num))
        self.built = True

    def call(self, x, training=None, **kwargs):
        _ndim_check(x, 3)
        Z = ops.field_pool(x, 'max' if self.pooling_op == 'max' else 'mean')
        A1 = self.dense_att1(Z)
        A2 = self.dense_att2(A1)
        return ops.field_scale(x, A2)

    def get_config(self):
        c = super().get_config()
        c.update(reduction_ratio=self.reduction_ratio, pooling_op=self.pooling_op)
        return c


# =============================================================================================
# BilinearInteraction — layers.py:311-382
# =============================================================================================
class BilinearInteraction(Layer):
    """Weights are kept as one stacked tensor `W` [nW, D, D] whose slices are the Keras variables
    bilinear_weight / bilinear_weight{i} / bilinear_weight{i}_{j} in creation order (layers.py:346-359)."""

    def __init__(self, bilinear_type='field_interaction', **kwargs):
        self.bilinear_type = bilinear_type
        super().__init__(**kwargs)

    def weight_names(self):
        F = self.field_num
        if self.bilinear_type == 'field_all':
            return ['bilinear_weight']
        if self.bilinear_type == 'field_each':
            return [f'bilinear_weight{i}' for i in range(F - 1)]
        return [f'bilinear_weight{i}_{j}' for i, j in itertools.combinations(range(F), 2)]

    def build(self, input_shape):
        D = int(input_shape[-1])
        self.field_num = int(input_shape[1])
        n = len(self.weight_names())
        self.W = nn.Parameter(torch.stack([initialize((D, D), 'glorot_uniform') for _ in range(n)], 0))
        self.built = True

    def compute_output_shape(self, input_shape):
        F = int(input_shape[1])
        return (input_shape[0], F * (F - 1) // 2, input_shape[-1])

    def call(self, x, **kwargs):
        _ndim_check(x, 3)
        btype = self.bilinear_type if self.bilinear_type in ('field_all', 'field_each') else 'field_interaction'
        return ops.bilinear_interaction(x, self.W, btype)

    def get_config(self):
        c = super().get_config()
        c['bilinear_type'] = self.bilinear_type
        return c


# =============================================================================================
# FGCNN — layers.py:161-242
# =============================================================================================
def _same_pad(size, k, stride):
    """Keras/TF 'same' padding along one axis -> (out, pad_before, pad_after)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


class FGCNN(Layer):
    """Feature generation: Conv2D((h,1), same) -> MaxPooling2D((p,1), same) -> Flatten -> Dense recombination.
    A (h,1) convolution along the field axis is a GEMM over the h stacked field taps: the taps are gathered
    ([B*F*D, h*Cin], a strided view copy) and multiplied by the [h*Cin, filters] kernel on the library's own
    fp32-MFMA Dense kernel (csrc/dense.hip) — no MIOpen/vendor convolution; the pooling is a reshape + max.
    Tensors keep the Keras channels-last layout [B,F,D,C] (the Flatten order feeds the recombination Dense),
    kernel stored in the Keras layout [h,1,Cin,Cout]."""

    def __init__(self, filters, kernel_height, new_filters, pool_height, activation='tanh', **kwargs):
        self.filters = filters
        self.kernel_height = kernel_height
        self.new_filters = new_filters
        self.pool_height = pool_height
        self.activation = activation
        super().__init__(**kwargs)

    def build(self, input_shape):
        _, F, D, C = [None if v is None else int(v) for v in input_shape]
        self.conv_kernel = self.add_weight('conv2d_kernel', (self.kernel_height, 1, C, self.filters), 'glorot_uniform')
        self.conv_bias = self.add_weight('conv2d_bias', (self.filters,), 'zeros')
        Fp = _same_pad(F, self.pool_height, self.pool_height)[0]
        self.dense_output = Dense(F * D * self.new_filters, activation=self.activation,
                                  kernel_initializer='glorot_uniform', name=f'{self.name}_dense_output')
        self.dense_output.build((None, Fp * D * self.filters))
        self._act = get_activation(self.activation)
        self.built = True

    def compute_output_shape(self, input_shape):
        B, F, D, C = input_shape
        Fp = _same_pad(int(F), self.pool_height, self.pool_height)[0]
        return [(B, Fp, int(D), self.filters), (B, int(F) * self.new_filters, int(D))]

    def call(self, x, **kwargs):
        _ndim_check(x, 4)
        B, F, D, C = x.shape
        h = self.kernel_height
        _, pb, pa = _same_pad(F, h, 1)
        xp = torch.nn.functional.pad(x, (0, 0, 0, 0, pb, pa))             # zero-pad the field axis
        taps = xp.unfold(1, h, 1).permute(0, 1, 2, 4, 3).reshape(B * F * D, h * C)   # [B,F,D,C,h] -> rows of taps
        wk = self.conv_kernel.reshape(h * C, self.filters)
        if ops.dense_supported(taps, wk):
            out = ops.dense(taps, wk, self.conv_bias, None)
        else:
            from ..functional import note_vendor_gemm
            note_vendor_gemm(f'FGCNN {self.name!r} convolution', taps.shape, wk.shape)
            out = torch.addmm(self.conv_bias, taps, wk)
        if self._act is not None:
            out = self._act(out)
        out = out.reshape(B, F, D, self.filters)
        ph = self.pool_height
        Fp, qb, qa = _same_pad(F, ph, ph)
        if qb or qa:
            out = torch.nn.functional.pad(out, (0, 0, 0, 0, qb, qa), value=float('-inf'))
        pooling_output = out.reshape(B, Fp, ph, D, self.filters).amax(dim=2)   # [B,F',D,filters]
        new_features = self.dense_output(pooling_output.reshape(B, -1))
        new_features = new_features.reshape(-1, F * self.new_filters, D)
        return [pooling_output, new_features]

    def get_config(self):
        c = super().get_config()
        c.update(filters=self.filters, kernel_height=self.kernel_height, new_filters=self.new_filters,
                 pool_height=self.pool_height, activation=self.activation)
        return c


# =============================================================================================
# VarLenColumnEmbedding — layers.py:925-980
# =============================================================================================
class VarLenColumnEmbedding(Layer):
    """One table shared by the L positions of a multi-valued column: [B,L] ids -> [B,1,L*D].
    The gather is the same HIP kernel as MultiColumnEmbedding with every 'field' pointing at one table."""

    def __init__(self, emb_vocab_size, emb_output_dim, embeddings_initializer='uniform',
                 embeddings_regularizer=None, activity_regularizer=None, dropout_rate=0., **kwargs):
        self.emb_vocab_size = int(emb_vocab_size)
        self.emb_output_dim = int(emb_output_dim)
        self.embeddings_initializer = embeddings_initializer
        self.embeddings_regularizer = embeddings_regularizer
        self.activity_regularizer = activity_regularizer
        self.dropout_rate = dropout_rate
        super().__init__(**kwargs)
        self.dropout = None

    def compute_output_shape(self, input_shape):
        return (input_shape[0], 1, self.emb_output_dim * int(input_shape[1]))

    def build(self, input_shape=None):
        L = int(input_shape[1])
        self.embeddings = nn.Parameter(initialize((self.emb_vocab_size, self.emb_output_dim),
                                                  self.embeddings_initializer))
        self.register_buffer('row_offset', torch.zeros(L, dtype=torch.int64))
        self.register_buffer('vocab', torch.full((L,), self.emb_vocab_size, dtype=torch.int32))
        self.dropout = SpatialDropout1D(self.dropout_rate, name='var_len_emb_dropout') \
            if self.dropout_rate > 0 else None
        self.built = True

    def call(self, inputs, **kwargs):
        out, _rows = ops.embedding_lookup(inputs, self.embeddings, self.row_offset, self.vocab, dense_grad=True)
        out = out.reshape(out.shape[0], 1, -1)
        return self.dropout(out) if self.dropout is not None else out

    def compute_mask(self, inputs, mask=None):
        return None

    def get_config(self):
        c = super().get_config()
        c.update(dropout_rate=self.dropout_rate, embeddings_initializer=self.embeddings_initializer,
                 embeddings_regularizer=self.embeddings_regularizer, emb_vocab_size=self.emb_vocab_size,
                 emb_output_dim=self.emb_output_dim)
        return c


# =============================================================================================
# losses shipped with the layers module — layers.py:983-1163.  Element-wise work on [B, classes]
# predictions (a few KB per step); expressed with torch ops on the device, no dedicated kernel.
# =============================================================================================
K_EPSILON = 1e-7   # keras.backend.epsilon()


class _Loss:
    """keras.losses.Loss protocol: __call__(y_true, y_pred) = reduction(call(...)); AUTO = mean over the batch."""

    def __init__(self, reduction='auto', name=None):
        self.reduction = reduction
        self.name = name
        self.__name__ = name or self.__class__.__name__

    def __call__(self, y_true, y_pred):
        v = self.call(y_true, y_pred)
        if self.reduction in ('none', None):
            return v
        if self.reduction == 'sum':
            return v.sum()
        return v.mean()

    def get_config(self):
        return {'reduction': self.reduction, 'name': self.name}


class BinaryFocalLoss(_Loss):
    """FL(p_t) = -alpha (1-p_t)^gamma log(p_t) on sigmoid probabilities (layers.py:983-1024)."""

    def __init__(self, gamma=2., alpha=.25, reduction='auto', name='focal_loss'):
        super().__init__(reduction=reduction, name=name)
        self.gamma = float(gamma)
        self.alpha = float(alpha)

    def call(self, y_true, y_pred):
        y_true = y_true.reshape(y_pred.shape).to(y_pred.dtype)
        pt_1 = torch.where(y_true == 1, y_pred, torch.ones_like(y_pred))
        pt_0 = torch.where(y_true == 0, y_pred, torch.zeros_like(y_pred))
        pt_1 = pt_1.clamp(K_EPSILON, 1. - K_EPSILON)
        pt_0 = pt_0.clamp(K_EPSILON, 1. - K_EPSILON)
        return -(self.alpha * (1. - pt_1) ** self.gamma * torch.log(pt_1)).mean() \
            - ((1 - self.alpha) * pt_0 ** self.gamma * torch.log(1. - pt_0)).mean()

    def get_config(self):
        return dict(super().get_config(), gamma=self.gamma, alpha=self.alpha)


class CategoricalFocalLoss(_Loss):
    """Softmax focal loss, summed over classes per sample (layers.py:1027-1081)."""

    def __init__(self, gamma=2., alpha=.25, reduction='auto', name='focal_loss'):
        super().__init__(reduction=reduction, name=name)
        self.gamma = float(gamma)
        self.alpha = float(alpha)

    def call(self, y_true, y_pred):
        y_true = y_true.to(y_pred.dtype)
        y_pred = y_pred / y_pred.sum(-1, keepdim=True)
        y_pred = y_pred.clamp(K_EPSILON, 1. - K_EPSILON)
        cross_entropy = -y_true * torch.log(y_pred)
        loss = self.alpha * (1. - y_pred) ** self.gamma * cross_entropy
        return loss.sum(1)

    def get_config(self):
        return dict(super().get_config(), gamma=self.gamma, alpha=self.alpha)


class GHMCLoss:
    """Gradient Harmonising Mechanism (classification) loss on logits with a momentum-smoothed histogram of
    gradient lengths (layers.py:1084-1163)."""

    def __init__(self, bins=10, momentum=0.75):
        self.bins = bins
        self.momentum = momentum
        self.edges_left, self.edges_right = self.get_edges(self.bins)
        if momentum > 0:
            self.acc_sum = self.get_acc_sum(self.bins)

    def get_edges(self, bins):
        edges_left = torch.tensor([float(x) / bins for x in range(bins)], dtype=torch.float32).reshape(bins, 1, 1)
        right = [float(x) / bins for x in range(1, bins + 1)]
        right[-1] += 1e-6
        edges_right = torch.tensor(right, dtype=torch.float32).reshape(bins, 1, 1)
        return edges_left, edges_right

    def get_acc_sum(self, bins):
        return torch.zeros(bins, dtype=torch.float32)

    def calc(self, input, target, mask=None, is_mask=False):
        dev = input.device
        target = target.reshape(input.shape).to(input.dtype)
        edges_left, edges_right = self.edges_left.to(dev), self.edges_right.to(dev)
        mmt = self.momentum
        self.g = (torch.sigmoid(input) - target).abs().detach()
        g = self.g.unsqueeze(0)
        in_bin = (g >= edges_left) & (g < edges_right)                      # [bins, B, C]
        if is_mask:
            mask_pos = mask > 0
            inds = (in_bin & mask_pos).to(torch.float32)
            tot = torch.clamp(mask_pos.to(torch.float32).sum(), min=1.0)
        else:
            inds = in_bin.to(torch.float32)
            tot = torch.tensor(max(float(input.shape[0] * input.shape[1]), 1.0), device=dev)
        num_in_bin = inds.sum(dim=(1, 2))
        nonempty = num_in_bin > 0
        num_valid_bin = nonempty.to(torch.float32).sum()
        if mmt > 0:
            self.acc_sum = torch.where(nonempty, mmt * self.acc_sum.to(dev) + (1 - mmt) * num_in_bin,
                                       self.acc_sum.to(dev))
            denom = self.acc_sum.reshape(-1, 1, 1)
        else:
            denom = num_in_bin.reshape(-1, 1, 1)
        weights = torch.where(inds == 1, tot / denom, torch.zeros_like(inds)).sum(0)
        weights = weights / num_valid_bin
        loss = torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-input.abs()))
        return (loss * weights).sum() / tot


# =============================================================================================
# custom-object registry — layers.py:1165-1186
# =============================================================================================
def register_custom_objects(objs_dict: dict):
    import logging
    for k, v in objs_dict.items():
        if dt_custom_objects.get(k) is None:
            dt_custom_objects[k] = v
        else:
            logging.getLogger(__name__).error(f'`register_custom_objects` cannot register duplicate key [{k}].')


dt_custom_objects = {
    'MultiColumnEmbedding': MultiColumnEmbedding,
    'FM': FM,
    'AFM': AFM,
    'CIN': CIN,
    'MultiheadAttention': MultiheadAttention,
    'FGCNN': FGCNN,
    'SENET': SENET,
    'BilinearInteraction': BilinearInteraction,
    'Cross': Cross,
    'InnerProduct': InnerProduct,
    'OuterProduct': OuterProduct,
    'VarLenColumnEmbedding': VarLenColumnEmbedding,
    'CategoricalFocalLoss': CategoricalFocalLoss,
    'BinaryFocalLoss': BinaryFocalLoss,
}
