# -*- coding:utf-8 -*-
"""CPU ORACLE — test infrastructure, never shipped, never measured as the product.

Op-for-op restatement (torch CPU, float64 or float32) of the reference's hot path:
`deeptables/models/layers.py`, the net functions of `deeptables/models/deepnets.py` that wire
them, and the graph assembled by `deeptables/models/deepmodel.py:259-317`.  Each function
mirrors the TF/Keras op SEQUENCE of the cited lines (split/concat/matmul/reduce in the same
order), not a simplified closed form; `oracle/closed_form.py` holds the independent closed
forms, and `tests/test_oracle.py` checks the two against each other and against the golden
vectors in `tests/golden/`.

PARITY STATUS — pinned to the reference's SOURCE, unpinned against TensorFlow's arithmetic.  The reference ships no
golden vectors / known-answer tests for this path (all its assertions are `AUC >= 0` or shapes:
deeptables/tests/models/nets_test.py:43-44, layers_test.py:28-29) and TensorFlow/Keras (requirements.txt:1
`tensorflow>=2.4`, CI pins 2.16.2-2.18.0) is not installable in this environment.  What IS checked: the reference's
own `layers.py`, `deepnets.py`, `config.py`, `metainfo.py` and `deepmodel.py`, imported unmodified from /root/reference
onto an in-process shim of the TF / Keras primitives they call (tests/golden/make_reference_golden.py), produce outputs
this oracle reproduces to 1e-12: every layer / loss class of layers.py, every net function of deepnets.py, and WHOLE
MODELS built by the reference's DeepModel.__build_model with the reference's ModelConfig defaults — the five BASELINE.json
configurations (FM, DeepFM, xDeepFM, AutoInt, DCN), every other preset, stacking add / concat, binary / regression /
multiclass heads, BatchNormalization towers — forward AND the gradients of the task loss with respect to every weight
(autograd through the reference's graph) (78 fixtures, tests/golden/reference_code_*.npz, replayed on every CPU run by
tests/test_oracle_reference_code.py; tests/test_reference_models_gpu.py compares the HIP path with the same files): op order, axes, splits, transposes, weight shapes and the graph wiring are the
reference's.  What is NOT checked: float32 rounding / reduction order inside a TensorFlow primitive, and the Keras
defaults listed in KERAS_DEFAULTS below (BatchNormalization epsilon / momentum, initializers, Adam, BCE clipping).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Gradients of the oracle come from torch.autograd on these functions.
"""
import itertools
import math

import torch

# Keras defaults this oracle assumes (each overridable by the caller):
KERAS_DEFAULTS = {
    'bn_epsilon': 1e-3,          # keras.layers.BatchNormalization(epsilon=0.001)
    'bn_momentum': 0.99,         # keras.layers.BatchNormalization(momentum=0.99)
    'bn_variance': 'biased',     # batch variance = mean((x-mean)^2); moving var fed with it
    'embeddings_initializer': ('uniform', -0.05, 0.05),   # Keras 'uniform' == RandomUniform(+-0.05)
    'dense_kernel_initializer': 'glorot_uniform', 'dense_bias_initializer': 'zeros',
    'add_weight_default_initializer': 'glorot_uniform',   # OuterProduct.kernel etc.
    'he_uniform_limit': 'sqrt(6/fan_in)',
    'adam': dict(lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7),
    # under model.fit's tf.function Keras' TensorFlow backend recovers the logits of a sigmoid output (the output tensor's
    # op is `Sigmoid`: backend/tensorflow/nn.py _get_logits) and evaluates sigmoid_cross_entropy_with_logits; run eagerly
    # it clips the probabilities to [1e-7, 1 - 1e-7] and takes logs.  The two agree to ~1e-7 unless |logit| > 16.1, where
    # the clipped form has loss 16.1 and gradient 0.  This oracle (and the product) use the logits form; the gradient
    # fixtures (reference_code_modelgrad_*) apply the probability formula at |logit| < 12, where the forms coincide.
    'bce': 'from the logits (graph mode); probabilities clipped to [1e-7, 1-1e-7] only when run eagerly',
    'float_to_int_cast': 'truncation toward zero',
    'oob_embedding_index': 'TF-CPU raises InvalidArgument; TF-GPU returns a zero row',
}


# ---------------------------------------------------------------------------------------------
# layers.py
# ---------------------------------------------------------------------------------------------
def multi_column_embedding(inputs, tables):
    """MultiColumnEmbedding.call — layers.py:889-904.
    inputs [B,F] float32/int; tables: list of F tensors (vocab_f, D_f) -> list of F x [B,1,D_f]."""
    if inputs.shape[1] == 0:                                   # :890-891
        return []
    if inputs.dtype not in (torch.int32, torch.int64):         # :893-895  cast(float->int32) truncates
        inputs = inputs.to(torch.int32)
    columns = torch.split(inputs, 1, dim=1)                    # :896 tf.split(inputs, F, axis=1)
    out = []
    for i, col in enumerate(columns):                          # :898-903
        emb = tables[i][col.long()]                            # embedding_lookup -> [B,1,D]
        out.append(emb)
    return out


def fm(x):
    """FM.call — layers.py:53-62.  x [B,F,D] -> [B,1]."""
    if x.dim() != 3:
        raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {x.dim()}.')
    square_of_sum = torch.square(torch.sum(x, dim=1, keepdim=True))      # :56-57
    sum_of_square = torch.sum(x * x, dim=1, keepdim=True)                # :58-59
    cross = square_of_sum - sum_of_square                                # :60
    cross = 0.5 * torch.sum(cross, dim=2, keepdim=False)                 # :61
    return cross


def cross(x, kernels, biases):
    """Cross.call — layers.py:428-436.  x [B,C]; kernels[i], biases[i]: (C,1)."""
    if x.dim() != 2:
        raise ValueError(f'Wrong dimensions of x, expected 2 but input {x.dim()}.')
    x_f = x.unsqueeze(-1)                                                # :431 expand_dims
    x_n = x_f
    for i in range(len(kernels)):                                        # :433-434
        xw = torch.tensordot(x_n, kernels[i], dims=([1], [0]))           # [B,1,1]
        x_n = torch.matmul(x_f, xw) + x_n + biases[i]
    return x_n.reshape(-1, x_f.shape[1])                                 # :435


def _pair_rows_cols(n):
    row, col = [], []
    for i in range(n - 1):
        for j in range(i + 1, n):
            row.append(i)
            col.append(j)
    return row, col


def inner_product(xs):
    """InnerProduct.call — layers.py:473-487.  xs: list of F x [B,1,D] -> [B, F(F-1)/2]."""
    if xs[0].dim() != 3:
        raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {xs[0].dim()}.')
    n = len(xs)
    num_pairs = int(n * (n - 1) / 2)
    row, col = _pair_rows_cols(n)                                        # :477-482
    p = torch.cat([xs[i] for i in row], dim=1)                           # :483
    q = torch.cat([xs[j] for j in col], dim=1)                           # :484
    return torch.sum(p * q, dim=-1).reshape(-1, num_pairs)               # :485


def outer_product(xs, kernel, kernel_type='mat'):
    """OuterProduct.call — layers.py:543-581.
    kernel: mat (D,P,D) | vec (P,D) | num (P,1)."""
    if xs[0].dim() != 3:
        raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {xs[0].dim()}.')
    row, col = _pair_rows_cols(len(xs))                                  # :546-552
    p = torch.cat([xs[i] for i in row], dim=1)                           # [B,P,D]
    q = torch.cat([xs[i] for i in col], dim=1)
    if kernel_type == 'mat':                                             # :557-574
        p = p.unsqueeze(1)                                               # [B,1,P,D]
        inner = torch.sum(p * kernel, dim=-1)                            # [B,D,P]   (p*kernel: [B,D,P,D])
        inner = inner.permute(0, 2, 1)                                   # [B,P,D]
        kp = torch.sum(inner * q, dim=-1)                                # [B,P]
    else:                                                                # :575-580
        k = kernel.unsqueeze(0)
        kp = torch.sum(p * q * k, dim=-1)
    return kp


# set to a list by oracle/headline.py: every relu evaluation appends min |input|
RELU_PROBE = None
RELU_NEAR = None        # set to a list next to RELU_PROBE: per relu evaluation, how many inputs lie within 1e-6 rms of zero
RELU_TOTAL = None       # set to a list next to RELU_PROBE: per relu evaluation, how many inputs it had


def _activation(name):
    if name is None or name == 'linear':
        return lambda t: t
    if name == 'relu':
        if RELU_PROBE is not None:              # test infrastructure: how close does a relu input come to its kink?
            def probed_relu(t):
                if t.numel():
                    a = t.detach().abs()
                    RELU_PROBE.append(float(a.min()))
                    if RELU_NEAR is not None:   # units within float32 rounding of the kink: |input| < 1e-6 of the tensor's rms
                        RELU_NEAR.append(int(((a > 0) & (a < 1e-6 * float(a.pow(2).mean().sqrt()))).sum()))
                    if RELU_TOTAL is not None:
                        RELU_TOTAL.append(int(a.numel()))
                return torch.relu(t)
            return probed_relu
        return torch.relu
    if name == 'tanh':
        return torch.tanh
    if name == 'sigmoid':
        return torch.sigmoid
    # keras.activations (keras/src/activations/activations.py) — the other names `Activation(name)` accepts in
    # CIN (layers.py:709) / AFM (layers.py:783)
    if name == 'elu':
        return torch.nn.functional.elu                                  # alpha = 1
    if name == 'selu':
        return torch.selu                                               # scale 1.0507.., alpha 1.6733..
    if name == 'softplus':
        return torch.nn.functional.softplus                             # log(exp(x) + 1)
    if name == 'softsign':
        return torch.nn.functional.softsign                             # x / (|x| + 1)
    if name == 'exponential':
        return torch.exp
    raise ValueError(name)


def cin_reduced_filter(f0, f__, layer_size, n_in):
    """the filter of a reduce_D layer from its low-rank factors — layers.py:696-701.
    f0 (1, L, F0, D), f__ (1, L, D, H_k) -> (1, F0*H_k, L), n_in = F0*H_k"""
    f_m = torch.matmul(f0, f__)                                           # :698
    f_o = f_m.reshape(1, layer_size, n_in)                                # :699
    return f_o.permute(0, 2, 1)                                           # :700


def cin(x, filters, biases, cross_layer_size, activation='relu', direct=False, dense_out=None,
        dense_out0=None, return_hidden=False, reduce_factors=None):
    """CIN.call — layers.py:680-734.  reduce_D=False: filters[k] (1, F*H_k, L_k).  reduce_D=True: filters is None and
    reduce_factors[k] = (f0_k (1, L_k, F, D), f___k (1, L_k, D, H_k)), the filter is their product (:696-701).
    x [B,F,D]; biases[k]: (L_k,) or None.
    dense_out = (kernel (sum_out,1), bias (1,)) is exFM_out; dense_out0 the use_residual Dense.
    Returns exFM_out [B,1] (or the pre-Dense `result` [B, sum_out] if dense_out is None)."""
    if x.dim() != 3:
        raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {x.dim()}.')
    dim = int(x.shape[-1])
    act = _activation(activation)
    field_nums = [int(x.shape[1])]
    hidden_nn_layers = [x]
    final_result = []
    split_tensor0 = torch.split(hidden_nn_layers[0], 1, dim=2)           # :688  D x [B,F,1]
    n_layers = len(cross_layer_size)
    for idx, layer_size in enumerate(cross_layer_size):
        split_tensor = torch.split(hidden_nn_layers[-1], 1, dim=2)       # :690
        # :691 tf.matmul(split_tensor0, split_tensor, transpose_b=True) -> [D,B,F,H]
        dot_result_m = torch.stack([torch.matmul(a, b.transpose(1, 2))
                                    for a, b in zip(split_tensor0, split_tensor)], dim=0)
        dot_result_o = dot_result_m.reshape(dim, -1, field_nums[0] * field_nums[idx])   # :692
        dot_result = dot_result_o.permute(1, 0, 2)                       # :693  [B,D,F*H]
        if reduce_factors is not None:
            filt = cin_reduced_filter(reduce_factors[idx][0], reduce_factors[idx][1], layer_size,
                                      field_nums[0] * field_nums[idx])
        else:
            filt = filters[idx]                                          # :702  (1, F*H, L)
        curr_out = torch.matmul(dot_result, filt[0])                     # :703 conv1d, width-1 kernel
        if biases is not None and biases[idx] is not None:               # :704-705
            curr_out = curr_out + biases[idx]
        curr_out = act(curr_out)                                         # :707
        curr_out = curr_out.permute(0, 2, 1)                             # :708  [B,L,D]
        if direct:                                                       # :710-712
            direct_connect = curr_out
            next_hidden = curr_out
            field_nums.append(layer_size)
        else:                                                            # :713-718
            if idx != n_layers - 1:
                next_hidden, direct_connect = torch.split(curr_out, [layer_size // 2] * 2, dim=1)
            else:
                direct_connect = curr_out
                next_hidden = None
            field_nums.append(layer_size // 2)
        final_result.append(direct_connect)                              # :720
        hidden_nn_layers.append(next_hidden)
    result = torch.cat(final_result, dim=1)                              # :723
    result = torch.sum(result, dim=-1)                                   # :724  [B, sum]
    if return_hidden:
        return result
    if dense_out is None:
        return result
    if dense_out0 is not None:                                           # :726-729 use_residual
        k0, b0 = dense_out0
        ex0 = act(result @ k0 + b0)
        result = torch.cat([ex0, result], dim=1)
    k, b = dense_out
    return result @ k + b                                                # :731


def keras_batchnorm(x, gamma, beta, moving_mean=None, moving_var=None, training=True,
                    eps=KERAS_DEFAULTS['bn_epsilon'], momentum=KERAS_DEFAULTS['bn_momentum']):
    """keras.layers.BatchNormalization over the last axis [ext].  Returns (y, new_mean, new_var)."""
    red = tuple(range(x.dim() - 1))
    if training:
        mean = x.mean(dim=red)
        var = ((x - mean) ** 2).mean(dim=red)                            # biased
        y = (x - mean) / torch.sqrt(var + eps) * gamma + beta
        nm = None if moving_mean is None else moving_mean * momentum + mean * (1 - momentum)
        nv = None if moving_var is None else moving_var * momentum + var * (1 - momentum)
        return y, nm, nv
    y = (x - moving_mean) / torch.sqrt(moving_var + eps) * gamma + beta
    return y, moving_mean, moving_var


def multihead_attention(x, w, num_heads=1, use_residual=True, training=True):
    """MultiheadAttention.call — layers.py:119-153 (dropout_rate=0).
    w: dict with Q,K,V,(R): (kernel (D,D), bias (D,)), bn: (gamma, beta)."""
    if x.dim() != 3:
        raise ValueError(f'Wrong dimensions of inputs, expected 3 but input {x.dim()}.')
    relu = _activation('relu')                                           # torch.relu (probed under oracle/headline.py)
    q = relu(x @ w['Q'][0] + w['Q'][1])                                  # :123  Dense(relu)
    k = relu(x @ w['K'][0] + w['K'][1])                                  # :124
    v = relu(x @ w['V'][0] + w['V'][1])                                  # :125
    if use_residual:
        v_res = relu(x @ w['R'][0] + w['R'][1])                          # :126-127
    D = x.shape[-1]
    hs = D // num_heads
    Q_ = torch.cat(torch.split(q, hs, dim=2), dim=0)                     # :130
    K_ = torch.cat(torch.split(k, hs, dim=2), dim=0)                     # :131
    V_ = torch.cat(torch.split(v, hs, dim=2), dim=0)                     # :132
    weights = torch.matmul(Q_, K_.transpose(1, 2))                       # :135
    weights = weights / (K_.shape[-1] ** 0.5)                            # :137
    weights = torch.softmax(weights, dim=-1)                             # :139
    outputs = torch.matmul(weights, V_)                                  # :143
    outputs = torch.cat(torch.split(outputs, x.shape[0], dim=0), dim=2)  # :145
    if use_residual:
        outputs = outputs + v_res                                        # :148-149
    outputs = torch.relu(outputs)                                        # :150 (a sum of non-negative terms: exactly 0 or
                                                                         # positive in any precision, not a kink -> unprobed)
    gamma, beta = w['bn'][0], w['bn'][1]
    mm = w['bn'][2] if len(w['bn']) > 2 else None
    mv = w['bn'][3] if len(w['bn']) > 3 else None
    outputs, _, _ = keras_batchnorm(outputs, gamma, beta, mm, mv, training=training)   # :152
    return outputs


def _mha_from_parts(x, parts, names, num_heads=1, use_residual=True, training=True):
    """multihead_attention with its weight dict given as parallel lists (tests/golden/make_reference_golden.py stores
    tensors, not dicts): names[i] in {'Q','K','V','R','bn'}, parts[i] = the tuple of that entry."""
    return multihead_attention(x, {n: tuple(p) for n, p in zip(names, parts)}, num_heads, use_residual, training)


# ---------------------------------------------------------------------------------------------
# deepnets.py net functions + deepmodel.py graph (explicit-weights functional form)
# ---------------------------------------------------------------------------------------------
def dnn(x, layers_w, activation='relu', training=True):
    """deepnets.dnn — deepnets.py:401-427 with (units, dropout=0, batch_norm) cells:
    Dense(use_bias = not batch_norm) -> [BatchNormalization] -> Activation.
    layers_w: list of (kernel, bias) or, for a batch_norm cell, (kernel, None, (gamma, beta[, moving_mean, moving_var]))."""
    act = _activation(activation)
    for cell in layers_w:
        k, b = cell[0], cell[1]
        x = x @ k                                                      # :414
        if b is not None:
            x = x + b
        if len(cell) > 2 and cell[2] is not None:                      # :420-421
            bn = cell[2]
            x, _, _ = keras_batchnorm(x, bn[0], bn[1], bn[2] if len(bn) > 2 else None,
                                      bn[3] if len(bn) > 3 else None, training=training)
        x = act(x)                                                     # :422
    return x


def linear_net(embeddings, dense, kernel):
    """deepnets.linear — deepnets.py:43-66."""
    x_emb = None
    if embeddings:
        concat = torch.cat(embeddings, dim=1) if len(embeddings) > 1 else embeddings[0]   # :49
        x_emb = torch.sum(concat, dim=-1)                                                   # :51
    if x_emb is not None and dense is not None:
        x = torch.cat([x_emb, dense], dim=-1)                                               # :55
    elif x_emb is not None:
        x = x_emb
    else:
        x = dense
    return x @ kernel                                                                      # :64 Dense(1,no bias)


def model_forward(weights, cat_idx, dense, nets, config=None, training=True, return_parts=False, var_len_idx=None):
    """DeepModel.__build_model — deepmodel.py:259-317, dropout 0.  config keys read here: 'stacking_op' ('add' |
    'concat', default 'add'), 'task' ('binary' | 'regression' | 'multiclass' | 'multilabel', default 'binary').
    weights: dict keyed with the Keras layer/weight names (see tests/golden/make_golden.py).
    cat_idx [B,F] float32 (reference input contract), dense [B,Nd] or None.
    Returns the pre-activation of `task_output` (the LOGIT, [B,units]) and its activation (sigmoid probability for
    binary / multilabel, softmax for multiclass, the value itself for regression)."""
    config = config or {}
    outs, concat_emb_dense = model_nets(weights, cat_idx, dense, nets, config, training, var_len_idx)
    if len(outs) > 1:                                                # :286-301
        logits = []
        for name, out in outs.items():
            if out.dim() > 2:
                out = out.reshape(out.shape[0], -1)                  # :289-290 Flatten
            if out.shape[-1] > 1:
                out = out @ weights[f'dense_logit_{name}']           # :291-292 Dense(1, no bias)
            logits.append(out)
        stacking = config.get('stacking_op', 'add')
        if stacking == 'add':                                        # :296-297
            xs = logits[0]
            for t in logits[1:]:
                xs = xs + t
        elif stacking == 'concat':                                   # :298-299
            xs = torch.cat(logits, dim=-1)
        else:
            raise ValueError(f'Unsupported stacking_op:{stacking}.')
    elif len(outs) == 1:                                             # :302-307
        xs = next(iter(outs.values()))
        if xs.dim() > 2:
            xs = xs.reshape(xs.shape[0], -1)
    else:
        raise ValueError(f'Unexpected logit output.{outs}')
    k, b = weights['task_output']                                    # :436-457 Dense(units, activation)
    logit = xs @ k
    if b is not None:
        logit = logit + b
    task = config.get('task', 'binary')
    if task in ('binary', 'multilabel'):
        prob = torch.sigmoid(logit)
    elif task == 'multiclass':
        prob = torch.softmax(logit, dim=-1)
    elif task == 'regression':
        prob = logit
    else:
        raise ValueError(f'Unknown task type:{task}')
    if return_parts:
        return logit, prob, outs, concat_emb_dense
    return logit, prob


def model_nets(weights, cat_idx, dense, nets, config=None, training=True, var_len_idx=None):
    """the front of DeepModel.__build_model: embeddings, concat + BatchNormalization, and the output of every net
    function of `nets` (deepmodel.py:259-285, deepnets.py) -> (OrderedDict net -> output, concat_emb_dense)"""
    config = config or {}
    act_name = config.get('dnn_activation', 'relu')
    tables = weights['emb_categorical_vars_all']                     # list of (V_f, D)
    embeddings = multi_column_embedding(cat_idx, tables) if cat_idx is not None else []   # :264 / :388-404
    for ids, table in zip(var_len_idx or [], weights.get('var_len_tables', [])):     # :406-418: one [B,1,L*D] block each
        embeddings.append(var_len_embedding(ids, table))
    flatten_emb = None
    if embeddings:
        flatten_emb = torch.cat(embeddings, dim=-1).reshape(embeddings[0].shape[0], -1)   # :269-274
    if flatten_emb is not None and dense is not None:                # :348-353
        x = torch.cat([flatten_emb, dense], dim=-1)
    elif flatten_emb is not None:
        x = flatten_emb
    else:
        x = dense
    concat_emb_dense = None
    bn = weights.get('bn_concat_emb_dense')                          # :359 (pruned by keras.Model when no net
    if bn is not None:                                               #       consumes it, e.g. nets=['autoint_nets'])
        concat_emb_dense, _, _ = keras_batchnorm(x, bn[0], bn[1], bn[2] if len(bn) > 2 else None,
                                                 bn[3] if len(bn) > 3 else None, training=training)
    outs = {}
    cursor = {'fgcnn': 0, 'afm': 0}          # weights['fgcnn'] / ['afm'] list the layers of ALL nets in creation order
    if len([n for n in nets if n in ('cin_nets', 'fgcnn_cin_nets')]) > 1 or \
            len([n for n in nets if n in ('fibi_nets', 'fibi_dnn_nets')]) > 1:
        raise ValueError('the weights dict holds one CIN and one SENET / Bilinear set')
    for net in nets:                                                 # :281-285
        if net == 'linear':
            outs[net] = linear_net(embeddings, dense, weights['linear_logit'])
        elif net == 'fm_nets':
            outs[net] = fm(torch.cat(embeddings, dim=1))             # deepnets.py:88-94
        elif net == 'dnn_nets':
            outs[net] = dnn(concat_emb_dense, weights['dnn'], act_name, training)
        elif net == 'cin_nets':
            cp = config['cin_params']
            outs[net] = cin(torch.cat(embeddings, dim=1), weights['cin_filters'], weights.get('cin_bias'),
                            cp['cross_layer_size'], cp.get('activation', 'relu'), cp.get('direct', False),
                            dense_out=weights['cin_exFM_out'])       # deepnets.py:75-80
        elif net == 'dcn_nets':                                      # deepnets.py:194-207
            cross_out = cross(concat_emb_dense, weights['dcn_cross_kernels'], weights['dcn_cross_bias'])
            dnn_out = dnn(concat_emb_dense, weights['dcn_dnn'], act_name, training)
            outs[net] = torch.cat([cross_out, dnn_out], dim=-1)
        elif net == 'autoint_nets':                                  # deepnets.py:210-224
            ap = config['autoint_params']
            o = torch.cat(embeddings, dim=1)
            for lw in weights['autoint_layers']:
                o = multihead_attention(o, lw, ap.get('num_heads', 1), ap.get('use_residual', True),
                                        training=training)
            outs[net] = o.reshape(o.shape[0], -1)
        elif net == 'cross_nets':                                    # deepnets.py:172-178
            outs[net] = cross(concat_emb_dense, weights['cross_kernels'], weights['cross_bias'])
        elif net == 'cross_dnn_nets':                                # deepnets.py:181-191
            c = cross(concat_emb_dense, weights['cross_dnn_kernels'], weights['cross_dnn_bias'])
            outs[net] = dnn(c, weights['cross_dnn'], act_name, training)
        elif net in ('opnn_nets', 'ipnn_nets', 'pnn_nets'):          # deepnets.py:110-160
            if len(embeddings) < 2:
                continue                                             # the net function returns None: dropped (:284)
            kt = config.get('pnn_params', {}).get('outer_product_kernel_type', 'mat')
            if net == 'opnn_nets':
                parts = [outer_product(embeddings, weights['opnn_kernel'], kt)]
            elif net == 'ipnn_nets':
                parts = [inner_product(embeddings)]
            else:
                parts = [inner_product(embeddings), outer_product(embeddings, weights['pnn_kernel'], kt)]
            outs[net] = dnn(torch.cat(parts + [concat_emb_dense], dim=-1), weights[net[:-5]], act_name, training)
        elif net == 'afm_nets':                                      # deepnets.py:99-107
            a = weights['afm'][cursor['afm']]
            cursor['afm'] += 1
            outs[net] = afm(embeddings, a['att_kernel'], a['att_bias'], a['projection_h'], a['out_kernel'],
                            a.get('activation', 'relu'))
        elif net in ('fibi_nets', 'fibi_dnn_nets'):                  # deepnets.py:344-386
            fp = config.get('fibinet_params', {})
            e = torch.cat(embeddings, dim=1)
            se = weights['senet'][0]
            senet_embedding = senet(e, se['att1'], se['att2'], fp.get('senet_pooling_op', 'mean'))
            btype = fp.get('bilinear_type', 'field_interaction')
            Ws, We = weights['bilinear']['senet'], weights['bilinear']['embedding']
            senet_bilinear_out = bilinear_interaction(senet_embedding, [Ws[i] for i in range(len(Ws))], btype)
            bilinear_out = bilinear_interaction(e, [We[i] for i in range(len(We))], btype)
            fibi = torch.cat([senet_bilinear_out, bilinear_out], dim=1)
            if net == 'fibi_nets':
                outs[net] = fibi
            else:
                outs[net] = dnn(torch.cat([fibi.reshape(fibi.shape[0], -1), dense], dim=-1), weights['fibi_dnn'],
                                act_name, training)
        elif net.startswith('fgcnn_') or net == 'fg_nets':           # deepnets.py:227-341
            gp = config.get('fgcnn_params', {})
            e = torch.cat(embeddings, dim=1)
            fg_inputs = e.unsqueeze(-1)
            new_features = []
            pools, nfs = gp.get('fg_pool_heights', (2, 2)), gp.get('fg_new_feat_filters', (2, 2))
            depth = min(len(gp.get('fg_filters', (14, 16))), len(gp.get('fg_heights', (7, 7))), len(pools), len(nfs))
            mine = weights['fgcnn'][cursor['fgcnn']:cursor['fgcnn'] + depth]     # every fg_nets call builds its own
            cursor['fgcnn'] += depth                                             # FGCNN layers (deepnets.py:251-258)
            for lw, pool, nf in zip(mine, pools, nfs):
                fg_inputs, nfeat = fgcnn(fg_inputs, lw['conv_kernel'], lw['conv_bias'], lw['dense_kernel'],
                                         lw['dense_bias'], pool, nf)
                new_features.append(nfeat)
            fg_output = torch.cat(new_features + [e], dim=1)
            flat = fg_output.reshape(fg_output.shape[0], -1)
            if net == 'fg_nets':
                outs[net] = fg_output
            elif net == 'fgcnn_fm_nets':
                outs[net] = fm(fg_output)
            elif net == 'fgcnn_cin_nets':
                cp = config['cin_params']
                outs[net] = cin(fg_output, weights['cin_filters'], weights.get('cin_bias'), cp['cross_layer_size'],
                                cp.get('activation', 'relu'), cp.get('direct', False),
                                dense_out=weights['cin_exFM_out'])
            elif net == 'fgcnn_afm_nets':
                a = weights['afm'][cursor['afm']]
                cursor['afm'] += 1
                outs[net] = afm(list(torch.split(fg_output, 1, dim=1)), a['att_kernel'], a['att_bias'],
                                a['projection_h'], a['out_kernel'], a.get('activation', 'relu'))
            elif net == 'fgcnn_ipnn_nets':
                parts = [flat, inner_product(list(torch.split(fg_output, 1, dim=1)))]
                if dense is not None:
                    parts.append(dense)
                outs[net] = dnn(torch.cat(parts, dim=-1), weights['fgcnn_ipnn'], act_name, training)
            elif net == 'fgcnn_dnn_nets':
                x_in = torch.cat([flat, dense], dim=-1) if dense is not None else flat
                outs[net] = dnn(x_in, weights['fgcnn_dnn'], act_name, training)
            else:
                raise ValueError(net)
        else:
            raise ValueError(net)
    return outs, concat_emb_dense


def _nets_from_parts(cat_idx, dense, tables, bn, names, parts, nets, config, net):
    """model_forward's per-net output `outs[net]` with the weight dict given as parallel lists
    (tests/golden/make_reference_golden.py stores tensors, not dicts).  names[i] is a key of the weights dict and
    parts[i] its value; 'autoint_layers' is a list of [Q, K, V, R|None, bn] lists."""
    w = {'emb_categorical_vars_all': list(tables), 'bn_concat_emb_dense': tuple(bn)}
    for n, p in zip(names, parts):
        if n == 'autoint_layers':
            w[n] = [{k: tuple(v) for k, v in zip(('Q', 'K', 'V', 'R', 'bn'), lw) if v is not None} for lw in p]
        elif n in ('dnn', 'dcn_dnn'):
            w[n] = [tuple(kb) for kb in p]
        elif n == 'cin_exFM_out':
            w[n] = tuple(p)
        else:
            w[n] = p
    outs, _ = model_nets(w, cat_idx, dense, [net], config, training=True)
    return outs[net]


def _model_from_parts(cat_idx, dense, weights, nets, config, var_len_idx=None):
    """model_forward in training mode -> [logit | activation] side by side (tests/golden/make_reference_golden.py stores
    the reference's `task_output` pre-activation and output the same way).  Lists that stand for (kernel, bias) pairs
    arrive as lists; the restatement only indexes them."""
    logit, prob = model_forward(weights, cat_idx, dense, nets, config, training=True, var_len_idx=var_len_idx)
    return torch.cat([logit, prob], dim=-1)


def _leaves(nest):
    """tensors of a nested dict / list / tuple in traversal order (dict: key order as stored), None skipped"""
    if nest is None or isinstance(nest, str):
        return []
    if isinstance(nest, dict):
        return [t for v in nest.values() for t in _leaves(v)]
    if isinstance(nest, (list, tuple)):
        return [t for v in nest for t in _leaves(v)]
    return [nest]


def _map_leaves(nest, fn):
    if nest is None or isinstance(nest, str):
        return nest
    if isinstance(nest, dict):
        return {k: _map_leaves(v, fn) for k, v in nest.items()}
    if isinstance(nest, (list, tuple)):
        return [_map_leaves(v, fn) for v in nest]
    return fn(nest)


def model_loss(weights, cat_idx, dense, y, nets, config=None, training=True, var_len_idx=None):
    """the loss DeepModel.__compile_model selects for the task (deepmodel.py:319-346): BinaryCrossentropy (from the
    logits), MeanSquaredError, CategoricalCrossentropy — each the mean over the batch (Keras SUM_OVER_BATCH_SIZE)"""
    config = config or {}
    logit, out = model_forward(weights, cat_idx, dense, nets, config, training=training, var_len_idx=var_len_idx)
    task = config.get('task', 'binary')
    if task in ('binary', 'multilabel'):
        z, t = logit, y.reshape(logit.shape).to(logit.dtype)
        return (torch.clamp(z, min=0) - z * t + torch.log1p(torch.exp(-torch.abs(z)))).mean()
    if task == 'regression':
        return ((out - y.reshape(out.shape).to(out.dtype)) ** 2).mean()
    if task == 'multiclass':
        return -(torch.log_softmax(logit, dim=-1) * y.to(logit.dtype)).sum(-1).mean()
    raise ValueError(task)


def _model_grads_from_parts(cat_idx, dense, weights, y, nets, config, var_len_idx=None):
    """d model_loss / d every tensor of `weights` (traversal order of _leaves), flattened into one vector; a weight the
    graph does not use (the concat BatchNormalization of a model none of whose nets reads it) contributes zeros"""
    w = _map_leaves(weights, lambda t: t.detach().clone().requires_grad_(True))
    leaves = _leaves(w)
    loss = model_loss(w, cat_idx, dense, y, nets, config, training=True, var_len_idx=var_len_idx)
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    return torch.cat([(torch.zeros_like(t) if g is None else g).reshape(-1) for t, g in zip(leaves, grads)])


def _ghmc_from_parts(input, target, acc_sum, bins=10, momentum=0.75):
    """ghmc_loss -> [loss, updated acc_sum...] as one vector"""
    loss, acc = ghmc_loss(input, target, acc_sum, bins, momentum)
    return torch.cat([loss.reshape(1), acc.reshape(-1)])


def _fgcnn_from_parts(x, conv_kernel, conv_bias, dense_kernel, dense_bias, pool_height, new_filters, activation='tanh'):
    """fgcnn -> [pooled | new_features] flattened side by side"""
    pooled, feats = fgcnn(x, conv_kernel, conv_bias, dense_kernel, dense_bias, pool_height, new_filters, activation)
    return torch.cat([pooled.reshape(pooled.shape[0], -1), feats.reshape(feats.shape[0], -1)], dim=-1)


# ---------------------------------------------------------------------------------------------
# layers.py — the f3 layer types (AFM, SENET, BilinearInteraction, FGCNN, VarLenColumnEmbedding, losses)
# ---------------------------------------------------------------------------------------------
def afm(xs, att_kernel, att_bias, projection_h, out_kernel, activation='relu'):
    """AFM.call — layers.py:786-807.  xs: list of F tensors [B,1,D]; att_kernel [D,H] (dense_afm_attention),
    projection_h [H,1], out_kernel [D,1] (Dense(1, use_bias=False)); dropout_rate 0."""
    row, col = [], []
    for r, c in itertools.combinations(xs, 2):                        # :792-794
        row.append(r)
        col.append(c)
    p = torch.cat(row, dim=1)                                          # :795
    q = torch.cat(col, dim=1)                                          # :796
    bi_interaction = p * q                                             # :797
    attention_2 = bi_interaction @ att_kernel                          # :799 Dense(hidden_factor, activation)
    if att_bias is not None:
        attention_2 = attention_2 + att_bias
    act = _activation(activation)
    if act is not None:
        attention_2 = act(attention_2)
    attention_score = torch.softmax(torch.tensordot(attention_2, projection_h, dims=([-1], [0])), dim=1)   # :800
    attention_out = torch.sum(attention_score * bi_interaction, dim=1)  # :801
    return attention_out @ out_kernel                                   # :803


def senet(x, att1, att2, pooling_op='mean'):
    """SENET.call — layers.py:291-302.  att1/att2 = (kernel, bias) of the two relu Dense layers."""
    if pooling_op == 'max':
        Z = torch.max(x, dim=-1).values                                # :296
    else:
        Z = torch.mean(x, dim=-1)                                      # :298
    A1 = torch.relu(Z @ att1[0] + att1[1])                             # :299
    A2 = torch.relu(A1 @ att2[0] + att2[1])                            # :300
    return x * A2.unsqueeze(2)                                         # :301


def bilinear_interaction(x, W_list, bilinear_type='field_interaction'):
    """BilinearInteraction.call — layers.py:363-377.  W_list: list of [D,D] in creation order
    (field_all: 1, field_each: F-1, field_interaction: one per pair)."""
    F = x.shape[1]
    xs = list(torch.split(x, 1, dim=1))                                # :366
    if bilinear_type == 'field_all':
        p = [torch.tensordot(v_i, W_list[0], dims=([-1], [0])) * v_j for v_i, v_j in itertools.combinations(xs, 2)]
    elif bilinear_type == 'field_each':
        p = [torch.tensordot(xs[i], W_list[i], dims=([-1], [0])) * xs[j]
             for i, j in itertools.combinations(range(F), 2)]
    else:
        p = [torch.tensordot(v[0], w, dims=([-1], [0])) * v[1]
             for v, w in zip(itertools.combinations(xs, 2), W_list)]
    return torch.cat(p, dim=1)                                         # :375


def _same_pad_1d(size, k, stride):
    """TF 'SAME' padding: out = ceil(size/stride); total = max((out-1)*stride + k - size, 0); before = total//2."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


def fgcnn(x, conv_kernel, conv_bias, dense_kernel, dense_bias, pool_height, new_filters, activation='tanh'):
    """FGCNN.call — layers.py:220-230.  x [B,F,D,C] channels-last; conv_kernel [h,1,C,filters] (Keras layout),
    Conv2D(strides 1, padding 'same', activation) -> MaxPooling2D((pool,1), padding 'same', stride = pool) ->
    Flatten (channels-last order) -> Dense(F*D*new_filters, activation) -> reshape [B, F*new_filters, D]."""
    B, F, D, C = x.shape
    h = conv_kernel.shape[0]
    act = _activation(activation)
    _, pb, pa = _same_pad_1d(F, h, 1)
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, pb, pa))              # zero-pad the field axis
    # explicit correlation: out[b,f,d,o] = sum_{t,c} xp[b,f+t,d,c] k[t,0,c,o] + bias[o]
    out = torch.zeros(B, F, D, conv_kernel.shape[3], dtype=x.dtype)
    for t in range(h):
        out = out + torch.einsum('bfdc,co->bfdo', xp[:, t:t + F], conv_kernel[t, 0])
    out = out + conv_bias
    if act is not None:
        out = act(out)
    Fp, qb, qa = _same_pad_1d(F, pool_height, pool_height)
    outp = torch.nn.functional.pad(out, (0, 0, 0, 0, qb, qa), value=float('-inf'))
    pooled = torch.stack([outp[:, i * pool_height:(i + 1) * pool_height].max(dim=1).values for i in range(Fp)],
                         dim=1)                                        # [B,Fp,D,filters]
    new_features = pooled.reshape(B, -1) @ dense_kernel + dense_bias   # :226-227
    if act is not None:
        new_features = act(new_features)
    new_features = new_features.reshape(-1, F * new_filters, D)        # :228-229
    return pooled, new_features


def var_len_embedding(inputs, table):
    """VarLenColumnEmbedding.call — layers.py:961-968 (dropout 0): Embedding then reshape [B,1,L*D]."""
    idx = inputs.to(torch.int64)
    e = table[idx]                                                     # [B,L,D]
    return e.reshape(e.shape[0], 1, -1)


K_EPSILON = 1e-7


def binary_focal_loss(y_true, y_pred, gamma=2., alpha=.25):
    """BinaryFocalLoss.call — layers.py:1006-1017."""
    pt_1 = torch.where(y_true == 1, y_pred, torch.ones_like(y_pred))
    pt_0 = torch.where(y_true == 0, y_pred, torch.zeros_like(y_pred))
    pt_1 = torch.clamp(pt_1, K_EPSILON, 1. - K_EPSILON)
    pt_0 = torch.clamp(pt_0, K_EPSILON, 1. - K_EPSILON)
    return -torch.mean(alpha * torch.pow(1. - pt_1, gamma) * torch.log(pt_1)) \
        - torch.mean((1 - alpha) * torch.pow(pt_0, gamma) * torch.log(1. - pt_0))


def categorical_focal_loss(y_true, y_pred, gamma=2., alpha=.25):
    """CategoricalFocalLoss.call — layers.py:1062-1076 (per-sample; Keras AUTO reduction then means)."""
    y_pred = y_pred / torch.sum(y_pred, dim=-1, keepdim=True)
    y_pred = torch.clamp(y_pred, K_EPSILON, 1. - K_EPSILON)
    cross_entropy = -y_true * torch.log(y_pred)
    loss = alpha * torch.pow(1. - y_pred, gamma) * cross_entropy
    return torch.sum(loss, dim=1)


def ghmc_loss(input, target, acc_sum, bins=10, momentum=0.75):
    """GHMCLoss.calc (is_mask False) — layers.py:1111-1162.  Returns (loss, updated acc_sum)."""
    edges_left = torch.tensor([float(x) / bins for x in range(bins)], dtype=input.dtype).reshape(bins, 1, 1)
    right = [float(x) / bins for x in range(1, bins + 1)]
    right[-1] += 1e-6
    edges_right = torch.tensor(right, dtype=input.dtype).reshape(bins, 1, 1)
    g = torch.abs(torch.sigmoid(input) - target).detach().unsqueeze(0)
    inds = ((g >= edges_left) & (g < edges_right)).to(input.dtype)
    tot = max(float(input.shape[0] * input.shape[1]), 1.0)
    num_in_bin = inds.sum(dim=(1, 2))
    nonempty = num_in_bin > 0
    num_valid_bin = nonempty.to(input.dtype).sum()
    if momentum > 0:
        acc_sum = torch.where(nonempty, momentum * acc_sum + (1 - momentum) * num_in_bin, acc_sum)
        denom = acc_sum.reshape(-1, 1, 1) + torch.zeros_like(inds)
    else:
        denom = num_in_bin.reshape(-1, 1, 1) + torch.zeros_like(inds)
    weights = torch.where(inds == 1, tot / denom, torch.zeros_like(inds)).sum(0)
    weights = weights / num_valid_bin
    loss = torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))
    return torch.sum(loss * weights) / tot, acc_sum


def binary_crossentropy_from_logits(logit, y):
    """keras.losses.BinaryCrossentropy on a sigmoid output, evaluated from logits (graph mode):
    mean(max(z,0) - z*y + log1p(exp(-|z|)))."""
    z = logit.reshape(-1)
    y = y.reshape(-1).to(z.dtype)
    return (torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-torch.abs(z)))).mean()


# ---------------------------------------------------------------------------------------------
# keras.optimizers.Adam (selected by DeepModel.__compile_model, deepmodel.py:321-322)
# ---------------------------------------------------------------------------------------------
def keras_adam_step(p, g, m, v, t, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    """One Keras Adam update (keras/src/optimizers/adam.py update_step): t is the 1-based step number.
        alpha = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t);  m += (g - m)(1 - beta_1);  v += (g^2 - v)(1 - beta_2)
        p -= alpha * m / (sqrt(v) + epsilon)
    Returns (p, m, v).  A sparse (IndexedSlices) gradient is densified by Keras 3 before this update; the product's
    row-sparse variant applies the same formula to the touched rows only (documented deviation)."""
    alpha = lr * math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
    m = m + (g - m) * (1.0 - beta_1)
    v = v + (g * g - v) * (1.0 - beta_2)
    p = p - alpha * m / (torch.sqrt(v) + epsilon)
    return p, m, v


# ---------------------------------------------------------------------------------------------
# Keras initializers (for building reference-shaped random weights)
# ---------------------------------------------------------------------------------------------
def glorot_uniform(shape, gen, fan_in=None, fan_out=None, dtype=torch.float32):
    if fan_in is None:
        fan_in, fan_out = _fans(shape)
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul(limit).to(dtype)


def he_uniform(shape, gen, fan_in=None, dtype=torch.float32):
    if fan_in is None:
        fan_in, _ = _fans(shape)
    limit = math.sqrt(6.0 / fan_in)
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul(limit).to(dtype)


def _fans(shape):
    """keras.initializers compute_fans: 2-D (in,out); N-D: receptive field * in / out."""
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = 1
    for s in shape[:-2]:
        rf *= s
    return shape[-2] * rf, shape[-1] * rf


def pairs(n):
    return list(itertools.combinations(range(n), 2))
