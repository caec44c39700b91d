# -*- coding:utf-8 -*-
"""CPU ORACLE (test infrastructure) — independent closed forms in numpy float64.

Second leg of the parity protocol: `reference_layers.py` transcribes the
reference's op sequence literally; the formulas below are the algebraic closed forms of the
same layers.  tests/test_oracle.py requires the two to agree before either
is trusted to judge the HIP kernels.  Citations: deeptables/models/layers.py line ranges.
"""
import numpy as np


def fm(x):                       # layers.py:53-62
    s = x.sum(axis=1)
    q = (x * x).sum(axis=1)
    return 0.5 * (s * s - q).sum(axis=1, keepdims=True)


def cross(x, w, b):              # layers.py:428-436 ; w,b [L,C]
    x0 = x
    xl = x
    for l in range(w.shape[0]):
        s = xl @ w[l]
        xl = x0 * s[:, None] + xl + b[l]
    return xl


def cross_scalar_form(xhat, gamma, beta, w, b, w3c, dz):
    """The algebra the fused DCN step runs (csrc/deepfm.hip, kernels C and E'), in numpy float64, tile = the whole batch.
    Inputs: xhat [B,C] = (X - mean) rstd, BN affine gamma / beta [C] (x0 = gamma xhat + beta), cross kernels / biases
    w, b [L,C], w3c [C] (the part of the output kernel applied to the cross output), dz [B] = d loss / d z.
    By induction on x_{l+1} = x0 (x_l . w_l) + x_l + b_l (layers.py:428-436):
        x_l = a_l x0 + c_l,  a_0 = 1,  c_l = b_0 + .. + b_{l-1};   s_l = a_l p_l + q_l,  p_l = x0 . w_l,  q_l = c_l . w_l
    Returns dict: z_c = w3c . x_L [B]; dXn = d loss / d x0 through the cross network [B,C]; dw [L,C], db [L,C], dw3c [C];
    sum_dx = sum_b dXn and sum_dx_xhat = sum_b dXn * xhat (the cross path's share of BatchNormalization's backward sums)."""
    L, C = w.shape
    x0 = xhat * gamma + beta
    Wc = np.concatenate([w, w3c[None]], 0)                    # [L+1, C]: w_0 .. w_{L-1}, w3c
    P = x0 @ Wc.T                                             # kernel C: one skinny GEMM
    BW = b @ Wc.T                                             # the Gram rows b_j . Wc_l (third row tile of that GEMM)
    c = np.concatenate([np.zeros((1, C)), np.cumsum(b, 0)], 0)                         # c_0 .. c_L
    q = np.array([BW[:l, l].sum() for l in range(L + 1)])                              # q_l = c_l . Wc_l
    a = np.ones((x0.shape[0], L + 1))
    for l in range(L):                                        # the L scalar steps per row
        a[:, l + 1] = a[:, l] + a[:, l] * P[:, l] + q[l]
    z_c = a[:, L] * P[:, L] + q[L]
    # backward: scalars per row
    coeff = np.zeros((x0.shape[0], L + 1))
    A_next = np.zeros((x0.shape[0], L))
    A = dz * P[:, L]
    coeff[:, L] = dz * a[:, L]
    for l in range(L - 1, -1, -1):
        coeff[:, l] = A * a[:, l]
        A_next[:, l] = A
        A = A * (1.0 + P[:, l])
    dXn = coeff @ Wc                                          # kernel C: rows to HBM
    G = xhat.T @ coeff                                        # [C, L+1]: the tile record (MFMA, K = rows)
    Sco, SA, Sdz = coeff.sum(0), A_next.sum(0), dz.sum()      # the record's scalar block (ones^T . M)
    # kernel E': per column
    dw = np.stack([gamma * G[:, l] + beta * Sco[l] + SA[l] * c[l] for l in range(L)])
    dw3c = gamma * G[:, L] + beta * Sco[L] + Sdz * c[L]
    db = np.zeros((L, C))
    suffix = Sdz * w3c
    for j in range(L - 1, -1, -1):
        db[j] = suffix
        suffix = suffix + SA[j] * w[j]
    return {'z_c': z_c, 'dXn': dXn, 'dw': dw, 'db': db, 'dw3c': dw3c,
            'sum_dx': (Sco[:, None] * Wc).sum(0), 'sum_dx_xhat': (Wc * G.T).sum(0)}


def pair_index(F):
    return [(i, j) for i in range(F - 1) for j in range(i + 1, F)]


def inner_product(x):            # layers.py:473-487 ; x [B,F,D]
    pi = pair_index(x.shape[1])
    return np.stack([(x[:, i] * x[:, j]).sum(-1) for i, j in pi], axis=1)


def outer_product(x, kernel, kernel_type='mat'):   # layers.py:543-581
    pi = pair_index(x.shape[1])
    P = len(pi)
    out = np.zeros((x.shape[0], P), dtype=x.dtype)
    for p, (i, j) in enumerate(pi):
        if kernel_type == 'mat':    # kernel [D,P,D]:  sum_a sum_d x_i[d] K[a,p,d] x_j[a]
            out[:, p] = np.einsum('bd,ad,ba->b', x[:, i], kernel[:, p, :], x[:, j])
        elif kernel_type == 'vec':
            out[:, p] = (x[:, i] * x[:, j] * kernel[p]).sum(-1)
        else:
            out[:, p] = kernel[p, 0] * (x[:, i] * x[:, j]).sum(-1)
    return out


def cin_layer(x0, xk, W, bias=None, relu=True):    # layers.py:689-710 ; W [F0*Hk, L]
    B, F0, D = x0.shape
    Hk = xk.shape[1]
    Wr = W.reshape(F0, Hk, -1)
    y = np.einsum('bid,bjd,ijl->bld', x0, xk, Wr)
    if bias is not None:
        y = y + bias[None, :, None]
    return np.maximum(y, 0) if relu else y


def cin(x, filters, cross_layer_size, direct=False, relu=True):   # -> result [B, sum] (layers.py:724)
    hidden = x
    outs = []
    n = len(cross_layer_size)
    for idx, L in enumerate(cross_layer_size):
        y = cin_layer(x, hidden, filters[idx], None, relu)
        if direct:
            outs.append(y)
            hidden = y
        elif idx != n - 1:
            hidden, dc = y[:, :L // 2], y[:, L // 2:]
            outs.append(dc)
        else:
            outs.append(y)
    return np.concatenate(outs, axis=1).sum(-1)


def mha_core(q, k, v, H):        # layers.py:129-145 ; q,k,v [B,F,D] already relu(Dense)
    B, F, D = q.shape
    dh = D // H
    qh = q.reshape(B, F, H, dh).transpose(0, 2, 1, 3)
    kh = k.reshape(B, F, H, dh).transpose(0, 2, 1, 3)
    vh = v.reshape(B, F, H, dh).transpose(0, 2, 1, 3)
    s = np.einsum('bhid,bhjd->bhij', qh, kh) / np.sqrt(dh)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = np.einsum('bhij,bhjd->bhid', p, vh)
    return o.transpose(0, 2, 1, 3).reshape(B, F, D)


def batchnorm_train(x, gamma, beta, eps=1e-3):
    mean = x.mean(0)
    var = x.var(0)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta
