# -*- coding:utf-8 -*-
"""CPU: the arithmetic behind the split-bf16 matrix-core kernels (csrc/tower_x3.h, csrc/cin_bf16.hip),
restated in numpy with an exact bf16 rounding — no GPU, no kernel: what the kernels RELY on.

  a = a1 + a2 + a3,  a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)      (exact: 8 + 8 + 8 mantissa bits)
  a . b = sum_{p,q} a_p b_q;  bf16 x bf16 products are exact in fp32
  forward  (three parts, the six products with p + q <= 4): error of the fp32-rounding class (2^-24 of sum |a||b|)
  backward (two parts, three products):                      2^-17 class
  plain bf16 (one product):                                  2^-9 class (north_star's 1e-2 mode)"""
import numpy as np
import pytest


def bf16(x):
    """float32 -> the nearest bfloat16 (round to nearest even), returned as float32"""
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(x, parts):
    out, r = [], np.asarray(x, dtype=np.float32).copy()
    for _ in range(parts):
        p = bf16(r)
        out.append(p)
        r = (r - p).astype(np.float32)          # exact in fp32: p agrees with r in its leading bits
    return out, r


def test_three_bf16_parts_hold_every_bit_of_a_float32():
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.randn(20000).astype(np.float32) * s for s in (1e-3, 1.0, 37.0)] +
                       [np.float32([0.0, 1.0, -1.0, 1.0e-30, 65504.0, 1.0 + 2.0 ** -23])])   # (normal range: residuals of
    # values within 2^16 of the smallest normal number underflow — the kernels' inputs are activations and weights)
    parts, rest = split(x, 3)
    assert np.all(rest == 0)                                            # nothing left after three parts
    assert np.array_equal((parts[0].astype(np.float64) + parts[1] + parts[2]).astype(np.float32), x)
    two, rest2 = split(x, 2)
    big = np.abs(x) > 1e-30
    assert np.max(np.abs(rest2[big] / x[big])) <= 2.0 ** -16            # two parts: 16 bits


def test_a_bf16_product_is_exact_in_float32():
    rng = np.random.RandomState(1)
    a, b = bf16(rng.randn(100000)), bf16(rng.randn(100000))
    assert np.array_equal((a.astype(np.float64) * b.astype(np.float64)).astype(np.float32).astype(np.float64),
                          a.astype(np.float64) * b.astype(np.float64))   # 8 x 8 mantissa bits fit 24


@pytest.mark.parametrize('K', [32, 448, 1664])
def test_kept_products_bound_the_error_of_a_dot_product(K):
    """errors of the kept partial products against the exact dot product, relative to sum |a||b| (what a relu input's
    rounding is measured against): six products of three parts ~ 2^-24 (as fp32 accumulation itself), three products of
    two parts ~ 2^-17, one product ~ 2^-9"""
    rng = np.random.RandomState(K)
    n = 2000
    a = rng.randn(n, K).astype(np.float32)
    b = (rng.randn(n, K) * 0.1).astype(np.float32)
    exact = np.einsum('nk,nk->n', a.astype(np.float64), b.astype(np.float64))
    scale = np.einsum('nk,nk->n', np.abs(a).astype(np.float64), np.abs(b).astype(np.float64))
    a3, _ = split(a, 3)
    b3, _ = split(b, 3)

    def kept(order):
        acc = np.zeros(n, dtype=np.float64)
        for p in range(3):
            for q in range(3):
                if p + q <= order:
                    acc += np.einsum('nk,nk->n', a3[p].astype(np.float64), b3[q].astype(np.float64))
        return acc

    six, three, one = kept(2), kept(1), kept(0)
    e6 = np.max(np.abs(six - exact) / scale)
    e3 = np.max(np.abs(three - exact) / scale)
    e1 = np.max(np.abs(one - exact) / scale)
    assert e6 < 2.0 ** -23, e6                  # dropped terms: a2 b3, a3 b2, a3 b3 — 2^-24 .. 2^-32 of the products
    assert 2.0 ** -23 < e3 < 2.0 ** -15, e3     # dropped: the 2^-16 tier (random signs: ~ 2^-17 / sqrt(K) of sum |a||b|)
    assert 2.0 ** -12 < e1 < 2.0 ** -7, e1      # dropped: the 2^-8 tier
    # two-part operands give the same three products (a1 b1, a1 b2, a2 b1): the backward kernels' form
    a2, _ = split(a, 2)
    b2, _ = split(b, 2)
    back = sum(np.einsum('nk,nk->n', a2[p].astype(np.float64), b2[q].astype(np.float64))
               for p in range(2) for q in range(2) if p + q <= 1)
    assert np.allclose(back, three, rtol=0, atol=1e-12 * scale.max())


def test_relu_decisions_of_the_six_product_forward_are_fp32_class():
    """a relu unit flips when the arithmetic error exceeds |pre-activation|.  Dense128 on a Criteo-shaped input, 2 M units:
    fp32 arithmetic and the six-product forward flip none, three products a few (each one moves a weight gradient by a whole
    sample's term: the 2e-3 .. 0.11 gradient errors of the first version, DESIGN.md §3.5 — why the forward GEMMs keep six),
    plain bf16 about 0.07 % of the units (north_star's 1e-2 mode: its gradient bars are L2 bars for that reason)"""
    rng = np.random.RandomState(6)
    rows, K, H = 16384, 448, 128
    x = rng.randn(rows, K).astype(np.float32)
    w = (rng.randn(K, H) / np.sqrt(K)).astype(np.float32)
    exact = x.astype(np.float64) @ w.astype(np.float64)
    x3, _ = split(x, 3)
    w3, _ = split(w, 3)

    def kept(order):
        acc = np.zeros((rows, H), dtype=np.float64)
        for p in range(3):
            for q in range(3):
                if p + q <= order:
                    acc += x3[p].astype(np.float64) @ w3[q].astype(np.float64)
        return acc

    fp32 = (x @ w).astype(np.float64)                   # numpy's float32 GEMM: the rounding class the exact kernels have
    flips = lambda y: int(np.sum((y > 0) != (exact > 0)))
    f32_flips, six_flips, three_flips, one_flips = flips(fp32), flips(kept(2)), flips(kept(1)), flips(kept(0))
    assert six_flips <= f32_flips + 1, (six_flips, f32_flips)
    assert three_flips >= 1 and three_flips > six_flips, (three_flips, six_flips)
    assert 500 < one_flips < 5000, one_flips
