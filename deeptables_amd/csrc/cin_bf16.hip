// cin_bf16.hip — the CIN layer of cin.hip in bf16-MFMA mode (north_star "1e-2 bf16"): v_mfma_f32_32x32x16_bf16, bf16
// operands, fp32 accumulation.  Same math and interface as cin.hip (CIN.call, deeptables/models/layers.py:689-710):
//     Y[(b,d), l] = act( sum_{k=(i,j)} x0[b,i,d] xk[b,j,d] W[k, l] + bias[l] )
// An OPT-IN precision mode (cin_params['mfma_dtype'] = 'bf16'): results are within 1e-2 of the float64 oracle instead of
// 1e-4; the fp32 kernels of cin.hip stay the default and the benchmark headline never uses this file.
//
// A bf16 MFMA consumes 8 consecutive k per lane and instruction (K = 16), so the operands are laid out for 16-byte
// LDS reads:
//   * k' = i * Hp + j with Hp = Hk rounded up to 8 (zero columns): a lane's 8 k' share one i -> ONE x0 value times
//     eight consecutive xk values (two ds_read_b128 from an [m][j] tile), rounded once to bf16;
//   * the filter is re-laid once per call by k_cin_pack_w: WT [L][K'] (forward: B operand rows = filters, 8 k' per
//     read) and WN [(i, 32-j block)][L] (dgrad: A operand rows = (i,j), 8 l per read), both bf16, zero padded;
//   * wgrad contracts over the rows m = (b,d): x0 / xk / G are staged [i][m], [j][m], [l][m] so that 8 consecutive m
//     are one read.
// Tile shapes, the T-never-stored dgrad and the batch-split wgrad follow cin.hip.
//
// SPLIT-bf16 mode (round 4; dt_cin_layer_fwd_bf16x3 / _bwd_bf16x3, cin_params['mfma_dtype'] = 'bf16x3'): the same three
// kernels with every fp32 operand split into NP bf16 parts (a = a_1 + .. + a_NP, each the rounding of what the parts before
// it left; three parts = all 24 mantissa bits) and the products a_p b_q with p + q <= NP + 1 on the matrix cores, fp32
// accumulate — as csrc/tower_x3.h does for the Dense tower.  FORWARD: three parts, six products (the dropped terms are 2^-24
// of the product): fp32-class outputs, so the layer's relu units land on the other side of their kink no more often than
// with fp32 arithmetic.  BACKWARD (dgrad, wgrad): two parts, three products (2^-17 per product).  gfx950's fp32 MFMA runs
// at 1/16 of the bf16 rate: 6/16 resp. 3/16 of its time.
#include <stdlib.h>
#include "common.h"

namespace dt {

typedef float cb_f16v __attribute__((ext_vector_type(16)));
typedef float cb_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 cb_b8 __attribute__((ext_vector_type(8)));

constexpr int kBM = 128;     // (b,d) rows per block tile
constexpr int kBN = 128;     // filters per block tile
constexpr int kBK = 32;      // k' per LDS chunk of W^T (forward)
constexpr int kBWS = kBK + 8;    // bf16 row stride of a W^T chunk: 80 B -> 16 rows' 16-byte reads hit distinct banks
constexpr int kBMC = 64;     // rows m per wgrad LDS chunk
constexpr int kBKT = 2;      // 32-row k sub-tiles per wave (wgrad)

__device__ __forceinline__ cb_b8 cb_pack(const cb_f4& lo, const cb_f4& hi) {
    cb_b8 r;
    r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
    r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
    return r;
}
__device__ __forceinline__ cb_b8 cb_zero() {
    cb_b8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)0.f;
    return r;
}
// v = out[0] + .. + out[NP-1]: bf16 parts, each the rounding of what the parts before it left
template <int NP>
__device__ __forceinline__ void cb_split(const float (&v)[8], cb_b8 (&out)[NP]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float r = v[e];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const __bf16 a = (__bf16)r;
            out[k][e] = a;
            r -= (float)a;
        }
    }
}
// acc += sum_{p + q < NP} a[p] b[q], smallest terms first
template <int NP>
__device__ __forceinline__ void cb_mma(cb_f16v& acc, const cb_b8 (&a)[NP], const cb_b8 (&b)[NP]) {
#pragma unroll
    for (int t = NP - 1; t >= 0; --t)
#pragma unroll
        for (int pa = 0; pa <= t; ++pa) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[t - pa], acc, 0, 0, 0);
}
__host__ __device__ inline int cb_hp(int Hk) { return (Hk + 7) & ~7; }
__host__ __device__ inline int cb_lq(int L) { return (L + 15) & ~15; }

// W [F0*Hk][L] fp32 -> WT [Lp][Kp] bf16 (Lp = L rounded up to 128, Kp = F0*Hp, k' = i*Hp + j) and
//                      WN [F0*njb*32][Lq] bf16 (row (i*njb + jb)*32 + r <-> j = 32 jb + r), zero padded
template <int NP>
__global__ __launch_bounds__(256) void k_cin_pack_w(const float* __restrict__ W, int F0, int Hk, int L,
                                                    __bf16* __restrict__ WT, __bf16* __restrict__ WN) {
    const int Hp = cb_hp(Hk), Kp = F0 * Hp, Lp = (L + kBN - 1) / kBN * kBN, Lq = cb_lq(L), njb = (Hk + 31) / 32;
    const int64_t nT = (int64_t)Lp * Kp, nN = (int64_t)F0 * njb * 32 * Lq;        // elements of ONE part of each layout
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nT + nN; e += (int64_t)gridDim.x * blockDim.x) {
        float r;
        __bf16* dst;
        int64_t stride;
        if (e < nT) {
            if (!WT) continue;
            const int n = (int)(e / Kp), kp = (int)(e - (int64_t)n * Kp);
            const int i = kp / Hp, j = kp - i * Hp;
            r = (n < L && j < Hk) ? W[((int64_t)i * Hk + j) * L + n] : 0.f;
            dst = WT + e; stride = nT;
        } else {
            if (!WN) continue;
            const int64_t q = e - nT;
            const int row = (int)(q / Lq), l = (int)(q - (int64_t)row * Lq);
            const int ib = row >> 5, rr = row & 31;
            const int i = ib / njb, j = (ib - i * njb) * 32 + rr;
            r = (l < L && j < Hk) ? W[((int64_t)i * Hk + j) * L + l] : 0.f;
            dst = WN + q; stride = nN;
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const __bf16 a = (__bf16)r;
            dst[k * stride] = a;
            r -= (float)a;
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <bool kAnyAct, int NP>  // kAnyAct false: linear / relu only (see k_cin_fwd); NP: bf16 parts per operand (1: plain bf16 mode)
__global__ __launch_bounds__(256, 2) void k_cin_fwd_bf16(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const __bf16* __restrict__ WT, int64_t wt_part, const float* __restrict__ bias, int act, int B, int F0, int Hk, int L,
    int D, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KCH = NP == 1 ? kBK : 16;              // k' per LDS chunk of W^T (split mode: NP x the bytes per k')
    constexpr int WSb = KCH + 8;                         // bf16 row stride of a chunk row: 80 / 48 B -> distinct bank groups
    const int Hp = cb_hp(Hk), Kp = F0 * Hp;
    const int F0S = F0 | 1, HS = Hp + 4;
    const int64_t M = (int64_t)B * D;
    float* x0T = lds;                    // [kBM][F0S]
    float* xkT = x0T + kBM * F0S;        // [kBM][HS], zero beyond Hk
    __bf16* wtb = reinterpret_cast<__bf16*>(xkT + kBM * HS);     // [2][NP][kBN][WSb]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * kBM;
    const int n0 = blockIdx.y * kBN;

    for (int e = threadIdx.x; e < kBM * F0; e += 256) {
        const int i = e / kBM, r = e - i * kBM;
        const int64_t m = m0 + r;
        x0T[r * F0S + i] = m < M ? x0[(m / D) * x0_bs + (int64_t)i * D + (m % D)] : 0.f;
    }
    for (int e = threadIdx.x; e < kBM * Hp; e += 256) {
        const int j = e / kBM, r = e - j * kBM;
        const int64_t m = m0 + r;
        xkT[r * HS + j] = (m < M && j < Hk) ? xk[(m / D) * xk_bs + (int64_t)j * D + (m % D)] : 0.f;
    }
    // W^T chunk loader: NP parts x 128 filters x KCH k' bf16 in 16-byte pieces, kWR per thread
    constexpr int kPieces = NP * kBN * (KCH / 8), kWR = (kPieces + 255) / 256;
    cb_f4 wreg[kWR];
    auto load_w = [&](int chunk) {
#pragma unroll
        for (int u = 0; u < kWR; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int pc = e % (KCH / 8), n = (e / (KCH / 8)) % kBN, part = e / ((KCH / 8) * kBN);
            const int kp = chunk * KCH + 8 * pc;
            wreg[u] = (e < kPieces && kp < Kp)
                          ? *reinterpret_cast<const cb_f4*>(WT + part * wt_part + (int64_t)(n0 + n) * Kp + kp)
                          : cb_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < kWR; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int pc = e % (KCH / 8), n = (e / (KCH / 8)) % kBN, part = e / ((KCH / 8) * kBN);
            if (e < kPieces) *reinterpret_cast<cb_f4*>(wtb + ((buf * NP + part) * kBN + n) * WSb + 8 * pc) = wreg[u];
        }
    };
    cb_f16v acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int nchunks = (Kp + KCH - 1) / KCH;
    load_w(0);
    store_w(0);
    __syncthreads();
    const int row = wave * 32 + c;
    const float* x0r = x0T + row * F0S;
    const float* xkr = xkT + row * HS;
    const int nblocks_n = min(4, (L - n0 + 31) / 32);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) load_w(chunk + 1);
#pragma unroll
        for (int st = 0; st < KCH / 16; ++st) {
            const int kp = chunk * KCH + 16 * st + 8 * s;
            cb_b8 a[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) a[q] = cb_zero();
            if (kp < Kp) {
                const int i = kp / Hp, j0 = kp - i * Hp;
                const float xv = x0r[i];
                const cb_f4 lo = *reinterpret_cast<const cb_f4*>(xkr + j0) * xv, hi = *reinterpret_cast<const cb_f4*>(xkr + j0 + 4) * xv;
                const float z[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                cb_split<NP>(z, a);
            }
            const __bf16* wrow = wtb + (buf * NP * kBN + c) * WSb + 16 * st + 8 * s;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                if (nb < nblocks_n) {
                    cb_b8 b[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const cb_b8*>(wrow + (q * kBN + nb * 32) * WSb);
                    cb_mma<NP>(acc[nb], a, b);
                }
        }
        if (chunk + 1 < nchunks) store_w(buf ^ 1);
        __syncthreads();
    }
    // epilogue (as cin.hip): row = (r&3) + 8*(r>>2) + 4*s, col = c
    const bool vec_ok = (D % 4 == 0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb >= nblocks_n) continue;
        const int n = n0 + nb * 32 + c;
        if (n >= L) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t mr = m0 + wave * 32 + 8 * g + 4 * s;
            if (mr >= M) continue;
            const int64_t b = mr / D;
            const int dd = (int)(mr % D);
            float ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = kAnyAct ? act_apply(acc[nb][g * 4 + r] + bv, act)
                                : (act == DT_ACT_RELU ? fmaxf(acc[nb][g * 4 + r] + bv, 0.f) : acc[nb][g * 4 + r] + bv);
            if (vec_ok) {
                *reinterpret_cast<float4*>(y + (b * L + n) * D + dd) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            } else {
                for (int r = 0; r < 4; ++r) {
                    const int64_t mm = mr + r;
                    if (mm < M) y[((mm / D) * L + n) * D + (mm % D)] = ov[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward, split-bf16, WITHOUT forming Z (round 4).  Y[m, l] = sum_i x0[m, i] T_i[m, l],  T_i[m, l] = sum_j xk[m, j] W[(i, j), l]:
// the inner sums are F0 small GEMMs whose operands are x_k and W_i THEMSELVES — the A operand (the x_k row of the lane, three
// bf16 parts) is split ONCE per tile and stays in registers for all F0 GEMMs, the B operand is the pre-split filter — and the
// outer sum is one scalar x0[m, i] per accumulator row: 16 FMAs per 32x32 tile and i.  The Z-forming kernel above multiplies
// and splits eight products per lane and MFMA step (56 VALU operations for 24 MFMAs) and ran 404 us per 128-filter layer at
// the Criteo shape, three times its matrix time.
// Block = 128 rows m x 128 filters; W_i (three parts x 128 filters x Hp k) goes through ONE LDS buffer per block — two blocks
// per CU, the other block's MFMAs cover this block's refill.
// ------------------------------------------------------------------------------------------
// Four-wave form (128 rows per block, two blocks per CU), round 4's first version: small batches.
template <bool kAnyAct, int KS /* ceil(Hk / 16) upper bound: 2, 4 or 8 */>
__global__ __launch_bounds__(256, 2) void k_cin_fwd_noz4(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const __bf16* __restrict__ WT, int64_t wt_part, const float* __restrict__ bias, int act, int B, int F0, int Hk, int L,
    int D, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NP = 3;
    const int Hp = cb_hp(Hk), Kp = F0 * Hp;
    constexpr int HPM = 16 * KS;                          // k per i the kernel walks (Hp <= HPM; beyond Hp: zero operands)
    constexpr int WSb = HPM + 8;                          // bf16 row stride of a filter row: an odd number of 16-byte slots
    const int64_t M = (int64_t)B * D;
    float* x0T = lds;                                     // [F0][kBM]  x0[m][i], rows m contiguous
    __bf16* wtb = reinterpret_cast<__bf16*>(x0T + F0 * kBM);       // [NP][kBN][WSb]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * kBM;
    const int n0 = blockIdx.y * kBN;
    for (int e = threadIdx.x; e < kBM * F0; e += 256) {
        const int i = e / kBM, r = e - i * kBM;
        const int64_t m = m0 + r;
        x0T[i * kBM + r] = m < M ? x0[(m / D) * x0_bs + (int64_t)i * D + (m % D)] : 0.f;
    }
    // this lane's x_k row, split once: A operand of every GEMM (k = 16 ks + 8 s + e)
    cb_b8 a[KS][NP];
    {
        const int64_t m = m0 + wave * 32 + c;
        const bool ok = m < M;
        const float* src = xk + (ok ? (m / D) * xk_bs + (m % D) : 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 16 * ks + 8 * s + e;
                v[e] = (ok && j < Hk) ? src[(int64_t)j * D] : 0.f;
            }
            cb_split<NP>(v, a[ks]);
        }
    }
    // W_i loader: NP parts x 128 filters x (HPM / 8) 16-byte pieces
    constexpr int kPieces = NP * kBN * (HPM / 8), kWR = (kPieces + 255) / 256;
    cb_f4 wreg[kWR];
    auto load_w = [&](int i) {
#pragma unroll
        for (int u = 0; u < kWR; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int pc = e % (HPM / 8), n = (e / (HPM / 8)) % kBN, part = e / ((HPM / 8) * kBN);
            wreg[u] = (e < kPieces && 8 * pc < Hp)
                          ? *reinterpret_cast<const cb_f4*>(WT + part * wt_part + (int64_t)(n0 + n) * Kp + (int64_t)i * Hp + 8 * pc)
                          : cb_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < kWR; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int pc = e % (HPM / 8), n = (e / (HPM / 8)) % kBN, part = e / ((HPM / 8) * kBN);
            if (e < kPieces) *reinterpret_cast<cb_f4*>(wtb + (part * kBN + n) * WSb + 8 * pc) = wreg[u];
        }
    };
    cb_f16v acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int nblocks_n = min(4, (L - n0 + 31) / 32);
    load_w(0);
    for (int i = 0; i < F0; ++i) {
        __syncthreads();                 // every wave is done with W_{i-1} (and, first time, x0T is complete)
        store_w();
        __syncthreads();
        if (i + 1 < F0) load_w(i + 1);   // in flight under this i's MFMAs
        // x0[m][i] of the 16 accumulator rows of this lane: rows (r & 3) + 8 (r >> 2) + 4 s
        cb_f4 xq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xq[g] = *reinterpret_cast<const cb_f4*>(x0T + i * kBM + wave * 32 + 8 * g + 4 * s);
        const __bf16* wrow = wtb + c * WSb + 8 * s;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb >= nblocks_n) continue;
            cb_f16v t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                cb_b8 b[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const cb_b8*>(wrow + (q * kBN + nb * 32) * WSb + 16 * ks);
                cb_mma<NP>(t, a[ks], b);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] += xq[r >> 2][r & 3] * t[r];
        }
    }
    // epilogue (as cin.hip): row = (r&3) + 8*(r>>2) + 4*s, col = c
    const bool vec_ok = (D % 4 == 0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb >= nblocks_n) continue;
        const int n = n0 + nb * 32 + c;
        if (n >= L) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t mr = m0 + wave * 32 + 8 * g + 4 * s;
            if (mr >= M) continue;
            const int64_t b = mr / D;
            const int dd = (int)(mr % D);
            float ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = kAnyAct ? act_apply(acc[nb][g * 4 + r] + bv, act)
                                : (act == DT_ACT_RELU ? fmaxf(acc[nb][g * 4 + r] + bv, 0.f) : acc[nb][g * 4 + r] + bv);
            if (vec_ok) {
                *reinterpret_cast<float4*>(y + (b * L + n) * D + dd) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            } else {
                for (int r = 0; r < 4; ++r) {
                    const int64_t mm = mr + r;
                    if (mm < M) y[((mm / D) * L + n) * D + (mm % D)] = ov[r];
                }
            }
        }
    }
}


// WV = waves per block (32 rows m each).  WV = 8 (large batches): 256 rows per block — every block streams ALL of the filter's
// parts from L2 (1.28 MB at the Criteo shape), so the 128-row blocks moved 1.3 GB per layer and ran at the L2's pace, not
// the matrix cores'; twice the rows per block halve that stream.  One 512-thread block per CU then, so W_i is double
// buffered in LDS (one barrier per i; the other block of a CU no longer covers the refill).
template <bool kAnyAct, int KS /* ceil(Hk / 16) upper bound: 2, 4 or 8 */, int WV>
__global__ __launch_bounds__(64 * WV, WV == 4 ? 2 : 1) void k_cin_fwd_noz(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const __bf16* __restrict__ WT, int64_t wt_part, const float* __restrict__ bias, int act, int B, int F0, int Hk, int L,
    int D, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NP = 3, BM = 32 * WV, NT = 64 * WV, NBUF = WV == 8 ? 2 : 1;
    const int Hp = cb_hp(Hk), Kp = F0 * Hp;
    constexpr int HPM = 16 * KS;                          // k per i the kernel walks (Hp <= HPM; beyond Hp: zero operands)
    constexpr int WSb = HPM + 8;                          // bf16 row stride of a filter row: an odd number of 16-byte slots
    const int64_t M = (int64_t)B * D;
    float* x0T = lds;                                     // [F0][BM]  x0[m][i], rows m contiguous
    __bf16* wtb = reinterpret_cast<__bf16*>(x0T + F0 * BM);        // [NBUF][NP][kBN][WSb]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * kBN;
    for (int e = threadIdx.x; e < BM * F0; e += NT) {
        const int i = e / BM, r = e - i * BM;
        const int64_t m = m0 + r;
        x0T[i * BM + r] = m < M ? x0[(m / D) * x0_bs + (int64_t)i * D + (m % D)] : 0.f;
    }
    // this lane's x_k row, split once: A operand of every GEMM (k = 16 ks + 8 s + e)
    cb_b8 a[KS][NP];
    {
        const int64_t m = m0 + wave * 32 + c;
        const bool ok = m < M;
        const float* src = xk + (ok ? (m / D) * xk_bs + (m % D) : 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 16 * ks + 8 * s + e;
                v[e] = (ok && j < Hk) ? src[(int64_t)j * D] : 0.f;
            }
            cb_split<NP>(v, a[ks]);
        }
    }
    // W_i loader: NP parts x 128 filters x (HPM / 8) 16-byte pieces
    constexpr int kPieces = NP * kBN * (HPM / 8), kWR = (kPieces + NT - 1) / NT;
    cb_f4 wreg[kWR];
    auto load_w = [&](int i) {
#pragma unroll
        for (int u = 0; u < kWR; ++u) {
            const int e = threadIdx.x + NT * u;
            const int pc = e % (HPM / 8), n = (e / (HPM / 8)) % kBN, part = e / ((HPM / 8) * kBN);
            wreg[u] = (e < kPieces && 8 * pc < Hp)
                          ? *reinterpret_cast<const cb_f4*>(WT + part * wt_part + (int64_t)(n0 + n) * Kp + (int64_t)i * Hp + 8 * pc)
                          : cb_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < kWR; ++u) {
            const int e = threadIdx.x + NT * u;
            const int pc = e % (HPM / 8), n = (e / (HPM / 8)) % kBN, part = e / ((HPM / 8) * kBN);
            if (e < kPieces) *reinterpret_cast<cb_f4*>(wtb + ((buf * NP + part) * kBN + n) * WSb + 8 * pc) = wreg[u];
        }
    };
    cb_f16v acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int nblocks_n = min(4, (L - n0 + 31) / 32);
    auto gemm_i = [&](int i, int buf) {
        // x0[m][i] of the 16 accumulator rows of this lane: rows (r & 3) + 8 (r >> 2) + 4 s
        cb_f4 xq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xq[g] = *reinterpret_cast<const cb_f4*>(x0T + i * BM + wave * 32 + 8 * g + 4 * s);
        const __bf16* wrow = wtb + (buf * NP * kBN + c) * WSb + 8 * s;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb >= nblocks_n) continue;
            cb_f16v t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                cb_b8 b[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const cb_b8*>(wrow + (q * kBN + nb * 32) * WSb + 16 * ks);
                cb_mma<NP>(t, a[ks], b);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] += xq[r >> 2][r & 3] * t[r];
        }
    };
    load_w(0);
    if constexpr (NBUF == 1) {
        for (int i = 0; i < F0; ++i) {
            __syncthreads();                 // every wave is done with W_{i-1} (and, first time, x0T is complete)
            store_w(0);
            __syncthreads();
            if (i + 1 < F0) load_w(i + 1);   // in flight under this i's MFMAs
            gemm_i(i, 0);
        }
    } else {
        store_w(0);
        if (F0 > 1) load_w(1);
        __syncthreads();
        for (int i = 0; i < F0; ++i) {
            gemm_i(i, i & 1);
            // W_{i+1} (in registers since the last iteration) -> the other buffer: its last readers (i - 1) are behind the
            // barrier that ended that iteration
            if (i + 1 < F0) store_w((i + 1) & 1);
            if (i + 2 < F0) load_w(i + 2);
            __syncthreads();
        }
    }
    // epilogue (as cin.hip): row = (r&3) + 8*(r>>2) + 4*s, col = c
    const bool vec_ok = (D % 4 == 0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        if (nb >= nblocks_n) continue;
        const int n = n0 + nb * 32 + c;
        if (n >= L) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t mr = m0 + wave * 32 + 8 * g + 4 * s;
            if (mr >= M) continue;
            const int64_t b = mr / D;
            const int dd = (int)(mr % D);
            float ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = kAnyAct ? act_apply(acc[nb][g * 4 + r] + bv, act)
                                : (act == DT_ACT_RELU ? fmaxf(acc[nb][g * 4 + r] + bv, 0.f) : acc[nb][g * 4 + r] + bv);
            if (vec_ok) {
                *reinterpret_cast<float4*>(y + (b * L + n) * D + dd) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            } else {
                for (int r = 0; r < 4; ++r) {
                    const int64_t mm = mr + r;
                    if (mm < M) y[((mm / D) * L + n) * D + (mm % D)] = ov[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// dgrad: grad_x0 (=), grad_xk (=): both OVERWRITTEN (one block owns a 128-row tile of m and all of its (i, j)).  T^T[(i, 32 j's), m] = sum_l W[(i,j), l] G[m, l] per chunk, contracted in the
// lane that owns column m against xk / x0; grad_x0 is gathered in LDS and flushed once.
// ------------------------------------------------------------------------------------------
// Four-wave form (128 rows per block, two blocks per CU: small batches), as in round 3; the eight-wave form for large
// batches follows.  (A version that kept BOTH in one template and staged W through registers two chunks ahead left the
// four-wave form with one block per CU — 601 us instead of 345 — and the eight-wave form 16 % slower: gpurun_out/r04_*.)
template <int LSTEPS /* ceil(L/16) upper bound: 8 or 16 */, int JB /* ceil(Hk/32) upper bound */, int NP /* bf16 parts */>
__global__ __launch_bounds__(256) void k_cin_dgrad_bf16_4(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const __bf16* __restrict__ WN, int64_t wn_part, const float* __restrict__ y, const float* __restrict__ gy, int act,
    int B, int F0, int Hk, int L, int D, float* __restrict__ gx0, float* __restrict__ gxk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int64_t M = (int64_t)B * D;
    const int F0S = F0 | 1, Lq = cb_lq(L), WS = 16 * LSTEPS + 8;
    float* x0T = lds;                    // [kBM][F0S]
    float* g0T = x0T + kBM * F0S;        // [kBM][F0S] grad_x0 of the tile
    __bf16* wtl = reinterpret_cast<__bf16*>(g0T + kBM * F0S);   // [2][NP][32][WS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * kBM;
    for (int e = threadIdx.x; e < kBM * F0; e += 256) {
        const int i = e / kBM, r = e - i * kBM;
        const int64_t mm = m0 + r;
        x0T[r * F0S + i] = mm < M ? x0[(mm / D) * x0_bs + (int64_t)i * D + (mm % D)] : 0.f;
        g0T[r * F0S + i] = 0.f;
    }
    const int row = wave * 32 + c;
    const int64_t m = m0 + row;          // the column of T^T this lane owns
    const bool mvalid = m < M;
    const int64_t b = mvalid ? m / D : 0;
    const int d = mvalid ? (int)(m % D) : 0;

    // G[m][l] for l = 16 st + 8 s + e as the B operand, kept in registers for the whole tile
    cb_b8 G[LSTEPS][NP];
#pragma unroll
    for (int st = 0; st < LSTEPS; ++st) {
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = 16 * st + 8 * s + e;
            float g = 0.f;
            if (mvalid && l < L) {
                const int64_t o = (b * L + l) * D + d;
                g = gy[o] * act_grad_from_y(y[o], act);
            }
            gv[e] = g;
        }
        cb_split<NP>(gv, G[st]);
    }
    // xk values this lane needs: j = jb*32 + (r&3) + 8*(r>>2) + 4*s
    float xkv[JB][16], gxk_acc[JB][16];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * s;
            xkv[jb][r] = (mvalid && j < Hk) ? xk[b * xk_bs + (int64_t)j * D + d] : 0.f;
            gxk_acc[jb][r] = 0.f;
        }
    const int njb = (Hk + 31) / 32;
    const int nchunks = F0 * njb;
    // W chunk (i, jb): 32 rows x Lq bf16, contiguous in WN
    auto stage_w = [&](int chunk, int buf) {
        const __bf16* src = WN + (int64_t)chunk * 32 * Lq;
        const int pieces = Lq / 8;                      // 16-byte pieces per row
        for (int e = threadIdx.x; e < NP * 32 * pieces; e += 256) {
            const int part = e / (32 * pieces), q = e - part * 32 * pieces;
            const int r = q / pieces, pc = q - r * pieces;
            *reinterpret_cast<cb_f4*>(wtl + ((buf * NP + part) * 32 + r) * WS + 8 * pc) =
                *reinterpret_cast<const cb_f4*>(src + part * wn_part + r * Lq + 8 * pc);
        }
    };
    const int lsteps = Lq / 16;
    stage_w(0, 0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const int i = chunk / njb, jb = chunk - i * njb;
        if (chunk + 1 < nchunks) stage_w(chunk + 1, buf ^ 1);
        const __bf16* wrow = wtl + (buf * NP * 32 + c) * WS + 8 * s;
        cb_f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < LSTEPS; ++st)
            if (st < lsteps) {
                cb_b8 a[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const cb_b8*>(wrow + q * 32 * WS + 16 * st);
                cb_mma<NP>(acc, a, G[st]);
            }
        // contract T^T[j, m] (16 j's in this lane) against xk and x0
        const float x0v = x0T[row * F0S + i];
        float p = 0.f;
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
            if (jj == jb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p += xkv[jj][r] * acc[r];
                    gxk_acc[jj][r] += x0v * acc[r];
                }
            }
        p += __shfl_xor(p, 32, 64);
        if (s == 0) g0T[row * F0S + i] += p;  // unique owner of (m, i)
        __syncthreads();
    }
    if (mvalid) {
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (j < Hk) gxk[(b * Hk + j) * D + d] = gxk_acc[jb][r];
            }
    }
    for (int e = threadIdx.x; e < kBM * F0; e += 256) {
        const int i = e / kBM, r = e - i * kBM;
        const int64_t mm = m0 + r;
        if (mm < M) gx0[((mm / D) * F0 + i) * D + (mm % D)] = g0T[r * F0S + i];     // this block is the only writer of (m, i)
    }
}


// WV = waves per block, 32 columns m each.  Every block streams ALL of W_N through LDS (852 KB at the Criteo shape with two
// parts): 1024 blocks of 128 rows moved 870 MB per layer and the kernel ran at the L2's pace (345 us, 19 % of the matrix
// rate); WV = 8 (large batches) halves the stream.  The next chunk's pieces wait in registers while this chunk's MFMAs run.
template <int LSTEPS /* ceil(L/16) upper bound: 8 or 16 */, int JB /* ceil(Hk/32) upper bound */, int NP /* bf16 parts */, int WV>
__global__ __launch_bounds__(64 * WV) void k_cin_dgrad_bf16(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const __bf16* __restrict__ WN, int64_t wn_part, const float* __restrict__ y, const float* __restrict__ gy, int act,
    int B, int F0, int Hk, int L, int D, float* __restrict__ gx0, float* __restrict__ gxk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = 32 * WV, NT = 64 * WV;
    const int64_t M = (int64_t)B * D;
    const int F0S = F0 | 1, Lq = cb_lq(L);
    constexpr int WS = 16 * LSTEPS + 8;
    float* x0T = lds;                    // [BM][F0S]
    float* g0T = x0T + BM * F0S;         // [BM][F0S] grad_x0 of the tile
    __bf16* wtl = reinterpret_cast<__bf16*>(g0T + BM * F0S);    // [2][NP][32][WS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    for (int e = threadIdx.x; e < BM * F0; e += NT) {
        const int i = e / BM, r = e - i * BM;
        const int64_t mm = m0 + r;
        x0T[r * F0S + i] = mm < M ? x0[(mm / D) * x0_bs + (int64_t)i * D + (mm % D)] : 0.f;
        g0T[r * F0S + i] = 0.f;
    }
    const int row = wave * 32 + c;
    const int64_t m = m0 + row;          // the column of T^T this lane owns
    const bool mvalid = m < M;
    const int64_t b = mvalid ? m / D : 0;
    const int d = mvalid ? (int)(m % D) : 0;

    // G[m][l] for l = 16 st + 8 s + e as the B operand, kept in registers for the whole tile
    cb_b8 G[LSTEPS][NP];
#pragma unroll
    for (int st = 0; st < LSTEPS; ++st) {
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int l = 16 * st + 8 * s + e;
            float g = 0.f;
            if (mvalid && l < L) {
                const int64_t o = (b * L + l) * D + d;
                g = gy[o] * act_grad_from_y(y[o], act);
            }
            gv[e] = g;
        }
        cb_split<NP>(gv, G[st]);
    }
    // xk values this lane needs: j = jb*32 + (r&3) + 8*(r>>2) + 4*s
    float xkv[JB][16], gxk_acc[JB][16];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * s;
            xkv[jb][r] = (mvalid && j < Hk) ? xk[b * xk_bs + (int64_t)j * D + d] : 0.f;
            gxk_acc[jb][r] = 0.f;
        }
    const int njb = (Hk + 31) / 32;
    const int nchunks = F0 * njb;
    // W chunk (i, jb): NP parts x 32 rows x Lq bf16, contiguous in WN; 16-byte pieces, kWP per thread
    const int pieces = Lq / 8;                          // per row
    constexpr int kWP = (NP * 32 * 2 * LSTEPS + NT - 1) / NT;
    cb_f4 wreg[kWP];
    auto load_w = [&](int chunk) {
        const __bf16* src = WN + (int64_t)chunk * 32 * Lq;
#pragma unroll
        for (int u = 0; u < kWP; ++u) {
            const int e = threadIdx.x + NT * u;
            const int part = e / (32 * pieces), q = e - part * 32 * pieces;
            const int r = q / pieces, pc = q - r * pieces;
            if (e < NP * 32 * pieces) wreg[u] = *reinterpret_cast<const cb_f4*>(src + part * wn_part + r * Lq + 8 * pc);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < kWP; ++u) {
            const int e = threadIdx.x + NT * u;
            const int part = e / (32 * pieces), q = e - part * 32 * pieces;
            const int r = q / pieces, pc = q - r * pieces;
            if (e < NP * 32 * pieces) *reinterpret_cast<cb_f4*>(wtl + ((buf * NP + part) * 32 + r) * WS + 8 * pc) = wreg[u];
        }
    };
    const int lsteps = Lq / 16;
    load_w(0);
    store_w(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const int i = chunk / njb, jb = chunk - i * njb;
        if (chunk + 1 < nchunks) load_w(chunk + 1);
        const __bf16* wrow = wtl + (buf * NP * 32 + c) * WS + 8 * s;
        cb_f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < LSTEPS; ++st)
            if (st < lsteps) {
                cb_b8 a[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const cb_b8*>(wrow + q * 32 * WS + 16 * st);
                cb_mma<NP>(acc, a, G[st]);
            }
        // contract T^T[j, m] (16 j's in this lane) against xk and x0
        const float x0v = x0T[row * F0S + i];
        float p = 0.f;
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
            if (jj == jb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p += xkv[jj][r] * acc[r];
                    gxk_acc[jj][r] += x0v * acc[r];
                }
            }
        p += __shfl_xor(p, 32, 64);
        if (s == 0) g0T[row * F0S + i] += p;  // unique owner of (m, i)
        if (chunk + 1 < nchunks) store_w(buf ^ 1);   // (its last readers, chunk - 1, are behind the previous barrier)
        __syncthreads();
    }
    if (mvalid) {
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (j < Hk) gxk[(b * Hk + j) * D + d] = gxk_acc[jb][r];
            }
    }
    for (int e = threadIdx.x; e < BM * F0; e += NT) {
        const int i = e / BM, r = e - i * BM;
        const int64_t mm = m0 + r;
        if (mm < M) gx0[((mm / D) * F0 + i) * D + (mm % D)] = g0T[r * F0S + i];     // this block is the only writer of (m, i)
    }
}

// ------------------------------------------------------------------------------------------
// wgrad: grad_W[k, l] += sum_m Z[m,k] G[m,l]
// ------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256, 2) void k_cin_wgrad_bf16(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const float* __restrict__ y, const float* __restrict__ gy, int act, int B, int F0, int Hk, int L,
    int D, int64_t rows_per_split, float* __restrict__ gW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = F0 * Hk;
    const int64_t M = (int64_t)B * D;
    constexpr int XS = kBMC + 4, GS = kBMC + 8;
    float* x0s = lds;                    // [F0][XS]  (rows m contiguous)
    float* xks = x0s + F0 * XS;          // [Hk][XS]
    __bf16* gsb = reinterpret_cast<__bf16*>(xks + Hk * XS);     // [NP][kBN][GS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int kbase = blockIdx.x * (128 * kBKT) + wave * (32 * kBKT);
    int ki[kBKT], kj[kBKT];
    bool kvalid[kBKT];
#pragma unroll
    for (int u = 0; u < kBKT; ++u) {
        const int k = kbase + 32 * u + c;   // this lane's A row (i,j) in sub-tile u
        kvalid[u] = k < K;
        ki[u] = kvalid[u] ? k / Hk : 0;
        kj[u] = kvalid[u] ? k % Hk : 0;
    }
    const int n0 = blockIdx.z * kBN;
    const int nblocks_n = min(4, (L - n0 + 31) / 32);
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = min(M, m_begin + rows_per_split);
    cb_f16v acc[kBKT][4];
#pragma unroll
    for (int u = 0; u < kBKT; ++u)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][nb][r] = 0.f;

    // 16-byte staging needs 4 consecutive m inside one batch row (D % 4 == 0; chunk starts are multiples of 64) and aligned
    // addresses
    const bool vec4 = (D % 4 == 0) && (x0_bs % 4 == 0) && (xk_bs % 4 == 0) &&
                      ((((uintptr_t)x0 | (uintptr_t)xk | (uintptr_t)y | (uintptr_t)gy) & 15) == 0);
    for (int64_t mc = m_begin; mc < m_end; mc += kBMC) {
        __syncthreads();
        const int64_t b0 = mc / D;
        const int d0 = (int)(mc - b0 * D);
        const int rows = (int)min((int64_t)kBMC, m_end - mc);
        if (vec4) {
            // as cin.hip's k_cin_wgrad (round 3): 4 consecutive m of a tile row are 4 consecutive d of one (b, row) — one
            // 16-byte load, one 16-byte (x0 / x_k, fp32) or 8-byte (G, bf16) LDS store; a thread's loads of a batch are issued
            // before its first store.  The scalar form below waited for ~70 dependent 4-byte loads per thread and chunk — many
            // times the chunk's bf16 MFMA time.
            constexpr int QR = kBMC / 4;
            constexpr int kStB = 4;
            const int ntask = (F0 + Hk + kBN) * QR;
            for (int t0 = threadIdx.x; t0 < ntask; t0 += 256 * kStB) {
                cb_f4 v[kStB], yv[kStB];
                int dst[kStB];       // >= 0: float offset of a 16-byte fp32 store (x0s | xks are contiguous); <= -2: G, bf16 offset -dst - 2
#pragma unroll
                for (int u = 0; u < kStB; ++u) {
                    const int t = t0 + 256 * u;
                    dst[u] = -1;
                    v[u] = cb_f4{0.f, 0.f, 0.f, 0.f};
                    yv[u] = cb_f4{1.f, 1.f, 1.f, 1.f};
                    if (t >= ntask) continue;
                    const int row = t / QR, q = t - row * QR;
                    dst[u] = row < F0 + Hk ? row * XS + 4 * q : -2 - ((row - F0 - Hk) * GS + 4 * q);
                    if (4 * q >= rows) continue;
                    const int64_t m = mc + 4 * q, b = m / D;
                    const int d = (int)(m - b * D);
                    if (row < F0) v[u] = *reinterpret_cast<const cb_f4*>(x0 + b * x0_bs + (int64_t)row * D + d);
                    else if (row < F0 + Hk) v[u] = *reinterpret_cast<const cb_f4*>(xk + b * xk_bs + (int64_t)(row - F0) * D + d);
                    else if (n0 + row - F0 - Hk < L) {
                        const int64_t o = (b * L + n0 + row - F0 - Hk) * D + d;
                        v[u] = *reinterpret_cast<const cb_f4*>(gy + o);
                        if (act != DT_ACT_LINEAR) yv[u] = *reinterpret_cast<const cb_f4*>(y + o);
                    }
                }
#pragma unroll
                for (int u = 0; u < kStB; ++u) {
                    if (dst[u] == -1) continue;
                    cb_f4 o = v[u];
                    if (dst[u] >= 0) {
                        *reinterpret_cast<cb_f4*>(lds + dst[u]) = o;
                    } else {
                        if (act != DT_ACT_LINEAR) {
                            o.x *= act_grad_from_y(yv[u].x, act); o.y *= act_grad_from_y(yv[u].y, act);
                            o.z *= act_grad_from_y(yv[u].z, act); o.w *= act_grad_from_y(yv[u].w, act);
                        }
                        typedef __bf16 cb_b4 __attribute__((ext_vector_type(4)));
                        float rr[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            cb_b4 h;
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const __bf16 a = (__bf16)rr[e4];
                                h[e4] = a;
                                rr[e4] -= (float)a;
                            }
                            *reinterpret_cast<cb_b4*>(gsb + q * kBN * GS + (-dst[u] - 2)) = h;
                        }
                    }
                }
            }
        } else {
        for (int e = threadIdx.x; e < kBMC * F0; e += 256) {
            const int i = e / kBMC, r = e - i * kBMC;
            const int q = d0 + r, bq = q / D, dq = q - bq * D;
            x0s[i * XS + r] = r < rows ? x0[(b0 + bq) * x0_bs + i * D + dq] : 0.f;
        }
        for (int e = threadIdx.x; e < kBMC * Hk; e += 256) {
            const int j = e / kBMC, r = e - j * kBMC;
            const int q = d0 + r, bq = q / D, dq = q - bq * D;
            xks[j * XS + r] = r < rows ? xk[(b0 + bq) * xk_bs + j * D + dq] : 0.f;
        }
        for (int e = threadIdx.x; e < kBMC * kBN; e += 256) {
            const int l = e / kBMC, r = e - l * kBMC;
            const int q = d0 + r, bq = q / D, dq = q - bq * D;
            float g = 0.f;
            if (r < rows && n0 + l < L) {
                const int64_t o = ((b0 + bq) * L + n0 + l) * D + dq;
                g = gy[o] * act_grad_from_y(y[o], act);
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const __bf16 a = (__bf16)g;
                gsb[q * kBN * GS + l * GS + r] = a;
                g -= (float)a;
            }
        }
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < kBMC / 16; ++st) {
            const int r0 = 16 * st + 8 * s;
            cb_b8 a[kBKT][NP];
#pragma unroll
            for (int u = 0; u < kBKT; ++u) {
#pragma unroll
                for (int q = 0; q < NP; ++q) a[u][q] = cb_zero();
                if (kvalid[u]) {
                    const float* xp = x0s + ki[u] * XS + r0;
                    const float* kp = xks + kj[u] * XS + r0;
                    const cb_f4 lo = *reinterpret_cast<const cb_f4*>(xp) * *reinterpret_cast<const cb_f4*>(kp),
                                hi = *reinterpret_cast<const cb_f4*>(xp + 4) * *reinterpret_cast<const cb_f4*>(kp + 4);
                    const float z[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    cb_split<NP>(z, a[u]);
                }
            }
            const __bf16* grow = gsb + c * GS + r0;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                if (nb < nblocks_n) {
                    cb_b8 g[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) g[q] = *reinterpret_cast<const cb_b8*>(grow + (q * kBN + nb * 32) * GS);
#pragma unroll
                    for (int u = 0; u < kBKT; ++u) cb_mma<NP>(acc[u][nb], a[u], g);
                }
        }
    }
#pragma unroll
    for (int u = 0; u < kBKT; ++u)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb >= nblocks_n) continue;
            const int l = n0 + nb * 32 + c;
            if (l >= L) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kbase + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (kk < K) atomicAdd(&gW[(int64_t)kk * L + l], acc[u][nb][r]);
            }
        }
}

// wgrad, large-batch form (round 4).  The kernel above gives a block 256 rows of k and all 128 filters: each of its 7 k blocks
// reads ALL of G and y again (1.27 GB per layer at the Criteo shape) and the kernel ran at the memory system's pace (372 us,
// 18 % of the matrix rate).  Here a block is EIGHT waves x up to FOUR 32-row k sub-tiles (wave w owns sub-tiles w, w + 8, ..
// of the block's group) x 64 filters: 2 x 2 (k groups x filter halves) blocks per batch slice, 457 MB per layer; the next
// chunk's loads wait in registers while this chunk's MFMAs run (one block per CU: nobody else would cover them).
// Needs the 16-byte staging (D % 4 == 0, aligned pointers) and F0 + Hk <= 96 (three x pieces per thread).
constexpr int kW2Waves = 8, kW2KT = 4, kW2NB = 2;
template <int NP>
__global__ __launch_bounds__(512) void k_cin_wgrad_wide(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const float* __restrict__ y, const float* __restrict__ gy, int act, int B, int F0, int Hk, int L,
    int D, int64_t rows_per_split, int sub_per_group, float* __restrict__ gW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KT = kW2KT, NB = kW2NB, FN = 32 * NB, NT = 64 * kW2Waves;
    const int K = F0 * Hk;
    const int64_t M = (int64_t)B * D;
    constexpr int XS = kBMC + 4, GS = kBMC + 8;
    float* x0s = lds;                    // [F0][XS]  (rows m contiguous)
    float* xks = x0s + F0 * XS;          // [Hk][XS]
    __bf16* gsb = reinterpret_cast<__bf16*>(xks + Hk * XS);     // [NP][FN][GS]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int nsub = (K + 31) >> 5;
    const int sub0 = blockIdx.x * sub_per_group, sub1 = min(nsub, sub0 + sub_per_group);
    int ki[KT], kj[KT], kb[KT];
    bool has[KT], kvalid[KT];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
        const int su = sub0 + wave + kW2Waves * u;
        has[u] = su < sub1;                 // wave-uniform
        kb[u] = 32 * su;
        const int k = kb[u] + c;            // this lane's A row (i,j) in sub-tile u
        kvalid[u] = has[u] && k < K;
        ki[u] = kvalid[u] ? k / Hk : 0;
        kj[u] = kvalid[u] ? k % Hk : 0;
    }
    const int n0 = blockIdx.z * FN;
    const int nblocks_n = min(NB, (L - n0 + 31) / 32);
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = min(M, m_begin + rows_per_split);
    cb_f16v acc[KT][NB];
#pragma unroll
    for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][nb][r] = 0.f;

    // staging tasks: 16 four-row pieces per tile row; x rows (F0 + Hk <= 96: at most 3 per thread), G rows (64: exactly 2)
    constexpr int QR = kBMC / 4, XT = 3, GT = FN * QR / NT;
    static_assert(GT * NT == FN * QR, "G pieces per thread");
    cb_f4 vx[XT], vg[GT], vy[GT];
    const int ntx = (F0 + Hk) * QR;
    auto load_chunk = [&](int64_t mc) {
        const int rows = (int)min((int64_t)kBMC, m_end - mc);
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int t = threadIdx.x + NT * u;
            vx[u] = cb_f4{0.f, 0.f, 0.f, 0.f};
            if (t < ntx) {
                const int row = t / QR, q = t - row * QR;
                if (4 * q < rows) {
                    const int64_t m = mc + 4 * q, b = m / D;
                    const int d = (int)(m - b * D);
                    vx[u] = row < F0 ? *reinterpret_cast<const cb_f4*>(x0 + b * x0_bs + (int64_t)row * D + d)
                                     : *reinterpret_cast<const cb_f4*>(xk + b * xk_bs + (int64_t)(row - F0) * D + d);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GT; ++u) {
            const int t = threadIdx.x + NT * u;
            const int row = t / QR, q = t - row * QR;
            vg[u] = cb_f4{0.f, 0.f, 0.f, 0.f};
            vy[u] = cb_f4{1.f, 1.f, 1.f, 1.f};
            if (4 * q < rows && n0 + row < L) {
                const int64_t m = mc + 4 * q, b = m / D;
                const int d = (int)(m - b * D);
                const int64_t o = (b * L + n0 + row) * D + d;
                vg[u] = *reinterpret_cast<const cb_f4*>(gy + o);
                if (act != DT_ACT_LINEAR) vy[u] = *reinterpret_cast<const cb_f4*>(y + o);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int t = threadIdx.x + NT * u;
            if (t < ntx) {
                const int row = t / QR, q = t - row * QR;
                *reinterpret_cast<cb_f4*>(lds + row * XS + 4 * q) = vx[u];        // x0s | xks are contiguous
            }
        }
#pragma unroll
        for (int u = 0; u < GT; ++u) {
            const int t = threadIdx.x + NT * u;
            const int row = t / QR, q = t - row * QR;
            cb_f4 o = vg[u];
            if (act != DT_ACT_LINEAR) {
                o.x *= act_grad_from_y(vy[u].x, act); o.y *= act_grad_from_y(vy[u].y, act);
                o.z *= act_grad_from_y(vy[u].z, act); o.w *= act_grad_from_y(vy[u].w, act);
            }
            typedef __bf16 cb_b4 __attribute__((ext_vector_type(4)));
            float rr[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int qq = 0; qq < NP; ++qq) {
                cb_b4 h;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const __bf16 a = (__bf16)rr[e4];
                    h[e4] = a;
                    rr[e4] -= (float)a;
                }
                *reinterpret_cast<cb_b4*>(gsb + (qq * FN + row) * GS + 4 * q) = h;
            }
        }
    };
    if (m_begin < m_end) {
        load_chunk(m_begin);
        store_chunk();
    }
    __syncthreads();
    for (int64_t mc = m_begin; mc < m_end; mc += kBMC) {
        const bool more = mc + kBMC < m_end;
        if (more) load_chunk(mc + kBMC);
#pragma unroll
        for (int st = 0; st < kBMC / 16; ++st) {
            const int r0 = 16 * st + 8 * s;
            cb_b8 g[NB][NP];
            const __bf16* grow = gsb + c * GS + r0;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < NP; ++q) g[nb][q] = *reinterpret_cast<const cb_b8*>(grow + (q * FN + nb * 32) * GS);
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                if (!has[u]) continue;
                cb_b8 a[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) a[q] = cb_zero();
                if (kvalid[u]) {
                    const float* xp = x0s + ki[u] * XS + r0;
                    const float* kp = xks + kj[u] * XS + r0;
                    const cb_f4 lo = *reinterpret_cast<const cb_f4*>(xp) * *reinterpret_cast<const cb_f4*>(kp),
                                hi = *reinterpret_cast<const cb_f4*>(xp + 4) * *reinterpret_cast<const cb_f4*>(kp + 4);
                    const float z[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    cb_split<NP>(z, a);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    if (nb < nblocks_n) cb_mma<NP>(acc[u][nb], a, g[nb]);
            }
        }
        __syncthreads();                 // every wave is done with this chunk's tiles
        if (more) store_chunk();
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!has[u] || nb >= nblocks_n) continue;
            const int l = n0 + nb * 32 + c;
            if (l >= L) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kb[u] + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (kk < K) atomicAdd(&gW[(int64_t)kk * L + l], acc[u][nb][r]);
            }
        }
}

// grad_bias[l] += sum_{b,d} G[b,l,d]   (fp32, as cin.hip)
__global__ __launch_bounds__(256) void k_cin_bias_grad_b(const float* __restrict__ y, const float* __restrict__ gy,
                                                         int act, int B, int L, int D, float* __restrict__ gbias) {
    const int l = blockIdx.x;
    float sacc = 0.f;
    const int64_t n = (int64_t)B * D;
    for (int64_t e = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.y * blockDim.x) {
        const int64_t b = e / D;
        const int d = (int)(e % D);
        const int64_t o = (b * L + l) * D + d;
        sacc += gy[o] * act_grad_from_y(y[o], act);
    }
    sacc = wave_sum(sacc);
    if ((threadIdx.x & 63) == 0) atomicAdd(&gbias[l], sacc);
}

}  // namespace dt

using namespace dt;

static int cinb_check(const char* who, int B, int F0, int Hk, int L, int D) {
    DT_REQUIRE(B >= 0 && F0 > 0 && Hk > 0 && L > 0 && D > 0, "%s: bad sizes B=%d F0=%d Hk=%d L=%d D=%d", who, B, F0, Hk, L, D);
    DT_UNSUPPORTED(L > 256 || Hk > 128 || F0 > 128, "%s: the bf16 mode takes L <= 256, Hk <= 128, F0 <= 128 (L=%d Hk=%d F0=%d)",
                   who, L, Hk, F0);
    return DT_OK;
}

// workspace: NPF parts of WT [Lp][Kp] + NPB parts of WN [F0*njb*32][Lq], bf16
static int64_t cinb_nT(int F0, int Hk, int L) { return ((int64_t)(L + kBN - 1) / kBN * kBN) * ((int64_t)F0 * cb_hp(Hk)); }
static int64_t cinb_nN(int F0, int Hk, int L) { return (int64_t)F0 * ((Hk + 31) / 32) * 32 * cb_lq(L); }
static int64_t cinb_ws_bytes(int F0, int Hk, int L, int npf, int npb) {
    if (F0 <= 0 || Hk <= 0 || L <= 0) return 0;
    return 2 * (npf * cinb_nT(F0, Hk, L) + npb * cinb_nN(F0, Hk, L)) + 64;
}
extern "C" int64_t dt_cin_bf16_workspace_bytes(int F0, int Hk, int L) { return cinb_ws_bytes(F0, Hk, L, 1, 1); }
extern "C" int64_t dt_cin_bf16x3_workspace_bytes(int F0, int Hk, int L) { return cinb_ws_bytes(F0, Hk, L, 3, 2); }

// the filter's bf16 parts: WT (forward, NPF parts) at the start of ws, WN (dgrad, NPB parts) behind it; a NULL result is not packed
template <int NP>
static void cinb_pack(const float* W, int F0, int Hk, int L, __bf16* WT, __bf16* WN, hipStream_t st) {
    hipLaunchKernelGGL(k_cin_pack_w<NP>, dim3(512), dim3(256), 0, st, W, F0, Hk, L, WT, WN);
}

// large batches take the 256-row (eight-wave) blocks of the forward / dgrad: at least one block per CU that way
// (round 4's A/B, profiles/r04_xdeepfm_x3_narrow_kernel_stats.csv: 2.49 ms per step against 3.27 ms with the four-wave kernels)
static bool cinb_wide(int64_t M) { return M >= 256 * 128; }

template <int NP>
static int cinb_fwd(const char* who, const float* x0, const float* xk, const float* W, const float* bias, int act, int B, int F0,
                    int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* y, void* ws, void* stream) {
    int rc = cinb_check(who, B, F0, Hk, L, D);
    if (rc) return rc;
    if (B == 0) return DT_OK;
    DT_REQUIRE(x0 && xk && W && y && ws, "%s: null pointer", who);
    DT_REQUIRE(act >= 0 && act < DT_ACT_COUNT, "%s: act %d", who, act);
    hipStream_t st = as_stream(stream);
    __bf16* WT = reinterpret_cast<__bf16*>(ws);
    cinb_pack<NP>(W, F0, Hk, L, WT, nullptr, st);
    const int64_t M = (int64_t)B * D;
    dim3 grid((unsigned)((M + kBM - 1) / kBM), (unsigned)ceil_div(L, kBN));
    if (NP == 3) {
        // the forward that never forms Z (k_cin_fwd_noz): Hk up to 128
        const int ks = cb_hp(Hk) <= 32 ? 2 : cb_hp(Hk) <= 64 ? 4 : 8;
        const bool wide = cinb_wide(M);                  // 256-row blocks (eight waves), the filter double buffered in LDS
        const int bm = wide ? 256 : kBM;
        const size_t ldsn = (size_t)F0 * bm * sizeof(float) + (size_t)(wide ? 2 : 1) * 3 * kBN * (16 * ks + 8) * 2;
        const bool any = !(act == DT_ACT_LINEAR || act == DT_ACT_RELU);
        const dim3 gridn((unsigned)((M + bm - 1) / bm), (unsigned)ceil_div(L, kBN));
#define DT_NOZ(ANY, KSV)                                                                                                   \
    do {                                                                                                                   \
        if (wide) {                                                                                                        \
            hipFuncSetAttribute((const void*)k_cin_fwd_noz<ANY, KSV, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn); \
            hipLaunchKernelGGL((k_cin_fwd_noz<ANY, KSV, 8>), gridn, dim3(512), ldsn, st, x0, x0_bstride, xk, xk_bstride, WT, \
                               cinb_nT(F0, Hk, L), bias, act, B, F0, Hk, L, D, y);                                         \
        } else {                                                                                                           \
            hipFuncSetAttribute((const void*)k_cin_fwd_noz4<ANY, KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn); \
            hipLaunchKernelGGL((k_cin_fwd_noz4<ANY, KSV>), gridn, dim3(256), ldsn, st, x0, x0_bstride, xk, xk_bstride, WT, \
                               cinb_nT(F0, Hk, L), bias, act, B, F0, Hk, L, D, y);                                         \
        }                                                                                                                  \
    } while (0)
        // linear / relu epilogues and Hk <= 64 only: with the libm activations or eight k steps of A parts in registers the
        // kernel spills (254 VGPRs at four steps already) — those shapes keep the Z-forming kernel
        if (ldsn <= 160 * 1024 && !any && ks <= 4) {
            if (ks == 2) DT_NOZ(false, 2); else DT_NOZ(false, 4);
            return launch_status(who);
        }
#undef DT_NOZ
    }
    constexpr int KCH = NP == 1 ? kBK : 16;
    const size_t lds = ((size_t)kBM * ((F0 | 1) + cb_hp(Hk) + 4)) * sizeof(float) + (size_t)2 * NP * kBN * (KCH + 8) * 2;
    DT_UNSUPPORTED(lds > 160 * 1024, "%s: tiles need %zu B of LDS (> 160 KiB)", who, lds);
    if (act == DT_ACT_LINEAR || act == DT_ACT_RELU) {
        hipFuncSetAttribute((const void*)k_cin_fwd_bf16<false, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cin_fwd_bf16<false, NP>), grid, dim3(256), lds, st, x0, x0_bstride, xk, xk_bstride, WT,
                           cinb_nT(F0, Hk, L), bias, act, B, F0, Hk, L, D, y);
    } else {
        hipFuncSetAttribute((const void*)k_cin_fwd_bf16<true, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cin_fwd_bf16<true, NP>), grid, dim3(256), lds, st, x0, x0_bstride, xk, xk_bstride, WT,
                           cinb_nT(F0, Hk, L), bias, act, B, F0, Hk, L, D, y);
    }
    return launch_status(who);
}

extern "C" int dt_cin_layer_fwd_bf16(const float* x0, const float* xk, const float* W, const float* bias, int act, int B,
                                     int F0, int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* y,
                                     void* ws, void* stream) {
    return cinb_fwd<1>("dt_cin_layer_fwd_bf16", x0, xk, W, bias, act, B, F0, Hk, L, D, x0_bstride, xk_bstride, y, ws, stream);
}
extern "C" int dt_cin_layer_fwd_bf16x3(const float* x0, const float* xk, const float* W, const float* bias, int act, int B,
                                       int F0, int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* y,
                                       void* ws, void* stream) {
    return cinb_fwd<3>("dt_cin_layer_fwd_bf16x3", x0, xk, W, bias, act, B, F0, Hk, L, D, x0_bstride, xk_bstride, y, ws, stream);
}

// the eight-wave dgrad exists for L <= 128, Hk <= 64 (G of 256 filters or four j blocks of x_k values per lane spill at
// two waves per SIMD)
template <int LSTEPS, int JB>
constexpr bool dgrad_roomy() { return LSTEPS == 8 && JB <= 2; }
template <int LSTEPS, int NP, int WV>
static size_t dgrad_lds(int F0) {
    return (size_t)2 * 32 * WV * (F0 | 1) * sizeof(float) + (size_t)2 * NP * 32 * (16 * LSTEPS + 8) * 2;
}
template <int LSTEPS, int JB, int NP, int WV>
static int launch_dgrad_b(const float* x0, int64_t x0_bs, const float* xk, int64_t xk_bs, const __bf16* WN, int64_t wn_part,
                          const float* y, const float* gy, int act, int B, int F0, int Hk, int L, int D, float* gx0, float* gxk,
                          hipStream_t st) {
    constexpr int BM = 32 * WV;
    const size_t lds = dgrad_lds<LSTEPS, NP, WV>(F0);
    if (lds > 160 * 1024) {
        set_error("dt_cin_layer_bwd_bf16: tiles need %zu B of LDS (> 160 KiB)", lds);
        return DT_ERR_UNSUPPORTED;
    }
    const int64_t M = (int64_t)B * D;
    const dim3 grid((unsigned)((M + BM - 1) / BM));
    if constexpr (WV == 8) {
        hipFuncSetAttribute((const void*)k_cin_dgrad_bf16<LSTEPS, JB, NP, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cin_dgrad_bf16<LSTEPS, JB, NP, 8>), grid, dim3(512), lds, st, x0, x0_bs, xk, xk_bs, WN, wn_part, y,
                           gy, act, B, F0, Hk, L, D, gx0, gxk);
    } else {
        hipFuncSetAttribute((const void*)k_cin_dgrad_bf16_4<LSTEPS, JB, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_cin_dgrad_bf16_4<LSTEPS, JB, NP>), grid, dim3(256), lds, st, x0, x0_bs, xk, xk_bs, WN, wn_part, y, gy,
                           act, B, F0, Hk, L, D, gx0, gxk);
    }
    return launch_status("dt_cin_layer_bwd_bf16(dgrad)");
}

// NPF: parts the workspace holds for the forward layout (the backward's WN parts start behind them)
template <int NP, int NPF>
static int cinb_bwd(const char* who, const float* x0, const float* xk, const float* W, const float* y, const float* grad_y,
                    int act, int B, int F0, int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride, float* grad_x0,
                    float* grad_xk, float* grad_W, float* grad_bias, void* ws, void* stream) {
    int rc = cinb_check(who, B, F0, Hk, L, D);
    if (rc) return rc;
    if (B == 0) return DT_OK;
    DT_REQUIRE(x0 && xk && W && y && grad_y && grad_x0 && grad_xk && grad_W && ws, "%s: null pointer", who);
    hipStream_t st = as_stream(stream);
    __bf16* WN = reinterpret_cast<__bf16*>(ws) + ((NPF * cinb_nT(F0, Hk, L) + 7) & ~(int64_t)7);
    cinb_pack<NP>(W, F0, Hk, L, nullptr, WN, st);
    const int64_t wn_part = cinb_nN(F0, Hk, L);
    const int jb = ceil_div(Hk, 32);
    // 256-row blocks when the batch fills the chip with them and their tiles fit the LDS
    const bool big = cinb_wide((int64_t)B * D);
#define DT_DGRAD_B(LS, JBV)                                                                                          \
    do {                                                                                                             \
        if constexpr (dgrad_roomy<LS, JBV>()) {                                                                      \
            if (big && dgrad_lds<LS, NP, 8>(F0) <= 160 * 1024) {                                                     \
                rc = launch_dgrad_b<LS, JBV, NP, 8>(x0, x0_bstride, xk, xk_bstride, WN, wn_part, y, grad_y, act, B, F0, Hk, L, \
                                                    D, grad_x0, grad_xk, st);                                        \
                break;                                                                                               \
            }                                                                                                        \
        }                                                                                                            \
        rc = launch_dgrad_b<LS, JBV, NP, 4>(x0, x0_bstride, xk, xk_bstride, WN, wn_part, y, grad_y, act, B, F0, Hk, L, D, \
                                            grad_x0, grad_xk, st);                                                   \
    } while (0)
    if (L <= 128) {
        if (jb <= 1) DT_DGRAD_B(8, 1);
        else if (jb <= 2) DT_DGRAD_B(8, 2);
        else DT_DGRAD_B(8, 4);
    } else {
        if (jb <= 1) DT_DGRAD_B(16, 1);
        else if (jb <= 2) DT_DGRAD_B(16, 2);
        else DT_DGRAD_B(16, 4);
    }
#undef DT_DGRAD_B
    if (rc) return rc;
    const int K = F0 * Hk;
    const int64_t M = (int64_t)B * D;
    const bool vec4 = (D % 4 == 0) && (x0_bstride % 4 == 0) && (xk_bstride % 4 == 0) &&
                      ((((uintptr_t)x0 | (uintptr_t)xk | (uintptr_t)y | (uintptr_t)grad_y) & 15) == 0);
    if (vec4 && F0 + Hk <= 96) {
        // k_cin_wgrad_wide: groups of up to 32 k sub-tiles x 64 filters per block, the batch split over what is left of ~256 blocks
        const int nsub = ceil_div(K, 32), kgroups = ceil_div(nsub, kW2Waves * kW2KT), spg = ceil_div(nsub, kgroups);
        const int lgroups = ceil_div(L, 32 * kW2NB);
        int splits = 256 / (kgroups * lgroups);
        if (splits < 1) splits = 1;
        int64_t rps = (M + splits - 1) / splits;
        rps = (rps + kBMC - 1) / kBMC * kBMC;
        splits = (int)((M + rps - 1) / rps);
        const size_t lds = (size_t)(F0 + Hk) * (kBMC + 4) * sizeof(float) + (size_t)NP * 32 * kW2NB * (kBMC + 8) * 2;
        hipFuncSetAttribute((const void*)k_cin_wgrad_wide<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_cin_wgrad_wide<NP>, dim3(kgroups, splits, lgroups), dim3(512), lds, st, x0, x0_bstride, xk,
                           xk_bstride, y, grad_y, act, B, F0, Hk, L, D, rps, spg, grad_W);
    } else {
        const int kblocks = ceil_div(K, 128 * kBKT), nblocks = ceil_div(L, kBN);
        int splits = 512 / (kblocks * nblocks);
        if (splits < 1) splits = 1;
        int64_t rps = (M + splits - 1) / splits;
        rps = (rps + kBMC - 1) / kBMC * kBMC;
        splits = (int)((M + rps - 1) / rps);
        const size_t lds = (size_t)(F0 + Hk) * (kBMC + 4) * sizeof(float) + (size_t)NP * kBN * (kBMC + 8) * 2;
        hipFuncSetAttribute((const void*)k_cin_wgrad_bf16<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_cin_wgrad_bf16<NP>, dim3(kblocks, splits, nblocks), dim3(256), lds, st, x0, x0_bstride, xk, xk_bstride,
                           y, grad_y, act, B, F0, Hk, L, D, rps, grad_W);
    }
    if (grad_bias)
        hipLaunchKernelGGL(k_cin_bias_grad_b, dim3(L, 16), dim3(256), 0, st, y, grad_y, act, B, L, D, grad_bias);
    return launch_status(who);
}

extern "C" int dt_cin_layer_bwd_bf16(const float* x0, const float* xk, const float* W, const float* y, const float* grad_y,
                                     int act, int B, int F0, int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride,
                                     float* grad_x0, float* grad_xk, float* grad_W, float* grad_bias, void* ws,
                                     void* stream) {
    return cinb_bwd<1, 1>("dt_cin_layer_bwd_bf16", x0, xk, W, y, grad_y, act, B, F0, Hk, L, D, x0_bstride, xk_bstride, grad_x0,
                          grad_xk, grad_W, grad_bias, ws, stream);
}
extern "C" int dt_cin_layer_bwd_bf16x3(const float* x0, const float* xk, const float* W, const float* y, const float* grad_y,
                                       int act, int B, int F0, int Hk, int L, int D, int64_t x0_bstride, int64_t xk_bstride,
                                       float* grad_x0, float* grad_xk, float* grad_W, float* grad_bias, void* ws,
                                       void* stream) {
    return cinb_bwd<2, 3>("dt_cin_layer_bwd_bf16x3", x0, xk, W, y, grad_y, act, B, F0, Hk, L, D, x0_bstride, xk_bstride, grad_x0,
                          grad_xk, grad_W, grad_bias, ws, stream);
}
