// autoint.hip — the AutoInt interacting layer (MultiheadAttention.call, deeptables/models/layers.py:119-153) as ONE
// forward and ONE backward kernel on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
//
//   Q,K,V,R = relu(x W_* + b_*)                      layers.py:104-108, :123-127   (four Keras Dense layers)
//   per head h:  P = softmax(Q_h K_h^T / sqrt(d_h));  P = dropout(P);  O_h = P V_h      layers.py:129-143
//   a = relu(concat_h(O_h) + R)                      layers.py:145-150   (BatchNormalization :151 stays bn.hip)
//
// One WAVE owns one batch row at a time (F <= 32 fields padded to two 16-row MFMA tiles); nothing is exchanged
// between waves, so there is no workgroup barrier anywhere.  Per row:
//   * the projections run as [32 x D] . [D x 4D] with the weights held in registers for the wave's whole life
//     (D^2/16 registers per lane) and x read straight from HBM as the A operand (one 128-byte row per four lanes:
//     the contraction index is permuted so that a lane's k are contiguous — k = (D/4) q + t);
//   * Q | K | V | R land in the wave's private LDS slab [32][4D+4]: they never reach HBM (round 1 wrote and
//     re-read 109 MB per layer and direction);
//   * scores are formed TRANSPOSED, S^T[j][i] with the key index j on the accumulator registers and the query i on
//     the lanes: the softmax over keys is then a lane-local loop plus two shuffles, and the probability registers
//     ARE the A operand of P.V (step (J,r) <-> key 16J + 4q + r) — P never leaves registers;
//   * the backward recomputes Q, K, V, R and the probabilities (both orientations: queries-on-lanes feeds dQ,
//     keys-on-lanes feeds dV and dK), and leaves the gradient of the four pre-activation projections [B,F,4D] for
//     the Dense backward (dt_dense_bwd: grad_x = dY Wcat^T, grad_W = x^T dY).
// Attention-weight dropout (layers.py:141): keep-mask from a counter hash of (seed, row, head, query, key), the
// same function in both kernels and in deeptables_amd/ops.py (tests rebuild the mask from it).
#include "common.h"

namespace dt {

typedef float ai_f4 __attribute__((ext_vector_type(4)));
constexpr int kAiPad = 4;

__host__ __device__ inline unsigned ai_hash(unsigned seed, unsigned b, unsigned h, unsigned i, unsigned j) {
    unsigned x = seed ^ (b * 0x9E3779B1u) ^ ((h * 64u * 64u + i * 64u + j) * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// keep-probability threshold: keep iff hash >= thr, thr = rate * 2^32
__host__ __device__ inline float ai_keep(unsigned seed, unsigned thr, unsigned b, unsigned h, unsigned i, unsigned j,
                                         float inv_keep) {
    return ai_hash(seed, b, h, i, j) >= thr ? inv_keep : 0.f;
}

__device__ __forceinline__ void ai_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the kernels / biases of dense_Q | dense_K | dense_V [| dense_residual]: separate Keras variables ([D,D] and [D] each);
// "Wcat[k][m]" below = W[m / D][k][m % D]
struct AiW {
    const float* W[4];
    const float* b[4];
};
// optional BatchNormalization backward folded into the load of the incoming gradient (layers.py:151: the layer's output
// is BN(a)): g_a = c1 (g_y - c2 - (a - mean) c3) with c1 = gamma rstd, c2 = sum_g / N, c3 = rstd sum_gx / N
struct AiBn {
    const float *gamma, *mean, *rstd, *sum_g, *sum_gx;   // gamma may be NULL (scale = False); mean == NULL: no BN
    float inv_n;
    const double* sums64;                                // != NULL: sum_g | sum_gx as the doubles an AiPrev pass left ([2 D])
};
// optional: the BatchNormalization-backward batch sums of the layer BELOW (whose output y = BN(a_prev) is this layer's input x)
// formed here, while this layer's dX — the gradient w.r.t. y — leaves: sum_b dX and sum_b dX xhat_prev per channel, added as
// doubles into sums [2 D] (zeroed by the caller).  Replaces that layer's dt_bn_train_bwd_stats pass over a_prev and dX (two
// launches, 54 MB read at the Criteo shape) by one more read of a_prev rows in this kernel's epilogue.
struct AiPrev {
    const float *a, *mean, *rstd;       // a == NULL: off
    double* sums;                        // [D] sum_g | [D] sum_gx
};
// optional: the layer's input is handed over UN-normalised — x = a_prev, the output of the interacting layer below BEFORE its
// BatchNormalization — and normalised while it is loaded: x_used = s x + t per channel, s = rstd gamma, t = beta - mean s
// (one fma per element; within two roundings of bn.hip k_bn_apply's (x - mean) rstd gamma + beta, and the SAME values in the
// forward and in the backward's recomputation, which is what the relu masks need).  The normalised tensor y_prev is then never
// written or read (27 MB each way per layer at the Criteo shape) and the layer below runs without its apply pass.  The
// constants sit in the block's LDS (xc = [D] s | [D] t), not in registers.
struct AiXn {
    const float *mean, *rstd, *gamma, *beta;     // mean == NULL: off; gamma / beta may be NULL (1 / 0)
};
template <int D>
__device__ __forceinline__ void ai_xn_fill(const AiXn& xn, float* xc) {
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float sc = xn.rstd[c] * (xn.gamma ? xn.gamma[c] : 1.f);
        xc[c] = sc;
        xc[D + c] = (xn.beta ? xn.beta[c] : 0.f) - xn.mean[c] * sc;
    }
}
// the A operand of the projections (ai_load_x's layout: xa[T][t] = channel (D/4) q + t), normalised in place
template <int D>
__device__ __forceinline__ void ai_xn_rows(const float* xc, int q, float (&xa)[2][D / 4]) {
#pragma unroll
    for (int t = 0; t < D / 4; t += 4) {
        const int c = (D / 4) * q + t;
        const ai_f4 sc = *reinterpret_cast<const ai_f4*>(xc + c), sh = *reinterpret_cast<const ai_f4*>(xc + D + c);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[T][t + e] = fmaf(xa[T][t + e], sc[e], sh[e]);
    }
}
template <int D>
__device__ __forceinline__ float ai_w(const AiW& w, int k, int m) { return w.W[m / D][k * D + (m % D)]; }
template <int D>
__device__ __forceinline__ float ai_b(const AiW& w, int m) { return w.b[m / D][m % D]; }

// reductions over the four lanes l, l^16, l^32, l^48 (the q groups of a 16x16x4 accumulator column) without LDS:
// v_permlane16_swap exchanges odd 16-lane rows of one register with even rows of another, v_permlane32_swap the halves
__device__ __forceinline__ float ai_qsum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float ai_qmax(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int D, int DH>
struct AiCfg {
    static constexpr int H = D / DH;
    static constexpr int TK = D / 4;        // MFMA steps of a projection (k = TK q + t)
    static constexpr int SK = DH / 4;       // MFMA steps of a score tile  (d = SK q + t)
    static constexpr int YS = 4 * D + kAiPad;
};

// x rows of batch row b as the A operand of both row tiles: xa[T][t] = x[b][16T + n][TK q + t]  (rows >= F: row F-1)
template <int D>
__device__ __forceinline__ void ai_load_x(const float* __restrict__ x, int64_t b, int F, int n, int q, float (&xa)[2][D / 4]) {
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int row = min(16 * T + n, F - 1);
        const float* p = x + ((int64_t)b * F + row) * D + (D / 4) * q;
#pragma unroll
        for (int t = 0; t < D / 4; t += 4) {
            const ai_f4 v = *reinterpret_cast<const ai_f4*>(p + t);
            xa[T][t] = v.x; xa[T][t + 1] = v.y; xa[T][t + 2] = v.z; xa[T][t + 3] = v.w;
        }
    }
}

// ---- north_star's "1e-2 bf16" mode of the layer (mfma_mode = DT_AI_BF16; D = 32 only) ------------------------------------
// The three projection-shaped products of the layer — Y = x Wcat (forward, and its recomputation in the backward), dX = dY
// Wcat^T and the weight gradient x^T dY — run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 70 % of the layer's matrix
// work at 1/16 of the fp32-MFMA time per product.  The two backward products take plain bf16 operands (2^-9 per operand).
// The FORWARD product takes two-part operands (a = hi + lo, three products hi hi + hi lo + lo hi: 2^-17) — measured first
// with plain operands: outputs 3e-3, but ~0.3 % of the relu units of the four projections sat within that noise of zero and
// took the other derivative: gradients 4-36 % off in the largest entry, 8 % in L2, logits 1.6e-2 after three layers.  With
// fp32-class pre-activations the relu decisions are the oracle's and only the backward's own rounding is left.
// mfma_mode = DT_AI_BF16X2 (split-bf16, the construction of tower_x3.h / cin_bf16.hip for this layer): the forward product with
// THREE-part operands (all 24 mantissa bits, six products: fp32-class pre-activations — with two parts the logits of the
// three-layer AutoInt graph were 8e-5 off and relu units flipped: dense gradients 7e-3), the two backward products with two
// parts (2^-17, gradients ~1e-5 of the tensor max): the exact kernels' parity bars on the bf16 matrix cores.  With D = 32 the operand registers the fp32 kernels
// already hold ARE the 16x16x32 layout: a lane (n, q) owns the 8 contraction indices k = 8 q + t of row / column n — eight
// floats become one bf16x8 operand, one MFMA replaces eight.  The score / probability products (d_h = 8: two steps of the
// fp32 MFMA) stay exact fp32, and so do softmax, relu masks and BatchNormalization: results within 1e-2 of the float64
// oracle (of each tensor's largest entry), tests/test_autoint_gpu.py.
typedef __bf16 ai_b8 __attribute__((ext_vector_type(8)));
#define AI_MFMA_BF16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)
__device__ __forceinline__ ai_b8 ai_pack8(const float (&v)[8]) {
    ai_b8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    return o;
}
// v = hi + lo with 16 mantissa bits kept (lo = the bf16 rounding of what hi left)
__device__ __forceinline__ void ai_split8(const float (&v)[8], ai_b8& hi, ai_b8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}
// v = hi + mid + lo EXACTLY (8 + 8 + 8 mantissa bits)
__device__ __forceinline__ void ai_split8_3(const float (&v)[8], ai_b8& hi, ai_b8& mid, ai_b8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];
        const float r1 = v[e] - (float)h;
        const __bf16 m = (__bf16)r1;
        hi[e] = h; mid[e] = m;
        lo[e] = (__bf16)(r1 - (float)m);
    }
}
// Y tile += x W with split operands: PARTS = 2 (16 bits: hi hi + hi lo + lo hi, 2^-17) or 3 (all 24 bits: six products, the
// dropped terms 2^-24 of the product — fp32-class pre-activations, the oracle's relu decisions)
template <int PARTS>
__device__ __forceinline__ void ai_split_mma(const float (&x0)[8], const float (&x1)[8], const float (&w)[8], ai_f4& c0, ai_f4& c1) {
    ai_f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;               // the small products, added last
    if constexpr (PARTS == 2) {
        ai_b8 xh0, xl0, xh1, xl1, wh, wlo;
        ai_split8(x0, xh0, xl0); ai_split8(x1, xh1, xl1); ai_split8(w, wh, wlo);
        AI_MFMA_BF16(s0, xh0, wlo); AI_MFMA_BF16(s1, xh1, wlo);
        AI_MFMA_BF16(s0, xl0, wh); AI_MFMA_BF16(s1, xl1, wh);
        AI_MFMA_BF16(c0, xh0, wh); AI_MFMA_BF16(c1, xh1, wh);
        c0 += s0; c1 += s1;
    } else {
        ai_b8 xh0, xm0, xl0, xh1, xm1, xl1, wh, wm, wlo;
        ai_split8_3(x0, xh0, xm0, xl0); ai_split8_3(x1, xh1, xm1, xl1); ai_split8_3(w, wh, wm, wlo);
        ai_f4 t0 = s0, t1 = s0;
        AI_MFMA_BF16(t0, xh0, wlo); AI_MFMA_BF16(t1, xh1, wlo);
        AI_MFMA_BF16(t0, xm0, wm); AI_MFMA_BF16(t1, xm1, wm);
        AI_MFMA_BF16(t0, xl0, wh); AI_MFMA_BF16(t1, xl1, wh);
        AI_MFMA_BF16(s0, xh0, wm); AI_MFMA_BF16(s1, xh1, wm);
        AI_MFMA_BF16(s0, xm0, wh); AI_MFMA_BF16(s1, xm1, wh);
        AI_MFMA_BF16(c0, xh0, wh); AI_MFMA_BF16(c1, xh1, wh);
        c0 += s0 + t0; c1 += s1 + t1;
    }
}
__device__ __forceinline__ ai_b8 ai_pack8(const ai_f4& a, const ai_f4& b) {
    ai_b8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)a[e]; o[4 + e] = (__bf16)b[e]; }
    return o;
}

// Y = relu(x Wcat + b) -> the wave's LDS slab ys[32][YS]   (NP = 3 or 4 projections: q | k | v [| residual])
template <int D, int BF = 0>
__device__ __forceinline__ void ai_project(const float (&xa)[2][D / 4], const float (&wr)[D / 4][D / 4],
                                           const float (&br)[D / 4], int NP, float* ys, int n, int q) {
    constexpr int TK = D / 4, YS = 4 * D + kAiPad;
    if constexpr (BF != 0) static_assert(D == 32, "bf16 modes: D = 32");
#pragma unroll
    for (int ct = 0; ct < D / 4; ++ct) {                  // 4D / 16 column tiles
        if (ct * 16 >= NP * D) break;
        ai_f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
        if constexpr (BF != 0) {
            // (the splits of x repeat per column tile: the compiler keeps them in registers across the unrolled loop)
            ai_split_mma<BF == 2 ? 3 : 2>(xa[0], xa[1], wr[ct], c0, c1);
        } else {
#pragma unroll
        for (int t = 0; t < TK; ++t) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][t], wr[ct][t], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1][t], wr[ct][t], c1, 0, 0, 0);
        }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ys[(4 * q + r) * YS + 16 * ct + n] = fmaxf(c0[r] + br[ct], 0.f);
            ys[(16 + 4 * q + r) * YS + 16 * ct + n] = fmaxf(c1[r] + br[ct], 0.f);
        }
    }
}

// the same with the weights read from the block's LDS copy wl[D][WS] (B operand of step t: W[TK q + t][16 ct + n])
template <int D, int BF = 0>
__device__ __forceinline__ void ai_project_lds(const float (&xa)[2][D / 4], const float* wl, int WS,
                                               const float (&br)[D / 4], int NP, float* ys, int n, int q) {
    constexpr int TK = D / 4, YS = 4 * D + kAiPad;
    if constexpr (BF != 0) static_assert(D == 32, "bf16 modes: D = 32");
#pragma unroll
    for (int ct = 0; ct < D / 4; ++ct) {
        if (ct * 16 >= NP * D) break;
        float wv[TK];
#pragma unroll
        for (int t = 0; t < TK; ++t) wv[t] = wl[(TK * q + t) * WS + 16 * ct + n];
        ai_f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
        if constexpr (BF != 0) {                             // the forward's products, operation for operation (same relu masks)
            ai_split_mma<BF == 2 ? 3 : 2>(xa[0], xa[1], wv, c0, c1);
        } else {
#pragma unroll
        for (int t = 0; t < TK; ++t) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][t], wv[t], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1][t], wv[t], c1, 0, 0, 0);
        }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ys[(4 * q + r) * YS + 16 * ct + n] = fmaxf(c0[r] + br[ct], 0.f);
            ys[(16 + 4 * q + r) * YS + 16 * ct + n] = fmaxf(c1[r] + br[ct], 0.f);
        }
    }
}

// S^T tiles of head h in the queries-on-lanes orientation: st[J][I][r] = sum_d A[16J + 4q + r][d] B[16I + n][d]
// (A = K, B = Q for the scores; A = V, B = dO for dP).  acol / bcol: first column of the head inside the slab.
template <int DH>
__device__ __forceinline__ void ai_tiles(const float* ys, int YS, int acol, int bcol, int n, int q, ai_f4 (&st)[2][2]) {
    constexpr int SK = DH / 4;
    float av[2][SK], bv[2][SK];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int t = 0; t < SK; ++t) {
            av[T][t] = ys[(16 * T + n) * YS + acol + SK * q + t];
            bv[T][t] = ys[(16 * T + n) * YS + bcol + SK * q + t];
        }
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            ai_f4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < SK; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[J][t], bv[I][t], c, 0, 0, 0);
            st[J][I] = c;
        }
}

// out[I][r] (row 16I + 4q + r, column n) = sum over the 32 "register-side" indices of p[J][I][r'] * M[16J + 4q + r'][col0 + (n % DH)]
template <int DH>
__device__ __forceinline__ void ai_apply(const ai_f4 (&p)[2][2], const float* ys, int YS, int col0, int n, int q,
                                         ai_f4 (&out)[2]) {
    float bv[2][4];
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[J][r] = ys[(16 * J + 4 * q + r) * YS + col0 + (n & (DH - 1))];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        ai_f4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(p[J][I][r], bv[J][r], c, 0, 0, 0);
        out[I] = c;
    }
}

template <int D, int DH, bool DROP, int BF = 0>
__global__ __launch_bounds__(256) void k_autoint_fwd(const float* __restrict__ x, AiW w4, int B, int F, int NP,
                                                     float* __restrict__ out_a, float* __restrict__ lse_out,
                                                     unsigned drop_thr, float inv_keep, unsigned seed,
                                                     const float* __restrict__ bn_shift, float* __restrict__ bn_part,
                                                     AiXn xn) {
    using C = AiCfg<D, DH>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    float* ys = lds + wave * 32 * C::YS;
    const float* xc = lds + 4 * 32 * C::YS;                 // [2 D] input-normalisation constants (AiXn)
    if (xn.mean) {
        ai_xn_fill<D>(xn, lds + 4 * 32 * C::YS);
        __syncthreads();
    }
    // BatchNormalization statistics of the layer's output (layers.py:151) ride along (bn_part != NULL): every lane owns the
    // float4 column lane % (D/4) of all the rows it stores and keeps sum / sum of squares of (a - K), K = bn_shift (the
    // moving mean: near the batch mean after a few steps, and the same for every block, so partials add up without a
    // merge formula).  bn_part = [D] copy of K | [gridDim.x][2][D] block sums; dt_autoint_fwd_bn's second launch finishes.
    ai_f4 bnK = {0.f, 0.f, 0.f, 0.f}, bnS = bnK, bnQ = bnK;
    if (bn_part && bn_shift) {
        bnK = *reinterpret_cast<const ai_f4*>(bn_shift + 4 * (lane % (D / 4)));
        if (blockIdx.x == 0 && threadIdx.x < D / 4) *reinterpret_cast<ai_f4*>(bn_part + 4 * threadIdx.x) = bnK;
    } else if (bn_part && blockIdx.x == 0 && threadIdx.x < D / 4) {
        *reinterpret_cast<ai_f4*>(bn_part + 4 * threadIdx.x) = bnK;
    }
    const int M = NP * D;                                   // columns of Wcat
    float wr[D / 4][C::TK], br[D / 4];
#pragma unroll
    for (int ct = 0; ct < D / 4; ++ct) {
        const int col = min(16 * ct + n, M - 1);
        br[ct] = ai_b<D>(w4, col);
#pragma unroll
        for (int t = 0; t < C::TK; ++t) wr[ct][t] = ai_w<D>(w4, C::TK * q + t, col);
    }
    const float scale = 1.0f / sqrtf((float)DH);
    const int nwaves = gridDim.x * 4;
    float xa[2][C::TK];
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b < B) ai_load_x<D>(x, b, F, n, q, xa);
    for (; b < B; b += nwaves) {
        if (xn.mean) ai_xn_rows<D>(xc, q, xa);              // (here, not behind the load: the prefetch must stay in flight)
        ai_project<D, BF>(xa, wr, br, NP, ys, n, q);
        if (b + nwaves < B) ai_load_x<D>(x, b + nwaves, F, n, q, xa);      // next row's operand while this one is attended
        ai_fence();
#pragma unroll
        for (int h = 0; h < C::H; ++h) {
            ai_f4 st[2][2];
            ai_tiles<DH>(ys, C::YS, D + DH * h, DH * h, n, q, st);         // A = K_h, B = Q_h
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                float m = -3.0e38f;
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = 16 * J + 4 * q + r < F;
                        st[J][I][r] = ok ? st[J][I][r] * scale : -3.0e38f;
                        m = fmaxf(m, st[J][I][r]);
                    }
                m = ai_qmax(m);
                float l = 0.f;
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = 16 * J + 4 * q + r < F ? __expf(st[J][I][r] - m) : 0.f;
                        st[J][I][r] = e;
                        l += e;
                    }
                l = ai_qsum(l);
                const float inv = 1.0f / l;
                const int i = 16 * I + n;
                if (lse_out && q == 0 && i < F) lse_out[((int64_t)b * C::H + h) * F + i] = m + logf(l);
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = st[J][I][r] * inv;
                        if (DROP) pv *= ai_keep(seed, drop_thr, (unsigned)b, h, i, 16 * J + 4 * q + r, inv_keep);
                        st[J][I][r] = pv;
                    }
            }
            ai_f4 o[2];
            ai_apply<DH>(st, ys, C::YS, 2 * D + DH * h, n, q, o);          // O_h = P V_h
            if (n < DH) {
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * I + 4 * q + r;
                        const float res = NP == 4 ? ys[i * C::YS + 3 * D + DH * h + n] : 0.f;
                        ys[i * C::YS + DH * h + n] = fmaxf(o[I][r] + res, 0.f);    // over Q_h, which is dead now
                    }
            }
        }
        ai_fence();
        // the row block [F][D] leaves as whole rows
        for (int e = lane; e < F * (D / 4); e += 64) {
            const int i = e / (D / 4), c4 = e - i * (D / 4);
            const ai_f4 v = *reinterpret_cast<const ai_f4*>(ys + i * C::YS + 4 * c4);
            *reinterpret_cast<ai_f4*>(out_a + ((int64_t)b * F + i) * D + 4 * c4) = v;
            const ai_f4 dlt = v - bnK;
            bnS += dlt;
            bnQ += dlt * dlt;
        }
        ai_fence();
    }
    if (bn_part) {
        // lanes with the same float4 column -> lane % (D/4); waves through their (dead) slabs; one record per block
#pragma unroll
        for (int o = D / 4; o < 64; o <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bnS[c] += __shfl_xor(bnS[c], o, 64);
                bnQ[c] += __shfl_xor(bnQ[c], o, 64);
            }
        }
        if (lane < D / 4) {
            *reinterpret_cast<ai_f4*>(ys + 4 * lane) = bnS;
            *reinterpret_cast<ai_f4*>(ys + D + 4 * lane) = bnQ;
        }
        __syncthreads();
        if (threadIdx.x < 2 * D) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += lds[w * 32 * C::YS + threadIdx.x];
            bn_part[D + (int64_t)blockIdx.x * 2 * D + threadIdx.x] = v;
        }
    }
}

// second launch of dt_autoint_fwd_bn: every block adds up the forward's block sums (<= 512 records of 2 D floats: L2 reads,
// 1024 threads, eight independent loads in flight each), forms mean / rstd, and normalises its share of a -> y; block 0 also
// leaves save_mean / save_rstd for the backward and moves the moving statistics (bn.hip k_bn_finalize's arithmetic).  The
// shift K is read from the record's copy, NOT from moving_mean (block 0 rewrites that while others are still here).
__global__ __launch_bounds__(1024) void k_autoint_bn_apply(const float* __restrict__ a, int total4, int D,
                                                           const float* __restrict__ part, int nparts, float inv_n,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float momentum, float* __restrict__ moving_mean,
                                                           float* __restrict__ moving_var, float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd, float* __restrict__ y,
                                                           double* __restrict__ zero_sums) {
    // zero_sums [2 D] (may be NULL): the double accumulators the layer ABOVE adds this normalisation's backward sums into
    // (AiPrev) start every step at zero — here, not in a fill launch of their own
    if (zero_sums && blockIdx.x == 0 && threadIdx.x < 2 * D) zero_sums[threadIdx.x] = 0.0;
    __shared__ __attribute__((aligned(16))) float red[1024];
    __shared__ __attribute__((aligned(16))) float cmean[32], crstd[32], cg[32], cb[32];
    const int t = threadIdx.x, W2 = 2 * D;                  // W2 = 32 or 64 values per record
    const int grp = t / W2, ngrp = 1024 / W2, col = t - grp * W2;
    const float* pp = part + D + col;
    float tot = 0.f;
    for (int k = grp; k < nparts; k += 32 * ngrp) {         // 32 records in flight per thread: one round trip for 512 records
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = k + u * ngrp < nparts ? pp[(int64_t)(k + u * ngrp) * W2] : 0.f;
#pragma unroll
        for (int u = 16; u > 0; u >>= 1)
#pragma unroll
            for (int w = 0; w < u; ++w) v[w] += v[w + u];
        tot += v[0];
    }
    red[t] = tot;
    __syncthreads();
    if (t < D) {
        float S = 0.f, Q = 0.f;
        for (int g2 = 0; g2 < ngrp; ++g2) {
            S += red[g2 * W2 + t];
            Q += red[g2 * W2 + D + t];
        }
        const float K = part[t];
        const float dm = S * inv_n;                         // mean - K
        const float mean = K + dm;
        const float var = fmaxf(Q * inv_n - dm * dm, 0.f);
        const float rstd = 1.0f / sqrtf(var + eps);
        cmean[t] = mean;
        crstd[t] = rstd;
        cg[t] = gamma ? gamma[t] : 1.f;
        cb[t] = beta ? beta[t] : 0.f;
        if (blockIdx.x == 0) {
            save_mean[t] = mean;
            save_rstd[t] = rstd;
            if (moving_mean) moving_mean[t] = moving_mean[t] * momentum + mean * (1.f - momentum);
            if (moving_var) moving_var[t] = moving_var[t] * momentum + var * (1.f - momentum);
        }
    }
    __syncthreads();
    if (!y) return;                                          // statistics only: the layer above normalises on load (AiXn)
    // the float4 column of a thread is the same in every iteration (the stride is a multiple of D/4)
    const int c4 = t & (D / 4 - 1);
    const ai_f4 m = *reinterpret_cast<const ai_f4*>(cmean + 4 * c4), r = *reinterpret_cast<const ai_f4*>(crstd + 4 * c4);
    const ai_f4 g = *reinterpret_cast<const ai_f4*>(cg + 4 * c4), bb = *reinterpret_cast<const ai_f4*>(cb + 4 * c4);
    const ai_f4* a4 = reinterpret_cast<const ai_f4*>(a);
    ai_f4* y4 = reinterpret_cast<ai_f4*>(y);
    const int stride = gridDim.x * 1024;
    int e = blockIdx.x * 1024 + t;
    for (; e + 3 * stride < total4; e += 4 * stride) {       // four 16-byte loads in flight per thread
        const ai_f4 v0 = a4[e], v1 = a4[e + stride], v2 = a4[e + 2 * stride], v3 = a4[e + 3 * stride];
        y4[e] = (v0 - m) * r * g + bb;                       // the rounding sequence of bn.hip k_bn_apply
        y4[e + stride] = (v1 - m) * r * g + bb;
        y4[e + 2 * stride] = (v2 - m) * r * g + bb;
        y4[e + 3 * stride] = (v3 - m) * r * g + bb;
    }
    for (; e < total4; e += stride) y4[e] = (a4[e] - m) * r * g + bb;
}

// Backward.  g = gradient w.r.t. a = relu(O + R) (after the BatchNormalization backward), a = the saved forward output.
// dY [B,F,NP*D] = gradient w.r.t. the PRE-activations of q | k | v [| residual] (the weight gradient x^T dY is a
// batch reduction and stays dt_dense_bwd's); dX [B,F,D] = dY Wcat^T is formed here, from the slab.
// 8 waves per block (two per SIMD); the weights live ONCE per block in LDS (wl [D][NP*D+4]), not in registers.
// (Round 2 tried a block-shared LDS accumulator for x^T dY fed by ds_add_f32: ~1 lane-atomic per 3 cycles, 358 us against
// 189 + 42 us for this kernel + the Dense weight-gradient kernel.  Round 3's WG variant below keeps the accumulators in
// registers — 245 VGPRs, no spill, with the per-element dropout code compiled out.)
// WG (round 3): the kernel / bias gradients x^T dY and colsum(dY) (the batch reduction dt_dense_bwd ran as its own launch,
// re-reading the 109 MB of dY this kernel had just written) are accumulated HERE, per wave in 64 + 8 accumulator registers,
// from the dY slab while it is still in LDS: dWc[d][m] += sum_field x[field][d] dY[field][m] as 2 x (M/16) MFMA tiles x 7
// field steps per batch row.  The block's eight waves sum their accumulators through the (dead) slabs at the end and leave
// ONE partial [D*M + M] per block in wpart; dt_autoint_bwd_w's second launch adds the <= 256 partials and writes the four
// Keras variables' gradients.  dY itself never reaches HBM (pass dY = NULL).
template <int D, int DH, bool WG, bool DROP, int BF = 0>
__device__ __forceinline__ void ai_bwd_body(const float* __restrict__ x, AiW w4, const float* __restrict__ a,
                                            const float* __restrict__ g, int B, int F, int NP,
                                            float* __restrict__ dY, float* __restrict__ dX, AiBn bn,
                                            unsigned drop_thr, float inv_keep, unsigned seed, float* __restrict__ wpart,
                                            AiPrev pv, AiXn xn, const float* __restrict__ g1w = nullptr) {
    // g1w != NULL: the incoming gradient is RANK ONE — the layer's (normalised, flattened) output feeds a Dense(1) and nothing
    // else, so dL/dy[b][i][c] = g[b] g1w[i D + c] with g [B] the gradient of that unit's output and g1w [F D] its kernel: formed
    // here from one scalar and the kernel's L1-resident rows instead of a [B,F,D] tensor written and read back (54 MB)
    using C = AiCfg<D, DH>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    float* wl = lds;                                       // [D][WS] Wcat, shared by the block's 8 waves
    constexpr int WS = 4 * D + kAiPad;
    float* ys = lds + D * WS + wave * (32 * C::YS + 96);
    float* gs = ys + 3 * D;                                // dZ = g * (a > 0) (= dO of every head) replaces R in the slab
    float* stat = ys + 32 * C::YS;                         // [32] m | [32] 1/l | [32] delta  of the current head
    constexpr int GS = C::YS;
    const int M = NP * D;
    for (int e = threadIdx.x; e < D * M; e += blockDim.x) wl[(e / M) * WS + (e % M)] = ai_w<D>(w4, e / M, e % M);
    float br[D / 4];
#pragma unroll
    for (int ct = 0; ct < D / 4; ++ct) br[ct] = ai_b<D>(w4, min(16 * ct + n, M - 1));
    const float* xc = lds + D * WS + 8 * (32 * C::YS + 96) + 8 * 2 * D;     // [2 D] input-normalisation constants (AiXn)
    const bool xnorm = xn.mean != nullptr;
    if (xnorm) ai_xn_fill<D>(xn, lds + D * WS + 8 * (32 * C::YS + 96) + 8 * 2 * D);
    __syncthreads();
    // BN-backward constants of this lane's 4 channels (the float4 column of the g / a loads below is lane % (D/4))
    ai_f4 bc1 = {1.f, 1.f, 1.f, 1.f}, bc2 = {0.f, 0.f, 0.f, 0.f}, bc3 = bc2, bmu = bc2;
    if (bn.mean) {
        const int c4 = lane % (D / 4);
        const ai_f4 rs = *reinterpret_cast<const ai_f4*>(bn.rstd + 4 * c4);
        const ai_f4 ga = bn.gamma ? *reinterpret_cast<const ai_f4*>(bn.gamma + 4 * c4) : bc1;
        bmu = *reinterpret_cast<const ai_f4*>(bn.mean + 4 * c4);
        bc1 = ga * rs;
        ai_f4 sg, sgx;
        if (bn.sums64) {                                     // (the float values a conversion launch would have left)
#pragma unroll
            for (int e = 0; e < 4; ++e) { sg[e] = (float)bn.sums64[4 * c4 + e]; sgx[e] = (float)bn.sums64[D + 4 * c4 + e]; }
        } else {
            sg = *reinterpret_cast<const ai_f4*>(bn.sum_g + 4 * c4);
            sgx = *reinterpret_cast<const ai_f4*>(bn.sum_gx + 4 * c4);
        }
        bc2 = sg * bn.inv_n;
        bc3 = rs * sgx * bn.inv_n;
    }
    const float scale = 1.0f / sqrtf((float)DH);
    const int nwaves = gridDim.x * 8;
    constexpr int WT = WG ? D / 4 : 1;                      // column tiles of dY (4D / 16)
    constexpr int WSTEPS = 7;                               // field steps of the weight-gradient product (fields 4s + q < 28)
    ai_f4 wacc[D / 16][WT];                                 // wacc[T][ct][r] = dWc[16T + 4q + r][16ct + n]
    float bacc[2] = {0.f, 0.f};                             // column sums of dY: columns lane and 64 + lane
    // AiPrev: the wave's running sums live in LDS (pacc [2 D] behind the eight slabs), not in eight registers all kernel long
    float* pacc = lds + D * WS + 8 * (32 * C::YS + 96) + wave * 2 * D;
    if (pv.a && lane < 2 * D / 4) *reinterpret_cast<ai_f4*>(pacc + 4 * lane) = ai_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < WT; ++ct) {
#pragma unroll
        for (int T = 0; T < D / 16; ++T) wacc[T][ct] = ai_f4{0.f, 0.f, 0.f, 0.f};
    }
    float xa[2][C::TK];
    int64_t b = (int64_t)blockIdx.x * 8 + wave;
    if (b < B) ai_load_x<D>(x, b, F, n, q, xa);
    for (; b < B; b += nwaves) {
        if (xnorm) ai_xn_rows<D>(xc, q, xa);                // (here, not behind the prefetch; its constants die before gz lives)
        // dZ rows -> LDS (rows >= F: zero), issued before the projections so the loads fly under them
        ai_f4 gz[(32 * (D / 4) + 63) / 64];
#pragma unroll
        for (int u = 0; u < (32 * (D / 4) + 63) / 64; ++u) {
            const int e = lane + 64 * u;
            const int i = e / (D / 4), c4 = e - i * (D / 4);
            gz[u] = ai_f4{0.f, 0.f, 0.f, 0.f};
            if (i < F) {
                ai_f4 gv;
                if (g1w) gv = *reinterpret_cast<const ai_f4*>(g1w + i * D + 4 * c4) * g[b];
                else gv = *reinterpret_cast<const ai_f4*>(g + ((int64_t)b * F + i) * D + 4 * c4);
                const ai_f4 av = *reinterpret_cast<const ai_f4*>(a + ((int64_t)b * F + i) * D + 4 * c4);
                if (bn.mean) gv = bc1 * (gv - bc2 - (av - bmu) * bc3);
                gz[u] = ai_f4{av.x > 0.f ? gv.x : 0.f, av.y > 0.f ? gv.y : 0.f, av.z > 0.f ? gv.z : 0.f,
                              av.w > 0.f ? gv.w : 0.f};
            }
        }
        ai_project_lds<D, BF>(xa, wl, WS, br, NP, ys, n, q);
        ai_fence();
        // residual branch: d(pre-activation of R) = dZ * (R > 0), straight to HBM
        unsigned rmask = 0;                                  // relu mask of this lane's R elements, 4 bits per float4
        if (NP == 4) {
#pragma unroll
            for (int u = 0; u < (32 * (D / 4) + 63) / 64; ++u) {
                const int e = lane + 64 * u;
                const int i = e / (D / 4), c4 = e - i * (D / 4);
                if (i < F) {
                    const ai_f4 rv = *reinterpret_cast<const ai_f4*>(ys + i * C::YS + 3 * D + 4 * c4);
                    const ai_f4 gv = gz[u];
                    rmask |= ((rv.x > 0.f ? 1u : 0u) | (rv.y > 0.f ? 2u : 0u) | (rv.z > 0.f ? 4u : 0u) | (rv.w > 0.f ? 8u : 0u)) << (4 * u);
                    if (dY)
                        *reinterpret_cast<ai_f4*>(dY + ((int64_t)b * F + i) * M + 3 * D + 4 * c4) =
                            ai_f4{rv.x > 0.f ? gv.x : 0.f, rv.y > 0.f ? gv.y : 0.f, rv.z > 0.f ? gv.z : 0.f,
                                  rv.w > 0.f ? gv.w : 0.f};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < (32 * (D / 4) + 63) / 64; ++u) {
            const int e = lane + 64 * u;
            const int i = e / (D / 4), c4 = e - i * (D / 4);
            if (i < 32) *reinterpret_cast<ai_f4*>(gs + i * GS + 4 * c4) = gz[u];
        }
        ai_fence();
#pragma unroll 1
        for (int h = 0; h < C::H; ++h) {        // not unrolled: the heads' live ranges must not overlap (256 registers)
            const int qc = DH * h, kc = D + DH * h, vc = 2 * D + DH * h;
            // ---- queries on lanes: P^T, dP^T -> delta, dS^T -> dQ ----
            ai_f4 pt[2][2], dpt[2][2];
            ai_tiles<DH>(ys, C::YS, kc, qc, n, q, pt);                     // S^T = K Q^T
            {
                // dP^T[j][i] = V_j . dO_i: A = V_h rows (slab), B = dO rows (gs slab, other stride): inline variant
                constexpr int SK = DH / 4;
                float av[2][SK], bv[2][SK];
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int t = 0; t < SK; ++t) {
                        av[T][t] = ys[(16 * T + n) * C::YS + vc + SK * q + t];
                        bv[T][t] = gs[(16 * T + n) * GS + qc + SK * q + t];
                    }
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int I = 0; I < 2; ++I) {
                        ai_f4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < SK; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[J][t], bv[I][t], c, 0, 0, 0);
                        dpt[J][I] = c;
                    }
            }
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                float m = -3.0e38f;
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = 16 * J + 4 * q + r < F;
                        pt[J][I][r] = ok ? pt[J][I][r] * scale : -3.0e38f;
                        m = fmaxf(m, pt[J][I][r]);
                    }
                m = ai_qmax(m);
                float l = 0.f;
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = 16 * J + 4 * q + r < F ? __expf(pt[J][I][r] - m) : 0.f;
                        pt[J][I][r] = e;
                        l += e;
                    }
                l = ai_qsum(l);
                const float inv = 1.0f / l;
                const int i = 16 * I + n;
                float delta = 0.f;
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = pt[J][I][r] * inv;
                        const float keep = DROP ? ai_keep(seed, drop_thr, (unsigned)b, h, i, 16 * J + 4 * q + r, inv_keep) : 1.f;
                        dpt[J][I][r] *= keep;                               // gradient w.r.t. the un-dropped probability
                        pt[J][I][r] = pv;
                        delta += pv * dpt[J][I][r];
                    }
                delta = ai_qsum(delta);
                if (q == 0) { stat[i] = m; stat[32 + i] = inv; stat[64 + i] = delta; }
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[J][I][r] = pt[J][I][r] * (dpt[J][I][r] - delta) * scale;   // dS^T
            }
            ai_f4 dq[2];
            ai_apply<DH>(pt, ys, C::YS, kc, n, q, dq);                     // dQ_h = dS K_h
            ai_fence();
            // ---- keys on lanes: P, dP -> dS -> dV = P'^T dO, dK = dS^T Q ----
            ai_f4 pn[2][2], dpn[2][2];
            {
                constexpr int SK = DH / 4;
                float qa[2][SK], kb[2][SK], ga[2][SK], vb[2][SK];
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int t = 0; t < SK; ++t) {
                        qa[T][t] = ys[(16 * T + n) * C::YS + qc + SK * q + t];
                        kb[T][t] = ys[(16 * T + n) * C::YS + kc + SK * q + t];
                        ga[T][t] = gs[(16 * T + n) * GS + qc + SK * q + t];
                        vb[T][t] = ys[(16 * T + n) * C::YS + vc + SK * q + t];
                    }
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int J = 0; J < 2; ++J) {
                        ai_f4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
#pragma unroll
                        for (int t = 0; t < SK; ++t) {
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[I][t], kb[J][t], c, 0, 0, 0);   // S[i][j], i = 16I+4q+r, j = 16J+n
                            d = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[I][t], vb[J][t], d, 0, 0, 0);   // dP[i][j]
                        }
                        pn[I][J] = c;
                        dpn[I][J] = d;
                    }
            }
            ai_f4 pdrop[2][2];                                              // dropped probabilities (A operand of dV)
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * I + 4 * q + r;
                    const float m = stat[i], inv = stat[32 + i], delta = stat[64 + i];
#pragma unroll
                    for (int J = 0; J < 2; ++J) {
                        const int j = 16 * J + n;
                        const float pv = j < F ? __expf(pn[I][J][r] * scale - m) * inv : 0.f;
                        const float keep = DROP ? ai_keep(seed, drop_thr, (unsigned)b, h, i, j, inv_keep) : 1.f;
                        pdrop[I][J][r] = i < F ? pv * keep : 0.f;
                        pn[I][J][r] = i < F ? pv * (dpn[I][J][r] * keep - delta) * scale : 0.f;            // dS[i][j]
                    }
                }
            // dV[j][d] = sum_i P'[i][j] dO[i][d];  dK[j][d] = sum_i dS[i][j] Q[i][d]: the register-side index is the query i
            ai_f4 dv[2], dk[2];
            {
                float gb[2][4], qb[2][4];
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gb[I][r] = gs[(16 * I + 4 * q + r) * GS + qc + (n & (DH - 1))];
                        qb[I][r] = ys[(16 * I + 4 * q + r) * C::YS + qc + (n & (DH - 1))];
                    }
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    ai_f4 c = {0.f, 0.f, 0.f, 0.f}, d = c;
#pragma unroll
                    for (int I = 0; I < 2; ++I)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(pdrop[I][J][r], gb[I][r], c, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x4f32(pn[I][J][r], qb[I][r], d, 0, 0, 0);
                        }
                    dv[J] = c;
                    dk[J] = d;
                }
            }
            ai_fence();
            // relu masks of the projections, gradients over the head's own (now dead) Q / K / V columns
            if (n < DH) {
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * T + 4 * q + r;
                        float* yr = ys + row * C::YS;
                        const float qv = yr[qc + n], kv = yr[kc + n], vv = yr[vc + n];
                        yr[qc + n] = qv > 0.f ? dq[T][r] : 0.f;
                        yr[kc + n] = kv > 0.f ? dk[T][r] : 0.f;
                        yr[vc + n] = vv > 0.f ? dv[T][r] : 0.f;
                    }
            }
            ai_fence();
        }
        // dQ | dK | dV rows [F][3D] leave as whole rows (when the caller wants dY)
        for (int e = lane; dY && e < F * (3 * D / 4); e += 64) {
            const int i = e / (3 * D / 4), c4 = e - i * (3 * D / 4);
            *reinterpret_cast<ai_f4*>(dY + ((int64_t)b * F + i) * M + 4 * c4) =
                *reinterpret_cast<const ai_f4*>(ys + i * C::YS + 4 * c4);
        }
        // x^T as the A operand of the weight-gradient product: xw[T][s] = x[b][4s + q][16T + n] (fields >= F: 0); these hit L2
        // (the wave read the row a few microseconds ago) and fly under the dX product
        float xw[D / 16][WG ? WSTEPS : 1];
        if (WG) {
#pragma unroll
            for (int s = 0; s < WSTEPS; ++s)
#pragma unroll
                for (int T = 0; T < D / 16; ++T)
                    xw[T][s] = 4 * s + q < F ? x[((int64_t)b * F + 4 * s + q) * D + 16 * T + n] : 0.f;
        }
        if (b + nwaves < B) ai_load_x<D>(x, b + nwaves, F, n, q, xa);      // next row's operand flies under the dX product
        if (dX || WG) {
            // the residual block of the slab becomes d(pre-activation of R) = dZ * (R > 0): the slab row is now dY
            if (NP == 4) {
#pragma unroll
                for (int u = 0; u < (32 * (D / 4) + 63) / 64; ++u) {
                    const int e = lane + 64 * u;
                    const int i = e / (D / 4), c4 = e - i * (D / 4);
                    if (i < 32) {
                        const unsigned mk = rmask >> (4 * u);
                        const ai_f4 zv = *reinterpret_cast<const ai_f4*>(gs + i * GS + 4 * c4);
                        *reinterpret_cast<ai_f4*>(gs + i * GS + 4 * c4) =
                            ai_f4{mk & 1u ? zv.x : 0.f, mk & 2u ? zv.y : 0.f, mk & 4u ? zv.z : 0.f, mk & 8u ? zv.w : 0.f};
                    }
                }
            }
            ai_fence();
        }
        if (WG && BF) {
            // bf16 mode: the field index is the contraction index — field(g = q, j) = 4 j + q, j < 7 (the eighth is zero):
            // the seven fp32 steps' operands of a lane are ONE bf16x8 operand
            if constexpr (BF) {
                ai_b8 xa8[D / 16], xa8l[D / 16];
#pragma unroll
                for (int T = 0; T < D / 16; ++T) {
                    float v[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) v[s] = s < WSTEPS ? xw[T][s < WSTEPS ? s : 0] : 0.f;
                    if constexpr (BF == 2) ai_split8(v, xa8[T], xa8l[T]); else xa8[T] = ai_pack8(v);
                }
#pragma unroll
                for (int ct = 0; ct < WT; ++ct) {
                    if (16 * ct >= M) break;
                    float v[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) v[s] = s < WSTEPS ? ys[(4 * (s < WSTEPS ? s : 0) + q) * C::YS + n + 16 * ct] : 0.f;
                    if constexpr (BF == 2) {
                        ai_b8 dyh, dyl;
                        ai_split8(v, dyh, dyl);
#pragma unroll
                        for (int T = 0; T < D / 16; ++T) {
                            AI_MFMA_BF16(wacc[T][ct], xa8[T], dyl);
                            AI_MFMA_BF16(wacc[T][ct], xa8l[T], dyh);
                            AI_MFMA_BF16(wacc[T][ct], xa8[T], dyh);
                        }
                    } else {
                        const ai_b8 dy8 = ai_pack8(v);
#pragma unroll
                        for (int T = 0; T < D / 16; ++T) AI_MFMA_BF16(wacc[T][ct], xa8[T], dy8);
                    }
                }
            }
        } else if (WG) {
#pragma unroll
            for (int s = 0; s < WSTEPS; ++s) {
                const float* yr = ys + (4 * s + q) * C::YS + n;
#pragma unroll
                for (int ct = 0; ct < WT; ++ct) {
                    if (16 * ct >= M) break;
                    const float dv_ = yr[16 * ct];
#pragma unroll
                    for (int T = 0; T < D / 16; ++T)
                        wacc[T][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[T][s], dv_, wacc[T][ct], 0, 0, 0);
                }
            }
        }
        if (WG) {
            for (int i = 0; i < F; ++i) {
                bacc[0] += ys[i * C::YS + lane];
                if (64 + lane < M) bacc[1] += ys[i * C::YS + 64 + lane];
            }
        }
        if (dX) {
            // dX[i][k] = sum_m dY[i][m] W[k][m]; the contraction index is permuted so that a lane's m are contiguous
            // (m = (M/4) q + t): A and B operands are 16-byte LDS reads
            ai_f4 dx[2][D / 16];
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int ct = 0; ct < D / 16; ++ct) dx[T][ct] = ai_f4{0.f, 0.f, 0.f, 0.f};
            const int MQ = M / 4;                            // 24 or 32 (D = 32), 12 or 16 (D = 16): multiples of 4
            if constexpr (BF) {
                // bf16 mode: eight consecutive m of a lane per step (MQ = 24 or 32: three or four steps)
                auto ld8 = [](const float* p, float (&v)[8]) {
                    const ai_f4 u = *reinterpret_cast<const ai_f4*>(p), w = *reinterpret_cast<const ai_f4*>(p + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = u[e]; v[4 + e] = w[e]; }
                };
                for (int t = 0; t < MQ; t += 8) {
                    const float* ap = ys + n * C::YS + MQ * q + t;
                    float av0[8], av1[8];
                    ld8(ap, av0);
                    ld8(ap + 16 * C::YS, av1);
                    if constexpr (BF == 2) {
                        ai_b8 a0h, a0l, a1h, a1l;
                        ai_split8(av0, a0h, a0l);
                        ai_split8(av1, a1h, a1l);
#pragma unroll
                        for (int ct = 0; ct < D / 16; ++ct) {
                            float wv8[8];
                            ld8(wl + (16 * ct + n) * WS + MQ * q + t, wv8);
                            ai_b8 wh, wlo;
                            ai_split8(wv8, wh, wlo);
                            AI_MFMA_BF16(dx[0][ct], a0h, wlo); AI_MFMA_BF16(dx[1][ct], a1h, wlo);
                            AI_MFMA_BF16(dx[0][ct], a0l, wh); AI_MFMA_BF16(dx[1][ct], a1l, wh);
                            AI_MFMA_BF16(dx[0][ct], a0h, wh); AI_MFMA_BF16(dx[1][ct], a1h, wh);
                        }
                    } else {
                        const ai_b8 a0 = ai_pack8(av0), a1 = ai_pack8(av1);
#pragma unroll
                        for (int ct = 0; ct < D / 16; ++ct) {
                            float wv8[8];
                            ld8(wl + (16 * ct + n) * WS + MQ * q + t, wv8);
                            const ai_b8 w8 = ai_pack8(wv8);
                            AI_MFMA_BF16(dx[0][ct], a0, w8);
                            AI_MFMA_BF16(dx[1][ct], a1, w8);
                        }
                    }
                }
            } else
            for (int t = 0; t < MQ; t += 4) {
                const ai_f4 a0 = *reinterpret_cast<const ai_f4*>(ys + n * C::YS + MQ * q + t);
                const ai_f4 a1 = *reinterpret_cast<const ai_f4*>(ys + (16 + n) * C::YS + MQ * q + t);
#pragma unroll
                for (int ct = 0; ct < D / 16; ++ct) {
                    const ai_f4 w = *reinterpret_cast<const ai_f4*>(wl + (16 * ct + n) * WS + MQ * q + t);
                    dx[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w.x, dx[0][ct], 0, 0, 0);
                    dx[1][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w.x, dx[1][ct], 0, 0, 0);
                    dx[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w.y, dx[0][ct], 0, 0, 0);
                    dx[1][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w.y, dx[1][ct], 0, 0, 0);
                    dx[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w.z, dx[0][ct], 0, 0, 0);
                    dx[1][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w.z, dx[1][ct], 0, 0, 0);
                    dx[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w.w, dx[0][ct], 0, 0, 0);
                    dx[1][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w.w, dx[1][ct], 0, 0, 0);
                }
            }
            ai_fence();
            // stage over the (dead) first D columns of the slab, leave as whole rows
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int ct = 0; ct < D / 16; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ys[(16 * T + 4 * q + r) * C::YS + 16 * ct + n] = dx[T][ct][r];
            ai_fence();
            // (mean / rstd of the lane's channels are re-read per row — L1 hits — instead of living in eight registers all kernel long)
            ai_f4 pmu = {0.f, 0.f, 0.f, 0.f}, prs = pmu, psg = pmu, psgx = pmu;
            if (pv.a) {
                pmu = *reinterpret_cast<const ai_f4*>(pv.mean + 4 * (lane % (D / 4)));
                prs = *reinterpret_cast<const ai_f4*>(pv.rstd + 4 * (lane % (D / 4)));
            }
            for (int e = lane; e < F * (D / 4); e += 64) {
                const int i = e / (D / 4), c4 = e - i * (D / 4);
                const ai_f4 dxv = *reinterpret_cast<const ai_f4*>(ys + i * C::YS + 4 * c4);
                *reinterpret_cast<ai_f4*>(dX + ((int64_t)b * F + i) * D + 4 * c4) = dxv;
                if (pv.a) {
                    const ai_f4 av = *reinterpret_cast<const ai_f4*>(pv.a + ((int64_t)b * F + i) * D + 4 * c4);
                    psg += dxv;
                    psgx += dxv * ((av - pmu) * prs);
                }
            }
            if (pv.a) {          // the row's sums: lanes with the same float4 column meet by shuffles, lanes < D/4 add them up
#pragma unroll
                for (int o = D / 4; o < 64; o <<= 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        psg[c] += __shfl_xor(psg[c], o, 64);
                        psgx[c] += __shfl_xor(psgx[c], o, 64);
                    }
                }
                if (lane < D / 4) {
                    *reinterpret_cast<ai_f4*>(pacc + 4 * lane) += psg;
                    *reinterpret_cast<ai_f4*>(pacc + D + 4 * lane) += psgx;
                }
            }
        }
        ai_fence();
    }
    if (pv.a) {
        // the block's eight waves meet; ONE double atomic per channel and sum per block (<= 256 blocks: ~1.5 us of arrivals
        // on 2 D addresses)
        __syncthreads();
        if (threadIdx.x < 2 * D) {
            const float* p0 = lds + D * WS + 8 * (32 * C::YS + 96);
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += (double)p0[w * 2 * D + threadIdx.x];
            unsafeAtomicAdd(pv.sums + threadIdx.x, v);
        }
    }
    if (WG) {
        __syncthreads();
        float* out = wpart + (int64_t)blockIdx.x * (D * M + M);
        constexpr int SLAB = 32 * C::YS + 96;
        const float* s0 = lds + D * WS;
        // bias partials first: every wave leaves its column sums in row 0 of its slab; the block's sums also stay in LDS
        // (over the weights, dead now) for the AiXn correction below
        ys[lane] = bacc[0];
        ys[64 + lane] = bacc[1];
        __syncthreads();
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s0[w * SLAB + m];
            out[D * M + m] = v;
            wl[m] = v;
        }
        __syncthreads();
        // the eight waves' accumulators -> their slabs ([D][YS]: row d, column m), summed by the whole block
#pragma unroll
        for (int ct = 0; ct < WT; ++ct) {
            if (16 * ct >= M) break;
#pragma unroll
            for (int T = 0; T < D / 16; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) ys[(16 * T + 4 * q + r) * C::YS + 16 * ct + n] = wacc[T][ct][r];
        }
        __syncthreads();
        // AiXn: the accumulators hold a_prev^T dY (the x^T operand was loaded un-normalised); with x = s a_prev + t per
        // channel (s = rstd gamma, t = beta - mean s):  x^T dY = s (a_prev^T dY) + t colsum(dY) — one correction per block
        // partial instead of a normalisation of every operand element in the row loop
        for (int e = threadIdx.x; e < D * M; e += blockDim.x) {
            const int d = e / M, m = e - d * M;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s0[w * SLAB + d * C::YS + m];
            if (xnorm) v = xc[d] * v + xc[D + d] * wl[m];
            out[e] = v;
        }
    }
}

template <int D, int DH, bool DROP, int BF = 0>
__global__ __launch_bounds__(512) void k_autoint_bwd(const float* __restrict__ x, AiW w4, const float* __restrict__ a,
                                                     const float* __restrict__ g, int B, int F, int NP,
                                                     float* __restrict__ dY, float* __restrict__ dX, AiBn bn,
                                                     unsigned drop_thr, float inv_keep, unsigned seed, AiPrev pv, AiXn xn) {
    ai_bwd_body<D, DH, false, DROP, BF>(x, w4, a, g, B, F, NP, dY, dX, bn, drop_thr, inv_keep, seed, nullptr, pv, xn);
}
template <int D, int DH, bool DROP, int BF = 0>
__global__ __launch_bounds__(512) void k_autoint_bwd_w(const float* __restrict__ x, AiW w4, const float* __restrict__ a,
                                                       const float* __restrict__ g, int B, int F, int NP,
                                                       float* __restrict__ dX, AiBn bn, unsigned drop_thr, float inv_keep,
                                                       unsigned seed, float* __restrict__ wpart, AiPrev pv, AiXn xn,
                                                       const float* __restrict__ g1w) {
    ai_bwd_body<D, DH, true, DROP, BF>(x, w4, a, g, B, F, NP, nullptr, dX, bn, drop_thr, inv_keep, seed, wpart, pv, xn, g1w);
}

// sum of the per-block partials -> the gradients of the NP Keras kernels [NP][D][D] (gW[p][k][j] = dWc[k][p D + j]) and
// biases [NP][D]; nparts <= 256
__global__ __launch_bounds__(1024) void k_autoint_wgrad_reduce(const float* __restrict__ wpart, int nparts, int D, int M,
                                                               float* __restrict__ gW, float* __restrict__ gb,
                                                               const double* __restrict__ bn_sums64,
                                                               float* __restrict__ bn_grads) {
    // bn_grads [2 D] (may be NULL) = gradient of beta | gamma of the layer's BatchNormalization = the two sums as floats
    if (bn_grads && bn_sums64 && blockIdx.x == 0 && threadIdx.x < 2 * D) bn_grads[threadIdx.x] = (float)bn_sums64[threadIdx.x];
    // a block owns 64 consecutive elements (coalesced 256-byte reads of every partial); its 16 waves split the partials
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = D * M + M;
    const int e = blockIdx.x * 64 + lane;
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int p = wave + 16 * u;
        v[u] = (e < total && p < nparts) ? wpart[(int64_t)p * total + e] : 0.f;
    }
#pragma unroll
    for (int u = 8; u > 0; u >>= 1)
#pragma unroll
        for (int w = 0; w < u; ++w) v[w] += v[w + u];
    red[wave][lane] = v[0];
    __syncthreads();
    if (wave == 0 && e < total) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        if (e < D * M) {
            const int k = e / M, m = e - k * M;
            gW[((int64_t)(m / D) * D + k) * D + (m % D)] = t;
        } else {
            gb[e - D * M] = t;
        }
    }
}

// ---- the head of the AutoInt graph: BatchNormalization (pending: AiXn) -> Flatten -> Dense(1) (deepnets.py:222-224 +
// deepmodel.py:131-143 `task_output` / `dense_logit_*`) without the normalised tensor -------------------------------------------
// forward: z[b] = sum_k (s a[b][k] + t) w[k] + bias, one wave per row (four rows per wave at the Criteo shape); s, t are formed
// once per block in LDS, a wave's share of w, s, t stays in registers
constexpr int kAiHeadK4 = 4;            // float4 chunks per lane: K <= 1024
__global__ __launch_bounds__(256) void k_autoint_head_fwd(const float* __restrict__ a, const float* __restrict__ w,
                                                          const float* __restrict__ bias, AiXn xn, int B, int K, int D,
                                                          float* __restrict__ z) {
    __shared__ __attribute__((aligned(16))) float cs[64], ct[64];
    if ((int)threadIdx.x < D) {
        const int c = threadIdx.x;
        const float s1 = xn.rstd[c] * (xn.gamma ? xn.gamma[c] : 1.f);
        cs[c] = s1;
        ct[c] = (xn.beta ? xn.beta[c] : 0.f) - xn.mean[c] * s1;                          // ai_xn_fill's arithmetic
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, K4 = K >> 2, D4 = D >> 2;
    ai_f4 wv[kAiHeadK4], sc[kAiHeadK4], sh[kAiHeadK4];
#pragma unroll
    for (int u = 0; u < kAiHeadK4; ++u) {
        const int k4 = lane + 64 * u;
        wv[u] = sc[u] = sh[u] = ai_f4{0.f, 0.f, 0.f, 0.f};
        if (k4 < K4) {
            wv[u] = reinterpret_cast<const ai_f4*>(w)[k4];
            sc[u] = *reinterpret_cast<const ai_f4*>(cs + 4 * (k4 % D4));
            sh[u] = *reinterpret_cast<const ai_f4*>(ct + 4 * (k4 % D4));
        }
    }
    const float b0 = bias ? bias[0] : 0.f;
    const int wpb = blockDim.x >> 6;
    for (int64_t n = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); n < B; n += (int64_t)gridDim.x * wpb) {
        const ai_f4* ap = reinterpret_cast<const ai_f4*>(a) + n * K4;
        ai_f4 av[kAiHeadK4];
#pragma unroll
        for (int u = 0; u < kAiHeadK4; ++u) av[u] = lane + 64 * u < K4 ? ap[lane + 64 * u] : ai_f4{0.f, 0.f, 0.f, 0.f};
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < kAiHeadK4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += fmaf(av[u][e], sc[u][e], sh[u][e]) * wv[u][e];
        acc = wave_sum(acc);
        if (lane == 0) z[n] = acc + b0;
    }
}

// backward, launch 1: a block owns 32 rows; thread k4 accumulates R[k] = sum_rows gz[row] a[row][k] over its float4 column
// -> part[block][K + 4] (element K: the block's sum of gz)
constexpr int kAiHeadRows = 32;
__global__ __launch_bounds__(256) void k_autoint_head_bwd(const float* __restrict__ a, const float* __restrict__ gz, int B,
                                                          int K, float* __restrict__ part) {
    __shared__ float gs[kAiHeadRows];
    const int64_t r0 = (int64_t)blockIdx.x * kAiHeadRows;
    const int nr = (int)(min((int64_t)B, r0 + kAiHeadRows) - r0);
    if ((int)threadIdx.x < kAiHeadRows) gs[threadIdx.x] = (int)threadIdx.x < nr ? gz[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    float* prow = part + (int64_t)blockIdx.x * (K + 4);
    const int K4 = K >> 2;
    for (int k4 = threadIdx.x; k4 < K4; k4 += blockDim.x) {
        const ai_f4* xp = reinterpret_cast<const ai_f4*>(a) + r0 * K4 + k4;
        ai_f4 acc = {0.f, 0.f, 0.f, 0.f};
        int n = 0;
        for (; n + 7 < nr; n += 8) {                         // eight 16-byte loads in flight per thread
            ai_f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xp[(int64_t)(n + u) * K4];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u] * gs[n + u];
        }
        for (; n < nr; ++n) acc += xp[(int64_t)n * K4] * gs[n];
        reinterpret_cast<ai_f4*>(prow)[k4] = acc;
    }
    if (threadIdx.x < 64) {
        float bsum = (int)threadIdx.x < kAiHeadRows ? gs[threadIdx.x] : 0.f;
        bsum = wave_sum(bsum);
        if (threadIdx.x == 0) prow[K] = bsum;
    }
}

// backward, launch 2 (a block per 64 columns): R[k], S = sum gz from the partials (doubles); the Dense kernel's gradient
// gW[k] = s R + t S (x = s a + t), its bias gradient S, and the two batch sums of the pending BatchNormalization's backward
// for the rank-one gradient dL/dy = gz w:  sum_g[c] = S sum_f w[f][c],  sum_gx[c] = sum_f w[f][c] rstd_c (R[f][c] - mean_c S)
// — ADDED into bn_sums (doubles, zero when the step starts: dt_autoint_fwd_bn's zero_sums), one atomic per block and channel
__global__ __launch_bounds__(1024) void k_autoint_head_finish(const float* __restrict__ part, int nparts, int K, int D,
                                                              const float* __restrict__ w, AiXn xn, float* __restrict__ gW,
                                                              float* __restrict__ gb, double* __restrict__ bn_sums) {
    __shared__ double red[16][64];
    __shared__ double hS;
    const int t = threadIdx.x, lane = t & 63, grp = t >> 6;
    const int64_t stride = K + 4;
    double s = 0.0;
    for (int p = t; p < nparts; p += 1024) s += (double)part[(int64_t)p * stride + K];
    red[grp][lane] = s;
    __syncthreads();
    if (t < 64) {
        double v = 0.0;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) v += red[g2][t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (t == 0) { hS = v; if (gb && blockIdx.x == 0) gb[0] = (float)v; }
    }
    __syncthreads();
    const double S = hS;
    const int k = blockIdx.x * 64 + lane;
    double R = 0.0;
    if (k < K) {
        const float* pp = part + k;
        int p = grp;
        for (; p + 16 * 7 < nparts; p += 16 * 8) {           // eight loads in flight per thread
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(int64_t)(p + 16 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) R += (double)v[u];
        }
        for (; p < nparts; p += 16) R += (double)pp[(int64_t)p * stride];
    }
    __syncthreads();
    red[grp][lane] = R;
    __syncthreads();
    double hx = 0.0, hw = 0.0;
    if (grp == 0 && k < K) {
        R = 0.0;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) R += red[g2][lane];
        const int c = k % D;
        const float rs = xn.rstd[c], mu = xn.mean[c];
        const float s1 = rs * (xn.gamma ? xn.gamma[c] : 1.f);
        const float t1 = (xn.beta ? xn.beta[c] : 0.f) - mu * s1;
        gW[k] = (float)((double)s1 * R + (double)t1 * S);
        hw = (double)w[k];
        hx = hw * (double)rs * (R - (double)mu * S);
    }
    __syncthreads();
    if (grp == 0) { red[0][lane] = hw; red[1][lane] = hx; }
    __syncthreads();
    // the block's 64 columns are 64 / D whole fields (the block's first column is a multiple of 64, D divides 64)
    if (t < D) {
        double sw = 0.0, sx = 0.0;
        for (int j = t; j < 64; j += D) { sw += red[0][j]; sx += red[1][j]; }
        unsafeAtomicAdd(bn_sums + t, S * sw);
        unsafeAtomicAdd(bn_sums + D + t, sx);
    }
}

}  // namespace dt

using namespace dt;

extern "C" int dt_autoint_supported(int F, int D, int H) {
    if (F < 1 || F > 32 || H < 1 || D % H) return 0;
    const int dh = D / H;
    return ((D == 32 && (dh == 8 || dh == 16)) || (D == 16 && (dh == 4 || dh == 8 || dh == 16))) ? 1 : 0;
}

extern "C" unsigned dt_autoint_dropout_hash(unsigned seed, unsigned b, unsigned h, unsigned i, unsigned j) {
    return ai_hash(seed, b, h, i, j);
}

// the attention-weight dropout is a compile-time variant (DROP): with rate 0 the per-element mask code and its 32 branches
// per head are not in the kernel at all
#define DT_AI_LAUNCH(KERNEL, DV, HV, WAVES, ...)                                                                           \
    do {                                                                                                                   \
        if (thr) {                                                                                                         \
            hipFuncSetAttribute((const void*)KERNEL<DV, HV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
            hipLaunchKernelGGL((KERNEL<DV, HV, true>), dim3(blocks), dim3(64 * (WAVES)), lds, st, __VA_ARGS__);             \
        } else {                                                                                                           \
            hipFuncSetAttribute((const void*)KERNEL<DV, HV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            hipLaunchKernelGGL((KERNEL<DV, HV, false>), dim3(blocks), dim3(64 * (WAVES)), lds, st, __VA_ARGS__);            \
        }                                                                                                                  \
    } while (0)
#define DT_AI_LAUNCH_BF(KERNEL, DV, HV, WAVES, ...)                                                                        \
    do {                                                                                                                   \
        if (thr) {                                                                                                         \
            if (mfma_mode == DT_AI_BF16X2) {                                                                              \
                hipFuncSetAttribute((const void*)KERNEL<DV, HV, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                hipLaunchKernelGGL((KERNEL<DV, HV, true, 2>), dim3(blocks), dim3(64 * (WAVES)), lds, st, __VA_ARGS__);     \
            } else {                                                                                                       \
                hipFuncSetAttribute((const void*)KERNEL<DV, HV, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                hipLaunchKernelGGL((KERNEL<DV, HV, true, 1>), dim3(blocks), dim3(64 * (WAVES)), lds, st, __VA_ARGS__);     \
            }                                                                                                              \
        } else {                                                                                                           \
            if (mfma_mode == DT_AI_BF16X2) {                                                                              \
                hipFuncSetAttribute((const void*)KERNEL<DV, HV, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                hipLaunchKernelGGL((KERNEL<DV, HV, false, 2>), dim3(blocks), dim3(64 * (WAVES)), lds, st, __VA_ARGS__);    \
            } else {                                                                                                       \
                hipFuncSetAttribute((const void*)KERNEL<DV, HV, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                hipLaunchKernelGGL((KERNEL<DV, HV, false, 1>), dim3(blocks), dim3(64 * (WAVES)), lds, st, __VA_ARGS__);    \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
#define DT_AI_DISPATCH(KERNEL, WAVES, LDS_FLOATS, ...)                                                             \
    do {                                                                                                           \
        const int dh = D / H;                                                                                      \
        int blocks = (int)((B + (WAVES) - 1) / (WAVES));                                                           \
        if (blocks > 2048 / (WAVES)) blocks = 2048 / (WAVES);                                                      \
        const size_t lds = (size_t)(LDS_FLOATS) * sizeof(float);                                                   \
        if (mfma_mode != DT_AI_F32 && D == 32 && dh == 8) DT_AI_LAUNCH_BF(KERNEL, 32, 8, WAVES, __VA_ARGS__);      \
        else if (mfma_mode != DT_AI_F32 && D == 32 && dh == 16) DT_AI_LAUNCH_BF(KERNEL, 32, 16, WAVES, __VA_ARGS__); \
        else if (D == 32 && dh == 8) DT_AI_LAUNCH(KERNEL, 32, 8, WAVES, __VA_ARGS__);                              \
        else if (D == 32 && dh == 16) DT_AI_LAUNCH(KERNEL, 32, 16, WAVES, __VA_ARGS__);                            \
        else if (D == 16 && dh == 4) DT_AI_LAUNCH(KERNEL, 16, 4, WAVES, __VA_ARGS__);                              \
        else if (D == 16 && dh == 8) DT_AI_LAUNCH(KERNEL, 16, 8, WAVES, __VA_ARGS__);                              \
        else DT_AI_LAUNCH(KERNEL, 16, 16, WAVES, __VA_ARGS__);                                                     \
    } while (0)

static bool ai_drop(float rate, unsigned* thr, float* inv_keep) {
    *thr = 0; *inv_keep = 1.f;
    if (rate <= 0.f) return true;
    if (rate >= 1.f) return false;
    double t = (double)rate * 4294967296.0;
    *thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (*thr == 0) *thr = 1;
    *inv_keep = 1.0f / (1.0f - rate);
    return true;
}

static bool ai_weights(const float* const* W, const float* const* b, int NP, AiW* w) {
    for (int i = 0; i < 4; ++i) {
        w->W[i] = i < NP ? W[i] : W[0];
        w->b[i] = i < NP ? b[i] : b[0];
        if (i < NP && (!W[i] || !b[i])) return false;
    }
    return true;
}

extern "C" int dt_autoint_fwd(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr,
                              const float* bq, const float* bk, const float* bv, const float* br, int64_t B, int F,
                              int D, int H, float dropout_rate, unsigned seed, float* out_a, float* lse, const float* xn_mean,
                              const float* xn_rstd, const float* xn_gamma, const float* xn_beta, int mfma_mode, void* stream) {
    DT_UNSUPPORTED(mfma_mode != DT_AI_F32 && !((mfma_mode == DT_AI_BF16 || mfma_mode == DT_AI_BF16X2) && D == 32), "dt_autoint_fwd: mfma_mode %d (DT_AI_BF16 / DT_AI_BF16X2 need D = 32; D = %d)", mfma_mode, D);
    DT_UNSUPPORTED(!dt_autoint_supported(F, D, H), "dt_autoint_fwd: unsupported shape F=%d D=%d H=%d", F, D, H);
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && out_a && B > 0 && B < (1LL << 31), "dt_autoint_fwd: null pointer / bad batch");
    const int NP = Wr ? 4 : 3;
    const float* Ws[4] = {Wq, Wk, Wv, Wr};
    const float* bs[4] = {bq, bk, bv, br};
    AiW w4;
    DT_REQUIRE(ai_weights(Ws, bs, NP, &w4), "dt_autoint_fwd: null weight pointer");
    unsigned thr; float inv_keep;
    DT_REQUIRE(ai_drop(dropout_rate, &thr, &inv_keep), "dt_autoint_fwd: dropout_rate %f", dropout_rate);
    hipStream_t st = as_stream(stream);
    DT_REQUIRE(!xn_mean || xn_rstd, "dt_autoint_fwd: xn_mean needs xn_rstd");
    const AiXn xn{xn_mean, xn_rstd, xn_gamma, xn_beta};
    DT_AI_DISPATCH(k_autoint_fwd, 4, 4 * 32 * (4 * D + kAiPad) + 2 * D, x, w4, (int)B, F, NP, out_a, lse, thr, inv_keep, seed,
                   (const float*)nullptr, (float*)nullptr, xn);
    return launch_status("dt_autoint_fwd");
}

extern "C" int dt_autoint_bwd(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr,
                              const float* bq, const float* bk, const float* bv, const float* br, const float* a,
                              const float* g, int64_t B, int F, int D, int H, float dropout_rate, unsigned seed,
                              const float* bn_gamma, const float* bn_mean, const float* bn_rstd, const float* bn_sums,
                              float* dY, float* dX, const float* prev_a, const float* prev_mean, const float* prev_rstd,
                              double* prev_sums, const float* xn_mean, const float* xn_rstd, const float* xn_gamma,
                              const float* xn_beta, int mfma_mode, void* stream) {
    DT_UNSUPPORTED(mfma_mode != DT_AI_F32 && !((mfma_mode == DT_AI_BF16 || mfma_mode == DT_AI_BF16X2) && D == 32), "dt_autoint_bwd: mfma_mode %d (DT_AI_BF16 / DT_AI_BF16X2 need D = 32; D = %d)", mfma_mode, D);
    DT_UNSUPPORTED(!dt_autoint_supported(F, D, H), "dt_autoint_bwd: unsupported shape F=%d D=%d H=%d", F, D, H);
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && a && g && B > 0 && B < (1LL << 31), "dt_autoint_bwd: null pointer / bad batch");
    const int NP = Wr ? 4 : 3;
    const float* Ws[4] = {Wq, Wk, Wv, Wr};
    const float* bs[4] = {bq, bk, bv, br};
    AiW w4;
    DT_REQUIRE(ai_weights(Ws, bs, NP, &w4), "dt_autoint_bwd: null weight pointer");
    unsigned thr; float inv_keep;
    DT_REQUIRE(ai_drop(dropout_rate, &thr, &inv_keep), "dt_autoint_bwd: dropout_rate %f", dropout_rate);
    hipStream_t st = as_stream(stream);
    DT_REQUIRE(!bn_mean || (bn_rstd && bn_sums), "dt_autoint_bwd: incomplete BatchNormalization arguments");
    const AiBn bn{bn_gamma, bn_mean, bn_rstd, bn_sums, bn_sums ? bn_sums + D : nullptr, 1.0f / ((float)B * (float)F), nullptr};
    DT_REQUIRE(!prev_a || (prev_mean && prev_rstd && prev_sums && dX), "dt_autoint_bwd: prev_a needs prev_mean / prev_rstd / prev_sums and dX");
    const AiPrev pv{prev_a, prev_mean, prev_rstd, prev_sums};
    DT_REQUIRE(!xn_mean || xn_rstd, "dt_autoint_bwd: xn_mean needs xn_rstd");
    const AiXn xn{xn_mean, xn_rstd, xn_gamma, xn_beta};
    DT_AI_DISPATCH(k_autoint_bwd, 8, D * (4 * D + kAiPad) + 8 * (32 * (4 * D + kAiPad) + 96) + 8 * 2 * D + 2 * D, x, w4, a, g, (int)B, F, NP, dY,
                   dX, bn, thr, inv_keep, seed, pv, xn);
    return launch_status("dt_autoint_bwd");
}

// dt_autoint_bwd with the four Dense layers' kernel / bias gradients formed inside the same launch (+ one small reduction):
// replaces dt_autoint_bwd(dY) + dt_dense_bwd(x, dY).  gW [NP][D][D], gb [NP][D] (NP = 3 or 4: q | k | v [| residual]) are
// OVERWRITTEN; workspace: dt_autoint_bwd_workspace_bytes(B, D) bytes.  dX may be NULL (input without gradient).
extern "C" int64_t dt_autoint_bwd_workspace_bytes(int64_t B, int D) {
    int64_t blocks = (B + 7) / 8;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    return blocks * (int64_t)(D * 4 * D + 4 * D) * (int64_t)sizeof(float);
}

extern "C" int dt_autoint_bwd_w(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr,
                                const float* bq, const float* bk, const float* bv, const float* br, const float* a,
                                const float* g, int64_t B, int F, int D, int H, float dropout_rate, unsigned seed,
                                const float* bn_gamma, const float* bn_mean, const float* bn_rstd, const float* bn_sums,
                                float* dX, float* gW, float* gb, void* workspace, const float* prev_a,
                                const float* prev_mean, const float* prev_rstd, double* prev_sums, const float* xn_mean,
                                const float* xn_rstd, const float* xn_gamma, const float* xn_beta,
                                const double* bn_sums_f64, float* bn_grads, const float* g_rank1_w, int mfma_mode,
                                void* stream) {
    DT_UNSUPPORTED(mfma_mode != DT_AI_F32 && !((mfma_mode == DT_AI_BF16 || mfma_mode == DT_AI_BF16X2) && D == 32), "dt_autoint_bwd_w: mfma_mode %d (DT_AI_BF16 / DT_AI_BF16X2 need D = 32; D = %d)", mfma_mode, D);
    DT_UNSUPPORTED(!dt_autoint_supported(F, D, H), "dt_autoint_bwd_w: unsupported shape F=%d D=%d H=%d", F, D, H);
    DT_UNSUPPORTED(F > 28, "dt_autoint_bwd_w: F=%d > 28 fields (use dt_autoint_bwd + dt_dense_bwd)", F);
    DT_REQUIRE(gW && gb && workspace, "dt_autoint_bwd_w: null gradient / workspace pointer");
    const int NP = Wr ? 4 : 3;
    const int M = NP * D;
    hipStream_t st = as_stream(stream);
    if (B == 0) {
        hipMemsetAsync(gW, 0, sizeof(float) * D * M, st);
        hipMemsetAsync(gb, 0, sizeof(float) * M, st);
        if (bn_grads) hipMemsetAsync(bn_grads, 0, sizeof(float) * 2 * D, st);
        return launch_status("dt_autoint_bwd_w");
    }
    DT_REQUIRE(x && a && g && B > 0 && B < (1LL << 31), "dt_autoint_bwd_w: null pointer / bad batch");
    const float* Ws[4] = {Wq, Wk, Wv, Wr};
    const float* bs[4] = {bq, bk, bv, br};
    AiW w4;
    DT_REQUIRE(ai_weights(Ws, bs, NP, &w4), "dt_autoint_bwd_w: null weight pointer");
    unsigned thr; float inv_keep;
    DT_REQUIRE(ai_drop(dropout_rate, &thr, &inv_keep), "dt_autoint_bwd_w: dropout_rate %f", dropout_rate);
    DT_REQUIRE(!bn_mean || (bn_rstd && (bn_sums || bn_sums_f64)), "dt_autoint_bwd_w: incomplete BatchNormalization arguments");
    const AiBn bn{bn_gamma, bn_mean, bn_rstd, bn_sums, bn_sums ? bn_sums + D : nullptr, 1.0f / ((float)B * (float)F),
                  bn_mean ? bn_sums_f64 : nullptr};
    DT_REQUIRE(!xn_mean || xn_rstd, "dt_autoint_bwd_w: xn_mean needs xn_rstd");
    const AiXn xn{xn_mean, xn_rstd, xn_gamma, xn_beta};
    float* wpart = static_cast<float*>(workspace);
    DT_REQUIRE(!prev_a || (prev_mean && prev_rstd && prev_sums && dX), "dt_autoint_bwd_w: prev_a needs prev_mean / prev_rstd / prev_sums and dX");
    const AiPrev pv{prev_a, prev_mean, prev_rstd, prev_sums};
    DT_AI_DISPATCH(k_autoint_bwd_w, 8, D * (4 * D + kAiPad) + 8 * (32 * (4 * D + kAiPad) + 96) + 8 * 2 * D + 2 * D, x, w4, a, g, (int)B, F, NP,
                   dX, bn, thr, inv_keep, seed, wpart, pv, xn, g_rank1_w);
    int nparts = (int)((B + 7) / 8);
    if (nparts > 256) nparts = 256;
    const int total = D * M + M;
    hipLaunchKernelGGL(k_autoint_wgrad_reduce, dim3((total + 63) / 64), dim3(1024), 0, st, wpart, nparts, D, M, gW, gb,
                       bn_sums_f64, bn_grads);
    return launch_status("dt_autoint_bwd_w");
}

// dt_autoint_fwd followed by the layer's training-mode BatchNormalization (layers.py:151) in TWO launches instead of four:
// the statistics are block sums written by the attention kernel's epilogue, the second launch finishes them in its
// prologue and normalises.  out_a = relu(attention + residual) (kept for the backward), out_y = BN(out_a); save_mean /
// save_rstd [D] for dt_autoint_bwd*; moving_mean / moving_var updated as dt_bn_train_fwd does.
// workspace: dt_autoint_fwd_bn_workspace_bytes(B, D) bytes.
extern "C" int64_t dt_autoint_fwd_bn_workspace_bytes(int64_t B, int D) {
    int64_t blocks = (B + 3) / 4;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    return (int64_t)sizeof(float) * (D + blocks * 2 * D);
}

extern "C" int dt_autoint_fwd_bn(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wr,
                                 const float* bq, const float* bk, const float* bv, const float* br, int64_t B, int F,
                                 int D, int H, float dropout_rate, unsigned seed, const float* gamma, const float* beta,
                                 float eps, float momentum, float* moving_mean, float* moving_var, float* out_a,
                                 float* out_y, float* save_mean, float* save_rstd, void* workspace, const float* xn_mean,
                                 const float* xn_rstd, const float* xn_gamma, const float* xn_beta, double* zero_sums,
                                 int mfma_mode, void* stream) {
    DT_UNSUPPORTED(mfma_mode != DT_AI_F32 && !((mfma_mode == DT_AI_BF16 || mfma_mode == DT_AI_BF16X2) && D == 32), "dt_autoint_fwd_bn: mfma_mode %d (DT_AI_BF16 / DT_AI_BF16X2 need D = 32; D = %d)", mfma_mode, D);
    DT_UNSUPPORTED(!dt_autoint_supported(F, D, H), "dt_autoint_fwd_bn: unsupported shape F=%d D=%d H=%d", F, D, H);
    DT_REQUIRE(x && out_a && save_mean && save_rstd && workspace && B > 0 && B < (1LL << 31),
               "dt_autoint_fwd_bn: null pointer / bad batch");
    const int NP = Wr ? 4 : 3;
    const float* Ws[4] = {Wq, Wk, Wv, Wr};
    const float* bs[4] = {bq, bk, bv, br};
    AiW w4;
    DT_REQUIRE(ai_weights(Ws, bs, NP, &w4), "dt_autoint_fwd_bn: null weight pointer");
    unsigned thr; float inv_keep;
    DT_REQUIRE(ai_drop(dropout_rate, &thr, &inv_keep), "dt_autoint_fwd_bn: dropout_rate %f", dropout_rate);
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    float* lse = nullptr;
    DT_REQUIRE(!xn_mean || xn_rstd, "dt_autoint_fwd_bn: xn_mean needs xn_rstd");
    const AiXn xn{xn_mean, xn_rstd, xn_gamma, xn_beta};
    DT_AI_DISPATCH(k_autoint_fwd, 4, 4 * 32 * (4 * D + kAiPad) + 2 * D, x, w4, (int)B, F, NP, out_a, lse, thr, inv_keep, seed,
                   (const float*)moving_mean, part, xn);
    int nparts = (int)((B + 3) / 4);
    if (nparts > 512) nparts = 512;
    const int64_t total4 = B * F * (D / 4);
    DT_REQUIRE(total4 < (1LL << 31), "dt_autoint_fwd_bn: tensor too large");
    int blocks = (int)((total4 + 1023) / 1024);
    if (blocks > 256) blocks = 256;
    if (!out_y) blocks = 1;                                  // statistics only: the layer above normalises on load (xn_*)
    hipLaunchKernelGGL(k_autoint_bn_apply, dim3(blocks), dim3(1024), 0, st, out_a, (int)total4, D, part, nparts,
                       1.0f / ((float)B * (float)F), gamma, beta, eps, momentum, moving_mean, moving_var, save_mean,
                       save_rstd, out_y, zero_sums);
    return launch_status("dt_autoint_fwd_bn");
}

// ---- the head of the AutoInt graph (see k_autoint_head_*): a [B, K = F D] un-normalised output of the top interacting layer,
// xn_* its pending BatchNormalization, w [K] / bias [1] (may be NULL) the Dense(1) that consumes the flattened result
extern "C" int64_t dt_autoint_head_workspace_bytes(int64_t B, int K) {
    const int64_t blocks = (B + kAiHeadRows - 1) / kAiHeadRows;
    return (blocks < 1 ? 1 : blocks) * (int64_t)(K + 4) * (int64_t)sizeof(float);
}

extern "C" int dt_autoint_head_fwd(const float* a, const float* w, const float* bias, const float* xn_mean,
                                   const float* xn_rstd, const float* xn_gamma, const float* xn_beta, int64_t B, int K, int D,
                                   float* z, void* stream) {
    DT_UNSUPPORTED(K < 4 || K > 256 * kAiHeadK4 || (K & 3) || D < 4 || D > 64 || 64 % D || K % D, "dt_autoint_head_fwd: K=%d D=%d (K <= 1024, D in {4..64} dividing 64 and K)", K, D);
    if (B == 0) return DT_OK;
    DT_REQUIRE(a && w && z && xn_mean && xn_rstd && B > 0 && B < (1LL << 31), "dt_autoint_head_fwd: null pointer / bad batch");
    hipStream_t st = as_stream(stream);
    const AiXn xn{xn_mean, xn_rstd, xn_gamma, xn_beta};
    int blocks = (int)((B + 3) / 4);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_autoint_head_fwd, dim3(blocks), dim3(256), 0, st, a, w, bias, xn, (int)B, K, D, z);
    return launch_status("dt_autoint_head_fwd");
}

// gz [B] = gradient of the unit's output.  gW [K], gb [1] (may be NULL): OVERWRITTEN.  bn_sums [2 D] doubles: the pending
// normalisation's backward sums for dL/dy = gz w are ADDED (zero them first: dt_autoint_fwd_bn's zero_sums does) — pass them
// as dt_autoint_bwd_w(bn_sums_f64 = bn_sums, g = gz, g_rank1_w = w).
extern "C" int dt_autoint_head_bwd(const float* a, const float* w, const float* gz, const float* xn_mean,
                                   const float* xn_rstd, const float* xn_gamma, const float* xn_beta, int64_t B, int K, int D,
                                   float* gW, float* gb, double* bn_sums, void* workspace, void* stream) {
    DT_UNSUPPORTED(K < 4 || K > 1024 || (K & 3) || D < 4 || D > 64 || 64 % D || K % D, "dt_autoint_head_bwd: K=%d D=%d (K <= 1024, D in {4..64} dividing 64 and K)", K, D);
    DT_REQUIRE(gW && bn_sums && workspace && w && xn_mean && xn_rstd, "dt_autoint_head_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (B == 0) {
        hipMemsetAsync(gW, 0, sizeof(float) * K, st);
        if (gb) hipMemsetAsync(gb, 0, sizeof(float), st);
        return launch_status("dt_autoint_head_bwd");
    }
    DT_REQUIRE(a && gz && B > 0 && B < (1LL << 31), "dt_autoint_head_bwd: null pointer / bad batch");
    float* part = static_cast<float*>(workspace);
    const int blocks = (int)((B + kAiHeadRows - 1) / kAiHeadRows);
    hipLaunchKernelGGL(k_autoint_head_bwd, dim3(blocks), dim3(256), 0, st, a, gz, (int)B, K, part);
    const AiXn xn{xn_mean, xn_rstd, xn_gamma, xn_beta};
    hipLaunchKernelGGL(k_autoint_head_finish, dim3((K + 63) / 64), dim3(1024), 0, st, part, blocks, K, D, w, xn, gW, gb, bn_sums);
    return launch_status("dt_autoint_head_bwd");
}
