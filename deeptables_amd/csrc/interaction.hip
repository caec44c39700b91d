// interaction.hip — the remaining pairwise field-interaction layers of deeptables/models/layers.py
// (SURVEY §8 f3): AFM attention pooling, BilinearInteraction (FiBiNet), SENET re-weighting.
//
//   AFM.call                  layers.py:789-807   softmax over the P = F(F-1)/2 pair interactions
//   BilinearInteraction.call  layers.py:363-377   (v_i . W) * v_j  for every pair
//   SENET.call                layers.py:291-302   squeeze (mean|max over D) ... scale
//
// Same design as product.hip: a batch row's [F, D] block is staged once in LDS and every pair is formed from
// it; the reference materialises p, q = [B, P, D] (and the AFM attention tensor [B, P, H]).  Parameter
// gradients are accumulated per block (persistent blocks looping over rows) and flushed once with atomics.
#include "common.h"

namespace dt {

__device__ __forceinline__ int pair_of(int i, int j, int F) { return i * F - (i * (i + 1)) / 2 + (j - i - 1); }

__device__ __forceinline__ float block_sum(float v, float* scratch /* >= 4 floats */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, scratch[w]);
    return t;
}

// ---------------------------------------------------------------------------------------------
// AFM:  bi[p] = x_i * x_j ; a[p,h] = act(bi[p] . Wa[:,h] + ba[h]) ; logit[p] = a[p] . pv ;
//       score = softmax_p(logit) ; out[d] = sum_p score[p] bi[p,d]
// Persistent blocks, one batch row at a time in LDS.  The attention weights live in LDS padded to a compile-time
// stride HMAX (zeros beyond H), so the inner loops are guard-free and read 16 bytes at a time.
// ---------------------------------------------------------------------------------------------
template <bool kAnyAct>
__device__ __forceinline__ float afm_act(float v, int act) {
    if (kAnyAct) return act_apply(v, act);
    return act == DT_ACT_RELU ? fmaxf(v, 0.f) : v;
}
template <bool kAnyAct>
__device__ __forceinline__ float afm_act_grad(float y, int act) {
    if (kAnyAct) return act_grad_from_y(y, act);
    return (act == DT_ACT_RELU && !(y > 0.f)) ? 0.f : 1.f;
}

template <int HMAX>
__device__ __forceinline__ void afm_att_preact(const float* __restrict__ xi, const float* __restrict__ xj,
                                               const float* __restrict__ wa, const float* __restrict__ bb, int D,
                                               float (&acc)[HMAX]) {
#pragma unroll
    for (int h = 0; h < HMAX; ++h) acc[h] = bb[h];
    for (int d = 0; d < D; ++d) {
        const float bi = xi[d] * xj[d];
        const float4* w4 = reinterpret_cast<const float4*>(wa + d * HMAX);
#pragma unroll
        for (int q = 0; q < HMAX / 4; ++q) {
            const float4 w = w4[q];
            acc[4 * q + 0] += bi * w.x; acc[4 * q + 1] += bi * w.y;
            acc[4 * q + 2] += bi * w.z; acc[4 * q + 3] += bi * w.w;
        }
    }
}

template <int HMAX, bool kAnyAct>  // kAnyAct = false: linear / relu only (keeps libm code out of the unrolled loops)
__global__ __launch_bounds__(256) void k_afm_fwd(const float* __restrict__ x, const float* __restrict__ Wa,
                                                 const float* __restrict__ ba, const float* __restrict__ pv,
                                                 int act, int B, int F, int D, int H, float* __restrict__ out,
                                                 float* __restrict__ score_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    float* wa = lds;                   // [D][HMAX]   (16-byte aligned rows)
    float* bb = wa + D * HMAX;         // [HMAX]
    float* pp = bb + HMAX;             // [HMAX]
    float* red = pp + HMAX;            // [4][16]
    float* scratch = red + 64;         // [8]
    float* xr = scratch + 8;           // [F*D]
    float* sc = xr + F * D;            // [P]
    short* pi = reinterpret_cast<short*>(sc + P);
    short* pj = pi + P;
    for (int e = threadIdx.x; e < D * HMAX; e += blockDim.x) {
        const int d = e / HMAX, h = e - d * HMAX;
        wa[e] = h < H ? Wa[d * H + h] : 0.f;
    }
    for (int e = threadIdx.x; e < HMAX; e += blockDim.x) {
        bb[e] = (ba && e < H) ? ba[e] : 0.f;
        pp[e] = e < H ? pv[e] : 0.f;
    }
    for (int i = threadIdx.x; i < F; i += blockDim.x)
        for (int j = i + 1; j < F; ++j) { const int p = pair_of(i, j, F); pi[p] = (short)i; pj[p] = (short)j; }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = threadIdx.x; e < F * D; e += blockDim.x) xr[e] = x[(int64_t)b * F * D + e];
        __syncthreads();
        float lmax = -INFINITY;
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            float acc[HMAX];
            afm_att_preact<HMAX>(xr + pi[p] * D, xr + pj[p] * D, wa, bb, D, acc);
            float lg = 0.f;
#pragma unroll
            for (int h = 0; h < HMAX; ++h) lg += afm_act<kAnyAct>(acc[h], act) * pp[h];
            sc[p] = lg;
            lmax = fmaxf(lmax, lg);
        }
        lmax = block_max(lmax, scratch);
        float lsum = 0.f;
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            const float e = expf(sc[p] - lmax);
            sc[p] = e;
            lsum += e;
        }
        lsum = block_sum(lsum, scratch);
        const float inv = 1.0f / lsum;
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            const float s = sc[p] * inv;
            sc[p] = s;
            if (score_out) score_out[(int64_t)b * P + p] = s;
        }
        __syncthreads();
        // pooled sum over pairs: every thread folds ITS pairs into a private [16] vector, the vectors meet through wave
        // shuffles and a [waves][16] LDS buffer (a d-per-thread loop over all P pairs leaves 240 of 256 lanes idle)
        for (int d0 = 0; d0 < D; d0 += 16) {
            float part[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) part[k] = 0.f;
            for (int p = threadIdx.x; p < P; p += blockDim.x) {
                const float s = sc[p];
                const float* xi = xr + pi[p] * D + d0;
                const float* xj = xr + pj[p] * D + d0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (d0 + k < D) part[k] += s * xi[k] * xj[k];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) part[k] = wave_sum(part[k]);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) red[(threadIdx.x >> 6) * 16 + k] = part[k];
            }
            __syncthreads();
            if (threadIdx.x < 16 && d0 + threadIdx.x < D) {
                float o = 0.f;
                for (int w = 0; w < (int)(blockDim.x >> 6); ++w) o += red[w * 16 + threadIdx.x];
                out[(int64_t)b * D + d0 + threadIdx.x] = o;
            }
        }
    }
}

// Backward.  Per row: (1) S = sum_p score_p dscore_p; (2) per pair: recompute the attention pre-activations, form
// datt[p,:] and dbi[p,:] in LDS; (3) grad_Wa += bi^T datt as 4x4 register blocks — thread = (d-block, h-block, pair
// group), accumulators persist over the block's rows; (4) grad_x from dbi.  grad_pv / grad_ba accumulate per thread.
template <int HMAX, bool kAnyAct>
__global__ __launch_bounds__(256) void k_afm_bwd(const float* __restrict__ x, const float* __restrict__ Wa,
                                                 const float* __restrict__ ba, const float* __restrict__ pv,
                                                 const float* __restrict__ score, const float* __restrict__ gout,
                                                 int act, int B, int F, int D, int H, float* __restrict__ gx,
                                                 float* __restrict__ gWa, float* __restrict__ gba,
                                                 float* __restrict__ gpv) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    const int DP4 = (D + 3) & ~3;          // bi / dbi rows padded to 16 bytes
    float* wa = lds;                       // [D][HMAX]
    float* bb = wa + D * HMAX;             // [HMAX]
    float* pp = bb + HMAX;                 // [HMAX]
    float* accb = pp + HMAX;               // [HMAX]
    float* accp = accb + HMAX;             // [HMAX]
    float* datt = accp + HMAX;             // [P][HMAX]
    float* bi = datt + P * HMAX;           // [P][DP4]
    float* dbi = bi + P * DP4;             // [P][DP4]
    float* accW = dbi + P * DP4;           // [D][HMAX]
    float* scratch = accW + D * HMAX;      // [8]
    float* xr = scratch + 8;               // [F*D]
    float* gr = xr + F * D;                // [D]
    short* pi = reinterpret_cast<short*>(gr + D);
    short* pj = pi + P;
    for (int e = threadIdx.x; e < D * HMAX; e += blockDim.x) {
        const int d = e / HMAX, h = e - d * HMAX;
        wa[e] = h < H ? Wa[d * H + h] : 0.f;
        accW[e] = 0.f;
    }
    for (int e = threadIdx.x; e < HMAX; e += blockDim.x) {
        bb[e] = (ba && e < H) ? ba[e] : 0.f;
        pp[e] = e < H ? pv[e] : 0.f;
        accb[e] = 0.f;
        accp[e] = 0.f;
    }
    for (int i = threadIdx.x; i < F; i += blockDim.x)
        for (int j = i + 1; j < F; ++j) { const int p = pair_of(i, j, F); pi[p] = (short)i; pj[p] = (short)j; }
    // grad_Wa blocking: thread = (4x4 block of (d,h), pair group); needs D % 4 == 0
    const int nblk = (DP4 / 4) * (HMAX / 4);
    const bool blocked = (D % 4 == 0) && nblk <= 256;
    const int pgroups = blocked ? 256 / nblk : 0;
    const int blk = threadIdx.x % (blocked ? nblk : 1), pg = blocked ? threadIdx.x / nblk : 0;
    const int db = blk / (HMAX / 4), hb = blk % (HMAX / 4);
    float wacc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) wacc[k] = 0.f;
    float my_dp[HMAX], my_db[HMAX];
#pragma unroll
    for (int h = 0; h < HMAX; ++h) { my_dp[h] = 0.f; my_db[h] = 0.f; }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = threadIdx.x; e < F * D; e += blockDim.x) xr[e] = x[(int64_t)b * F * D + e];
        for (int e = threadIdx.x; e < D; e += blockDim.x) gr[e] = gout[(int64_t)b * D + e];
        __syncthreads();
        // (1) bi[p,:] and S = sum_q score_q (g . bi_q)
        float part = 0.f;
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            const float* xi = xr + pi[p] * D;
            const float* xj = xr + pj[p] * D;
            float ds = 0.f;
            for (int d = 0; d < D; ++d) {
                const float v = xi[d] * xj[d];
                bi[p * DP4 + d] = v;
                ds += gr[d] * v;
            }
            part += score[(int64_t)b * P + p] * ds;
        }
        const float S = block_sum(part, scratch);
        // (2) datt, dbi
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            const float sp = score[(int64_t)b * P + p];
            float acc[HMAX];
            afm_att_preact<HMAX>(xr + pi[p] * D, xr + pj[p] * D, wa, bb, D, acc);
            float ds = 0.f;
            for (int d = 0; d < D; ++d) ds += gr[d] * bi[p * DP4 + d];
            const float dlogit = sp * (ds - S);
#pragma unroll
            for (int h = 0; h < HMAX; ++h) {
                const float a = afm_act<kAnyAct>(acc[h], act);
                const float da = dlogit * pp[h] * afm_act_grad<kAnyAct>(a, act);
                my_dp[h] += dlogit * a;
                my_db[h] += da;
                acc[h] = da;
            }
#pragma unroll
            for (int q = 0; q < HMAX / 4; ++q)
                *reinterpret_cast<float4*>(datt + p * HMAX + 4 * q) =
                    make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            for (int d = 0; d < D; ++d) {
                const float4* w4 = reinterpret_cast<const float4*>(wa + d * HMAX);
                float v = sp * gr[d];
#pragma unroll
                for (int q = 0; q < HMAX / 4; ++q) {
                    const float4 w = w4[q];
                    v += acc[4 * q] * w.x + acc[4 * q + 1] * w.y + acc[4 * q + 2] * w.z + acc[4 * q + 3] * w.w;
                }
                dbi[p * DP4 + d] = v;
            }
        }
        __syncthreads();
        // (3) grad_Wa[d,h] += sum_p bi[p,d] datt[p,h]
        if (blocked) {
            if (pg < pgroups) {
                for (int p = pg; p < P; p += pgroups) {
                    const float4 bv = *reinterpret_cast<const float4*>(bi + p * DP4 + 4 * db);
                    const float4 av = *reinterpret_cast<const float4*>(datt + p * HMAX + 4 * hb);
                    wacc[0] += bv.x * av.x; wacc[1] += bv.x * av.y; wacc[2] += bv.x * av.z; wacc[3] += bv.x * av.w;
                    wacc[4] += bv.y * av.x; wacc[5] += bv.y * av.y; wacc[6] += bv.y * av.z; wacc[7] += bv.y * av.w;
                    wacc[8] += bv.z * av.x; wacc[9] += bv.z * av.y; wacc[10] += bv.z * av.z; wacc[11] += bv.z * av.w;
                    wacc[12] += bv.w * av.x; wacc[13] += bv.w * av.y; wacc[14] += bv.w * av.z; wacc[15] += bv.w * av.w;
                }
            }
        } else {
            for (int e = threadIdx.x; e < D * HMAX; e += blockDim.x) {
                const int d = e / HMAX, h = e - d * HMAX;
                float v = 0.f;
                for (int p = 0; p < P; ++p) v += bi[p * DP4 + d] * datt[p * HMAX + h];
                accW[e] += v;
            }
        }
        // (4) grad_x[f,d] = sum_{o != f} dbi[p(f,o), d] x[o,d]
        for (int e = threadIdx.x; e < F * D; e += blockDim.x) {
            const int f = e / D, d = e - f * D;
            float v = 0.f;
            for (int o = 0; o < F; ++o) {
                if (o == f) continue;
                const int p = o > f ? pair_of(f, o, F) : pair_of(o, f, F);
                v += dbi[p * DP4 + d] * xr[o * D + d];
            }
            gx[(int64_t)b * F * D + e] = v;
        }
    }
    __syncthreads();
    if (blocked && pg < pgroups) {
#pragma unroll
        for (int k = 0; k < 16; ++k) atomicAdd(&accW[(4 * db + (k >> 2)) * HMAX + 4 * hb + (k & 3)], wacc[k]);
    }
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        const float a = wave_sum(my_dp[h]), c = wave_sum(my_db[h]);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&accp[h], a);
            atomicAdd(&accb[h], c);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D * H; e += blockDim.x) {
        const int d = e / H, h = e - d * H;
        atomicAdd(gWa + e, accW[d * HMAX + h]);
    }
    for (int e = threadIdx.x; e < H; e += blockDim.x) {
        if (gba) atomicAdd(gba + e, accb[e]);
        atomicAdd(gpv + e, accp[e]);
    }
}

// ---------------------------------------------------------------------------------------------
// BilinearInteraction: out[b,p,:] = (x_i . W_q) * x_j ,  q = p ('field_interaction'), i ('field_each'), 0 ('field_all')
//   wave = one pair x 64 batch rows (lane = row); W_q [D,D] broadcast from LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bilinear_fwd(const float* __restrict__ x, const float* __restrict__ W,
                                                     int wtype, int B, int F, int D, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    const int DP = D + 1;
    float* Wq = lds;             // [D][D]
    float* xi = lds + D * D;     // [64][DP]
    float* xj = xi + 64 * DP;    // [64][DP]
    const int lane = threadIdx.x, p = blockIdx.y, b0 = blockIdx.x * 64;
    int i = 0, rem = p;
    while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const int q = wtype == 0 ? p : (wtype == 1 ? i : 0);
    for (int e = lane; e < D * D; e += 64) Wq[e] = W[(int64_t)q * D * D + e];
    for (int e = lane; e < 64 * D; e += 64) {
        const int r = e / D, d = e - r * D;
        const int b = b0 + r;
        xi[r * DP + d] = b < B ? x[((int64_t)b * F + i) * D + d] : 0.f;
        xj[r * DP + d] = b < B ? x[((int64_t)b * F + j) * D + d] : 0.f;
    }
    __syncthreads();
    if (b0 + lane >= B) return;
    for (int d2 = 0; d2 < D; ++d2) {
        float u = 0.f;
        for (int d = 0; d < D; ++d) u += xi[lane * DP + d] * Wq[d * D + d2];
        out[((int64_t)(b0 + lane) * P + p) * D + d2] = u * xj[lane * DP + d2];
    }
}

// grad_x for field f (all partners), lane = row
__global__ __launch_bounds__(64) void k_bilinear_bwd_x(const float* __restrict__ x, const float* __restrict__ W,
                                                       const float* __restrict__ gout, int wtype, int B, int F,
                                                       int D, float* __restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    const int DP = D + 1;
    float* Wq = lds;              // [D][D]
    float* xo = lds + D * D;      // [64][DP] partner rows
    float* xf = xo + 64 * DP;     // [64][DP] own rows
    float* acc = xf + 64 * DP;    // [64][DP]
    const int lane = threadIdx.x, f = blockIdx.y, b0 = blockIdx.x * 64;
    const int b = b0 + lane;
    for (int e = lane; e < 64 * D; e += 64) {
        const int r = e / D, d = e - r * D;
        xf[r * DP + d] = (b0 + r) < B ? x[((int64_t)(b0 + r) * F + f) * D + d] : 0.f;
    }
    for (int d = 0; d < D; ++d) acc[lane * DP + d] = 0.f;
    for (int o = 0; o < F; ++o) {
        if (o == f) continue;
        const bool f_is_i = f < o;
        const int i = f_is_i ? f : o;
        const int p = f_is_i ? pair_of(f, o, F) : pair_of(o, f, F);
        const int q = wtype == 0 ? p : (wtype == 1 ? i : 0);
        __syncthreads();
        for (int e = lane; e < D * D; e += 64) Wq[e] = W[(int64_t)q * D * D + e];
        for (int e = lane; e < 64 * D; e += 64) {
            const int r = e / D, d = e - r * D;
            xo[r * DP + d] = (b0 + r) < B ? x[((int64_t)(b0 + r) * F + o) * D + d] : 0.f;
        }
        __syncthreads();
        if (b >= B) continue;
        const float* g = gout + ((int64_t)b * P + p) * D;
        if (f_is_i) {   // out[d2] = (sum_d x_f[d] W[d,d2]) x_o[d2]  ->  grad x_f[d] += sum_d2 g[d2] x_o[d2] W[d,d2]
            for (int d = 0; d < D; ++d) {
                float v = 0.f;
                for (int d2 = 0; d2 < D; ++d2) v += g[d2] * xo[lane * DP + d2] * Wq[d * D + d2];
                acc[lane * DP + d] += v;
            }
        } else {        // f = j: grad x_f[d2] += g[d2] * (sum_d x_o[d] W[d,d2])
            for (int d2 = 0; d2 < D; ++d2) {
                float u = 0.f;
                for (int d = 0; d < D; ++d) u += xo[lane * DP + d] * Wq[d * D + d2];
                acc[lane * DP + d2] += g[d2] * u;
            }
        }
    }
    if (b < B)
        for (int d = 0; d < D; ++d) gx[((int64_t)b * F + f) * D + d] = acc[lane * DP + d];
}

// grad_W[q][d,d2] += sum_b x_i[b,d] g[b,p,d2] x_j[b,d2];  block = (pair, batch split), thread = (d,d2) entries
__global__ __launch_bounds__(256) void k_bilinear_bwd_w(const float* __restrict__ x, const float* __restrict__ gout,
                                                        int wtype, int B, int F, int D, int rows_per_split,
                                                        float* __restrict__ gW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    constexpr int TB = 64;
    float* xi = lds;              // [TB][D]
    float* gj = lds + TB * D;     // [TB][D]   g * x_j
    const int p = blockIdx.x;
    int i = 0, rem = p;
    while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const int q = wtype == 0 ? p : (wtype == 1 ? i : 0);
    const int r_begin = blockIdx.y * rows_per_split;
    const int r_end = min(B, r_begin + rows_per_split);
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int r0 = r_begin; r0 < r_end; r0 += TB) {
        __syncthreads();
        for (int e = threadIdx.x; e < TB * D; e += blockDim.x) {
            const int r = e / D, d = e - r * D;
            const int b = r0 + r;
            const bool ok = b < r_end;
            xi[e] = ok ? x[((int64_t)b * F + i) * D + d] : 0.f;
            gj[e] = ok ? gout[((int64_t)b * P + p) * D + d] * x[((int64_t)b * F + j) * D + d] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = threadIdx.x + k * 256;
            if (t < D * D) {
                const int d = t / D, d2 = t - d * D;
                float sacc = 0.f;
                // (unrolled four deep only: with all 64 iterations unrolled in each of the 16 k blocks the scheduler hoisted the
                // LDS reads of several blocks and spilled 572 registers — profiles/r04_kernel_resources.txt)
#pragma unroll 4
                for (int r = 0; r < TB; ++r) sacc += xi[r * D + d] * gj[r * D + d2];
                acc[k] += sacc;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = threadIdx.x + k * 256;
        if (t < D * D) atomicAdd(&gW[(int64_t)q * D * D + t], acc[k]);
    }
}

// ---------------------------------------------------------------------------------------------
// BilinearInteraction, D == 16: the three per-pair 16x16 products on v_mfma_f32_16x16x4_f32 (exact fp32).
//   A operand lane l: A[m = l&15][k = l>>4]; B operand: B[k = l>>4][n = l&15]; C/D: col = l&15, row = 4*(l>>4) + r.
// A wave owns a 16-row batch tile; pairs are dealt round-robin to the block's waves.
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void pair_ij(int p, int F, int& i, int& j) {
    i = 0;
    int rem = p;
    while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
    j = i + 1 + rem;
}
__device__ __forceinline__ int bil_q(int wtype, int p, int i) { return wtype == 0 ? p : (wtype == 1 ? i : 0); }

// forward, pair-major: out[b,p,:] = (x_i W_q) * x_j.  grid (P, splits); W_q sits in 4 registers per lane (the MFMA B
// operand) while the block's waves walk the 16-row tiles of their split.
__global__ __launch_bounds__(256) void k_bil16_fwd(const float* __restrict__ x, const float* __restrict__ W, int wtype,
                                                   int B, int F, int tiles_per_split, float* __restrict__ out) {
    constexpr int D = 16;
    const int P = F * (F - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x;
    int i, j;
    pair_ij(p, F, i, j);
    const float* Wq = W + (int64_t)bil_q(wtype, p, i) * D * D;
    float wb[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) wb[s4] = Wq[(kq + 4 * s4) * D + m];
    const int ntiles = (B + 15) / 16;
    const int tile0 = blockIdx.y * tiles_per_split, tile1 = min(ntiles, tile0 + tiles_per_split);
    for (int t = tile0 + wave; t < tile1; t += 4) {
        const int b0 = t * 16;
        const int brow = b0 + m;
        const bool row_ok = brow < B;
        float xa[4], xj4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xa[s4] = row_ok ? x[((int64_t)brow * F + i) * D + kq + 4 * s4] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + 4 * kq + r;
            xj4[r] = b < B ? x[((int64_t)b * F + j) * D + m] : 0.f;
        }
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) u = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s4], wb[s4], u, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {               // C layout: row 4*kq + r, col m
            const int b = b0 + 4 * kq + r;
            if (b < B) out[((int64_t)b * P + p) * D + m] = u[r] * xj4[r];
        }
    }
}

// grad_x.  grid (ceil(B/16)), block 256; the tile's grad rows [16][F][16] accumulate in LDS.
//   u = x_i W ; grad x_j = g * u ; t = g * x_j ; grad x_i = t W^T
struct Bil16Regs {          // everything one (pair, 16-row tile) step reads from memory
    float xi[4], tj[4], wf[4], wt[4], gc[4];
    int i, j;
};

__device__ __forceinline__ void bil16_load(Bil16Regs& r, const float* __restrict__ x, const float* __restrict__ W,
                                           const float* __restrict__ gout, const short* __restrict__ pij, int wtype,
                                           int p, int P, int F, int B, int b0, int m, int kq) {
    constexpr int D = 16;
    r.i = pij[2 * p];
    r.j = pij[2 * p + 1];
    const float* Wq = W + (int64_t)bil_q(wtype, p, r.i) * D * D;
    const int brow = b0 + m;
    const bool row_ok = brow < B;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int k = kq + 4 * s4;
        r.xi[s4] = row_ok ? x[((int64_t)brow * F + r.i) * D + k] : 0.f;
        r.tj[s4] = row_ok ? gout[((int64_t)brow * P + p) * D + k] * x[((int64_t)brow * F + r.j) * D + k] : 0.f;
        r.wf[s4] = Wq[k * D + m];      // forward:  B = W[k][n = m]
        r.wt[s4] = Wq[m * D + k];      // grad x_i: B = W^T[k = d2][n = d] = W[d = m][d2 = k]
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = b0 + 4 * kq + q;
        r.gc[q] = b < B ? gout[((int64_t)b * P + p) * D + m] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_bil16_bwd_x(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ gout, int wtype, int B, int F,
                                                     float* __restrict__ gx) {
    constexpr int D = 16;
    extern __shared__ __attribute__((aligned(16))) float acc[];     // [4 waves][16][F*D] + pair table
    const int P = F * (F - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int FD = F * D;
    short* pij = reinterpret_cast<short*>(acc + 4 * 16 * FD);       // [P][2]
    for (int e = threadIdx.x; e < 4 * 16 * FD; e += blockDim.x) acc[e] = 0.f;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        int i, j;
        pair_ij(p, F, i, j);
        pij[2 * p] = (short)i;
        pij[2 * p + 1] = (short)j;
    }
    __syncthreads();
    // Each wave owns a contiguous range of pairs and a PRIVATE copy of the tile's gradient rows, so its updates are
    // plain LDS read-modify-writes (LDS atomics issue at ~0.5 lane/clk and were the whole cost of this kernel);
    // combinations order keeps field i constant over long runs, so grad x_i accumulates in registers and is flushed
    // when i changes.  The next pair's loads are in flight while this pair's MFMAs run.
    float* mine = acc + (size_t)wave * 16 * FD;
    const int p_begin = (int)((int64_t)P * wave / 4), p_end = (int)((int64_t)P * (wave + 1) / 4);
    Bil16Regs cur, nxt;
    if (p_begin < p_end) bil16_load(cur, x, W, gout, pij, wtype, p_begin, P, F, B, b0, m, kq);
    f32x4 dxi = {0.f, 0.f, 0.f, 0.f};
    int run_i = p_begin < p_end ? cur.i : -1;
    for (int p = p_begin; p < p_end; ++p) {
        if (p + 1 < p_end) bil16_load(nxt, x, W, gout, pij, wtype, p + 1, P, F, B, b0, m, kq);
        if (cur.i != run_i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(4 * kq + r) * FD + run_i * D + m] += dxi[r];
            dxi = f32x4{0.f, 0.f, 0.f, 0.f};
            run_i = cur.i;
        }
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            u = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.xi[s4], cur.wf[s4], u, 0, 0, 0);
            dxi = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.tj[s4], cur.wt[s4], dxi, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(4 * kq + r) * FD + cur.j * D + m] += cur.gc[r] * u[r];
        cur = nxt;
    }
    if (run_i >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(4 * kq + r) * FD + run_i * D + m] += dxi[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * FD; e += blockDim.x) {
        const int row = e / FD;
        if (b0 + row < B)
            gx[(int64_t)(b0 + row) * FD + (e - row * FD)] =
                (acc[e] + acc[16 * FD + e]) + (acc[2 * 16 * FD + e] + acc[3 * 16 * FD + e]);
    }
}

// grad_W[q] += x_i^T (g * x_j).   grid (P, splits), block 256: the block's waves stride over the 16-row tiles of the
// split; A = x_i^T: A[m = d][k = row], B = t[k = row][n = d2]; one 16x16 accumulator per wave, met in LDS.
__global__ __launch_bounds__(256) void k_bil16_bwd_w(const float* __restrict__ x, const float* __restrict__ gout,
                                                     int wtype, int B, int F, int tiles_per_split,
                                                     float* __restrict__ gW) {
    constexpr int D = 16;
    __shared__ float red[4][256];
    const int P = F * (F - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x;
    int i, j;
    pair_ij(p, F, i, j);
    const int tile0 = blockIdx.y * tiles_per_split;
    const int ntiles = (B + 15) / 16;
    const int tile1 = min(ntiles, tile0 + tiles_per_split);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int t = tile0 + wave; t < tile1; t += 4) {
        const int b0 = t * 16;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int b = b0 + kq + 4 * s4;          // k index = batch row
            float a = 0.f, tv = 0.f;
            if (b < B) {
                a = x[((int64_t)b * F + i) * D + m];                                  // x_i[b][d = m]
                tv = gout[((int64_t)b * P + p) * D + m] * x[((int64_t)b * F + j) * D + m];   // t[b][d2 = m]
            }
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tv, c, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * kq + r) * D + m] = c[r];      // dW[d = row][d2 = col]
    __syncthreads();
    const int e = threadIdx.x;
    const float v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    atomicAdd(&gW[(int64_t)bil_q(wtype, p, i) * D * D + e], v);
}

// ---------------------------------------------------------------------------------------------
// SENET: z[b,f] = mean_d | max_d x[b,f,d]  ;  v[b,f,d] = x[b,f,d] * a[b,f]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_field_pool_fwd(const float* __restrict__ x, int64_t n_fields, int D,
                                                        int use_max, float* __restrict__ z, int* __restrict__ arg) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_fields) return;
    const float* p = x + t * D;
    if (use_max) {
        float m = p[0];
        int am = 0;
        for (int d = 1; d < D; ++d)
            if (p[d] > m) { m = p[d]; am = d; }
        z[t] = m;
        if (arg) arg[t] = am;
    } else {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += p[d];
        z[t] = s / (float)D;
    }
}

__global__ __launch_bounds__(256) void k_field_pool_bwd(const float* __restrict__ gz, const int* __restrict__ arg,
                                                        int64_t total, int D, int use_max, float* __restrict__ gx) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = t / D;
        const int d = (int)(t - f * D);
        gx[t] = use_max ? (arg[f] == d ? gz[f] : 0.f) : gz[f] / (float)D;
    }
}

__global__ __launch_bounds__(256) void k_field_scale(const float* __restrict__ x, const float* __restrict__ a,
                                                     int64_t total, int D, float* __restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = x[t] * a[t / D];
}

// grad_a[b,f] = sum_d g[b,f,d] x[b,f,d]
__global__ __launch_bounds__(256) void k_field_scale_bwd_a(const float* __restrict__ x, const float* __restrict__ g,
                                                           int64_t n_fields, int D, float* __restrict__ ga) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_fields) return;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += g[t * D + d] * x[t * D + d];
    ga[t] = s;
}

static int ew_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return b < 1 ? 1 : (int)b;
}

}  // namespace dt

using namespace dt;

static int afm_hmax(int H) { return H <= 16 ? 16 : (H <= 32 ? 32 : 64); }
static size_t afm_lds_fwd(int F, int D, int H) {
    const int P = F * (F - 1) / 2, HM = afm_hmax(H);
    return ((size_t)D * HM + 2 * HM + 64 + 8 + (size_t)F * D + P) * sizeof(float) + 2 * (size_t)P * sizeof(short) + 16;
}
static size_t afm_lds_bwd(int F, int D, int H) {
    const int P = F * (F - 1) / 2, HM = afm_hmax(H), DP4 = (D + 3) & ~3;
    return (2 * (size_t)D * HM + 4 * HM + (size_t)P * HM + 2 * (size_t)P * DP4 + 8 + (size_t)F * D + D) * sizeof(float) +
           2 * (size_t)P * sizeof(short) + 16;
}

extern "C" int dt_afm_fwd(const float* x, const float* Wa, const float* ba, const float* pv, int act, int B, int F,
                          int D, int H, float* out, float* score, void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0 && H > 0, "dt_afm_fwd: bad sizes B=%d F=%d D=%d H=%d", B, F, D, H);
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && Wa && pv && out, "dt_afm_fwd: null pointer");
    DT_UNSUPPORTED(H > 64, "dt_afm_fwd: attention factor %d > 64", H);
    const size_t lds = afm_lds_fwd(F, D, H);
    DT_UNSUPPORTED(lds > 150 * 1024, "dt_afm_fwd: needs %zu B of LDS", lds);
    int blocks = B < 2048 ? B : 2048;
    hipStream_t st = as_stream(stream);
    const bool simple_act = act == DT_ACT_LINEAR || act == DT_ACT_RELU;
#define DT_AFM_F(HM)                                                                                          \
    do {                                                                                                      \
        auto kern = simple_act ? k_afm_fwd<HM, false> : k_afm_fwd<HM, true>;                                  \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, x, Wa, ba, pv, act, B, F, D, H, out,       \
                           score);                                                                            \
    } while (0)
    if (H <= 16) DT_AFM_F(16); else if (H <= 32) DT_AFM_F(32); else DT_AFM_F(64);
#undef DT_AFM_F
    return launch_status("dt_afm_fwd");
}

extern "C" int dt_afm_bwd(const float* x, const float* Wa, const float* ba, const float* pv, const float* score,
                          const float* grad_out, int act, int B, int F, int D, int H, float* grad_x,
                          float* grad_Wa, float* grad_ba, float* grad_pv, void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0 && H > 0, "dt_afm_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && Wa && pv && score && grad_out && grad_x && grad_Wa && grad_pv, "dt_afm_bwd: null pointer");
    DT_UNSUPPORTED(H > 64, "dt_afm_bwd: attention factor %d > 64", H);
    const size_t lds = afm_lds_bwd(F, D, H);
    DT_UNSUPPORTED(lds > 150 * 1024, "dt_afm_bwd: needs %zu B of LDS", lds);
    int blocks = B < 512 ? B : 512;
    hipStream_t st = as_stream(stream);
    const bool simple_act = act == DT_ACT_LINEAR || act == DT_ACT_RELU;
#define DT_AFM_B(HM)                                                                                          \
    do {                                                                                                      \
        auto kern = simple_act ? k_afm_bwd<HM, false> : k_afm_bwd<HM, true>;                                  \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, x, Wa, ba, pv, score, grad_out, act, B, F, \
                           D, H, grad_x, grad_Wa, grad_ba, grad_pv);                                          \
    } while (0)
    if (H <= 16) DT_AFM_B(16); else if (H <= 32) DT_AFM_B(32); else DT_AFM_B(64);
#undef DT_AFM_B
    return launch_status("dt_afm_bwd");
}

extern "C" int dt_bilinear_fwd(const float* x, const float* W, int wtype, int B, int F, int D, float* out,
                               void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0 && wtype >= 0 && wtype <= 2, "dt_bilinear_fwd: bad arguments");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && W && out, "dt_bilinear_fwd: null pointer");
    if (D == 16) {
        const int ntiles = ceil_div(B, 16);
        int splits16 = ntiles >= 64 ? 8 : 1;
        const int tps = ceil_div(ntiles, splits16);
        splits16 = ceil_div(ntiles, tps);
        hipLaunchKernelGGL(k_bil16_fwd, dim3(F * (F - 1) / 2, splits16), dim3(256), 0, as_stream(stream), x, W, wtype, B,
                           F, tps, out);
        return launch_status("dt_bilinear_fwd");
    }
    const size_t lds = ((size_t)D * D + 2 * 64 * (D + 1)) * sizeof(float);
    DT_UNSUPPORTED(lds > 64 * 1024, "dt_bilinear_fwd: D=%d too large", D);
    hipLaunchKernelGGL(k_bilinear_fwd, dim3(ceil_div(B, 64), F * (F - 1) / 2), dim3(64), lds, as_stream(stream), x, W,
                       wtype, B, F, D, out);
    return launch_status("dt_bilinear_fwd");
}

extern "C" int dt_bilinear_bwd(const float* x, const float* W, const float* grad_out, int wtype, int B, int F,
                               int D, float* grad_x, float* grad_W, void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0 && wtype >= 0 && wtype <= 2, "dt_bilinear_bwd: bad arguments");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && W && grad_out && grad_x && grad_W, "dt_bilinear_bwd: null pointer");
    DT_UNSUPPORTED(D > 64, "dt_bilinear_bwd: D=%d > 64", D);
    hipStream_t st = as_stream(stream);
    if (D == 16 && (size_t)4 * 16 * F * 16 * sizeof(float) + (size_t)F * F * 2 <= 150 * 1024) {
        const size_t lds16 = (size_t)4 * 16 * F * 16 * sizeof(float) + (size_t)F * (F - 1) * sizeof(short) + 16;
        hipFuncSetAttribute((const void*)k_bil16_bwd_x, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipLaunchKernelGGL(k_bil16_bwd_x, dim3(ceil_div(B, 16)), dim3(256), lds16, st, x, W, grad_out, wtype, B, F,
                           grad_x);
        const int ntiles = ceil_div(B, 16);
        int splits16 = ntiles >= 64 ? 8 : 1;
        const int tps = ceil_div(ntiles, splits16);
        splits16 = ceil_div(ntiles, tps);
        hipLaunchKernelGGL(k_bil16_bwd_w, dim3(F * (F - 1) / 2, splits16), dim3(256), 0, st, x, grad_out, wtype, B, F, tps,
                           grad_W);
        return launch_status("dt_bilinear_bwd");
    }
    const size_t lds = ((size_t)D * D + 3 * 64 * (D + 1)) * sizeof(float);
    hipLaunchKernelGGL(k_bilinear_bwd_x, dim3(ceil_div(B, 64), F), dim3(64), lds, st, x, W, grad_out, wtype, B, F, D,
                       grad_x);
    int splits = ceil_div(B, 1024);
    if (splits > 16) splits = 16;
    const int rps = ceil_div(ceil_div(B, splits), 64) * 64;
    splits = ceil_div(B, rps);
    hipLaunchKernelGGL(k_bilinear_bwd_w, dim3(F * (F - 1) / 2, splits), dim3(256), (size_t)2 * 64 * D * sizeof(float), st,
                       x, grad_out, wtype, B, F, D, rps, grad_W);
    return launch_status("dt_bilinear_bwd");
}

extern "C" int dt_field_pool_fwd(const float* x, int B, int F, int D, int use_max, float* z, int* argmax,
                                 void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0, "dt_field_pool_fwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && z && (!use_max || argmax), "dt_field_pool_fwd: null pointer");
    const int64_t n = (int64_t)B * F;
    hipLaunchKernelGGL(k_field_pool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, n, D,
                       use_max, z, argmax);
    return launch_status("dt_field_pool_fwd");
}

extern "C" int dt_field_pool_bwd(const float* grad_z, const int* argmax, int B, int F, int D, int use_max,
                                 float* grad_x, void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0, "dt_field_pool_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(grad_z && grad_x && (!use_max || argmax), "dt_field_pool_bwd: null pointer");
    const int64_t total = (int64_t)B * F * D;
    hipLaunchKernelGGL(k_field_pool_bwd, dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), grad_z, argmax, total,
                       D, use_max, grad_x);
    return launch_status("dt_field_pool_bwd");
}

extern "C" int dt_field_scale_fwd(const float* x, const float* a, int B, int F, int D, float* out, void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0, "dt_field_scale_fwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && a && out, "dt_field_scale_fwd: null pointer");
    const int64_t total = (int64_t)B * F * D;
    hipLaunchKernelGGL(k_field_scale, dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), x, a, total, D, out);
    return launch_status("dt_field_scale_fwd");
}

extern "C" int dt_field_scale_bwd(const float* x, const float* a, const float* grad_out, int B, int F, int D,
                                  float* grad_x, float* grad_a, void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0, "dt_field_scale_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && a && grad_out && grad_x && grad_a, "dt_field_scale_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int64_t total = (int64_t)B * F * D, n = (int64_t)B * F;
    hipLaunchKernelGGL(k_field_scale, dim3(ew_blocks(total)), dim3(256), 0, st, grad_out, a, total, D, grad_x);
    hipLaunchKernelGGL(k_field_scale_bwd_a, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, grad_out, n, D,
                       grad_a);
    return launch_status("dt_field_scale_bwd");
}
