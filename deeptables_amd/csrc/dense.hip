// dense.hip — Keras `Dense` (y = act(x W + b)) forward/backward for the shapes this model family uses.
//
// The reference's Dense layers (deepnets.dnn deepnets.py:401-427; the per-net Dense(1) logits and the
// `task_output` of deepmodel.py:291-292,455; the Q/K/V/residual projections of MultiheadAttention,
// layers.py:104-108) are Keras built-ins.  A vendor GEMM is the obvious stand-in, but at these shapes
// (N = 8192..212,992 rows, K <= ~1000, M in {1, 32, 64, 128}) rocBLAS picks pathological tiles — measured
// 55-72 us for every [8192 x K] x [K x 1] product and 549 us for [212,992 x 32] x [32 x 32]
// (profiles/r01_*_kernel_stats.csv) — so the layer is hand-written here:
//   M >= 2 : fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32): block = 4 waves, a [rows x K] tile of x staged
//            once in LDS, each wave owns 32x32 output tiles; bias + relu fused.  The same kernel computes
//            grad_x = G W^T with the masked gradient G = grad_y * act'(y) formed while staging.
//   M == 1 : one wavefront per row (coalesced row read, shuffle reduction): HBM-bound.
//   grad_W = x^T G: 32x32 tiles, batch split over blocks, operands streamed from global (coalesced along the
//            tile's 32 columns), partial tiles merged with a few float atomics per address; grad_b rides along.
#include "common.h"

namespace dt {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int kDCH = 16;  // MFMA steps per operand chunk

__device__ __forceinline__ float dact(float g, float y, int act) {
    return (act == DT_ACT_RELU && !(y > 0.f)) ? 0.f : g;
}

// ---------------------------------------------------------------------------------------------
// out[N, Mo] = act( A[N, Kd] . Bm[Kd, Mo] + bias )   with A optionally masked: A = ga * act'(ya)
//   block: 4 waves = NBW waves along the output columns x RW row tiles of 32 rows.
// ---------------------------------------------------------------------------------------------
template <bool MASKED>
__global__ __launch_bounds__(256) void k_dense_mfma(const float* __restrict__ A, const float* __restrict__ ya,
                                                    int mask_act, const float* __restrict__ Bm,
                                                    const float* __restrict__ bias, int act, int N, int Kd, int Mo,
                                                    int NBW, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int RW = 4 / NBW;
    const int rows_blk = 32 * RW;
    const int KS = (Kd + 1) | 1;  // odd LDS row stride >= Kd + 1 (one zero pad column for odd Kd)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane >> 5, c = lane & 31;
    const int wn = wave % NBW, wr = wave / NBW;
    const int64_t m0 = (int64_t)blockIdx.x * rows_blk;

    // ---- stage the A tile (coalesced along K), zero padded ----
    if ((Kd & 3) == 0) {   // 16-byte loads: 4 consecutive k per thread
        const int q4 = Kd >> 2;
        for (int e = threadIdx.x; e < rows_blk * q4; e += blockDim.x) {
            const int r = e / q4, q = e - r * q4;
            const int64_t m = m0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < N) {
                v = *reinterpret_cast<const float4*>(A + m * Kd + 4 * q);
                if (MASKED) {
                    const float4 yv = *reinterpret_cast<const float4*>(ya + m * Kd + 4 * q);
                    v.x = dact(v.x, yv.x, mask_act); v.y = dact(v.y, yv.y, mask_act);
                    v.z = dact(v.z, yv.z, mask_act); v.w = dact(v.w, yv.w, mask_act);
                }
            }
            float* dst = lds + r * KS + 4 * q;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        for (int r = threadIdx.x; r < rows_blk; r += blockDim.x)
            for (int k = Kd; k < KS; ++k) lds[r * KS + k] = 0.f;
    } else {
        for (int e = threadIdx.x; e < rows_blk * KS; e += blockDim.x) {
            const int r = e / KS, k = e - r * KS;
            const int64_t m = m0 + r;
            float v = 0.f;
            if (m < N && k < Kd) {
                v = A[m * Kd + k];
                if (MASKED) v = dact(v, ya[m * Kd + k], mask_act);
            }
            lds[e] = v;
        }
    }
    __syncthreads();

    const float* arow = lds + (wr * 32 + c) * KS + s;
    const int nblocks = (Mo + 31) >> 5;
    const int steps = (Kd + 1) >> 1;         // k = 2 st + s; odd Kd: the last s = 1 operand is the zero pad / guarded
    const int full = (Kd >> 1);              // steps whose both k are < Kd
    for (int nb = wn; nb < nblocks; nb += NBW) {
        const int col = 32 * nb + c;
        const bool cok = col < Mo;
        const float* bcol = Bm + (int64_t)s * Mo + (cok ? col : 0);
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        int st = 0;
        const int nch = full / kDCH;
        if (nch > 0) {
            float a0[kDCH], b0[kDCH], a1[kDCH], b1[kDCH];
#pragma unroll
            for (int i = 0; i < kDCH; ++i) { a0[i] = arow[2 * i]; b0[i] = bcol[(int64_t)2 * i * Mo]; }
            for (int ch = 0; ch < nch; ch += 2) {
                const bool has1 = ch + 1 < nch, has2 = ch + 2 < nch;
                if (has1) {
                    const float* an = arow + 2 * kDCH * (ch + 1);
                    const float* bn = bcol + (int64_t)2 * kDCH * (ch + 1) * Mo;
#pragma unroll
                    for (int i = 0; i < kDCH; ++i) { a1[i] = an[2 * i]; b1[i] = bn[(int64_t)2 * i * Mo]; }
                }
#pragma unroll
                for (int i = 0; i < kDCH; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc, 0, 0, 0);
                if (has2) {
                    const float* a2 = arow + 2 * kDCH * (ch + 2);
                    const float* b2 = bcol + (int64_t)2 * kDCH * (ch + 2) * Mo;
#pragma unroll
                    for (int i = 0; i < kDCH; ++i) { a0[i] = a2[2 * i]; b0[i] = b2[(int64_t)2 * i * Mo]; }
                }
                if (has1) {
#pragma unroll
                    for (int i = 0; i < kDCH; ++i)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc, 0, 0, 0);
                }
            }
            st = nch * kDCH;
        }
        for (; st < steps; ++st) {   // remainder (< 16 steps) and the odd-K tail, guarded
            const int k = 2 * st + s;
            const float b = k < Kd ? bcol[(int64_t)2 * st * Mo] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * st], b, acc, 0, 0, 0);
        }
        if (cok) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (m < N) {
                    float v = acc[r] + bv;
                    if (act == DT_ACT_RELU) v = fmaxf(v, 0.f);
                    out[m * Mo + col] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// M == 1: y[n] = act(x[n,:] . w + b)      (one wave per row)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int act, int N, int K,
                                                    float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int64_t n = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); n < N; n += (int64_t)gridDim.x * wpb) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) acc += x[n * K + k] * w[k];
        acc = wave_sum(acc);
        if (lane == 0) {
            float v = acc + (bias ? bias[0] : 0.f);
            if (act == DT_ACT_RELU) v = fmaxf(v, 0.f);
            y[n] = v;
        }
    }
}

// grad_x[n,k] = G[n] w[k];  grad_w[k] += sum_n G[n] x[n,k];  grad_b += sum_n G[n]
// A block owns `rows_per_block` (<= 64) rows: their G sit in LDS, a thread owns 4 adjacent columns (16-byte lanes)
// and walks the rows 4 at a time; the block's column sums go to part[block][K] (no atomics: every block of the
// grid would otherwise hit the same K addresses) and k_dense1_reduce adds the blocks up.
constexpr int kD1Rows = 64;
typedef float d1_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_dense1_bwd(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ y, const float* __restrict__ gy,
                                                    int act, int N, int K, int rows_per_block,
                                                    float* __restrict__ gx, float* __restrict__ part, int vec4) {
    __shared__ float gs[kD1Rows];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int nr = (int)(min((int64_t)N, r0 + rows_per_block) - r0);
    if ((int)threadIdx.x < kD1Rows)
        gs[threadIdx.x] = (int)threadIdx.x < nr ? dact(gy[r0 + threadIdx.x], y[r0 + threadIdx.x], act) : 0.f;
    __syncthreads();
    float* prow = part + (int64_t)blockIdx.x * (K + 4);
    if (vec4) {
        const int K4 = K >> 2;
        for (int k4 = threadIdx.x; k4 < K4; k4 += blockDim.x) {
            const d1_f4 wk = reinterpret_cast<const d1_f4*>(w)[k4];
            const d1_f4* xp = reinterpret_cast<const d1_f4*>(x) + r0 * K4 + k4;
            d1_f4* gp = gx ? reinterpret_cast<d1_f4*>(gx) + r0 * K4 + k4 : nullptr;
            d1_f4 acc = {0.f, 0.f, 0.f, 0.f};
            int n = 0;
            for (; n + 3 < nr; n += 4) {
                const d1_f4 x0 = xp[(int64_t)n * K4], x1 = xp[(int64_t)(n + 1) * K4], x2 = xp[(int64_t)(n + 2) * K4],
                            x3 = xp[(int64_t)(n + 3) * K4];
                const float g0 = gs[n], g1 = gs[n + 1], g2 = gs[n + 2], g3 = gs[n + 3];
                if (gp) {
                    gp[(int64_t)n * K4] = wk * g0; gp[(int64_t)(n + 1) * K4] = wk * g1;
                    gp[(int64_t)(n + 2) * K4] = wk * g2; gp[(int64_t)(n + 3) * K4] = wk * g3;
                }
                acc += (x0 * g0 + x1 * g1) + (x2 * g2 + x3 * g3);
            }
            for (; n < nr; ++n) {
                const float g = gs[n];
                if (gp) gp[(int64_t)n * K4] = wk * g;
                acc += xp[(int64_t)n * K4] * g;
            }
            reinterpret_cast<d1_f4*>(prow)[k4] = acc;
        }
    } else {
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            const float wk = w[k];
            float acc = 0.f;
            for (int n = 0; n < nr; ++n) {
                const float g = gs[n];
                if (gx) gx[(r0 + n) * K + k] = g * wk;
                acc += g * x[(r0 + n) * K + k];
            }
            prow[k] = acc;
        }
    }
    if (threadIdx.x < 64) {
        float bsum = gs[threadIdx.x];
        bsum = wave_sum(bsum);
        if (threadIdx.x == 0) prow[K] = bsum;
    }
}

// grad_w[k] += sum_blocks part[block][k]  (k == K: the bias).  64 columns x 4 block-groups per workgroup, 16 loads in
// flight per thread
__global__ __launch_bounds__(256) void k_dense1_reduce(const float* __restrict__ part, int blocks, int K,
                                                       float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const int64_t stride = K + 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (k <= K) {
        const float* p = part + k;
        int i = grp;
        for (; i + 60 < blocks; i += 64) {
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u & 3] += p[(int64_t)(i + 4 * u) * stride];
        }
        for (; i < blocks; i += 4) a[0] += p[(int64_t)i * stride];
    }
    red[grp][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (grp != 0 || k > K) return;
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (k < K) gw[k] += v;
    else if (gb) gb[0] += v;
}

__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ W, int K, int M,
                                                   float* __restrict__ WT) {
    const int64_t total = (int64_t)K * M;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / K), k = (int)(e - (int64_t)m * K);   // WT[m][k]
        WT[e] = W[(int64_t)k * M + m];
    }
}

// ---------------------------------------------------------------------------------------------
// grad_W[k, n] += sum_m x[m,k] G[m,n]; grad_b[n] += sum_m G[m,n]   (G = gy * act'(y))
//   grid (tiles = ceil(K/32) x ceil(M/32), row splits); 4 waves take quarters of the split's rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_wgrad(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ gy, int act, int N, int K, int M,
                                                     int row_splits, float* __restrict__ gW,
                                                     float* __restrict__ gb) {
    __shared__ float red[4][32][33];
    __shared__ float bred[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane >> 5, c = lane & 31;
    const int nbm = (M + 31) >> 5;
    const int kb = blockIdx.x / nbm, nb = blockIdx.x % nbm;
    const int kcol = 32 * kb + c, ncol = 32 * nb + c;
    const bool kok = kcol < K, nok = ncol < M;
    const int64_t rows_per_split = (((int64_t)N + row_splits - 1) / row_splits + 7) & ~(int64_t)7;
    const int64_t rq = rows_per_split >> 2;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_split + wave * rq;
    const int64_t r_end = min((int64_t)N, r_begin + rq);
    const float* xa = x + (kok ? kcol : 0);
    const float* ga = gy + (nok ? ncol : 0);
    const float* ya = y + (nok ? ncol : 0);

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    constexpr int CH = kDCH;
    for (int64_t base = r_begin; base < r_end; base += 2 * CH) {
        float aq[CH], gq[CH], yq[CH];
        const bool fullc = base + 2 * CH <= r_end;   // wave-uniform
        if (fullc) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int64_t m = base + 2 * i + s;
                aq[i] = xa[m * K];
                gq[i] = ga[m * M];
                yq[i] = act == DT_ACT_RELU ? ya[m * M] : 1.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int64_t m = base + 2 * i + s;
                const bool ok = m < r_end;
                aq[i] = ok ? xa[m * K] : 0.f;
                gq[i] = ok ? ga[m * M] : 0.f;
                yq[i] = (ok && act == DT_ACT_RELU) ? ya[m * M] : 1.f;
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float g = (nok && yq[i] > 0.f) ? gq[i] : 0.f;
            bsum += g;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kok ? aq[i] : 0.f, g, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * s][c] = acc[r];
    bsum += __shfl_xor(bsum, 32, 64);
    if (s == 0) bred[wave][c] = bsum;
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += blockDim.x) {
        const int i = e >> 5, j = e & 31;
        const int kk = 32 * kb + i, nn = 32 * nb + j;
        if (kk < K && nn < M)
            atomicAdd(gW + (int64_t)kk * M + nn, (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]));
    }
    if (gb && kb == 0 && threadIdx.x < 32 && 32 * nb + threadIdx.x < M)
        atomicAdd(gb + 32 * nb + threadIdx.x,
                  (bred[0][threadIdx.x] + bred[1][threadIdx.x]) + (bred[2][threadIdx.x] + bred[3][threadIdx.x]));
}

static int dense_nbw(int M) {
    const int nb = (M + 31) / 32;
    return nb >= 4 ? 4 : (nb >= 2 ? 2 : 1);
}
static size_t dense_lds(int K, int nbw) { return (size_t)(32 * (4 / nbw)) * ((K + 1) | 1) * sizeof(float); }

}  // namespace dt

using namespace dt;

extern "C" int dt_dense_supported(int N, int K, int M) {
    if (N <= 0 || K <= 0 || M <= 0) return 0;
    if (M == 1) return 1;
    return dense_lds(K, dense_nbw(M)) <= 150 * 1024 && dense_lds(M, dense_nbw(K)) <= 150 * 1024;
}

static int dense1_rows(int N) {          // rows per block of the Dense(1) backward: ~1024 blocks, 32..64 rows each
    int rpb = ceil_div(N, 1024);
    if (rpb < 32) rpb = 32;
    if (rpb > kD1Rows) rpb = kD1Rows;
    return rpb;
}

extern "C" int64_t dt_dense_workspace_bytes(int N, int K, int M) {
    if (M == 1) return (int64_t)sizeof(float) * ceil_div(N > 0 ? N : 1, dense1_rows(N)) * (K + 4);   // per-block column sums
    return (int64_t)sizeof(float) * K * M;   // W^T for grad_x
}

extern "C" int dt_dense_fwd(const float* x, const float* W, const float* bias, int act, int N, int K, int M,
                            float* y, void* stream) {
    DT_REQUIRE(N >= 0 && K > 0 && M > 0, "dt_dense_fwd: bad sizes N=%d K=%d M=%d", N, K, M);
    DT_REQUIRE(act == DT_ACT_LINEAR || act == DT_ACT_RELU, "dt_dense_fwd: act %d", act);
    if (N == 0) return DT_OK;
    DT_REQUIRE(x && W && y, "dt_dense_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (M == 1) {
        int blocks = ceil_div(N, 4);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_dense1_fwd, dim3(blocks), dim3(256), 0, st, x, W, bias, act, N, K, y);
        return launch_status("dt_dense_fwd(gemv)");
    }
    const int nbw = dense_nbw(M);
    const size_t lds = dense_lds(K, nbw);
    DT_UNSUPPORTED(lds > 150 * 1024, "dt_dense_fwd: K=%d too large for the LDS row tile", K);
    hipFuncSetAttribute((const void*)k_dense_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_dense_mfma<false>), dim3(ceil_div(N, 32 * (4 / nbw))), dim3(256), lds, st, x, nullptr, 0, W,
                       bias, act, N, K, M, nbw, y);
    return launch_status("dt_dense_fwd");
}

extern "C" int dt_dense_bwd(const float* x, const float* W, const float* y, const float* grad_y, int act, int N,
                            int K, int M, float* grad_x, float* grad_W, float* grad_b, void* ws, void* stream) {
    DT_REQUIRE(N >= 0 && K > 0 && M > 0, "dt_dense_bwd: bad sizes");
    if (N == 0) return DT_OK;
    DT_REQUIRE(x && W && y && grad_y && grad_W, "dt_dense_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (M == 1) {
        DT_REQUIRE(ws, "dt_dense_bwd: null workspace");
        const int rpb = dense1_rows(N), blocks = ceil_div(N, rpb);
        const int vec4 = (K % 4 == 0) && (((uintptr_t)x | (uintptr_t)W | (uintptr_t)grad_x | (uintptr_t)ws) % 16 == 0);
        float* part = reinterpret_cast<float*>(ws);
        hipLaunchKernelGGL(k_dense1_bwd, dim3(blocks), dim3(256), 0, st, x, W, y, grad_y, act, N, K, rpb, grad_x, part,
                           vec4);
        hipLaunchKernelGGL(k_dense1_reduce, dim3(ceil_div(K + 1, 64)), dim3(256), 0, st, part, blocks, K, grad_W,
                           grad_b);
        return launch_status("dt_dense_bwd(gemv)");
    }
    if (grad_x) {
        DT_REQUIRE(ws, "dt_dense_bwd: null workspace");
        float* WT = reinterpret_cast<float*>(ws);
        hipLaunchKernelGGL(k_transpose, dim3(ceil_div((int64_t)K * M, 256)), dim3(256), 0, st, W, K, M, WT);
        const int nbw = dense_nbw(K);
        const size_t lds = dense_lds(M, nbw);
        DT_UNSUPPORTED(lds > 150 * 1024, "dt_dense_bwd: M=%d too large for the LDS row tile", M);
        hipFuncSetAttribute((const void*)k_dense_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_dense_mfma<true>), dim3(ceil_div(N, 32 * (4 / nbw))), dim3(256), lds, st, grad_y, y, act,
                           WT, nullptr, DT_ACT_LINEAR, N, M, K, nbw, grad_x);
    }
    const int tiles = ceil_div(K, 32) * ceil_div(M, 32);
    // enough (tile, batch-split) blocks for ~4 waves per SIMD; each wave keeps >= 64 rows of work
    int splits = 1024 / tiles;
    if (splits < 1) splits = 1;
    if (splits > 512) splits = 512;
    while (splits > 1 && N / splits < 256) splits >>= 1;
    hipLaunchKernelGGL(k_dense_wgrad, dim3(tiles, splits), dim3(256), 0, st, x, y, grad_y, act, N, K, M, splits,
                       grad_W, grad_b);
    return launch_status("dt_dense_bwd");
}
