// bn.hip — Keras-semantics BatchNormalization over the rows of x[N,C].
//
// Replaces keras BatchNormalization(name='bn_concat_emb_dense') in
// deeptables/models/deepmodel.py:359 (and the BatchNormalization at the end of
// MultiheadAttention.call, deeptables/models/layers.py:152, with N = batch*fields):
// biased batch variance, epsilon 1e-3 and momentum 0.99 by default (passed in), moving
// statistics updated with the biased variance.
//
// HBM-bound column reduction: a block owns a chunk of rows and all C columns; threads are laid
// out (row-lane, column) with the column index fastest so every wave reads contiguous row
// segments.  Per column the block accumulates SHIFTED sums (shift = first row of the chunk) so
// the M2 it reports does not suffer E[x^2]-E[x]^2 cancellation; chunk results are merged with
// Chan's parallel formula in a one-thread-per-column finalize kernel.
#include <initializer_list>
#include "common.h"

namespace dt {

constexpr int kBnThreads = 256;
constexpr int kBnMaxChunks = 512;

static int bn_chunks(int N) {
    int c = ceil_div(N, 32);
    if (c > kBnMaxChunks) c = kBnMaxChunks;
    if (c < 1) c = 1;
    return c;
}
static int bn_col_width(int C) {  // power of two in [1,256]
    int w = 1;
    while (w < C && w < kBnThreads) w <<= 1;
    return w;
}

// partial[chunk][0..2][C] = {count, mean, M2}
__global__ __launch_bounds__(kBnThreads) void k_bn_stats(const float* __restrict__ x, int N, int C,
                                                         int CW, int rows_per_chunk,
                                                         float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][RS][CW]
    const int RS = kBnThreads / CW;
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(N, r0 + rows_per_chunk);
    float* ps = lds;
    float* pq = lds + RS * CW;
    for (int c0 = 0; c0 < C; c0 += CW) {
        const int col = c0 + tx;
        float s = 0.f, q = 0.f, K = 0.f;
        if (col < C && r0 < r1) {
            K = x[(int64_t)r0 * C + col];
            for (int r = r0 + ty; r < r1; r += RS) {
                const float d = x[(int64_t)r * C + col] - K;
                s += d;
                q += d * d;
            }
        }
        ps[ty * CW + tx] = s;
        pq[ty * CW + tx] = q;
        __syncthreads();
        if (ty == 0 && col < C) {
            for (int k = 1; k < RS; ++k) {
                s += ps[k * CW + tx];
                q += pq[k * CW + tx];
            }
            const float n = (float)max(r1 - r0, 0);
            float mean = 0.f, m2 = 0.f;
            if (n > 0.f) {
                mean = K + s / n;
                m2 = fmaxf(q - s * s / n, 0.f);
            }
            float* p = partial + (int64_t)blockIdx.x * 3 * C;
            p[col] = n;
            p[C + col] = mean;
            p[2 * C + col] = m2;
        }
        __syncthreads();
    }
}

// one wavefront per column: lanes stride over the chunk partials, then a Chan-merge butterfly
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ partial, int chunks,
                                                     int C, float eps, float momentum,
                                                     float* __restrict__ moving_mean,
                                                     float* __restrict__ moving_var,
                                                     float* __restrict__ save_mean,
                                                     float* __restrict__ save_rstd) {
    const int col = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (col >= C) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int k = lane; k < chunks; k += 64) {
        const float* p = partial + (int64_t)k * 3 * C;
        const float nb = p[col];
        if (nb <= 0.f) continue;
        const float mb = p[C + col], m2b = p[2 * C + col];
        const float nt = n + nb;
        const float delta = mb - mean;
        mean += delta * (nb / nt);
        m2 += m2b + delta * delta * (n * nb / nt);
        n = nt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), m2b = __shfl_xor(m2, o, 64);
        const float nt = n + nb;
        if (nt > 0.f) {
            const float delta = mb - mean;
            // symmetric form so both partners compute the same merged value
            const float new_mean = (n * mean + nb * mb) / nt;
            m2 = m2 + m2b + delta * delta * (n * nb / nt);
            mean = new_mean;
        }
        n = nt;
    }
    if (lane != 0) return;
    const float var = n > 0.f ? m2 / n : 0.f;
    save_mean[col] = mean;
    save_rstd[col] = 1.0f / sqrtf(var + eps);
    if (moving_mean) moving_mean[col] = moving_mean[col] * momentum + mean * (1.f - momentum);
    if (moving_var) moving_var[col] = moving_var[col] * momentum + var * (1.f - momentum);
}

// y = (x-mean)*rstd*gamma+beta   (a = rstd*gamma, b = beta-mean*a evaluated per element to keep
// the rounding sequence of the unfused formula)
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, int64_t total, int C,
                                                  const float* __restrict__ gamma,
                                                  const float* __restrict__ beta,
                                                  const float* __restrict__ mean,
                                                  const float* __restrict__ rstd,
                                                  float* __restrict__ y) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const float g = gamma ? gamma[c] : 1.f;
        const float bb = beta ? beta[c] : 0.f;
        y[t] = (x[t] - mean[c]) * rstd[c] * g + bb;
    }
}

__global__ __launch_bounds__(256) void k_bn_infer(const float* __restrict__ x, int64_t total, int C,
                                                  const float* __restrict__ gamma,
                                                  const float* __restrict__ beta,
                                                  const float* __restrict__ mmean,
                                                  const float* __restrict__ mvar, float eps,
                                                  float* __restrict__ y) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const float g = gamma ? gamma[c] : 1.f;
        const float bb = beta ? beta[c] : 0.f;
        y[t] = (x[t] - mmean[c]) * (1.0f / sqrtf(mvar[c] + eps)) * g + bb;
    }
}

// backward partial sums: partial[chunk][0..1][C] = {sum g, sum g*xhat}
__global__ __launch_bounds__(kBnThreads) void k_bn_bwd_stats(
    const float* __restrict__ x, const float* __restrict__ gy, int N, int C, int CW,
    int rows_per_chunk, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int RS = kBnThreads / CW;
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(N, r0 + rows_per_chunk);
    float* ps = lds;
    float* pq = lds + RS * CW;
    for (int c0 = 0; c0 < C; c0 += CW) {
        const int col = c0 + tx;
        float s = 0.f, q = 0.f;
        if (col < C) {
            const float m = mean[col], rs = rstd[col];
            for (int r = r0 + ty; r < r1; r += RS) {
                const float g = gy[(int64_t)r * C + col];
                s += g;
                q += g * ((x[(int64_t)r * C + col] - m) * rs);
            }
        }
        ps[ty * CW + tx] = s;
        pq[ty * CW + tx] = q;
        __syncthreads();
        if (ty == 0 && col < C) {
            for (int k = 1; k < RS; ++k) {
                s += ps[k * CW + tx];
                q += pq[k * CW + tx];
            }
            float* p = partial + (int64_t)blockIdx.x * 2 * C;
            p[col] = s;
            p[C + col] = q;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ partial,
                                                         int chunks, int C,
                                                         float* __restrict__ sum_g,
                                                         float* __restrict__ sum_gx,
                                                         float* __restrict__ grad_gamma,
                                                         float* __restrict__ grad_beta) {
    const int col = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (col >= C) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f, q = 0.f;
    for (int k = lane; k < chunks; k += 64) {
        s += partial[(int64_t)k * 2 * C + col];
        q += partial[(int64_t)k * 2 * C + C + col];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane != 0) return;
    sum_g[col] = s;
    sum_gx[col] = q;
    if (grad_gamma) grad_gamma[col] = q;
    if (grad_beta) grad_beta[col] = s;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(
    const float* __restrict__ x, const float* __restrict__ gy, int64_t total, int C, float inv_n,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ sum_g, const float* __restrict__ sum_gx, float* __restrict__ gx) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const float xhat = (x[t] - mean[c]) * rstd[c];
        const float g = gamma ? gamma[c] : 1.f;
        gx[t] = g * rstd[c] * (gy[t] - sum_g[c] * inv_n - xhat * (sum_gx[c] * inv_n));
    }
}

// ---- 16-byte variants for C % 4 == 0 (the [batch*fields, embedding] normalisations of the AutoInt layers: C = 16 / 32).
// A thread owns 4 adjacent columns; a wave-load covers whole 128-byte (C = 32) rows instead of 4-byte elements, the
// row loop is unrolled 4x so four independent 16-byte loads per thread are in flight, and the element-wise passes
// index with 32-bit arithmetic (the scalar versions spend most of their time in a 64-bit modulo).
typedef float bn_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBnThreads) void k_bn_stats_v4(const float* __restrict__ x, int N, int C, int CW4,
                                                            int rows_per_chunk, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][RS][CW4] float4
    const int C4 = C >> 2, RS = kBnThreads / CW4;
    const int tx = threadIdx.x % CW4, ty = threadIdx.x / CW4;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(N, r0 + rows_per_chunk);
    bn_f4* ps = reinterpret_cast<bn_f4*>(lds);
    bn_f4* pq = ps + RS * CW4;
    for (int c0 = 0; c0 < C4; c0 += CW4) {
        const int c4 = c0 + tx;
        bn_f4 s = {0.f, 0.f, 0.f, 0.f}, q = s, K = s;
        if (c4 < C4 && r0 < r1) {
            const bn_f4* xp = reinterpret_cast<const bn_f4*>(x) + c4;
            K = xp[(int64_t)r0 * C4];
            int r = r0 + ty;
            for (; r + 3 * RS < r1; r += 4 * RS) {
                const bn_f4 v0 = xp[(int64_t)r * C4], v1 = xp[(int64_t)(r + RS) * C4],
                            v2 = xp[(int64_t)(r + 2 * RS) * C4], v3 = xp[(int64_t)(r + 3 * RS) * C4];
                const bn_f4 d0 = v0 - K, d1 = v1 - K, d2 = v2 - K, d3 = v3 - K;
                s += (d0 + d1) + (d2 + d3);
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            for (; r < r1; r += RS) {
                const bn_f4 d = xp[(int64_t)r * C4] - K;
                s += d;
                q += d * d;
            }
        }
        ps[ty * CW4 + tx] = s;
        pq[ty * CW4 + tx] = q;
        __syncthreads();
        if (ty == 0 && c4 < C4) {
            for (int k = 1; k < RS; ++k) {
                s += ps[k * CW4 + tx];
                q += pq[k * CW4 + tx];
            }
            const float n = (float)max(r1 - r0, 0);
            float* p = partial + (int64_t)blockIdx.x * 3 * C + 4 * c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float mean = 0.f, m2 = 0.f;
                if (n > 0.f) {
                    mean = K[e] + s[e] / n;
                    m2 = fmaxf(q[e] - s[e] * s[e] / n, 0.f);
                }
                p[e] = n;
                p[C + e] = mean;
                p[2 * C + e] = m2;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bn_apply_v4(const float* __restrict__ x, int total4, int C4,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ y) {
    const bn_f4* xp = reinterpret_cast<const bn_f4*>(x);
    bn_f4* yp = reinterpret_cast<bn_f4*>(y);
    const bn_f4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += gridDim.x * blockDim.x) {
        const int c4 = t % C4;
        const bn_f4 g = gamma ? reinterpret_cast<const bn_f4*>(gamma)[c4] : one;
        const bn_f4 bb = beta ? reinterpret_cast<const bn_f4*>(beta)[c4] : zero;
        const bn_f4 m = reinterpret_cast<const bn_f4*>(mean)[c4], rs = reinterpret_cast<const bn_f4*>(rstd)[c4];
        yp[t] = (xp[t] - m) * rs * g + bb;
    }
}

__global__ __launch_bounds__(kBnThreads) void k_bn_bwd_stats_v4(
    const float* __restrict__ x, const float* __restrict__ gy, int N, int C, int CW4, int rows_per_chunk,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C4 = C >> 2, RS = kBnThreads / CW4;
    const int tx = threadIdx.x % CW4, ty = threadIdx.x / CW4;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(N, r0 + rows_per_chunk);
    bn_f4* ps = reinterpret_cast<bn_f4*>(lds);
    bn_f4* pq = ps + RS * CW4;
    for (int c0 = 0; c0 < C4; c0 += CW4) {
        const int c4 = c0 + tx;
        bn_f4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
        if (c4 < C4) {
            const bn_f4 m = reinterpret_cast<const bn_f4*>(mean)[c4], rs = reinterpret_cast<const bn_f4*>(rstd)[c4];
            const bn_f4* xp = reinterpret_cast<const bn_f4*>(x) + c4;
            const bn_f4* gp = reinterpret_cast<const bn_f4*>(gy) + c4;
            int r = r0 + ty;
            for (; r + RS < r1; r += 2 * RS) {
                const bn_f4 g0 = gp[(int64_t)r * C4], x0 = xp[(int64_t)r * C4];
                const bn_f4 g1 = gp[(int64_t)(r + RS) * C4], x1 = xp[(int64_t)(r + RS) * C4];
                s += g0 + g1;
                q += g0 * ((x0 - m) * rs) + g1 * ((x1 - m) * rs);
            }
            for (; r < r1; r += RS) {
                const bn_f4 g = gp[(int64_t)r * C4];
                s += g;
                q += g * ((xp[(int64_t)r * C4] - m) * rs);
            }
        }
        ps[ty * CW4 + tx] = s;
        pq[ty * CW4 + tx] = q;
        __syncthreads();
        if (ty == 0 && c4 < C4) {
            for (int k = 1; k < RS; ++k) {
                s += ps[k * CW4 + tx];
                q += pq[k * CW4 + tx];
            }
            float* p = partial + (int64_t)blockIdx.x * 2 * C + 4 * c4;
            *reinterpret_cast<bn_f4*>(p) = s;
            *reinterpret_cast<bn_f4*>(p + C) = q;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply_v4(
    const float* __restrict__ x, const float* __restrict__ gy, int total4, int C4, float inv_n,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ sum_g, const float* __restrict__ sum_gx, float* __restrict__ gx) {
    const bn_f4 one = {1.f, 1.f, 1.f, 1.f};
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += gridDim.x * blockDim.x) {
        const int c4 = t % C4;
        const bn_f4 m = reinterpret_cast<const bn_f4*>(mean)[c4], rs = reinterpret_cast<const bn_f4*>(rstd)[c4];
        const bn_f4 g = gamma ? reinterpret_cast<const bn_f4*>(gamma)[c4] : one;
        const bn_f4 sg = reinterpret_cast<const bn_f4*>(sum_g)[c4], sgx = reinterpret_cast<const bn_f4*>(sum_gx)[c4];
        const bn_f4 xhat = (reinterpret_cast<const bn_f4*>(x)[t] - m) * rs;
        reinterpret_cast<bn_f4*>(gx)[t] = g * rs * (reinterpret_cast<const bn_f4*>(gy)[t] - sg * inv_n - xhat * (sgx * inv_n));
    }
}

static bool bn_vec4(int N, int C, std::initializer_list<const void*> ptrs) {
    if (C % 4 || C > 1024 || (int64_t)N * C / 4 >= (1LL << 31)) return false;
    for (const void* p : ptrs)
        if (p && ((uintptr_t)p & 15)) return false;
    return true;
}
static int bn_col_width4(int C) {  // power of two >= C/4, <= 256
    int w = 1;
    while (w < C / 4 && w < kBnThreads) w <<= 1;
    return w;
}

static int elementwise_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace dt

using namespace dt;

extern "C" int64_t dt_bn_workspace_bytes(int N, int C) {
    if (N < 0 || C <= 0) return 0;
    // stats partials (3 floats) dominate the backward partials (2 floats) + 2*C finalized sums
    return (int64_t)sizeof(float) * ((int64_t)kBnMaxChunks * 3 * C + 2 * C);
}

extern "C" int dt_bn_train_fwd(const float* x, int N, int C, const float* gamma, const float* beta,
                               float eps, float momentum, float* moving_mean, float* moving_var,
                               float* y, float* save_mean, float* save_rstd, void* ws,
                               void* stream) {
    DT_REQUIRE(N > 0 && C > 0, "dt_bn_train_fwd: bad sizes N=%d C=%d", N, C);
    DT_REQUIRE(x && y && save_mean && save_rstd && ws, "dt_bn_train_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int chunks = bn_chunks(N);
    const int rpc = ceil_div(N, chunks);
    const int CW = bn_col_width(C);
    const size_t lds = 2 * kBnThreads * sizeof(float);
    float* partial = reinterpret_cast<float*>(ws);
    const bool v4 = bn_vec4(N, C, {x, y, gamma, beta, save_mean, save_rstd, ws});
    if (v4) {
        const int CW4 = bn_col_width4(C);
        hipLaunchKernelGGL(k_bn_stats_v4, dim3(chunks), dim3(kBnThreads), 8 * kBnThreads * sizeof(float), st, x, N, C, CW4,
                           rpc, partial);
    } else {
        hipLaunchKernelGGL(k_bn_stats, dim3(chunks), dim3(kBnThreads), lds, st, x, N, C, CW, rpc, partial);
    }
    hipLaunchKernelGGL(k_bn_finalize, dim3(ceil_div(C, 4)), dim3(256), 0, st, partial, chunks, C,
                       eps, momentum, moving_mean, moving_var, save_mean, save_rstd);
    const int64_t total = (int64_t)N * C;
    if (v4)
        hipLaunchKernelGGL(k_bn_apply_v4, dim3(elementwise_blocks(total / 4)), dim3(256), 0, st, x, (int)(total / 4), C / 4,
                           gamma, beta, save_mean, save_rstd, y);
    else
        hipLaunchKernelGGL(k_bn_apply, dim3(elementwise_blocks(total)), dim3(256), 0, st, x, total, C,
                           gamma, beta, save_mean, save_rstd, y);
    return launch_status("dt_bn_train_fwd");
}

extern "C" int dt_bn_infer_fwd(const float* x, int N, int C, const float* gamma, const float* beta,
                               float eps, const float* moving_mean, const float* moving_var,
                               float* y, void* stream) {
    DT_REQUIRE(N >= 0 && C > 0, "dt_bn_infer_fwd: bad sizes");
    if (N == 0) return DT_OK;
    DT_REQUIRE(x && y && moving_mean && moving_var, "dt_bn_infer_fwd: null pointer");
    const int64_t total = (int64_t)N * C;
    hipLaunchKernelGGL(k_bn_infer, dim3(elementwise_blocks(total)), dim3(256), 0, as_stream(stream),
                       x, total, C, gamma, beta, moving_mean, moving_var, eps, y);
    return launch_status("dt_bn_infer_fwd");
}

extern "C" int dt_bn_train_bwd(const float* x, const float* grad_y, int N, int C,
                               const float* gamma, const float* save_mean, const float* save_rstd,
                               float* grad_x, float* grad_gamma, float* grad_beta, void* ws,
                               void* stream) {
    DT_REQUIRE(N > 0 && C > 0, "dt_bn_train_bwd: bad sizes");
    DT_REQUIRE(x && grad_y && save_mean && save_rstd && grad_x && ws,
               "dt_bn_train_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int chunks = bn_chunks(N);
    const int rpc = ceil_div(N, chunks);
    const int CW = bn_col_width(C);
    const size_t lds = 2 * kBnThreads * sizeof(float);
    float* partial = reinterpret_cast<float*>(ws);
    float* sums = partial + (int64_t)kBnMaxChunks * 3 * C;
    const bool v4 = bn_vec4(N, C, {x, grad_y, gamma, save_mean, save_rstd, grad_x, ws}) &&
                    (((int64_t)kBnMaxChunks * 3 * C) & 3) == 0;
    if (v4)
        hipLaunchKernelGGL(k_bn_bwd_stats_v4, dim3(chunks), dim3(kBnThreads), 8 * kBnThreads * sizeof(float), st, x, grad_y,
                           N, C, bn_col_width4(C), rpc, save_mean, save_rstd, partial);
    else
        hipLaunchKernelGGL(k_bn_bwd_stats, dim3(chunks), dim3(kBnThreads), lds, st, x, grad_y, N, C, CW,
                           rpc, save_mean, save_rstd, partial);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(ceil_div(C, 4)), dim3(256), 0, st, partial, chunks,
                       C, sums, sums + C, grad_gamma, grad_beta);
    const int64_t total = (int64_t)N * C;
    if (v4)
        hipLaunchKernelGGL(k_bn_bwd_apply_v4, dim3(elementwise_blocks(total / 4)), dim3(256), 0, st, x, grad_y,
                           (int)(total / 4), C / 4, 1.0f / (float)N, gamma, save_mean, save_rstd, sums, sums + C, grad_x);
    else
        hipLaunchKernelGGL(k_bn_bwd_apply, dim3(elementwise_blocks(total)), dim3(256), 0, st, x, grad_y,
                           total, C, 1.0f / (float)N, gamma, save_mean, save_rstd, sums, sums + C,
                           grad_x);
    return launch_status("dt_bn_train_bwd");
}

/* The reduction half of dt_bn_train_bwd: sum_g[c] = sum_n grad_y, sum_gx[c] = sum_n grad_y * xhat (= grad_beta, grad_gamma),
 * for a consumer that applies  grad_x = gamma rstd (grad_y - sum_g / N - xhat sum_gx / N)  itself while it reads
 * grad_y anyway (dt_autoint_bwd).  sums: [2*C] = sum_g | sum_gx. */
extern "C" int dt_bn_train_bwd_stats(const float* x, const float* grad_y, int N, int C, const float* save_mean,
                                     const float* save_rstd, float* sums, float* grad_gamma, float* grad_beta, void* ws,
                                     void* stream) {
    DT_REQUIRE(N > 0 && C > 0, "dt_bn_train_bwd_stats: bad sizes");
    DT_REQUIRE(x && grad_y && save_mean && save_rstd && sums && ws, "dt_bn_train_bwd_stats: null pointer");
    hipStream_t st = as_stream(stream);
    const int chunks = bn_chunks(N);
    const int rpc = ceil_div(N, chunks);
    float* partial = reinterpret_cast<float*>(ws);
    if (bn_vec4(N, C, {x, grad_y, save_mean, save_rstd, ws}))
        hipLaunchKernelGGL(k_bn_bwd_stats_v4, dim3(chunks), dim3(kBnThreads), 8 * kBnThreads * sizeof(float), st, x, grad_y,
                           N, C, bn_col_width4(C), rpc, save_mean, save_rstd, partial);
    else
        hipLaunchKernelGGL(k_bn_bwd_stats, dim3(chunks), dim3(kBnThreads), 2 * kBnThreads * sizeof(float), st, x, grad_y, N,
                           C, bn_col_width(C), rpc, save_mean, save_rstd, partial);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(ceil_div(C, 4)), dim3(256), 0, st, partial, chunks, C, sums, sums + C,
                       grad_gamma, grad_beta);
    return launch_status("dt_bn_train_bwd_stats");
}
