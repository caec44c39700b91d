// cross.hip — depth-fused CrossNet.
//
// Replaces Cross.call, deeptables/models/layers.py:428-436:
//     x_n = tf.matmul(x_f, tf.tensordot(x_n, kernels[i], axes=(1,0))) + x_n + bias[i]
// i.e. per row  x_{l+1} = x_0 * (x_l . w_l) + x_l + b_l,  l = 0..L-1.
// The reference runs >= 4 full passes over [B,C] per layer; here one wavefront keeps a whole
// row (x_0 and x_l, C/64 floats per lane each) in registers across all L layers: one read and
// one write of the row, one wave reduction per layer.  Only the L scalars s_l = x_l.w_l are
// saved for the backward, which recomputes x_l from x_0 (exactly the forward's values).
//
// Backward (t_l = g_{l+1} . x_0):  g_l = g_{l+1} + w_l t_l ;  grad_w_l = sum_b x_l t_l ;
// grad_b_l = sum_b g_{l+1} ;  grad_x = g_0 + sum_l g_{l+1} s_l.
// grad_w/grad_b are accumulated per block in LDS, written as per-block partials into the
// workspace and summed by a second kernel (no global float atomics).
#include "common.h"

namespace dt {

constexpr int kCrossMaxBlocks = 512;

template <int PER>
__global__ __launch_bounds__(256) void k_cross_fwd(const float* __restrict__ x,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ bias, int B, int C,
                                                   int L, float* __restrict__ out,
                                                   float* __restrict__ save_s) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int b = blockIdx.x * wpb + (threadIdx.x >> 6); b < B; b += gridDim.x * wpb) {
        float x0[PER], xl[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int col = k * 64 + lane;
            x0[k] = col < C ? x[(int64_t)b * C + col] : 0.f;
            xl[k] = x0[k];
        }
        for (int l = 0; l < L; ++l) {
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int col = k * 64 + lane;
                if (col < C) p += xl[k] * w[(int64_t)l * C + col];
            }
            const float s = wave_sum(p);
            if (save_s && lane == 0) save_s[(int64_t)b * L + l] = s;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int col = k * 64 + lane;
                if (col < C) xl[k] = x0[k] * s + xl[k] + bias[(int64_t)l * C + col];
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int col = k * 64 + lane;
            if (col < C) out[(int64_t)b * C + col] = xl[k];
        }
    }
}

template <int PER>
__global__ __launch_bounds__(256) void k_cross_bwd(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ save_s, const float* __restrict__ gout, int B, int C, int L,
    float* __restrict__ gx, float* __restrict__ partial /* [grid][2][L][C] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][L][C]
    float* lgw = lds;
    float* lgb = lds + (int64_t)L * C;
    for (int i = threadIdx.x; i < 2 * L * C; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int b = blockIdx.x * wpb + (threadIdx.x >> 6); b < B; b += gridDim.x * wpb) {
        float x0[PER], g[PER], acc[PER], xl[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int col = k * 64 + lane;
            x0[k] = col < C ? x[(int64_t)b * C + col] : 0.f;
            g[k] = col < C ? gout[(int64_t)b * C + col] : 0.f;
            acc[k] = 0.f;
        }
        for (int l = L - 1; l >= 0; --l) {
            // recompute x_l from x_0 with the saved scalars (same arithmetic as the forward)
#pragma unroll
            for (int k = 0; k < PER; ++k) xl[k] = x0[k];
            for (int m = 0; m < l; ++m) {
                const float sm = save_s[(int64_t)b * L + m];
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int col = k * 64 + lane;
                    if (col < C) xl[k] = x0[k] * sm + xl[k] + bias[(int64_t)m * C + col];
                }
            }
            const float s = save_s[(int64_t)b * L + l];
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) p += g[k] * x0[k];
            const float t = wave_sum(p);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int col = k * 64 + lane;
                if (col < C) {
                    atomicAdd(&lgw[(int64_t)l * C + col], xl[k] * t);
                    atomicAdd(&lgb[(int64_t)l * C + col], g[k]);
                    acc[k] += g[k] * s;
                    g[k] += w[(int64_t)l * C + col] * t;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int col = k * 64 + lane;
            if (col < C) gx[(int64_t)b * C + col] = g[k] + acc[k];
        }
    }
    __syncthreads();
    float* p = partial + (int64_t)blockIdx.x * 2 * L * C;
    for (int i = threadIdx.x; i < 2 * L * C; i += blockDim.x) p[i] = lds[i];
}

// Register-accumulating variant for L <= LMAX: each wave keeps its grad_w / grad_b contributions in registers
// across all of its rows and touches LDS once at the end (the generic kernel above does 2 LDS atomics per
// (row, layer, column)).  x_0 .. x_{L-1} of a row are rebuilt ONCE, in forward order, into registers (LMAX * PER of
// them) — round 1 rebuilt x_l from x_0 for every l (O(L^2) passes with an LDS bias read per element: 95 us at
// B = 8192, C = 429, L = 6).
template <int PER, int LMAX>
__global__ __launch_bounds__(256) void k_cross_bwd_reg(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ save_s, const float* __restrict__ gout, int B, int C, int L,
    float* __restrict__ gx, float* __restrict__ partial /* [grid][2][L][C] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][L][C] accumulators, then [2][L][C] w|bias
    float* wl = lds + (int64_t)2 * L * C;
    float* bl = wl + (int64_t)L * C;
    for (int i = threadIdx.x; i < 2 * L * C; i += blockDim.x) lds[i] = 0.f;
    for (int i = threadIdx.x; i < L * C; i += blockDim.x) { wl[i] = w[i]; bl[i] = bias[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    float gwa[LMAX][PER], gba[LMAX][PER];
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
#pragma unroll
        for (int k = 0; k < PER; ++k) { gwa[l][k] = 0.f; gba[l][k] = 0.f; }
    for (int b = blockIdx.x * wpb + (threadIdx.x >> 6); b < B; b += gridDim.x * wpb) {
        float xs[LMAX][PER], g[PER], acc[PER], sv[LMAX];          // xs[l] = x_l (xs[0] = x_0)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int col = k * 64 + lane;
            xs[0][k] = col < C ? x[(int64_t)b * C + col] : 0.f;
            g[k] = col < C ? gout[(int64_t)b * C + col] : 0.f;
            acc[k] = 0.f;
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) sv[l] = l < L ? save_s[(int64_t)b * L + l] : 0.f;
#pragma unroll
        for (int l = 1; l < LMAX; ++l) {
            if (l >= L) break;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int col = k * 64 + lane;
                xs[l][k] = col < C ? xs[0][k] * sv[l - 1] + xs[l - 1][k] + bl[(l - 1) * C + col] : 0.f;   // the forward's arithmetic
            }
        }
#pragma unroll
        for (int l = LMAX - 1; l >= 0; --l) {
            if (l >= L) continue;
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) p += g[k] * xs[0][k];
            const float t = wave_sum(p);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int col = k * 64 + lane;
                if (col < C) {
                    gwa[l][k] += xs[l][k] * t;
                    gba[l][k] += g[k];
                    acc[k] += g[k] * sv[l];
                    g[k] += wl[l * C + col] * t;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int col = k * 64 + lane;
            if (col < C) gx[(int64_t)b * C + col] = g[k] + acc[k];
        }
    }
    // the block's waves add their registers into the LDS accumulator ONE WAVE AT A TIME with plain read-modify-writes
    // (LDS float atomics run at about one lane-atomic per three cycles on gfx950: 84 of them per lane cost more than
    // the rows themselves)
    for (int turn = 0; turn < wpb; ++turn) {
        if ((int)(threadIdx.x >> 6) == turn) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l >= L) continue;
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int col = k * 64 + lane;
                    if (col < C) {
                        lds[(int64_t)l * C + col] += gwa[l][k];
                        lds[(int64_t)(L + l) * C + col] += gba[l][k];
                    }
                }
            }
        }
        __syncthreads();
    }
    float* pp = partial + (int64_t)blockIdx.x * 2 * L * C;
    for (int i = threadIdx.x; i < 2 * L * C; i += blockDim.x) pp[i] = lds[i];
}

// one wavefront per (layer, column) element: lanes stride over the block partials
__global__ __launch_bounds__(256) void k_cross_bwd_reduce(const float* __restrict__ partial,
                                                          int nblocks, int LC,
                                                          float* __restrict__ grad_w,
                                                          float* __restrict__ grad_b) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= 2 * LC) return;
    const int lane = threadIdx.x & 63;
    float s0 = 0.f, s1 = 0.f;
    int k = lane;
    for (; k + 64 < nblocks; k += 128) {
        s0 += partial[(int64_t)k * 2 * LC + i];
        s1 += partial[(int64_t)(k + 64) * 2 * LC + i];
    }
    if (k < nblocks) s0 += partial[(int64_t)k * 2 * LC + i];
    const float s = wave_sum(s0 + s1);
    if (lane != 0) return;
    if (i < LC) {
        if (grad_w) grad_w[i] += s;
    } else {
        if (grad_b) grad_b[i - LC] += s;
    }
}

static int cross_blocks(int B) {
    int g = ceil_div(B, 4);
    if (g > kCrossMaxBlocks) g = kCrossMaxBlocks;
    return g < 1 ? 1 : g;
}
static int cross_per(int C) {          // 64-column register slices per row: exact up to 8, powers of two beyond
    const int need = ceil_div(C, 64);
    if (need <= 8) return need;
    int per = 16;
    while (per < need) per <<= 1;
    return per;
}

}  // namespace dt

using namespace dt;

extern "C" int64_t dt_cross_workspace_bytes(int B, int C, int L) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    return (int64_t)sizeof(float) * cross_blocks(B) * 2 * L * C;
}

extern "C" int dt_cross_fwd(const float* x, const float* w, const float* b, int B, int C, int L,
                            float* out, float* save_s, void* stream) {
    DT_REQUIRE(B >= 0 && C > 0 && L >= 0, "dt_cross_fwd: bad sizes B=%d C=%d L=%d", B, C, L);
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && out && (L == 0 || (w && b)), "dt_cross_fwd: null pointer");
    const int per = cross_per(C);
    DT_UNSUPPORTED(per > 32, "dt_cross_fwd: C=%d exceeds 2048 columns per row", C);
    hipStream_t st = as_stream(stream);
    int fwd_blocks = ceil_div(B, 4);
    if (fwd_blocks > 2048) fwd_blocks = 2048;
    dim3 grid(fwd_blocks), block(256);
#define DT_CROSS_FWD(P)                                                                         \
    case P:                                                                                     \
        hipLaunchKernelGGL((k_cross_fwd<P>), grid, block, 0, st, x, w, b, B, C, L, out, save_s); \
        break;
    switch (per) {
        DT_CROSS_FWD(1) DT_CROSS_FWD(2) DT_CROSS_FWD(3) DT_CROSS_FWD(4) DT_CROSS_FWD(5) DT_CROSS_FWD(6) DT_CROSS_FWD(7)
        DT_CROSS_FWD(8) DT_CROSS_FWD(16) DT_CROSS_FWD(32)
    }
#undef DT_CROSS_FWD
    return launch_status("dt_cross_fwd");
}

extern "C" int dt_cross_bwd(const float* x, const float* w, const float* b, const float* save_s,
                            const float* grad_out, int B, int C, int L, float* grad_x,
                            float* grad_w, float* grad_b, void* ws, void* stream) {
    DT_REQUIRE(B >= 0 && C > 0 && L >= 0, "dt_cross_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && grad_out && grad_x && (L == 0 || (w && b && save_s && ws)),
               "dt_cross_bwd: null pointer");
    const int per = cross_per(C);
    DT_UNSUPPORTED(per > 32, "dt_cross_bwd: C=%d exceeds 2048 columns per row", C);
    const size_t lds = (size_t)2 * L * C * sizeof(float);
    DT_UNSUPPORTED(lds > 64 * 1024, "dt_cross_bwd: L*C=%d exceeds the 8192-float LDS accumulator",
                   L * C);
    hipStream_t st = as_stream(stream);
    const int nblocks = cross_blocks(B);
    dim3 grid(nblocks), block(256);
    float* partial = reinterpret_cast<float*>(ws);
#define DT_CROSS_BWD(P)                                                                               \
    case P:                                                                                           \
        if (L <= 4 && P <= 8 && 2 * lds <= 64 * 1024)                                                 \
            hipLaunchKernelGGL((k_cross_bwd_reg<(P <= 8 ? P : 8), 4>), grid, block, 2 * lds, st, x, w, b, \
                               save_s, grad_out, B, C, L, grad_x, partial);                           \
        else if (L <= 8 && P <= 8 && 2 * lds <= 64 * 1024)                                            \
            hipLaunchKernelGGL((k_cross_bwd_reg<(P <= 8 ? P : 8), 8>), grid, block, 2 * lds, st, x, w, b, \
                               save_s, grad_out, B, C, L, grad_x, partial);                           \
        else                                                                                          \
            hipLaunchKernelGGL((k_cross_bwd<P>), grid, block, lds, st, x, w, b, save_s, grad_out, B,  \
                               C, L, grad_x, partial);                                                \
        break;
    switch (per) {
        DT_CROSS_BWD(1) DT_CROSS_BWD(2) DT_CROSS_BWD(3) DT_CROSS_BWD(4) DT_CROSS_BWD(5) DT_CROSS_BWD(6) DT_CROSS_BWD(7)
        DT_CROSS_BWD(8) DT_CROSS_BWD(16) DT_CROSS_BWD(32)
    }
#undef DT_CROSS_BWD
    if (L > 0 && (grad_w || grad_b))
        hipLaunchKernelGGL(k_cross_bwd_reduce, dim3(ceil_div(2 * L * C, 4)), dim3(256), 0, st,
                           partial, nblocks, L * C, grad_w, grad_b);
    return launch_status("dt_cross_bwd");
}
