// product.hip — PNN product layers: InnerProduct and OuterProduct.
//
// Replaces InnerProduct.call  deeptables/models/layers.py:473-487  and
//          OuterProduct.call  deeptables/models/layers.py:543-581.
// The reference materialises p,q = [B,P,D] (P = F(F-1)/2 pairs, 170 MB each at the Criteo
// shape) and, for kernel_type 'mat', a [B,D,P,D] product (2.7 GB).  Here a batch row's
// F x D block is staged once in LDS and every pair is formed from it; nothing of size B*P*D
// ever exists.  Pair order is the reference's: (i<j) row-major, p(i,j) = i*F - i(i+1)/2 + j-i-1.
//
// LDS-tiled elementwise + reduction work (VALU), HBM traffic = x in, [B,P] out.
#include "common.h"

namespace dt {

__device__ __forceinline__ int pair_index(int i, int j, int F) {  // i < j
    return i * F - (i * (i + 1)) / 2 + (j - i - 1);
}

// out[b,p] = scale_p * sum_d x_i[d] x_j[d] * kvec[p,d]    (kvec/knum optional)
__global__ __launch_bounds__(256) void k_pair_dot_fwd(const float* __restrict__ x,
                                                      const float* __restrict__ kvec,
                                                      const float* __restrict__ knum, int B, int F,
                                                      int D, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    const int DP = D + 1;  // padded row stride: pair lanes walk different rows
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float* row = lds + (int64_t)wave * F * DP;
    short* pi = reinterpret_cast<short*>(lds + (int64_t)wpb * F * DP);
    short* pj = pi + P;
    for (int i = threadIdx.x; i < F; i += blockDim.x)
        for (int j = i + 1; j < F; ++j) {
            const int p = pair_index(i, j, F);
            pi[p] = (short)i;
            pj[p] = (short)j;
        }
    __syncthreads();
    for (int b = blockIdx.x * wpb + wave; b < B; b += gridDim.x * wpb) {
        for (int e = lane; e < F * D; e += 64) {
            const int f = e / D, d = e - f * D;
            row[f * DP + d] = x[(int64_t)b * F * D + e];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
        for (int p = lane; p < P; p += 64) {
            const float* a = row + pi[p] * DP;
            const float* c = row + pj[p] * DP;
            float acc = 0.f;
            if (kvec) {
                for (int d = 0; d < D; ++d) acc += a[d] * c[d] * kvec[(int64_t)p * D + d];
            } else {
                for (int d = 0; d < D; ++d) acc += a[d] * c[d];
                if (knum) acc *= knum[p];
            }
            out[(int64_t)b * P + p] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// grad_x[b,f,d] = sum_{o != f} g[b,p(f,o)] * coef(p,d) * x[b,o,d]
// grad_k (vec: [P,D], num: [P]) accumulated in LDS per block, flushed with global atomics.
__global__ __launch_bounds__(256) void k_pair_dot_bwd(
    const float* __restrict__ x, const float* __restrict__ kvec, const float* __restrict__ knum,
    const float* __restrict__ gout, int B, int F, int D, float* __restrict__ gx,
    float* __restrict__ gkvec, float* __restrict__ gknum) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float* row = lds + (int64_t)wave * (F * D + P);
    float* grow = row + F * D;
    float* gk = lds + (int64_t)wpb * (F * D + P);  // [P*D] or [P]
    const int nk = gkvec ? P * D : (gknum ? P : 0);
    for (int i = threadIdx.x; i < nk; i += blockDim.x) gk[i] = 0.f;
    __syncthreads();
    for (int b = blockIdx.x * wpb + wave; b < B; b += gridDim.x * wpb) {
        for (int e = lane; e < F * D; e += 64) row[e] = x[(int64_t)b * F * D + e];
        for (int p = lane; p < P; p += 64) grow[p] = gout[(int64_t)b * P + p];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int e = lane; e < F * D; e += 64) {
            const int f = e / D, d = e - f * D;
            float acc = 0.f;
            for (int o = 0; o < F; ++o) {
                if (o == f) continue;
                const int p = o > f ? pair_index(f, o, F) : pair_index(o, f, F);
                float coef = grow[p];
                if (kvec) coef *= kvec[(int64_t)p * D + d];
                if (knum) coef *= knum[p];
                acc += coef * row[o * D + d];
            }
            gx[(int64_t)b * F * D + e] = acc;
        }
        if (gkvec) {
            for (int t = lane; t < P * D; t += 64) {
                const int p = t / D, d = t - p * D;
                // invert p -> (i,j)
                int i = 0, rem = p;
                while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
                const int j = i + 1 + rem;
                atomicAdd(&gk[t], grow[p] * row[i * D + d] * row[j * D + d]);
            }
        } else if (gknum) {
            for (int p = lane; p < P; p += 64) {
                int i = 0, rem = p;
                while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
                const int j = i + 1 + rem;
                float dot = 0.f;
                for (int d = 0; d < D; ++d) dot += row[i * D + d] * row[j * D + d];
                atomicAdd(&gk[p], grow[p] * dot);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    float* dst = gkvec ? gkvec : gknum;
    for (int i = threadIdx.x; i < nk; i += blockDim.x) atomicAdd(dst + i, gk[i]);
}

// ---- OuterProduct 'mat':  out[b,p] = x_j^T K_p x_i,  K_p[a][d] = K[a,p,d]  ---------------------
// One wave = one pair p x 64 batch rows (lane = row): K_p is wave-uniform (LDS broadcast reads),
// x_i/x_j tiles sit in LDS with a padded stride so lane-private rows are conflict free.
__global__ __launch_bounds__(64) void k_op_mat_fwd(const float* __restrict__ x,
                                                   const float* __restrict__ K, int B, int F, int D,
                                                   float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    const int DP = D + 1;
    float* Kp = lds;                 // [D][D]
    float* xi = lds + D * D;         // [64][DP]
    float* xj = xi + 64 * DP;        // [64][DP]
    const int lane = threadIdx.x;
    const int p = blockIdx.y;
    const int b0 = blockIdx.x * 64;
    int i = 0, rem = p;
    while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
    const int j = i + 1 + rem;
    for (int e = lane; e < D * D; e += 64) {
        const int a = e / D, d = e - a * D;
        Kp[e] = K[((int64_t)a * P + p) * D + d];
    }
    for (int e = lane; e < 64 * D; e += 64) {
        const int r = e / D, d = e - r * D;
        const int b = b0 + r;
        xi[r * DP + d] = b < B ? x[((int64_t)b * F + i) * D + d] : 0.f;
        xj[r * DP + d] = b < B ? x[((int64_t)b * F + j) * D + d] : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    for (int a = 0; a < D; ++a) {
        float u = 0.f;
        for (int d = 0; d < D; ++d) u += xi[lane * DP + d] * Kp[a * D + d];
        acc += u * xj[lane * DP + a];
    }
    if (b0 + lane < B) out[(int64_t)(b0 + lane) * P + p] = acc;
}

// grad wrt x for field f of 64 rows: loops over all partner fields o.
__global__ __launch_bounds__(64) void k_op_mat_bwd_x(const float* __restrict__ x,
                                                     const float* __restrict__ K,
                                                     const float* __restrict__ gout, int B, int F,
                                                     int D, float* __restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    const int DP = D + 1;
    float* Kp = lds;              // [D][D]
    float* xo = lds + D * D;      // [64][DP] partner rows
    float* acc = xo + 64 * DP;    // [64][DP] lane-private accumulators
    const int lane = threadIdx.x;
    const int f = blockIdx.y;
    const int b0 = blockIdx.x * 64;
    const int b = b0 + lane;
    for (int d = 0; d < D; ++d) acc[lane * DP + d] = 0.f;
    for (int o = 0; o < F; ++o) {
        if (o == f) continue;
        const bool f_is_i = f < o;
        const int p = f_is_i ? pair_index(f, o, F) : pair_index(o, f, F);
        __syncthreads();
        for (int e = lane; e < D * D; e += 64) {
            const int a = e / D, d = e - a * D;
            Kp[e] = K[((int64_t)a * P + p) * D + d];
        }
        for (int e = lane; e < 64 * D; e += 64) {
            const int r = e / D, d = e - r * D;
            xo[r * DP + d] = (b0 + r) < B ? x[((int64_t)(b0 + r) * F + o) * D + d] : 0.f;
        }
        __syncthreads();
        const float g = b < B ? gout[(int64_t)b * P + p] : 0.f;
        if (f_is_i) {  // f = i: grad_xi[d] += g * sum_a K[a][d] xj[a]
            for (int d = 0; d < D; ++d) {
                float v = 0.f;
                for (int a = 0; a < D; ++a) v += Kp[a * D + d] * xo[lane * DP + a];
                acc[lane * DP + d] += g * v;
            }
        } else {  // f = j: grad_xj[a] += g * sum_d K[a][d] xi[d]
            for (int a = 0; a < D; ++a) {
                float u = 0.f;
                for (int d = 0; d < D; ++d) u += Kp[a * D + d] * xo[lane * DP + d];
                acc[lane * DP + a] += g * u;
            }
        }
    }
    if (b < B)
        for (int d = 0; d < D; ++d) gx[((int64_t)b * F + f) * D + d] = acc[lane * DP + d];
}

// grad_K[a,p,d] += sum_b g[b,p] x_i[b,d] x_j[b,a]; block = (pair p, batch split), thread = (a,d)
__global__ __launch_bounds__(256) void k_op_mat_bwd_k(const float* __restrict__ x,
                                                      const float* __restrict__ gout, int B, int F,
                                                      int D, int rows_per_split,
                                                      float* __restrict__ gK) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = F * (F - 1) / 2;
    constexpr int TB = 64;
    float* xi = lds;             // [TB][D]
    float* gxj = lds + TB * D;   // [TB][D]  g * x_j
    const int p = blockIdx.x;
    int i = 0, rem = p;
    while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const int r_begin = blockIdx.y * rows_per_split;
    const int r_end = min(B, r_begin + rows_per_split);
    const int nout = D * D;
    // each thread owns outputs t, t+256, ... (a = t / D, d = t % D); up to 16 per thread (D<=64)
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int r0 = r_begin; r0 < r_end; r0 += TB) {
        __syncthreads();
        for (int e = threadIdx.x; e < TB * D; e += blockDim.x) {
            const int r = e / D, d = e - r * D;
            const int b = r0 + r;
            const bool ok = b < r_end;
            const float g = ok ? gout[(int64_t)b * P + p] : 0.f;
            xi[e] = ok ? x[((int64_t)b * F + i) * D + d] : 0.f;
            gxj[e] = ok ? g * x[((int64_t)b * F + j) * D + d] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = threadIdx.x + k * 256;
            if (t < nout) {
                const int a = t / D, d = t - a * D;
                float s = 0.f;
                // (four deep: fully unrolled in each of the 16 k blocks the LDS reads were hoisted across blocks — 572 spilled
                // registers, one wave per SIMD; profiles/r04_kernel_resources.txt)
#pragma unroll 4
                for (int r = 0; r < TB; ++r) s += gxj[r * D + a] * xi[r * D + d];
                acc[k] += s;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = threadIdx.x + k * 256;
        if (t < nout) {
            const int a = t / D, d = t - a * D;
            atomicAdd(&gK[((int64_t)a * P + p) * D + d], acc[k]);
        }
    }
}

static int row_grid(int B, int wpb, int cap) {
    int g = ceil_div(B, wpb);
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}

// ---------------------------------------------------------------------------------------------
// OuterProduct 'mat', D == 16, on v_mfma_f32_16x16x4_f32 (exact fp32).  Kp[a][d] = K[a,p,d].
//   u = x_i Kp^T ; out = sum_a u[a] x_j[a] ; grad x_j = g u ; grad x_i = g (x_j Kp) ; grad Kp = (g x_j)^T x_i
// A operand lane l: A[m = l&15][k = l>>4]; B: B[k = l>>4][n = l&15]; C/D: col = l&15, row = 4*(l>>4) + r.
// A wave owns a 16-row batch tile; pairs are dealt round-robin to the block's 4 waves.
// ---------------------------------------------------------------------------------------------
typedef float op_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void op_pair_ij(int p, int F, int& i, int& j) {
    i = 0;
    int rem = p;
    while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
    j = i + 1 + rem;
}

// forward, pair-major: grid (P, splits); the pair's Kp^T sits in 4 registers per lane (the MFMA B operand) while
// the block's waves walk the 16-row tiles of their split.
__global__ __launch_bounds__(256) void k_op16_fwd(const float* __restrict__ x, const float* __restrict__ K, int B,
                                                  int F, int tiles_per_split, float* __restrict__ out) {
    constexpr int D = 16;
    const int P = F * (F - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x;
    int i, j;
    op_pair_ij(p, F, i, j);
    float kb[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) kb[s4] = K[((int64_t)m * P + p) * D + kq + 4 * s4];   // Kp^T[k = d][n = a = m]
    const int ntiles = (B + 15) / 16;
    const int tile0 = blockIdx.y * tiles_per_split, tile1 = min(ntiles, tile0 + tiles_per_split);
    for (int t = tile0 + wave; t < tile1; t += 4) {
        const int b0 = t * 16;
        const int brow = b0 + m;
        const bool row_ok = brow < B;
        op_f32x4 u = {0.f, 0.f, 0.f, 0.f};
        float xa[4], xj4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xa[s4] = row_ok ? x[((int64_t)brow * F + i) * D + kq + 4 * s4] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + 4 * kq + r;
            xj4[r] = b < B ? x[((int64_t)b * F + j) * D + m] : 0.f;
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) u = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s4], kb[s4], u, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // row 4*kq + r, col a = m: dot with x_j over the 16 cols
            const int b = b0 + 4 * kq + r;
            const float v = group_sum<16>(u[r] * xj4[r]);
            if (m == 0 && b < B) out[(int64_t)b * P + p] = v;
        }
    }
}

struct Op16Regs {           // everything one (pair, 16-row tile) step reads from memory
    float xi[4], xj[4], kt[4], kn[4], g[4];
    int i, j;
};

__device__ __forceinline__ void op16_load(Op16Regs& r, const float* __restrict__ x, const float* __restrict__ K,
                                          const float* __restrict__ gout, const short* __restrict__ pij, int p, int P,
                                          int F, int B, int b0, int m, int kq) {
    constexpr int D = 16;
    r.i = pij[2 * p];
    r.j = pij[2 * p + 1];
    const int brow = b0 + m;
    const bool row_ok = brow < B;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int k = kq + 4 * s4;
        r.xi[s4] = row_ok ? x[((int64_t)brow * F + r.i) * D + k] : 0.f;    // A[row][k = d]
        r.xj[s4] = row_ok ? x[((int64_t)brow * F + r.j) * D + k] : 0.f;    // A[row][k = a]
        r.kt[s4] = K[((int64_t)m * P + p) * D + k];                        // Kp^T[k = d][n = a = m]
        r.kn[s4] = K[((int64_t)k * P + p) * D + m];                        // Kp[k = a][n = d = m]
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = b0 + 4 * kq + q;
        r.g[q] = b < B ? gout[(int64_t)b * P + p] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_op16_bwd_x(const float* __restrict__ x, const float* __restrict__ K,
                                                    const float* __restrict__ gout, int B, int F,
                                                    float* __restrict__ gx) {
    constexpr int D = 16;
    extern __shared__ __attribute__((aligned(16))) float acc[];     // [4 waves][16][F*D] + pair table
    const int P = F * (F - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int FD = F * D;
    short* pij = reinterpret_cast<short*>(acc + 4 * 16 * FD);
    for (int e = threadIdx.x; e < 4 * 16 * FD; e += blockDim.x) acc[e] = 0.f;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        int i, j;
        op_pair_ij(p, F, i, j);
        pij[2 * p] = (short)i;
        pij[2 * p + 1] = (short)j;
    }
    __syncthreads();
    // private per-wave gradient rows + contiguous pair ranges: plain LDS read-modify-writes instead of LDS atomics,
    // grad x_i accumulated in registers while field i stays the same (see k_bil16_bwd_x)
    float* mine = acc + (size_t)wave * 16 * FD;
    const int p_begin = (int)((int64_t)P * wave / 4), p_end = (int)((int64_t)P * (wave + 1) / 4);
    Op16Regs cur, nxt;
    if (p_begin < p_end) op16_load(cur, x, K, gout, pij, p_begin, P, F, B, b0, m, kq);
    float dxi[4] = {0.f, 0.f, 0.f, 0.f};
    int run_i = p_begin < p_end ? cur.i : -1;
    for (int p = p_begin; p < p_end; ++p) {
        if (p + 1 < p_end) op16_load(nxt, x, K, gout, pij, p + 1, P, F, B, b0, m, kq);
        if (cur.i != run_i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { mine[(4 * kq + r) * FD + run_i * D + m] += dxi[r]; dxi[r] = 0.f; }
            run_i = cur.i;
        }
        op_f32x4 u = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            u = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.xi[s4], cur.kt[s4], u, 0, 0, 0);   // x_i Kp^T
            w = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.xj[s4], cur.kn[s4], w, 0, 0, 0);   // x_j Kp
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dxi[r] += cur.g[r] * w[r];                                                 // grad x_i[d = m]
            mine[(4 * kq + r) * FD + cur.j * D + m] += cur.g[r] * u[r];               // grad x_j[a = m]
        }
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

__global__ __launch_bounds__(256) void k_op16_bwd_k(const float* __restrict__ x, const float* __restrict__ gout, int B,
                                                    int F, int tiles_per_split, float* __restrict__ gK) {
    constexpr int D = 16;
    __shared__ float red[4][256];
    const int P = F * (F - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x;
    int i, j;
    op_pair_ij(p, F, i, j);
    const int tile0 = blockIdx.y * tiles_per_split;
    const int ntiles = (B + 15) / 16;
    const int tile1 = min(ntiles, tile0 + tiles_per_split);
    op_f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int t = tile0 + wave; t < tile1; t += 4) {
        const int b0 = t * 16;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int b = b0 + kq + 4 * s4;                  // k = batch row
            float a = 0.f, bv = 0.f;
            if (b < B) {
                a = gout[(int64_t)b * P + p] * x[((int64_t)b * F + j) * D + m];    // A[m = a][k = b] = g x_j[b][a]
                bv = x[((int64_t)b * F + i) * D + m];                              // B[k = b][n = d] = x_i[b][d]
            }
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, c, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * kq + r) * D + m] = c[r];      // [a = row][d = col]
    __syncthreads();
    const int e = threadIdx.x;
    const int a = e / D, d = e - a * D;
    atomicAdd(&gK[((int64_t)a * P + p) * D + d], red[0][e] + red[1][e] + red[2][e] + red[3][e]);
}

}  // namespace dt

using namespace dt;

static int pair_fwd(const float* x, const float* kvec, const float* knum, int B, int F, int D,
                    float* out, void* stream, const char* who) {
    const int P = F * (F - 1) / 2;
    const size_t lds = (size_t)4 * F * (D + 1) * sizeof(float) + (size_t)2 * P * sizeof(short) + 16;
    DT_UNSUPPORTED(lds > 64 * 1024, "%s: F=%d D=%d needs %zu B of LDS (> 64 KiB)", who, F, D, lds);
    hipLaunchKernelGGL(k_pair_dot_fwd, dim3(row_grid(B, 4, 2048)), dim3(256), lds,
                       as_stream(stream), x, kvec, knum, B, F, D, out);
    return launch_status(who);
}

static int pair_bwd(const float* x, const float* kvec, const float* knum, const float* gout, int B,
                    int F, int D, float* gx, float* gkvec, float* gknum, void* stream,
                    const char* who) {
    const int P = F * (F - 1) / 2;
    const int nk = gkvec ? P * D : (gknum ? P : 0);
    const size_t lds = ((size_t)4 * (F * D + P) + nk) * sizeof(float);
    DT_UNSUPPORTED(lds > 64 * 1024, "%s: F=%d D=%d needs %zu B of LDS (> 64 KiB)", who, F, D, lds);
    hipLaunchKernelGGL(k_pair_dot_bwd, dim3(row_grid(B, 4, nk ? 256 : 2048)), dim3(256), lds,
                       as_stream(stream), x, kvec, knum, gout, B, F, D, gx, gkvec, gknum);
    return launch_status(who);
}

extern "C" int dt_inner_product_fwd(const float* x, int B, int F, int D, float* out,
                                    void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0, "dt_inner_product_fwd: bad sizes B=%d F=%d D=%d", B, F, D);
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && out, "dt_inner_product_fwd: null pointer");
    return pair_fwd(x, nullptr, nullptr, B, F, D, out, stream, "dt_inner_product_fwd");
}

extern "C" int dt_inner_product_bwd(const float* x, const float* grad_out, int B, int F, int D,
                                    float* grad_x, void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0, "dt_inner_product_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && grad_out && grad_x, "dt_inner_product_bwd: null pointer");
    return pair_bwd(x, nullptr, nullptr, grad_out, B, F, D, grad_x, nullptr, nullptr, stream,
                    "dt_inner_product_bwd");
}

extern "C" int dt_outer_product_fwd(const float* x, const float* kernel, int kernel_type, int B,
                                    int F, int D, float* out, void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0, "dt_outer_product_fwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && kernel && out, "dt_outer_product_fwd: null pointer");
    if (kernel_type == DT_OP_KERNEL_VEC)
        return pair_fwd(x, kernel, nullptr, B, F, D, out, stream, "dt_outer_product_fwd(vec)");
    if (kernel_type == DT_OP_KERNEL_NUM)
        return pair_fwd(x, nullptr, kernel, B, F, D, out, stream, "dt_outer_product_fwd(num)");
    DT_REQUIRE(kernel_type == DT_OP_KERNEL_MAT, "dt_outer_product_fwd: kernel_type %d", kernel_type);
    const int P = F * (F - 1) / 2;
    if (D == 16) {
        const int ntiles = ceil_div(B, 16);
        int splits16 = ntiles >= 64 ? 8 : 1;
        const int tps = ceil_div(ntiles, splits16);
        splits16 = ceil_div(ntiles, tps);
        hipLaunchKernelGGL(k_op16_fwd, dim3(P, splits16), dim3(256), 0, as_stream(stream), x, kernel, B, F, tps, out);
        return launch_status("dt_outer_product_fwd(mat)");
    }
    const size_t lds = ((size_t)D * D + 2 * 64 * (D + 1)) * sizeof(float);
    DT_UNSUPPORTED(lds > 64 * 1024, "dt_outer_product_fwd(mat): D=%d too large for LDS tiling", D);
    hipLaunchKernelGGL(k_op_mat_fwd, dim3(ceil_div(B, 64), P), dim3(64), lds, as_stream(stream), x,
                       kernel, B, F, D, out);
    return launch_status("dt_outer_product_fwd(mat)");
}

extern "C" int dt_outer_product_bwd(const float* x, const float* kernel, int kernel_type,
                                    const float* grad_out, int B, int F, int D, float* grad_x,
                                    float* grad_kernel, void* stream) {
    DT_REQUIRE(B >= 0 && F >= 2 && D > 0, "dt_outer_product_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && kernel && grad_out && grad_x, "dt_outer_product_bwd: null pointer");
    if (kernel_type == DT_OP_KERNEL_VEC)
        return pair_bwd(x, kernel, nullptr, grad_out, B, F, D, grad_x, grad_kernel, nullptr, stream,
                        "dt_outer_product_bwd(vec)");
    if (kernel_type == DT_OP_KERNEL_NUM)
        return pair_bwd(x, nullptr, kernel, grad_out, B, F, D, grad_x, nullptr, grad_kernel, stream,
                        "dt_outer_product_bwd(num)");
    DT_REQUIRE(kernel_type == DT_OP_KERNEL_MAT, "dt_outer_product_bwd: kernel_type %d", kernel_type);
    const int P = F * (F - 1) / 2;
    const size_t lds = ((size_t)D * D + 2 * 64 * (D + 1)) * sizeof(float);
    DT_UNSUPPORTED(lds > 64 * 1024 || D > 64,
                   "dt_outer_product_bwd(mat): D=%d too large for LDS tiling", D);
    hipStream_t st = as_stream(stream);
    if (D == 16 && (size_t)4 * 16 * F * 16 * sizeof(float) + (size_t)F * F * 2 <= 150 * 1024) {
        const size_t lds16 = (size_t)4 * 16 * F * 16 * sizeof(float) + (size_t)F * (F - 1) * sizeof(short) + 16;
        hipFuncSetAttribute((const void*)k_op16_bwd_x, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipLaunchKernelGGL(k_op16_bwd_x, dim3(ceil_div(B, 16)), dim3(256), lds16, st, x, kernel, grad_out, B, F, grad_x);
        if (grad_kernel) {
            const int ntiles = ceil_div(B, 16);
            int splits16 = ntiles >= 64 ? 8 : 1;
            const int tps = ceil_div(ntiles, splits16);
            splits16 = ceil_div(ntiles, tps);
            hipLaunchKernelGGL(k_op16_bwd_k, dim3(P, splits16), dim3(256), 0, st, x, grad_out, B, F, tps, grad_kernel);
        }
        return launch_status("dt_outer_product_bwd(mat)");
    }
    hipLaunchKernelGGL(k_op_mat_bwd_x, dim3(ceil_div(B, 64), F), dim3(64), lds, st, x, kernel,
                       grad_out, B, F, D, grad_x);
    if (grad_kernel) {
        int splits = ceil_div(B, 1024);
        if (splits > 16) splits = 16;
        const int rps = ceil_div(ceil_div(B, splits), 64) * 64;
        splits = ceil_div(B, rps);
        hipLaunchKernelGGL(k_op_mat_bwd_k, dim3(P, splits), dim3(256),
                           (size_t)2 * 64 * D * sizeof(float), st, x, grad_out, B, F, D, rps,
                           grad_kernel);
    }
    return launch_status("dt_outer_product_bwd(mat)");
}
