// cin.hip — one Compressed-Interaction-Network layer on the matrix cores.
//
// Replaces the per-layer op chain of CIN.call, deeptables/models/layers.py:689-710:
//   split x0 / x_k into D column tensors, tf.matmul(outer product) -> Z [B, D, F0*Hk],
//   tf.nn.conv1d(Z, filters[1, F0*Hk, L]) (+bias), activation, transpose -> [B, L, D].
// The reference materialises Z (354-872 MB per layer at the Criteo shape).  Here the layer is
// the GEMM   Y[(b,d), l] = sum_{k=(i,j)} Z[(b,d), k] * W[k, l],   M = B*D, K = F0*Hk, N = L
// with the A operand Z[(b,d),(i,j)] = x0[b,i,d] * xk[b,j,d] formed in registers (one multiply
// per MFMA step) from LDS-resident x0/xk tiles — Z never exists in memory.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate): bitwise an f32 fmaf chain, so
// the layer meets the reference's fp32 tolerance (1e-4) without a precision mode switch.
// fp32-MFMA roof: 157.3 TFLOP/s.
//
// Three kernels (wave64, 4 waves per block, one 32-row MFMA tile per wave):
//   fwd    block tile 128 rows (b,d) x 128 cols l; W streamed through LDS in 16-row K chunks.
//   dgrad  T^T[(i,j), (b,d)] = sum_l W[(i,j), l] G[(b,d), l] per (i, 32 j's) chunk; the lane that
//          owns column (b,d) contracts its 16 T values against xk / x0 in registers:
//          grad_x0[b,i,d] += sum_j xk T,  grad_xk[b,j,d] += x0 T  — T is never stored either.
//   wgrad  grad_W[(i,j), l] = sum_{(b,d)} Z G with Z regenerated; split over the batch, partial
//          tiles added with float atomics.
#include "common.h"

namespace dt {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kCinKC = 16;    // K rows of W per LDS chunk (fwd)
constexpr int kCinTileM = 128;
constexpr int kCinTileN = 128;

__host__ __device__ inline int cin_slab(int F, int D) {  // floats per batch row in an LDS tile
    int s = F * D;
    if (D < 32) s += ((D - (s % 32)) % 32 + 32) % 32;  // s % 32 == D % 32: two b's never collide
    return s;
}
__host__ __device__ inline int cin_nb(int D) {  // batch rows an m-tile of 128 (b,d) rows can touch
    return (kCinTileM % D == 0) ? kCinTileM / D : kCinTileM / D + 2;
}


template <bool kAnyAct>
__device__ __forceinline__ float cin_act(float v, int act) {
    if (kAnyAct) return act_apply(v, act);
    return act == DT_ACT_RELU ? fmaxf(v, 0.f) : v;
}

// copy the [nb] batch rows starting at b_first of x[B, F, D] (b stride `bstride`) into LDS slabs
__device__ __forceinline__ void stage_rows(const float* __restrict__ x, int64_t bstride, int B, int F,
                                           int D, int b_first, int nb, int slab, float* lds) {
    const int FD = F * D;
    for (int e = threadIdx.x; e < nb * FD; e += blockDim.x) {
        const int bl = e / FD, r = e - bl * FD;
        const int b = b_first + bl;
        lds[bl * slab + r] = b < B ? x[(int64_t)b * bstride + r] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// Round 3: every MFMA operand is a 16-byte LDS read.  An MFMA contraction index may be permuted freely as long as A and
// B agree, so lane (c, s) takes FOUR consecutive k (one float4) and spends them on four consecutive 32x32x2 steps:
// step 4g + jj <-> k' = 8g + 4s + jj.  For the A operand Z[m][(i,j)] = x0[m][i] xk[m][j] that means four consecutive j of
// ONE i: the contraction runs over k' = i HkP + j with HkP = Hk rounded up to 4 (x_k padded with zeros, W rows of a padded j
// read as zero: +7.7 % steps at Hk = 26, none at 64), the x_k tile sits in LDS as [m][j] (j contiguous) and the W chunk
// TRANSPOSED as [n][k'] (k' contiguous).  Round 2's loop read one float per operand and MFMA step (24 ds_read_b32 + the
// (i, j) walk per 16 MFMAs: 42 % of the fp32-MFMA rate); this one 5 ds_read_b128.
// kAnyAct = false: only linear / relu reach the epilogue (no libm code next to the 64 accumulators; with the full
// activation switch inlined there the kernel spills the accumulators and runs 7x slower)
typedef float cin_f4 __attribute__((ext_vector_type(4)));
constexpr int kCinKC4 = 16;                       // k' per W chunk
constexpr int kCinWS = kCinKC4 + 4;               // LDS row stride of a W chunk [n][k']: 20 = 4 x odd -> conflict-free 16-byte reads
__host__ __device__ inline int cin_hkp(int Hk) { return (Hk + 3) & ~3; }
__host__ __device__ inline int cin_xks(int Hk) {  // row stride of the x_k tile [m][j]: 4 x odd
    const int h = cin_hkp(Hk);
    return ((h >> 2) & 1) ? h : h + 4;
}

template <bool kAnyAct>
__global__ __launch_bounds__(256, 2) void k_cin_fwd(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const float* __restrict__ W, const float* __restrict__ bias, int act, int B, int F0, int Hk, int L,
    int D, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int HkP = cin_hkp(Hk), XS = cin_xks(Hk), F0S = F0 | 1;
    const int KP = F0 * HkP;                                // padded contraction length
    const int64_t M = (int64_t)B * D;
    float* xkt = lds;                                       // [128][XS]   x_k[m][j], zero for j >= Hk
    float* wt = xkt + kCinTileM * XS;                       // [2][128][kCinWS]  W chunk, transposed
    float* x0t = wt + 2 * kCinTileN * kCinWS;               // [128][F0S]  x_0[m][i]

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * kCinTileM;
    const int n0 = blockIdx.y * kCinTileN;
    // stage the tiles: element (m, j) from x[b][j][d], m = b D + d (lanes along m: 64-byte pieces of a row at D = 16)
    {
        const int ml = threadIdx.x & (kCinTileM - 1), h = threadIdx.x >> 7;          // 2 threads per row m
        const int64_t m = m0 + ml;
        const bool ok = m < M;
        const int64_t b = ok ? m / D : 0;
        const int d = ok ? (int)(m - b * D) : 0;
        const float* pk = xk + b * xk_bs + d;
        const float* p0 = x0 + b * x0_bs + d;
        for (int j = h; j < XS; j += 2) xkt[ml * XS + j] = (ok && j < Hk) ? pk[(int64_t)j * D] : 0.f;
        for (int i = h; i < F0; i += 2) x0t[ml * F0S + i] = ok ? p0[(int64_t)i * D] : 0.f;
    }
    const int ml = wave * 32 + c;                           // this lane's A-operand row
    const float* xkrow = xkt + ml * XS;
    const float* x0row = x0t + ml * F0S;

    floatx16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    const int nchunks = (KP + kCinKC4 - 1) / kCinKC4;
    // W chunk loader: 256 threads x 8 floats = 16 k' x 128 n, read along n (coalesced), stored transposed
    float wreg[8];
    auto load_w = [&](int chunk) {
        const int kp0 = chunk * kCinKC4;                    // wave-uniform: ONE division per chunk, the rows walk on from it
        const int i0 = kp0 / HkP, j0 = kp0 - i0 * HkP;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = threadIdx.x + 256 * r;
            const int kk = idx >> 7, n = idx & 127;
            int i = i0, j = j0 + kk;
            while (j >= HkP) { j -= HkP; ++i; }
            wreg[r] = (i < F0 && j < Hk && n0 + n < L) ? W[((int64_t)i * Hk + j) * L + n0 + n] : 0.f;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = threadIdx.x + 256 * r;
            wt[buf * kCinTileN * kCinWS + (idx & 127) * kCinWS + (idx >> 7)] = wreg[r];
        }
    };
    load_w(0);
    store_w(0);
    __syncthreads();
    const int nblocks_n = min(4, (L - n0 + 31) / 32);
    int ki = 0, kj = 4 * s;                                 // (i, j0) of this lane's k' = 8g + 4s (+ chunk base), advanced by 8: the two
                                                            // halves of a wave (s = 0, 1) cross into the next i at different steps
    while (kj >= HkP) { kj -= HkP; ++ki; }                  // HkP = 4: the upper half starts in row i = 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) load_w(chunk + 1);
        const float* wb = wt + buf * kCinTileN * kCinWS + c * kCinWS + 4 * s;
#pragma unroll
        for (int g = 0; g < kCinKC4 / 8; ++g) {
            cin_f4 a = {0.f, 0.f, 0.f, 0.f};
            if (ki < F0) a = *reinterpret_cast<const cin_f4*>(xkrow + kj) * x0row[ki];
            kj += 8;
            while (kj >= HkP) { kj -= HkP; ++ki; }       // (HkP may be 4: two rows of i per group)
            cin_f4 bq[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bq[nb] = *reinterpret_cast<const cin_f4*>(wb + nb * 32 * kCinWS + 8 * g);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                if (nb < nblocks_n) {
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[nb].x, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[nb].y, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[nb].z, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[nb].w, acc[nb], 0, 0, 0);
                }
        }
        if (chunk + 1 < nchunks) store_w(buf ^ 1);
        __syncthreads();
    }

    // epilogue: row = (r&3) + 8*(r>>2) + 4*s  (4 consecutive rows per register quad), col = c
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
            float4 o;
            o.x = cin_act<kAnyAct>(acc[nb][g * 4 + 0] + bv, act);
            o.y = cin_act<kAnyAct>(acc[nb][g * 4 + 1] + bv, act);
            o.z = cin_act<kAnyAct>(acc[nb][g * 4 + 2] + bv, act);
            o.w = cin_act<kAnyAct>(acc[nb][g * 4 + 3] + bv, act);
            if (vec_ok) {
                *reinterpret_cast<float4*>(y + (b * L + n) * D + dd) = o;
            } else {
                const float ov[4] = {o.x, o.y, o.z, o.w};
                for (int r = 0; r < 4; ++r) {
                    const int64_t mm = mr + r;
                    if (mm < M) y[((mm / D) * L + n) * D + (mm % D)] = ov[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// dgrad: grad_x0 (+=), grad_xk (=)
// ------------------------------------------------------------------------------------------
template <int LH /* ceil(L/2) upper bound: 64 or 128 */, int JB /* ceil(Hk/32) upper bound */>
__global__ __launch_bounds__(256) void k_cin_dgrad(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const float* __restrict__ W, const float* __restrict__ y, const float* __restrict__ gy, int act,
    int B, int F0, int Hk, int L, int D, float* __restrict__ gx0, float* __restrict__ gxk, int overwrite) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int64_t M = (int64_t)B * D;
    const int S0 = cin_slab(F0, D), Sk = cin_slab(Hk, D), NB = cin_nb(D);
    constexpr int LP = 2 * LH + 4;  // padded W row (lanes walk rows): 4 x odd floats -> conflict-free 16-byte reads
    constexpr int NG = LH / 4;      // groups of 8 l: lane (c, s) holds l = 8g + 4s + jj, jj = 0..3, as ONE float4 (round 3:
                                    // the K permutation of k_cin_fwd — 16 ds_read_b128 per 64 MFMAs instead of 64 ds_read_b32)
    // x_0 is NOT staged (round 3): a lane needs ONE x_0 value per chunk, fetched from L2 before the chunk's 64 MFMAs; without its
    // 14 KB the kernel's LDS (67 KB at the Criteo shape) fits twice per CU
    float* xkt = lds;
    float* wt = lds + ((NB * Sk + 3) & ~3);  // [2][32][LP], 16-byte aligned

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * kCinTileM;
    const int b_first = (int)(m0 / D);
    stage_rows(xk, xk_bs, B, Hk, D, b_first, NB, Sk, xkt);

    const int64_t m = m0 + wave * 32 + c;  // the column of T^T this lane owns
    const bool mvalid = m < M;
    const int64_t b = mvalid ? m / D : 0;
    const int d = mvalid ? (int)(m % D) : 0;
    const int bl = (int)b - b_first;

    // G[m][l] for l = 8g + 4s + jj, kept in registers for the whole tile
    cin_f4 G[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int l = 8 * g + 4 * s + jj;
            float gv = 0.f;
            if (mvalid && l < L) {
                const int64_t o = (b * L + l) * D + d;
                gv = gy[o];
                gv *= act_grad_from_y(y[o], act);
            }
            G[g][jj] = gv;
        }
    }
    __syncthreads();
    // xk values this lane needs: j = jb*32 + (r&3) + 8*(r>>2) + 4*s
    float xkv[JB][16], gxk_acc[JB][16];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * s;
            xkv[jb][r] = (mvalid && j < Hk) ? xkt[bl * Sk + j * D + d] : 0.f;
            gxk_acc[jb][r] = 0.f;
        }

    const int njb = (Hk + 31) / 32;
    const int nchunks = F0 * njb;
    // W chunk (i, jb): rows k = i*Hk + jb*32 + r (r<32, j<Hk), cols l < L  -> wt[r][l]
    const bool wvec = (L % 4 == 0);
    auto stage_w = [&](int chunk, int buf) {
        const int i = chunk / njb, jb = chunk - i * njb;
        float* dst = wt + buf * 32 * LP;
        for (int e = threadIdx.x; e < 32 * (2 * LH / 4); e += 256) {        // 16-byte pieces: rows of W are contiguous in l
            const int r = e / (2 * LH / 4), l = 4 * (e - r * (2 * LH / 4));
            const int j = jb * 32 + r;
            const float* src = W + ((int64_t)i * Hk + j) * L + l;
            cin_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (j < Hk) {
                if (wvec && l + 3 < L) v = *reinterpret_cast<const cin_f4*>(src);
                else {
                    if (l < L) v.x = src[0];
                    if (l + 1 < L) v.y = src[1];
                    if (l + 2 < L) v.z = src[2];
                    if (l + 3 < L) v.w = src[3];
                }
            }
            *reinterpret_cast<cin_f4*>(dst + r * LP + l) = v;
        }
    };
    stage_w(0, 0);
    __syncthreads();
    // grad_x0[b,i,d] += sum_j ...: summed over the i's j-blocks in a register, ONE read-modify-write per i whose read is issued
    // at the i's first chunk (round 2: a dependent global load + store in every chunk)
    float p_acc = 0.f, gx0_old = 0.f;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const int i = chunk / njb, jb = chunk - i * njb;
        if (jb == 0 && mvalid && s == 0 && !overwrite) gx0_old = gx0[(b * F0 + i) * D + d];
        const float x0v = mvalid ? x0[b * x0_bs + (int64_t)i * D + d] : 0.f;
        if (chunk + 1 < nchunks) stage_w(chunk + 1, buf ^ 1);
        const float* wrow = wt + buf * 32 * LP + c * LP + 4 * s;
        floatx16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const cin_f4 a = *reinterpret_cast<const cin_f4*>(wrow + 8 * g);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, G[g].x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, G[g].y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, G[g].z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, G[g].w, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
        // contract T^T[j, m] (16 j's in this lane) against xk and x0
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
        p_acc += p;
        if (jb == njb - 1) {
            if (mvalid && s == 0) gx0[(b * F0 + i) * D + d] = gx0_old + p_acc;  // unique owner of (b,i,d)
            p_acc = 0.f;
        }
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
}

// ------------------------------------------------------------------------------------------
// wgrad: grad_W[k, l] += sum_m Z[m,k] G[m,l]
// ------------------------------------------------------------------------------------------
constexpr int kCinMC = 64;  // (b,d) rows per LDS chunk

constexpr int kCinKT = 2;   // 32-row K sub-tiles per wave: every staged G value feeds 2 MFMAs

// Round 3: the three tiles sit in LDS TRANSPOSED — x0T[i][m], xkT[j][m], gT[l][m], m contiguous — so that lane (c, s) reads
// the four consecutive m = 8g + 4s + jj of its K-permuted MFMA steps as ONE float4 per operand: A = x0T[ki] * xkT[kj]
// (elementwise, Z regenerated), B = gT[l].  8 ds_read_b128 per 32 MFMAs (round 2: 32 ds_read_b32 + 8 multiplies).
constexpr int kCinMS = kCinMC + 4;     // row stride: 68 = 4 x odd

__global__ __launch_bounds__(256, 2) void k_cin_wgrad(
    const float* __restrict__ x0, int64_t x0_bs, const float* __restrict__ xk, int64_t xk_bs,
    const float* __restrict__ y, const float* __restrict__ gy, int act, int B, int F0, int Hk, int L,
    int D, int64_t rows_per_split, float* __restrict__ gW, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = F0 * Hk;
    const int64_t M = (int64_t)B * D;
    float* x0T = lds;                      // [F0][MS]
    float* xkT = x0T + F0 * kCinMS;        // [Hk][MS]
    float* gT = xkT + Hk * kCinMS;         // [128][MS]

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 5, c = lane & 31;
    const int kbase = blockIdx.x * (128 * kCinKT) + wave * (32 * kCinKT);
    const float* arow0[kCinKT];
    const float* arowk[kCinKT];
    float amask[kCinKT];
#pragma unroll
    for (int u = 0; u < kCinKT; ++u) {
        const int k = kbase + 32 * u + c;   // this lane's A row (i,j) in sub-tile u
        const bool kv = k < K;
        const int ki = kv ? k / Hk : 0, kj = kv ? k % Hk : 0;
        arow0[u] = x0T + ki * kCinMS + 4 * s;
        arowk[u] = xkT + kj * kCinMS + 4 * s;
        amask[u] = kv ? 1.f : 0.f;
    }
    const int n0 = blockIdx.z * kCinTileN;
    const int nblocks_n = min(4, (L - n0 + 31) / 32);
    const float* brow = gT + c * kCinMS + 4 * s;
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = min(M, m_begin + rows_per_split);
    // 16-byte staging needs 4 consecutive m inside one batch row and aligned addresses
    const bool vec4 = (D % 4 == 0) && (x0_bs % 4 == 0) && (xk_bs % 4 == 0) &&
                      ((((uintptr_t)x0 | (uintptr_t)xk | (uintptr_t)y | (uintptr_t)gy) & 15) == 0);

    floatx16 acc[kCinKT][4];
#pragma unroll
    for (int u = 0; u < kCinKT; ++u)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][nb][r] = 0.f;

    for (int64_t mc = m_begin; mc < m_end; mc += kCinMC) {
        __syncthreads();
        // stage the transposed tiles: rows mm = mc + r.  A thread always stages the same r = tid % 64 (256 threads, 64 rows per
        // chunk): its (batch row, d) offset is computed once per chunk; lanes run along m, so every LDS store is contiguous
        if (vec4) {
            // 16-byte staging: 4 consecutive m of one tile row are 4 consecutive d of one (b, row) — a float4 load and a
            // float4 LDS store; the thread's loads of a batch are issued before its first store (round 3: the scalar form below
            // waited for ~55 dependent 4-byte loads per thread and chunk, three times the chunk's MFMA time)
            constexpr int QR = kCinMC / 4;                                   // 16-byte pieces per tile row
            constexpr int kStB = 4;                                          // pieces per thread and batch (7: the 128 accumulators spill)
            const int ntask = (F0 + Hk + kCinTileN) * QR;
            const int rows = (int)min((int64_t)kCinMC, m_end - mc);
            for (int t0 = threadIdx.x; t0 < ntask; t0 += 256 * kStB) {
                cin_f4 v[kStB], yv[kStB];
                int dst[kStB];
#pragma unroll
                for (int u = 0; u < kStB; ++u) {
                    const int t = t0 + 256 * u;
                    dst[u] = -1;
                    v[u] = cin_f4{0.f, 0.f, 0.f, 0.f};
                    yv[u] = cin_f4{1.f, 1.f, 1.f, 1.f};
                    if (t >= ntask) continue;
                    const int row = t / QR, q = t - row * QR;
                    dst[u] = row * kCinMS + 4 * q;
                    if (4 * q >= rows) continue;
                    const int64_t m = mc + 4 * q, b = m / D;
                    const int d = (int)(m - b * D);
                    if (row < F0) v[u] = *reinterpret_cast<const cin_f4*>(x0 + b * x0_bs + (int64_t)row * D + d);
                    else if (row < F0 + Hk) v[u] = *reinterpret_cast<const cin_f4*>(xk + b * xk_bs + (int64_t)(row - F0) * D + d);
                    else if (n0 + row - F0 - Hk < L) {
                        const int64_t o = (b * L + n0 + row - F0 - Hk) * D + d;
                        v[u] = *reinterpret_cast<const cin_f4*>(gy + o);
                        if (act != DT_ACT_LINEAR) yv[u] = *reinterpret_cast<const cin_f4*>(y + o);
                    }
                }
#pragma unroll
                for (int u = 0; u < kStB; ++u) {
                    if (dst[u] < 0) continue;
                    cin_f4 o = v[u];
                    if (act != DT_ACT_LINEAR && dst[u] >= (F0 + Hk) * kCinMS) {
                        o.x *= act_grad_from_y(yv[u].x, act); o.y *= act_grad_from_y(yv[u].y, act);
                        o.z *= act_grad_from_y(yv[u].z, act); o.w *= act_grad_from_y(yv[u].w, act);
                    }
                    *reinterpret_cast<cin_f4*>(lds + dst[u]) = o;
                }
            }
        } else {
        const int64_t b0 = mc / D;
        const int d0 = (int)(mc - b0 * D);
        const int rows = (int)min((int64_t)kCinMC, m_end - mc);
        const int r_ = threadIdx.x & (kCinMC - 1), i_first = threadIdx.x / kCinMC;        // element (i, r), i = i_first + 4 k
        const int q_ = d0 + r_, bq_ = q_ / D, dq_ = q_ - bq_ * D;
        const bool rok = r_ < rows;
        const float* x0p = x0 + (b0 + bq_) * x0_bs + dq_;
        const float* xkp = xk + (b0 + bq_) * xk_bs + dq_;
        for (int i = i_first; i < F0; i += 256 / kCinMC) x0T[i * kCinMS + r_] = rok ? x0p[(int64_t)i * D] : 0.f;
        for (int j = i_first; j < Hk; j += 256 / kCinMC) xkT[j * kCinMS + r_] = rok ? xkp[(int64_t)j * D] : 0.f;
        {
            const int64_t ob = ((b0 + bq_) * L + n0) * D + dq_;
            for (int l = i_first; l < kCinTileN; l += 256 / kCinMC) {
                float g = 0.f;
                if (rok && n0 + l < L) {
                    const int64_t o = ob + (int64_t)l * D;
                    g = gy[o];
                    g *= act_grad_from_y(y[o], act);
                }
                gT[l * kCinMS + r_] = g;
            }
        }
        }
        __syncthreads();
#pragma unroll 2
        for (int g = 0; g < kCinMC / 8; ++g) {
            cin_f4 a[kCinKT];
#pragma unroll
            for (int u = 0; u < kCinKT; ++u)
                a[u] = *reinterpret_cast<const cin_f4*>(arow0[u] + 8 * g) * *reinterpret_cast<const cin_f4*>(arowk[u] + 8 * g) * amask[u];
            cin_f4 bq[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bq[nb] = *reinterpret_cast<const cin_f4*>(brow + nb * 32 * kCinMS + 8 * g);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                if (nb < nblocks_n) {
#pragma unroll
                    for (int u = 0; u < kCinKT; ++u) {
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, bq[nb].x, acc[u][nb], 0, 0, 0);
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, bq[nb].y, acc[u][nb], 0, 0, 0);
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, bq[nb].z, acc[u][nb], 0, 0, 0);
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, bq[nb].w, acc[u][nb], 0, 0, 0);
                    }
                }
        }
    }
    // out: row = k_local = (r&3)+8*(r>>2)+4*s, col = l_local = c
#pragma unroll
    for (int u = 0; u < kCinKT; ++u)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb >= nblocks_n) continue;
            const int l = n0 + nb * 32 + c;
            if (l >= L) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kbase + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (kk >= K) continue;
                // part: this batch split's own [K][L] slab (plain 128-byte row stores), summed by k_cin_wgrad_reduce — the
                // 512 blocks of a layer otherwise issue 512 x 256 x 128 = 16.7 M float atomics (round 3: 848 -> see DESIGN)
                if (part) part[((int64_t)blockIdx.y * K + kk) * L + l] = acc[u][nb][r];
                else atomicAdd(&gW[(int64_t)kk * L + l], acc[u][nb][r]);
            }
        }
}

// grad_W[k][l] += sum over the batch splits of part[split][k][l]   (16-byte lanes, the splits' loads in flight together)
__global__ __launch_bounds__(256) void k_cin_wgrad_reduce(const float* __restrict__ part, int splits, int64_t n4,
                                                          float* __restrict__ gW, int overwrite) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    cin_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const cin_f4* p = reinterpret_cast<const cin_f4*>(part) + i;
    int sp = 0;
    for (; sp + 8 <= splits; sp += 8) {
        cin_f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sp + u) * n4];
        acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; sp < splits; ++sp) acc += p[(int64_t)sp * n4];
    cin_f4* g = reinterpret_cast<cin_f4*>(gW) + i;
    *g = overwrite ? acc : *g + acc;
}

// grad_bias[l] += sum_{b,d} G[b,l,d]
__global__ __launch_bounds__(256) void k_cin_bias_grad(const float* __restrict__ y,
                                                       const float* __restrict__ gy, int act, int B,
                                                       int L, int D, float* __restrict__ gbias) {
    const int l = blockIdx.x;
    float sacc = 0.f;
    const int64_t n = (int64_t)B * D;
    for (int64_t e = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.y * blockDim.x) {
        const int64_t b = e / D;
        const int d = (int)(e % D);
        const int64_t o = (b * L + l) * D + d;
        float g = gy[o];
        g *= act_grad_from_y(y[o], act);
        sacc += g;
    }
    sacc = wave_sum(sacc);
    if ((threadIdx.x & 63) == 0) atomicAdd(&gbias[l], sacc);
}

// CIN `direct=False` bookkeeping of one layer output y [B,L,D] (layers.py:713-721): channels [half, L) leave the stack and only
// their sum over D is ever used (tf.reduce_sum(result, -1), :726).  One thread per (b, l): D/4 16-byte loads of one row,
// consecutive threads on consecutive rows (torch's reduce over the strided slice ran at 1 TB/s: 34 us per layer).
__global__ __launch_bounds__(256) void k_cin_pool(const float* __restrict__ y, int64_t n, int L, int D, int half,
                                                  float* __restrict__ pooled) {
    const int LH = L - half, D4 = D >> 2;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / LH;
        const int l = (int)(t - b * LH);
        const cin_f4* p = reinterpret_cast<const cin_f4*>(y + (b * L + half + l) * (int64_t)D);
        cin_f4 a = p[0];
#pragma unroll 4
        for (int i = 1; i < D4; ++i) a += p[i];
        pooled[t] = (a.x + a.y) + (a.z + a.w);
    }
}
// its backward: the layer's incoming gradient gy [B,L,D] assembled in ONE pass — channels < half from g_hidden [B,half,D] (NULL:
// zero), the others the pooled gradient g_pooled [B,L-half] broadcast over D (NULL: zero).  One thread per float4 of gy.
__global__ __launch_bounds__(256) void k_cin_pool_bwd(const float* __restrict__ g_hidden, const float* __restrict__ g_pooled,
                                                      int64_t n4, int L, int D, int half, float* __restrict__ gy) {
    const int D4 = D >> 2, LH = L - half;
    const int64_t row4 = (int64_t)L * D4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / row4;
        const int r = (int)(t - b * row4);
        const int l = r / D4;
        cin_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (l < half) {
            if (g_hidden) v = reinterpret_cast<const cin_f4*>(g_hidden)[b * (int64_t)half * D4 + r];
        } else if (g_pooled) {
            const float g = g_pooled[b * LH + (l - half)];
            v = cin_f4{g, g, g, g};
        }
        reinterpret_cast<cin_f4*>(gy)[t] = v;
    }
}

}  // namespace dt

using namespace dt;

static int cin_check(const char* who, int B, int F0, int Hk, int L, int D) {
    DT_REQUIRE(B >= 0 && F0 > 0 && Hk > 0 && L > 0 && D > 0, "%s: bad sizes B=%d F0=%d Hk=%d L=%d D=%d",
               who, B, F0, Hk, L, D);
    DT_UNSUPPORTED(D > kCinTileM, "%s: D=%d > %d", who, D, kCinTileM);
    return DT_OK;
}

extern "C" int dt_cin_layer_fwd(const float* x0, const float* xk, const float* W, const float* bias,
                                int act, int B, int F0, int Hk, int L, int D, int64_t x0_bstride,
                                int64_t xk_bstride, float* y, void* stream) {
    int rc = cin_check("dt_cin_layer_fwd", B, F0, Hk, L, D);
    if (rc) return rc;
    if (B == 0) return DT_OK;
    DT_REQUIRE(x0 && xk && W && y, "dt_cin_layer_fwd: null pointer");
    DT_REQUIRE(act >= 0 && act < DT_ACT_COUNT, "dt_cin_layer_fwd: act %d", act);
    const size_t lds = ((size_t)kCinTileM * cin_xks(Hk) + 2 * kCinTileN * kCinWS + (size_t)kCinTileM * (F0 | 1)) * sizeof(float);
    DT_UNSUPPORTED(lds > 160 * 1024, "dt_cin_layer_fwd: tiles need %zu B of LDS (> 160 KiB)", lds);
    const int64_t M = (int64_t)B * D;
    dim3 grid((unsigned)((M + kCinTileM - 1) / kCinTileM), (unsigned)ceil_div(L, kCinTileN));
    if (act == DT_ACT_LINEAR || act == DT_ACT_RELU) {
        hipFuncSetAttribute((const void*)k_cin_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_cin_fwd<false>, grid, dim3(256), lds, as_stream(stream), x0, x0_bstride, xk,
                           xk_bstride, W, bias, act, B, F0, Hk, L, D, y);
    } else {
        hipFuncSetAttribute((const void*)k_cin_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_cin_fwd<true>, grid, dim3(256), lds, as_stream(stream), x0, x0_bstride, xk,
                           xk_bstride, W, bias, act, B, F0, Hk, L, D, y);
    }
    return launch_status("dt_cin_layer_fwd");
}

template <int LH, int JB>
static int launch_dgrad(const float* x0, int64_t x0_bs, const float* xk, int64_t xk_bs, const float* W,
                        const float* y, const float* gy, int act, int B, int F0, int Hk, int L, int D,
                        float* gx0, float* gxk, hipStream_t st, int overwrite) {
    const size_t lds = ((size_t)cin_nb(D) * cin_slab(Hk, D) + 2 * 32 * (2 * LH + 4) + 4) * sizeof(float);
    if (lds > 160 * 1024) {
        set_error("dt_cin_layer_bwd: tiles need %zu B of LDS (> 160 KiB)", lds);
        return DT_ERR_UNSUPPORTED;
    }
    const int64_t M = (int64_t)B * D;
    dim3 grid((unsigned)((M + kCinTileM - 1) / kCinTileM));
    hipFuncSetAttribute((const void*)k_cin_dgrad<LH, JB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    hipLaunchKernelGGL((k_cin_dgrad<LH, JB>), grid, dim3(256), lds, st, x0, x0_bs, xk, xk_bs, W, y, gy,
                       act, B, F0, Hk, L, D, gx0, gxk, overwrite);
    return launch_status("dt_cin_layer_bwd(dgrad)");
}

static int cin_wgrad_splits(int B, int F0, int Hk, int L, int D, int64_t* rps_out) {
    const int K = F0 * Hk;
    const int64_t M = (int64_t)B * D;
    const int kblocks = ceil_div(K, 128 * kCinKT), nblocks = ceil_div(L, kCinTileN);
    // exactly one residency round: 256 CUs x 2 blocks (<= 256 registers per lane, 59 KB of LDS per block)
    int splits = 512 / (kblocks * nblocks);
    if (splits < 1) splits = 1;
    int64_t rps = (M + splits - 1) / splits;
    rps = (rps + kCinMC - 1) / kCinMC * kCinMC;
    splits = (int)((M + rps - 1) / rps);
    if (rps_out) *rps_out = rps;
    return splits;
}

// bytes of the optional workspace of dt_cin_layer_bwd_ws: one [K][L] slab per batch split of the weight-gradient kernel
extern "C" int64_t dt_cin_bwd_workspace_bytes(int B, int F0, int Hk, int L, int D) {
    if (B <= 0 || F0 <= 0 || Hk <= 0 || L <= 0 || D <= 0) return 0;
    return (int64_t)cin_wgrad_splits(B, F0, Hk, L, D, nullptr) * F0 * Hk * L * (int64_t)sizeof(float);
}

static int cin_layer_bwd(const float* x0, const float* xk, const float* W, const float* y,
                         const float* grad_y, int act, int B, int F0, int Hk, int L, int D,
                         int64_t x0_bstride, int64_t xk_bstride, float* grad_x0, float* grad_xk,
                         float* grad_W, float* grad_bias, float* ws, void* stream);

extern "C" int dt_cin_layer_bwd(const float* x0, const float* xk, const float* W, const float* y,
                                const float* grad_y, int act, int B, int F0, int Hk, int L, int D,
                                int64_t x0_bstride, int64_t xk_bstride, float* grad_x0, float* grad_xk,
                                float* grad_W, float* grad_bias, void* stream) {
    return cin_layer_bwd(x0, xk, W, y, grad_y, act, B, F0, Hk, L, D, x0_bstride, xk_bstride, grad_x0, grad_xk, grad_W,
                         grad_bias, nullptr, stream);
}

// the same with a workspace of dt_cin_bwd_workspace_bytes (16-byte aligned): the weight-gradient kernel's batch splits store
// their partial [K][L] tiles there and one reduction adds them to grad_W — no float atomics, deterministic
extern "C" int dt_cin_layer_bwd_ws(const float* x0, const float* xk, const float* W, const float* y,
                                   const float* grad_y, int act, int B, int F0, int Hk, int L, int D,
                                   int64_t x0_bstride, int64_t xk_bstride, float* grad_x0, float* grad_xk,
                                   float* grad_W, float* grad_bias, void* ws, void* stream) {
    DT_REQUIRE(ws && ((uintptr_t)ws | (uintptr_t)grad_W) % 16 == 0, "dt_cin_layer_bwd_ws: ws / grad_W null or not 16-byte aligned");
    return cin_layer_bwd(x0, xk, W, y, grad_y, act, B, F0, Hk, L, D, x0_bstride, xk_bstride, grad_x0, grad_xk, grad_W,
                         grad_bias, reinterpret_cast<float*>(ws), stream);
}

static int cin_layer_bwd(const float* x0, const float* xk, const float* W, const float* y,
                         const float* grad_y, int act, int B, int F0, int Hk, int L, int D,
                         int64_t x0_bstride, int64_t xk_bstride, float* grad_x0, float* grad_xk,
                         float* grad_W, float* grad_bias, float* ws, void* stream) {
    int rc = cin_check("dt_cin_layer_bwd", B, F0, Hk, L, D);
    if (rc) return rc;
    if (B == 0) return DT_OK;
    DT_REQUIRE(x0 && xk && W && y && grad_y && grad_x0 && grad_xk && grad_W,
               "dt_cin_layer_bwd: null pointer");
    DT_UNSUPPORTED(L > 256, "dt_cin_layer_bwd: L=%d > 256 filters per layer", L);
    DT_UNSUPPORTED(Hk > 128, "dt_cin_layer_bwd: Hk=%d > 128 hidden feature maps", Hk);
    hipStream_t st = as_stream(stream);
    const int jb = ceil_div(Hk, 32);
#define DT_DGRAD(LHV, JBV)                                                                          \
    rc = launch_dgrad<LHV, JBV>(x0, x0_bstride, xk, xk_bstride, W, y, grad_y, act, B, F0, Hk, L, D, \
                                grad_x0, grad_xk, st, ws ? 1 : 0)
    if (L <= 128) {
        if (jb <= 1) DT_DGRAD(64, 1);
        else if (jb <= 2) DT_DGRAD(64, 2);
        else DT_DGRAD(64, 4);
    } else {
        if (jb <= 1) DT_DGRAD(128, 1);
        else if (jb <= 2) DT_DGRAD(128, 2);
        else DT_DGRAD(128, 4);
    }
#undef DT_DGRAD
    if (rc) return rc;

    const int K = F0 * Hk;
    const int kblocks = ceil_div(K, 128 * kCinKT), nblocks = ceil_div(L, kCinTileN);
    int64_t rps;
    const int splits = cin_wgrad_splits(B, F0, Hk, L, D, &rps);
    const bool slabs = ws && ((int64_t)K * L) % 4 == 0;
    if (ws && !slabs) hipMemsetAsync(grad_W, 0, (size_t)K * L * sizeof(float), st);      // the _ws contract: grad_W is overwritten
    const size_t lds = (size_t)kCinMS * (F0 + Hk + kCinTileN) * sizeof(float);
    hipFuncSetAttribute((const void*)k_cin_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_cin_wgrad, dim3(kblocks, splits, nblocks), dim3(256), lds, st, x0, x0_bstride,
                       xk, xk_bstride, y, grad_y, act, B, F0, Hk, L, D, rps, grad_W, slabs ? ws : nullptr);
    if (slabs) {
        const int64_t n4 = (int64_t)K * L / 4;
        hipLaunchKernelGGL(k_cin_wgrad_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, ws, splits, n4, grad_W, 1);
    }
    if (grad_bias)
        hipLaunchKernelGGL(k_cin_bias_grad, dim3(L, 16), dim3(256), 0, st, y, grad_y, act, B, L, D,
                           grad_bias);
    return launch_status("dt_cin_layer_bwd(wgrad)");
}

// y [B,L,D] -> pooled [B, L-half] = sum_D y[:, half:, :]   (the CIN result channels of one layer; half = 0: all of them)
extern "C" int dt_cin_pool(const float* y, int64_t B, int L, int D, int half, float* pooled, void* stream) {
    DT_REQUIRE(B >= 0 && L > 0 && D > 0 && D % 4 == 0 && half >= 0 && half < L, "dt_cin_pool: bad sizes B=%lld L=%d D=%d half=%d",
               (long long)B, L, D, half);
    if (B == 0) return DT_OK;
    DT_REQUIRE(y && pooled && ((uintptr_t)y % 16 == 0), "dt_cin_pool: null / unaligned pointer");
    const int64_t n = B * (L - half);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_cin_pool, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), y, n, L, D, half, pooled);
    return launch_status("dt_cin_pool");
}

// gy [B,L,D] = concat(g_hidden [B,half,D] or 0, broadcast_D(g_pooled [B,L-half]) or 0): the backward of the split + pool
extern "C" int dt_cin_pool_bwd(const float* g_hidden, const float* g_pooled, int64_t B, int L, int D, int half, float* gy,
                               void* stream) {
    DT_REQUIRE(B >= 0 && L > 0 && D > 0 && D % 4 == 0 && half >= 0 && half < L, "dt_cin_pool_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(gy && ((uintptr_t)gy % 16 == 0) && ((uintptr_t)g_hidden % 16 == 0), "dt_cin_pool_bwd: null / unaligned pointer");
    const int64_t n4 = B * L * (D / 4);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_cin_pool_bwd, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g_hidden, g_pooled, n4, L, D,
                       half, gy);
    return launch_status("dt_cin_pool_bwd");
}
