// deepfm.hip — the whole DeepFM train step (forward + BCE + backward) as SIX fused launches.
//
// DeepFM = nets ['linear','fm_nets','dnn_nets'] (deeptables/models/deepnets.py:15) assembled by
// DeepModel.__build_model (deeptables/models/deepmodel.py:259-317):
//   emb   = MultiColumnEmbedding(cat)                                   layers.py:889-904
//   x     = Concatenate([Flatten(Concatenate(emb)), dense])             deepmodel.py:269-274,348-353
//   xn    = BatchNormalization('bn_concat_emb_dense')(x)                deepmodel.py:359
//   lin   = Dense(1,no bias)(Concatenate([sum_D(emb), dense]))          deepnets.py:43-66
//   fm    = FM()(Concatenate(emb, axis=1))                              layers.py:53-62
//   dnn   = Dense(1,no bias)(relu(Dense(64)(relu(Dense(128)(xn)))))     deepnets.py:401-427, deepmodel.py:291-292
//   logit = Dense(1, bias)(Add([lin, fm, dnn]))  (sigmoid applied by the loss)   deepmodel.py:296-297,455
// The reference runs this as ~100 small TF ops per step; the generic path of this repo as ~60
// launches.  Here:
//   A  k_sparse_fwd   gather + FM + linear + concat row X + per-block BN statistics      (HBM-bound)
//   B  k_prep + k_bn_final   two-level BN reduction (mean/rstd, moving stats), zero-padded W1 and W1^T
//   C  k_mlp_fwd      X -> BN -> Dense128 -> relu -> Dense64 -> relu -> logits, BCE, dlogit  (fp32 MFMA)
//   D  k_mlp_bwd      dH2, dH1, dXn tiles + per-tile partial sums of every small gradient
//   E  k_wgrad        dW1 = Xn^T dH1, dW2 = H1^T dH2 (fp32 MFMA) + reduction of D's partial sums
//   G  k_sparse_bwd   BN backward + embedding row-gradients (the IndexedSlices values)
// Matrix work uses v_mfma_f32_32x32x2_f32 (exact fp32) so logits stay within 1e-4 of the oracle.
// Every wave owns one 32x32 output tile so that all 1024 SIMDs are busy at B=8192 (256 row tiles x 4).
// All GEMM operands are zero-padded to K = CP (a multiple of 64) so the MFMA loops are straight-line:
// 16-step chunks, operands for the next chunk already in flight (no per-load predication, counted waits).
// No same-address float atomics: per-tile partials + one reduction pass (dW1/dW2 tiles: 4 adds/address).
#include <stdlib.h>
#include "common.h"

namespace dt {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kH1 = 128;  // dnn_params hidden_units[0]
constexpr int kH2 = 64;   // dnn_params hidden_units[1]
constexpr int kTM = 32;   // rows per MLP tile

struct DeepFmDims {
    int B, F, D, Nd, C, CP;  // C = F*D+Nd; CP = C rounded up to 64: row stride of X / dXn / W1T and the padded GEMM K
};

// accumulator buffer layout (floats), zeroed once per step by one memset
struct DeepFmAccum {
    int64_t dW1, dW2, db1, db2, dw3, dwo, dbo, loss, dgamma, dbeta, dwlin, slin, total;
};
__host__ __device__ inline DeepFmAccum deepfm_accum_layout(int C, int CP, int F, int Nd) {
    DeepFmAccum a;
    int64_t o = 0;
    a.dW1 = o; o += (int64_t)C * kH1;
    a.dW2 = o; o += (int64_t)kH1 * kH2;
    a.db1 = o; o += kH1;
    a.db2 = o; o += kH2;
    a.dw3 = o; o += kH2;
    a.dwo = o; o += 1;
    a.dbo = o; o += 1;
    a.loss = o; o += 2;
    a.dgamma = o; o += CP;   // = sum_b dXn * xhat   (also the BN-backward column sum)
    a.dbeta = o; o += CP;    // = sum_b dXn
    a.dwlin = o; o += F + Nd;
    o = (o + 3) & ~(int64_t)3;
    a.slin = o; o += CP;     // scratch: sum_b dz * X per column (field-reduced into dwlin by kernel G)
    a.total = (o + 3) & ~(int64_t)3;
    return a;
}

// Optional row dedupe inside the step (dt_deepfm_train_step, dedupe_ws != NULL): the sparse gradient leaves the step
// with every table row appearing ONCE, so the row-sparse optimizer needs no dedupe pass of its own.
//   A: every valid lookup inserts (row+1) << 32 | occurrence into an open-addressing hash with a 64-bit CAS.  The
//      winner owns the row (mark = its slot); a later lookup of the same row is a duplicate: mark = -(owner)-2, it sets
//      flags[owner], zeroes the owner's gradient row and reports row -1.
//   G: owners without duplicates store their gradient row; owners with duplicates and the duplicates themselves
//      atomicAdd into the owner's row; owners clear their hash slot and flag, so the workspace is all-zero again.
struct DedupeWs {
    unsigned long long* slots;   // [1 << slots_log2], zero outside a step
    int slots_log2;
    int* mark;                   // [B*F]
    int* flags;                  // [B*F], zero outside a step
};

// ---------------------------------------------------------------------------------------------
// A: sparse forward.  One wave = one batch row, 16 waves (16 rows) per block: 8192 waves at B = 8192, all
//    resident at once (2 blocks of 1024 threads per CU) so the two dependent HBM round trips of a gather
//    (ids -> table rows) overlap across 32 waves per CU.  The block's 16 rows meet in LDS for the BN statistics.
// ---------------------------------------------------------------------------------------------
constexpr int kRowsPerBlockA = 16;
constexpr int kMaxC = 544;

template <int KIND, int LPR>
__global__ __launch_bounds__(1024) void k_sparse_fwd(
    const void* __restrict__ idx, const float4* __restrict__ table, const int64_t* __restrict__ row_offset,
    const int32_t* __restrict__ vocab, const float* __restrict__ dense, const float* __restrict__ wlin,
    DeepFmDims dm, float* __restrict__ X, float* __restrict__ lin_out, float* __restrict__ fm_out,
    int64_t* __restrict__ rows_out, int* __restrict__ oob, float* __restrict__ bn_partial, DedupeWs dd,
    float* __restrict__ grad_rows) {
    __shared__ __attribute__((aligned(16))) float rowbuf[kRowsPerBlockA][kMaxC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & (LPR - 1);
    const int NV = dm.F * LPR;  // float4 per row (<= 128)
    const int D = 4 * LPR;
    const int b = blockIdx.x * kRowsPerBlockA + wave;
    if (b < dm.B) {
        float4 v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = lane + 64 * t;
            v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < NV) {
                const int f = j / LPR;
                const int id = load_id<KIND>(idx, (int64_t)b * dm.F + f);
                const bool ok = (unsigned)id < (unsigned)vocab[f];
                const int64_t row = ok ? row_offset[f] + id : (int64_t)-1;
                if (ok) v[t] = table[row * LPR + c];
                if (c == 0) {
                    const int64_t occ = (int64_t)b * dm.F + f;
                    int64_t row_w = row;
                    if (dd.slots) {
                        // ownership of the row for this step (see DedupeWs): the first lookup to claim the hash
                        // slot owns it, later ones become duplicates that G folds into the owner's gradient row
                        int mk = -1;
                        if (ok) {
                            const unsigned mask = (1u << dd.slots_log2) - 1u;
                            unsigned h = hash_row((unsigned)row, 32 - dd.slots_log2);
                            const unsigned long long mine =
                                ((unsigned long long)(row + 1) << 32) | (unsigned long long)(unsigned)occ;
                            for (;;) {
                                const unsigned long long prev = atomicCAS(&dd.slots[h], 0ULL, mine);
                                if (prev == 0ULL) { mk = (int)h; break; }
                                if ((prev >> 32) == (unsigned long long)(row + 1)) {
                                    const int64_t owner = (int64_t)(prev & 0xffffffffULL);
                                    mk = -(int)owner - 2;
                                    dd.flags[owner] = 1;                     // the owner's row is accumulated atomically:
                                    float4* z = reinterpret_cast<float4*>(grad_rows + owner * D);   // start it from 0
                                    for (int q = 0; q < LPR; ++q) z[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                                    row_w = -1;                              // not a separate row of the gradient
                                    break;
                                }
                                h = (h + 1) & mask;
                            }
                        }
                        dd.mark[occ] = mk;
                    }
                    rows_out[occ] = row_w;
                    if (!ok && oob) atomicAdd(oob, 1);
                }
            }
        }
        const float dv = lane < dm.Nd ? dense[(int64_t)b * dm.Nd + lane] : 0.f;
        float lp = dv * (lane < dm.Nd ? wlin[dm.F + lane] : 0.f);
        float4 S = make_float4(0.f, 0.f, 0.f, 0.f), Q = S;
        float* xrow = X + (int64_t)b * dm.CP;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = lane + 64 * t;
            const float4 x = v[t];
            if (j < NV) {
                *reinterpret_cast<float4*>(xrow + 4 * j) = x;
                *reinterpret_cast<float4*>(&rowbuf[wave][4 * j]) = x;
                lp += ((x.x + x.y) + (x.z + x.w)) * wlin[j / LPR];
            }
            S.x += x.x; S.y += x.y; S.z += x.z; S.w += x.w;
            Q.x += x.x * x.x; Q.y += x.y * x.y; Q.z += x.z * x.z; Q.w += x.w * x.w;
        }
        for (int k = lane; k < dm.CP - dm.F * D; k += 64)   // dense columns, then zero padding up to CP
            xrow[dm.F * D + k] = k < dm.Nd ? dv : 0.f;
        if (lane < dm.Nd) rowbuf[wave][dm.F * D + lane] = dv;
        S.x = wave_sum_strided<LPR>(S.x); S.y = wave_sum_strided<LPR>(S.y);
        S.z = wave_sum_strided<LPR>(S.z); S.w = wave_sum_strided<LPR>(S.w);
        Q.x = wave_sum_strided<LPR>(Q.x); Q.y = wave_sum_strided<LPR>(Q.y);
        Q.z = wave_sum_strided<LPR>(Q.z); Q.w = wave_sum_strided<LPR>(Q.w);
        float ts = ((S.x * S.x - Q.x) + (S.y * S.y - Q.y)) + ((S.z * S.z - Q.z) + (S.w * S.w - Q.w));
        ts = group_sum<LPR>(ts);
        lp = wave_sum(lp);
        if (lane == 0) {
            fm_out[b] = 0.5f * ts;
            lin_out[b] = lp;
        }
    }
    __syncthreads();
    // ---- BN statistics of this block's rows: exact two-pass {n, mean, M2} per column ----
    const int nrows = min(kRowsPerBlockA, dm.B - (int)blockIdx.x * kRowsPerBlockA);
    for (int col = threadIdx.x; col < dm.C; col += blockDim.x) {
        float sum = 0.f;
        for (int w = 0; w < nrows; ++w) sum += rowbuf[w][col];
        const float mean = nrows > 0 ? sum / (float)nrows : 0.f;
        float m2 = 0.f;
        for (int w = 0; w < nrows; ++w) {
            const float d = rowbuf[w][col] - mean;
            m2 += d * d;
        }
        float* p = bn_partial + (int64_t)blockIdx.x * 3 * dm.C;
        p[col] = (float)max(nrows, 0);
        p[dm.C + col] = mean;
        p[2 * dm.C + col] = m2;
    }
}

// ---------------------------------------------------------------------------------------------
// B: BN finalize + zero-padded W1P [CP][H1] and W1T [H1][CP] + zeroing of dW1|dW2
//    BN blocks: 64 columns x 16 waves; lanes run along the columns (256-byte coalesced partial rows), every
//    wave Chan-merges a slice of the chunk list, the 16 slices meet in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int kBnSlices = 16;   // level-1 BN reduction blocks per 64-column group

struct PrepOut {
    float *mean, *rstd, *sc, *beta, *W1P, *W1T;   // all padded to CP
    float* bn2;                                   // [kBnSlices][3][C] level-1 results
};

__global__ __launch_bounds__(1024) void k_prep(const float* __restrict__ partial, int chunks, DeepFmDims dm,
                                               float eps, float momentum, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float* __restrict__ moving_mean,
                                               float* __restrict__ moving_var, const float* __restrict__ W1,
                                               PrepOut o, int bn_blocks, float* __restrict__ zero_region,
                                               int zero_floats) {
    if ((int)blockIdx.x >= bn_blocks) {  // weight copies + zeroing of the atomically accumulated dW1/dW2
        __shared__ float tile[32][33];
        const int wb = (int)blockIdx.x - bn_blocks, nwb = gridDim.x - bn_blocks;
        for (int e = wb * blockDim.x + threadIdx.x; e < zero_floats; e += nwb * blockDim.x) zero_region[e] = 0.f;
        const int total = kH1 * dm.CP;
        for (int e = wb * blockDim.x + threadIdx.x; e < total; e += nwb * blockDim.x) {   // W1P[col][k]: plain copy
            const int col = e / kH1;
            o.W1P[e] = col < dm.C ? W1[e] : 0.f;
        }
        // W1T[k][col] = W1[col][k]: 32x32 tiles through LDS so that both the read (along k) and the write
        // (along col) are coalesced
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 32 threads
        const int ctiles = dm.CP >> 5, ktiles = kH1 >> 5;
        for (int t = wb; t < ctiles * ktiles; t += nwb) {
            const int c0 = (t / ktiles) << 5, k0 = (t % ktiles) << 5;
            __syncthreads();
            tile[ty][tx] = (c0 + ty) < dm.C ? W1[(int64_t)(c0 + ty) * kH1 + k0 + tx] : 0.f;
            __syncthreads();
            o.W1T[(int64_t)(k0 + ty) * dm.CP + c0 + tx] = tile[tx][ty];
        }
        return;
    }
    // level 1 of the BN reduction: block = (64-column group, slice of the chunk list), ONE wave, lanes along the
    // columns (256-byte coalesced partial rows).  A CU pulls only ~20 GB/s from HBM, so the 2.6 MB of partials
    // must be spread over >= 100 CUs; level 2 (k_bn_final) merges the kBnSlices results per column.
    const int lane = threadIdx.x & 63;
    if (threadIdx.x >= 64) return;
    const int cgroups = (dm.C + 63) >> 6;
    const int cg = blockIdx.x % cgroups, slice = blockIdx.x / cgroups;
    const int col = cg * 64 + lane;
    const bool cok = col < dm.C;
    const int per = (chunks + kBnSlices - 1) / kBnSlices;
    const int k0 = slice * per, k1 = min(chunks, k0 + per);
    float n = 0.f, mean = 0.f, m2 = 0.f;
    constexpr int UB = 16;
    for (int kb = k0; kb < k1; kb += UB) {
        float nb[UB], mb[UB], qb[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = kb + u;
            const bool ok = cok && k < k1;
            const float* p = partial + (int64_t)(ok ? k : 0) * 3 * dm.C + (cok ? col : 0);
            // unconditional loads from a clamped (always valid) address, selected afterwards
            const float v0 = p[0], v1 = p[dm.C], v2 = p[2 * dm.C];
            nb[u] = ok ? v0 : 0.f;
            mb[u] = ok ? v1 : 0.f;
            qb[u] = ok ? v2 : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (nb[u] <= 0.f) continue;
            const float nt = n + nb[u];
            const float delta = mb[u] - mean;
            mean += delta * (nb[u] / nt);
            m2 += qb[u] + delta * delta * (n * nb[u] / nt);
            n = nt;
        }
    }
    if (cok) {
        float* q = o.bn2 + (int64_t)slice * 3 * dm.C;
        q[col] = n;
        q[dm.C + col] = mean;
        q[2 * dm.C + col] = m2;
    }
}

// level 2: merge the kBnSlices per-column results, publish mean / rstd / scale / beta (padded) + moving stats
__global__ __launch_bounds__(256) void k_bn_final(DeepFmDims dm, float eps, float momentum,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                  PrepOut o) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= dm.CP) return;
    if (col >= dm.C) {  // pad columns
        o.mean[col] = 0.f; o.rstd[col] = 0.f; o.sc[col] = 0.f; o.beta[col] = 0.f;
        return;
    }
    float nb[kBnSlices], mb[kBnSlices], qb[kBnSlices];
#pragma unroll
    for (int w = 0; w < kBnSlices; ++w) {
        const float* q = o.bn2 + (int64_t)w * 3 * dm.C + col;
        nb[w] = q[0]; mb[w] = q[dm.C]; qb[w] = q[2 * dm.C];
    }
    float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
    for (int w = 0; w < kBnSlices; ++w) {
        if (nb[w] <= 0.f) continue;
        const float nt = n + nb[w];
        const float delta = mb[w] - mean;
        mean += delta * (nb[w] / nt);
        m2 += qb[w] + delta * delta * (n * nb[w] / nt);
        n = nt;
    }
    const float var = n > 0.f ? m2 / n : 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    o.mean[col] = mean;
    o.rstd[col] = rstd;
    o.sc[col] = rstd * gamma[col];
    o.beta[col] = beta[col];
    if (moving_mean) moving_mean[col] = moving_mean[col] * momentum + mean * (1.f - momentum);
    if (moving_var) moving_var[col] = moving_var[col] * momentum + var * (1.f - momentum);
}

// ---------------------------------------------------------------------------------------------
// per-tile partial sums written by C/D and reduced by E (layout of one tile's record, floats)
// ---------------------------------------------------------------------------------------------
struct PartLayout {
    int sg, sgx, slin, db1, db2, dw3, dwo, dbo, loss, stride;
};
__host__ __device__ inline PartLayout part_layout(int CP) {
    PartLayout l;
    l.sg = 0; l.sgx = CP; l.slin = 2 * CP;
    l.db1 = 3 * CP; l.db2 = l.db1 + kH1; l.dw3 = l.db2 + kH2;
    l.dwo = l.dw3 + kH2; l.dbo = l.dwo + 1; l.loss = l.dbo + 1;
    l.stride = (l.loss + 1 + 3) & ~3;
    return l;
}

// ---------------------------------------------------------------------------------------------
// C: MLP forward on a 32-row tile.  4 waves; GEMM1: wave w owns hidden units [32w, 32w+32).
// ---------------------------------------------------------------------------------------------
struct MlpParams {
    const float *W1P, *b1, *W2, *b2, *w3, *wo, *bo, *gamma, *mean, *rstd, *sc, *betap;
};

constexpr int kCH = 16;   // MFMA steps per operand chunk

// phase timestamps (s_memtime, shader cycles) of wave 0 of every block: ws region `stamps` [blocks][8] u64,
// read back by tools/phase_times.py; costs one scalar load + store per phase
#define DT_STAMP(buf, slot)                                                            \
    do {                                                                               \
        if ((buf) && threadIdx.x == 0)                                                 \
            (buf)[(int64_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();    \
    } while (0)

__global__ __launch_bounds__(256) void k_mlp_fwd(const float* __restrict__ X, MlpParams p, DeepFmDims dm,
                                                 const float* __restrict__ lin, const float* __restrict__ fm,
                                                 const float* __restrict__ y, float* __restrict__ H1,
                                                 float* __restrict__ H2, float* __restrict__ z_out,
                                                 float* __restrict__ logit_out, float* __restrict__ dlogit,
                                                 float* __restrict__ part, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    DT_STAMP(stamps, 0);
    const int XS = dm.CP + 1;  // odd stride: lanes walk rows conflict-free
    float* xn = lds;                    // [32][XS]
    float* h1 = xn + kTM * XS;          // [32][129]
    float* red = h1 + kTM * (kH1 + 1);  // [2][32][65]
    float* h2 = red + 2 * kTM * (kH2 + 1);  // [32][65]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane >> 5, c = lane & 31;
    const int m0 = blockIdx.x * kTM;
    const PartLayout pl = part_layout(dm.CP);

    // GEMM2's B operands (W2, L2-resident) do not depend on anything computed here: fetch them now so their
    // latency hides under the staging and GEMM1 phases (one block per CU = no other wave to hide it)
    float w2q[32];
    {
        const int nb2 = wave & 1, kh2 = wave >> 1;
        const float* bcol2 = p.W2 + (int64_t)(64 * kh2 + s) * kH2 + 32 * nb2 + c;
#pragma unroll
        for (int st = 0; st < 32; ++st) w2q[st] = bcol2[(int64_t)2 * st * kH2];
    }
    const float bias1 = p.b1[32 * wave + c];

    // ---- stage Xn = BN(X): pad columns carry sc = beta = 0 -> Xn = 0.  The X tile was written by another
    // XCD a moment ago (HBM/MALL latency): issue every load of this thread before touching any result. ----
    const int q4 = dm.CP >> 2;
    {
        constexpr int U = 8;
        const int total = kTM * q4;
        for (int e0 = threadIdx.x; e0 < total; e0 += blockDim.x * U) {
            float4 xv[U], mu[U], sc[U], be[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * blockDim.x;
                const int r = e / q4, q = e - r * q4;
                const int m = m0 + r;
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < total) {
                    if (m < dm.B) xv[u] = *reinterpret_cast<const float4*>(X + (int64_t)m * dm.CP + 4 * q);
                    mu[u] = *reinterpret_cast<const float4*>(p.mean + 4 * q);
                    sc[u] = *reinterpret_cast<const float4*>(p.sc + 4 * q);
                    be[u] = *reinterpret_cast<const float4*>(p.betap + 4 * q);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * blockDim.x;
                if (e < total) {
                    const int r = e / q4, q = e - r * q4;
                    float* dst = xn + r * XS + 4 * q;
                    dst[0] = (xv[u].x - mu[u].x) * sc[u].x + be[u].x;
                    dst[1] = (xv[u].y - mu[u].y) * sc[u].y + be[u].y;
                    dst[2] = (xv[u].z - mu[u].z) * sc[u].z + be[u].z;
                    dst[3] = (xv[u].w - mu[u].w) * sc[u].w + be[u].w;
                }
            }
        }
    }
    lds_barrier();
    DT_STAMP(stamps, 1);

    // ---- GEMM1: [32 x CP] . [CP x 128], zero padded: straight-line chunks of 16 MFMA steps ----
    // A dependent MFMA chain on ONE accumulator pays ~+43 cycles for every instruction (s_waitcnt, address math)
    // the compiler leaves between two links; alternating two accumulators hides that gap behind the other chain.
    floatx16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    {
        const float* arow = xn + c * XS + s;                       // A[row c][k = 2 st + s]
        const float* bcol = p.W1P + (int64_t)s * kH1 + 32 * wave + c;   // B[k][col]
        const int nchunks = dm.CP / (2 * kCH);
        // THREE operand buffers: chunk j+2 is issued while chunk j runs.  With two buffers the compiler's
        // s_waitcnt in front of a chain also waits for the first load of the chunk issued just before it
        // (vmcnt is conservative by one across the loop back-edge) and the chain starts a full memory latency late.
        float a0[kCH], b0[kCH], a1[kCH], b1[kCH], a2[kCH], b2[kCH];
        auto load = [&](float (&aq)[kCH], float (&bq)[kCH], int ch) {
            const float* an = arow + 2 * kCH * ch;
            const float* bn = bcol + (int64_t)2 * kCH * ch * kH1;
#pragma unroll
            for (int i = 0; i < kCH; ++i) { aq[i] = an[2 * i]; bq[i] = bn[(int64_t)2 * i * kH1]; }
        };
        auto run = [&](const float (&aq)[kCH], const float (&bq)[kCH]) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kCH; i += 2) {   // two independent accumulator chains (see acc2 above)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i], bq[i], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i + 1], bq[i + 1], acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        load(a0, b0, 0);
        if (nchunks > 1) load(a1, b1, 1);
        for (int ch = 0; ch < nchunks; ch += 3) {
            if (ch + 2 < nchunks) load(a2, b2, ch + 2);
            run(a0, b0);
            if (ch + 1 >= nchunks) break;
            if (ch + 3 < nchunks) load(a0, b0, ch + 3);
            run(a1, b1);
            if (ch + 2 >= nchunks) break;
            if (ch + 4 < nchunks) load(a1, b1, ch + 4);
            run(a2, b2);
        }
    }
    DT_STAMP(stamps, 2);
    {
        const float bias = bias1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
            const float h = fmaxf((acc[r] + acc2[r]) + bias, 0.f);
            h1[row * (kH1 + 1) + 32 * wave + c] = h;
            if (m0 + row < dm.B) H1[(int64_t)(m0 + row) * kH1 + 32 * wave + c] = h;
        }
    }
    lds_barrier();
    DT_STAMP(stamps, 3);

    // ---- GEMM2: [32 x 128] . [128 x 64]; wave = (n-block nb, k-half kh) ----
    {
        const int nb = wave & 1, kh = wave >> 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* arow = h1 + c * (kH1 + 1) + 64 * kh + s;
        float aq[32];
        const float (&bq)[32] = w2q;
#pragma unroll
        for (int st = 0; st < 32; ++st) aq[st] = arow[2 * st];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 32; st += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st], bq[st], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st + 1], bq[st + 1], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
            red[(kh * kTM + row) * (kH2 + 1) + 32 * nb + c] = acc[r] + acc2[r];
        }
    }
    lds_barrier();
    for (int e = threadIdx.x; e < kTM * kH2; e += blockDim.x) {
        const int row = e / kH2, n = e - row * kH2;
        const float v = red[row * (kH2 + 1) + n] + red[(kTM + row) * (kH2 + 1) + n] + p.b2[n];
        const float h = fmaxf(v, 0.f);
        h2[row * (kH2 + 1) + n] = h;
        if (m0 + row < dm.B) H2[(int64_t)(m0 + row) * kH2 + n] = h;
    }
    lds_barrier();
    DT_STAMP(stamps, 4);

    // ---- logits, loss, dlogit (wave 0) ----
    if (wave == 0) {
        float pt = 0.f;
        const float* hrow = h2 + c * (kH2 + 1) + 32 * s;
#pragma unroll 8
        for (int n = 0; n < 32; ++n) pt += hrow[n] * p.w3[32 * s + n];
        pt += __shfl_xor(pt, 32, 64);
        const int m = m0 + c;
        float loss = 0.f;
        if (s == 0 && m < dm.B) {
            const float z = (lin[m] + fm[m]) + pt;   // Add([linear, fm, dnn]) order
            const float lg = z * p.wo[0] + (p.bo ? p.bo[0] : 0.f);
            const float yy = y[m];
            const float pr = 1.0f / (1.0f + expf(-lg));
            loss = fmaxf(lg, 0.f) - lg * yy + log1pf(expf(-fabsf(lg)));
            z_out[m] = z;
            logit_out[m] = lg;
            dlogit[m] = (pr - yy) / (float)dm.B;
        }
        loss = wave_sum(loss);
        if (lane == 0) part[(int64_t)blockIdx.x * pl.stride + pl.loss] = loss / (float)dm.B;
    }
    DT_STAMP(stamps, 5);
}

// ---------------------------------------------------------------------------------------------
// D: MLP backward on a 32-row tile
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mlp_bwd(const float* __restrict__ X, MlpParams p,
                                                 const float* __restrict__ W1T, DeepFmDims dm,
                                                 const float* __restrict__ H1, const float* __restrict__ H2,
                                                 const float* __restrict__ z, const float* __restrict__ dlogit,
                                                 float* __restrict__ dH1, float* __restrict__ dH2,
                                                 float* __restrict__ dXn, float* __restrict__ dz_out,
                                                 float* __restrict__ part, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    DT_STAMP(stamps, 0);
    float* w2s = lds;                        // [128][65]
    float* dh2 = w2s + kH1 * (kH2 + 1);      // [32][65]
    float* dh1 = dh2 + kTM * (kH2 + 1);      // [32][129]
    float* dzs = dh1 + kTM * (kH1 + 1);      // [32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane >> 5, c = lane & 31;
    const int m0 = blockIdx.x * kTM;
    const PartLayout pl = part_layout(dm.CP);
    float* prec = part + (int64_t)blockIdx.x * pl.stride;

    // W2 (32 KB) and this tile's H2 (8 KB): all 16-byte loads of a thread in flight together
    float4 w2r[(kH1 * kH2 / 4) / 256], h2r[(kTM * kH2 / 4) / 256];
#pragma unroll
    for (int u = 0; u < (kH1 * kH2 / 4) / 256; ++u)
        w2r[u] = *reinterpret_cast<const float4*>(p.W2 + 4 * (threadIdx.x + 256 * u));
#pragma unroll
    for (int u = 0; u < (kTM * kH2 / 4) / 256; ++u) {
        const int e4 = threadIdx.x + 256 * u;                    // float4 index in the [32][64] tile
        const int m = m0 + e4 / (kH2 / 4);
        h2r[u] = m < dm.B ? *reinterpret_cast<const float4*>(H2 + (int64_t)m * kH2 + 4 * (e4 % (kH2 / 4)))
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // everything that does not depend on values computed in this kernel is requested now, AFTER the loads the prologue needs first
    // (vmcnt retires in order) (one block per CU:
    // nothing else hides HBM/MALL latency): the relu masks of dH1, and the first dXn operand block
    float h1v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * s;
        h1v[r] = m < dm.B ? H1[(int64_t)m * kH1 + 32 * wave + c] : 0.f;
    }
    float bq0[kH1 / 2], bq1[kH1 / 2], xv0[16], xv1[16], mr0[2], mr1[2];
    auto load_nb = [&](float (&bq)[kH1 / 2], float (&xv)[16], float (&mr)[2], int nb) {
        const int col = 32 * nb + c;
        mr[0] = p.mean[col];
        mr[1] = p.rstd[col];
        const float* bcol = W1T + (int64_t)s * dm.CP + col;
#pragma unroll
        for (int st = 0; st < kH1 / 2; ++st) bq[st] = bcol[(int64_t)2 * st * dm.CP];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
            xv[r] = X[(int64_t)(m0 + row) * dm.CP + col];      // workspace rows are padded to the tile: in range
        }
    };
    const int nblocks = dm.CP >> 5;
    if (wave < nblocks) load_nb(bq0, xv0, mr0, wave);
    if (threadIdx.x < kTM) {
        const int m = m0 + threadIdx.x;
        float dl = 0.f, zz = 0.f;
        if (m < dm.B) { dl = dlogit[m]; zz = z[m]; }
        const float dzv = dl * p.wo[0];
        dzs[threadIdx.x] = dzv;
        if (m < dm.B) dz_out[m] = dzv;
        float a = dl * zz, b = dl;   // d task_output kernel / bias
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
        if (threadIdx.x == 0) { prec[pl.dwo] = a; prec[pl.dbo] = b; }
    }
#pragma unroll
    for (int u = 0; u < (kH1 * kH2 / 4) / 256; ++u) {
        const int e = 4 * (threadIdx.x + 256 * u);
        const int k = e / kH2, n = e - k * kH2;
        float* dst = w2s + k * (kH2 + 1) + n;
        dst[0] = w2r[u].x; dst[1] = w2r[u].y; dst[2] = w2r[u].z; dst[3] = w2r[u].w;
    }
    lds_barrier();
    // dH2 tile (H2 parked in the not-yet-used dh1 buffer for the column sums below)
#pragma unroll
    for (int u = 0; u < (kTM * kH2 / 4) / 256; ++u) {
        const int e4 = threadIdx.x + 256 * u;
        const int row = e4 / (kH2 / 4), n = 4 * (e4 % (kH2 / 4));
        const int m = m0 + row;
        const float hv[4] = {h2r[u].x, h2r[u].y, h2r[u].z, h2r[u].w};
        float gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gv[i] = hv[i] > 0.f ? dzs[row] * p.w3[n + i] : 0.f;
            dh2[row * (kH2 + 1) + n + i] = gv[i];
            dh1[row * (kH1 + 1) + n + i] = hv[i];
        }
        if (m < dm.B) *reinterpret_cast<float4*>(dH2 + (int64_t)m * kH2 + n) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    }
    lds_barrier();
    if (threadIdx.x < kH2) {
        const int n = threadIdx.x;
        float sw = 0.f, sb = 0.f;
        for (int row = 0; row < kTM; ++row) {
            sw += dzs[row] * dh1[row * (kH1 + 1) + n];
            sb += dh2[row * (kH2 + 1) + n];
        }
        prec[pl.dw3 + n] = sw;
        prec[pl.db2 + n] = sb;
    }
    lds_barrier();
    DT_STAMP(stamps, 1);

    // dH1 = dH2 . W2^T  (wave w -> hidden units [32w, 32w+32))
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const float* arow = dh2 + c * (kH2 + 1) + s;
        const float* brow = w2s + (32 * wave + c) * (kH2 + 1) + s;
        float aq[kH2 / 2], bq[kH2 / 2];
#pragma unroll
        for (int st = 0; st < kH2 / 2; ++st) { aq[st] = arow[2 * st]; bq[st] = brow[2 * st]; }
        floatx16 accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
        for (int st = 0; st < kH2 / 2; st += 2) {   // two accumulator chains (see k_mlp_fwd)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st], bq[st], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st + 1], bq[st + 1], accb, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += accb[r];
        float colsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
            const int m = m0 + row;
            const float g = h1v[r] > 0.f ? acc[r] : 0.f;
            dh1[row * (kH1 + 1) + 32 * wave + c] = g;
            if (m < dm.B) dH1[(int64_t)m * kH1 + 32 * wave + c] = g;
            colsum += g;
        }
        colsum += __shfl_xor(colsum, 32, 64);
        if (s == 0) prec[pl.db1 + 32 * wave + c] = colsum;
    }
    lds_barrier();
    DT_STAMP(stamps, 2);

    // dXn = dH1 . W1^T : column blocks of 32 (CP/32 of them), round-robin over the 4 waves
    const float* arow = dh1 + c * (kH1 + 1) + s;
    float aq[kH1 / 2];
#pragma unroll
    for (int st = 0; st < kH1 / 2; ++st) aq[st] = arow[2 * st];   // A operand is the same for every column block
    // operands of column block nb: 64 W1T values + this lane's 16 X values; the next block's operands are
    // issued before the current block's MFMA chain so HBM/L2 latency hides under 4096 MFMA cycles
    float dzr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dzr[r] = dzs[(r & 3) + 8 * (r >> 2) + 4 * s];
    auto run_nb = [&](const float (&bq)[kH1 / 2], const float (&xv)[16], const float (&mr)[2], int nb) {
        const int col = 32 * nb + c;
        floatx16 accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < kH1 / 2; st += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st], bq[st], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st + 1], bq[st + 1], accb, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += accb[r];
        float sg = 0.f, sgx = 0.f, sl = 0.f;
        const float mu = mr[0], rs = mr[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
            const int m = m0 + row;
            const float g = acc[r];                 // pad columns: W1T column is zero -> g = 0
            if (m < dm.B) {
                dXn[(int64_t)m * dm.CP + col] = g;
                sg += g;
                sgx += g * ((xv[r] - mu) * rs);
                sl += dzr[r] * xv[r];
            }
        }
        sg += __shfl_xor(sg, 32, 64);
        sgx += __shfl_xor(sgx, 32, 64);
        sl += __shfl_xor(sl, 32, 64);
        if (s == 0) {
            prec[pl.sg + col] = sg;
            prec[pl.sgx + col] = sgx;
            prec[pl.slin + col] = sl;
        }
    };
    int nb = wave;
    while (nb < nblocks) {
        if (nb + 4 < nblocks) load_nb(bq1, xv1, mr1, nb + 4);
        __builtin_amdgcn_sched_barrier(0);
        run_nb(bq0, xv0, mr0, nb);
        __builtin_amdgcn_sched_barrier(0);
        nb += 4;
        if (nb >= nblocks) break;
        if (nb + 4 < nblocks) load_nb(bq0, xv0, mr0, nb + 4);
        __builtin_amdgcn_sched_barrier(0);
        run_nb(bq1, xv1, mr1, nb);
        __builtin_amdgcn_sched_barrier(0);
        nb += 4;
    }
    DT_STAMP(stamps, 3);
}

// ---------------------------------------------------------------------------------------------
// E: weight gradients + reduction of the per-tile partial sums.
//   blockIdx.x < ntiles: tile t < T1: dW1[32 cb.., 32 kb..] = sum_m Xn[m][c] dH1[m][k]
//                        else        : dW2[32 kb.., 32 nb..] = sum_m H1[m][k] dH2[m][n]
//     block = (tile, row split blockIdx.y); 4 waves take quarters of the split's rows, reduce via LDS.
//   blockIdx.x >= ntiles (blockIdx.y == 0 only): one wave per reduced output element.
// ---------------------------------------------------------------------------------------------
// operand loads for one chunk of kCH MFMA steps: rows base + 2i (+ s folded into the lane offset).  The row
// base is wave-uniform (SGPR) and the lane part constant, so every load is saddr + voffset with no 64-bit VALU.
template <bool GUARD>
__device__ __forceinline__ void wg_load(float (&aq)[kCH], float (&bq)[kCH], const float* pa, int sa, int offa,
                                        const float* pb, int sb, int offb, int base, int s, int r_end) {
#pragma unroll
    for (int i = 0; i < kCH; ++i) {
        const int row = base + 2 * i;                       // uniform
        const float* ra = pa + (int64_t)row * sa;           // uniform pointer
        const float* rb = pb + (int64_t)row * sb;
        if (GUARD) {
            const bool ok = row + s < r_end;
            aq[i] = ok ? ra[offa] : 0.f;
            bq[i] = ok ? rb[offb] : 0.f;
        } else {
            aq[i] = ra[offa];
            bq[i] = rb[offb];
        }
    }
}

__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ X, MlpParams p, DeepFmDims dm,
                                               const float* __restrict__ H1, const float* __restrict__ dH1,
                                               const float* __restrict__ dH2, int row_splits, int ntiles,
                                               const float* __restrict__ part, int nparts,
                                               float* __restrict__ accum, DeepFmAccum al) {
    __shared__ float red[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PartLayout pl = part_layout(dm.CP);
    const int nheavy = ntiles * row_splits;     // 1-D grid: heavy (tile, split) blocks first, then reducers
    if ((int)blockIdx.x >= nheavy) {
        // ---- reduction of the per-tile partials: element id -> destination in accum ----
        const int nsimple = 3 * dm.CP + kH1 + 2 * kH2 + 3;          // sg, sgx, slin, db1, db2, dw3, dwo, dbo, loss
        const int e = ((int)blockIdx.x - nheavy) * 4 + wave;
        if (e < nsimple) {
            int src; int64_t dst;
            if (e < dm.CP) { src = pl.sg + e; dst = al.dbeta + e; }
            else if (e < 2 * dm.CP) { src = pl.sgx + (e - dm.CP); dst = al.dgamma + (e - dm.CP); }
            else if (e < 3 * dm.CP) { src = pl.slin + (e - 2 * dm.CP); dst = al.slin + (e - 2 * dm.CP); }
            else {
                const int q = e - 3 * dm.CP;
                if (q < kH1) { src = pl.db1 + q; dst = al.db1 + q; }
                else if (q < kH1 + kH2) { src = pl.db2 + (q - kH1); dst = al.db2 + (q - kH1); }
                else if (q < kH1 + 2 * kH2) { src = pl.dw3 + (q - kH1 - kH2); dst = al.dw3 + (q - kH1 - kH2); }
                else if (q == kH1 + 2 * kH2) { src = pl.dwo; dst = al.dwo; }
                else if (q == kH1 + 2 * kH2 + 1) { src = pl.dbo; dst = al.dbo; }
                else { src = pl.loss; dst = al.loss; }
            }
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;   // 4 independent loads in flight per lane
            int t = lane;
            for (; t + 192 < nparts; t += 256) {
                v0 += part[(int64_t)t * pl.stride + src];
                v1 += part[(int64_t)(t + 64) * pl.stride + src];
                v2 += part[(int64_t)(t + 128) * pl.stride + src];
                v3 += part[(int64_t)(t + 192) * pl.stride + src];
            }
            for (; t < nparts; t += 64) v0 += part[(int64_t)t * pl.stride + src];
            const float v = wave_sum((v0 + v1) + (v2 + v3));
            if (lane == 0) accum[dst] = v;
        }
        return;
    }
    const int s = lane >> 5, c = lane & 31;
    const int cblocks = (dm.C + 31) >> 5;
    const int T1 = cblocks * (kH1 / 32);
    // XCD-aware (tile, split) assignment: workgroups are dealt round-robin to the 8 XCDs (id % 8) and every XCD
    // has its own 4 MB L2.  All tiles of one batch split read the same rows of X / dH1, so a split's tiles are
    // given ids with the same id % 8: each L2 then holds 1/8 of X instead of thrashing through all of it.
    int tile, split;
    {
        const int hid = blockIdx.x;                                // 0 .. ntiles*row_splits-1
        if ((row_splits & 7) == 0) {
            const int xcd = hid & 7, j = hid >> 3;
            const int per = row_splits >> 3;                       // splits per XCD
            tile = j % ntiles;
            split = xcd * per + j / ntiles;
        } else {
            tile = hid % ntiles;
            split = hid / ntiles;
        }
    }
    const int rows_per_split = ((dm.B + row_splits - 1) / row_splits + 7) & ~7;
    const int rq = rows_per_split >> 2;  // rows per wave (even)
    const int r_begin = split * rows_per_split + wave * rq;
    const int r_end = min(dm.B, r_begin + rq);

    floatx16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    const bool first = tile < T1;
    const float* pa; const float* pb; int sa, sb, offa, offb;
    float mu = 0.f, sc = 1.f, be = 0.f;
    if (first) {
        const int cb = tile / (kH1 / 32), kb = tile % (kH1 / 32);
        const int col = 32 * cb + c;           // < CP always (CP is a multiple of 64 >= C)
        mu = p.mean[col]; sc = p.sc[col]; be = p.betap[col];
        pa = X; sa = dm.CP; offa = s * dm.CP + col;
        pb = dH1; sb = kH1; offb = s * kH1 + 32 * kb + c;
    } else {
        const int t2 = tile - T1;
        const int kb = t2 / (kH2 / 32), nb = t2 % (kH2 / 32);
        pa = H1; sa = kH1; offa = s * kH1 + 32 * kb + c;
        pb = dH2; sb = kH2; offb = s * kH2 + 32 * nb + c;
    }
    {
        float a0[kCH], b0[kCH], a1[kCH], b1[kCH], a2[kCH], b2[kCH];
        const int span = 2 * kCH;                    // rows per chunk
        const int nfull = (r_end - r_begin) / span;  // unguarded chunks
        auto load = [&](float (&aq)[kCH], float (&bq)[kCH], int ch) {
            wg_load<false>(aq, bq, pa, sa, offa, pb, sb, offb, r_begin + ch * span, s, r_end);
        };
        auto run = [&](const float (&aq)[kCH], const float (&bq)[kCH]) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kCH; i += 2) {   // two accumulator chains (see k_mlp_fwd)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32((aq[i] - mu) * sc + be, bq[i], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((aq[i + 1] - mu) * sc + be, bq[i + 1], acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (nfull > 0) load(a0, b0, 0);
        if (nfull > 1) load(a1, b1, 1);
        for (int ch = 0; ch < nfull; ch += 3) {      // three operand buffers: see k_mlp_fwd
            if (ch + 2 < nfull) load(a2, b2, ch + 2);
            run(a0, b0);
            if (ch + 1 >= nfull) break;
            if (ch + 3 < nfull) load(a0, b0, ch + 3);
            run(a1, b1);
            if (ch + 2 >= nfull) break;
            if (ch + 4 < nfull) load(a1, b1, ch + 4);
            run(a2, b2);
        }
        for (int base = r_begin + nfull * span; base < r_end; base += span) {   // ragged tail
            wg_load<true>(a0, b0, pa, sa, offa, pb, sb, offb, base, s, r_end);
#pragma unroll
            for (int i = 0; i < kCH; ++i) {
                const bool ok = base + 2 * i + s < r_end;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? (a0[i] - mu) * sc + be : 0.f, b0[i], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * s][c] = acc[r] + acc2[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += blockDim.x) {
        const int i = e >> 5, j = e & 31;
        const float v = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
        if (first) {
            const int cb = tile / (kH1 / 32), kb = tile % (kH1 / 32);
            const int col = 32 * cb + i;
            if (col < dm.C) atomicAdd(accum + al.dW1 + (int64_t)col * kH1 + 32 * kb + j, v);
        } else {
            const int t2 = tile - T1;
            const int kb = t2 / (kH2 / 32), nb = t2 % (kH2 / 32);
            atomicAdd(accum + al.dW2 + (int64_t)(32 * kb + i) * kH2 + 32 * nb + j, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// G: sparse backward.  One wave per batch row, pure streaming (no atomics).
//   dX[c]  = gamma rstd (dXn - mean_b(dXn) - xhat mean_b(dXn xhat))
//   grad_rows[b,f,d] = dX[b,f*D+d] + dz[b] w_lin[f] + dz[b] (S[b,d] - E[b,f,d])
// ---------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void k_sparse_bwd(const float* __restrict__ X, const float* __restrict__ dXn,
                                                    const float* __restrict__ dz, MlpParams p,
                                                    const float* __restrict__ wlin, DeepFmDims dm,
                                                    const float* accum, DeepFmAccum al,
                                                    float* __restrict__ grad_rows, float* dwlin_out, DedupeWs dd,
                                                    float grad_scale, int field_major) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NV = dm.F * LPR;
    const int D = 4 * LPR;
    const float invN = 1.0f / (float)dm.B;
    if (blockIdx.x == 0) {   // d linear_logit kernel: field f = its D columns of the reduced slin, dense k = one column
        for (int q = threadIdx.x; q < dm.F + dm.Nd; q += blockDim.x) {
            float v = 0.f;
            if (q < dm.F) {
                for (int d = 0; d < D; ++d) v += accum[al.slin + q * D + d];
            } else {
                v = accum[al.slin + dm.F * D + (q - dm.F)];
            }
            dwlin_out[q] = v;
        }
    }
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= dm.B) return;
    const float g = dz[b];
    const float* xrow = X + (int64_t)b * dm.CP;
    const float* grow = dXn + (int64_t)b * dm.CP;
    float4 x[2], gx[2];
    float4 ca[2], cm1[2], cmu[2], cm2[2];
    float wl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        x[t] = gx[t] = ca[t] = cm1[t] = cmu[t] = cm2[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        wl[t] = 0.f;
        if (j < NV) {
            x[t] = *reinterpret_cast<const float4*>(xrow + 4 * j);
            gx[t] = *reinterpret_cast<const float4*>(grow + 4 * j);
            const float4 sc = *reinterpret_cast<const float4*>(p.sc + 4 * j);        // gamma * rstd
            const float4 r = *reinterpret_cast<const float4*>(p.rstd + 4 * j);
            const float4 sg = *reinterpret_cast<const float4*>(accum + al.dbeta + 4 * j);
            const float4 sgx = *reinterpret_cast<const float4*>(accum + al.dgamma + 4 * j);
            ca[t] = sc;
            cm1[t] = make_float4(sg.x * invN, sg.y * invN, sg.z * invN, sg.w * invN);
            cm2[t] = make_float4(r.x * sgx.x * invN, r.y * sgx.y * invN, r.z * sgx.z * invN, r.w * sgx.w * invN);
            cmu[t] = *reinterpret_cast<const float4*>(p.mean + 4 * j);
            wl[t] = wlin[j / LPR];
        }
    }
    float4 S = make_float4(x[0].x + x[1].x, x[0].y + x[1].y, x[0].z + x[1].z, x[0].w + x[1].w);
    S.x = wave_sum_strided<LPR>(S.x); S.y = wave_sum_strided<LPR>(S.y);
    S.z = wave_sum_strided<LPR>(S.z); S.w = wave_sum_strided<LPR>(S.w);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        if (j >= NV) continue;
        const float lin_g = g * wl[t];
        float4 o;
        o.x = ca[t].x * (gx[t].x - cm1[t].x - (x[t].x - cmu[t].x) * cm2[t].x) + lin_g + g * (S.x - x[t].x);
        o.y = ca[t].y * (gx[t].y - cm1[t].y - (x[t].y - cmu[t].y) * cm2[t].y) + lin_g + g * (S.y - x[t].y);
        o.z = ca[t].z * (gx[t].z - cm1[t].z - (x[t].z - cmu[t].z) * cm2[t].z) + lin_g + g * (S.z - x[t].z);
        o.w = ca[t].w * (gx[t].w - cm1[t].w - (x[t].w - cmu[t].w) * cm2[t].w) + lin_g + g * (S.w - x[t].w);
        const int f = j / LPR, c = j - f * LPR;
        if (field_major) {       // model-parallel tables: [F,B,D], already divided by the world size
            o.x *= grad_scale; o.y *= grad_scale; o.z *= grad_scale; o.w *= grad_scale;
            *reinterpret_cast<float4*>(grad_rows + ((int64_t)f * dm.B + b) * D + 4 * c) = o;
            continue;
        }
        if (!dd.slots) {
            *reinterpret_cast<float4*>(grad_rows + (int64_t)b * dm.F * D + 4 * j) = o;
            continue;
        }
        const int64_t occ = (int64_t)b * dm.F + f;
        const int mk = dd.mark[occ];
        int64_t target = occ;
        bool atomic = false;
        if (mk >= 0) {                       // owner
            atomic = dd.flags[occ] != 0;
            if (c == 0) {
                dd.slots[mk] = 0ULL;
                if (atomic) dd.flags[occ] = 0;
            }
        } else if (mk <= -2) {               // duplicate: into the owner's (zero-started) row
            target = (int64_t)(-mk - 2);
            atomic = true;
        }
        float* dst = grad_rows + target * D + 4 * c;
        if (atomic) {
            atomicAdd(dst, o.x); atomicAdd(dst + 1, o.y); atomicAdd(dst + 2, o.z); atomicAdd(dst + 3, o.w);
        } else {
            *reinterpret_cast<float4*>(dst) = o;
        }
    }
}

}  // namespace dt

using namespace dt;

static bool deepfm_dims(int B, int F, int D, int Nd, DeepFmDims* dm, int* lpr) {
    if (B <= 0 || F <= 0 || D <= 0 || Nd < 0 || D % 4) return false;
    const int l = D / 4;
    if (l < 1 || l > 64 || (l & (l - 1))) return false;
    if (F * l > 128 || Nd > 64) return false;
    dm->B = B; dm->F = F; dm->D = D; dm->Nd = Nd;
    dm->C = F * D + Nd;
    dm->CP = (dm->C + 63) & ~63;
    if (dm->C > 544) return false;
    *lpr = l;
    return true;
}

extern "C" int dt_deepfm_supported(int B, int F, int D, int Nd, int H1, int H2) {
    DeepFmDims dm; int lpr;
    return (H1 == kH1 && H2 == kH2 && deepfm_dims(B, F, D, Nd, &dm, &lpr)) ? 1 : 0;
}

// workspace layout (floats)
struct DeepFmWs {
    int64_t X, dXn, H1, dH1, H2, dH2, lin, fm, z, dz, dlogit, mean, rstd, sc, betap, W1P, W1T, bnp, bn2, part, stamps, total;
};
static DeepFmWs deepfm_ws_layout(const DeepFmDims& dm) {
    DeepFmWs w;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 3) & ~(int64_t)3; return r; };
    const int blocksA = ceil_div(dm.B, kRowsPerBlockA);
    const int tiles = ceil_div(dm.B, kTM);
    const int64_t rows = (int64_t)tiles * kTM;
    w.X = take(rows * dm.CP);
    w.dXn = take(rows * dm.CP);
    w.H1 = take(rows * kH1);
    w.dH1 = take(rows * kH1);
    w.H2 = take(rows * kH2);
    w.dH2 = take(rows * kH2);
    w.lin = take(rows); w.fm = take(rows); w.z = take(rows); w.dz = take(rows); w.dlogit = take(rows);
    w.mean = take(dm.CP); w.rstd = take(dm.CP); w.sc = take(dm.CP); w.betap = take(dm.CP);
    w.W1P = take((int64_t)dm.CP * kH1);
    w.W1T = take((int64_t)kH1 * dm.CP);
    w.bnp = take((int64_t)blocksA * 3 * dm.C);
    w.bn2 = take((int64_t)kBnSlices * 3 * dm.C);
    w.part = take((int64_t)tiles * part_layout(dm.CP).stride);
    w.stamps = take((int64_t)2 * tiles * 8 * 2);   // u64 [2 kernels][tiles][8]
    w.total = o;
    return w;
}

extern "C" int64_t dt_deepfm_workspace_bytes(int B, int F, int D, int Nd) {
    DeepFmDims dm; int lpr;
    if (!deepfm_dims(B, F, D, Nd, &dm, &lpr)) return -1;
    return deepfm_ws_layout(dm).total * (int64_t)sizeof(float);
}

extern "C" int64_t dt_deepfm_stamps_offset_floats(int B, int F, int D, int Nd) {
    DeepFmDims dm; int lpr;
    if (!deepfm_dims(B, F, D, Nd, &dm, &lpr)) return -1;
    return deepfm_ws_layout(dm).stamps;
}

extern "C" int64_t dt_deepfm_accum_floats(int F, int D, int Nd) {
    const int C = F * D + Nd, CP = (C + 63) & ~63;
    return deepfm_accum_layout(C, CP, F, Nd).total;
}

// offsets (in floats) of each gradient inside the accumulator buffer, in the order:
// dW1, dW2, db1, db2, dw3, dwo, dbo, loss, dgamma, dbeta, dwlin
extern "C" int dt_deepfm_accum_offsets(int F, int D, int Nd, int64_t* out11) {
    const int C = F * D + Nd, CP = (C + 63) & ~63;
    const DeepFmAccum a = deepfm_accum_layout(C, CP, F, Nd);
    const int64_t v[11] = {a.dW1, a.dW2, a.db1, a.db2, a.dw3, a.dwo, a.dbo, a.loss, a.dgamma, a.dbeta, a.dwlin};
    for (int i = 0; i < 11; ++i) out11[i] = v[i];
    return DT_OK;
}

extern "C" int64_t dt_deepfm_dedupe_slots(int B, int F) {
    int64_t s = 1024;
    while (s < 8LL * B * F) s <<= 1;   // load <= 1/8 keeps the serialised probe chains short
    return s;
}

extern "C" int64_t dt_deepfm_dedupe_bytes(int B, int F) {
    return dt_deepfm_dedupe_slots(B, F) * 8 + 2LL * B * F * 4;
}

extern "C" int dt_deepfm_train_step(
    const void* idx, int idx_kind, const float* table, const int64_t* row_offset, const int32_t* vocab,
    const float* dense, const float* y, int B, int F, int D, int Nd,
    const float* w_lin, const float* bn_gamma, const float* bn_beta, float* bn_moving_mean,
    float* bn_moving_var, float bn_eps, float bn_momentum, const float* W1, const float* b1, const float* W2,
    const float* b2, const float* w3, const float* w_out, const float* b_out,
    float* logit_out, int64_t* rows_out, float* grad_rows, float* accum, void* workspace, int* oob_count,
    void* dedupe_ws, int64_t dedupe_slots, float grad_rows_scale, int grad_rows_field_major, int phases,
    void* stream) {
    DeepFmDims dm; int lpr;
    DT_UNSUPPORTED(!deepfm_dims(B, F, D, Nd, &dm, &lpr), "dt_deepfm_train_step: unsupported shape B=%d F=%d D=%d Nd=%d",
                   B, F, D, Nd);
    DT_REQUIRE(idx && table && row_offset && vocab && y && w_lin && bn_gamma && bn_beta && W1 && b1 && W2 && b2 &&
                   w3 && w_out && logit_out && rows_out && grad_rows && accum && workspace,
               "dt_deepfm_train_step: null pointer");
    DT_REQUIRE(Nd == 0 || dense, "dt_deepfm_train_step: dense is null");
    DT_REQUIRE(idx_kind == DT_IDX_F32 || idx_kind == DT_IDX_I32, "dt_deepfm_train_step: idx_kind %d", idx_kind);
    hipStream_t st = as_stream(stream);
    const DeepFmWs wl = deepfm_ws_layout(dm);
    const DeepFmAccum al = deepfm_accum_layout(dm.C, dm.CP, F, Nd);
    float* ws = reinterpret_cast<float*>(workspace);
    MlpParams mp{ws + wl.W1P, b1, W2, b2, w3, w_out, b_out, bn_gamma, ws + wl.mean, ws + wl.rstd, ws + wl.sc,
                 ws + wl.betap};
    const int blocksA = ceil_div(B, kRowsPerBlockA);
    const int tiles = ceil_div(B, kTM);
    DedupeWs dd{nullptr, 0, nullptr, nullptr};
    DT_REQUIRE(!(dedupe_ws && grad_rows_field_major), "dt_deepfm_train_step: dedupe and field-major row gradients "
                                                      "are mutually exclusive");
    if (dedupe_ws && phases >= 2) {          // forward-only calls never reach G, which empties the hash again
        int lg = 0;
        while ((1LL << lg) < dedupe_slots) ++lg;
        DT_REQUIRE((1LL << lg) == dedupe_slots && dedupe_slots >= 2LL * B * F && lg <= 31 &&
                       (int64_t)B * F < (1LL << 31),
                   "dt_deepfm_train_step: dedupe_slots=%lld must be a power of two >= 2*B*F", (long long)dedupe_slots);
        dd.slots = reinterpret_cast<unsigned long long*>(dedupe_ws);
        dd.slots_log2 = lg;
        dd.mark = reinterpret_cast<int*>(dd.slots + dedupe_slots);
        dd.flags = dd.mark + (int64_t)B * F;
    }
    static const bool stamps_on = getenv("DT_DEEPFM_STAMPS") != nullptr;   // phase timestamps (tools/phase_times.py)

    // A  (DT_A_DYNLDS: experiment knob — extra dynamic LDS caps residency to one 1024-thread block per CU)
    static const size_t a_dyn_lds = getenv("DT_A_DYNLDS") ? (size_t)atoi(getenv("DT_A_DYNLDS")) : 0;
#define DT_A(KIND, L)                                                                                        \
    hipLaunchKernelGGL((k_sparse_fwd<KIND, L>), dim3(blocksA), dim3(1024), a_dyn_lds, st, idx, (const float4*)table,  \
                       row_offset, vocab, dense, w_lin, dm, ws + wl.X, ws + wl.lin, ws + wl.fm, rows_out,    \
                       oob_count, ws + wl.bnp, dd, grad_rows)
#define DT_A_L(KIND)                                                                  \
    switch (lpr) {                                                                    \
        case 1: DT_A(KIND, 1); break; case 2: DT_A(KIND, 2); break;                   \
        case 4: DT_A(KIND, 4); break; case 8: DT_A(KIND, 8); break;                   \
        case 16: DT_A(KIND, 16); break; case 32: DT_A(KIND, 32); break;               \
        default: DT_A(KIND, 64); break;                                               \
    }
    if (idx_kind == DT_IDX_F32) { DT_A_L(DT_IDX_F32) } else { DT_A_L(DT_IDX_I32) }
#undef DT_A_L
#undef DT_A
    // B
    const int bn_blocks = ceil_div(dm.C, 64) * kBnSlices;
    PrepOut po{ws + wl.mean, ws + wl.rstd, ws + wl.sc, ws + wl.betap, ws + wl.W1P, ws + wl.W1T, ws + wl.bn2};
    hipLaunchKernelGGL(k_prep, dim3(bn_blocks + 56), dim3(1024), 0, st, ws + wl.bnp, blocksA, dm, bn_eps, bn_momentum,
                       bn_gamma, bn_beta, bn_moving_mean, bn_moving_var, W1, po, bn_blocks, accum + al.dW1,
                       (int)(al.db1 - al.dW1));
    hipLaunchKernelGGL(k_bn_final, dim3(ceil_div(dm.CP, 256)), dim3(256), 0, st, dm, bn_eps, bn_momentum, bn_gamma,
                       bn_beta, bn_moving_mean, bn_moving_var, po);
    // C
    const size_t ldsC = ((size_t)kTM * (dm.CP + 1) + kTM * (kH1 + 1) + 3 * kTM * (kH2 + 1)) * sizeof(float);
    hipFuncSetAttribute((const void*)k_mlp_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsC);
    hipLaunchKernelGGL(k_mlp_fwd, dim3(tiles), dim3(256), ldsC, st, ws + wl.X, mp, dm, ws + wl.lin, ws + wl.fm, y,
                       ws + wl.H1, ws + wl.H2, ws + wl.z, logit_out, ws + wl.dlogit, ws + wl.part,
                       stamps_on ? reinterpret_cast<unsigned long long*>(ws + wl.stamps) : nullptr);
    // D (forward-only calls still need the loss reduced: E runs with zero tiles below)
    const int ntiles_w = ((dm.C + 31) >> 5) * (kH1 / 32) + (kH1 / 32) * (kH2 / 32);
    const PartLayout pl = part_layout(dm.CP);
    const int nred = 3 * dm.CP + kH1 + 2 * kH2 + 3;
    const int red_blocks = ceil_div(nred, 4);
    (void)pl;
    if (phases >= 2) {
        const size_t ldsD = ((size_t)kH1 * (kH2 + 1) + kTM * (kH2 + 1) + kTM * (kH1 + 1) + kTM) * sizeof(float);
        hipFuncSetAttribute((const void*)k_mlp_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsD);
        hipLaunchKernelGGL(k_mlp_bwd, dim3(tiles), dim3(256), ldsD, st, ws + wl.X, mp, ws + wl.W1T, dm, ws + wl.H1,
                           ws + wl.H2, ws + wl.z, ws + wl.dlogit, ws + wl.dH1, ws + wl.dH2, ws + wl.dXn,
                           ws + wl.dz, ws + wl.part,
                           stamps_on ? reinterpret_cast<unsigned long long*>(ws + wl.stamps) + (int64_t)tiles * 8 : nullptr);
        // E
        // 2 blocks (8 waves) per CU: two waves per SIMD so one wave's operand waits hide under the other's MFMAs.
        // 512 blocks = exactly one residency round at that occupancy (250 registers per lane): 8 batch splits
        // measured 21.2 us against 23.0 (16 splits, two rounds), 24.9 (4) and 25.4 (32)
        int splits = 512 / ntiles_w;
        if (splits < 1) splits = 1;
        if (splits > 32) splits = 32;
        while (splits > 1 && (B + splits - 1) / splits < 128) splits >>= 1;
        static const int splits_env = getenv("DT_WGRAD_SPLITS") ? atoi(getenv("DT_WGRAD_SPLITS")) : 0;   // experiment knob
        if (splits_env > 0) splits = splits_env;
        hipLaunchKernelGGL(k_wgrad, dim3(ntiles_w * splits + red_blocks), dim3(256), 0, st, ws + wl.X, mp, dm,
                           ws + wl.H1, ws + wl.dH1, ws + wl.dH2, splits, ntiles_w, ws + wl.part, tiles, accum, al);
        // G
        const int gblocks = ceil_div(B, 4);
#define DT_G(L)                                                                                              \
    case L:                                                                                                  \
        hipLaunchKernelGGL((k_sparse_bwd<L>), dim3(gblocks), dim3(256), 0, st, ws + wl.X, ws + wl.dXn,       \
                           ws + wl.dz, mp, w_lin, dm, accum, al, grad_rows, accum + al.dwlin, dd, grad_rows_scale,            \
                           grad_rows_field_major);                                                           \
        break;
        switch (lpr) { DT_G(1) DT_G(2) DT_G(4) DT_G(8) DT_G(16) DT_G(32) DT_G(64) }
#undef DT_G
    } else {
        // forward only: reduce just the loss (the other partial slots are stale and ignored by the caller)
        hipLaunchKernelGGL(k_wgrad, dim3(red_blocks), dim3(256), 0, st, ws + wl.X, mp, dm, ws + wl.H1, ws + wl.dH1,
                           ws + wl.dH2, 1, 0, ws + wl.part, tiles, accum, al);
    }
    return launch_status("dt_deepfm_train_step");
}
