// embedding.hip — categorical embedding gather, FM second-order term and the linear field-sum,
// separately and fused.  HBM-bound: one wavefront owns one batch row.
//
// Reference op sequences replaced here (paths relative to the DeepTables checkout):
//   MultiColumnEmbedding.call   deeptables/models/layers.py:889-904
//   FM.call                     deeptables/models/layers.py:53-62
//   deepnets.linear (sum_D)     deeptables/models/deepnets.py:49-51
//   Flatten(Concatenate(emb)) + Concatenate([emb, dense])   deeptables/models/deepmodel.py:269-274,348-353
//
// Data layout: a batch row's F embedding vectors are [F*D] contiguous floats = NV = F*D/4
// float4.  Lane l of the wave owns float4 j = l, l+64, ...; with LPR = D/4 lanes per embedding
// vector (a power of two) a lane always sees the same d-chunk c = l % LPR, so the FM sums
// S[d] = sum_f x, Q[d] = sum_f x^2 accumulate in registers and finish with a strided
// wave-shuffle reduction; the per-field sum is a reduction inside aligned LPR-lane groups.
// One wave-load instruction therefore covers 64/LPR table rows of 16*LPR bytes each (64-byte
// rows at D=16) and the output stores are fully coalesced 1 KiB lines.
#include "common.h"

namespace dt {

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ------------------------------------------------------------------------------------------
// Fast path: D = 4*LPR, LPR in {1,2,4,...,64}.
// ------------------------------------------------------------------------------------------
template <int KIND, int LPR, bool GATHER>
__global__ __launch_bounds__(256) void k_row_fwd(
    const void* __restrict__ idx, const float4* __restrict__ src /* table or x */,
    const int64_t* __restrict__ row_offset, const int32_t* __restrict__ vocab,
    const float* __restrict__ dense, int B, int F, int Nd, float4* __restrict__ emb_out,
    float* __restrict__ concat_out, float* __restrict__ field_sum, float* __restrict__ fm_out,
    int64_t* __restrict__ rows_out, int* __restrict__ oob_count) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;  // wave-uniform
    const int c = lane & (LPR - 1);
    const int NV = F * LPR;
    const int D = 4 * LPR;
    const int trips = (NV + 63) >> 6;
    const int64_t cstride = (int64_t)F * D + Nd;

    float4 S = f4_zero(), Q = f4_zero();
    for (int t = 0; t < trips; ++t) {
        const int j = lane + (t << 6);
        const bool act = j < NV;
        const int f = j / LPR;
        float4 v = f4_zero();
        if (act) {
            if (GATHER) {
                const int id = load_id<KIND>(idx, (int64_t)b * F + f);
                const bool ok = (unsigned)id < (unsigned)vocab[f];
                const int64_t row = ok ? row_offset[f] + id : (int64_t)-1;
                if (ok) v = src[row * LPR + c];
                if (c == 0) {
                    if (rows_out) rows_out[(int64_t)b * F + f] = row;
                    if (!ok && oob_count) atomicAdd(oob_count, 1);
                }
            } else {
                v = src[(int64_t)b * NV + j];
            }
            if (emb_out) emb_out[(int64_t)b * NV + j] = v;
            if (concat_out) {  // row stride F*D+Nd floats is only 4-byte aligned
                float* dst = concat_out + (int64_t)b * cstride + (int64_t)j * 4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        }
        if (field_sum) {
            float s = (v.x + v.y) + (v.z + v.w);
            s = group_sum<LPR>(s);
            if (act && c == 0) field_sum[(int64_t)b * F + f] = s;
        }
        S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
        Q.x += v.x * v.x; Q.y += v.y * v.y; Q.z += v.z * v.z; Q.w += v.w * v.w;
    }
    if (concat_out && Nd > 0) {
        for (int k = lane; k < Nd; k += 64)
            concat_out[(int64_t)b * cstride + (int64_t)F * D + k] = dense[(int64_t)b * Nd + k];
    }
    if (fm_out) {
        S.x = wave_sum_strided<LPR>(S.x); S.y = wave_sum_strided<LPR>(S.y);
        S.z = wave_sum_strided<LPR>(S.z); S.w = wave_sum_strided<LPR>(S.w);
        Q.x = wave_sum_strided<LPR>(Q.x); Q.y = wave_sum_strided<LPR>(Q.y);
        Q.z = wave_sum_strided<LPR>(Q.z); Q.w = wave_sum_strided<LPR>(Q.w);
        float tsum = ((S.x * S.x - Q.x) + (S.y * S.y - Q.y)) + ((S.z * S.z - Q.z) + (S.w * S.w - Q.w));
        tsum = group_sum<LPR>(tsum);
        if (lane == 0) fm_out[b] = 0.5f * tsum;
    }
}

// grad_rows[b,f,d] = g_emb + g_concat[b,f*D+d] + g_field_sum[b,f] + g_fm[b]*(S[b,d]-emb[b,f,d])
template <int LPR>
__global__ __launch_bounds__(256) void k_row_bwd(
    const float4* __restrict__ emb, const float4* __restrict__ g_emb,
    const float* __restrict__ g_concat, int concat_stride, const float* __restrict__ g_field_sum,
    const float* __restrict__ g_fm, int B, int F, float4* __restrict__ grad_rows) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int NV = F * LPR;
    const int trips = (NV + 63) >> 6;
    float4 S = f4_zero();
    float gf = 0.f;
    if (g_fm) {
        gf = g_fm[b];
        for (int t = 0; t < trips; ++t) {
            const int j = lane + (t << 6);
            if (j < NV) {
                float4 v = emb[(int64_t)b * NV + j];
                S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
            }
        }
        S.x = wave_sum_strided<LPR>(S.x); S.y = wave_sum_strided<LPR>(S.y);
        S.z = wave_sum_strided<LPR>(S.z); S.w = wave_sum_strided<LPR>(S.w);
    }
    for (int t = 0; t < trips; ++t) {
        const int j = lane + (t << 6);
        if (j >= NV) continue;
        float4 g = f4_zero();
        if (g_emb) g = g_emb[(int64_t)b * NV + j];
        if (g_concat) {
            const float* s = g_concat + (int64_t)b * concat_stride + (int64_t)j * 4;
            g.x += s[0]; g.y += s[1]; g.z += s[2]; g.w += s[3];
        }
        if (g_field_sum) {
            const float gs = g_field_sum[(int64_t)b * F + j / LPR];
            g.x += gs; g.y += gs; g.z += gs; g.w += gs;
        }
        if (g_fm) {
            float4 v = emb[(int64_t)b * NV + j];
            g.x += gf * (S.x - v.x); g.y += gf * (S.y - v.y);
            g.z += gf * (S.z - v.z); g.w += gf * (S.w - v.w);
        }
        grad_rows[(int64_t)b * NV + j] = g;
    }
}

// ------------------------------------------------------------------------------------------
// Generic path (any D): one wave per batch row, the row staged through LDS.
// ------------------------------------------------------------------------------------------
template <int KIND, bool GATHER>
__global__ __launch_bounds__(64) void k_row_fwd_generic(
    const void* __restrict__ idx, const float* __restrict__ src,
    const int64_t* __restrict__ row_offset, const int32_t* __restrict__ vocab,
    const float* __restrict__ dense, int B, int F, int D, int Nd, float* __restrict__ emb_out,
    float* __restrict__ concat_out, float* __restrict__ field_sum, float* __restrict__ fm_out,
    int64_t* __restrict__ rows_out, int* __restrict__ oob_count) {
    extern __shared__ __attribute__((aligned(16))) float row_lds[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int N = F * D;
    const int64_t cstride = (int64_t)N + Nd;
    for (int e = lane; e < N; e += 64) {
        const int f = e / D, d = e - f * D;
        float v = 0.f;
        if (GATHER) {
            const int id = load_id<KIND>(idx, (int64_t)b * F + f);
            const bool ok = (unsigned)id < (unsigned)vocab[f];
            const int64_t row = ok ? row_offset[f] + id : (int64_t)-1;
            if (ok) v = src[row * D + d];
            if (d == 0) {
                if (rows_out) rows_out[(int64_t)b * F + f] = row;
                if (!ok && oob_count) atomicAdd(oob_count, 1);
            }
        } else {
            v = src[(int64_t)b * N + e];
        }
        row_lds[e] = v;
        if (emb_out) emb_out[(int64_t)b * N + e] = v;
        if (concat_out) concat_out[(int64_t)b * cstride + e] = v;
    }
    if (concat_out)
        for (int k = lane; k < Nd; k += 64)
            concat_out[(int64_t)b * cstride + N + k] = dense[(int64_t)b * Nd + k];
    __syncthreads();
    if (field_sum)
        for (int f = lane; f < F; f += 64) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) s += row_lds[f * D + d];
            field_sum[(int64_t)b * F + f] = s;
        }
    if (fm_out) {
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) {
            float s = 0.f, q = 0.f;
            for (int f = 0; f < F; ++f) {
                const float v = row_lds[f * D + d];
                s += v; q += v * v;
            }
            acc += s * s - q;
        }
        acc = wave_sum(acc);
        if (lane == 0) fm_out[b] = 0.5f * acc;
    }
}

__global__ __launch_bounds__(64) void k_row_bwd_generic(
    const float* __restrict__ emb, const float* __restrict__ g_emb,
    const float* __restrict__ g_concat, int concat_stride, const float* __restrict__ g_field_sum,
    const float* __restrict__ g_fm, int B, int F, int D, float* __restrict__ grad_rows) {
    extern __shared__ __attribute__((aligned(16))) float s_lds[];  // S[d]
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int N = F * D;
    float gf = 0.f;
    if (g_fm) {
        gf = g_fm[b];
        for (int d = lane; d < D; d += 64) {
            float s = 0.f;
            for (int f = 0; f < F; ++f) s += emb[(int64_t)b * N + f * D + d];
            s_lds[d] = s;
        }
        __syncthreads();
    }
    for (int e = lane; e < N; e += 64) {
        const int f = e / D, d = e - f * D;
        float g = 0.f;
        if (g_emb) g = g_emb[(int64_t)b * N + e];
        if (g_concat) g += g_concat[(int64_t)b * concat_stride + e];
        if (g_field_sum) g += g_field_sum[(int64_t)b * F + f];
        if (g_fm) g += gf * (s_lds[d] - emb[(int64_t)b * N + e]);
        grad_rows[(int64_t)b * N + e] = g;
    }
}

// plain gather, 16-byte lanes: thread -> (lookup, chunk).  Used when only the rows are wanted.
template <int KIND>
__global__ __launch_bounds__(256) void k_gather_vec4(
    const void* __restrict__ idx, const float4* __restrict__ table,
    const int64_t* __restrict__ row_offset, const int32_t* __restrict__ vocab, int64_t n_lookups,
    int F, int LPR, float4* __restrict__ out, int64_t* __restrict__ rows_out,
    int* __restrict__ oob_count) {
    const int64_t total = n_lookups * LPR;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / LPR;
        const int c = (int)(t - r * LPR);
        const int f = (int)(r % F);
        const int id = load_id<KIND>(idx, r);
        const bool ok = (unsigned)id < (unsigned)vocab[f];
        const int64_t row = ok ? row_offset[f] + id : (int64_t)-1;
        float4 v = f4_zero();
        if (ok) v = table[row * LPR + c];
        out[t] = v;
        if (c == 0) {
            if (rows_out) rows_out[r] = row;
            if (!ok && oob_count) atomicAdd(oob_count, 1);
        }
    }
}

// Owner-side gather of the model-parallel exchange (parallel.ShardedEmbeddingStrategy): the ids of ALL W minibatches
// [W,B,F] for the fields [f0,f1) this rank owns -> rows [W, F_own, B, D] (field-major inside each minibatch, so the
// piece for rank w is contiguous) + the packed row ids for the optimizer.  One launch instead of a slice / permute /
// range-check / offset chain of element-wise kernels followed by a gather.
template <int KIND>
__global__ __launch_bounds__(256) void k_gather_owned(const void* __restrict__ idx_all,
                                                      const float4* __restrict__ table,
                                                      const int64_t* __restrict__ row_offset,
                                                      const int32_t* __restrict__ vocab, int W, int B, int F, int f0,
                                                      int f1, int LPR, float4* __restrict__ out,
                                                      int64_t* __restrict__ rows_out, int* __restrict__ oob_count) {
    const int Fo = f1 - f0;
    const int64_t total = (int64_t)W * Fo * B * LPR;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / LPR;                  // output lookup index in [W, Fo, B] order
        const int c = (int)(t - r * LPR);
        const int b = (int)(r % B);
        const int fo = (int)((r / B) % Fo);
        const int w = (int)(r / ((int64_t)B * Fo));
        const int f = f0 + fo;
        const int id = load_id<KIND>(idx_all, ((int64_t)w * B + b) * F + f);
        const bool ok = (unsigned)id < (unsigned)vocab[f];
        const int64_t row = ok ? row_offset[f] + id : (int64_t)-1;
        float4 v = f4_zero();
        if (ok) v = table[row * LPR + c];
        out[t] = v;
        if (c == 0) {
            rows_out[r] = row;
            if (!ok && oob_count) atomicAdd(oob_count, 1);
        }
    }
}

__global__ __launch_bounds__(256) void k_scatter_add_rows(const int64_t* __restrict__ rows,
                                                         const float* __restrict__ g,
                                                         int64_t total, int D,
                                                         float* __restrict__ grad_table) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / D;
        const int d = (int)(t - r * D);
        const int64_t row = rows[r];
        if (row >= 0) atomicAdd(grad_table + row * D + d, g[t]);
    }
}

// ---- dispatch helpers ------------------------------------------------------------------------
static bool fast_lpr(int D, int* lpr) {
    if (D % 4) return false;
    const int l = D / 4;
    if (l < 1 || l > 64 || (l & (l - 1))) return false;
    *lpr = l;
    return true;
}

template <int KIND, bool GATHER>
static int launch_row_fwd(const void* idx, const float* src, const int64_t* row_offset,
                          const int32_t* vocab, const float* dense, int B, int F, int D, int Nd,
                          float* emb_out, float* concat_out, float* field_sum, float* fm_out,
                          int64_t* rows_out, int* oob, hipStream_t st) {
    int lpr = 0;
    if (fast_lpr(D, &lpr)) {
        dim3 grid(ceil_div(B, 4)), block(256);
#define DT_ROW_FWD(L)                                                                          \
    case L:                                                                                    \
        hipLaunchKernelGGL((k_row_fwd<KIND, L, GATHER>), grid, block, 0, st, idx,              \
                           (const float4*)src, row_offset, vocab, dense, B, F, Nd,             \
                           (float4*)emb_out, concat_out, field_sum, fm_out, rows_out, oob);    \
        break;
        switch (lpr) {
            DT_ROW_FWD(1) DT_ROW_FWD(2) DT_ROW_FWD(4) DT_ROW_FWD(8) DT_ROW_FWD(16)
            DT_ROW_FWD(32) DT_ROW_FWD(64)
        }
#undef DT_ROW_FWD
    } else {
        const size_t lds = (size_t)F * D * sizeof(float);
        if (lds > 64 * 1024) {
            set_error("row kernel: F*D=%d floats exceeds the 64 KiB LDS row buffer", F * D);
            return DT_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL((k_row_fwd_generic<KIND, GATHER>), dim3(B), dim3(64), lds, st, idx, src,
                           row_offset, vocab, dense, B, F, D, Nd, emb_out, concat_out, field_sum,
                           fm_out, rows_out, oob);
    }
    return launch_status("row_fwd");
}

static int launch_row_bwd(const float* emb, const float* g_emb, const float* g_concat,
                          int concat_stride, const float* g_field_sum, const float* g_fm, int B,
                          int F, int D, float* grad_rows, hipStream_t st) {
    int lpr = 0;
    if (fast_lpr(D, &lpr)) {
        dim3 grid(ceil_div(B, 4)), block(256);
#define DT_ROW_BWD(L)                                                                          \
    case L:                                                                                    \
        hipLaunchKernelGGL((k_row_bwd<L>), grid, block, 0, st, (const float4*)emb,             \
                           (const float4*)g_emb, g_concat, concat_stride, g_field_sum, g_fm, B, \
                           F, (float4*)grad_rows);                                             \
        break;
        switch (lpr) {
            DT_ROW_BWD(1) DT_ROW_BWD(2) DT_ROW_BWD(4) DT_ROW_BWD(8) DT_ROW_BWD(16)
            DT_ROW_BWD(32) DT_ROW_BWD(64)
        }
#undef DT_ROW_BWD
    } else {
        hipLaunchKernelGGL(k_row_bwd_generic, dim3(B), dim3(64), (size_t)D * sizeof(float), st, emb,
                           g_emb, g_concat, concat_stride, g_field_sum, g_fm, B, F, D, grad_rows);
    }
    return launch_status("row_bwd");
}

}  // namespace dt

using namespace dt;

extern "C" int dt_embedding_fwd(const void* idx, int idx_kind, const float* table,
                                const int64_t* row_offset, const int32_t* vocab, int B, int F,
                                int D, float* out, int64_t* rows_out, int* oob_count,
                                void* stream) {
    DT_REQUIRE(B >= 0 && F >= 0 && D > 0, "dt_embedding_fwd: bad sizes B=%d F=%d D=%d", B, F, D);
    DT_REQUIRE(idx_kind == DT_IDX_F32 || idx_kind == DT_IDX_I32, "dt_embedding_fwd: idx_kind %d",
               idx_kind);
    if (B == 0 || F == 0) return DT_OK;
    DT_REQUIRE(idx && table && row_offset && vocab && out, "dt_embedding_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (D % 4 == 0) {
        const int lpr = D / 4;
        const int64_t total = (int64_t)B * F * lpr;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 256 * 16) blocks = 256 * 16;
        if (idx_kind == DT_IDX_F32)
            hipLaunchKernelGGL((k_gather_vec4<DT_IDX_F32>), dim3(blocks), dim3(256), 0, st, idx,
                               (const float4*)table, row_offset, vocab, (int64_t)B * F, F, lpr,
                               (float4*)out, rows_out, oob_count);
        else
            hipLaunchKernelGGL((k_gather_vec4<DT_IDX_I32>), dim3(blocks), dim3(256), 0, st, idx,
                               (const float4*)table, row_offset, vocab, (int64_t)B * F, F, lpr,
                               (float4*)out, rows_out, oob_count);
        return launch_status("dt_embedding_fwd");
    }
    if (idx_kind == DT_IDX_F32)
        return launch_row_fwd<DT_IDX_F32, true>(idx, table, row_offset, vocab, nullptr, B, F, D, 0,
                                                out, nullptr, nullptr, nullptr, rows_out, oob_count,
                                                st);
    return launch_row_fwd<DT_IDX_I32, true>(idx, table, row_offset, vocab, nullptr, B, F, D, 0, out,
                                            nullptr, nullptr, nullptr, rows_out, oob_count, st);
}

extern "C" int dt_embedding_gather_owned(const void* idx_all, int idx_kind, const float* table,
                                         const int64_t* row_offset, const int32_t* vocab, int W, int B, int F,
                                         int f_begin, int f_end, int D, float* out, int64_t* rows_out,
                                         int* oob_count, void* stream) {
    DT_REQUIRE(W > 0 && B >= 0 && F > 0 && D > 0 && D % 4 == 0 && 0 <= f_begin && f_begin <= f_end && f_end <= F,
               "dt_embedding_gather_owned: bad sizes W=%d B=%d F=%d fields [%d,%d) D=%d", W, B, F, f_begin, f_end, D);
    DT_REQUIRE(idx_kind == DT_IDX_F32 || idx_kind == DT_IDX_I32, "dt_embedding_gather_owned: idx_kind %d", idx_kind);
    const int64_t total = (int64_t)W * (f_end - f_begin) * B * (D / 4);
    if (total == 0) return DT_OK;
    DT_REQUIRE(idx_all && table && row_offset && vocab && out && rows_out, "dt_embedding_gather_owned: null pointer");
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (idx_kind == DT_IDX_F32)
        hipLaunchKernelGGL((k_gather_owned<DT_IDX_F32>), dim3(blocks), dim3(256), 0, as_stream(stream), idx_all,
                           (const float4*)table, row_offset, vocab, W, B, F, f_begin, f_end, D / 4, (float4*)out,
                           rows_out, oob_count);
    else
        hipLaunchKernelGGL((k_gather_owned<DT_IDX_I32>), dim3(blocks), dim3(256), 0, as_stream(stream), idx_all,
                           (const float4*)table, row_offset, vocab, W, B, F, f_begin, f_end, D / 4, (float4*)out,
                           rows_out, oob_count);
    return launch_status("dt_embedding_gather_owned");
}

extern "C" int dt_embedding_bwd_dense(const int64_t* rows, const float* grad_out, int n_lookups,
                                      int D, float* grad_table, void* stream) {
    DT_REQUIRE(n_lookups >= 0 && D > 0, "dt_embedding_bwd_dense: bad sizes");
    if (n_lookups == 0) return DT_OK;
    DT_REQUIRE(rows && grad_out && grad_table, "dt_embedding_bwd_dense: null pointer");
    const int64_t total = (int64_t)n_lookups * D;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_scatter_add_rows, dim3(blocks), dim3(256), 0, as_stream(stream), rows,
                       grad_out, total, D, grad_table);
    return launch_status("dt_embedding_bwd_dense");
}

extern "C" int dt_fm_fwd(const float* x, int B, int F, int D, float* out, void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0, "dt_fm_fwd: bad sizes B=%d F=%d D=%d", B, F, D);
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && out, "dt_fm_fwd: null pointer");
    return launch_row_fwd<DT_IDX_I32, false>(nullptr, x, nullptr, nullptr, nullptr, B, F, D, 0,
                                             nullptr, nullptr, nullptr, out, nullptr, nullptr,
                                             as_stream(stream));
}

extern "C" int dt_fm_bwd(const float* x, const float* grad_out, int B, int F, int D,
                         float* grad_x, void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0, "dt_fm_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(x && grad_out && grad_x, "dt_fm_bwd: null pointer");
    return launch_row_bwd(x, nullptr, nullptr, 0, nullptr, grad_out, B, F, D, grad_x,
                          as_stream(stream));
}

extern "C" int dt_embed_fm_linear_fwd(const void* idx, int idx_kind, const float* table,
                                      const int64_t* row_offset, const int32_t* vocab,
                                      const float* dense, int B, int F, int D, int Nd,
                                      float* emb_out, float* concat_out, float* field_sum,
                                      float* fm_out, int64_t* rows_out, int* oob_count,
                                      void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0 && Nd >= 0, "dt_embed_fm_linear_fwd: bad sizes");
    DT_REQUIRE(idx_kind == DT_IDX_F32 || idx_kind == DT_IDX_I32,
               "dt_embed_fm_linear_fwd: idx_kind %d", idx_kind);
    if (B == 0) return DT_OK;
    DT_REQUIRE(idx && table && row_offset && vocab, "dt_embed_fm_linear_fwd: null pointer");
    DT_REQUIRE(!(concat_out && Nd > 0 && !dense), "dt_embed_fm_linear_fwd: dense is null, Nd=%d",
               Nd);
    hipStream_t st = as_stream(stream);
    if (idx_kind == DT_IDX_F32)
        return launch_row_fwd<DT_IDX_F32, true>(idx, table, row_offset, vocab, dense, B, F, D, Nd,
                                                emb_out, concat_out, field_sum, fm_out, rows_out,
                                                oob_count, st);
    return launch_row_fwd<DT_IDX_I32, true>(idx, table, row_offset, vocab, dense, B, F, D, Nd,
                                            emb_out, concat_out, field_sum, fm_out, rows_out,
                                            oob_count, st);
}

extern "C" int dt_embed_fm_linear_bwd(const float* emb, const float* g_emb, const float* g_concat,
                                      int concat_stride, const float* g_field_sum,
                                      const float* g_fm, int B, int F, int D, float* grad_rows,
                                      void* stream) {
    DT_REQUIRE(B >= 0 && F > 0 && D > 0, "dt_embed_fm_linear_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(grad_rows, "dt_embed_fm_linear_bwd: null grad_rows");
    DT_REQUIRE(!(g_fm && !emb), "dt_embed_fm_linear_bwd: g_fm needs emb");
    DT_REQUIRE(!(g_concat && concat_stride < F * D), "dt_embed_fm_linear_bwd: concat_stride %d",
               concat_stride);
    return launch_row_bwd(emb, g_emb, g_concat, concat_stride, g_field_sum, g_fm, B, F, D,
                          grad_rows, as_stream(stream));
}

// ---------------------------------------------------------------------------------------------
// dt_feed_gather: batch assembly on the device (include/dt_hip.h).  One thread per dword of a destination row set; the
// dwords of a row are consecutive threads, so every block row is a contiguous segment on both sides.
// ---------------------------------------------------------------------------------------------
namespace dt {
struct FeedBlocks {
    const uint32_t* src[8];
    uint32_t* dst[8];
    int first[9];            // prefix sums of the blocks' dwords per row
    int n;
};
// (round 4, first version: a one-thread launch advanced the cursor behind the gather — 4.7 us for one store; now the last
// block of the gather to FINISH does it: every block has read the cursor by then)
// one wave per destination row: lane q copies dword q of the row's blocks (q += 64 for wider rows); the row index is one
// scalar load per wave, no division per element (the first version — a thread per dword with a 64-bit division and a
// dependent index load each — took 17 us for the 13 MB of ten 8192-row steps)
constexpr int kFeedGroups = 32;   // ticket groups of the cursor (DT_FEED_CURSOR_WORDS = 16 (1 + groups) int64 words)
__global__ __launch_bounds__(256) void k_feed_gather(const int64_t* __restrict__ sel, int64_t n_rows, FeedBlocks fb,
                                                     int64_t* __restrict__ cursor) {
    const int64_t base = cursor ? *cursor : 0;
    sel += base;
    const int per = fb.first[fb.n];
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int R = 8;                          // rows per wave and pass: R index loads, then R source loads in flight
    for (int64_t r0 = wave * R; r0 < n_rows; r0 += nwaves * R) {
        int64_t srow[R];
#pragma unroll
        for (int u = 0; u < R; ++u) srow[u] = sel[min(r0 + u, n_rows - 1)];
        for (int q = lane; q < per; q += 64) {
            int b = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) b += (k < fb.n && q >= fb.first[k]) ? 1 : 0;
            const int w = fb.first[b + 1] - fb.first[b], c = q - fb.first[b];
            uint32_t v[R];
#pragma unroll
            for (int u = 0; u < R; ++u) v[u] = fb.src[b][srow[u] * w + c];
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (r0 + u < n_rows) fb.dst[b][(r0 + u) * w + c] = v[u];
        }
    }
    if (cursor) {
        // Arrival tickets (all zero between launches), two levels — 2560 blocks adding to ONE word serialize in the L2's
        // atomic unit (7 ns each: the gather took 31.6 us instead of 12.8): block b adds to the ticket of group b % 32 (its
        // own 128-byte line), a group's last arrival adds to the top ticket, the top's last arrival moves the position.
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned G = gridDim.x < (unsigned)kFeedGroups ? gridDim.x : (unsigned)kFeedGroups;
            const unsigned g = blockIdx.x % G, members = (gridDim.x - g + G - 1) / G;
            unsigned* mine = reinterpret_cast<unsigned*>(cursor + 16 * (1 + g));
            if (atomicAdd(mine, 1u) == members - 1) {
                *mine = 0u;
                unsigned* top = reinterpret_cast<unsigned*>(cursor + 1);
                if (atomicAdd(top, 1u) == G - 1) {
                    *top = 0u;
                    *cursor = base + n_rows;
                }
            }
        }
    }
}
}  // namespace dt

extern "C" int dt_feed_gather(const int64_t* sel, int64_t n_rows, int n_blocks, const void* const* src, void* const* dst,
                              const int* row_bytes, int64_t* cursor, void* stream) {
    DT_REQUIRE(n_rows >= 0 && n_blocks >= 1 && n_blocks <= 8 && src && dst && row_bytes, "dt_feed_gather: bad arguments");
    if (n_rows == 0) return DT_OK;
    DT_REQUIRE(sel, "dt_feed_gather: null pointer");
    dt::FeedBlocks fb;
    fb.n = n_blocks;
    fb.first[0] = 0;
    for (int b = 0; b < 8; ++b) {
        fb.src[b] = nullptr; fb.dst[b] = nullptr;
        if (b < n_blocks) {
            DT_REQUIRE(src[b] && dst[b] && row_bytes[b] > 0 && row_bytes[b] % 4 == 0, "dt_feed_gather: block %d: null pointer or a "
                       "row size that is not a positive multiple of 4 bytes (%d)", b, row_bytes[b]);
            fb.src[b] = reinterpret_cast<const uint32_t*>(src[b]);
            fb.dst[b] = reinterpret_cast<uint32_t*>(dst[b]);
            fb.first[b + 1] = fb.first[b] + row_bytes[b] / 4;
        } else {
            fb.first[b + 1] = fb.first[b];
        }
    }
    int64_t blocks = (n_rows + 31) / 32;               // four waves of eight rows per block
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(dt::k_feed_gather, dim3((unsigned)blocks), dim3(256), 0, dt::as_stream(stream), sel, n_rows, fb, cursor);
    return dt::launch_status("dt_feed_gather");
}
