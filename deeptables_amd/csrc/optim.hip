// optim.hip — Keras-semantics Adam steps.
//
// Replaces keras.optimizers.Adam(learning_rate=0.001) selected by
// DeepModel.__compile_model, deeptables/models/deepmodel.py:321-322.  Keras formulation:
//     m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= lr_t * m / (sqrt(v) + eps)
//     lr_t = lr * sqrt(1-b2^t)/(1-b1^t)   (computed by the host, eps defaults to 1e-7)
// (torch.optim.Adam puts eps after the bias correction of v and defaults to 1e-8 — different.)
//
//  * dt_adam_dense_step : every element, used for dense layers and for exact (dense) Adam on
//    small embedding tables.
//  * dt_adam_rows_step  : "lazy" row-sparse variant for 1M-row tables: only rows looked up in
//    this step are touched (m/v of other rows do not decay — a documented deviation from
//    Keras' dense semantics).  Duplicate lookups of a row are merged through a small
//    hash of the step's row ids (see below); each distinct row is updated exactly once.
#include <stdlib.h>
#include "common.h"
#include "adam_dev.h"

namespace dt {

const float* adam_state_lr_t(const void* state) { return &reinterpret_cast<const AdamState*>(state)->lr_t; }

__global__ void k_adam_state_init(AdamState* st, float lr, float b1, float b2, int steps_done) {
    st->t = steps_done + 1;
    st->lr_t = adam_lr_t(lr, b1, b2, steps_done + 1);
    st->done = 0u;
    st->pad = 0;
    for (int k = 0; k < kAdamSub; ++k) st->sub[k] = 0u;
}

__global__ void k_adam_advance(AdamState* st, float lr, float b1, float b2) {
    const int t = st->t + 1;
    st->t = t;
    st->lr_t = adam_lr_t(lr, b1, b2, t);
}

__global__ __launch_bounds__(256) void k_adam_dense(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    float lr_host, AdamState* __restrict__ st, float b1, float b2,
                                                    float eps, int advance, float lr) {
    unsigned ticket;
    const float lr_t = adam_read_lr(st, lr_host, advance, ticket);
    const DenseTail d{p, g, m, v, n};
    adam_dense_range(d, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, lr_t, b1, b2,
                     eps);
    adam_finish(st, ticket, lr, b1, b2);
}

// Several parameter tensors in ONE launch (the layer-by-layer path has one small tensor per Dense / BN / Cross
// variable: ~4.5 us of launch each).  The tensor descriptors travel as kernel arguments (no device-side table to keep
// in sync, replayable from a hipGraph); a block finds its tensor by scanning the block offsets.
constexpr int kMultiMax = 32;
constexpr int kMultiPerBlock = 1024;          // elements per block (256 threads x 4)
struct AdamMulti {
    float* p[kMultiMax];
    const float* g[kMultiMax];
    float* m[kMultiMax];
    float* v[kMultiMax];
    int n[kMultiMax];
    int block_start[kMultiMax + 1];
    int count;
};

__global__ __launch_bounds__(256) void k_adam_multi(AdamMulti d, float lr_host, AdamState* __restrict__ st, float b1,
                                                    float b2, float eps, int advance, float lr) {
    unsigned ticket;
    const float lr_t = adam_read_lr(st, lr_host, advance, ticket);
    int t = 0;
    while (t + 1 < d.count && (int)blockIdx.x >= d.block_start[t + 1]) ++t;
    const DenseTail tail{d.p[t], d.g[t], d.m[t], d.v[t], (int64_t)d.n[t]};
    const int64_t first = (int64_t)(blockIdx.x - d.block_start[t]) * kMultiPerBlock + threadIdx.x;
    const int64_t end = min((int64_t)d.n[t], (int64_t)(blockIdx.x - d.block_start[t] + 1) * kMultiPerBlock);
    for (int64_t i = first; i < end; i += 256) {
        const float gi = tail.g[i];
        const float mi = b1 * tail.m[i] + (1.f - b1) * gi;
        const float vi = b2 * tail.v[i] + (1.f - b2) * gi * gi;
        tail.m[i] = mi;
        tail.v[i] = vi;
        tail.p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
    adam_finish(st, ticket, lr, b1, b2);
}

// ---- row-sparse ("lazy") Adam on (rows, values) pairs --------------------------------------------------------
// Pass 1 (k_rows_dedupe): one thread per looked-up row occurrence inserts its table row into a small open-addressing
// hash (64-bit slots: (row+1) << 32 | occurrence).  The first occurrence of a row owns it; later duplicates add
// their gradient row into the owner's (atomics on an L2/MALL-resident [n,D] buffer — duplicates are rare) and mark
// themselves done.  Pass 2 (k_adam_rows_owner): D/4 lanes per owner read the merged gradient (coalesced), update
// p/m/v of that one table row (16-byte pieces), and the owner clears its hash slot so the table is empty again for
// the next step — no dense gradient scratch table, no per-row epoch array, no memset.

// dst[0..D) += src[0..D) with atomics.  dst and src live in the same buffer, so the compiler must keep every load
// behind the previous atomic; loading a chunk into registers first keeps the loads independent (one latency, not D).
template <int CH>
__device__ __forceinline__ int merge_chunks(float* dst, const float* src, int d, int D) {
    for (; d + CH <= D; d += CH) {
        float t[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) t[k] = src[d + k];
#pragma unroll
        for (int k = 0; k < CH; ++k) atomicAdd(dst + d + k, t[k]);
    }
    return d;
}
__device__ __forceinline__ void merge_row(float* dst, const float* src, int D) {
    int d = merge_chunks<16>(dst, src, 0, D);
    d = merge_chunks<4>(dst, src, d, D);
    merge_chunks<1>(dst, src, d, D);
}

__global__ __launch_bounds__(256) void k_rows_dedupe(const int64_t* __restrict__ rows, float* __restrict__ values,
                                                     int64_t n, int D, unsigned long long* __restrict__ slots,
                                                     int slots_log2, int* __restrict__ mark) {
    const int64_t occ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (occ >= n) return;
    const int64_t row = rows[occ];
    if (row < 0) {            // out-of-range lookup: contributes nothing
        mark[occ] = -1;
        return;
    }
    const unsigned mask = (1u << slots_log2) - 1u;
    unsigned h = hash_row((unsigned)row, 32 - slots_log2);
    const unsigned long long mine = ((unsigned long long)(row + 1) << 32) | (unsigned long long)(unsigned)occ;
    for (;;) {
        const unsigned long long prev = atomicCAS(&slots[h], 0ULL, mine);
        if (prev == 0ULL) {   // owner
            mark[occ] = (int)h;
            return;
        }
        if ((prev >> 32) == (unsigned long long)(row + 1)) {   // duplicate of an earlier occurrence
            const int64_t owner = (int64_t)(prev & 0xffffffffULL);
            merge_row(values + owner * D, values + occ * D, D);
            mark[occ] = -1;
            return;
        }
        h = (h + 1) & mask;
    }
}

// Field-local variant of pass 1 for lookups of a PACKED table laid out [.., fields]: occurrence occ belongs to field
// occ % fields and the fields' row ranges are disjoint, so each field dedupes on its own — one workgroup per field
// with the hash in LDS (16K slots: the key word, and one word holding the owner's index in the low 13 bits and the
// duplicate count above them).  Rows looked up kHotMin+ times in the step (skewed ids: the head of a Zipf
// distribution, low-cardinality columns) get an LDS accumulator: their occurrences are summed with LDS atomics and
// the owner's gradient row is overwritten with the sum, so the hot rows never serialise on one global address.  The
// remaining (rare) duplicates add into their owner's row with global atomics.
constexpr int kFieldSlotsLog2 = 14;
constexpr int kFieldSlots = 1 << kFieldSlotsLog2;
constexpr int kOwnerBits = 13;                  // up to 8192 lookups per field
constexpr int kHotMin = 4;                      // duplicates (beyond the owner) that make a row "hot"
constexpr int kHotBytes = 30 * 1024;            // LDS left for the hot accumulators

__global__ __launch_bounds__(1024) void k_rows_dedupe_fields(const int64_t* __restrict__ rows,
                                                             float* __restrict__ values, int64_t n, int D, int fields,
                                                             int* __restrict__ mark) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_u[];
    unsigned* keys = lds_u;                                  // [kFieldSlots] row+1, 0 = empty; later: hot index+1
    unsigned* oc = lds_u + kFieldSlots;                      // [kFieldSlots] owner index | dup count << 13
    float* acc = reinterpret_cast<float*>(lds_u + 2 * kFieldSlots);   // [hot_cap][D+1]
    __shared__ int n_hot;
    const int f = blockIdx.x;
    const int cnt = (int)(n / fields);
    const int accs = D + 1;                                  // odd stride: hot rows spread over the banks
    const int hot_cap = kHotBytes / (4 * accs);
    // this thread's (up to 8) lookups: all loads issued before the first LDS atomic, kept in registers
    constexpr int kPer = kFieldSlots / 2 / 1024;
    int64_t row[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = threadIdx.x + k * 1024;
        row[k] = i < cnt ? rows[(int64_t)i * fields + f] : -1;
    }
    for (int e = threadIdx.x; e < kFieldSlots; e += blockDim.x) { keys[e] = 0u; oc[e] = 0u; }
    for (int e = threadIdx.x; e < hot_cap * accs; e += blockDim.x) acc[e] = 0.f;
    if (threadIdx.x == 0) n_hot = 0;
    __syncthreads();
    constexpr unsigned mask = kFieldSlots - 1;
    unsigned slot[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        if (row[k] < 0) continue;
        const unsigned key = (unsigned)row[k] + 1u;
        unsigned h = hash_row((unsigned)row[k], 32 - kFieldSlotsLog2);
        unsigned prev;
        for (;;) {
            prev = atomicCAS(&keys[h], 0u, key);
            if (prev == 0u || prev == key) break;
            h = (h + 1) & mask;
        }
        // exactly one inserter per slot: it deposits its index; everyone else counts as a duplicate
        atomicAdd(&oc[h], prev == 0u ? (unsigned)(threadIdx.x + k * 1024) : (1u << kOwnerBits));
        slot[k] = h;
    }
    __syncthreads();
    // hot slots get an accumulator; `keys` is free now (every lookup remembers its slot) and holds hot index + 1
    for (int e = threadIdx.x; e < kFieldSlots; e += blockDim.x) {
        unsigned hi = 0u;
        if ((oc[e] >> kOwnerBits) >= (unsigned)kHotMin) {
            const int idx = atomicAdd(&n_hot, 1);
            if (idx < hot_cap) hi = (unsigned)idx + 1u;
        }
        keys[e] = hi;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = threadIdx.x + k * 1024;
        if (i >= cnt) continue;
        const int64_t occ = (int64_t)i * fields + f;
        if (row[k] < 0) { mark[occ] = -1; continue; }
        const unsigned hi = keys[slot[k]];
        const int o = (int)(oc[slot[k]] & ((1u << kOwnerBits) - 1u));
        if (hi) {                                            // hot: every occurrence (the owner too) adds in LDS
            float* a = acc + (hi - 1u) * accs;
            const float* src = values + occ * D;
            int d = 0;
            for (; d + 16 <= D; d += 16) {
                float t[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) t[q] = src[d + q];
#pragma unroll
                for (int q = 0; q < 16; ++q) atomicAdd(a + d + q, t[q]);
            }
            for (; d < D; ++d) atomicAdd(a + d, src[d]);
            mark[occ] = (o == i) ? 0 : -1;
        } else if (o == i) {
            mark[occ] = 0;
        } else {
            merge_row(values + ((int64_t)o * fields + f) * D, values + occ * D, D);
            mark[occ] = -1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k) {                          // hot owners: gradient row := LDS sum
        const int i = threadIdx.x + k * 1024;
        if (i >= cnt || row[k] < 0) continue;
        const unsigned hi = keys[slot[k]];
        if (!hi || (int)(oc[slot[k]] & ((1u << kOwnerBits) - 1u)) != i) continue;
        const float* a = acc + (hi - 1u) * accs;
        float* dst = values + ((int64_t)i * fields + f) * D;
        for (int d = 0; d < D; ++d) dst[d] = a[d];
    }
}

template <int VW>
__device__ __forceinline__ void adam_rows_owner_body(float* __restrict__ table, float* __restrict__ m,
                                                     float* __restrict__ v, const int64_t* __restrict__ rows,
                                                     const float* __restrict__ values, int64_t n, int D,
                                                     unsigned long long* __restrict__ slots,
                                                     const int* __restrict__ mark, float lr_t, float b1, float b2,
                                                     float eps, int sstride);

template <int VW>
__global__ __launch_bounds__(256) void k_adam_rows_owner(float* __restrict__ table, float* __restrict__ m,
                                                         float* __restrict__ v, const int64_t* __restrict__ rows,
                                                         const float* __restrict__ values, int64_t n, int D,
                                                         unsigned long long* __restrict__ slots,
                                                         const int* __restrict__ mark, float lr_host,
                                                         AdamState* __restrict__ st, float b1, float b2, float eps,
                                                         int row_blocks, DenseTail tail, int advance, float lr,
                                                         int sstride, SegTail seg, int segments_only) {
    unsigned ticket;
    const float lr_t = adam_read_lr(st, lr_host, advance, ticket);
    if ((int)blockIdx.x >= row_blocks) {      // trailing blocks: the model's dense parameters (one flat buffer)
        adam_dense_range(tail, (int64_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x,
                         (int64_t)(gridDim.x - row_blocks) * blockDim.x, lr_t, b1, b2, eps);
    } else {
        // the segments of rows looked up several times ride on the row blocks' waves (extra blocks would each take the
        // state's arrival ticket: +6 us for 1024 of them); a wave's region count is loaded before its own rows
        int nseg0 = 0;
        if (VW == 4 && seg.nseg) nseg0 = seg.nseg[((int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6)) % seg.regions];
        // segments_only: the rows looked up once were updated inside the train step (dt_deepfm_train_step_adam)
        if (!segments_only) adam_rows_owner_body<VW>(table, m, v, rows, values, n, D, slots, mark, lr_t, b1, b2, eps, sstride);
        if (VW == 4 && seg.nseg) adam_segments(seg, row_blocks, nseg0, table, m, v, values, D, lr_t, b1, b2, eps, sstride);
    }
    adam_finish(st, ticket, lr, b1, b2);
}

constexpr int kAdamPieces = 1;   // pieces per thread: 4 measured slower (29-32 us vs 28 us on the DeepFM step)
template <int VW>
__device__ __forceinline__ void adam_rows_owner_body(float* __restrict__ table, float* __restrict__ m,
                                                     float* __restrict__ v, const int64_t* __restrict__ rows,
                                                     const float* __restrict__ values, int64_t n, int D,
                                                     unsigned long long* __restrict__ slots,
                                                     const int* __restrict__ mark, float lr_t, float b1, float b2,
                                                     float eps, int sstride) {
    // kAdamPieces pieces (VW floats of one row) per thread, the loads of all of them in flight before the first update:
    // 4x fewer blocks take the state's arrival ticket (each ~350 ns when they queue on one address) and every lane has
    // up to 16 independent loads outstanding
    const int lpr = D / VW;
    const int64_t base = (int64_t)blockIdx.x * blockDim.x * kAdamPieces + threadIdx.x;
    float gi[kAdamPieces][VW], mi[kAdamPieces][VW], vi[kAdamPieces][VW], pi[kAdamPieces][VW];
    int64_t i0[kAdamPieces], s0[kAdamPieces];
    int slot[kAdamPieces];
    bool lead[kAdamPieces];
#pragma unroll
    for (int q = 0; q < kAdamPieces; ++q) {
        const int64_t gt = base + (int64_t)q * blockDim.x;
        const int64_t occ = gt / lpr;
        const int part = (int)(gt - occ * lpr);
        slot[q] = -1;
        lead[q] = part == 0;
        if (occ < n) {
            const int64_t row = rows[occ];
            slot[q] = mark ? mark[occ] : (row >= 0 ? 0 : -1);   // no mark: rows are already distinct
            if (slot[q] >= 0) {
                i0[q] = row * D + part * VW;
                s0[q] = row * sstride + part * VW;      // the row's slot record: m | v interleaved (sstride = 2D) or separate (D)
                const float* gsrc = values + occ * D + part * VW;
                if (VW == 4) {
                    *reinterpret_cast<float4*>(gi[q]) = *reinterpret_cast<const float4*>(gsrc);
                    *reinterpret_cast<float4*>(mi[q]) = *reinterpret_cast<const float4*>(m + s0[q]);
                    *reinterpret_cast<float4*>(vi[q]) = *reinterpret_cast<const float4*>(v + s0[q]);
                    *reinterpret_cast<float4*>(pi[q]) = *reinterpret_cast<const float4*>(table + i0[q]);
                } else {
#pragma unroll
                    for (int k = 0; k < VW; ++k) {
                        gi[q][k] = gsrc[k]; mi[q][k] = m[s0[q] + k]; vi[q][k] = v[s0[q] + k]; pi[q][k] = table[i0[q] + k];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kAdamPieces; ++q) {
        if (slot[q] < 0) continue;
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            mi[q][k] = b1 * mi[q][k] + (1.f - b1) * gi[q][k];
            vi[q][k] = b2 * vi[q][k] + (1.f - b2) * gi[q][k] * gi[q][k];
            pi[q][k] -= lr_t * mi[q][k] / (sqrtf(vi[q][k]) + eps);
        }
        if (VW == 4) {
            *reinterpret_cast<float4*>(m + s0[q]) = *reinterpret_cast<float4*>(mi[q]);
            *reinterpret_cast<float4*>(v + s0[q]) = *reinterpret_cast<float4*>(vi[q]);
            *reinterpret_cast<float4*>(table + i0[q]) = *reinterpret_cast<float4*>(pi[q]);
        } else {
#pragma unroll
            for (int k = 0; k < VW; ++k) { m[s0[q] + k] = mi[q][k]; v[s0[q] + k] = vi[q][k]; table[i0[q] + k] = pi[q][k]; }
        }
        if (lead[q] && slots) slots[slot[q]] = 0ULL;   // global-hash variant: leave the hash empty for the next step
    }
}

// ---- BinaryCrossentropy from logits (the loss Keras evaluates for a sigmoid output in graph mode, deepmodel.py:326-328):
//   loss = mean(max(z,0) - z*y + log1p(exp(-|z|))),  dloss/dz = (sigmoid(z) - y) / n      (one launch for both)
__global__ __launch_bounds__(256) void k_bce_logits(const float* __restrict__ z, const float* __restrict__ y, int64_t n,
                                                    float* __restrict__ loss, float* __restrict__ dz) {
    const float inv_n = 1.0f / (float)n;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float zi = z[i], yi = y[i];
        const float e = expf(-fabsf(zi));
        acc += fmaxf(zi, 0.f) - zi * yi + log1pf(e);
        const float sig = zi >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        dz[i] = (sig - yi) * inv_n;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv_n);
}
// the same as ONE workgroup (n <= 32 K: a minibatch's logits): the loss is stored, not accumulated — no zero-fill node ahead
// of the launch, no atomics, a fixed summation order
__global__ __launch_bounds__(1024) void k_bce_logits_one(const float* __restrict__ z, const float* __restrict__ y, int n,
                                                         float* __restrict__ loss, float* __restrict__ dz) {
    __shared__ float red[16];
    const float inv_n = 1.0f / (float)n;
    float acc = 0.f;
    for (int base = 0; base < n; base += 8 * 1024) {        // eight independent loads of z and y in flight per thread
        float zv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(base + u * 1024 + (int)threadIdx.x, n - 1);
            zv[u] = z[i];
            yv[u] = y[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 1024 + (int)threadIdx.x;
            if (i < n) {
                const float zi = zv[u], yi = yv[u];
                const float e = expf(-fabsf(zi));
                acc += fmaxf(zi, 0.f) - zi * yi + log1pf(e);
                const float sig = zi >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
                dz[i] = (sig - yi) * inv_n;
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = t * inv_n;
    }
}

// ---- SGD (keras.optimizers.SGD, momentum 0): p -= lr * g; duplicates of a row simply add up ----------------
__global__ __launch_bounds__(256) void k_sgd_dense(float* __restrict__ p, const float* __restrict__ g, int64_t n,
                                                   float lr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] -= lr * g[i];
}

__global__ __launch_bounds__(256) void k_sgd_rows(float* __restrict__ table, const int64_t* __restrict__ rows,
                                                  const float* __restrict__ values, int64_t total, int D, float lr) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t occ = t / D;
        const int64_t row = rows[occ];
        if (row >= 0) atomicAdd(table + row * D + (t - occ * D), -lr * values[t]);
    }
}

// ---- bucketed sparse gradient for the data-parallel exchange (north_star: "bucketed sparse embedding grads") -------------
// A fused step hands out the rows looked up once as per-lookup entries (rows >= 0) and the rows looked up several times as
// segments (DedupeWs, csrc/deepfm.hip).  k_rows_compact turns that into UNIQUE (row, summed gradient x scale) entries
// packed at the front of (out_rows, out_vals): what a rank puts on the wire instead of one entry per lookup — with
// Zipf ids half as many entries, and every receiver applies W x (unique rows) updates instead of W x B x F.
//   leading blocks : one lane group (D/4 lanes) per lookup; the wave's valid lookups take consecutive slots behind ONE
//                    atomicAdd on the counter (ballot + popcount)
//   trailing blocks: one wave per segment (the walk of adam_segments): members summed, one slot per segment
// Entries beyond `cap` are dropped and counted in counter[1] (the caller checks it on the host outside the step);
// out_rows must be pre-filled with -1.
__global__ __launch_bounds__(256) void k_rows_compact(const int64_t* __restrict__ rows, const float* __restrict__ values,
                                                      int64_t n, int D, SegTail sg, int row_blocks, int seg_blocks,
                                                      float scale, int64_t cap, int64_t* __restrict__ out_rows,
                                                      float* __restrict__ out_vals, int* __restrict__ counter) {
    const int lane = threadIdx.x & 63;
    const int lpr = D >> 2, groups = 64 / lpr, grp = lane / lpr, part = lane - grp * lpr;
    if ((int)blockIdx.x < row_blocks) {
        const int64_t occ = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / lpr;
        const bool in = occ < n;
        const int64_t row = in ? rows[occ] : -1;
        const bool lead = part == 0 && row >= 0;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= 0) g = *reinterpret_cast<const float4*>(values + occ * D + 4 * part);
        const unsigned long long mask = __ballot(lead);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(&counter[0], __popcll(mask));
        base = __builtin_amdgcn_readfirstlane(base);
        // slot of this lane's lookup: its lead lane's rank among the wave's lead lanes
        const int lead_lane = grp * lpr;
        const int pos = base + __popcll(mask & ((1ULL << lead_lane) - 1ULL));
        if (row >= 0) {
            if (pos < cap) {
                if (part == 0) out_rows[pos] = row;
                *reinterpret_cast<float4*>(out_vals + (int64_t)pos * D + 4 * part) =
                    make_float4(g.x * scale, g.y * scale, g.z * scale, g.w * scale);
            } else if (part == 0) {
                atomicAdd(&counter[1], 1);
            }
        }
        return;
    }
    if (!sg.nseg) return;
    const int gw = ((int)blockIdx.x - row_blocks) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    const int nw = seg_blocks * (int)(blockDim.x >> 6);
    const int wpr = nw >= sg.regions ? nw / sg.regions : 1, lw = nw >= sg.regions ? gw / sg.regions : 0;
    if (lw >= wpr) return;
    for (int e = gw % sg.regions; e < sg.regions; e += (nw >= sg.regions ? sg.regions : nw)) {
        const int nseg = sg.nseg[e];
        for (int sl = lw; sl < nseg; sl += wpr) {
            const int s = e * sg.cap + sl;
            const int64_t row = sg.row[s];
            const int off = sg.off[s], cnt = sg.cnt[s];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = grp; i < cnt; i += groups) {
                const int o0 = sg.list[off + i];
                const float4 g0 = *reinterpret_cast<const float4*>(values + (int64_t)o0 * D + 4 * part);
                acc.x += g0.x; acc.y += g0.y; acc.z += g0.z; acc.w += g0.w;
            }
            for (int o = lpr; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
                acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
            }
            int pos = 0;
            if (lane == 0) pos = atomicAdd(&counter[0], 1);
            pos = __builtin_amdgcn_readfirstlane(pos);
            if (grp == 0) {
                if (pos < cap) {
                    if (part == 0) out_rows[pos] = row;
                    *reinterpret_cast<float4*>(out_vals + (int64_t)pos * D + 4 * part) =
                        make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
                } else if (part == 0) {
                    atomicAdd(&counter[1], 1);
                }
            }
        }
    }
}

// The cheap form of the bucket (no slot counter: 18 K returning atomics on one word cost 200 us): every segment's members are
// summed INTO the gradient row of its first member, whose entry of `rows` gets the table row back; the other members
// keep row -1.  The (rows, values) pair then holds one entry per distinct row of the rank (and holes the consumers skip)
// in its original [.., fields] layout: same wire size, but a receiver applies W x (distinct rows) updates.
__global__ __launch_bounds__(256) void k_rows_merge_segments(int64_t* __restrict__ rows, float* __restrict__ values, int D,
                                                             SegTail sg, int seg_blocks) {
    const int lane = threadIdx.x & 63;
    const int lpr = D >> 2, groups = 64 / lpr, grp = lane / lpr, part = lane - grp * lpr;
    const int gw = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    const int nw = seg_blocks * (int)(blockDim.x >> 6);
    const int wpr = nw >= sg.regions ? nw / sg.regions : 1, lw = nw >= sg.regions ? gw / sg.regions : 0;
    if (lw >= wpr) return;
    for (int e = gw % sg.regions; e < sg.regions; e += (nw >= sg.regions ? sg.regions : nw)) {
        const int nseg = sg.nseg[e];
        for (int sl = lw; sl < nseg; sl += wpr) {
            const int s = e * sg.cap + sl;
            const int64_t row = sg.row[s];
            const int off = sg.off[s], cnt = sg.cnt[s];
            const int first = sg.list[off];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int i = grp;
            for (; i + 3 * groups < cnt; i += 4 * groups) {      // four members per group in flight
                const int o0 = sg.list[off + i], o1 = sg.list[off + i + groups], o2 = sg.list[off + i + 2 * groups],
                          o3 = sg.list[off + i + 3 * groups];
                const float4 g0 = *reinterpret_cast<const float4*>(values + (int64_t)o0 * D + 4 * part);
                const float4 g1 = *reinterpret_cast<const float4*>(values + (int64_t)o1 * D + 4 * part);
                const float4 g2 = *reinterpret_cast<const float4*>(values + (int64_t)o2 * D + 4 * part);
                const float4 g3 = *reinterpret_cast<const float4*>(values + (int64_t)o3 * D + 4 * part);
                acc.x += (g0.x + g1.x) + (g2.x + g3.x); acc.y += (g0.y + g1.y) + (g2.y + g3.y);
                acc.z += (g0.z + g1.z) + (g2.z + g3.z); acc.w += (g0.w + g1.w) + (g2.w + g3.w);
            }
            for (; i < cnt; i += groups) {
                const int o0 = sg.list[off + i];
                const float4 g0 = *reinterpret_cast<const float4*>(values + (int64_t)o0 * D + 4 * part);
                acc.x += g0.x; acc.y += g0.y; acc.z += g0.z; acc.w += g0.w;
            }
            for (int o = lpr; o < 64; o <<= 1) {
                acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
                acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
            }
            // every read of the members' rows (this wave's, above) is complete before the first member's row is overwritten
            if (grp == 0) {
                *reinterpret_cast<float4*>(values + (int64_t)first * D + 4 * part) = acc;
                if (part == 0) rows[first] = row;
            }
        }
    }
}

}  // namespace dt

using namespace dt;

extern "C" int dt_rows_merge_segments(int64_t* rows, float* values, int D, const int* seg_nseg, const int64_t* seg_row,
                                      const int* seg_off, const int* seg_cnt, const int* seg_list, int seg_regions,
                                      int seg_cap, void* stream) {
    DT_REQUIRE(rows && values && seg_nseg && seg_row && seg_off && seg_cnt && seg_list && seg_regions > 0 && seg_cap > 0,
               "dt_rows_merge_segments: bad arguments");
    const int lpr = D / 4;
    DT_UNSUPPORTED(D % 4 || lpr < 1 || lpr > 64 || (lpr & (lpr - 1)), "dt_rows_merge_segments: D = 4 * 2^k <= 256 (D=%d)", D);
    const SegTail sg{seg_nseg, seg_row, seg_off, seg_cnt, seg_list, seg_regions, seg_cap};
    const int seg_blocks = 1024;
    hipLaunchKernelGGL(k_rows_merge_segments, dim3(seg_blocks), dim3(256), 0, as_stream(stream), rows, values, D, sg,
                       seg_blocks);
    return launch_status("dt_rows_merge_segments");
}

extern "C" int dt_rows_compact(const int64_t* rows, const float* values, int64_t n_rows, int D, const int* seg_nseg,
                               const int64_t* seg_row, const int* seg_off, const int* seg_cnt, const int* seg_list,
                               int seg_regions, int seg_cap, float scale, int64_t cap, int64_t* out_rows, float* out_vals,
                               int* counter2, void* stream) {
    DT_REQUIRE(n_rows >= 0 && cap > 0 && rows && values && out_rows && out_vals && counter2, "dt_rows_compact: bad arguments");
    const int lpr = D / 4;
    DT_UNSUPPORTED(D % 4 || lpr < 1 || lpr > 64 || (lpr & (lpr - 1)), "dt_rows_compact: D = 4 * 2^k <= 256 (D=%d)", D);
    DT_REQUIRE(!seg_nseg || (seg_row && seg_off && seg_cnt && seg_list && seg_regions > 0 && seg_cap > 0),
               "dt_rows_compact: bad segment arrays");
    hipStream_t st = as_stream(stream);
    // only the slot counter restarts: counter2[1] (entries that did not fit) is a RUNNING total over the calls, so a host check
    // after many steps still sees an overflow of any of them (the caller zeroes the pair once, when it allocates it)
    hipMemsetAsync(counter2, 0, sizeof(int), st);
    hipMemsetAsync(out_rows, 0xff, (size_t)cap * sizeof(int64_t), st);          // -1: unused slots are skipped by every consumer
    const SegTail sg{seg_nseg, seg_row, seg_off, seg_cnt, seg_list, seg_regions, seg_cap};
    const int row_blocks = (int)((n_rows * lpr + 255) / 256);
    const int seg_blocks = seg_nseg ? 1024 : 0;
    if (row_blocks + seg_blocks == 0) return DT_OK;
    hipLaunchKernelGGL(k_rows_compact, dim3((unsigned)(row_blocks + seg_blocks)), dim3(256), 0, st, rows, values, n_rows, D, sg,
                       row_blocks, seg_blocks, scale, cap, out_rows, out_vals, counter2);
    return launch_status("dt_rows_compact");
}

extern "C" int dt_bce_logits(const float* z, const float* y, int64_t n, float* loss, float* dz, void* stream) {
    DT_REQUIRE(n > 0 && z && y && loss && dz, "dt_bce_logits: bad arguments");
    hipStream_t st = as_stream(stream);
    if (n <= 32768) {
        hipLaunchKernelGGL(k_bce_logits_one, dim3(1), dim3(1024), 0, st, z, y, (int)n, loss, dz);
        return launch_status("dt_bce_logits");
    }
    hipMemsetAsync(loss, 0, sizeof(float), st);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_bce_logits, dim3((unsigned)blocks), dim3(256), 0, st, z, y, n, loss, dz);
    return launch_status("dt_bce_logits");
}

extern "C" int dt_sgd_dense_step(float* p, const float* g, int64_t n, float lr, void* stream) {
    DT_REQUIRE(n >= 0, "dt_sgd_dense_step: n < 0");
    if (n == 0) return DT_OK;
    DT_REQUIRE(p && g, "dt_sgd_dense_step: null pointer");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_sgd_dense, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, n, lr);
    return launch_status("dt_sgd_dense_step");
}

extern "C" int dt_sgd_rows_step(float* table, const int64_t* rows, const float* values, int64_t n_rows, int D,
                                float lr, void* stream) {
    DT_REQUIRE(n_rows >= 0 && D > 0, "dt_sgd_rows_step: bad sizes");
    if (n_rows == 0) return DT_OK;
    DT_REQUIRE(table && rows && values, "dt_sgd_rows_step: null pointer");
    const int64_t total = n_rows * D;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_sgd_rows, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), table, rows, values, total, D,
                       lr);
    return launch_status("dt_sgd_rows_step");
}


extern "C" int dt_adam_state_init(void* state, float lr, float beta1, float beta2, int steps_done, void* stream) {
    DT_REQUIRE(state && steps_done >= 0, "dt_adam_state_init: bad arguments");
    hipLaunchKernelGGL(k_adam_state_init, dim3(1), dim3(1), 0, as_stream(stream), (AdamState*)state, lr, beta1, beta2,
                       steps_done);
    return launch_status("dt_adam_state_init");
}

extern "C" int dt_adam_advance(void* state, float lr, float beta1, float beta2, void* stream) {
    DT_REQUIRE(state, "dt_adam_advance: null state");
    hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(1), 0, as_stream(stream), (AdamState*)state, lr, beta1, beta2);
    return launch_status("dt_adam_advance");
}

extern "C" int dt_adam_dense_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t,
                                  float beta1, float beta2, float eps, void* state, int advance, float lr,
                                  void* stream) {
    DT_REQUIRE(n >= 0, "dt_adam_dense_step: n < 0");
    DT_REQUIRE(!advance || state, "dt_adam_dense_step: advance needs the device state");
    if (n == 0) {
        if (advance) return dt_adam_advance(state, lr, beta1, beta2, stream);
        return DT_OK;
    }
    DT_REQUIRE(p && g && m && v, "dt_adam_dense_step: null pointer");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_adam_dense, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr_t,
                       (AdamState*)state, beta1, beta2, eps, advance, lr);
    return launch_status("dt_adam_dense_step");
}

extern "C" int dt_adam_multi_step(int count, float* const* p, const float* const* g, float* const* m, float* const* v,
                                  const int64_t* n, float lr_t, float beta1, float beta2, float eps, void* state,
                                  int advance, float lr, void* stream) {
    DT_REQUIRE(count >= 0 && (count == 0 || (p && g && m && v && n)), "dt_adam_multi_step: bad arguments");
    DT_REQUIRE(!advance || state, "dt_adam_multi_step: advance needs the device state");
    if (count == 0) return advance ? dt_adam_advance(state, lr, beta1, beta2, stream) : DT_OK;
    hipStream_t st = as_stream(stream);
    for (int c0 = 0; c0 < count; c0 += kMultiMax) {
        AdamMulti d;
        d.count = count - c0 < kMultiMax ? count - c0 : kMultiMax;
        int blocks = 0;
        for (int t = 0; t < d.count; ++t) {
            const int64_t nt = n[c0 + t];
            DT_REQUIRE(nt >= 0 && nt < (1LL << 31) && p[c0 + t] && g[c0 + t] && m[c0 + t] && v[c0 + t],
                       "dt_adam_multi_step: tensor %d: bad size or null pointer", c0 + t);
            d.p[t] = p[c0 + t]; d.g[t] = g[c0 + t]; d.m[t] = m[c0 + t]; d.v[t] = v[c0 + t];
            d.n[t] = (int)nt;
            d.block_start[t] = blocks;
            blocks += (int)((nt + kMultiPerBlock - 1) / kMultiPerBlock);
        }
        d.block_start[d.count] = blocks;
        const bool last_chunk = c0 + kMultiMax >= count;
        if (blocks == 0) {
            if (advance && last_chunk) return dt_adam_advance(state, lr, beta1, beta2, stream);
            continue;
        }
        hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, st, d, lr_t, (AdamState*)state, beta1, beta2, eps,
                           (advance && last_chunk) ? 1 : 0, lr);
    }
    return launch_status("dt_adam_multi_step");
}

extern "C" int64_t dt_adam_rows_slots(int64_t n_rows) {
    int64_t s = 1024;
    while (s < 8 * n_rows) s <<= 1;   // load <= 1/8: probe chains are serialised device atomics (21 -> 12 us at 213 K rows)
    return s;
}

extern "C" int dt_adam_rows_step_seg(float* table, float* m, float* v, const int64_t* rows, float* values,
                                     int64_t n_rows, int D, int fields, void* slots, int64_t n_slots, int* mark,
                                     float lr_t, float beta1, float beta2, float eps, void* state, float* dense_p,
                                     const float* dense_g, float* dense_m, float* dense_v, int64_t dense_n, int advance,
                                     float lr, const int* seg_nseg, const int64_t* seg_row, const int* seg_off,
                                     const int* seg_cnt, const int* seg_list, int seg_regions, int seg_cap,
                                     int slot_stride, void* stream) {
    DT_REQUIRE(n_rows >= 0 && D > 0 && fields >= -2 && dense_n >= 0, "dt_adam_rows_step: bad sizes");
    DT_REQUIRE(slot_stride == D || slot_stride == 2 * D, "dt_adam_rows_step_seg: slot_stride %d (D = %d or 2 D)", slot_stride, D);
    // fields == -2: as -1 (rows distinct) AND the rows with an entry in `rows` were already updated inside the train step
    // (dt_deepfm_train_step_adam): only the segments and the dense tail are left
    const int segments_only = fields == -2 ? 1 : 0;
    if (segments_only) {
        fields = -1;
        DT_REQUIRE(D % 4 == 0, "dt_adam_rows_step_seg: fields = -2 needs D %% 4 == 0 (D=%d)", D);
        if (!seg_nseg) n_rows = 0;            // nothing sparse left: the dense tail runs as a plain dense step
    }
    DT_REQUIRE(!advance || state, "dt_adam_rows_step: advance needs the device state");
    DT_REQUIRE(dense_n == 0 || (dense_p && dense_g && dense_m && dense_v), "dt_adam_rows_step: null dense tail");
    if (n_rows == 0)      // nothing sparse this step: the tail (and the advance) run as a plain dense step
        return (dense_n > 0 || advance)
                   ? dt_adam_dense_step(dense_p, dense_g, dense_m, dense_v, dense_n, lr_t, beta1, beta2, eps, state,
                                        advance, lr, stream)
                   : DT_OK;
    hipStream_t st = as_stream(stream);
    AdamState* as = (AdamState*)state;
    const DenseTail tail{dense_p, dense_g, dense_m, dense_v, dense_n};
    int tail_blocks = (int)((dense_n + 255) / 256);
    if (tail_blocks > 1024) tail_blocks = 1024;
    DT_REQUIRE(table && m && v && rows && values, "dt_adam_rows_step: null pointer");
    DT_REQUIRE(n_rows < (1LL << 31), "dt_adam_rows_step: %lld occurrences do not fit the 32-bit slot field",
               (long long)n_rows);
    SegTail seg{seg_nseg, seg_row, seg_off, seg_cnt, seg_list, seg_regions, seg_cap};
    if (seg_nseg) {
        DT_REQUIRE(seg_row && seg_off && seg_cnt && seg_list && seg_regions > 0 && seg_cap > 0,
                   "dt_adam_rows_step_seg: bad segment arrays");
        const int lpr = D / 4;
        DT_UNSUPPORTED(D % 4 || lpr > 64 || (lpr & (lpr - 1)), "dt_adam_rows_step_seg: segments need D = 4 * 2^k <= 256 (D=%d)", D);
    }
    // slot layout, stated by the caller: slot_stride = D: two separate [V, D] arrays; 2 D: ONE [V, 2, D] array with m and v of a
    // row side by side (v = m + D) — a row's m and v then share a 128-byte line and the update touches two random locations
    // per row instead of three
    const int sstride = slot_stride;
    DT_REQUIRE(sstride == D || v == m + D, "dt_adam_rows_step_seg: slot_stride 2 D needs v == m + D (interleaved slots)");
    unsigned long long* gslots = nullptr;
    int* mk = mark;
    if (fields == -1) {
        mk = nullptr;                                      // rows are already distinct: no dedupe pass
    } else {
        DT_REQUIRE(mark, "dt_adam_rows_step: null mark");
        // field-local LDS hashes run one workgroup per field: with few fields and a large batch they leave most of
        // the chip idle (26 fields x 8192 lookups: 41 us against ~15 us for the global hash), so that case takes the
        // global hash when the caller provided one
        const bool few_blocks = fields < 64 && n_rows >= 65536 && slots && n_slots >= 2 * n_rows;
        const bool field_local = fields > 0 && n_rows % fields == 0 && n_rows / fields <= kFieldSlots / 2 && !few_blocks;
        if (field_local) {
            const size_t lds = (size_t)kFieldSlots * 8 + kHotBytes;
            DT_UNSUPPORTED(D + 1 > kHotBytes / 4, "dt_adam_rows_step: D=%d too large", D);
            hipFuncSetAttribute((const void*)k_rows_dedupe_fields, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_rows_dedupe_fields, dim3(fields), dim3(1024), lds, st, rows, values, n_rows, D, fields,
                               mark);
        } else {
            DT_REQUIRE(slots, "dt_adam_rows_step: null slots");
            int lg = 0;
            while ((1LL << lg) < n_slots) ++lg;
            DT_REQUIRE((1LL << lg) == n_slots && n_slots >= 2 * n_rows && lg <= 31 && lg >= 1,
                       "dt_adam_rows_step: n_slots=%lld must be a power of two >= 2*n_rows", (long long)n_slots);
            gslots = (unsigned long long*)slots;
            hipLaunchKernelGGL(k_rows_dedupe, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, rows, values,
                               n_rows, D, gslots, lg, mark);
        }
    }
    if (D % 4 == 0) {
        int row_blocks = (int)((n_rows * (D / 4) + 256 * kAdamPieces - 1) / (256 * kAdamPieces));
        // segments only: enough waves to walk the regions (every block takes the state's arrival ticket: few blocks)
        if (segments_only) row_blocks = row_blocks < 512 ? row_blocks : 512;
        hipLaunchKernelGGL(k_adam_rows_owner<4>, dim3((unsigned)(row_blocks + tail_blocks)), dim3(256), 0, st,
                           table, m, v, rows, values, n_rows, D, gslots, mk, lr_t, as, beta1, beta2, eps, row_blocks, tail,
                           advance, lr, sstride, seg, segments_only);
    } else {
        const int row_blocks = (int)((n_rows * D + 256 * kAdamPieces - 1) / (256 * kAdamPieces));
        hipLaunchKernelGGL(k_adam_rows_owner<1>, dim3((unsigned)(row_blocks + tail_blocks)), dim3(256), 0, st, table, m,
                           v, rows, values, n_rows, D, gslots, mk, lr_t, as, beta1, beta2, eps, row_blocks, tail, advance,
                           lr, sstride, seg, 0);
    }
    return launch_status("dt_adam_rows_step");
}

extern "C" int dt_adam_rows_step(float* table, float* m, float* v, const int64_t* rows, float* values, int64_t n_rows,
                                 int D, int fields, void* slots, int64_t n_slots, int* mark, float lr_t,
                                 float beta1, float beta2, float eps, void* state, float* dense_p,
                                 const float* dense_g, float* dense_m, float* dense_v, int64_t dense_n, int advance,
                                 float lr, void* stream) {
    return dt_adam_rows_step_seg(table, m, v, rows, values, n_rows, D, fields, slots, n_slots, mark, lr_t, beta1, beta2,
                                 eps, state, dense_p, dense_g, dense_m, dense_v, dense_n, advance, lr, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, 0, 0,
                                 (m && v == m + D) ? 2 * D : D /* this older entry point has no stride argument: v = m + D means
                                                                  interleaved [V, 2, D] slots */,
                                 stream);
}
