// adam_dev.h — device-side pieces of the Keras-Adam step shared by optim.hip (the optimizer launches) and deepfm.hip (the
// pipelined DeepFM step finishes its own optimizer step): the device-resident step state, the dense-range update,
// the arrival ticket that advances the state, and the segment walk of the rows looked up several times.
#pragma once
#include "common.h"

namespace dt {

// device-resident step state (16 bytes): t = the step number the NEXT update uses (1-based), lr_t = its
// bias-corrected rate, done = block counter of the kernel that advances the state.  Keeping it on the device makes the
// whole optimizer step replayable from a hipGraph (no host scalar baked into the captured launches).  The state is
// advanced at the END of a step by the last block of the last kernel the host launches for that step (every block
// has read lr_t by then), so advancing costs no launch of its own.
constexpr int kAdamSub = 64;
struct AdamState {
    int t;
    float lr_t;
    unsigned done;            // sub-counters completed
    int pad;
    unsigned sub[kAdamSub];   // blocks arrived, by blockIdx % kAdamSub (spreads the arrivals over 64 addresses)
};

__device__ __forceinline__ float adam_lr_t(float lr, float b1, float b2, int t) {
    // float is enough: the factor multiplies a 1e-3 step (tests/test_optim_gpu.py checks 5 steps to 2e-6 absolute),
    // and a double pow costs microseconds on one lane at the tail of the step's last kernel
    return lr * sqrtf(1.0f - powf(b2, (float)t)) / (1.0f - powf(b1, (float)t));
}


struct DenseTail {              // a dense Adam update riding along another launch's trailing blocks
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
};

__device__ __forceinline__ void adam_dense_range(const DenseTail& d, int64_t first, int64_t stride, float lr_t,
                                                 float b1, float b2, float eps) {
    for (int64_t i = first; i < d.n; i += stride) {
        const float gi = d.g[i];
        const float mi = b1 * d.m[i] + (1.f - b1) * gi;
        const float vi = b2 * d.v[i] + (1.f - b2) * gi * gi;
        d.m[i] = mi;
        d.v[i] = vi;
        d.p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

__device__ __forceinline__ void adam_one(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int64_t i,
                                         float g, float lr_t, float b1, float b2, float eps) {
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

// Every block reads lr_t once (thread 0, broadcast through LDS).  In the step's LAST launch (advance != 0) thread 0 of
// every block takes an arrival ticket at the block's end; the block whose ticket completes the count advances the
// state — every other block has read lr_t by then.
//   no fence: nothing a block WROTE has to be seen by the advancing block; a device-wide fence here writes back the
//   whole L2 (+4 us).  Arrivals on ONE address serialise (~6 ns each when they stream, ~350 ns when blocks queue on the
//   returned value): two levels, 64 addresses.  Measured alternatives (DeepFM step, 3.6K-block row update): ticket taken
//   right after the lr_t read 32.6 us, at the block's end 28.1-30.2 us, no ticket + a one-thread kernel 29.0 + 4.0 us.
constexpr unsigned kNoTicket = 0xffffffffu;
__device__ __forceinline__ float adam_read_lr(AdamState* st, float lr_host, int advance, unsigned& ticket) {
    __shared__ float s_lr;
    ticket = kNoTicket;
    if (!st) return lr_host;
    if (threadIdx.x == 0) {
        s_lr = st->lr_t;
        if (advance) ticket = 0u;                 // taken in adam_finish
    }
    __syncthreads();
    return s_lr;
}
__device__ __forceinline__ void adam_finish(AdamState* st, unsigned ticket, float lr, float b1, float b2) {
    if (ticket == kNoTicket) return;              // threads other than 0, or a launch that does not advance
    const unsigned k = blockIdx.x % kAdamSub;
    const unsigned expect = (gridDim.x + kAdamSub - 1 - k) / kAdamSub;       // blocks with this residue
    if (atomicAdd(&st->sub[k], 1u) == expect - 1) {
        st->sub[k] = 0u;
        const unsigned groups = gridDim.x < (unsigned)kAdamSub ? gridDim.x : (unsigned)kAdamSub;
        if (atomicAdd(&st->done, 1u) == groups - 1) {
            st->done = 0u;
            const int t = st->t + 1;
            st->t = t;
            st->lr_t = adam_lr_t(lr, b1, b2, t);
        }
    }
}


// Segments (rows looked up several times in the step; built by the fused steps' election, csrc/deepfm.hip DedupeWs):
// the waves of the launch's row blocks walk them after their own rows — a wave's lanes form 64 / (D/4) row groups, every group
// sums every (64 / (D/4))-th member's gradient row (16-byte lanes), the groups meet by lane shuffles, and group 0 applies
// the Adam update to the table row.  No lookup ever adds into a shared row: hot rows (Zipf ids, low-cardinality columns)
// cost one wave a few loads instead of hundreds of same-address atomics.
struct SegTail {
    const int* nseg;             // [regions] segments per region (read on the device); NULL: no segments
    const int64_t* row;          // [regions][cap]
    const int *off, *cnt, *list;
    int regions, cap;
};
// DEEP: eight members per group in flight for the rows with many members (k_finish_step; +28 VGPRs — the optimizer launch of
// the layer-by-layer graphs, whose main body is latency-bound on its occupancy, keeps the four-deep loop only)
template <bool DEEP = false>
__device__ __forceinline__ void adam_segments(const SegTail& sg, int row_blocks, int nseg0, float* __restrict__ table,
                                              float* __restrict__ m, float* __restrict__ v,
                                              const float* __restrict__ values, int D, float lr_t, float b1, float b2,
                                              float eps, int sstride, int blk = -1) {
    const int lane = threadIdx.x & 63;
    const int lpr = D >> 2;                               // lanes per row (a power of two <= 64: checked by the host)
    const int groups = 64 / lpr, grp = lane / lpr, part = lane - grp * lpr;
    // blk: the block's index among the `row_blocks` blocks that walk segments (default: the launch's leading blocks)
    const int gw = (blk >= 0 ? blk : (int)blockIdx.x) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    const int nw = row_blocks * (int)(blockDim.x >> 6);
    // region e = gw % regions is shared by wpr waves (local index lw); fewer waves than regions: a wave walks several
    const int wpr = nw >= sg.regions ? nw / sg.regions : 1, lw = nw >= sg.regions ? gw / sg.regions : 0;
    if (lw >= wpr) return;
    const int e0 = gw % sg.regions;
    // (Round 5 tried a SUB-wave of 16 lanes per segment — four segments per wave sharing the round trips: with Zipf ids the
    // launch went from 24.9 to 35.8 us.  A wave runs its sub-waves in lockstep, so one hot row's hundreds of members held
    // the three other sub-waves' next segments back; with one wave per segment only that wave's own queue waits.)
    // Dependent round trips per segment: (segment record) -> (member list | p, m, v of the row) -> (members' gradient rows)
    // -> stores.  The record of the wave's NEXT segment is fetched while the current one is summed (and the first one
    // before its region's count is even known: index e cap + lw is always inside the arrays), and the row's p / m / v
    // do not wait for the member sums (round 2 walked list -> values -> p, m, v: five trips).
    // Round 5: TWO segments of the wave's queue per iteration (sl and sl + wpr) — their records, member lists, p / m / v and
    // members' rows travel in the same round trips.  With Zipf ids a batch of 8192 has ~15 K segments, 3.7 per wave: walking
    // them one after the other made the finishing launch 29.8 us against 16.2 us with uniform ids (878 segments: at most one
    // per wave, for which nothing changes).  Every load is unconditional from a clamped (valid) address.
    struct SegRec { int64_t row; int off, cnt; };
    auto rec = [&](int e, int sl, int nseg) {
        const int s = e * sg.cap + min(sl, sg.cap - 1);
        SegRec r{sg.row[s], sg.off[s], sg.cnt[s]};
        if (sl >= nseg) { r.row = 0; r.off = 0; r.cnt = 0; }       // (beyond the region's segments: nothing to walk)
        return r;
    };
    auto adam_row = [&](const SegRec& r, float4 acc, float4 p, float4 mi, float4 vi) {
        for (int o = lpr; o < 64; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (grp == 0 && r.cnt > 0) {
            const float g[4] = {acc.x, acc.y, acc.z, acc.w};
            float* pp = reinterpret_cast<float*>(&p);
            float* pm = reinterpret_cast<float*>(&mi);
            float* pv = reinterpret_cast<float*>(&vi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pm[k] = b1 * pm[k] + (1.f - b1) * g[k];
                pv[k] = b2 * pv[k] + (1.f - b2) * g[k] * g[k];
                pp[k] -= lr_t * pm[k] / (sqrtf(pv[k]) + eps);
            }
            *reinterpret_cast<float4*>(m + r.row * sstride + 4 * part) = mi;
            *reinterpret_cast<float4*>(v + r.row * sstride + 4 * part) = vi;
            *reinterpret_cast<float4*>(table + r.row * D + 4 * part) = p;
        }
    };
    for (int e = e0; e < sg.regions; e += (nw >= sg.regions ? sg.regions : nw)) {
        int sl = lw;
        // (the first records are requested before the region's count is known: index e cap + sl is always inside the arrays;
        // what lies beyond the count is dropped once it is)
        SegRec A, Bq;
        {
            const int sa = e * sg.cap + min(sl, sg.cap - 1), sb = e * sg.cap + min(sl + wpr, sg.cap - 1);
            A = SegRec{sg.row[sa], sg.off[sa], sg.cnt[sa]};
            Bq = SegRec{sg.row[sb], sg.off[sb], sg.cnt[sb]};
        }
        const int nseg = (e == e0 ? nseg0 : sg.nseg[e]);
        if (sl >= nseg) A = SegRec{0, 0, 0};
        if (sl + wpr >= nseg) Bq = SegRec{0, 0, 0};
        while (sl < nseg) {
            const SegRec An = rec(e, sl + 2 * wpr, nseg), Bn = rec(e, sl + 3 * wpr, nseg);
            const SegRec R[2] = {A, Bq};
            float4 p[2], mi[2], vi[2], acc[2];
            int o[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t i0 = R[q].row * D + 4 * part, s0 = R[q].row * sstride + 4 * part;
                p[q] = *reinterpret_cast<const float4*>(table + i0);
                mi[q] = *reinterpret_cast<const float4*>(m + s0);
                vi[q] = *reinterpret_cast<const float4*>(v + s0);
#pragma unroll
                for (int k = 0; k < 4; ++k)                    // the first four members of every group (clamped: cnt >= 2 or 0)
                    o[q][k] = sg.list[R[q].off + min(grp + k * groups, max(R[q].cnt - 1, 0))];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = *reinterpret_cast<const float4*>(values + (int64_t)o[q][k] * D + 4 * part);
                acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float w = grp + k * groups < R[q].cnt ? 1.f : 0.f;
                    acc[q].x += w * g[k].x; acc[q].y += w * g[k].y; acc[q].z += w * g[k].z; acc[q].w += w * g[k].w;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {                        // a row with more than 4 x groups members: the rest
                int i = grp + 4 * groups;
                // a HOT row (Zipf ids: the top row of a field takes 8 % of its lookups — 650 members at B = 8192) is the launch's
                // longest chain on its one wave: eight members per group in flight per round trip (sixteen: 202 VGPRs, two waves per SIMD)
                for (; DEEP && i + 7 * groups < R[q].cnt; i += 8 * groups) {
                    int oo[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) oo[k] = sg.list[R[q].off + i + k * groups];
                    float4 gg[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) gg[k] = *reinterpret_cast<const float4*>(values + (int64_t)oo[k] * D + 4 * part);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { acc[q].x += gg[k].x; acc[q].y += gg[k].y; acc[q].z += gg[k].z; acc[q].w += gg[k].w; }
                }
                for (; i + 3 * groups < R[q].cnt; i += 4 * groups) {
                    const int o0 = sg.list[R[q].off + i], o1 = sg.list[R[q].off + i + groups],
                              o2 = sg.list[R[q].off + i + 2 * groups], o3 = sg.list[R[q].off + i + 3 * groups];
                    const float4 g0 = *reinterpret_cast<const float4*>(values + (int64_t)o0 * D + 4 * part);
                    const float4 g1 = *reinterpret_cast<const float4*>(values + (int64_t)o1 * D + 4 * part);
                    const float4 g2 = *reinterpret_cast<const float4*>(values + (int64_t)o2 * D + 4 * part);
                    const float4 g3 = *reinterpret_cast<const float4*>(values + (int64_t)o3 * D + 4 * part);
                    acc[q].x += (g0.x + g1.x) + (g2.x + g3.x); acc[q].y += (g0.y + g1.y) + (g2.y + g3.y);
                    acc[q].z += (g0.z + g1.z) + (g2.z + g3.z); acc[q].w += (g0.w + g1.w) + (g2.w + g3.w);
                }
                for (; i < R[q].cnt; i += groups) {
                    const int o0 = sg.list[R[q].off + i];
                    const float4 g0 = *reinterpret_cast<const float4*>(values + (int64_t)o0 * D + 4 * part);
                    acc[q].x += g0.x; acc[q].y += g0.y; acc[q].z += g0.z; acc[q].w += g0.w;
                }
            }
            adam_row(R[0], acc[0], p[0], mi[0], vi[0]);
            adam_row(R[1], acc[1], p[1], mi[1], vi[1]);
            sl += 2 * wpr;
            A = An; Bq = Bn;
        }
    }
}

}  // namespace dt
