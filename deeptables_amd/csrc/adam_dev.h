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
    for (int e = e0; e < sg.regions; e += (nw >= sg.regions ? sg.regions : nw)) {
        int sl = lw;
        int s = e * sg.cap + min(sl, sg.cap - 1);
        int64_t row = sg.row[s];
        int off = sg.off[s], cnt = sg.cnt[s];
        const int nseg = (e == e0 ? nseg0 : sg.nseg[e]);
        while (sl < nseg) {
            const int sn = e * sg.cap + min(sl + wpr, sg.cap - 1);
            const int64_t row_n = sg.row[sn];
            const int off_n = sg.off[sn], cnt_n = sg.cnt[sn];
            const int64_t i0 = row * D + 4 * part, s0 = row * sstride + 4 * part;
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f), mi = p, vi = p;
            if (grp == 0) {
                p = *reinterpret_cast<const float4*>(table + i0);
                mi = *reinterpret_cast<const float4*>(m + s0);
                vi = *reinterpret_cast<const float4*>(v + s0);
            }
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
            if (grp == 0) {
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
                *reinterpret_cast<float4*>(m + s0) = mi;
                *reinterpret_cast<float4*>(v + s0) = vi;
                *reinterpret_cast<float4*>(table + i0) = p;
            }
            sl += wpr;
            row = row_n; off = off_n; cnt = cnt_n;
        }
    }
}

}  // namespace dt
