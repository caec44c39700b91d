// mha.hip — per-row multi-head FIELD self-attention core.
//
// Replaces the op chain of MultiheadAttention.call, deeptables/models/layers.py:129-145:
//   split heads on the last axis / concat on batch, QK^T, / sqrt(d_h), softmax, (dropout),
//   .V, merge heads.   "Sequence length" is the field count F (26 at the Criteo shape), heads
//   are d_h = D/H wide (8), so one attention problem is a 26x26 score tile: far too small for
//   a matrix-core tile per problem, and the Q/K/V/residual projections (the FLOP-heavy part,
//   plain Dense layers in the reference) stay library GEMMs.  One lane owns one
//   (batch row, head, field) task; K/V rows of a (b,h) are wave-broadcast 16-byte loads out of
//   L1.  Nothing of size [B,H,F,F] touches HBM: the forward saves only the per-row
//   log-sum-exp, the backward recomputes the probabilities (flash-attention style).
#include "common.h"

namespace dt {

// attention-weight dropout (layers.py:141): the keep-mask hash of csrc/autoint.hip (same function, same seeds)
__device__ __forceinline__ float mha_keep(unsigned seed, unsigned thr, unsigned b, unsigned h, unsigned i, unsigned j,
                                          float inv_keep) {
    unsigned x = seed ^ (b * 0x9E3779B1u) ^ ((h * 64u * 64u + i * 64u + j) * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x >= thr ? inv_keep : 0.f;
}

template <int DHMAX, bool VEC4>
struct RowIO {
    // load d_h floats of row `p` into r[]
    static __device__ __forceinline__ void load(const float* __restrict__ p, int dh, float* r) {
        if (VEC4) {
#pragma unroll
            for (int d = 0; d < DHMAX; d += 4) {
                if (d < dh) {
                    const float4 v = *reinterpret_cast<const float4*>(p + d);
                    r[d] = v.x; r[d + 1] = v.y; r[d + 2] = v.z; r[d + 3] = v.w;
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < DHMAX; ++d) r[d] = d < dh ? p[d] : 0.f;
        }
    }
    static __device__ __forceinline__ void store(float* __restrict__ p, int dh, const float* r) {
        if (VEC4) {
#pragma unroll
            for (int d = 0; d < DHMAX; d += 4)
                if (d < dh)
                    *reinterpret_cast<float4*>(p + d) = make_float4(r[d], r[d + 1], r[d + 2], r[d + 3]);
        } else {
#pragma unroll
            for (int d = 0; d < DHMAX; ++d)
                if (d < dh) p[d] = r[d];
        }
    }
};

template <int DHMAX>
__device__ __forceinline__ float dotn(const float* a, const float* b, int dh) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DHMAX; ++d)
        if (d < dh) s += a[d] * b[d];
    return s;
}

template <int DHMAX, bool VEC4>
__global__ __launch_bounds__(256) void k_mha_fwd(const float* __restrict__ q,
                                                 const float* __restrict__ k,
                                                 const float* __restrict__ v, int B, int F, int D,
                                                 int H, int ld, float scale, float* __restrict__ out,
                                                 float* __restrict__ lse, unsigned drop_thr, float inv_keep,
                                                 unsigned seed) {
    const int dh = D / H;
    const int64_t total = (int64_t)B * H * F;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int i = (int)(t % F);
    const int h = (int)((t / F) % H);
    const int64_t b = t / ((int64_t)F * H);
    const int64_t base = b * F * D + (int64_t)h * dh;   // out: contiguous rows of D
    const int64_t ib = b * F * ld + (int64_t)h * dh;    // q / k / v: rows `ld` floats apart
    float qi[DHMAX], kr[DHMAX], acc[DHMAX];
#pragma unroll
    for (int d = 0; d < DHMAX; ++d) { qi[d] = 0.f; kr[d] = 0.f; acc[d] = 0.f; }
    RowIO<DHMAX, VEC4>::load(q + ib + (int64_t)i * ld, dh, qi);
    float m = -INFINITY;
    for (int j = 0; j < F; ++j) {
        RowIO<DHMAX, VEC4>::load(k + ib + (int64_t)j * ld, dh, kr);
        m = fmaxf(m, dotn<DHMAX>(qi, kr, dh) * scale);
    }
    float l = 0.f;
    for (int j = 0; j < F; ++j) {
        RowIO<DHMAX, VEC4>::load(k + ib + (int64_t)j * ld, dh, kr);
        const float p = expf(dotn<DHMAX>(qi, kr, dh) * scale - m);
        l += p;
        const float pk = drop_thr ? p * mha_keep(seed, drop_thr, (unsigned)b, h, i, j, inv_keep) : p;
        RowIO<DHMAX, VEC4>::load(v + ib + (int64_t)j * ld, dh, kr);
#pragma unroll
        for (int d = 0; d < DHMAX; ++d)
            if (d < dh) acc[d] += pk * kr[d];
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < DHMAX; ++d) acc[d] *= inv;
    RowIO<DHMAX, VEC4>::store(out + base + (int64_t)i * D, dh, acc);
    if (lse) lse[t] = m + logf(l);
}

template <int DHMAX, bool VEC4>
__global__ __launch_bounds__(256) void k_mha_bwd(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ gout,
    int B, int F, int D, int H, int ld, int ldg, float scale, float* __restrict__ gq, float* __restrict__ gk,
    float* __restrict__ gv, unsigned drop_thr, float inv_keep, unsigned seed) {
    const int dh = D / H;
    const int64_t total = (int64_t)B * H * F;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int r = (int)(t % F);
    const int h = (int)((t / F) % H);
    const int64_t b = t / ((int64_t)F * H);
    const int64_t base = b * F * D + (int64_t)h * dh;     // out / grad_out: contiguous rows of D
    const int64_t ib = b * F * ld + (int64_t)h * dh;      // q / k / v rows `ld` apart
    const int64_t gb = b * F * ldg + (int64_t)h * dh;     // grad_q / grad_k / grad_v rows `ldg` apart
    const int64_t lbase = (b * H + h) * F;
    using IO = RowIO<DHMAX, VEC4>;
    float a0[DHMAX], a1[DHMAX], a2[DHMAX], r0[DHMAX], r1[DHMAX], r2[DHMAX];
#pragma unroll
    for (int d = 0; d < DHMAX; ++d) { a0[d] = a1[d] = a2[d] = 0.f; r0[d] = r1[d] = r2[d] = 0.f; }

    // ---- role 1: row r as a QUERY -> grad_q[r] = sum_j dS_rj k_j ----
    IO::load(q + ib + (int64_t)r * ld, dh, r0);      // q_r
    IO::load(gout + base + (int64_t)r * D, dh, r1);   // dO_r
    IO::load(out + base + (int64_t)r * D, dh, r2);    // O_r
    const float delta_r = dotn<DHMAX>(r1, r2, dh);
    const float lse_r = lse[lbase + r];
    for (int j = 0; j < F; ++j) {
        IO::load(k + ib + (int64_t)j * ld, dh, a0);
        IO::load(v + ib + (int64_t)j * ld, dh, a1);
        const float p = expf(dotn<DHMAX>(r0, a0, dh) * scale - lse_r);
        const float kp = drop_thr ? mha_keep(seed, drop_thr, (unsigned)b, h, r, j, inv_keep) : 1.f;
        const float ds = p * (dotn<DHMAX>(r1, a1, dh) * kp - delta_r) * scale;
#pragma unroll
        for (int d = 0; d < DHMAX; ++d)
            if (d < dh) a2[d] += ds * a0[d];
    }
    IO::store(gq + gb + (int64_t)r * ldg, dh, a2);

    // ---- role 2: row r as a KEY/VALUE -> grad_k[r] = sum_i dS_ir q_i ; grad_v[r] = sum_i p_ir dO_i
    IO::load(k + ib + (int64_t)r * ld, dh, r0);      // k_r
    IO::load(v + ib + (int64_t)r * ld, dh, r1);      // v_r
#pragma unroll
    for (int d = 0; d < DHMAX; ++d) { a2[d] = 0.f; r2[d] = 0.f; }  // a2 = grad_k, r2 = grad_v
    for (int i = 0; i < F; ++i) {
        IO::load(q + ib + (int64_t)i * ld, dh, a0);     // q_i
        IO::load(gout + base + (int64_t)i * D, dh, a1);  // dO_i
        const float p = expf(dotn<DHMAX>(a0, r0, dh) * scale - lse[lbase + i]);
        const float kp = drop_thr ? mha_keep(seed, drop_thr, (unsigned)b, h, i, r, inv_keep) : 1.f;
        const float dP = dotn<DHMAX>(a1, r1, dh) * kp;
#pragma unroll
        for (int d = 0; d < DHMAX; ++d)
            if (d < dh) r2[d] += p * kp * a1[d];
        // delta_i = dO_i . O_i
        float oi[DHMAX];
#pragma unroll
        for (int d = 0; d < DHMAX; ++d) oi[d] = 0.f;
        IO::load(out + base + (int64_t)i * D, dh, oi);
        const float ds = p * (dP - dotn<DHMAX>(a1, oi, dh)) * scale;
#pragma unroll
        for (int d = 0; d < DHMAX; ++d)
            if (d < dh) a2[d] += ds * a0[d];
    }
    IO::store(gk + gb + (int64_t)r * ldg, dh, a2);
    IO::store(gv + gb + (int64_t)r * ldg, dh, r2);
}

}  // namespace dt

using namespace dt;

#define DT_MHA_LAUNCH(KERNEL, DHM, ...)                                                           \
    do {                                                                                          \
        if (vec4)                                                                                 \
            hipLaunchKernelGGL((KERNEL<DHM, true>), grid, dim3(256), 0, st, __VA_ARGS__);         \
        else                                                                                      \
            hipLaunchKernelGGL((KERNEL<DHM, false>), grid, dim3(256), 0, st, __VA_ARGS__);        \
    } while (0)

static bool mha_drop(float rate, unsigned* thr, float* inv_keep) {
    *thr = 0; *inv_keep = 1.f;
    if (rate <= 0.f) return true;
    if (rate >= 1.f) return false;
    double t = (double)rate * 4294967296.0;
    *thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (*thr == 0) *thr = 1;
    *inv_keep = 1.0f / (1.0f - rate);
    return true;
}

extern "C" int dt_mha_core_fwd(const float* q, const float* k, const float* v, int B, int F, int D,
                               int H, int ld, float dropout_rate, unsigned seed, float* out, float* lse,
                               void* stream) {
    unsigned thr; float inv_keep;
    DT_REQUIRE(mha_drop(dropout_rate, &thr, &inv_keep), "dt_mha_core_fwd: dropout_rate %f", dropout_rate);
    DT_REQUIRE(thr == 0 || F <= 64, "dt_mha_core_fwd: attention dropout supports up to 64 fields");
    DT_REQUIRE(ld >= D, "dt_mha_core_fwd: row stride %d < D=%d", ld, D);
    DT_REQUIRE(B >= 0 && F > 0 && D > 0 && H > 0 && D % H == 0,
               "dt_mha_core_fwd: bad sizes B=%d F=%d D=%d H=%d", B, F, D, H);
    if (B == 0) return DT_OK;
    DT_REQUIRE(q && k && v && out, "dt_mha_core_fwd: null pointer");
    const int dh = D / H;
    DT_UNSUPPORTED(dh > 64, "dt_mha_core_fwd: head width %d > 64", dh);
    const bool vec4 = (dh % 4 == 0) && (D % 4 == 0) && (ld % 4 == 0);
    const float scale = 1.0f / sqrtf((float)dh);
    const int64_t total = (int64_t)B * H * F;
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = as_stream(stream);
    if (dh <= 4) DT_MHA_LAUNCH(k_mha_fwd, 4, q, k, v, B, F, D, H, ld, scale, out, lse, thr, inv_keep, seed);
    else if (dh <= 8) DT_MHA_LAUNCH(k_mha_fwd, 8, q, k, v, B, F, D, H, ld, scale, out, lse, thr, inv_keep, seed);
    else if (dh <= 16) DT_MHA_LAUNCH(k_mha_fwd, 16, q, k, v, B, F, D, H, ld, scale, out, lse, thr, inv_keep, seed);
    else if (dh <= 32) DT_MHA_LAUNCH(k_mha_fwd, 32, q, k, v, B, F, D, H, ld, scale, out, lse, thr, inv_keep, seed);
    else DT_MHA_LAUNCH(k_mha_fwd, 64, q, k, v, B, F, D, H, ld, scale, out, lse, thr, inv_keep, seed);
    return launch_status("dt_mha_core_fwd");
}

extern "C" int dt_mha_core_bwd(const float* q, const float* k, const float* v, const float* out,
                               const float* lse, const float* grad_out, int B, int F, int D, int H, int ld,
                               int ldg, float dropout_rate, unsigned seed, float* grad_q, float* grad_k,
                               float* grad_v, void* stream) {
    unsigned thr; float inv_keep;
    DT_REQUIRE(mha_drop(dropout_rate, &thr, &inv_keep), "dt_mha_core_bwd: dropout_rate %f", dropout_rate);
    DT_REQUIRE(B >= 0 && F > 0 && D > 0 && H > 0 && D % H == 0 && ld >= D && ldg >= D, "dt_mha_core_bwd: bad sizes");
    if (B == 0) return DT_OK;
    DT_REQUIRE(q && k && v && out && lse && grad_out && grad_q && grad_k && grad_v,
               "dt_mha_core_bwd: null pointer");
    const int dh = D / H;
    DT_UNSUPPORTED(dh > 32, "dt_mha_core_bwd: head width %d > 32", dh);
    const bool vec4 = (dh % 4 == 0) && (D % 4 == 0) && (ld % 4 == 0) && (ldg % 4 == 0);
    const float scale = 1.0f / sqrtf((float)dh);
    const int64_t total = (int64_t)B * H * F;
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = as_stream(stream);
    if (dh <= 4)
        DT_MHA_LAUNCH(k_mha_bwd, 4, q, k, v, out, lse, grad_out, B, F, D, H, ld, ldg, scale, grad_q, grad_k, grad_v, thr, inv_keep, seed);
    else if (dh <= 8)
        DT_MHA_LAUNCH(k_mha_bwd, 8, q, k, v, out, lse, grad_out, B, F, D, H, ld, ldg, scale, grad_q, grad_k, grad_v, thr, inv_keep, seed);
    else if (dh <= 16)
        DT_MHA_LAUNCH(k_mha_bwd, 16, q, k, v, out, lse, grad_out, B, F, D, H, ld, ldg, scale, grad_q, grad_k, grad_v, thr, inv_keep, seed);
    else
        DT_MHA_LAUNCH(k_mha_bwd, 32, q, k, v, out, lse, grad_out, B, F, D, H, ld, ldg, scale, grad_q, grad_k, grad_v, thr, inv_keep, seed);
    return launch_status("dt_mha_core_bwd");
}
