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
//    Keras' dense semantics).  Duplicate lookups of a row are merged beforehand by
//    dt_embedding_bwd_dense into a dense gradient table; the first thread to claim the row
//    (atomicCAS on row_epoch) applies the update and re-zeroes the gradient row.
#include "common.h"

namespace dt {

// device-resident step state (8 bytes): int32 t, float lr_t.  Keeping it on the device makes the whole
// optimizer step replayable from a hipGraph (no host scalar baked into the captured launches).
struct AdamState {
    int t;
    float lr_t;
};

__global__ void k_adam_advance(AdamState* st, float lr, float b1, float b2) {
    const int t = st->t + 1;
    st->t = t;
    st->lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
}

__global__ __launch_bounds__(256) void k_adam_dense(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    float lr_host, const AdamState* __restrict__ st, float b1,
                                                    float b2, float eps) {
    const float lr_t = st ? st->lr_t : lr_host;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// 4 floats per lane, LPR = D/4 lanes per row (power of two): the 4 row-sized streams (table, m, v, merged
// gradient) are touched as contiguous 16-byte pieces by neighbouring lanes.  The group's first lane claims the row
// and shares the verdict by shuffle.
__global__ __launch_bounds__(256) void k_adam_rows_v4(float* __restrict__ table, float* __restrict__ m,
                                                      float* __restrict__ v, float* __restrict__ grad_table,
                                                      const int64_t* __restrict__ rows, int n_rows, int lpr_log2,
                                                      int* __restrict__ row_epoch, int epoch_host, float lr_host,
                                                      const AdamState* __restrict__ st, float b1, float b2,
                                                      float eps) {
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = gt >> lpr_log2;
    const int part = (int)(gt & ((1 << lpr_log2) - 1));
    if (r >= n_rows) return;                                  // whole groups leave together (groups are aligned)
    const int64_t row = rows[r];
    const int epoch = st ? st->t : epoch_host;
    const float lr_t = st ? st->lr_t : lr_host;
    int won = 0;
    if (part == 0 && row >= 0) won = atomicExch(&row_epoch[row], epoch) != epoch;
    won = __shfl(won, (int)(threadIdx.x & 63) - part, 64);
    if (!won) return;
    const int64_t i0 = (row << (lpr_log2 + 2)) + part * 4;
    const float4 g4 = *reinterpret_cast<const float4*>(grad_table + i0);
    float4 m4 = *reinterpret_cast<const float4*>(m + i0);
    float4 v4 = *reinterpret_cast<const float4*>(v + i0);
    float4 p4 = *reinterpret_cast<const float4*>(table + i0);
#define DT_ADAM1(c)                                     \
    m4.c = b1 * m4.c + (1.f - b1) * g4.c;               \
    v4.c = b2 * v4.c + (1.f - b2) * g4.c * g4.c;        \
    p4.c -= lr_t * m4.c / (sqrtf(v4.c) + eps);
    DT_ADAM1(x) DT_ADAM1(y) DT_ADAM1(z) DT_ADAM1(w)
#undef DT_ADAM1
    *reinterpret_cast<float4*>(grad_table + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(m + i0) = m4;
    *reinterpret_cast<float4*>(v + i0) = v4;
    *reinterpret_cast<float4*>(table + i0) = p4;
}

// any D: one lane per row
__global__ __launch_bounds__(256) void k_adam_rows(float* __restrict__ table, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ grad_table,
                                                   const int64_t* __restrict__ rows, int n_rows, int D,
                                                   int* __restrict__ row_epoch, int epoch_host, float lr_host,
                                                   const AdamState* __restrict__ st, float b1, float b2,
                                                   float eps) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows) return;
    const int64_t row = rows[t];
    if (row < 0) return;
    const int epoch = st ? st->t : epoch_host;
    const float lr_t = st ? st->lr_t : lr_host;
    if (atomicExch(&row_epoch[row], epoch) == epoch) return;  // someone else owns this row
    for (int d = 0; d < D; ++d) {
        const int64_t i = row * D + d;
        const float gi = grad_table[i];
        grad_table[i] = 0.f;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        table[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

}  // namespace dt

using namespace dt;

extern "C" int dt_adam_advance(void* state, float lr, float beta1, float beta2, void* stream) {
    DT_REQUIRE(state, "dt_adam_advance: null state");
    hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(1), 0, as_stream(stream), (AdamState*)state, lr, beta1, beta2);
    return launch_status("dt_adam_advance");
}

extern "C" int dt_adam_dense_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t,
                                  float beta1, float beta2, float eps, const void* state, void* stream) {
    DT_REQUIRE(n >= 0, "dt_adam_dense_step: n < 0");
    if (n == 0) return DT_OK;
    DT_REQUIRE(p && g && m && v, "dt_adam_dense_step: null pointer");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_adam_dense, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr_t,
                       (const AdamState*)state, beta1, beta2, eps);
    return launch_status("dt_adam_dense_step");
}

extern "C" int dt_adam_rows_step(float* table, float* m, float* v, float* grad_table_dense, const int64_t* rows,
                                 int n_rows, int D, int* row_epoch, int epoch, float lr_t, float beta1,
                                 float beta2, float eps, const void* state, void* stream) {
    DT_REQUIRE(n_rows >= 0 && D > 0, "dt_adam_rows_step: bad sizes");
    if (n_rows == 0) return DT_OK;
    DT_REQUIRE(table && m && v && grad_table_dense && rows && row_epoch, "dt_adam_rows_step: null pointer");
    const AdamState* st = (const AdamState*)state;
    const int lpr = D / 4;
    if (D % 4 == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0) {
        int lg = 0;
        while ((1 << lg) < lpr) ++lg;
        const int64_t threads = (int64_t)n_rows * lpr;
        hipLaunchKernelGGL(k_adam_rows_v4, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream),
                           table, m, v, grad_table_dense, rows, n_rows, lg, row_epoch, epoch, lr_t, st, beta1, beta2,
                           eps);
    } else {
        hipLaunchKernelGGL(k_adam_rows, dim3(ceil_div(n_rows, 256)), dim3(256), 0, as_stream(stream), table, m, v,
                           grad_table_dense, rows, n_rows, D, row_epoch, epoch, lr_t, st, beta1, beta2, eps);
    }
    return launch_status("dt_adam_rows_step");
}
