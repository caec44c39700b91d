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

__global__ __launch_bounds__(256) void k_adam_dense(float* __restrict__ p,
                                                    const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n, float lr_t, float b1, float b2,
                                                    float eps) {
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

__global__ __launch_bounds__(256) void k_adam_rows(float* __restrict__ table,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ grad_table,
                                                   const int64_t* __restrict__ rows, int n_rows,
                                                   int D, int* __restrict__ row_epoch, int epoch,
                                                   float lr_t, float b1, float b2, float eps) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows) return;
    const int64_t row = rows[t];
    if (row < 0) return;
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

extern "C" int dt_adam_dense_step(float* p, const float* g, float* m, float* v, int64_t n,
                                  float lr_t, float beta1, float beta2, float eps, void* stream) {
    DT_REQUIRE(n >= 0, "dt_adam_dense_step: n < 0");
    if (n == 0) return DT_OK;
    DT_REQUIRE(p && g && m && v, "dt_adam_dense_step: null pointer");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_adam_dense, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m,
                       v, n, lr_t, beta1, beta2, eps);
    return launch_status("dt_adam_dense_step");
}

extern "C" int dt_adam_rows_step(float* table, float* m, float* v, float* grad_table_dense,
                                 const int64_t* rows, int n_rows, int D, int* row_epoch, int epoch,
                                 float lr_t, float beta1, float beta2, float eps, void* stream) {
    DT_REQUIRE(n_rows >= 0 && D > 0, "dt_adam_rows_step: bad sizes");
    if (n_rows == 0) return DT_OK;
    DT_REQUIRE(table && m && v && grad_table_dense && rows && row_epoch,
               "dt_adam_rows_step: null pointer");
    hipLaunchKernelGGL(k_adam_rows, dim3(ceil_div(n_rows, 256)), dim3(256), 0, as_stream(stream),
                       table, m, v, grad_table_dense, rows, n_rows, D, row_epoch, epoch, lr_t, beta1,
                       beta2, eps);
    return launch_status("dt_adam_rows_step");
}
