// Random-access ceilings of HBM3E for the two table-bound kernels of the DeepFM step (k_sparse_fwd, k_adam_rows_owner):
// what does the memory system deliver for 64-byte row gathers / read-modify-writes at random addresses of a 1.7 GB table,
// as a function of the number of rows per launch and the loads kept in flight per lane?  Streaming copy for reference.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_ubench.hip -o tools/ub_gather && tools/ub_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 4 lanes per 64-byte row; R rows per lane group in flight
template <int R>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, const int* __restrict__ rows,
                                                int64_t n, float4* __restrict__ out) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    const int64_t base = g * R;
    if (base >= n) return;
    int id[R];
    float4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) id[r] = base + r < n ? rows[base + r] : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = table[(int64_t)id[r] * 4 + c];
#pragma unroll
    for (int r = 0; r < R; ++r) if (base + r < n) out[(base + r) * 4 + c] = v[r];
}

// row update in place: p = p * 0.999f + 1 (64-byte read + 64-byte write per row)
__global__ __launch_bounds__(256) void k_rmw(float4* __restrict__ table, const int* __restrict__ rows, int64_t n) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    if (g >= n) return;
    float4* p = table + (int64_t)rows[g] * 4 + c;
    float4 v = *p;
    v.x = v.x * 0.999f + 1.f; v.y = v.y * 0.999f + 1.f; v.z = v.z * 0.999f + 1.f; v.w = v.w * 0.999f + 1.f;
    *p = v;
}

// Adam-shaped: gradient row streamed, table row (64 B) and slot record (128 B) read + written at random addresses
__global__ __launch_bounds__(256) void k_adamlike(float4* __restrict__ table, float4* __restrict__ mv,
                                                  const int* __restrict__ rows, const float4* __restrict__ grad,
                                                  int64_t n) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    if (g >= n) return;
    const int64_t row = rows[g];
    float4 gr = grad[g * 4 + c];
    float4 p = table[row * 4 + c], m = mv[row * 8 + c], v = mv[row * 8 + 4 + c];
    m.x = 0.9f * m.x + 0.1f * gr.x; m.y = 0.9f * m.y + 0.1f * gr.y; m.z = 0.9f * m.z + 0.1f * gr.z; m.w = 0.9f * m.w + 0.1f * gr.w;
    v.x = 0.99f * v.x + 0.01f * gr.x * gr.x; v.y = 0.99f * v.y + 0.01f * gr.y * gr.y;
    v.z = 0.99f * v.z + 0.01f * gr.z * gr.z; v.w = 0.99f * v.w + 0.01f * gr.w * gr.w;
    p.x -= 1e-3f * m.x / (sqrtf(v.x) + 1e-7f); p.y -= 1e-3f * m.y / (sqrtf(v.y) + 1e-7f);
    p.z -= 1e-3f * m.z / (sqrtf(v.z) + 1e-7f); p.w -= 1e-3f * m.w / (sqrtf(v.w) + 1e-7f);
    table[row * 4 + c] = p; mv[row * 8 + c] = m; mv[row * 8 + 4 + c] = v;
}

// the same update on ONE interleaved record per row: p | m | v = 192 contiguous bytes (one random location, not two)
__global__ __launch_bounds__(256) void k_adamlike3(float4* __restrict__ pmv, const int* __restrict__ rows,
                                                   const float4* __restrict__ grad, int64_t n) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    if (g >= n) return;
    const int64_t row = rows[g];
    float4 gr = grad[g * 4 + c];
    float4 p = pmv[row * 12 + c], m = pmv[row * 12 + 4 + c], v = pmv[row * 12 + 8 + c];
    m.x = 0.9f * m.x + 0.1f * gr.x; m.y = 0.9f * m.y + 0.1f * gr.y; m.z = 0.9f * m.z + 0.1f * gr.z; m.w = 0.9f * m.w + 0.1f * gr.w;
    v.x = 0.99f * v.x + 0.01f * gr.x * gr.x; v.y = 0.99f * v.y + 0.01f * gr.y * gr.y;
    v.z = 0.99f * v.z + 0.01f * gr.z * gr.z; v.w = 0.99f * v.w + 0.01f * gr.w * gr.w;
    p.x -= 1e-3f * m.x / (sqrtf(v.x) + 1e-7f); p.y -= 1e-3f * m.y / (sqrtf(v.y) + 1e-7f);
    p.z -= 1e-3f * m.z / (sqrtf(v.z) + 1e-7f); p.w -= 1e-3f * m.w / (sqrtf(v.w) + 1e-7f);
    pmv[row * 12 + c] = p; pmv[row * 12 + 4 + c] = m; pmv[row * 12 + 8 + c] = v;
}

// gather of the 64-byte p part out of 192-byte records (what the forward gather would see with that layout)
__global__ __launch_bounds__(256) void k_gather3(const float4* __restrict__ pmv, const int* __restrict__ rows,
                                                 int64_t n, float4* __restrict__ out) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    if (g >= n) return;
    out[g * 4 + c] = pmv[(int64_t)rows[g] * 12 + c];
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}

// launch(i) must use a FRESH set of rows per repetition: 213K rows x 192 B = 41 MB would otherwise sit in the 256 MB
// Infinity Cache after the first pass and the numbers would be cache numbers, not HBM numbers
template <class F>
static float time_us(F launch, int reps = 16) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main() {
    const int64_t V = 26LL * 1000 * 1000;   // table rows of 64 bytes: 1.66 GB
    float4 *table, *mv, *out, *grad;
    CK(hipMalloc(&table, V * 64)); CK(hipMalloc(&mv, V * 192));
    CK(hipMemset(table, 0, V * 64)); CK(hipMemset(mv, 0, V * 192));
    const int64_t NMAX = 1 << 24;
    CK(hipMalloc(&out, NMAX * 64)); CK(hipMalloc(&grad, NMAX * 64));
    CK(hipMemset(grad, 0, NMAX * 64));
    std::vector<int> h(NMAX);
    uint64_t s = 88172645463325252ULL;
    for (int64_t i = 0; i < NMAX; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (uint64_t)V); }
    int* rows;
    CK(hipMalloc(&rows, NMAX * 4));
    CK(hipMemcpy(rows, h.data(), NMAX * 4, hipMemcpyHostToDevice));
    printf("random 64-byte rows of a %.2f GB table (one launch; time includes the launch gap of back-to-back launches)\n", V * 64 / 1e9);
    for (int64_t n : {212992LL, 1LL << 20}) {
        const unsigned blocks1 = (unsigned)((n * 4 + 255) / 256);
        float t1 = time_us([&](int i) { hipLaunchKernelGGL(k_gather<1>, dim3(blocks1), dim3(256), 0, 0, table, rows + i * n, n, out); });
        float t2 = time_us([&](int i) { hipLaunchKernelGGL(k_gather<2>, dim3((blocks1 + 1) / 2), dim3(256), 0, 0, table, rows + i * n, n, out); });
        float t4 = time_us([&](int i) { hipLaunchKernelGGL(k_gather<4>, dim3((blocks1 + 3) / 4), dim3(256), 0, 0, table, rows + i * n, n, out); });
        float tr = time_us([&](int i) { hipLaunchKernelGGL(k_rmw, dim3(blocks1), dim3(256), 0, 0, table, rows + i * n, n); });
        float ta = time_us([&](int i) { hipLaunchKernelGGL(k_adamlike, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + i * n, grad, n); });
        float ta3 = time_us([&](int i) { hipLaunchKernelGGL(k_adamlike3, dim3(blocks1), dim3(256), 0, 0, mv, rows + i * n, grad, n); });
        float tg3 = time_us([&](int i) { hipLaunchKernelGGL(k_gather3, dim3(blocks1), dim3(256), 0, 0, mv, rows + i * n, n, out); });
        printf("n=%8lld  interleaved p|m|v records (192 B): adam-like %7.1f us, gather of the p part %7.1f us\n", (long long)n, ta3, tg3);
        printf("n=%8lld  gather R=1 %7.1f us (%5.2f TB/s gathered+written)  R=2 %7.1f us  R=4 %7.1f us | rmw %7.1f us (%5.2f TB/s r+w) | adam-like %7.1f us (%5.2f TB/s)\n",
               (long long)n, t1, n * 128 / t1 / 1e6, t2, t4, tr, n * 128 / tr / 1e6, ta, n * 448 / ta / 1e6);
    }
    for (int64_t bytes : {28LL << 20, 256LL << 20}) {
        float tc = time_us([&](int i) { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, table + (int64_t)i * (bytes / 16), (float4*)mv + (int64_t)i * (bytes / 16), bytes / 16); }, 4);
        printf("streaming copy of %lld MB: %7.1f us (%5.2f TB/s read+write)\n", (long long)(bytes >> 20), tc, 2.0 * bytes / tc / 1e6);
    }
    return 0;
}
