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

// round 3: does reading a step's p / [m|v] rows EARLIER (while the matrix pipes work) make the later Adam-shaped
// read-modify-write cheaper?  k_touch reads the rows (all loads in flight, nothing stored unless the impossible happens)
template <bool WITH_P>
__global__ __launch_bounds__(256) void k_touch(const float4* __restrict__ table, const float4* __restrict__ mv,
                                               const int* __restrict__ rows, int64_t n, float4* __restrict__ out) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    if (g >= n) return;
    const int64_t row = rows[g];
    float4 m = mv[row * 8 + c], v = mv[row * 8 + 4 + c];
    float s = m.x + v.x;
    if (WITH_P) s += table[row * 4 + c].x;
    if (s == 12345.678f) out[0] = m;
}
// the write half alone: p (64 B) and [m|v] (128 B) stored at random rows, nothing read from the table
__global__ __launch_bounds__(256) void k_store_rows(float4* __restrict__ table, float4* __restrict__ mv,
                                                    const int* __restrict__ rows, const float4* __restrict__ grad, int64_t n) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int c = threadIdx.x & 3;
    if (g >= n) return;
    const int64_t row = rows[g];
    const float4 gr = grad[g * 4 + c];
    table[row * 4 + c] = gr; mv[row * 8 + c] = gr; mv[row * 8 + 4 + c] = gr;
}

// [m|v] read as ONE 128-byte request per row (8 lanes x 16 B in one instruction) instead of two 64-byte halves
__global__ __launch_bounds__(256) void k_touch128(const float4* __restrict__ mv, const int* __restrict__ rows, int64_t n,
                                                  float4* __restrict__ out) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int c = threadIdx.x & 7;
    if (g >= n) return;
    const float4 m = mv[(int64_t)rows[g] * 8 + c];
    if (m.x == 12345.678f) out[0] = m;
}
// Adam-shaped update with the slot record handled by 8 lanes (one 128-byte load + one 128-byte store per row) and p by 4
__global__ __launch_bounds__(256) void k_adamlike128(float4* __restrict__ table, float4* __restrict__ mv,
                                                     const int* __restrict__ rows, const float4* __restrict__ grad,
                                                     int64_t n) {
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int c = threadIdx.x & 7, c4 = c & 3;
    if (g >= n) return;
    const int64_t row = rows[g];
    const float4 gr = grad[g * 4 + c4];
    float4 s = mv[row * 8 + c];
    float4 p = table[row * 4 + c4];
    if (c < 4) { s.x = 0.9f * s.x + 0.1f * gr.x; s.y = 0.9f * s.y + 0.1f * gr.y; s.z = 0.9f * s.z + 0.1f * gr.z; s.w = 0.9f * s.w + 0.1f * gr.w; }
    else { s.x = 0.99f * s.x + 0.01f * gr.x * gr.x; s.y = 0.99f * s.y + 0.01f * gr.y * gr.y; s.z = 0.99f * s.z + 0.01f * gr.z * gr.z; s.w = 0.99f * s.w + 0.01f * gr.w * gr.w; }
    // lanes 0..3 hold m, lanes 4..7 v of the same 4 columns: exchange through the wave
    const float ox = __shfl_xor(s.x, 4, 64), oy = __shfl_xor(s.y, 4, 64), oz = __shfl_xor(s.z, 4, 64), ow = __shfl_xor(s.w, 4, 64);
    mv[row * 8 + c] = s;
    if (c < 4) {
        p.x -= 1e-3f * s.x / (sqrtf(ox) + 1e-7f); p.y -= 1e-3f * s.y / (sqrtf(oy) + 1e-7f);
        p.z -= 1e-3f * s.z / (sqrtf(oz) + 1e-7f); p.w -= 1e-3f * s.w / (sqrtf(ow) + 1e-7f);
        table[row * 4 + c4] = p;
    }
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
    {
        const int64_t n = 212992;
        const unsigned blocks1 = (unsigned)((n * 4 + 255) / 256);
        float tt = time_us([&](int i) { hipLaunchKernelGGL(k_touch<false>, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + i * n, n, out); });
        float ttp = time_us([&](int i) { hipLaunchKernelGGL(k_touch<true>, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (16 + i) * n, n, out); });
        float tw = time_us([&](int i) { hipLaunchKernelGGL(k_store_rows, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (32 + i) * n, grad, n); });
        // pairs on fresh rows: touch, then a streaming copy of 56 MB standing in for the step's traffic between the two, then the update
        float tpair = time_us([&](int i) {
            hipLaunchKernelGGL(k_touch<true>, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (48 + i) * n, n, out);
            hipLaunchKernelGGL(k_adamlike, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (48 + i) * n, grad, n); });
        float tpair2 = time_us([&](int i) {
            hipLaunchKernelGGL(k_touch<true>, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (20 + i) * n, n, out);
            hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, out, out + (56LL << 20) / 16, (56LL << 20) / 16);
            hipLaunchKernelGGL(k_adamlike, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (20 + i) * n, grad, n); });
        float tcp = time_us([&](int i) { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, out, out + (56LL << 20) / 16, (56LL << 20) / 16); });
        float tpair3 = time_us([&](int i) {
            hipLaunchKernelGGL(k_touch<false>, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (60 + i) * n, n, out);
            hipLaunchKernelGGL(k_adamlike, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + (60 + i) * n, grad, n); });
        float thot = time_us([&](int i) { hipLaunchKernelGGL(k_adamlike, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + 70 * n, grad, n); });
        float ttouchhot = time_us([&](int i) { hipLaunchKernelGGL(k_touch<true>, dim3(blocks1), dim3(256), 0, 0, table, mv, rows + 70 * n, n, out); });
        float t128 = time_us([&](int i) { hipLaunchKernelGGL(k_touch128, dim3(2 * blocks1), dim3(256), 0, 0, mv, rows + i * n, n, out); });
        float ta128 = time_us([&](int i) { hipLaunchKernelGGL(k_adamlike128, dim3(2 * blocks1), dim3(256), 0, 0, table, mv, rows + (16 + i) * n, grad, n); });
        printf("n=%8lld  the SAME rows every launch (cache numbers): adam-like %6.1f us, touch p + [m|v] %6.1f us | fresh rows, [m|v] as one 128-byte "
               "request per row: touch %6.1f us, adam-like %6.1f us\n", (long long)n, thot, ttouchhot, t128, ta128);
        printf("n=%8lld  touch [m|v] %6.1f us, touch p + [m|v] %6.1f us, store-only p + [m|v] %6.1f us\n", (long long)n, tt, ttp, tw);
        printf("n=%8lld  touch(p,m,v) + adam-like on the same fresh rows %6.1f us (adam-like after the touch: %6.1f us); with 112 MB of "
               "streaming traffic between them %6.1f us (copy alone %6.1f us -> adam-like %6.1f us); touch(m,v) + adam-like %6.1f us\n",
               (long long)n, tpair, tpair - ttp, tpair2, tcp, tpair2 - ttp - tcp, tpair3);
    }
    for (int64_t bytes : {28LL << 20, 256LL << 20}) {
        float tc = time_us([&](int i) { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, table + (int64_t)i * (bytes / 16), (float4*)mv + (int64_t)i * (bytes / 16), bytes / 16); }, 4);
        printf("streaming copy of %lld MB: %7.1f us (%5.2f TB/s read+write)\n", (long long)(bytes >> 20), tc, 2.0 * bytes / tc / 1e6);
    }
    return 0;
}
