// micro-benchmark: cost of v_mfma_f32_32x32x2_f32 chains (gfx950).  hipcc --offload-arch=gfx950 -O3 -o ub mfma_ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSOP>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* in) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
    const float* lp = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float a = av, b = bv;
            if (LDSOP) { a = lp[(it * 16 + u) & 1023]; b = lp[((it * 16 + u) & 1023) + 1024]; }
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDSOP>
void run(const char* name, int wpb_blocks, float* out, float* in) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDSOP>), dim3(wpb_blocks), dim3(256), 0, 0, out, iters, in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDSOP>), dim3(wpb_blocks), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * 16;
    const double ns_per = ms * 1e6 / mfma_per_wave;
    const double tflops = (double)wpb_blocks * 4 * mfma_per_wave * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
    printf("%-34s blocks=%4d  %.1f ns per MFMA per wave (%.0f cycles @2.1GHz)  %.1f TFLOP/s\n", name, wpb_blocks, ns_per,
           ns_per * 2.1, tflops);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 8192 * 4);
    hipMemset(in, 0, 8192 * 4);
    run<1, false>("1 acc, reg operands, 1 wave/SIMD", 256, out, in);
    run<2, false>("2 acc, reg operands, 1 wave/SIMD", 256, out, in);
    run<4, false>("4 acc, reg operands, 1 wave/SIMD", 256, out, in);
    run<1, true>("1 acc, LDS operands, 1 wave/SIMD", 256, out, in);
    run<2, true>("2 acc, LDS operands, 1 wave/SIMD", 256, out, in);
    run<4, true>("4 acc, LDS operands, 1 wave/SIMD", 256, out, in);
    run<1, false>("1 acc, reg operands, 2 waves/SIMD", 512, out, in);
    run<1, true>("1 acc, LDS operands, 2 waves/SIMD", 512, out, in);
    run<1, true>("1 acc, LDS operands, 4 waves/SIMD", 1024, out, in);
    return 0;
}
