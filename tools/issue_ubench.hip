// micro-benchmark (gfx950): what one wave per SIMD pays, in shader cycles (s_memtime), for the instruction mixes the fused
// DeepFM tower is built from.  One 256-thread block per CU (256 blocks), every number = cycles per instruction of
// wave 0, averaged over blocks.   hipcc --offload-arch=gfx950 -O3 -o tools/ub_issue tools/issue_ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define T0() const unsigned long long t0 = __builtin_amdgcn_s_memtime()
#define T1(n)                                                                            \
    do {                                                                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                      \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = (float)(t1 - t0) / (float)(n); \
    } while (0)

// ---- MFMA issue cadence ----
template <int CH>
__global__ __launch_bounds__(256) void k_mfma16(float* cyc, float* sink, const float* in) {
    floatx4 acc[CH];
    for (int a = 0; a < CH; ++a) acc[a] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
    T0();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[u % CH], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < CH; ++a) s += acc[a][0] + acc[a][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    T1(64 * 16);
}
template <int CH>
__global__ __launch_bounds__(256) void k_mfma32(float* cyc, float* sink, const float* in) {
    floatx16 acc[CH];
    for (int a = 0; a < CH; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float av = in[threadIdx.x], bv = in[threadIdx.x + 256];
    T0();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u % CH], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < CH; ++a) s += acc[a][0] + acc[a][15];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    T1(64 * 16);
}
// ---- MFMA 32x32x2 (2 chains) with one LDS read / one L2 load per GROUP MFMAs, issued DIST groups ahead ----
template <int KIND, int GROUP>      // KIND 0: ds_read_b128, 1: global_load_dwordx4 (coalesced, L2 resident), 2: both
__global__ __launch_bounds__(512) void k_mix(float* cyc, float* sink, const float* in, const floatx4* big) {
    __shared__ floatx4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = floatx4{in[i], 0.f, 0.f, 0.f};
    __syncthreads();
    floatx16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    const floatx4* gp = big + (threadIdx.x >> 6) * 64 + (threadIdx.x & 63);
    floatx4 a0 = lds[threadIdx.x], a1 = lds[(threadIdx.x + 256) & 2047], b0 = gp[0], b1 = gp[256];
    T0();
    for (int it = 0; it < 128; ++it) {
        floatx4 an = a0, bn = b0;
        if (KIND == 0 || KIND == 2) an = lds[(threadIdx.x + 64 * it) & 2047];
        if (KIND == 1 || KIND == 2) bn = gp[((it + 2) & 255) * 256];
#pragma unroll
        for (int u = 0; u < GROUP; u += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u & 3], b0[u & 3], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[(u + 1) & 3], b0[(u + 1) & 3], acc2, 0, 0, 0);
        }
        a0 = a1; a1 = an; b0 = b1; b1 = bn;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc2[0];
    T1(128 * GROUP);
}
// ---- memory instruction issue cost, 4 waves per CU all doing the same ----
template <int KIND>   // 0: coalesced float4 loads (1 KiB per wave-instruction, L2 resident), 1: row-strided float4 loads (64 lines),
                      // 2: coalesced float4 stores, 3: dword stores as 4 x 64 B segments, 4: dword loads 2 x 128 B
__global__ __launch_bounds__(256) void k_mem(float* cyc, float* sink, float* buf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = buf + (size_t)(blockIdx.x % 64) * 65536 + wave * 16384;     // 256 KiB per block slot, 64 slots (L2 resident)
    floatx4 s4 = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f;
    T0();
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        if (KIND == 0) s4 += *reinterpret_cast<floatx4*>(base + u * 256 + lane * 4);
        if (KIND == 1) s4 += *reinterpret_cast<floatx4*>(base + lane * 256 + u * 4);
        if (KIND == 2) *reinterpret_cast<floatx4*>(base + u * 256 + lane * 4) = floatx4{1.f, 2.f, 3.f, (float)u};
        if (KIND == 3) base[(lane >> 4) * 2048 + u * 16 + (lane & 15)] = (float)u;
        if (KIND == 4) s1 += base[(lane >> 5) * 2048 + u * 32 + (lane & 31)];
    }
    T1(32);
    sink[blockIdx.x * 256 + threadIdx.x] = s4[0] + s4[1] + s4[2] + s4[3] + s1;
}

static float* d_cyc; static float* d_sink; static float* d_in; static float* d_big;
template <typename F>
static void report(const char* name, F launch) {
    launch();
    hipDeviceSynchronize();
    launch();
    hipDeviceSynchronize();
    std::vector<float> h(256 * 8);
    hipMemcpy(h.data(), d_cyc, 256 * 8 * 4, hipMemcpyDeviceToHost);
    double s = 0, s0 = 0;
    for (int b = 0; b < 256; ++b) {          // slowest wave of every block (and wave 0 alone), averaged over blocks
        float mx = 0;
        for (int w = 0; w < 8; ++w) mx = h[b * 8 + w] > mx ? h[b * 8 + w] : mx;
        s += mx; s0 += h[b * 8];
    }
    printf("%-100s slowest wave %8.1f   wave 0 %8.1f cycles/instr\n", name, s / 256, s0 / 256);
    hipMemset(d_cyc, 0, 256 * 8 * 4);
}

int main() {
    hipMalloc(&d_cyc, 256 * 8 * 4); hipMemset(d_cyc, 0, 256 * 8 * 4); hipMalloc(&d_sink, 256 * 256 * 4); hipMalloc(&d_in, 8192 * 4);
    hipMalloc(&d_big, (size_t)64 * 65536 * 4);
    hipMemset(d_in, 0, 8192 * 4); hipMemset(d_big, 0, (size_t)64 * 65536 * 4);
    dim3 g(256), b(256);
#define R(name, K, ...) report(name, [&] { hipLaunchKernelGGL(K, g, b, 0, 0, __VA_ARGS__); })
    R("mfma 16x16x4 f32, 1 chain", (k_mfma16<1>), d_cyc, d_sink, d_in);
    R("mfma 16x16x4 f32, 2 chains", (k_mfma16<2>), d_cyc, d_sink, d_in);
    R("mfma 16x16x4 f32, 4 chains", (k_mfma16<4>), d_cyc, d_sink, d_in);
    R("mfma 32x32x2 f32, 1 chain", (k_mfma32<1>), d_cyc, d_sink, d_in);
    R("mfma 32x32x2 f32, 2 chains", (k_mfma32<2>), d_cyc, d_sink, d_in);
    R("mfma 32x32x2 x4 + one ds_read_b128 per 4", (k_mix<0, 4>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
    R("mfma 32x32x2 x4 + one coalesced L2 float4 load per 4", (k_mix<1, 4>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
    R("mfma 32x32x2 x4 + both per 4", (k_mix<2, 4>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
    R("mfma 32x32x2 x2 + both per 2", (k_mix<2, 2>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
#define R8(name, K, ...) report(name, [&] { hipLaunchKernelGGL(K, g, dim3(512), 0, 0, __VA_ARGS__); })
    R8("8 waves/CU: mfma 32x32x2 x4 + one ds_read_b128 per 4 (cycles per MFMA of ONE wave; 2 share a SIMD)", (k_mix<0, 4>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
    R8("8 waves/CU: mfma 32x32x2 x4 + one L2 float4 load per 4", (k_mix<1, 4>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
    R8("8 waves/CU: mfma 32x32x2 x4 + both per 4", (k_mix<2, 4>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
    R8("8 waves/CU: mfma 32x32x2 x2 + both per 2", (k_mix<2, 2>), d_cyc, d_sink, d_in, (const floatx4*)d_big);
    R("32 coalesced float4 loads (1 KiB / wave-instr), 4 waves/CU", (k_mem<0>), d_cyc, d_sink, d_big);
    R("32 row-strided float4 loads (64 lines / wave-instr), 4 waves/CU", (k_mem<1>), d_cyc, d_sink, d_big);
    R("32 coalesced float4 stores, 4 waves/CU", (k_mem<2>), d_cyc, d_sink, d_big);
    R("32 dword stores as 4 x 64 B segments, 4 waves/CU", (k_mem<3>), d_cyc, d_sink, d_big);
    R("32 dword loads as 2 x 128 B segments, 4 waves/CU", (k_mem<4>), d_cyc, d_sink, d_big);
    return 0;
}
