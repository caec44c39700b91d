// micro-benchmark of the DeepFM GEMM1 inner loop variants (gfx950): A from LDS, B from global (L2-resident W).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int CH = 16, KP = 448, H1 = 128, XS = KP + 1;

// MODE 0: B from global, 3 register buffers (current kernel)   MODE 1: B from global, loads all issued per chunk
// just-in-time (no prefetch)   MODE 2: B staged through LDS by the block (cooperative, double buffered)
// MODE 3: B from global with 32-bit offsets from a uniform base (buffer-style addressing)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, float* out, int reps) {
    extern __shared__ float lds[];
    float* xn = lds;                 // [32][XS]
    float* wt = lds + 32 * XS;       // [2][32][H1]  (MODE 2)
    for (int i = threadIdx.x; i < 32 * XS; i += 256) xn[i] = 0.001f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane >> 5, c = lane & 31;
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* arow = xn + c * XS + s;
    const float* bcol = W + (int64_t)s * H1 + 32 * wave + c;
    const int nch = KP / (2 * CH);
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
            float a0[CH], b0[CH], a1[CH], b1[CH], a2[CH], b2[CH];
            auto load = [&](float (&aq)[CH], float (&bq)[CH], int ch) {
#pragma unroll
                for (int i = 0; i < CH; ++i) { aq[i] = arow[2 * CH * ch + 2 * i]; bq[i] = bcol[(int64_t)(2 * CH * ch + 2 * i) * H1]; }
            };
            auto run = [&](const float (&aq)[CH], const float (&bq)[CH]) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < CH; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i], bq[i], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            load(a0, b0, 0); load(a1, b1, 1);
            for (int ch = 0; ch < nch; ch += 3) {
                if (ch + 2 < nch) load(a2, b2, ch + 2);
                run(a0, b0);
                if (ch + 1 >= nch) break;
                if (ch + 3 < nch) load(a0, b0, ch + 3);
                run(a1, b1);
                if (ch + 2 >= nch) break;
                if (ch + 4 < nch) load(a1, b1, ch + 4);
                run(a2, b2);
            }
        } else if (MODE == 1) {
            for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * CH * ch + 2 * i], bcol[(int64_t)(2 * CH * ch + 2 * i) * H1], acc, 0, 0, 0);
            }
        } else if (MODE == 2) {
            // cooperative staging: chunk = 32 k-rows x 128 cols = 4096 floats = 4 float4 per thread
            float4 wreg[4];
            auto gload = [&](int ch) {
#pragma unroll
                for (int r = 0; r < 4; ++r) wreg[r] = *reinterpret_cast<const float4*>(W + (int64_t)ch * 32 * H1 + (threadIdx.x + 256 * r) * 4);
            };
            auto lstore = [&](int buf) {
#pragma unroll
                for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(wt + buf * 32 * H1 + (threadIdx.x + 256 * r) * 4) = wreg[r];
            };
            gload(0); lstore(0); __syncthreads();
            for (int ch = 0; ch < nch; ++ch) {
                const int buf = ch & 1;
                if (ch + 1 < nch) gload(ch + 1);
                const float* wb = wt + buf * 32 * H1 + s * H1 + 32 * wave + c;
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * CH * ch + 2 * i], wb[2 * i * H1], acc, 0, 0, 0);
                if (ch + 1 < nch) lstore(buf ^ 1);
                __syncthreads();
            }
        }
    }
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int MODE>
void run(const char* name, float* W, float* out) {
    const int reps = 20;
    const size_t ldsb = (32 * XS + 2 * 32 * H1) * 4;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), ldsb, 0, W, out, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), ldsb, 0, W, out, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.2f us per GEMM1 pass (224 MFMA steps per wave; floor ~7.3 us)\n", name, ms * 1e3 / reps);
}

int main() {
    float *W, *out;
    hipMalloc(&W, KP * H1 * 4 + 4096); hipMalloc(&out, 256 * 256 * 4);
    hipMemset(W, 0, KP * H1 * 4);
    run<0>("B global, 3 register buffers", W, out);
    run<1>("B global, just-in-time loads", W, out);
    run<2>("B staged through LDS (coop, 2 buffers)", W, out);
    return 0;
}
