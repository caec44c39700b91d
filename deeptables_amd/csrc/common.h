// common.h — shared helpers for the gfx950 kernels of libdt_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/dt_hip.h"

namespace dt {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return DT_ERR_LAUNCH;
    }
    return DT_OK;
}

#define DT_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            dt::set_error(__VA_ARGS__);       \
            return DT_ERR_INVALID_ARG;        \
        }                                     \
    } while (0)

#define DT_UNSUPPORTED(cond, ...)             \
    do {                                      \
        if (cond) {                           \
            dt::set_error(__VA_ARGS__);       \
            return DT_ERR_UNSUPPORTED;        \
        }                                     \
    } while (0)

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- wave-level reductions (64 lanes, butterfly via DPP-lowered shuffles) -------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the lanes that share (lane % group), i.e. strides group, 2*group, ... 32
template <int GROUP>
__device__ __forceinline__ float wave_sum_strided(float v) {
#pragma unroll
    for (int o = 32; o >= GROUP; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum inside aligned groups of GROUP consecutive lanes
template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() lowers to
// `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`, i.e. it also drains every global load in flight; this form waits
// for the LDS traffic alone, so operand prefetches issued before the barrier stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// multiplicative hash of a packed table row id into `32 - shift` bits (row-dedupe hash tables)
__device__ __forceinline__ unsigned hash_row(unsigned row, int shift) { return (row * 0x9E3779B1u) >> shift; }

// categorical id -> int, reproducing keras.ops.cast(float32 -> int32) (truncation toward zero)
template <int KIND>
__device__ __forceinline__ int load_id(const void* idx, int64_t i) {
    if (KIND == DT_IDX_F32) return (int)(reinterpret_cast<const float*>(idx)[i]);
    return reinterpret_cast<const int32_t*>(idx)[i];
}

}  // namespace dt
