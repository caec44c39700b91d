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

// address of lr_t (the bias-corrected rate the CURRENT step uses) inside the device-resident Adam step state (optim.hip)
const float* adam_state_lr_t(const void* state);

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

// ---- wave-level reductions (64 lanes) without LDS -------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip per step; six dependent ones per wave_sum).  DPP
// moves cover the steps inside a 16-lane row, v_permlane16_swap / v_permlane32_swap (gfx950) the two across rows.
#define DT_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (ctrl), 0xf, 0xf, true))
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7 - i inside groups of 8
constexpr int kDppMirror = 0x140;       // lane i <-> 15 - i inside rows of 16

__device__ __forceinline__ float row_pair16(float v, bool is_max) {   // combine across the 16-lane rows and the halves
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = is_max ? fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])) : __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return is_max ? fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1])) : __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_sum(float v) {
    v += DT_DPP_F(v, kDppXor1);
    v += DT_DPP_F(v, kDppXor2);
    v += DT_DPP_F(v, kDppHalfMirror);     // quads hold equal values: pairing quad 0 with quad 1 (mirrored) is enough
    v += DT_DPP_F(v, kDppMirror);
    return row_pair16(v, false);
}
// sum over the lanes that share (lane % group), i.e. strides group, 2*group, ... 32
template <int GROUP>
__device__ __forceinline__ float wave_sum_strided(float v) {
    if (GROUP <= 16) {
        // strides >= 16 first (exact lane pairing through the permlane swaps), then xor 8 / 4 / ... inside the row
        v = row_pair16(v, false);
#pragma unroll
        for (int o = 8; o >= GROUP; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
#pragma unroll
    for (int o = 32; o >= GROUP; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum inside aligned groups of GROUP consecutive lanes
template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
    if (GROUP >= 2) v += DT_DPP_F(v, kDppXor1);
    if (GROUP >= 4) v += DT_DPP_F(v, kDppXor2);
    if (GROUP >= 8) v += DT_DPP_F(v, kDppHalfMirror);
    if (GROUP >= 16) v += DT_DPP_F(v, kDppMirror);
    if (GROUP == 32) v += __shfl_xor(v, 16, 64);
    if (GROUP == 64) v = row_pair16(v, false);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, DT_DPP_F(v, kDppXor1));
    v = fmaxf(v, DT_DPP_F(v, kDppXor2));
    v = fmaxf(v, DT_DPP_F(v, kDppHalfMirror));
    v = fmaxf(v, DT_DPP_F(v, kDppMirror));
    return row_pair16(v, true);
}

// ---- keras.activations fused by the CIN / AFM kernels (codes: include/dt_hip.h) ----------------------------------
constexpr float kSeluAlpha = 1.6732632423543772f, kSeluScale = 1.0507009873554805f;
__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case DT_ACT_RELU: return fmaxf(v, 0.f);
        case DT_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case DT_ACT_TANH: return tanhf(v);
        case DT_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case DT_ACT_SELU: return kSeluScale * (v > 0.f ? v : kSeluAlpha * expm1f(v));
        case DT_ACT_SOFTPLUS: return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
        case DT_ACT_SOFTSIGN: return v / (1.0f + fabsf(v));
        case DT_ACT_EXPONENTIAL: return expf(v);
        default: return v;
    }
}
// d act / d pre-activation as a function of the OUTPUT y = act(v)
__device__ __forceinline__ float act_grad_from_y(float y, int act) {
    switch (act) {
        case DT_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case DT_ACT_SIGMOID: return y * (1.f - y);
        case DT_ACT_TANH: return 1.f - y * y;
        case DT_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        case DT_ACT_SELU: return y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha;
        case DT_ACT_SOFTPLUS: return 1.f - expf(-y);
        case DT_ACT_SOFTSIGN: { const float t = 1.f - fabsf(y); return t * t; }
        case DT_ACT_EXPONENTIAL: return y;
        default: return 1.f;
    }
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
