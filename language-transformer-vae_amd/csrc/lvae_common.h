// Shared helpers for the lvae HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/lvae.h"

#define LVAE_LOG2E 1.4426950408889634f
#define LVAE_LN2 0.6931471805599453
#define LVAE_LOG_2PI 1.8378770664093453f

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// float -> unsigned key with the same ordering (for atomicMax on signed floats) and back
__device__ __forceinline__ unsigned lvae_float_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float lvae_key_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (lvae_api.hip owns the storage) -------------------------------------------
int lvae_set_error(int status, const char* fmt, ...);

#define LVAE_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return lvae_set_error(LVAE_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define LVAE_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return lvae_set_error(LVAE_ERR_HIP, "%s failed: %s (%s:%d)", #call,                \
                                  hipGetErrorString(e__), __FILE__, __LINE__);                 \
    } while (0)

#define LVAE_LAUNCH_CHECK() LVAE_HIP(hipGetLastError())

static inline int64_t lvae_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t lvae_round_up(int64_t a, int64_t b) { return lvae_ceil_div(a, b) * b; }

// Padded contraction depth per half: the MFMA path supports D <= 128 by zero-padding D.
static inline int lvae_pad_dim(int D) {
    if (D <= 16) return 16;
    if (D <= 32) return 32;
    if (D <= 64) return 64;
    if (D <= 128) return 128;
    return 0;
}

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float lvae_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
