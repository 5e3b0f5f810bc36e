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

// ---- development switches (environment variables), read ONCE per process ---------------------------------------
// Sweeps and tests that change the environment mid-process call lvae_dev_reload_env() afterwards (lvae_api.hip owns
// the storage); product calls never touch getenv.
struct LvaeSwitches {
    long long nsplit, dsplit;     // LVAE_NSPLIT / LVAE_DSPLIT: forced split counts of the plan (0: the plan decides)
    double small_pairs;           // LVAE_SMALL_PAIRS: threshold of the small path (< 0: built-in)
    bool no_small_path;           // LVAE_NO_SMALL_PATH
    int prep_rb;                  // LVAE_PREP_RB: blocks per workgroup of the operand prep (0: unset; set = separate launches)
    bool no_fused_tail;           // LVAE_NO_FUSED_TAIL
    bool no_fused_mmd;            // LVAE_NO_FUSED_MMD
    int bwd_path;                 // LVAE_BWD_PATH: 0 unset, 1 valu, 2 mfma
    int iw_pipe;                  // LVAE_IW_PIPE: -1 unset, 0 / 1
    int iw_u, iw_tpb;             // LVAE_IW_U / LVAE_IW_TPB (0: unset)
    bool iw_no_stage;             // LVAE_IW_NO_STAGE
    bool iw_plain_stores;         // LVAE_IW_PLAIN_STORES: the staged [B,K] rows leave as plain stores (default: write-through, sc1)
    long long mmd_bx_min_pairs;   // LVAE_MMD_BX_MIN_PAIRS: smallest n1*n2 at which AUTO takes the split-bf16 kernel sums (< 0: built-in)
};
const LvaeSwitches& lvae_switches();

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
