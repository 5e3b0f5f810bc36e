// Deterministic float32 -> float64 reductions (sum, sum of exp) used for the scalar results
// (means of per-sample log-densities, kernel-matrix sums).  HBM-bound: 4 bytes per element.
#include "lvae_common.h"

#define RED_BLOCKS 1024        // partials fit LVAE_REDUCE_WS_BYTES (8 KiB)

// mode 0: sum x ; mode 1: sum exp(x).  VEC: 16-byte loads, four independent ones in flight per lane (x must be
// 16-byte aligned; the last n % 4 elements are added by block 0).  The summation order only depends on n and
// the grid, so the result is reproducible.
template <int MODE, bool VEC>
__global__ void __launch_bounds__(256) reduce_stage1(const float* __restrict__ x, int64_t n,
                                                     double* __restrict__ part) {
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto term = [](float v) -> double { return MODE == 1 ? exp((double)v) : (double)v; };
    if (VEC) {
        const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
        const int64_t n4 = n >> 2;
        int64_t i = tid;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
            acc += (term(a[0]) + term(a[1])) + (term(a[2]) + term(a[3]));
            acc += (term(b[0]) + term(b[1])) + (term(b[2]) + term(b[3]));
            acc += (term(c[0]) + term(c[1])) + (term(c[2]) + term(c[3]));
            acc += (term(d[0]) + term(d[1])) + (term(d[2]) + term(d[3]));
        }
        for (; i < n4; i += stride) {
            const f32x4 a = x4[i];
            acc += (term(a[0]) + term(a[1])) + (term(a[2]) + term(a[3]));
        }
        if (tid < (n & 3)) acc += term(x[(n4 << 2) + tid]);
    } else {
        for (int64_t i = tid; i < n; i += stride) acc += term(x[i]);
    }
    acc = wave_sum_d(acc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void __launch_bounds__(256) reduce_stage2(const double* __restrict__ part, int nparts,
                                                     double scale, double* __restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    acc = wave_sum_d(acc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (sm[0] + sm[1] + sm[2] + sm[3]) * scale;
}

int lvae_launch_reduce(int mode, const float* x, int64_t n, double scale, double* out, double* part,
                       hipStream_t st) {
    int blocks = (int)lvae_ceil_div(n, 256 * 16);
    if (blocks < 1) blocks = 1;
    if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
    const bool vec = ((uintptr_t)x & 15) == 0 && n >= 1024;
#define LVAE_RED(M, V) hipLaunchKernelGGL((reduce_stage1<M, V>), dim3(blocks), dim3(256), 0, st, x, n, part)
    if (mode == 1) { if (vec) LVAE_RED(1, true); else LVAE_RED(1, false); }
    else { if (vec) LVAE_RED(0, true); else LVAE_RED(0, false); }
#undef LVAE_RED
    LVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, st, part, blocks, scale, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
