// Deterministic float32 -> float64 reductions (sum, sum of exp) used for the scalar results
// (means of per-sample log-densities, kernel-matrix sums).  HBM-bound: 4 bytes per element.
#include "lvae_common.h"

#define RED_BLOCKS 1024

// mode 0: sum x ; mode 1: sum exp(x)
template <int MODE>
__global__ void __launch_bounds__(256) reduce_stage1(const float* __restrict__ x, int64_t n,
                                                     double* __restrict__ part) {
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        acc += MODE == 1 ? exp((double)v) : (double)v;
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
    int blocks = (int)lvae_ceil_div(n, 256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
    if (mode == 1)
        hipLaunchKernelGGL(reduce_stage1<1>, dim3(blocks), dim3(256), 0, st, x, n, part);
    else
        hipLaunchKernelGGL(reduce_stage1<0>, dim3(blocks), dim3(256), 0, st, x, n, part);
    LVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, st, part, blocks, scale, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
