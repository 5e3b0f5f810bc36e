// Two-sample Kolmogorov-Smirnov statistic of two fp32 vectors (the IW log-likelihood vectors of the
// validation pass): what scipy.stats.ks_2samp returns as `statistic` at utils_train.py:496-497.
//
//   1. each sample is copied into a power-of-two buffer padded with +inf and sorted ascending by a bitonic
//      network: all stages whose partner distance fits a 4096-element tile run inside LDS (one global
//      read + write per group of stages), only the wider exchanges are separate HBM passes;
//   2. one thread per pooled value v counts #a <= v and #b <= v by binary search in the sorted buffers and
//      forms F_a(v) - F_b(v) in fp64; block-wide max / min, then one block folds the per-block extremes into
//      D = max(max_v(F_a - F_b), clip(-min_v(F_a - F_b), 0, 1))        (scipy's two-sided statistic).
// Counts are integers and the two divisions are fp64, so the statistic is bit-identical to scipy's.
#include <cmath>
#include "pair_plan.h"

#define KS_TILE 4096          // elements sorted per workgroup in LDS (16 KiB)
#define KS_THREADS 256

__global__ void __launch_bounds__(256)
ks_pad_copy_kernel(const float* __restrict__ src, int64_t n, int64_t npad, float* __restrict__ dst,
                   int* __restrict__ nan_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    float v = i < n ? src[i] : INFINITY;
    if (v != v) {                 // a NaN has no place in the order: flag it (the statistic becomes NaN, as in
        *nan_flag = 1;            // scipy) and keep the network's comparisons total
        v = INFINITY;
    }
    dst[i] = v;
}

__device__ __forceinline__ void ks_cmpswap(float& a, float& b, bool ascending) {
    const bool sw = ascending ? (a > b) : (a < b);
    if (sw) { const float t = a; a = b; b = t; }
}

// Runs, inside LDS, every compare-exchange whose partner distance is < KS_TILE for the stage range
// [k_first, k_last] (k = size of the bitonic sequences being merged; both powers of two).  For k > KS_TILE
// only the tail j = KS_TILE/2 .. 1 of that stage is done here (the wider exchanges ran as global passes).
__global__ void __launch_bounds__(KS_THREADS)
ks_bitonic_tile_kernel(float* __restrict__ data, int64_t k_first, int64_t k_last) {
    __shared__ float s[KS_TILE];
    const int64_t base = (int64_t)blockIdx.x * KS_TILE;
    for (int t = threadIdx.x; t < KS_TILE; t += KS_THREADS) s[t] = data[base + t];
    __syncthreads();
    for (int64_t k = k_first; k <= k_last; k <<= 1) {
        int j0 = (int)(k >> 1);
        if (j0 > KS_TILE / 2) j0 = KS_TILE / 2;
        for (int j = j0; j >= 1; j >>= 1) {
            for (int t = threadIdx.x; t < KS_TILE / 2; t += KS_THREADS) {
                const int lo = 2 * t - (t & (j - 1));            // index with bit j clear
                const bool asc = ((base + lo) & k) == 0;
                ks_cmpswap(s[lo], s[lo + j], asc);
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < KS_TILE; t += KS_THREADS) data[base + t] = s[t];
}

// One exchange of distance j >= KS_TILE of stage k, in global memory.
__global__ void __launch_bounds__(256)
ks_bitonic_global_kernel(float* __restrict__ data, int64_t npad, int64_t j, int64_t k) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= npad / 2) return;
    const int64_t lo = 2 * t - (t & (j - 1));
    float a = data[lo], b = data[lo + j];
    const bool asc = (lo & k) == 0;
    const bool sw = asc ? (a > b) : (a < b);
    if (sw) { data[lo] = b; data[lo + j] = a; }
}

__device__ __forceinline__ int64_t ks_upper_bound(const float* __restrict__ s, int64_t n, float v) {
    int64_t lo = 0, hi = n;                       // first index with s[idx] > v
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (s[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
ks_cdf_diff_kernel(const float* __restrict__ sa, int64_t n1, const float* __restrict__ sb, int64_t n2,
                   double* __restrict__ block_ext) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double dmax = -2.0, dmin = 2.0;
    if (i < n1 + n2) {
        const float v = i < n1 ? sa[i] : sb[i - n1];
        const double d = (double)ks_upper_bound(sa, n1, v) / (double)n1 -
                         (double)ks_upper_bound(sb, n2, v) / (double)n2;
        dmax = d; dmin = d;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
        dmin = fmin(dmin, __shfl_xor(dmin, o, 64));
    }
    __shared__ double smx[4], smn[4];
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = dmax; smn[threadIdx.x >> 6] = dmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_ext[2 * blockIdx.x] = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
        block_ext[2 * blockIdx.x + 1] = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
    }
}

__global__ void __launch_bounds__(256)
ks_finalize_kernel(const double* __restrict__ block_ext, int64_t nblocks, const int* __restrict__ nan_flag,
                   double* __restrict__ out) {
    double dmax = -2.0, dmin = 2.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += 256) {
        dmax = fmax(dmax, block_ext[2 * b]);
        dmin = fmin(dmin, block_ext[2 * b + 1]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
        dmin = fmin(dmin, __shfl_xor(dmin, o, 64));
    }
    __shared__ double smx[4], smn[4];
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = dmax; smn[threadIdx.x >> 6] = dmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mx = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
        const double mn = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
        const double min_s = fmin(fmax(-mn, 0.0), 1.0);
        const bool bad = *nan_flag != 0;
        out[0] = bad ? (double)NAN : fmax(min_s, mx);
        out[1] = bad ? (double)NAN : mx;           // D+  (max of F_a - F_b)
        out[2] = bad ? (double)NAN : min_s;        // D-
    }
}

// ---- host side -----------------------------------------------------------------------------------------
int64_t lvae_ks_padded(int64_t n) {
    int64_t p = KS_TILE;
    while (p < n) p <<= 1;
    return p;
}

size_t lvae_ks_workspace(int64_t n1, int64_t n2) {
    const int64_t blocks = lvae_ceil_div(n1 + n2, 256);
    return lvae_align256((size_t)lvae_ks_padded(n1) * sizeof(float)) +
           lvae_align256((size_t)lvae_ks_padded(n2) * sizeof(float)) + lvae_align256((size_t)blocks * 2 * sizeof(double)) +
           256;                   // (+ the NaN flag)
}

static int ks_sort(const float* src, int64_t n, float* buf, int* nan_flag, hipStream_t st) {
    const int64_t npad = lvae_ks_padded(n);
    hipLaunchKernelGGL(ks_pad_copy_kernel, dim3((unsigned)lvae_ceil_div(npad, 256)), dim3(256), 0, st, src, n, npad,
                       buf, nan_flag);
    const unsigned tiles = (unsigned)(npad / KS_TILE);
    hipLaunchKernelGGL(ks_bitonic_tile_kernel, dim3(tiles), dim3(KS_THREADS), 0, st, buf, (int64_t)2,
                       (int64_t)KS_TILE);
    for (int64_t k = 2 * KS_TILE; k <= npad; k <<= 1) {
        for (int64_t j = k >> 1; j >= KS_TILE; j >>= 1)
            hipLaunchKernelGGL(ks_bitonic_global_kernel, dim3((unsigned)lvae_ceil_div(npad / 2, 256)), dim3(256), 0,
                               st, buf, npad, j, k);
        hipLaunchKernelGGL(ks_bitonic_tile_kernel, dim3(tiles), dim3(KS_THREADS), 0, st, buf, k, k);
    }
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_ks_2samp(const float* a, int64_t n1, const float* b, int64_t n2, double* out, void* ws,
                         hipStream_t st) {
    char* base = (char*)ws;
    float* sa = (float*)base;
    float* sb = (float*)(base + lvae_align256((size_t)lvae_ks_padded(n1) * sizeof(float)));
    double* ext = (double*)((char*)sb + lvae_align256((size_t)lvae_ks_padded(n2) * sizeof(float)));
    const int64_t blocks = lvae_ceil_div(n1 + n2, 256);
    int* nan_flag = (int*)((char*)ext + lvae_align256((size_t)blocks * 2 * sizeof(double)));
    LVAE_HIP(hipMemsetAsync(nan_flag, 0, sizeof(int), st));
    int rc = ks_sort(a, n1, sa, nan_flag, st);
    if (rc) return rc;
    rc = ks_sort(b, n2, sb, nan_flag, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ks_cdf_diff_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sa, n1, sb, n2, ext);
    hipLaunchKernelGGL(ks_finalize_kernel, dim3(1), dim3(256), 0, st, ext, blocks, nan_flag, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
