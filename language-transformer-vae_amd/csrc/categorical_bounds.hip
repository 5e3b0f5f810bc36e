// InfoNCE lower / MBU upper bound with Categorical critics: calc_upper_and_lower_bound_generative_mi
// (mutual_information.py:251-278), i.e. mi_lower_bound (:135-173) and mi_upper_bound (:176-215) on
// P[j, i] = Categorical(probs = row j).log_prob(sample_i).  torch's Categorical normalises the row
// (probs / probs.sum(-1)) and takes log(clamp(p, eps, 1 - eps)) with eps = 2^-23, so
//   P[j, i] = log(clamp(probs[j, s_i] / rowsum_j, eps, 1 - eps)).
// Bandwidth-bound: one coalesced pass over [K, V] for the row sums, then a K x K gather.
#include "lvae_common.h"

#define CB_EPS 1.1920928955078125e-07f

__global__ void __launch_bounds__(256)
cat_rowsum_kernel(const float* __restrict__ probs, int64_t V, float* __restrict__ rowsum) {
    const float* row = probs + (int64_t)blockIdx.x * V;
    float acc = 0.f;
    for (int64_t v = threadIdx.x; v < V; v += 256) acc += row[v];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rowsum[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// One workgroup per sample i: logsumexp over the K distributions of P[j, i], with and without j == i.
__global__ void __launch_bounds__(256)
cat_column_kernel(const float* __restrict__ probs, const int64_t* __restrict__ samples,
                  const float* __restrict__ rowsum, int64_t K, int64_t V, float* __restrict__ terms) {
    const int64_t i = blockIdx.x;
    const int64_t s = samples[i];
    __shared__ float red[4];
    __shared__ float bcast;
    float mx = -INFINITY;
    for (int64_t j = threadIdx.x; j < K; j += 256) {
        const float p = fminf(fmaxf(probs[j * V + s] / rowsum[j], CB_EPS), 1.0f - CB_EPS);
        mx = fmaxf(mx, logf(p));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bcast = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    mx = bcast;
    float sum_all = 0.f, sum_off = 0.f, diag = 0.f;
    for (int64_t j = threadIdx.x; j < K; j += 256) {
        const float lp = logf(fminf(fmaxf(probs[j * V + s] / rowsum[j], CB_EPS), 1.0f - CB_EPS));
        const float e = expf(lp - mx);
        sum_all += e;
        if (j == i) diag = lp; else sum_off += e;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sum_all += __shfl_xor(sum_all, o, 64);
        sum_off += __shfl_xor(sum_off, o, 64);
        diag += __shfl_xor(diag, o, 64);
    }
    __shared__ float r3[3][4];
    if ((threadIdx.x & 63) == 0) {
        r3[0][threadIdx.x >> 6] = sum_all; r3[1][threadIdx.x >> 6] = sum_off; r3[2][threadIdx.x >> 6] = diag;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sa = (r3[0][0] + r3[0][1]) + (r3[0][2] + r3[0][3]);
        const float so = (r3[1][0] + r3[1][1]) + (r3[1][2] + r3[1][3]);
        const float dg = (r3[2][0] + r3[2][1]) + (r3[2][2] + r3[2][3]);
        terms[i] = dg - (mx + logf(sa) - logf((float)K));                // InfoNCE term (:165-171)
        terms[K + i] = dg - (mx + logf(so) - logf((float)(K - 1)));      // MBU term (:204-213)
    }
}

__global__ void __launch_bounds__(256)
cat_mean_kernel(const float* __restrict__ terms, int64_t K, double* __restrict__ out) {
    double lo = 0.0, up = 0.0;
    for (int64_t i = threadIdx.x; i < K; i += 256) { lo += terms[i]; up += terms[K + i]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { lo += __shfl_xor(lo, o, 64); up += __shfl_xor(up, o, 64); }
    __shared__ double sm[2][4];
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = lo; sm[1][threadIdx.x >> 6] = up; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = ((sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3])) / (double)K;
        out[1] = ((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3])) / (double)K;
    }
}

// samples must lie in [0, V): checked on the device before the gather (an out-of-range index would read
// outside probs); a bad index sets out = NaN and the gather is skipped for that column.
__global__ void __launch_bounds__(256)
cat_check_samples_kernel(const int64_t* __restrict__ samples, int64_t K, int64_t V, int64_t* __restrict__ clean,
                         int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const int64_t s = samples[i];
    const bool ok = s >= 0 && s < V;
    clean[i] = ok ? s : 0;
    if (!ok) atomicOr(bad, 1);
}

__global__ void cat_poison_kernel(const int* __restrict__ bad, double* __restrict__ out) {
    if (*bad) { out[0] = __longlong_as_double(0x7ff8000000000000LL); out[1] = out[0]; }
}

size_t lvae_cat_bounds_workspace(int64_t K) {
    return (size_t)256 + (((size_t)K * sizeof(int64_t) + 255) & ~(size_t)255) + (size_t)K * 3 * sizeof(float);
}

int lvae_launch_cat_bounds(const float* probs, const int64_t* samples, int64_t K, int64_t V, double* out, void* ws,
                           hipStream_t st) {
    char* base = (char*)ws;
    int* bad = (int*)base;
    int64_t* clean = (int64_t*)(base + 256);
    float* rowsum = (float*)(base + 256 + (((size_t)K * sizeof(int64_t) + 255) & ~(size_t)255));
    float* terms = rowsum + K;
    LVAE_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
    hipLaunchKernelGGL(cat_check_samples_kernel, dim3((unsigned)lvae_ceil_div(K, 256)), dim3(256), 0, st, samples, K,
                       V, clean, bad);
    hipLaunchKernelGGL(cat_rowsum_kernel, dim3((unsigned)K), dim3(256), 0, st, probs, V, rowsum);
    hipLaunchKernelGGL(cat_column_kernel, dim3((unsigned)K), dim3(256), 0, st, probs, clean, rowsum, K, V, terms);
    hipLaunchKernelGGL(cat_mean_kernel, dim3(1), dim3(256), 0, st, terms, K, out);
    hipLaunchKernelGGL(cat_poison_kernel, dim3(1), dim3(1), 0, st, bad, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
