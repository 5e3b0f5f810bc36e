// Backward of the element-wise estimators the reference differentiates in its training objectives:
// sample_log_likelihood (loss_and_optimisation.py:79-110) and kl_divergence (:14-43).  HBM-bound passes.
#include "pair_plan.h"

// d z[b,k,d] = -g * (z - mu) / var   (posterior branch)   or   -g * z   (prior branch, mu == NULL)
// g is per (b,k) (reduce_latent_dim=True) or per element.
__global__ void __launch_bounds__(256)
sample_ll_bwd_z_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                       const float* __restrict__ logvar, const float* __restrict__ g, int g_elem,
                       int64_t total, int64_t KD, int D, float* __restrict__ d_z) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const float gg = g_elem ? g[idx] : g[idx / D];
    float v = z[idx];
    if (mu) {
        const int64_t b = idx / KD;
        const int d = (int)(idx % D);
        v = (v - mu[b * D + d]) * expf(-logvar[b * D + d]);
    }
    d_z[idx] = -gg * v;
}

// d mu[b,d] = sum_k g (z - mu)/var ;  d logvar[b,d] = 0.5 sum_k g ((z - mu)^2/var - 1).  One thread per (b,d).
__global__ void __launch_bounds__(256)
sample_ll_bwd_params_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                            const float* __restrict__ logvar, const float* __restrict__ g, int g_elem,
                            int64_t B, int64_t K, int D, float* __restrict__ d_mu,
                            float* __restrict__ d_logvar) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * D) return;
    const int64_t b = idx / D;
    const int d = (int)(idx - b * D);
    const float m = mu[idx], iv = expf(-logvar[idx]);
    float am = 0.f, al = 0.f;
    for (int64_t k = 0; k < K; ++k) {
        const int64_t e = (b * K + k) * D + d;
        const float gg = g_elem ? g[e] : g[b * K + k];
        const float df = z[e] - m;
        am = fmaf(gg, df * iv, am);
        al = fmaf(gg, 0.5f * (df * df * iv - 1.0f), al);
    }
    if (d_mu) d_mu[idx] = am;
    if (d_logvar) d_logvar[idx] = al;
}

// kl = 0.5 (mu^2 + e^logvar - logvar - 1); hinge keeps kl where kl > lambda (lambda <= 0: hinge == kl).
// g_kl / g_hinge: upstream gradients, per element [N,D] (row_mode 0) or per row [N] (row_mode 1); may be NULL.
__global__ void __launch_bounds__(256)
kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ g_kl,
              const float* __restrict__ g_hinge, int row_mode, int64_t total, int D, float lambda,
              float* __restrict__ d_mu, float* __restrict__ d_logvar) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int64_t gi = row_mode ? idx / D : idx;
    const float m = mu[idx], lv = logvar[idx], ev = expf(lv);
    const float kl = 0.5f * (m * m + ev - lv - 1.0f);
    float w = g_kl ? g_kl[gi] : 0.f;
    if (g_hinge && (lambda <= 0.f || kl > lambda)) w += g_hinge[gi];
    if (d_mu) d_mu[idx] = w * m;
    if (d_logvar) d_logvar[idx] = w * 0.5f * (ev - 1.0f);
}

int lvae_launch_sample_ll_bwd(const float* z, const float* mu, const float* logvar, const float* g, int g_elem,
                              int64_t B, int64_t K, int D, float* d_z, float* d_mu, float* d_logvar,
                              hipStream_t st) {
    const int64_t total = B * K * D;
    if (d_z) {
        hipLaunchKernelGGL(sample_ll_bwd_z_kernel, dim3((unsigned)lvae_ceil_div(total, 256)), dim3(256), 0, st, z,
                           mu, logvar, g, g_elem, total, K * D, D, d_z);
        LVAE_LAUNCH_CHECK();
    }
    if (mu && (d_mu || d_logvar)) {
        hipLaunchKernelGGL(sample_ll_bwd_params_kernel, dim3((unsigned)lvae_ceil_div(B * D, 256)), dim3(256), 0,
                           st, z, mu, logvar, g, g_elem, B, K, D, d_mu, d_logvar);
        LVAE_LAUNCH_CHECK();
    }
    return LVAE_OK;
}

int lvae_launch_kl_bwd(const float* mu, const float* logvar, const float* g_kl, const float* g_hinge,
                       int row_mode, int64_t n, int D, float lambda, float* d_mu, float* d_logvar,
                       hipStream_t st) {
    const int64_t total = n * D;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)lvae_ceil_div(total, 256)), dim3(256), 0, st, mu, logvar,
                       g_kl, g_hinge, row_mode, total, D, lambda, d_mu, d_logvar);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// reparameterize (modules/encoder.py:148-157): z = mu + eps * exp(0.5 logvar).
// d mu[b,d] = sum_k g[b,k,d] ;  d logvar[b,d] = 0.5 exp(0.5 logvar) sum_k g[b,k,d] eps[b,k,d].
__global__ void __launch_bounds__(256)
reparam_bwd_kernel(const float* __restrict__ logvar, const float* __restrict__ eps,
                   const float* __restrict__ g, int64_t B, int64_t K, int D, float* __restrict__ d_mu,
                   float* __restrict__ d_logvar) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * D) return;
    const int64_t b = idx / D;
    const int d = (int)(idx - b * D);
    float am = 0.f, al = 0.f;
    for (int64_t k = 0; k < K; ++k) {
        const int64_t e = (b * K + k) * D + d;
        am += g[e];
        al = fmaf(g[e], eps[e], al);
    }
    if (d_mu) d_mu[idx] = am;
    if (d_logvar) d_logvar[idx] = 0.5f * expf(0.5f * logvar[idx]) * al;
}

int lvae_launch_reparam_bwd(const float* logvar, const float* eps, const float* g, int64_t B, int64_t K, int D,
                            float* d_mu, float* d_logvar, hipStream_t st) {
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3((unsigned)lvae_ceil_div(B * D, 256)), dim3(256), 0, st, logvar,
                       eps, g, B, K, D, d_mu, d_logvar);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
