// The training-time tail of EncoderNewsVAE.encode (NewsVAE/modules/encoder.py:79-107) for ONE training batch, as ONE
// launch forward and ONE launch backward.  The reference runs this every step at batch size 32: reparameterize
// (:148-157), two sample_log_likelihood calls (:93-98), approximate_log_q_z(method="chen", prod_marginals=True) with
// its two means (:102-104, loss_and_optimisation.py:139-221) and maximum_mean_discrepancy (:107, :46-76).  At that
// size the job is a few thousand pairs: launch-bound.  The general kernels of this library need ~40 launches for it
// (and as many again for the backward); here one workgroup of 16 waves keeps the batch in LDS and does everything in
// the direct form (z - mu)^2 / var -- no expansion, no conditioning concerns, float32 like the reference.
//
// Envelope: B <= 128, B * D <= 4096 (five [B, D] float arrays and the [B, B] softmax weights in LDS for the backward, four
// (row, dimension) elements per thread), B * K * D <= 2^20.
// Outside it the host mirror falls back to the general kernels.
#include <mutex>

#include "lvae_common.h"

#define TINY_THREADS 1024
#define TINY_WAVES 16
#define TINY_MAX_B 128
#define TINY_EPT 4                // (row, dimension) elements per thread: B * D <= 4096

struct TinyGeom {
    int B, K, D;
    int DS;                       // LDS row stride of the [B, D] arrays: D | 1 (odd -> a column walk is conflict-free)
    float log_norm;
};

// z[b,k,d] = mu + eps * exp(0.5 logvar): separate multiply and add, like the reference's torch.mul / + (:157)
__device__ __forceinline__ float tiny_z(float mu, float lv, float eps) {
    return __fadd_rn(mu, __fmul_rn(eps, expf(0.5f * lv)));
}

// t_ijd = -0.5 (lv_jd + log 2pi + (z_id - mu_jd)^2 iv_jd)
#define TINY_TERM(zv, j, d) (-0.5f * (s_lv[(j) * DS + (d)] + LVAE_LOG_2PI + ((zv) - s_mu[(j) * DS + (d)]) * ((zv) - s_mu[(j) * DS + (d)]) * s_iv[(j) * DS + (d)]))

// Shared first phase: (mu, logvar, 1/var, first sample z0) of the batch into LDS, row stride DS.
__device__ __forceinline__ void tiny_load(const TinyGeom& G, const float* __restrict__ mu, const float* __restrict__ logvar,
                                          const float* __restrict__ eps, float* s_mu, float* s_lv, float* s_iv,
                                          float* s_z0) {
    const int D = G.D, DS = G.DS, BD = G.B * D;
    for (int e = threadIdx.x; e < BD; e += TINY_THREADS) {
        const int b = e / D, d = e % D;
        const float m = mu[e], lv = logvar[e];
        s_mu[b * DS + d] = m;
        s_lv[b * DS + d] = lv;
        s_iv[b * DS + d] = expf(-lv);
        s_z0[b * DS + d] = tiny_z(m, lv, eps[(int64_t)b * G.K * D + d]);       // first sample (:81)
    }
}

// Joint logsumexp over the posteriors for every sample: a wave per sample i, a lane per posterior j (two when
// B > 64), the sum over dimensions in the lane; ONE pair of wave reductions per sample.  Leaves lq[i] = LSE_j v_ij
// (un-normalised) in s_lq and, if s_P != NULL, P[i][j] = wq * exp(v_ij - lq_i) (the joint softmax weights).
__device__ __forceinline__ void tiny_joint(const TinyGeom& G, const float* s_mu, const float* s_lv, const float* s_iv,
                                           const float* s_z0, float* s_lq, float* s_P, float wq) {
    const int B = G.B, D = G.D, DS = G.DS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < B; i += TINY_WAVES) {
        float v[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            if (j < B) {
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc += TINY_TERM(s_z0[i * DS + d], j, d);
                v[c] = acc;
            }
        }
        const float mx = wave_max(fmaxf(v[0], v[1]));
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (lane + 64 * c < B) se += expf(v[c] - mx);
        se = wave_sum(se);
        const float lq = mx + logf(se);
        if (lane == 0) s_lq[i] = lq;
        if (s_P) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (lane + 64 * c < B) s_P[i * B + lane + 64 * c] = wq * expf(v[c] - lq);
        }
    }
}

// ---- forward -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TINY_THREADS)
encode_tail_tiny_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
                        const float* __restrict__ prior, const TinyGeom G, float* __restrict__ z_out,
                        float* __restrict__ log_q_z_x, float* __restrict__ log_p_z, float* __restrict__ log_qz,
                        float* __restrict__ log_qz_pm, float* __restrict__ scalars) {
    extern __shared__ float sh[];
    const int B = G.B, K = G.K, D = G.D, DS = G.DS, BS = B * DS, BD = B * D;
    float* s_mu = sh;
    float* s_lv = s_mu + BS;
    float* s_iv = s_lv + BS;
    float* s_z0 = s_iv + BS;
    float* s_x = s_z0 + BS;            // prior draw, later reused for the per-(sample, dimension) logsumexps
    float* s_lq = s_x + BS;            // [B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    tiny_load(G, mu, logvar, eps, s_mu, s_lv, s_iv, s_z0);
    if (prior)
        for (int e = tid; e < BD; e += TINY_THREADS) s_x[(e / D) * DS + (e % D)] = prior[e];
    if (z_out) {
        const int total = B * K * D;
        for (int e = tid; e < total; e += TINY_THREADS) {
            const int b = e / (K * D), d = e % D;
            z_out[e] = tiny_z(mu[b * D + d], logvar[b * D + d], eps[e]);
        }
    }
    __syncthreads();

    // per-(b, k) densities under the own posterior and the prior (:93-98; loss_and_optimisation.py:93-101)
    if (log_q_z_x || log_p_z) {
        for (int p = wave; p < B * K; p += TINY_WAVES) {
            const int b = p / K;
            float aq = 0.f, ap = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float m = s_mu[b * DS + d], lv = s_lv[b * DS + d];
                const float zz = tiny_z(m, lv, eps[(int64_t)p * D + d]);
                const float df = zz - m;
                aq += -0.5f * (lv + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv)));
                ap += -0.5f * (LVAE_LOG_2PI + zz * zz);
            }
            aq = wave_sum(aq);
            ap = wave_sum(ap);
            if (lane == 0) {
                if (log_q_z_x) log_q_z_x[p] = aq;
                if (log_p_z) log_p_z[p] = ap;
            }
        }
    }

    // biased Gaussian-kernel MMD of the first samples against the prior draw (loss_and_optimisation.py:46-76):
    // k(a, b) = exp(-|a - b|^2 / D^2), mean(Kxx) + mean(Kyy) - 2 mean(Kxy)   (before s_x is reused below)
    if (prior && scalars) {
        double acc[3] = {0.0, 0.0, 0.0};
        const float inv = 1.0f / ((float)D * (float)D);
        const int BB = B * B;
        for (int p = tid; p < 3 * BB; p += TINY_THREADS) {
            const int which = p / BB, i = (p % BB) / B, j = p % B;
            const float* a = which == 1 ? s_z0 : s_x;
            const float* b = which == 0 ? s_x : s_z0;
            float d2 = 0.f;
            for (int d = 0; d < D; ++d) {
                const float df = a[i * DS + d] - b[j * DS + d];
                d2 += df * df;
            }
            acc[which] += (double)__expf(-d2 * inv);
        }
        __shared__ double sm[3][TINY_WAVES];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const double t = wave_sum_d(acc[w]);
            if (lane == 0) sm[w][wave] = t;
        }
        __syncthreads();
        if (tid == 0) {
            double t[3] = {0.0, 0.0, 0.0};
            for (int w = 0; w < 3; ++w)
                for (int k = 0; k < TINY_WAVES; ++k) t[w] += sm[w][k];
            scalars[2] = (float)((t[0] + t[1] - 2.0 * t[2]) / ((double)B * (double)B));
        }
    }
    if (!(log_qz || log_qz_pm || scalars)) return;
    __syncthreads();

    // log q(z_i) over the batch's posteriors (loss_and_optimisation.py:178-198): joint term per sample ...
    tiny_joint(G, s_mu, s_lv, s_iv, s_z0, s_lq, nullptr, 0.f);
    // ... and the per-(sample, dimension) logsumexp of the product of marginals (:193-195): a thread per (i, d), the
    // posteriors in a loop (two passes: maximum, then one expf per term); no cross-lane traffic
    float* s_lse = s_x;
    for (int e = tid; e < BD; e += TINY_THREADS) {
        const int i = e / D, d = e % D;
        const float zv = s_z0[i * DS + d];
        float mx = -INFINITY;
        for (int j = 0; j < B; ++j) mx = fmaxf(mx, TINY_TERM(zv, j, d));
        float se = 0.f;
        for (int j = 0; j < B; ++j) se += __expf(TINY_TERM(zv, j, d) - mx);
        s_lse[i * DS + d] = mx + logf(se) - G.log_norm;
    }
    __syncthreads();
    if (wave == 0) {
        float a = 0.f, b = 0.f;
        for (int i = lane; i < B; i += 64) {
            float pm = 0.f;
            for (int d = 0; d < D; ++d) pm += s_lse[i * DS + d];
            const float lq = s_lq[i] - G.log_norm;
            if (log_qz) log_qz[i] = lq;
            if (log_qz_pm) log_qz_pm[i] = pm;
            a += lq;
            b += pm;
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (lane == 0 && scalars) {
            scalars[0] = a / (float)B;           // the .mean() of :217-219
            scalars[1] = b / (float)B;
        }
    }
}

// ---- backward ------------------------------------------------------------------------------------------------
// Upstream gradients (each may be NULL): g_z [B,K,D], g_lqx / g_lpz [B,K], g_lq / g_pm: one float each in device
// memory (on the two MEANS).  Everything is a function of (mu, logvar) directly and through z = mu + eps sigma.
// A thread owns up to TINY_EPT (row, dimension) elements for the whole kernel: it accumulates that element's
// gradient as a sample (sum over posteriors), as a posterior (sum over samples) and in the element-wise tail, in
// registers; deterministic.  With w_ijd = wq softmax_j(v_ij) + wp softmax_j(t_ijd) and a = (z_id - mu_jd) / var_jd:
//   d/dz_id = -sum_j w a,   d/dmu_jd = sum_i w a,   d/dlogvar_jd = sum_i w (-1/2)(1 - (z_id - mu_jd) a).
__global__ void __launch_bounds__(TINY_THREADS)
encode_tail_tiny_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                            const float* __restrict__ eps, const TinyGeom G, const float* __restrict__ g_z,
                            const float* __restrict__ g_lqx, const float* __restrict__ g_lpz,
                            const float* __restrict__ g_lq, const float* __restrict__ g_pm,
                            float* __restrict__ d_mu, float* __restrict__ d_lv) {
    extern __shared__ float sh[];
    const int B = G.B, K = G.K, D = G.D, DS = G.DS, BS = B * DS, BD = B * D;
    float* s_mu = sh;
    float* s_lv = s_mu + BS;
    float* s_iv = s_lv + BS;
    float* s_z0 = s_iv + BS;
    float* s_lse = s_z0 + BS;          // [B][DS] per-(sample, dimension) logsumexp over the posteriors (un-normalised)
    float* s_lq = s_lse + BS;          // [B] joint logsumexp (un-normalised)
    float* s_P = s_lq + B;             // [B][B] wq * softmax_j(v_ij)
    const int tid = threadIdx.x;
    const bool pair_terms = g_lq || g_pm;
    const float wq = g_lq ? g_lq[0] / (float)B : 0.f;       // weight of every sample's joint term in the mean
    const float wp = g_pm ? g_pm[0] / (float)B : 0.f;

    tiny_load(G, mu, logvar, eps, s_mu, s_lv, s_iv, s_z0);
    __syncthreads();

    float dz0[TINY_EPT], dm[TINY_EPT], dv[TINY_EPT];
#pragma unroll
    for (int u = 0; u < TINY_EPT; ++u) dz0[u] = dm[u] = dv[u] = 0.f;

    if (pair_terms) {
        tiny_joint(G, s_mu, s_lv, s_iv, s_z0, s_lq, s_P, wq);
#pragma unroll
        for (int u = 0; u < TINY_EPT; ++u) {
            const int e = tid + u * TINY_THREADS;
            if (e < BD) {
                const int i = e / D, d = e % D;
                const float zv = s_z0[i * DS + d];
                float mx = -INFINITY;
                for (int j = 0; j < B; ++j) mx = fmaxf(mx, TINY_TERM(zv, j, d));
                float se = 0.f;
                for (int j = 0; j < B; ++j) se += __expf(TINY_TERM(zv, j, d) - mx);
                s_lse[i * DS + d] = mx + logf(se);
            }
        }
        __syncthreads();
        // as a sample: sum over the posteriors
#pragma unroll
        for (int u = 0; u < TINY_EPT; ++u) {
            const int e = tid + u * TINY_THREADS;
            if (e < BD) {
                const int i = e / D, d = e % D;
                const float zv = s_z0[i * DS + d], ls = s_lse[i * DS + d];
                float acc = 0.f;
                for (int j = 0; j < B; ++j) {
                    const float df = zv - s_mu[j * DS + d];
                    const float a = df * s_iv[j * DS + d];
                    const float t = -0.5f * (s_lv[j * DS + d] + LVAE_LOG_2PI + df * a);
                    acc -= (s_P[i * B + j] + wp * __expf(t - ls)) * a;
                }
                dz0[u] = acc;
            }
        }
        // as a posterior: sum over the samples
#pragma unroll
        for (int u = 0; u < TINY_EPT; ++u) {
            const int e = tid + u * TINY_THREADS;
            if (e < BD) {
                const int j = e / D, d = e % D;
                const float m = s_mu[j * DS + d], iv = s_iv[j * DS + d], base = s_lv[j * DS + d] + LVAE_LOG_2PI;
                float am = 0.f, av = 0.f;
                for (int i = 0; i < B; ++i) {
                    const float df = s_z0[i * DS + d] - m;
                    const float a = df * iv;
                    const float t = -0.5f * (base + df * a);
                    const float w = s_P[i * B + j] + wp * __expf(t - s_lse[i * DS + d]);
                    am += w * a;
                    av += w * (-0.5f) * (1.0f - df * a);
                }
                dm[u] = am;
                dv[u] = av;
            }
        }
    }

    // element-wise tail: everything that reaches z[b,k,d], chained through z = mu + eps sigma, plus the direct
    // (mu, logvar) dependence of log q(z|x)
#pragma unroll
    for (int u = 0; u < TINY_EPT; ++u) {
        const int e = tid + u * TINY_THREADS;
        if (e < BD) {
            const int b = e / D, d = e % D;
            const float m = s_mu[b * DS + d], lv = s_lv[b * DS + d], iv = s_iv[b * DS + d];
            const float sd = expf(0.5f * lv);
            float am = dm[u], av = dv[u];
            for (int k = 0; k < K; ++k) {
                const int64_t p = (int64_t)b * K + k;
                const float ep = eps[p * D + d];
                const float zz = __fadd_rn(m, __fmul_rn(ep, sd));
                const float df = zz - m;
                float gz = (g_z ? g_z[p * D + d] : 0.f) + (k == 0 ? dz0[u] : 0.f);
                if (g_lpz) gz -= g_lpz[p] * zz;
                if (g_lqx) {
                    const float gq = g_lqx[p], a = df * iv;
                    gz -= gq * a;
                    am += gq * a;
                    av += gq * (-0.5f) * (1.0f - df * a);
                }
                am += gz;
                av += gz * ep * 0.5f * sd;
            }
            if (d_mu) d_mu[e] = am;
            if (d_lv) d_lv[e] = av;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
bool lvae_encode_tail_tiny_ok(int64_t B, int64_t K, int D) {
    return B >= 1 && B <= TINY_MAX_B && K >= 1 && D >= 1 && B * D <= TINY_EPT * TINY_THREADS && B * K * D <= (1 << 20);
}

static TinyGeom tiny_geom(int64_t B, int64_t K, int D, float log_norm) {
    TinyGeom G;
    G.B = (int)B;
    G.K = (int)K;
    G.D = D;
    G.DS = D | 1;
    G.log_norm = log_norm;
    return G;
}

// dynamic LDS above 64 KB needs the attribute; it belongs to the DEVICE that is current when it is set, so it is
// set once per device ordinal (a process that drives several GPUs launches on each of them)
#define TINY_MAX_DEVICES 64
static int tiny_allow_lds() {
    static std::mutex mu;
    static bool done[TINY_MAX_DEVICES];
    int dev = -1;
    LVAE_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= TINY_MAX_DEVICES)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED, "device ordinal %d outside 0..%d", dev, TINY_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev]) return LVAE_OK;
    LVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_tail_tiny_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    LVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_tail_tiny_bwd_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done[dev] = true;
    return LVAE_OK;
}

int lvae_launch_encode_tail_tiny(const float* mu, const float* logvar, const float* eps, const float* prior, int64_t B,
                                 int64_t K, int D, float log_norm, float* z, float* log_q_z_x, float* log_p_z,
                                 float* log_qz, float* log_qz_pm, float* scalars, hipStream_t st) {
    int rc = tiny_allow_lds();
    if (rc) return rc;
    const TinyGeom G = tiny_geom(B, K, D, log_norm);
    const size_t lds = ((size_t)5 * B * G.DS + B) * sizeof(float);
    hipLaunchKernelGGL(encode_tail_tiny_kernel, dim3(1), dim3(TINY_THREADS), lds, st, mu, logvar, eps, prior, G, z,
                       log_q_z_x, log_p_z, log_qz, log_qz_pm, scalars);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_encode_tail_tiny_bwd(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K, int D,
                                     float log_norm, const float* g_z, const float* g_lqx, const float* g_lpz,
                                     const float* g_lq, const float* g_pm, float* d_mu, float* d_lv, hipStream_t st) {
    int rc = tiny_allow_lds();
    if (rc) return rc;
    const TinyGeom G = tiny_geom(B, K, D, log_norm);
    const size_t lds = ((size_t)5 * B * G.DS + B + B * B) * sizeof(float);
    hipLaunchKernelGGL(encode_tail_tiny_bwd_kernel, dim3(1), dim3(TINY_THREADS), lds, st, mu, logvar, eps, G, g_z, g_lqx,
                       g_lpz, g_lq, g_pm, d_mu, d_lv);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
