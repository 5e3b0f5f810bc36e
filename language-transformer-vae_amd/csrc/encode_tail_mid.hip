// The training-time tail of EncoderNewsVAE.encode (NewsVAE/modules/encoder.py:79-107) for batches beyond the
// one-workgroup form of encode_tail_tiny.hip (B = 128 takes that single workgroup 0.2 ms): the same direct-form
// arithmetic -- (z - mu)^2 / var, no expansion, float32 like the reference -- spread over B / 4 workgroups.
//   forward   one launch over tiles of four samples (a wave per sample): reparameterize (:148-157), log q(z|x) and
//             log p(z) per (b, k) (:93-98), log q(z) and its product of marginals for the first sample of every row
//             (:102-104, loss_and_optimisation.py:139-221, "chen"), the row sums of the three kernel matrices of
//             maximum_mean_discrepancy (:107, :46-76); one more launch of a single workgroup turns the per-sample
//             values into the three scalars
//   backward  two launches: A over tiles of four SAMPLES (the joint softmax weights P_ij, the per-(sample, dimension)
//             logsumexps and the gradient that reaches each first sample from the pair terms), B over tiles of four
//             POSTERIORS (the sums over samples, then the element-wise chain through z = mu + eps sigma)
// The other axis streams through LDS in chunks of 64 rows (row stride D | 1: a lane per row reads conflict-free).
// Nothing is O(B^2) except the [B, B] weights that A hands to B through the workspace.  Deterministic (no atomics).
// Envelope: D <= 64, B <= 4096.
#include <mutex>

#include "lvae_common.h"

#define MID_TI 4
#define MID_JC 64
#define MID_THREADS 256
#define MID_MAX_D 64
#define MID_MAX_B 4096
#define MID_EPT (MID_JC * MID_MAX_D / MID_THREADS)      // elements of a chunk array per thread, at most

struct MidGeom {
    int B, K, D;
    int DS;          // LDS row stride: D | 1
    int DPAD, JQ;    // lanes of a wave as (dimension, subset of the other axis): DPAD = pow2 >= D, JQ = 64 / DPAD
    float log_norm;
};

__device__ __forceinline__ float mid_z(float mu, float lv, float eps) {      // separate multiply and add (:157)
    return __fadd_rn(mu, __fmul_rn(eps, expf(0.5f * lv)));
}

__device__ __forceinline__ void mid_push(float& m, float& s, float w) {      // online logsumexp, natural base
    if (w > m) {
        s = s * __expf(m - w) + 1.0f;      // (exp(-inf) = 0 the first time)
        m = w;
    } else {
        s += __expf(w - m);
    }
}

__device__ __forceinline__ void mid_merge(float& m, float& s, float mo, float so) {
    const float mm = fmaxf(m, mo);
    s = (s > 0.f ? s * __expf(m - mm) : 0.f) + (so > 0.f ? so * __expf(mo - mm) : 0.f);
    m = mm;
}

// t_ijd = -0.5 (lv_jd + log 2pi + (z_id - mu_jd)^2 iv_jd) for chunk row r
#define MID_TERM(zv, r, d) (-0.5f * (s_lv[(r) * DS + (d)] + LVAE_LOG_2PI + ((zv) - s_mu[(r) * DS + (d)]) * ((zv) - s_mu[(r) * DS + (d)]) * s_iv[(r) * DS + (d)]))

// ---- forward -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(MID_THREADS)
encode_tail_mid_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
                       const float* __restrict__ prior, const MidGeom G, float* __restrict__ z_out,
                       float* __restrict__ log_q_z_x, float* __restrict__ log_p_z, float* __restrict__ log_qz,
                       float* __restrict__ log_qz_pm, float* __restrict__ lq_ws, float* __restrict__ pm_ws,
                       double* __restrict__ mmd_part, int want_pairs) {
    extern __shared__ float sh[];
    const int B = G.B, K = G.K, D = G.D, DS = G.DS;
    float* s_mu = sh;                       // chunk arrays [MID_JC][DS]
    float* s_lv = s_mu + MID_JC * DS;
    float* s_iv = s_lv + MID_JC * DS;
    float* s_z0 = s_iv + MID_JC * DS;
    float* s_x = s_z0 + MID_JC * DS;
    float* t_z0 = s_x + MID_JC * DS;        // tile arrays [MID_TI][DS]
    float* t_x = t_z0 + MID_TI * DS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * MID_TI;
    const int i = i0 + wave;
    const bool live = i < B;                // (wave-uniform; a wave past the end repeats the last row and publishes nothing)
    const int irow = live ? i : B - 1;

    // per-(b, k) densities under the own posterior and the prior, and z itself: a wave per row, a lane per dimension
    if (live && (z_out || log_q_z_x || log_p_z)) {
        float m = 0.f, lv = 0.f;
        if (lane < D) {
            m = mu[(int64_t)i * D + lane];
            lv = logvar[(int64_t)i * D + lane];
        }
        const float ev = expf(lv);
        for (int k = 0; k < K; ++k) {
            const int64_t p = (int64_t)i * K + k;
            float aq = 0.f, ap = 0.f;
            if (lane < D) {
                const float zz = mid_z(m, lv, eps[p * D + lane]);
                if (z_out) z_out[p * D + lane] = zz;
                const float df = zz - m;
                aq = -0.5f * (lv + LVAE_LOG_2PI + __fdiv_rn(df * df, ev));
                ap = -0.5f * (LVAE_LOG_2PI + zz * zz);
            }
            if (log_q_z_x || log_p_z) {
                aq = wave_sum(aq);
                ap = wave_sum(ap);
                if (lane == 0) {
                    if (log_q_z_x) log_q_z_x[p] = aq;
                    if (log_p_z) log_p_z[p] = ap;
                }
            }
        }
    }
    if (!want_pairs) return;

    for (int e = tid; e < MID_TI * D; e += MID_THREADS) {
        const int w = e / D, d = e - w * D;
        const int row = i0 + w < B ? i0 + w : B - 1;
        t_z0[w * DS + d] = mid_z(mu[(int64_t)row * D + d], logvar[(int64_t)row * D + d], eps[(int64_t)row * K * D + d]);
        t_x[w * DS + d] = prior ? prior[(int64_t)row * D + d] : 0.f;
    }
    float jm = -INFINITY, js = 0.f;         // joint logsumexp: lane <-> posterior
    float pmx = -INFINITY, psum = 0.f;      // per-dimension logsumexp: lane <-> (dimension, subset of posteriors)
    double a_xx = 0.0, a_yy = 0.0, a_xy = 0.0;
    const int dd = lane % G.DPAD, jq = lane / G.DPAD;
    const bool d_ok = dd < D;
    const float inv = 1.0f / ((float)D * (float)D);
    (void)irow;
    // the chunk after the one being consumed is already on its way into registers (MID_EPT elements per thread and
    // array): a chunk's global loads would otherwise be a latency of their own between two barriers, for every chunk
    float pf_m[MID_EPT], pf_lv[MID_EPT], pf_e[MID_EPT], pf_x[MID_EPT];
    auto fetch = [&](int j0) __attribute__((always_inline)) {
        const int jn = B - j0 < MID_JC ? B - j0 : MID_JC;
#pragma unroll
        for (int u = 0; u < MID_EPT; ++u) {
            const int e = tid + u * MID_THREADS;
            if (e < jn * D) {
                const int r = e / D, d = e - r * D;
                const int64_t j = j0 + r;
                pf_m[u] = mu[j * D + d];
                pf_lv[u] = logvar[j * D + d];
                pf_e[u] = eps[j * K * D + d];
                pf_x[u] = prior ? prior[j * D + d] : 0.f;
            }
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < B; j0 += MID_JC) {
        __syncthreads();                    // the previous chunk is consumed (first trip: the tile rows are written)
        const int jn = B - j0 < MID_JC ? B - j0 : MID_JC;
#pragma unroll
        for (int u = 0; u < MID_EPT; ++u) {
            const int e = tid + u * MID_THREADS;
            if (e < jn * D) {
                const int r = e / D, d = e - r * D;
                s_mu[r * DS + d] = pf_m[u];
                s_lv[r * DS + d] = pf_lv[u];
                s_iv[r * DS + d] = expf(-pf_lv[u]);
                s_z0[r * DS + d] = mid_z(pf_m[u], pf_lv[u], pf_e[u]);
                s_x[r * DS + d] = pf_x[u];
            }
        }
        __syncthreads();
        if (j0 + MID_JC < B) fetch(j0 + MID_JC);
        if (lane < jn) {
            float acc = 0.f, dxx = 0.f, dyy = 0.f, dxy = 0.f;
            if (prior) {
#pragma unroll 4
                for (int d = 0; d < D; ++d) {
                    const float zv = t_z0[wave * DS + d], xv = t_x[wave * DS + d];
                    acc += MID_TERM(zv, lane, d);
                    const float xj = s_x[lane * DS + d], zj = s_z0[lane * DS + d];
                    dxx += (xv - xj) * (xv - xj);
                    dyy += (zv - zj) * (zv - zj);
                    dxy += (xv - zj) * (xv - zj);
                }
                a_xx += (double)__expf(-dxx * inv);
                a_yy += (double)__expf(-dyy * inv);
                a_xy += (double)__expf(-dxy * inv);
            } else {
#pragma unroll 4
            for (int d = 0; d < D; ++d) acc += MID_TERM(t_z0[wave * DS + d], lane, d);
            }
            mid_push(jm, js, acc);
        }
        if (d_ok) {
            const float zv = t_z0[wave * DS + dd];
#pragma unroll 4
            for (int r = jq; r < jn; r += G.JQ) mid_push(pmx, psum, MID_TERM(zv, r, dd));
        }
    }
    // joint: merge the lanes
    const float mx = wave_max(jm);
    const float se = wave_sum(js > 0.f ? js * expf(jm - mx) : 0.f);
    const float lq = mx + logf(se) - G.log_norm;
    // product of marginals: merge the subsets of every dimension, then sum the dimensions
    for (int o = G.DPAD; o < 64; o <<= 1) {
        const float mo = __shfl_xor(pmx, o, 64), so = __shfl_xor(psum, o, 64);
        mid_merge(pmx, psum, mo, so);
    }
    const float pm = wave_sum((d_ok && jq == 0) ? pmx + logf(psum) - G.log_norm : 0.f);
    a_xx = wave_sum_d(a_xx);
    a_yy = wave_sum_d(a_yy);
    a_xy = wave_sum_d(a_xy);
    if (lane == 0) {
        if (live) {
            lq_ws[i] = lq;
            pm_ws[i] = pm;
            if (log_qz) log_qz[i] = lq;
            if (log_qz_pm) log_qz_pm[i] = pm;
        }
        double* mp = mmd_part + ((int64_t)blockIdx.x * MID_TI + wave) * 3;
        mp[0] = live ? a_xx : 0.0;
        mp[1] = live ? a_yy : 0.0;
        mp[2] = live ? a_xy : 0.0;
    }
}

// scalars = {mean log q(z), mean product of marginals, mmd} from the per-sample values (one workgroup, float64 sums)
__global__ void __launch_bounds__(MID_THREADS)
encode_tail_mid_finalize_kernel(const float* __restrict__ lq_ws, const float* __restrict__ pm_ws,
                                const double* __restrict__ mmd_part, int B, int nparts, int have_prior,
                                float* __restrict__ scalars) {
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < B; i += MID_THREADS) {
        a[0] += (double)lq_ws[i];
        a[1] += (double)pm_ws[i];
    }
    for (int p = threadIdx.x; p < nparts; p += MID_THREADS) {
        a[2] += mmd_part[p * 3 + 0];
        a[3] += mmd_part[p * 3 + 1];
        a[4] += mmd_part[p * 3 + 2];
    }
    __shared__ double sm[5][MID_THREADS / 64];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double t = wave_sum_d(a[q]);
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5];
        for (int q = 0; q < 5; ++q) t[q] = sm[q][0] + sm[q][1] + sm[q][2] + sm[q][3];
        scalars[0] = (float)(t[0] / (double)B);
        scalars[1] = (float)(t[1] / (double)B);
        if (have_prior) scalars[2] = (float)((t[2] + t[3] - 2.0 * t[4]) / ((double)B * (double)B));
    }
}

// ---- backward A: tiles of samples -------------------------------------------------------------------------------
// For sample i (a wave): the joint logsumexp lq_i and the per-dimension ones lse_id over all posteriors (pass 1), then
// (pass 2) P_ij = wq softmax_j(v_ij) -> workspace [B, B], and dz0_id = -sum_j (P_ij + wp softmax_j(t_ijd)) a_ijd with
// a_ijd = (z_id - mu_jd) / var_jd.  Also leaves z0 [B, D] and lse [B, D] in the workspace for B.
__global__ void __launch_bounds__(MID_THREADS)
encode_tail_mid_bwd_a_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                             const float* __restrict__ eps, const MidGeom G, const float* __restrict__ g_lq,
                             const float* __restrict__ g_pm, float* __restrict__ P_ws, float* __restrict__ lse_ws,
                             float* __restrict__ dz0_ws, float* __restrict__ z0_ws) {
    extern __shared__ float sh[];
    const int B = G.B, K = G.K, D = G.D, DS = G.DS;
    float* s_mu = sh;
    float* s_lv = s_mu + MID_JC * DS;
    float* s_iv = s_lv + MID_JC * DS;
    float* t_z0 = s_iv + MID_JC * DS;       // [MID_TI][DS]
    float* t_P = t_z0 + MID_TI * DS;        // [MID_TI][MID_JC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * MID_TI;
    const int i = i0 + wave;
    const bool live = i < B;
    const float wq = g_lq ? g_lq[0] / (float)B : 0.f;
    const float wp = g_pm ? g_pm[0] / (float)B : 0.f;
    for (int e = tid; e < MID_TI * D; e += MID_THREADS) {
        const int w = e / D, d = e - w * D;
        const int row = i0 + w < B ? i0 + w : B - 1;
        const float zz = mid_z(mu[(int64_t)row * D + d], logvar[(int64_t)row * D + d], eps[(int64_t)row * K * D + d]);
        t_z0[w * DS + d] = zz;
        if (i0 + w < B) z0_ws[(int64_t)row * D + d] = zz;
    }
    const int dd = lane % G.DPAD, jq = lane / G.DPAD;
    const bool d_ok = dd < D;
    // (the next chunk travels into registers while the current one is consumed, as in the forward kernel; the two
    //  passes form one sequence of 2 * ceil(B / 64) chunks)
    float pf_m[MID_EPT], pf_lv[MID_EPT];
    auto fetch = [&](int j0) __attribute__((always_inline)) {
        const int jn = B - j0 < MID_JC ? B - j0 : MID_JC;
#pragma unroll
        for (int u = 0; u < MID_EPT; ++u) {
            const int e = tid + u * MID_THREADS;
            if (e < jn * D) {
                const int r = e / D, d = e - r * D;
                pf_m[u] = mu[(int64_t)(j0 + r) * D + d];
                pf_lv[u] = logvar[(int64_t)(j0 + r) * D + d];
            }
        }
    };
    auto load_chunk = [&](int j0, int jn) __attribute__((always_inline)) {
        (void)j0;
#pragma unroll
        for (int u = 0; u < MID_EPT; ++u) {
            const int e = tid + u * MID_THREADS;
            if (e < jn * D) {
                const int r = e / D, d = e - r * D;
                s_mu[r * DS + d] = pf_m[u];
                s_lv[r * DS + d] = pf_lv[u];
                s_iv[r * DS + d] = expf(-pf_lv[u]);
            }
        }
    };
    fetch(0);
    // pass 1: the two kinds of logsumexp
    float jm = -INFINITY, js = 0.f, pmx = -INFINITY, psum = 0.f;
    for (int j0 = 0; j0 < B; j0 += MID_JC) {
        __syncthreads();
        const int jn = B - j0 < MID_JC ? B - j0 : MID_JC;
        load_chunk(j0, jn);
        __syncthreads();
        fetch(j0 + MID_JC < B ? j0 + MID_JC : 0);          // (after the last chunk: the first one again, for pass 2)
        if (lane < jn) {
            float acc = 0.f;
#pragma unroll 4
            for (int d = 0; d < D; ++d) acc += MID_TERM(t_z0[wave * DS + d], lane, d);
            mid_push(jm, js, acc);
        }
        if (d_ok) {
            const float zv = t_z0[wave * DS + dd];
#pragma unroll 4
            for (int r = jq; r < jn; r += G.JQ) mid_push(pmx, psum, MID_TERM(zv, r, dd));
        }
    }
    const float mx = wave_max(jm);
    const float lq = mx + logf(wave_sum(js > 0.f ? js * expf(jm - mx) : 0.f));      // un-normalised
    for (int o = G.DPAD; o < 64; o <<= 1) {
        const float mo = __shfl_xor(pmx, o, 64), so = __shfl_xor(psum, o, 64);
        mid_merge(pmx, psum, mo, so);
    }
    const float lse_d = pmx + logf(psum);                                           // un-normalised
    if (live && d_ok && jq == 0) lse_ws[(int64_t)i * D + dd] = lse_d;
    // pass 2: weights and the sample-side gradient
    float acc_dz = 0.f;
    const float zv = d_ok ? t_z0[wave * DS + dd] : 0.f;
    for (int j0 = 0; j0 < B; j0 += MID_JC) {
        __syncthreads();
        const int jn = B - j0 < MID_JC ? B - j0 : MID_JC;
        load_chunk(j0, jn);
        __syncthreads();
        if (j0 + MID_JC < B) fetch(j0 + MID_JC);
        if (lane < jn) {
            float v = 0.f;
#pragma unroll 4
            for (int d = 0; d < D; ++d) v += MID_TERM(t_z0[wave * DS + d], lane, d);
            const float P = wq * expf(v - lq);
            t_P[wave * MID_JC + lane] = P;
            if (live) P_ws[(int64_t)i * B + j0 + lane] = P;
        }
        __syncthreads();
        if (d_ok) {
#pragma unroll 4
            for (int r = jq; r < jn; r += G.JQ) {
                const float df = zv - s_mu[r * DS + dd];
                const float a = df * s_iv[r * DS + dd];
                const float t = -0.5f * (s_lv[r * DS + dd] + LVAE_LOG_2PI + df * a);
                acc_dz -= (t_P[wave * MID_JC + r] + wp * __expf(t - lse_d)) * a;
            }
        }
    }
    for (int o = G.DPAD; o < 64; o <<= 1) acc_dz += __shfl_xor(acc_dz, o, 64);
    if (live && d_ok && jq == 0) dz0_ws[(int64_t)i * D + dd] = acc_dz;
}

// ---- backward B: tiles of posteriors ----------------------------------------------------------------------------
// For posterior j (a wave), lane <-> (dimension, subset of samples): d mu_jd = sum_i w a, d logvar_jd = sum_i w
// (-1/2)(1 - (z_id - mu_jd) a) with w = P_ij + wp softmax_j(t_ijd), then the element-wise chain of everything that
// reaches z[j, k, d] (encode_tail_tiny.hip, same arithmetic).
__global__ void __launch_bounds__(MID_THREADS)
encode_tail_mid_bwd_b_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                             const float* __restrict__ eps, const MidGeom G, const float* __restrict__ g_z,
                             const float* __restrict__ g_lqx, const float* __restrict__ g_lpz,
                             const float* __restrict__ g_pm, int pair_terms, const float* __restrict__ P_ws,
                             const float* __restrict__ lse_ws, const float* __restrict__ dz0_ws,
                             const float* __restrict__ z0_ws, float* __restrict__ d_mu, float* __restrict__ d_lv) {
    extern __shared__ float sh[];
    const int B = G.B, K = G.K, D = G.D, DS = G.DS;
    float* c_z0 = sh;                        // [MID_JC][DS] samples of the chunk
    float* c_lse = c_z0 + MID_JC * DS;       // [MID_JC][DS]
    float* c_P = c_lse + MID_JC * DS;        // [MID_JC][MID_TI]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0t = blockIdx.x * MID_TI;
    const int j = j0t + wave;
    const bool live = j < B;
    const int jrow = live ? j : B - 1;
    const int dd = lane % G.DPAD, iq = lane / G.DPAD;
    const bool d_ok = dd < D;
    const float wp = g_pm ? g_pm[0] / (float)B : 0.f;
    float m = 0.f, lv = 0.f;
    if (d_ok) {
        m = mu[(int64_t)jrow * D + dd];
        lv = logvar[(int64_t)jrow * D + dd];
    }
    const float iv = expf(-lv), base = lv + LVAE_LOG_2PI;
    float am = 0.f, av = 0.f;
    if (pair_terms) {
        float pf_z[MID_EPT], pf_l[MID_EPT], pf_p = 0.f;
        auto fetch = [&](int i0) __attribute__((always_inline)) {
            const int in = B - i0 < MID_JC ? B - i0 : MID_JC;
#pragma unroll
            for (int u = 0; u < MID_EPT; ++u) {
                const int e = tid + u * MID_THREADS;
                if (e < in * D) {       // (z0 and lse are [B, D] row-major: a chunk is one contiguous run)
                    pf_z[u] = z0_ws[(int64_t)i0 * D + e];
                    pf_l[u] = lse_ws[(int64_t)i0 * D + e];
                }
            }
            if (tid < in * MID_TI) {
                const int r = tid / MID_TI, w = tid - r * MID_TI;
                pf_p = j0t + w < B ? P_ws[(int64_t)(i0 + r) * B + j0t + w] : 0.f;
            }
        };
        fetch(0);
        for (int i0 = 0; i0 < B; i0 += MID_JC) {
            __syncthreads();
            const int in = B - i0 < MID_JC ? B - i0 : MID_JC;
#pragma unroll
            for (int u = 0; u < MID_EPT; ++u) {
                const int e = tid + u * MID_THREADS;
                if (e < in * D) {
                    const int r = e / D, d = e - r * D;
                    c_z0[r * DS + d] = pf_z[u];
                    c_lse[r * DS + d] = pf_l[u];
                }
            }
            if (tid < in * MID_TI) c_P[tid] = pf_p;
            __syncthreads();
            if (i0 + MID_JC < B) fetch(i0 + MID_JC);
            if (d_ok) {
#pragma unroll 4
                for (int r = iq; r < in; r += G.JQ) {
                    const float df = c_z0[r * DS + dd] - m;
                    const float a = df * iv;
                    const float t = -0.5f * (base + df * a);
                    const float w = c_P[r * MID_TI + wave] + wp * __expf(t - c_lse[r * DS + dd]);
                    am += w * a;
                    av += w * (-0.5f) * (1.0f - df * a);
                }
            }
        }
        for (int o = G.DPAD; o < 64; o <<= 1) {
            am += __shfl_xor(am, o, 64);
            av += __shfl_xor(av, o, 64);
        }
    }
    if (!(live && d_ok && iq == 0)) return;
    // element-wise tail: everything that reaches z[j, k, d], chained through z = mu + eps sigma, plus the direct
    // (mu, logvar) dependence of log q(z|x)
    const float sd = expf(0.5f * lv);
    const float dz0 = pair_terms ? dz0_ws[(int64_t)j * D + dd] : 0.f;
    for (int k = 0; k < K; ++k) {
        const int64_t p = (int64_t)j * K + k;
        const float ep = eps[p * D + dd];
        const float zz = __fadd_rn(m, __fmul_rn(ep, sd));
        const float df = zz - m;
        float gz = (g_z ? g_z[p * D + dd] : 0.f) + (k == 0 ? dz0 : 0.f);
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
    if (d_mu) d_mu[(int64_t)j * D + dd] = am;
    if (d_lv) d_lv[(int64_t)j * D + dd] = av;
}

// ---- host side -------------------------------------------------------------------------------------------------
bool lvae_encode_tail_mid_ok(int64_t B, int64_t K, int D) {
    return B >= 1 && B <= MID_MAX_B && K >= 1 && D >= 1 && D <= MID_MAX_D && B * K * D <= ((int64_t)1 << 26);
}

struct MidLayout {
    size_t off_mmd, off_lq, off_pm, off_lse, off_dz0, off_z0, off_P, total;
};

static MidLayout mid_layout(int64_t B, int D) {
    MidLayout L{};
    const int64_t nwg = lvae_ceil_div(B, MID_TI);
    auto al = [](size_t o) { return (o + 255) & ~(size_t)255; };
    size_t o = 0;
    L.off_mmd = o;  o = al(o + (size_t)nwg * MID_TI * 3 * sizeof(double));
    L.off_lq = o;   o = al(o + (size_t)B * sizeof(float));
    L.off_pm = o;   o = al(o + (size_t)B * sizeof(float));
    L.off_lse = o;  o = al(o + (size_t)B * D * sizeof(float));
    L.off_dz0 = o;  o = al(o + (size_t)B * D * sizeof(float));
    L.off_z0 = o;   o = al(o + (size_t)B * D * sizeof(float));
    L.off_P = o;    o = al(o + (size_t)B * B * sizeof(float));
    L.total = o;
    return L;
}

int64_t lvae_encode_tail_mid_workspace(int64_t B, int D) { return (int64_t)mid_layout(B, D).total; }

static MidGeom mid_geom(int64_t B, int64_t K, int D, float log_norm) {
    MidGeom G;
    G.B = (int)B;
    G.K = (int)K;
    G.D = D;
    G.DS = D | 1;
    int p = 1;
    while (p < D) p <<= 1;
    G.DPAD = p;
    G.JQ = 64 / p;
    G.log_norm = log_norm;
    return G;
}

// dynamic LDS above 64 KB (D = 64: 85 KB) needs the attribute, per DEVICE (it belongs to the device that is current
// when it is set)
static int mid_allow_lds() {
    static std::mutex mtx;
    static bool done[64];
    int dev = -1;
    LVAE_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return lvae_set_error(LVAE_ERR_UNSUPPORTED, "device ordinal %d outside 0..63", dev);
    std::lock_guard<std::mutex> lk(mtx);
    if (done[dev]) return LVAE_OK;
    LVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_tail_mid_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    LVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_tail_mid_bwd_a_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    LVAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(encode_tail_mid_bwd_b_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    done[dev] = true;
    return LVAE_OK;
}

int lvae_launch_encode_tail_mid(const float* mu, const float* logvar, const float* eps, const float* prior, int64_t B,
                                int64_t K, int D, float log_norm, float* z, float* log_q_z_x, float* log_p_z,
                                float* log_qz, float* log_qz_pm, float* scalars, void* ws, hipStream_t st) {
    int rc = mid_allow_lds();
    if (rc) return rc;
    const MidGeom G = mid_geom(B, K, D, log_norm);
    const MidLayout L = mid_layout(B, D);
    char* base = (char*)ws;
    const int want_pairs = (log_qz || log_qz_pm || scalars) ? 1 : 0;
    const int nwg = (int)lvae_ceil_div(B, MID_TI);
    const size_t lds = ((size_t)5 * MID_JC * G.DS + 2 * MID_TI * G.DS) * sizeof(float);
    hipLaunchKernelGGL(encode_tail_mid_kernel, dim3(nwg), dim3(MID_THREADS), lds, st, mu, logvar, eps, prior, G, z,
                       log_q_z_x, log_p_z, log_qz, log_qz_pm, (float*)(base + L.off_lq), (float*)(base + L.off_pm),
                       (double*)(base + L.off_mmd), want_pairs);
    if (scalars)
        hipLaunchKernelGGL(encode_tail_mid_finalize_kernel, dim3(1), dim3(MID_THREADS), 0, st,
                           (const float*)(base + L.off_lq), (const float*)(base + L.off_pm),
                           (const double*)(base + L.off_mmd), (int)B, nwg * MID_TI, prior ? 1 : 0, scalars);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_encode_tail_mid_bwd(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K, int D,
                                    float log_norm, const float* g_z, const float* g_lqx, const float* g_lpz,
                                    const float* g_lq, const float* g_pm, float* d_mu, float* d_lv, void* ws,
                                    hipStream_t st) {
    int rc = mid_allow_lds();
    if (rc) return rc;
    const MidGeom G = mid_geom(B, K, D, log_norm);
    const MidLayout L = mid_layout(B, D);
    char* base = (char*)ws;
    const int pair_terms = (g_lq || g_pm) ? 1 : 0;
    const int nwg = (int)lvae_ceil_div(B, MID_TI);
    if (pair_terms) {
        const size_t lds_a = ((size_t)3 * MID_JC * G.DS + MID_TI * G.DS + MID_TI * MID_JC) * sizeof(float);
        hipLaunchKernelGGL(encode_tail_mid_bwd_a_kernel, dim3(nwg), dim3(MID_THREADS), lds_a, st, mu, logvar, eps, G, g_lq,
                           g_pm, (float*)(base + L.off_P), (float*)(base + L.off_lse), (float*)(base + L.off_dz0),
                           (float*)(base + L.off_z0));
    }
    const size_t lds_b = ((size_t)2 * MID_JC * G.DS + MID_JC * MID_TI) * sizeof(float);
    hipLaunchKernelGGL(encode_tail_mid_bwd_b_kernel, dim3(nwg), dim3(MID_THREADS), lds_b, st, mu, logvar, eps, G, g_z, g_lqx,
                       g_lpz, g_pm, pair_terms, (const float*)(base + L.off_P), (const float*)(base + L.off_lse),
                       (const float*)(base + L.off_dz0), (const float*)(base + L.off_z0), d_mu, d_lv);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
