// Small kernels of the latent-analysis path that are not worth a matrix-core formulation at the sizes
// the reference runs them at: analytic KL (K9), Gaussian KDE scores (K8), explicit pair matrices
// (log_prob_pairs / gaussian_kernel outputs), the diagonal terms and final means of the MI bounds (K3),
// column means, and the MMD bookkeeping around the kernel-sum launches.
#include "pair_plan.h"

// ---- analytic KL and its free-bits variant (loss_and_optimisation.py:14-43) ---------------------
// kl = 0.5*(mu^2 + exp(logvar) - logvar - 1); hinge keeps entries with kl > lambda (:27-30).
__global__ void __launch_bounds__(256)
kl_diag_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, int64_t n, int D,
               float lambda, float* __restrict__ kl_elem, float* __restrict__ hinge_elem,
               float* __restrict__ kl_row, float* __restrict__ hinge_row) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    float sk = 0.f, sh = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float m = mu[row * D + d], lv = logvar[row * D + d];
        const float kl = 0.5f * (m * m + expf(lv) - lv - 1.0f);
        const float hk = lambda > 0.f ? (kl > lambda ? kl : 0.f) : kl;
        if (kl_elem) kl_elem[row * D + d] = kl;
        if (hinge_elem) hinge_elem[row * D + d] = hk;
        sk += kl;
        sh += hk;
    }
    sk = wave_sum(sk);
    sh = wave_sum(sh);
    if (lane == 0) {
        if (kl_row) kl_row[row] = sk;
        if (hinge_row) hinge_row[row] = sh;
    }
}

int lvae_launch_kl(const float* mu, const float* logvar, int64_t n, int D, float lambda, float* kl_elem,
                   float* hinge_elem, float* kl_row, float* hinge_row, hipStream_t st) {
    hipLaunchKernelGGL(kl_diag_kernel, dim3((unsigned)lvae_ceil_div(n, 4)), dim3(256), 0, st, mu, logvar, n,
                       D, lambda, kl_elem, hinge_elem, kl_row, hinge_row);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- column means of a [n, c] matrix ---------------------------------------------------------------
__global__ void __launch_bounds__(256)
colmean_stage1(const float* __restrict__ x, int64_t n, int64_t c, int rows_per_block,
               double* __restrict__ acc) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= c) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n) r1 = n;
    double s = 0.0;
    for (int64_t r = r0; r < r1; ++r) s += (double)x[r * c + col];
    atomicAdd(&acc[col], s);
}
__global__ void colmean_stage2(const double* __restrict__ acc, int64_t c, double scale, float* __restrict__ out) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col < c) out[col] = (float)(acc[col] * scale);
}

int lvae_launch_colmean(const float* x, int64_t n, int64_t c, float* out, double* acc, hipStream_t st) {
    LVAE_HIP(hipMemsetAsync(acc, 0, sizeof(double) * c, st));
    const int rows_per_block = 512;
    const dim3 grid((unsigned)lvae_ceil_div(c, 256), (unsigned)lvae_ceil_div(n, rows_per_block));
    hipLaunchKernelGGL(colmean_stage1, grid, dim3(256), 0, st, x, n, c, rows_per_block, acc);
    LVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(colmean_stage2, dim3((unsigned)lvae_ceil_div(c, 256)), dim3(256), 0, st, acc, c,
                       1.0 / (double)n, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- Gaussian KDE scores (utils_latent_analysis_optimisation.py:55-58) ---------------------------
// out[j] = log( bw^-d * sum_i exp(logN((X_i - Y_j)/bw; 0, I)) / n ), float32.  Lane = query, the fit points
// are wave-uniform (scalar loads).  Base-2 form: term_ij = exp2(c0 - sum_d ((x_id - y_jd) s)^2) with
// s = sqrt(0.5 log2e) / bw and c0 = -0.5 d log2(2 pi).
// stabilise == 0: plain sum like the reference (quirk Q6: a query far from every point underflows to
//                 log(0) = -inf);
// stabilise != 0: `ref` [m] holds max_i of the exponent (first pass, KDE_MAXPASS), the sum is taken relative
//                 to it and the result is finite for any query.
// d == 1 (get_1d_kde_dim_marg_kl_estimate, loss_and_optimisation.py:625-637) takes two fit points per
// packed fp32 instruction: 2 VALU slots + one v_exp_f32 per pair.
template <bool MAXPASS>
__global__ void __launch_bounds__(256)
kde_1d_kernel(const float* __restrict__ X, const float* __restrict__ Y, int64_t n, int64_t m, float bw,
              int64_t rows_per_split, const float* __restrict__ ref, float* __restrict__ partial) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t jj = j < m ? j : m - 1;
    const float s = sqrtf(0.5f * LVAE_LOG2E) / bw;
    const float c0 = -0.5f * LVAE_LOG_2PI * LVAE_LOG2E - (ref ? ref[jj] : 0.f);
    const f32x2 y2 = {Y[jj], Y[jj]}, s2 = {s, s}, c2 = {c0, c0};
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_split;      // multiple of 8
    int64_t i1 = i0 + rows_per_split;
    if (i1 > n) i1 = n;
    f32x2 acc = MAXPASS ? (f32x2){-INFINITY, -INFINITY} : (f32x2){0.f, 0.f};
    int64_t i = i0;
    for (; i + 8 <= i1; i += 8) {
        const f32x2* __restrict__ xp = reinterpret_cast<const f32x2*>(X + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 u = (xp[k] - y2) * s2;
            const f32x2 t = __builtin_elementwise_fma(-u, u, c2);
            if (MAXPASS) {
                acc.x = fmaxf(acc.x, t.x);
                acc.y = fmaxf(acc.y, t.y);
            } else {
                f32x2 e;
                e.x = lvae_exp2(t.x);
                e.y = lvae_exp2(t.y);
                acc += e;
            }
        }
    }
    float a = MAXPASS ? fmaxf(acc.x, acc.y) : acc.x + acc.y;
    for (; i < i1; ++i) {
        const float u = (X[i] - y2.x) * s;
        const float t = fmaf(-u, u, c0);
        a = MAXPASS ? fmaxf(a, t) : a + lvae_exp2(t);
    }
    if (j < m) partial[(int64_t)blockIdx.y * m + j] = a;
}

// Any d: one exponential per pair.
template <bool MAXPASS>
__global__ void __launch_bounds__(256)
kde_nd_kernel(const float* __restrict__ X, const float* __restrict__ Y, int64_t n, int64_t m, int d, float bw,
              int64_t rows_per_split, const float* __restrict__ ref, float* __restrict__ partial) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t jj = j < m ? j : m - 1;
    const float s = sqrtf(0.5f * LVAE_LOG2E) / bw;
    const float c0 = -0.5f * (float)d * LVAE_LOG_2PI * LVAE_LOG2E - (ref ? ref[jj] : 0.f);
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t i1 = i0 + rows_per_split;
    if (i1 > n) i1 = n;
    float a = MAXPASS ? -INFINITY : 0.f;
    for (int64_t i = i0; i < i1; ++i) {
        float t = c0;
        for (int dd = 0; dd < d; ++dd) {
            const float u = (X[i * d + dd] - Y[jj * d + dd]) * s;
            t = fmaf(-u, u, t);
        }
        a = MAXPASS ? fmaxf(a, t) : a + lvae_exp2(t);
    }
    if (j < m) partial[(int64_t)blockIdx.y * m + j] = a;
}

// mode 0: ref[j] = max_k partial[k][j] - c0   (the exponent maximum without the constant, for the second pass)
// mode 1: out[j] = ln( bw^-d * sum_k partial[k][j] / n ) (+ ref[j] ln2 when stabilised)
__global__ void __launch_bounds__(256)
kde_finalize_kernel(const float* __restrict__ partial, int nsplit, int64_t m, int mode, float log_scale_over_n,
                    float c0, const float* __restrict__ ref_in, float* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    if (mode == 0) {
        float mx = -INFINITY;
        for (int k = 0; k < nsplit; ++k) mx = fmaxf(mx, partial[(int64_t)k * m + j]);
        out[j] = mx - c0;
    } else {
        double sacc = 0.0;
        for (int k = 0; k < nsplit; ++k) sacc += (double)partial[(int64_t)k * m + j];
        const float sum32 = (float)sacc;
        out[j] = logf(sum32) + log_scale_over_n + (ref_in ? ref_in[j] * (float)LVAE_LN2 : 0.f);
    }
}

static int kde_pass(bool maxpass, const float* X, const float* Y, int64_t n, int64_t m, int d, float bw,
                    int nsplit, int64_t rows, const float* ref, float* partial, hipStream_t st) {
    const dim3 grid((unsigned)lvae_ceil_div(m, 256), (unsigned)nsplit);
    if (d == 1) {
        if (maxpass) hipLaunchKernelGGL(kde_1d_kernel<true>, grid, dim3(256), 0, st, X, Y, n, m, bw, rows, ref, partial);
        else hipLaunchKernelGGL(kde_1d_kernel<false>, grid, dim3(256), 0, st, X, Y, n, m, bw, rows, ref, partial);
    } else {
        if (maxpass) hipLaunchKernelGGL(kde_nd_kernel<true>, grid, dim3(256), 0, st, X, Y, n, m, d, bw, rows, ref, partial);
        else hipLaunchKernelGGL(kde_nd_kernel<false>, grid, dim3(256), 0, st, X, Y, n, m, d, bw, rows, ref, partial);
    }
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// workspace: partial [nsplit][m] floats + ref [m] floats
void lvae_kde_plan(int64_t n, int64_t m, int* nsplit, int64_t* rows) {
    const int64_t waves = lvae_ceil_div(m, 64);
    int64_t ns = lvae_ceil_div(4096, waves);
    const int64_t cap = lvae_ceil_div(n, 64);
    if (ns > cap) ns = cap;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    int64_t r = lvae_round_up(lvae_ceil_div(n, ns), 8);
    *rows = r;
    *nsplit = (int)lvae_ceil_div(n, r);
}

int lvae_launch_kde(const float* X, const float* Y, int64_t n, int64_t n_norm, int64_t m, int d, float bw, int stabilise,
                    float* out, float* ws, hipStream_t st) {
    int nsplit;
    int64_t rows;
    lvae_kde_plan(n, m, &nsplit, &rows);
    float* partial = ws;
    float* ref = ws + (size_t)nsplit * m;
    const float c0 = -0.5f * (float)d * LVAE_LOG_2PI * LVAE_LOG2E;
    const float log_scale_over_n = -(float)d * logf(bw) - logf((float)n_norm);
    int rc;
    if (stabilise) {
        rc = kde_pass(true, X, Y, n, m, d, bw, nsplit, rows, nullptr, partial, st);
        if (rc) return rc;
        hipLaunchKernelGGL(kde_finalize_kernel, dim3((unsigned)lvae_ceil_div(m, 256)), dim3(256), 0, st, partial,
                           nsplit, m, 0, 0.f, c0, nullptr, ref);
        LVAE_LAUNCH_CHECK();
    }
    rc = kde_pass(false, X, Y, n, m, d, bw, nsplit, rows, stabilise ? ref : nullptr, partial, st);
    if (rc) return rc;
    hipLaunchKernelGGL(kde_finalize_kernel, dim3((unsigned)lvae_ceil_div(m, 256)), dim3(256), 0, st, partial,
                       nsplit, m, 1, log_scale_over_n, c0, stabilise ? ref : nullptr, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- explicit pair matrices ---------------------------------------------------------------------------
// mode 0: out[j, i] = log N(b_i; a_j, diag(var_j))          (log_prob_pairs, mutual_information.py:108-132;
//                                                             _gaussian_log_likelihood full branch :41)
// mode 1: out[j, i] = sum_k exp(-alpha_k * (eps + |a_j - b_i|^2))   (gaussian_kernel / MMD ret_matrix)
__global__ void __launch_bounds__(256)
pair_matrix_kernel(const float* __restrict__ a, const float* __restrict__ spread, int spread_kind,
                   const float* __restrict__ b, int64_t na, int64_t nb, int D, int mode,
                   const PairHeader* __restrict__ hdr, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int64_t j = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    if (mode == 0 && D > 32) {
        // Rows wider than 32: the 16 x 16 tile's operands travel through LDS 64 dimensions at a time, read coalesced,
        // with the variance, its logarithm and its reciprocal formed ONCE per (j, d) instead of once per (i, j, d) --
        // a thread walking its own pair of rows with an exponential, a logarithm and a division per element took
        // 124 us for a 32 x 32 matrix at D = 768.
        __shared__ float sa[16][65], sb[16][65], si[16][65], sl[16][65];
        __shared__ float slv[16];
        const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
        float acc = 0.f, lvsum = 0.f;
        for (int d0 = 0; d0 < D; d0 += 64) {
            __syncthreads();
            for (int e = tid; e < 16 * 64; e += 256) {
                const int r = e >> 6, c = e & 63, d = d0 + c;
                const int64_t jj = (int64_t)blockIdx.y * 16 + r, ii = (int64_t)blockIdx.x * 16 + r;
                float av = 0.f, iv = 0.f, lv = 0.f, bv = 0.f;
                if (d < D && jj < na) {
                    const float sp = spread[jj * D + d];
                    const float var = spread_kind == LVAE_SPREAD_VAR ? sp : expf(sp);
                    lv = (spread_kind == LVAE_SPREAD_VAR ? logf(sp) : sp) + LVAE_LOG_2PI;
                    iv = 1.0f / var;
                    av = a[jj * D + d];
                }
                if (d < D && ii < nb) bv = b[ii * D + d];
                sa[r][c] = av;
                si[r][c] = iv;
                sl[r][c] = lv;
                sb[r][c] = bv;
            }
            __syncthreads();
            if (tid < 16) {                        // the chunk's share of sum_d (log var + log 2 pi), one thread per j
                float t = 0.f;
                for (int c = 0; c < 64; ++c) t += sl[tid][c];
                slv[tid] = t;
            }
            __syncthreads();
            lvsum += slv[tj];
#pragma unroll 8
            for (int c = 0; c < 64; ++c) {
                const float df = sb[ti][c] - sa[tj][c];
                acc = fmaf(df * df, si[tj][c], acc);
            }
        }
        if (i < nb && j < na) out[j * nb + i] = -0.5f * (acc + lvsum);
        return;
    }
    if (i >= nb || j >= na) return;
    if (mode == 0) {
        float acc = 0.f;
        for (int d = 0; d < D; ++d) {
            const float sp = spread[j * D + d];
            const float var = spread_kind == LVAE_SPREAD_VAR ? sp : expf(sp);
            const float lv = spread_kind == LVAE_SPREAD_VAR ? logf(sp) : sp;
            const float df = b[i * D + d] - a[j * D + d];
            acc += lv + LVAE_LOG_2PI + __fdiv_rn(df * df, var);
        }
        out[j * nb + i] = -0.5f * acc;
    } else {
        float d2 = 0.f;
        for (int d = 0; d < D; ++d) {
            const float df = a[j * D + d] - b[i * D + d];
            d2 += df * df;
        }
        const KsumParams* kp = &hdr->ksum;
        const float v = fabsf(d2) + kp->eps;
        float f = 0.f;
        if (kp->poly) {
            const float t = exp2f(v * kp->nbase);
            f = kp->c[kp->n];
            for (int k = kp->n - 1; k >= 0; --k) f = fmaf(f, t, kp->c[k]);
        } else {
            for (int k = 0; k < kp->n; ++k) f += exp2f(v * kp->c[k]);
        }
        out[j * nb + i] = f;
    }
}

int lvae_launch_pair_matrix(const float* a, const float* spread, int spread_kind, const float* b,
                            int64_t na, int64_t nb, int D, int mode, const PairHeader* hdr, float* out,
                            hipStream_t st) {
    const dim3 grid((unsigned)lvae_ceil_div(nb, 16), (unsigned)lvae_ceil_div(na, 16));
    hipLaunchKernelGGL(pair_matrix_kernel, grid, dim3(256), 0, st, a, spread, spread_kind, b, na, nb, D, mode,
                       hdr, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- MI bounds: diagonal terms and final means ---------------------------------------------------------
// diag[i] = log N(z_i; mu_i, diag(var_i)); lower = mean(diag - lse_all), upper = mean(diag - lse_off)
// (mutual_information.py:165-171, :204-213; the -log K / -log(K-1) are already in the lse vectors).
__global__ void __launch_bounds__(256)
diag_ll_kernel(const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ spread,
               int spread_kind, int64_t n, int D, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        const float sp = spread[i * D + d];
        const float var = spread_kind == LVAE_SPREAD_VAR ? sp : expf(sp);
        const float lv = spread_kind == LVAE_SPREAD_VAR ? logf(sp) : sp;
        const float df = z[i * D + d] - mu[i * D + d];
        acc += lv + LVAE_LOG_2PI + __fdiv_rn(df * df, var);
    }
    out[i] = -0.5f * acc;
}

__global__ void __launch_bounds__(256)
bounds_finalize_kernel(const float* __restrict__ diag, const float* __restrict__ lse_all,
                       const float* __restrict__ lse_off, int64_t n, double* __restrict__ out) {
    double lo = 0.0, up = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        lo += (double)diag[i] - (double)lse_all[i];
        up += (double)diag[i] - (double)lse_off[i];
    }
    lo = wave_sum_d(lo);
    up = wave_sum_d(up);
    __shared__ double sm[2][4];
    if ((threadIdx.x & 63) == 0) {
        sm[0][threadIdx.x >> 6] = lo;
        sm[1][threadIdx.x >> 6] = up;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3]) / (double)n;
        out[1] = (sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3]) / (double)n;
    }
}

int lvae_launch_diag_ll(const float* z, const float* mu, const float* spread, int spread_kind, int64_t n,
                        int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(diag_ll_kernel, dim3((unsigned)lvae_ceil_div(n, 256)), dim3(256), 0, st, z, mu, spread,
                       spread_kind, n, D, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_bounds_finalize(const float* diag, const float* lse_all, const float* lse_off, int64_t n,
                                double* out, hipStream_t st) {
    hipLaunchKernelGGL(bounds_finalize_kernel, dim3(1), dim3(256), 0, st, diag, lse_all, lse_off, n, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- MMD bookkeeping -----------------------------------------------------------------------------------
__global__ void set_ksum_kernel(PairHeader* hdr, const KsumParams kp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) hdr->ksum = kp;
}

// sums: [Sxx, Syy, Sxy] of the kernels with alpha > 0; zero_alphas bandwidths with alpha == 0 add
// exactly 1 per counted pair.  out = [stat, Sxx, Syy, Sxy] (totals).
__global__ void mmd_stat_kernel(const double* __restrict__ sums, double n1, double n2, int unbiased,
                                double zero_alphas, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double c11 = unbiased ? n1 * (n1 - 1.0) : n1 * n1;
    const double c22 = unbiased ? n2 * (n2 - 1.0) : n2 * n2;
    const double c12 = n1 * n2;
    const double sxx = sums[0] + zero_alphas * c11;
    const double syy = sums[1] + zero_alphas * c22;
    const double sxy = sums[2] + zero_alphas * c12;
    out[0] = sxx / c11 + syy / c22 - 2.0 * sxy / c12;
    out[1] = sxx;
    out[2] = syy;
    out[3] = sxy;
}

// The same with the header cleared first and `nzero` float64 accumulators zeroed: the first launch of the fused MMD
// call (no separate fills).
__global__ void __launch_bounds__(256) ksum_init_kernel(PairHeader* hdr, const KsumParams kp, double* zero, int nzero) {
    unsigned* w = reinterpret_cast<unsigned*>(hdr);
    for (int i = threadIdx.x; i < (int)(sizeof(PairHeader) / sizeof(unsigned)); i += 256) w[i] = 0u;
    if ((int)threadIdx.x < nzero) zero[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) hdr->ksum = kp;
}

int lvae_launch_ksum_init(PairHeader* hdr, const KsumParams& kp, double* zero, int nzero, hipStream_t st) {
    hipLaunchKernelGGL(ksum_init_kernel, dim3(1), dim3(256), 0, st, hdr, kp, zero, nzero);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_set_ksum(PairHeader* hdr, const KsumParams& kp, hipStream_t st) {
    hipLaunchKernelGGL(set_ksum_kernel, dim3(1), dim3(64), 0, st, hdr, kp);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_mmd_stat(const double* sums, int64_t n1, int64_t n2, int unbiased, int zero_alphas,
                         double* out, hipStream_t st) {
    hipLaunchKernelGGL(mmd_stat_kernel, dim3(1), dim3(64), 0, st, sums, (double)n1, (double)n2, unbiased,
                       (double)zero_alphas, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- MMD permutation test -----------------------------------------------------------------------------
// 0/1 labels [P][n] (one row per labeling) -> bf16 fragments in the order the PERM mode of the MFMA pair
// kernel consumes them as B operands of v_mfma_f32_32x32x16_bf16: per 32-row block
// [4 labeling tiles][2 steps][64 lanes][8 values]; lane = labeling (low 5 bits) + 32 * k-half; value e of
// step s belongs to the row the accumulator layout puts in register 8 s + e of that half:
// (reg & 3) + 8 (reg >> 2) + 4 half.
__global__ void __launch_bounds__(256) prep_labels_kernel(const unsigned char* __restrict__ labels, int P,
                                                          int64_t n, int64_t nblk, uint4* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nblk * 512) return;
    const int lane = (int)(idx & 63);
    const int sp = (int)((idx >> 6) & 1);
    const int pt = (int)((idx >> 7) & 3);
    const int64_t blk = idx >> 9;
    const int perm = 32 * pt + (lane & 31);
    const int kh = lane >> 5;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (perm < P) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int reg = 8 * sp + e;
            const int64_t row = blk * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            if (row < n && labels[(int64_t)perm * n + row]) w[e >> 1] |= 0x3F80u << (16 * (e & 1));   // bf16 1.0
        }
    }
    out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Block p < P: lr_p = sum_i label[p][col_offset + i] * cs_i (+ ysum_p when sym); block P: T.
// cs_i = sum_k partial[k][i]: the kernel sums of kept element i over the visited rows -- all of the pooled
// sample, or with sym the rows beyond its diagonal tile, whose own sums are in diag_partial; then the
// missing half of lr_p is ysum_p = sum_c (K L_p)[c] over the visited tiles and T = 2 sum cs + sum diag.
// out = [q_0..q_{P-1}, lr_0..lr_{P-1}, T].
__global__ void __launch_bounds__(256) perm_lr_kernel(const double* __restrict__ partial,
                                                      const double* __restrict__ diag_partial, int nsplit,
                                                      int64_t cpad, int64_t C,
                                                      const unsigned char* __restrict__ labels, int64_t n,
                                                      int64_t col_offset, int P, int sym,
                                                      const double* __restrict__ qperm,
                                                      double* __restrict__ out) {
    const int p = blockIdx.x;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < C; i += 256) {
        if (p == P || labels[(int64_t)p * n + col_offset + i]) {
            double cs = 0.0, dg = 0.0;
            for (int k = 0; k < nsplit; ++k) {
                cs += partial[k * cpad + i];
                if (sym) dg += diag_partial[k * cpad + i];
            }
            acc += (p == P && sym) ? 2.0 * cs + dg : cs;
        }
    }
    acc = wave_sum_d(acc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = sm[0] + sm[1] + sm[2] + sm[3];
        if (p == P) out[2 * P] = tot;
        else {
            out[P + p] = tot + (sym ? qperm[128 + p] : 0.0);
            out[p] = qperm[p];
        }
    }
}

int lvae_launch_prep_labels(const unsigned char* labels, int P, int64_t n, int64_t nblk, void* out,
                            hipStream_t st) {
    const int64_t total = nblk * 512;
    hipLaunchKernelGGL(prep_labels_kernel, dim3((unsigned)lvae_ceil_div(total, 256)), dim3(256), 0, st, labels,
                       P, n, nblk, (uint4*)out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_perm_lr(const double* partial, const double* diag_partial, int nsplit, int64_t cpad, int64_t C,
                        const unsigned char* labels, int64_t n, int64_t col_offset, int P, int sym,
                        const double* qperm, double* out, hipStream_t st) {
    hipLaunchKernelGGL(perm_lr_kernel, dim3((unsigned)(P + 1)), dim3(256), 0, st, partial, diag_partial, nsplit,
                       cpad, C, labels, n, col_offset, P, sym, qperm, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- kernel-matrix sums for D > 128 (no MFMA variant: the pair kernel keeps a whole operand row in VGPRs) ---
// Direct differences, 32 x 32 tiles of pairs staged through LDS 32 dimensions at a time; a workgroup owns 32
// kept rows and a contiguous share of the reduced-axis tiles, and writes one float64 partial per kept row
// into the same [nsplit][cpad] layout the MFMA kernel-sum epilogues use (ksum_columns_kernel merges them).
// Sized for training-time batches with wide latents (the MMD of the reference's encode tail); O(na nb D) on
// the vector ALU.
__device__ __forceinline__ float ksum_eval(const KsumParams& kp, float d2) {
    const float v = fabsf(d2) + kp.eps;
    if (kp.poly) {
        const float t = lvae_exp2(v * kp.nbase);
        float acc = kp.c[kp.n];
        for (int k = kp.n - 1; k >= 0; --k) acc = fmaf(acc, t, kp.c[k]);
        return acc;
    }
    float f = 0.f;
    for (int k = 0; k < kp.n; ++k) f += lvae_exp2(v * kp.c[k]);
    return f;
}

__global__ void __launch_bounds__(256)
ksum_wide_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t na, int64_t nb, int D,
                 const PairHeader* __restrict__ hdr, int excl, int64_t excl_offset, int64_t tiles_per_split,
                 double* __restrict__ partial, int64_t cpad) {
    __shared__ float sa[32][33], sb[32][33];
    __shared__ KsumParams kp;
    if (threadIdx.x == 0) kp = hdr->ksum;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int64_t nbt = (nb + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.y * tiles_per_split;
    int64_t t1 = t0 + tiles_per_split;
    if (t1 > nbt) t1 = nbt;
    double rowacc[4] = {0.0, 0.0, 0.0, 0.0};
    __syncthreads();
    for (int64_t bt = t0; bt < t1; ++bt) {
        const int64_t j0 = bt * 32;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d0 = 0; d0 < D; d0 += 32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ty * 4 + r;
                const bool din = d0 + tx < D;
                sa[row][tx] = (din && i0 + row < na) ? a[(i0 + row) * D + d0 + tx] : 0.f;
                sb[row][tx] = (din && j0 + row < nb) ? b[(j0 + row) * D + d0 + tx] : 0.f;
            }
            __syncthreads();
#pragma unroll 8
            for (int dd = 0; dd < 32; ++dd) {
                const float bv = sb[tx][dd];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float df = sa[ty * 4 + r][dd] - bv;
                    acc[r] = fmaf(df, df, acc[r]);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + ty * 4 + r, j = j0 + tx;
            if (i < na && j < nb && !(excl && j == i + excl_offset)) rowacc[r] += (double)ksum_eval(kp, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = rowacc[r];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);     // the 32 lanes that share ty
        const int64_t i = i0 + ty * 4 + r;
        if (tx == 0 && i < cpad) partial[(int64_t)blockIdx.y * cpad + i] = v;
    }
}

int lvae_launch_ksum_wide(const float* a, const float* b, int64_t na, int64_t nb, int D, const PairHeader* hdr,
                          bool excl, int64_t excl_offset, int nsplit, int64_t cpad, double* partial,
                          hipStream_t st) {
    const int64_t nbt = lvae_ceil_div(nb, 32);
    const int64_t tps = lvae_ceil_div(nbt, nsplit);
    const dim3 grid((unsigned)lvae_ceil_div(cpad, 32), (unsigned)nsplit);
    hipLaunchKernelGGL(ksum_wide_kernel, grid, dim3(256), 0, st, a, b, na, nb, D, hdr, excl ? 1 : 0, excl_offset, tps,
                       partial, cpad);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- _gaussian_log_likelihood in full generality (mutual_information.py:26-41) ---------------------------
// The element-wise term of the branch named by which of mu / var is given, over the BROADCAST shape of
// (sample, mu, var), times a mask that was unsqueezed at `dim`, summed over `dim`.  Operands come as element
// strides over the broadcast shape (0 on broadcast axes), nothing is materialised.  One lane per output
// element (the broadcast shape without `dim`, row-major); sized for the small tensors the reference hands it.
#define LVAE_GLL_MAX_DIMS 6
struct GllArgs {
    int ndim, dim;
    int64_t shape[LVAE_GLL_MAX_DIMS];
    int64_t st_s[LVAE_GLL_MAX_DIMS], st_mu[LVAE_GLL_MAX_DIMS], st_var[LVAE_GLL_MAX_DIMS], st_mask[LVAE_GLL_MAX_DIMS];
};

__global__ void __launch_bounds__(256) gauss_ll_masked_kernel(const float* __restrict__ sample,
                                                              const float* __restrict__ mu,
                                                              const float* __restrict__ var,
                                                              const float* __restrict__ mask, const GllArgs a,
                                                              int64_t n_out, float* __restrict__ out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    int64_t rem = o, os = 0, om = 0, ov = 0, ok = 0;
    for (int k = a.ndim - 1; k >= 0; --k) {
        if (k == a.dim) continue;
        const int64_t i = rem % a.shape[k];
        rem /= a.shape[k];
        os += i * a.st_s[k];
        om += i * a.st_mu[k];
        ov += i * a.st_var[k];
        ok += i * a.st_mask[k];
    }
    const int64_t L = a.shape[a.dim];
    float acc = 0.f;
    for (int64_t i = 0; i < L; ++i) {
        const float s = sample[os + i * a.st_s[a.dim]];
        const float c = mu ? s - mu[om + i * a.st_mu[a.dim]] : s;
        float t;
        if (var) {
            const float v = var[ov + i * a.st_var[a.dim]];
            t = logf(6.283185307179586f * v) + c * c / v;          // :36, :41
        } else {
            t = LVAE_LOG_2PI + c * c;                              // :34, :38
        }
        acc += t * mask[ok + i * a.st_mask[a.dim]];
    }
    out[o] = -0.5f * acc;
}

int lvae_launch_gauss_ll_masked(const float* sample, const float* mu, const float* var, const float* mask, int ndim,
                                const int64_t* shape, const int64_t* st_s, const int64_t* st_mu,
                                const int64_t* st_var, const int64_t* st_mask, int dim, int64_t n_out, float* out,
                                hipStream_t st) {
    GllArgs a{};
    a.ndim = ndim;
    a.dim = dim;
    for (int k = 0; k < ndim; ++k) {
        a.shape[k] = shape[k];
        a.st_s[k] = st_s[k];
        a.st_mu[k] = mu ? st_mu[k] : 0;
        a.st_var[k] = var ? st_var[k] : 0;
        a.st_mask[k] = st_mask[k];
    }
    hipLaunchKernelGGL(gauss_ll_masked_kernel, dim3((unsigned)lvae_ceil_div(n_out, 256)), dim3(256), 0, st, sample,
                       mu, var, mask, a, n_out, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
