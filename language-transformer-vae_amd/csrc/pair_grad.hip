// Backward of the aggregate-posterior estimators (approximate_log_q_z, loss_and_optimisation.py:139-221,
// under autograd in the reference's training objectives :925-1018) on the vector ALU (gfx950).
//
// With t_ijd the log-density of z_id under posterior (mu_jd, var_jd), v_ij = sum_d t_ijd,
//     log_qz_i   = LSE_j v_ij - log_norm            d log_qz_i / d v_ij  = p_ij  = exp(v_ij - LSE_j v_ij)
//     log_pm_i   = sum_d (LSE_j t_ijd - log_norm)   d log_pm_i / d t_ijd = r_ijd = exp(t_ijd - LSE_j t_ijd)
// and upstream gradients g_i (on log_qz_i) and h_i (on log_pm_i), the weight on t_ijd is
//     G_ijd = g_i p_ij + h_i r_ijd
//     d z_id      = - sum_j G_ijd (z_id - mu_jd) / var_jd
//     d mu_jd     = + sum_i G_ijd (z_id - mu_jd) / var_jd
//     d logvar_jd = 0.5 sum_i G_ijd ((z_id - mu_jd)^2 / var_jd - 1)
// Two sweeps of the same shape, one per kept axis: a lane owns one kept element (its D-vector and its
// gradient accumulators live in registers), the other axis is wave-uniform and streams through scalar
// loads.  The softmax weights are recomputed from the forward results (flash-attention style): nothing of
// size M x N is stored.  Everything is in the base-2 domain of the direct parameters
// (w = sqrt(0.5 log2e / var), b = -0.5 log2e (logvar + log 2pi), y = (z - mu) w, t2 = b - y^2):
//     (z - mu)/var = 2 y w / log2e,   (z - mu)^2/var = 2 y^2 / log2e.
// Sized for training-time batches (the reference trains at B = 32) up to a few thousand rows.  D <= 64 takes
// the register-resident sweeps below; wider latents take grad_wide_* (one thread per (kept element, dimension),
// the joint weights read from an explicit N x M matrix of pair log-densities -- training batches only).
#include "pair_plan.h"

#define GRAD_REC_EXTRA 4      // L2_i, g_i, h_i, pad  (sample record = [z (DC) | lse2 (DC) | these])

// Per-sample records for the posterior-kept sweep (uniform rows -> scalar loads need aligned, padded rows).
__global__ void __launch_bounds__(256)
grad_sample_records_kernel(const float* __restrict__ z, const float* __restrict__ log_qz, float log_norm,
                           const float* __restrict__ g, const float* __restrict__ h,
                           const float* __restrict__ lse2, int64_t M, int D, int DC, float* __restrict__ rec) {
    const int RS = 2 * DC + GRAD_REC_EXTRA;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * RS) return;
    const int64_t i = idx / RS;
    const int c = (int)(idx - i * RS);
    float v = 0.f;
    if (c < DC) v = c < D ? z[i * D + c] : 0.f;
    else if (c < 2 * DC) v = lse2 ? lse2[i * DC + (c - DC)] : 0.f;
    else if (c == 2 * DC) v = (log_qz ? log_qz[i] + log_norm : 0.f) * LVAE_LOG2E;
    else if (c == 2 * DC + 1) v = g ? g[i] : 0.f;
    else if (c == 2 * DC + 2) v = h ? h[i] : 0.f;
    rec[idx] = v;
}

// Per-(sample, dimension) base-2 logsumexp over the posteriors from the prod-marginals partials
// [dsplit][DC][cpad] -> lse2 [M][DC] (no log_norm: only differences t - lse are used).
__global__ void __launch_bounds__(256)
finalize_pm_lse_kernel(const float2* __restrict__ partial, int nsplit, int64_t cpad, int64_t C, int DC,
                       float* __restrict__ lse2) {
    // (a thread per (sample, dimension): at a training batch of 256 and D = 768 the per-sample loop over the
    //  dimensions was one workgroup walking 768 double-precision logarithms, 0.93 ms)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C) return;
    {
        const int d = blockIdx.y;
        float mx = -INFINITY;
        for (int k = 0; k < nsplit; ++k) mx = fmaxf(mx, partial[((int64_t)k * DC + d) * cpad + i].x);
        double s = 0.0;
        for (int k = 0; k < nsplit; ++k) {
            const float2 p = partial[((int64_t)k * DC + d) * cpad + i];
            if (p.y > 0.f) s += (double)p.y * exp2((double)p.x - (double)mx);
        }
        lse2[i * DC + d] = (float)((double)mx + log2(s));
    }
}

// Lane = sample i; loop over posteriors j of this split (uniform direct parameters [3][DC] per row).
// partial_dz [nsplit][M][DC].
template <int DCT, bool PM>
__global__ void __launch_bounds__(256)
pair_grad_sample_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                        const float* __restrict__ log_qz, float log_norm, const float* __restrict__ g,
                        const float* __restrict__ h, const float* __restrict__ lse2, int64_t M, int64_t N,
                        int D, int DC, int64_t rows_per_split, float* __restrict__ partial_dz,
                        const PairHeader* __restrict__ gate_hdr, float cond_limit) {
    if (gate_hdr && !(__uint_as_float(gate_hdr->cond_bits) > cond_limit)) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = col < M ? col : M - 1;
    float zr[DCT], dz[DCT], ls[PM ? DCT : 1];
#pragma unroll
    for (int d = 0; d < DCT; ++d) {
        zr[d] = d < D ? z[i * D + d] : 0.f;
        dz[d] = 0.f;
        if (PM) ls[d] = lse2[i * DC + d];
    }
    const float L2 = (log_qz ? log_qz[i] + log_norm : 0.f) * LVAE_LOG2E;
    const float gi = g ? g[i] : 0.f;
    const float hi = (PM && h) ? h[i] : 0.f;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > N) j1 = N;
    for (int64_t j = j0; j < j1; ++j) {
        const float* __restrict__ pm = dparams + (j * 3 + 0) * DC;
        const float* __restrict__ pw = dparams + (j * 3 + 1) * DC;
        const float* __restrict__ pb = dparams + (j * 3 + 2) * DC;
        float y[DCT], t[PM ? DCT : 1];
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            y[d] = (zr[d] - pm[d]) * pw[d];
            const float td = fmaf(-y[d], y[d], pb[d]);
            if (PM) t[d] = td;
            v += td;
        }
        const float p = gi * lvae_exp2(v - L2);
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            const float G = PM ? fmaf(hi, lvae_exp2(t[d] - ls[d]), p) : p;
            dz[d] = fmaf(-(G * pw[d]), y[d], dz[d]);
        }
    }
    if (col < M) {
        float* __restrict__ o = partial_dz + ((int64_t)blockIdx.y * M + col) * DC;
#pragma unroll
        for (int d = 0; d < DCT; ++d) o[d] = dz[d] * (2.0f / LVAE_LOG2E);
    }
}

// Lane = posterior j; loop over the sample records of this split (uniform).
// partial_dp [nsplit][N][2*DC]: d mu (DC) | d logvar (DC).
template <int DCT, bool PM>
__global__ void __launch_bounds__(256)
pair_grad_posterior_kernel(const float* __restrict__ dparams, const float* __restrict__ rec, int64_t M,
                           int64_t N, int DC, int64_t rows_per_split, float* __restrict__ partial_dp,
                           const PairHeader* __restrict__ gate_hdr, float cond_limit) {
    if (gate_hdr && !(__uint_as_float(gate_hdr->cond_bits) > cond_limit)) return;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t j = col < N ? col : N - 1;
    const int RS = 2 * DC + GRAD_REC_EXTRA;
    float mu[DCT], w[DCT], b[DCT], dmu[DCT], dlv[DCT];
#pragma unroll
    for (int d = 0; d < DCT; ++d) {
        mu[d] = dparams[(j * 3 + 0) * DC + d];
        w[d] = dparams[(j * 3 + 1) * DC + d];
        b[d] = dparams[(j * 3 + 2) * DC + d];
        dmu[d] = 0.f;
        dlv[d] = 0.f;
    }
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t i1 = i0 + rows_per_split;
    if (i1 > M) i1 = M;
    for (int64_t i = i0; i < i1; ++i) {
        const float* __restrict__ r = rec + i * RS;
        const float L2 = r[2 * DC], gi = r[2 * DC + 1], hi = r[2 * DC + 2];
        float y[DCT], t[PM ? DCT : 1];
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            y[d] = (r[d] - mu[d]) * w[d];
            const float td = fmaf(-y[d], y[d], b[d]);
            if (PM) t[d] = td;
            v += td;
        }
        const float p = gi * lvae_exp2(v - L2);
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            const float G = PM ? fmaf(hi, lvae_exp2(t[d] - r[DC + d]), p) : p;
            dmu[d] = fmaf(G * w[d], y[d], dmu[d]);
            dlv[d] = fmaf(G, fmaf(y[d] * (1.0f / LVAE_LOG2E), y[d], -0.5f), dlv[d]);
        }
    }
    if (col < N) {
        float* __restrict__ o = partial_dp + ((int64_t)blockIdx.y * N + col) * 2 * DC;
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            o[d] = dmu[d] * (2.0f / LVAE_LOG2E);
            o[DC + d] = dlv[d];
        }
    }
}

// ---- D > 64: one thread per (kept element, dimension) ----------------------------------------------------
// The joint weight p_ij needs the density over ALL dimensions, which a (sample, dimension) thread does not see:
// it is read from P [N][M], the explicit matrix of natural-log pair densities (pair_matrix_kernel, the layout of
// lvae_log_prob_pairs_f32).  The prod-marginals weight r_ijd is local to the dimension.  No partials, no atomics.
// Sample-side sweep, GW_JT samples per workgroup (the mirror image of grad_wide_posterior_shared_kernel below): every
// (m, w, b)_jd a thread loads serves GW_JT samples, the joint weights of a chunk of 256 posteriors come from LDS.
#define GW_JT 8
__global__ void __launch_bounds__(256)
grad_wide_sample_shared_kernel(const float* __restrict__ z, const float* __restrict__ dparams, const float* __restrict__ P,
                               const float* __restrict__ log_qz, float log_norm, const float* __restrict__ g,
                               const float* __restrict__ h, const float* __restrict__ lse2, int64_t M, int64_t N, int D,
                               int DC, float* __restrict__ d_z, const PairHeader* __restrict__ gate_hdr, float cond_limit) {
    if (gate_hdr && !(__uint_as_float(gate_hdr->cond_bits) > cond_limit)) return;
    __shared__ float sW[GW_JT][256];
    const int d = blockIdx.y * 256 + threadIdx.x;
    const int64_t ibase = (int64_t)blockIdx.x * GW_JT;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    float zi[GW_JT], hi[GW_JT], ls[GW_JT], gi[GW_JT], Li[GW_JT], acc[GW_JT];
#pragma unroll
    for (int q = 0; q < GW_JT; ++q) {
        const int64_t i = ibase + q < M ? ibase + q : M - 1;
        zi[q] = z[i * D + dd];
        hi[q] = h ? h[i] : 0.f;
        ls[q] = h ? lse2[i * DC + dd] : 0.f;
        gi[q] = g ? g[i] : 0.f;
        Li[q] = log_qz ? log_qz[i] + log_norm : 0.f;
        acc[q] = 0.f;
    }
    for (int64_t j0 = 0; j0 < N; j0 += 256) {
        const int64_t jt = j0 + threadIdx.x;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GW_JT; ++q) {
            const int64_t i = ibase + q < M ? ibase + q : M - 1;
            sW[q][threadIdx.x] = (g && jt < N) ? gi[q] * __expf(P[jt * M + i] - Li[q]) : 0.f;
        }
        __syncthreads();
        const int cnt = (int)(N - j0 < 256 ? N - j0 : 256);
#pragma unroll 2
        for (int t = 0; t < cnt; ++t) {
            const int64_t j = j0 + t;
            const float m = dparams[(j * 3 + 0) * DC + dd];
            const float w = dparams[(j * 3 + 1) * DC + dd];
            const float b = dparams[(j * 3 + 2) * DC + dd];
#pragma unroll
            for (int q = 0; q < GW_JT; ++q) {
                const float y = (zi[q] - m) * w;
                float G = sW[q][t];
                if (h) G = fmaf(hi[q], lvae_exp2(fmaf(-y, y, b) - ls[q]), G);
                acc[q] = fmaf(-(G * w), y, acc[q]);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int q = 0; q < GW_JT; ++q)
        if (ibase + q < M) d_z[(ibase + q) * D + d] = acc[q] * (2.0f / LVAE_LOG2E);
}

// The posterior-side sweep, GW_JT posteriors per workgroup: a thread owns one dimension of all of them, so every z_id and
// lse_id it loads serves GW_JT posteriors (with one posterior per workgroup every posterior re-read the whole of z and of
// the per-dimension logsumexps from L2: 103 GB at B = 4,096, D = 768, which was the kernel's time), and the joint
// weights g_i exp(P_ji - L_i) -- identical for every dimension -- are formed once per workgroup and 256 samples (thread
// t takes sample i0 + t) and read back from LDS together with h_i.
__global__ void __launch_bounds__(256)
grad_wide_posterior_shared_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                                  const float* __restrict__ P, const float* __restrict__ log_qz, float log_norm,
                                  const float* __restrict__ g, const float* __restrict__ h,
                                  const float* __restrict__ lse2, int64_t M, int64_t N, int D, int DC,
                                  float* __restrict__ d_mu, float* __restrict__ d_logvar,
                                  const PairHeader* __restrict__ gate_hdr, float cond_limit) {
    if (gate_hdr && !(__uint_as_float(gate_hdr->cond_bits) > cond_limit)) return;
    __shared__ float sW[GW_JT][256], sH[256];
    const int d = blockIdx.y * 256 + threadIdx.x;
    const int64_t jbase = (int64_t)blockIdx.x * GW_JT;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    float m[GW_JT], w[GW_JT], b[GW_JT], dmu[GW_JT], dlv[GW_JT];
#pragma unroll
    for (int jj = 0; jj < GW_JT; ++jj) {
        const int64_t j = jbase + jj < N ? jbase + jj : N - 1;
        m[jj] = dparams[(j * 3 + 0) * DC + dd];
        w[jj] = dparams[(j * 3 + 1) * DC + dd];
        b[jj] = dparams[(j * 3 + 2) * DC + dd];
        dmu[jj] = 0.f;
        dlv[jj] = 0.f;
    }
    for (int64_t i0 = 0; i0 < M; i0 += 256) {
        const int64_t it = i0 + threadIdx.x;
        __syncthreads();
        {
            const bool ok = g && it < M;
            const float gi = ok ? g[it] : 0.f, Li = ok ? log_qz[it] + log_norm : 0.f;
#pragma unroll
            for (int jj = 0; jj < GW_JT; ++jj) {
                const int64_t j = jbase + jj < N ? jbase + jj : N - 1;
                sW[jj][threadIdx.x] = ok ? gi * __expf(P[j * M + it] - Li) : 0.f;
            }
            sH[threadIdx.x] = (h && it < M) ? h[it] : 0.f;
        }
        __syncthreads();
        const int cnt = (int)(M - i0 < 256 ? M - i0 : 256);
#pragma unroll 2
        for (int t = 0; t < cnt; ++t) {
            const int64_t i = i0 + t;
            const float zi = z[i * D + dd];
            const float ls = h ? lse2[i * DC + dd] : 0.f;
            const float hi = sH[t];
#pragma unroll
            for (int jj = 0; jj < GW_JT; ++jj) {
                const float y = (zi - m[jj]) * w[jj];
                float G = sW[jj][t];
                if (h) G = fmaf(hi, lvae_exp2(fmaf(-y, y, b[jj]) - ls), G);
                dmu[jj] = fmaf(G * w[jj], y, dmu[jj]);
                dlv[jj] = fmaf(G, fmaf(y * (1.0f / LVAE_LOG2E), y, -0.5f), dlv[jj]);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int jj = 0; jj < GW_JT; ++jj) {
        const int64_t j = jbase + jj;
        if (j < N) {
            if (d_mu) d_mu[j * D + d] = dmu[jj] * (2.0f / LVAE_LOG2E);
            if (d_logvar) d_logvar[j * D + d] = dlv[jj];
        }
    }
}

int lvae_launch_grad_wide(const float* z, const float* dparams, const float* P, const float* log_qz, float log_norm,
                          const float* g, const float* h, const float* lse2, int64_t M, int64_t N, int D, int DC,
                          float* d_z, float* d_mu, float* d_logvar, hipStream_t st, const PairHeader* gate_hdr,
                          float cond_limit) {
    const unsigned dy = (unsigned)lvae_ceil_div(D, 256);
    if (d_z)
        hipLaunchKernelGGL(grad_wide_sample_shared_kernel, dim3((unsigned)lvae_ceil_div(M, GW_JT), dy), dim3(256), 0, st, z,
                           dparams, P, log_qz, log_norm, g, h, lse2, M, N, D, DC, d_z, gate_hdr, cond_limit);
    if (d_mu || d_logvar)
        hipLaunchKernelGGL(grad_wide_posterior_shared_kernel, dim3((unsigned)lvae_ceil_div(N, GW_JT), dy), dim3(256), 0, st, z, dparams, P, log_qz,
                           log_norm, g, h, lse2, M, N, D, DC, d_mu, d_logvar, gate_hdr, cond_limit);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// out[row][d] = sum_k partial[k][row][off + d]   (d < D; partial rows are `stride` floats wide)
__global__ void __launch_bounds__(256)
grad_reduce_kernel(const float* __restrict__ partial, int nsplit, int64_t rows, int stride, int off, int D,
                   float* __restrict__ out, const PairHeader* __restrict__ gate_hdr, float cond_limit) {
    if (gate_hdr && !(__uint_as_float(gate_hdr->cond_bits) > cond_limit)) return;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * D) return;
    const int64_t row = idx / D;
    const int d = (int)(idx - row * D);
    float acc = 0.f;
    for (int k = 0; k < nsplit; ++k) acc += partial[((int64_t)k * rows + row) * stride + off + d];
    out[idx] = acc;
}

// ---- host launchers ----------------------------------------------------------------------------------
int lvae_launch_finalize_pm_lse(const float2* partial, int nsplit, int64_t cpad, int64_t C, int DC,
                                float* lse2, hipStream_t st) {
    hipLaunchKernelGGL(finalize_pm_lse_kernel, dim3((unsigned)lvae_ceil_div(C, 256), (unsigned)DC), dim3(256), 0, st, partial,
                       nsplit, cpad, C, DC, lse2);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_grad_records(const float* z, const float* log_qz, float log_norm, const float* g,
                             const float* h, const float* lse2, int64_t M, int D, int DC, float* rec,
                             hipStream_t st) {
    const int64_t total = M * (2 * DC + GRAD_REC_EXTRA);
    hipLaunchKernelGGL(grad_sample_records_kernel, dim3((unsigned)lvae_ceil_div(total, 256)), dim3(256), 0, st,
                       z, log_qz, log_norm, g, h, lse2, M, D, DC, rec);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_grad_sample(const float* z, const float* dparams, const float* log_qz, float log_norm,
                            const float* g, const float* h, const float* lse2, int64_t M, int64_t N, int D,
                            int DC, int nsplit, int64_t rows_per_split, float* partial_dz, hipStream_t st,
                            const PairHeader* gate_hdr, float cond_limit) {
    const dim3 grid((unsigned)lvae_ceil_div(M, 256), (unsigned)nsplit);
    const bool pm = h != nullptr;
#define LVAE_GS(DCT, P)                                                                                 \
    hipLaunchKernelGGL((pair_grad_sample_kernel<DCT, P>), grid, dim3(256), 0, st, z, dparams, log_qz,    \
                       log_norm, g, h, lse2, M, N, D, DC, rows_per_split, partial_dz, gate_hdr, cond_limit)
    if (DC == 16) { if (pm) LVAE_GS(16, true); else LVAE_GS(16, false); }
    else if (DC == 32) { if (pm) LVAE_GS(32, true); else LVAE_GS(32, false); }
    else if (DC == 64) { if (pm) LVAE_GS(64, true); else LVAE_GS(64, false); }
    else return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the backward pass needs D <= 64 (got D=%d)", D);
#undef LVAE_GS
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_grad_posterior(const float* dparams, const float* rec, bool pm, int64_t M, int64_t N, int DC,
                               int nsplit, int64_t rows_per_split, float* partial_dp, hipStream_t st,
                               const PairHeader* gate_hdr, float cond_limit) {
    const dim3 grid((unsigned)lvae_ceil_div(N, 256), (unsigned)nsplit);
#define LVAE_GP(DCT, P)                                                                                 \
    hipLaunchKernelGGL((pair_grad_posterior_kernel<DCT, P>), grid, dim3(256), 0, st, dparams, rec, M, N, \
                       DC, rows_per_split, partial_dp, gate_hdr, cond_limit)
    if (DC == 16) { if (pm) LVAE_GP(16, true); else LVAE_GP(16, false); }
    else if (DC == 32) { if (pm) LVAE_GP(32, true); else LVAE_GP(32, false); }
    else if (DC == 64) { if (pm) LVAE_GP(64, true); else LVAE_GP(64, false); }
    else return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the backward pass needs D <= 64 (got DC=%d)", DC);
#undef LVAE_GP
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_grad_reduce(const float* partial, int nsplit, int64_t rows, int stride, int off, int D,
                            float* out, hipStream_t st, const PairHeader* gate_hdr, float cond_limit) {
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)lvae_ceil_div(rows * D, 256)), dim3(256), 0, st,
                       partial, nsplit, rows, stride, off, D, out, gate_hdr, cond_limit);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
