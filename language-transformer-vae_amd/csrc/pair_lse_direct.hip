// Direct-form pair log-density + streaming logsumexp on the vector ALU (gfx950).
//
// One lane per kept element (sample z_i, its latent row in VGPRs); the reduced axis (posterior j) is
// wave-uniform, so its parameters (mu_j, w_j = sqrt(0.5*log2e/var_j), b_j = -0.5*log2e*(log var_j +
// log 2pi)) are read through uniform loads shared by the whole wave.  Per (i, j, d):
//     y = (z_id - mu_jd) * w_jd ;  t = b_jd - y*y            (base-2 log-density of one dimension)
// joint : v_ij = sum_d t, folded into the lane's running (max, sum)        -> log q(z_i)
// pm    : every (i, d) keeps its own sum over j                            -> prod-of-marginals term
//         (loss_and_optimisation.py:195: logsumexp of the [M,N,D] tensor over the posterior axis).
// This form has no cancellation (it is the reference's own formula), so it also serves as the
// fall-back of the MFMA path for badly conditioned inputs (peaked posteriors far from the origin) and
// for latent sizes the MFMA path does not cover.  Bound: VALU / transcendental rate (one v_exp_f32
// per (i,j,d) term with PM), not HBM.
#include "pair_plan.h"
#include <cstdlib>

#define DLSE_THR 8.0f
#define DLSE_MINIT (-1.0e30f)

template <int DCT>
__global__ void __launch_bounds__(256)
pair_direct_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                   float2* __restrict__ partial, const PairHeader* __restrict__ hdr, float cond_limit,
                   int use_if_above, int64_t C, int64_t cpad, int64_t R, int64_t rows_per_split, int D,
                   int DC, int excl, int64_t excl_offset) {
    if (cond_limit >= 0.f) {
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = col < C ? col : C - 1;

    float zr[DCT];
#pragma unroll
    for (int d = 0; d < DCT; ++d) zr[d] = d < D ? z[i * D + d] : 0.f;

    float mv = DLSE_MINIT, sv = 0.f;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > R) j1 = R;
    const int64_t jskip = excl ? i + excl_offset : -1;

    for (int64_t j = j0; j < j1; ++j) {
        const float* __restrict__ pm = dparams + (j * 3 + 0) * DC;
        const float* __restrict__ pw = dparams + (j * 3 + 1) * DC;
        const float* __restrict__ pb = dparams + (j * 3 + 2) * DC;
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            const float y = (zr[d] - pm[d]) * pw[d];
            v += fmaf(-y, y, pb[d]);
        }
        if (j == jskip) v = -INFINITY;
        const float xv = v - mv;
        if (__any(xv > DLSE_THR)) {
            if (xv > DLSE_THR) {
                sv = sv * lvae_exp2(-xv) + 1.0f;
                mv = v;
            } else {
                sv += lvae_exp2(xv);
            }
        } else {
            sv += lvae_exp2(xv);
        }
    }
    if (col < C) partial[(int64_t)blockIdx.y * cpad + col] = make_float2(mv, sv);
}

// Product-of-marginals sweep: every (sample i, dimension d) keeps its own sum over the posteriors j of
// exp2(t_ijd).  Fast pass: reference 0 (t <= b_jd = -0.5*log2(2 pi var), a few bits either side of 0
// for any sane variance), two dimensions per packed fp32 instruction -- per term 2.5 VALU slots and one
// v_exp_f32.  A wave in which any sum left the range [2^-60, 2^100] (a sample ~9 sigma away from every
// posterior of the split, or absurd variances) raises its redo flag instead of storing, and
// pair_pm_online_kernel repeats that wave's sweep with the running-maximum form.
#define PM_SUM_MIN 8.6736174e-19f   // 2^-60
#define PM_SUM_MAX 1.2676506e30f    // 2^100
template <int DCT>
__global__ void __launch_bounds__(256)
pair_pm_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
               float2* __restrict__ partial_pm, int* __restrict__ redo_flags, int64_t C, int64_t cpad,
               int64_t R, int64_t rows_per_split, int D, int DC, int d0_base,
               const PairHeader* __restrict__ hdr, float cond_limit) {
    // cond_limit >= 0: the matrix-pipe form (pair_pm_bx_kernel) covers well-conditioned inputs, this one the rest
    if (cond_limit >= 0.f && !(__uint_as_float(hdr->cond_bits) > cond_limit)) return;
    // blockIdx.z: the DCT-wide chunk of the latent row (all chunks of a wide row in ONE launch: D = 768 used to be
    // 24 launches of a grid that a training batch does not fill)
    const int d0 = d0_base + (int)blockIdx.z * DCT;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = col < C ? col : C - 1;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > R) j1 = R;

    f32x2 zr[DCT / 2], s2[DCT / 2];
#pragma unroll
    for (int h = 0; h < DCT / 2; ++h) {
        zr[h].x = (d0 + 2 * h < D) ? z[i * D + d0 + 2 * h] : 0.f;
        zr[h].y = (d0 + 2 * h + 1 < D) ? z[i * D + d0 + 2 * h + 1] : 0.f;
        s2[h] = (f32x2){0.f, 0.f};
    }
    // posterior parameters travel through scalar registers (wave-uniform loads)
    const float* __restrict__ pj = dparams + j0 * 3 * DC + d0;
    for (int64_t j = j0; j < j1; ++j, pj += 3 * DC) {
        const f32x2* __restrict__ pm = reinterpret_cast<const f32x2*>(pj);
        const f32x2* __restrict__ pw = reinterpret_cast<const f32x2*>(pj + DC);
        const f32x2* __restrict__ pb = reinterpret_cast<const f32x2*>(pj + 2 * DC);
        // four independent pairs per stage, so that no instruction waits on the one before it
#pragma unroll
        for (int hb = 0; hb < DCT / 2; hb += 4) {
            f32x2 y[4], e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = zr[hb + k] - pm[hb + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] *= pw[hb + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = __builtin_elementwise_fma(-y[k], y[k], pb[hb + k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                e[k].x = lvae_exp2(y[k].x);
                e[k].y = lvae_exp2(y[k].y);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) s2[hb + k] += e[k];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    bool ok = true;
#pragma unroll
    for (int h = 0; h < DCT / 2; ++h)
        ok = ok && s2[h].x >= PM_SUM_MIN && s2[h].x <= PM_SUM_MAX && s2[h].y >= PM_SUM_MIN &&
             s2[h].y <= PM_SUM_MAX;
    const bool redo = !__all(ok);
    if ((threadIdx.x & 63) == 0)
        redo_flags[((int64_t)blockIdx.y * (DC / 16) + d0 / 16) * (cpad / 64) + col / 64] = redo ? 1 : 0;
    if (redo || col >= C) return;
    float2* __restrict__ o = partial_pm + ((int64_t)blockIdx.y * DC + d0) * cpad + col;
#pragma unroll
    for (int h = 0; h < DCT / 2; ++h) {
        o[(int64_t)(2 * h) * cpad] = make_float2(0.f, s2[h].x);
        o[(int64_t)(2 * h + 1) * cpad] = make_float2(0.f, s2[h].y);
    }
}

// Running-maximum sweep (lazy: the reference only moves when a term exceeds it by 2^DLSE_THR) for the
// waves flagged by pair_pm_kernel; 16 dimensions per workgroup (blockIdx.z), flags indexed by the
// 16-dimension group the fast kernel's launch at dimension offset `d0 = 16 * (blockIdx.z / zper * zper)`
// covered.
__global__ void __launch_bounds__(256)
pair_pm_online_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                      float2* __restrict__ partial_pm, const int* __restrict__ redo_flags, int64_t C,
                      int64_t cpad, int64_t R, int64_t rows_per_split, int D, int DC, int zper) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int fgroup = (int)(blockIdx.z / zper) * zper;   // first 16-dim group of the fast launch
    if (redo_flags[((int64_t)blockIdx.y * (DC / 16) + fgroup) * (cpad / 64) + col / 64] == 0) return;
    const int64_t i = col < C ? col : C - 1;
    const int g0 = 16 * (int)blockIdx.z;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > R) j1 = R;
    float zr[16], m[16], s[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        zr[e] = (g0 + e < D) ? z[i * D + g0 + e] : 0.f;
        m[e] = DLSE_MINIT;
        s[e] = 0.f;
    }
    for (int64_t j = j0; j < j1; ++j) {
        const float* __restrict__ pm = dparams + (j * 3 + 0) * DC + g0;
        const float* __restrict__ pw = dparams + (j * 3 + 1) * DC + g0;
        const float* __restrict__ pb = dparams + (j * 3 + 2) * DC + g0;
        float t16[16];
        float ex = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float y = (zr[e] - pm[e]) * pw[e];
            t16[e] = fmaf(-y, y, pb[e]);
            ex = fmaxf(ex, t16[e] - m[e]);
        }
        if (__any(ex > DLSE_THR)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float x = t16[e] - m[e];
                if (x > DLSE_THR) {
                    s[e] = s[e] * lvae_exp2(-x) + 1.0f;
                    m[e] = t16[e];
                } else {
                    s[e] += lvae_exp2(x);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] += lvae_exp2(t16[e] - m[e]);
        }
    }
    if (col < C) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            partial_pm[((int64_t)blockIdx.y * DC + g0 + e) * cpad + col] = make_float2(m[e], s[e]);
    }
}

// Latent rows too wide for registers (DC > 64; D = 768 is the reference's module default, modules/encoder.py:9): the
// sweep is blocked JB posteriors x 16 dimensions.  A lane keeps JB partial sums v_ij (one per posterior of the batch)
// and walks the latent row in chunks of sixteen dimensions: its own sixteen z values come from global memory once per
// (batch, chunk) -- 64 B contiguous per lane, re-used for all JB posteriors --, the posteriors' parameters through
// wave-uniform (scalar) loads, three vector instructions per (i, j, d).  The per-dimension biases do not depend on the
// sample: lane l sums them for posterior l of the batch and the sums are broadcast once per batch.  Exact differences
// like the other direct kernels.  (The one-row-at-a-time generic kernel below re-read z_id from global memory, one
// 4-byte element per lane and 64 cache lines per wave-load, for every (j, d): 377 ms at 16,384^2, D = 768.)
template <int JB>
__global__ void __launch_bounds__(256)
pair_direct_wide_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                        float2* __restrict__ partial, const PairHeader* __restrict__ hdr,
                        float cond_limit, int use_if_above, int64_t C, int64_t cpad, int64_t R,
                        int64_t rows_per_split, int D, int DC, int excl, int64_t excl_offset) {
    if (cond_limit >= 0.f) {
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    static_assert(JB <= 64, "one lane per posterior of a batch sums its biases");
    const int lane = threadIdx.x & 63;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = col < C ? col : C - 1;
    const float* __restrict__ zi = z + i * D;
    const bool vec4 = (D & 3) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0;   // 16-byte loads of 16-byte aligned rows
    float mv = DLSE_MINIT, sv = 0.f;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > R) j1 = R;
    const int64_t jskip = excl ? i + excl_offset : -1;
    for (int64_t jb = j0; jb < j1; jb += JB) {
        // bias sums of the batch: lane l owns posterior jb + l
        float bsum = 0.f;
        if (lane < JB && jb + lane < j1) {
            const float* __restrict__ pb = dparams + ((jb + lane) * 3 + 2) * DC;
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            for (int d = 0; d < DC; d += 4) {
                const float4 q = *reinterpret_cast<const float4*>(pb + d);
                b0 += q.x; b1 += q.y; b2 += q.z; b3 += q.w;
            }
            bsum = (b0 + b1) + (b2 + b3);
        }
        float v[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) v[jj] = 0.f;
        for (int d0 = 0; d0 < DC; d0 += 16) {
            float zr[16];
            if (vec4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int d = d0 + 4 * q;
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d < D) t = *reinterpret_cast<const float4*>(zi + d);
                    zr[4 * q] = t.x; zr[4 * q + 1] = t.y; zr[4 * q + 2] = t.z; zr[4 * q + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) zr[e] = d0 + e < D ? zi[d0 + e] : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                // (a batch past the end of the split repeats its last posterior: its sums are dropped below)
                const int64_t j = jb + jj < j1 ? jb + jj : j1 - 1;
                const float* __restrict__ pm = dparams + (j * 3 + 0) * DC + d0;
                const float* __restrict__ pw = dparams + (j * 3 + 1) * DC + d0;
                float a = v[jj];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float y = (zr[e] - pm[e]) * pw[e];
                    a = fmaf(-y, y, a);
                }
                v[jj] = a;
            }
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            if (jb + jj < j1) {                                  // (wave-uniform)
                float vv = v[jj] + __shfl(bsum, jj, 64);
                if (jb + jj == jskip) vv = -INFINITY;
                const float xv = vv - mv;
                if (xv > DLSE_THR) {
                    sv = sv * lvae_exp2(-xv) + 1.0f;
                    mv = vv;
                } else {
                    sv += lvae_exp2(xv);
                }
            }
        }
    }
    if (col < C) partial[(int64_t)blockIdx.y * cpad + col] = make_float2(mv, sv);
}

// Wide rows, FEW kept elements (training batches: C <= 512): a lane per sample leaves the chip empty -- 32 samples at
// D = 768 are one partly filled wave walking 32 x 768 terms, 0.39 ms.  Here a WAVE owns SPW samples and its lanes run
// along the latent row (dimension lane + 64 s in register s): the posterior's parameters are read coalesced, once per
// posterior for the wave's SPW samples, three vector instructions per 64 terms and a butterfly per (sample, posterior).
// C / SPW waves x the plan's posterior splits.
template <int SPW>
__global__ void __launch_bounds__(256)
pair_direct_rowwave_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                           float2* __restrict__ partial, const PairHeader* __restrict__ hdr,
                           float cond_limit, int use_if_above, int64_t C, int64_t cpad, int64_t R,
                           int64_t rows_per_split, int D, int DC, int excl, int64_t excl_offset) {
    if (cond_limit >= 0.f) {
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    constexpr int MAXS = 16;                      // DC <= 1024
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * SPW;
    if (i0 >= C) return;
    float zr[SPW][MAXS];
#pragma unroll
    for (int q = 0; q < SPW; ++q)
#pragma unroll
        for (int sx = 0; sx < MAXS; ++sx) {
            const int d = sx * 64 + lane;
            zr[q][sx] = (d < D && i0 + q < C) ? z[(i0 + q) * D + d] : 0.f;
        }
    float mv[SPW], sv[SPW];
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        mv[q] = DLSE_MINIT;
        sv[q] = 0.f;
    }
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > R) j1 = R;
    for (int64_t j = j0; j < j1; ++j) {
        const float* __restrict__ pm = dparams + (j * 3 + 0) * DC;
        const float* __restrict__ pw = dparams + (j * 3 + 1) * DC;
        const float* __restrict__ pb = dparams + (j * 3 + 2) * DC;
        float acc[SPW];
#pragma unroll
        for (int q = 0; q < SPW; ++q) acc[q] = 0.f;
#pragma unroll
        for (int sx = 0; sx < MAXS; ++sx) {
            const int d = sx * 64 + lane;
            if (sx * 64 < DC) {                   // (uniform)
                float m = 0.f, w = 0.f, bb = 0.f;
                if (d < DC) {
                    m = pm[d];
                    w = pw[d];
                    bb = pb[d];
                }
#pragma unroll
                for (int q = 0; q < SPW; ++q) {
                    const float y = (zr[q][sx] - m) * w;
                    acc[q] += fmaf(-y, y, bb);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            float v = wave_sum(acc[q]);
            if (excl && j == i0 + q + excl_offset) v = -INFINITY;
            const float xv = v - mv[q];
            if (xv > DLSE_THR) {                  // (uniform: every lane holds the same v)
                sv[q] = sv[q] * lvae_exp2(-xv) + 1.0f;
                mv[q] = v;
            } else {
                sv[q] += lvae_exp2(xv);
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < SPW; ++q)
            if (i0 + q < C) partial[(int64_t)blockIdx.y * cpad + i0 + q] = make_float2(mv[q], sv[q]);
    }
}

// Any latent size: the sample row is re-read from global memory (L1/L2 resident) for every posterior.
__global__ void __launch_bounds__(256)
pair_direct_generic_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                           float2* __restrict__ partial, const PairHeader* __restrict__ hdr,
                           float cond_limit, int use_if_above, int64_t C, int64_t cpad, int64_t R,
                           int64_t rows_per_split, int D, int DC, int excl, int64_t excl_offset) {
    if (cond_limit >= 0.f) {
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = col < C ? col : C - 1;
    const float* __restrict__ zi = z + i * D;
    float mv = DLSE_MINIT, sv = 0.f;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > R) j1 = R;
    const int64_t jskip = excl ? i + excl_offset : -1;
    for (int64_t j = j0; j < j1; ++j) {
        const float* __restrict__ pm = dparams + (j * 3 + 0) * DC;
        const float* __restrict__ pw = dparams + (j * 3 + 1) * DC;
        const float* __restrict__ pb = dparams + (j * 3 + 2) * DC;
        float v = 0.f;
        for (int d = 0; d < D; ++d) {
            const float y = (zi[d] - pm[d]) * pw[d];
            v += fmaf(-y, y, pb[d]);
        }
        if (j == jskip) v = -INFINITY;
        const float xv = v - mv;
        if (xv > DLSE_THR) {
            sv = sv * lvae_exp2(-xv) + 1.0f;
            mv = v;
        } else {
            sv += lvae_exp2(xv);
        }
    }
    if (col < C) partial[(int64_t)blockIdx.y * cpad + col] = make_float2(mv, sv);
}

template <int DCT>
static int launch_joint(const PairPlan& p, const float* z, const float* dparams, float2* partial,
                        const PairHeader* hdr, float cond_limit, int use_if_above, int excl,
                        int64_t excl_offset, hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / 256), (unsigned)p.dsplit);
    hipLaunchKernelGGL((pair_direct_kernel<DCT>), grid, dim3(256), 0, st, z, dparams, partial, hdr,
                       cond_limit, use_if_above, p.C, p.cpad, p.R, p.rows_per_dsplit, p.D, p.DC, excl,
                       excl_offset);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

template <int DCT>
static int launch_pm(const PairPlan& p, const float* z, const float* dparams, float2* partial_pm,
                     int* redo_flags, int d0, int chunks, const PairHeader* hdr, float pm_cond_limit, hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / 256), (unsigned)p.dsplit, (unsigned)chunks);
    hipLaunchKernelGGL((pair_pm_kernel<DCT>), grid, dim3(256), 0, st, z, dparams, partial_pm, redo_flags,
                       p.C, p.cpad, p.R, p.rows_per_dsplit, p.D, p.DC, d0, hdr, pm_cond_limit);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// joint: produce the log q(z) partials (gated on the conditioning estimate when cond_limit >= 0);
// pm: produce the product-of-marginals partials; pm_fast_done: the fast pass has been run by
// lvae_launch_pm_bx (same partial layout, same flags), only the redo of flagged waves is launched here;
// pm_cond_limit >= 0: lvae_launch_pm_bx ran gated on the conditioning estimate, and the fast pass here runs
// with the opposite gate (exactly one of the two writes the partials and the flags).
// The direct parameters are padded (w = b = 0, a zero term) to DC columns: 16/32/64 for D <= 64, where
// the kernels keep the whole latent row in registers, else a multiple of 16.
int lvae_launch_pair_direct(const PairPlan& p, bool joint, bool pm, const float* z,
                            const float* dparams, float2* partial, float2* partial_pm, int* redo_flags,
                            const PairHeader* hdr, float cond_limit, int use_if_above, int excl,
                            int64_t excl_offset, hipStream_t st, bool pm_fast_done, float pm_cond_limit) {
    if (p.cpad % 256 != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "pair plan: cpad %lld not a multiple of 256",
                              (long long)p.cpad);
    const int DC = p.DC;
    int rc = LVAE_OK;
    if (joint) {
        if (DC == 16) rc = launch_joint<16>(p, z, dparams, partial, hdr, cond_limit, use_if_above, excl, excl_offset, st);
        else if (DC == 32) rc = launch_joint<32>(p, z, dparams, partial, hdr, cond_limit, use_if_above, excl, excl_offset, st);
        else if (DC == 64) rc = launch_joint<64>(p, z, dparams, partial, hdr, cond_limit, use_if_above, excl, excl_offset, st);
        else {
            const dim3 grid((unsigned)(p.cpad / 256), (unsigned)p.dsplit);
            // (LVAE_DIRECT_GENERIC=1, read once per process: the one-row-at-a-time kernel, for comparisons)
            static const bool force_generic = getenv("LVAE_DIRECT_GENERIC") != nullptr;
            if (DC <= 1024 && p.C <= 512 && !force_generic) {
                // few kept elements: a wave per sample (one per wave up to 128 samples, four beyond)
                if (p.C <= 128)
                    hipLaunchKernelGGL((pair_direct_rowwave_kernel<1>), dim3((unsigned)lvae_ceil_div(p.C, 4), (unsigned)p.dsplit),
                                       dim3(256), 0, st, z, dparams, partial, hdr, cond_limit, use_if_above, p.C, p.cpad, p.R,
                                       p.rows_per_dsplit, p.D, DC, excl, excl_offset);
                else
                    hipLaunchKernelGGL((pair_direct_rowwave_kernel<4>), dim3((unsigned)lvae_ceil_div(p.C, 16), (unsigned)p.dsplit),
                                       dim3(256), 0, st, z, dparams, partial, hdr, cond_limit, use_if_above, p.C, p.cpad, p.R,
                                       p.rows_per_dsplit, p.D, DC, excl, excl_offset);
            } else if (DC % 16 == 0 && !force_generic)
                hipLaunchKernelGGL((pair_direct_wide_kernel<32>), grid, dim3(256), 0, st, z, dparams, partial,
                                   hdr, cond_limit, use_if_above, p.C, p.cpad, p.R, p.rows_per_dsplit, p.D,
                                   DC, excl, excl_offset);
            else
                hipLaunchKernelGGL(pair_direct_generic_kernel, grid, dim3(256), 0, st, z, dparams, partial,
                                   hdr, cond_limit, use_if_above, p.C, p.cpad, p.R, p.rows_per_dsplit, p.D,
                                   DC, excl, excl_offset);
            LVAE_LAUNCH_CHECK();
        }
        if (rc != LVAE_OK) return rc;
    }
    if (pm) {
        // per-dimension states are independent: latent rows wider than 32 are swept in 32-wide chunks
        int zper = DC >= 32 ? 2 : 1;
        if (!pm_fast_done) {                                      // (pm_fast_done: another kernel wrote partials + flags)
            const int full = DC / 32;
            if (full > 0) rc = launch_pm<32>(p, z, dparams, partial_pm, redo_flags, 0, full, hdr, pm_cond_limit, st);
            if (rc != LVAE_OK) return rc;
            if (DC % 32) rc = launch_pm<16>(p, z, dparams, partial_pm, redo_flags, full * 32, 1, hdr, pm_cond_limit, st);
            if (rc != LVAE_OK) return rc;
        }
        // DC % 32 == 16: the last fast launch covers one group, and its flag sits at that group's index,
        // which (blockIdx.z / 2) * 2 also yields because the group index is even
        const dim3 grid((unsigned)(p.cpad / 256), (unsigned)p.dsplit, (unsigned)(DC / 16));
        hipLaunchKernelGGL(pair_pm_online_kernel, grid, dim3(256), 0, st, z, dparams, partial_pm,
                           redo_flags, p.C, p.cpad, p.R, p.rows_per_dsplit, p.D, DC, zper);
        LVAE_LAUNCH_CHECK();
    }
    return LVAE_OK;
}
