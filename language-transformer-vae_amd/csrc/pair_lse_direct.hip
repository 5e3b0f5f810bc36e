// Direct-form pair log-density + streaming logsumexp on the vector ALU (gfx950).
//
// One lane per kept element (sample z_i, its latent row in VGPRs); the reduced axis (posterior j) is
// wave-uniform, so its parameters (mu_j, w_j = sqrt(0.5*log2e/var_j), b_j = -0.5*log2e*(log var_j +
// log 2pi)) are read through uniform loads shared by the whole wave.  Per (i, j, d):
//     y = (z_id - mu_jd) * w_jd ;  t = b_jd - y*y            (base-2 log-density of one dimension)
// JOINT : v_ij = sum_d t, folded into the lane's running (max, sum)        -> log q(z_i)
// PM    : every (i, d) keeps its own running (max, sum) over j             -> prod-of-marginals term
//         (loss_and_optimisation.py:195: logsumexp of the [M,N,D] tensor over the posterior axis).
// This form has no cancellation (it is the reference's own formula), so it also serves as the
// fall-back of the MFMA path for badly conditioned inputs (peaked posteriors far from the origin) and
// for latent sizes the MFMA path does not cover.  Bound: VALU / transcendental rate (one v_exp_f32
// per (i,j,d) term with PM), not HBM.
#include "pair_plan.h"

#define DLSE_THR 8.0f
#define DLSE_MINIT (-1.0e30f)

template <int DCT, bool JOINT, bool PM>
__global__ void __launch_bounds__(256)
pair_direct_kernel(const float* __restrict__ z, const float* __restrict__ dparams,
                   float2* __restrict__ partial, float2* __restrict__ partial_pm,
                   const PairHeader* __restrict__ hdr, float cond_limit, int use_if_above,
                   int64_t C, int64_t cpad, int64_t R, int64_t rows_per_split, int D, int DC, int d0,
                   int excl, int64_t excl_offset) {
    if (cond_limit >= 0.f) {
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = col < C ? col : C - 1;

    float zr[DCT];
#pragma unroll
    for (int d = 0; d < DCT; ++d) zr[d] = (d0 + d < D) ? z[i * D + d0 + d] : 0.f;

    float mv = DLSE_MINIT, sv = 0.f;
    float m[PM ? DCT : 1], s[PM ? DCT : 1];
    if (PM) {
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            m[d] = DLSE_MINIT;
            s[d] = 0.f;
        }
    }

    const int64_t j0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t j1 = j0 + rows_per_split;
    if (j1 > R) j1 = R;
    const int64_t jskip = excl ? i + excl_offset : -1;

    for (int64_t j = j0; j < j1; ++j) {
        const float* __restrict__ pm = dparams + (j * 3 + 0) * DC + d0;
        const float* __restrict__ pw = dparams + (j * 3 + 1) * DC + d0;
        const float* __restrict__ pb = dparams + (j * 3 + 2) * DC + d0;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < DCT; g += 16) {
            float t16[16];
            float ex = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int d = g + e;
                const float y = (zr[d] - pm[d]) * pw[d];
                const float t = fmaf(-y, y, pb[d]);
                if (JOINT) v += t;
                if (PM) {
                    t16[e] = t;
                    ex = fmaxf(ex, t - m[d]);
                }
            }
            if (PM) {
                if (__any(ex > DLSE_THR)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int d = g + e;
                        const float x = t16[e] - m[d];
                        if (x > DLSE_THR) {
                            s[d] = s[d] * lvae_exp2(-x) + 1.0f;
                            m[d] = t16[e];
                        } else {
                            s[d] += lvae_exp2(x);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[g + e] += lvae_exp2(t16[e] - m[g + e]);
                }
            }
        }
        if (JOINT) {
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
    }

    if (col < C) {
        if (JOINT) partial[(int64_t)blockIdx.y * cpad + col] = make_float2(mv, sv);
        if (PM) {
#pragma unroll
            for (int d = 0; d < DCT; ++d)
                if (d0 + d < DC)
                    partial_pm[((int64_t)blockIdx.y * DC + d0 + d) * cpad + col] = make_float2(m[d], s[d]);
        }
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
static int launch_chunk(const PairPlan& p, bool joint, bool pm, const float* z, const float* dparams,
                        float2* partial, float2* partial_pm, const PairHeader* hdr, float cond_limit,
                        int use_if_above, int d0, int excl, int64_t excl_offset, hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / 256), (unsigned)p.dsplit);
#define LVAE_DIRECT(J, P)                                                                         \
    hipLaunchKernelGGL((pair_direct_kernel<DCT, J, P>), grid, dim3(256), 0, st, z, dparams, partial, \
                       partial_pm, hdr, cond_limit, use_if_above, p.C, p.cpad, p.R,                \
                       p.rows_per_dsplit, p.D, p.DC, d0, excl, excl_offset)
    if (joint && pm) LVAE_DIRECT(true, true);
    else if (joint) LVAE_DIRECT(true, false);
    else LVAE_DIRECT(false, true);
#undef LVAE_DIRECT
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// joint: produce the log q(z) partials; pm: produce the product-of-marginals partials.
// The direct parameters are padded (w = b = 0, a zero term) to DC columns: 16/32/64 for D <= 64, where
// one fused kernel keeps the whole latent row in registers, else a multiple of 16.
int lvae_launch_pair_direct(const PairPlan& p, bool joint, bool pm, const float* z,
                            const float* dparams, float2* partial, float2* partial_pm,
                            const PairHeader* hdr, float cond_limit, int use_if_above, int excl,
                            int64_t excl_offset, hipStream_t st) {
    if (p.cpad % 256 != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "pair plan: cpad %lld not a multiple of 256",
                              (long long)p.cpad);
    const int DC = p.DC;
    if (DC == 16 || DC == 32 || DC == 64) {
        // whole latent row in registers: joint and pm fused in one sweep
        if (DC == 16) return launch_chunk<16>(p, joint, pm, z, dparams, partial, partial_pm, hdr, cond_limit, use_if_above, 0, excl, excl_offset, st);
        if (DC == 32) return launch_chunk<32>(p, joint, pm, z, dparams, partial, partial_pm, hdr, cond_limit, use_if_above, 0, excl, excl_offset, st);
        return launch_chunk<64>(p, joint, pm, z, dparams, partial, partial_pm, hdr, cond_limit, use_if_above, 0, excl, excl_offset, st);
    }
    if (joint) {
        const dim3 grid((unsigned)(p.cpad / 256), (unsigned)p.dsplit);
        hipLaunchKernelGGL(pair_direct_generic_kernel, grid, dim3(256), 0, st, z, dparams, partial,
                           hdr, cond_limit, use_if_above, p.C, p.cpad, p.R, p.rows_per_dsplit, p.D,
                           DC, excl, excl_offset);
        LVAE_LAUNCH_CHECK();
    }
    if (pm) {
        // per-dimension states are independent: sweep the latent row in 16-wide chunks (DC % 16 == 0)
        for (int d0 = 0; d0 < DC; d0 += 16) {
            int rc = launch_chunk<16>(p, false, true, z, dparams, nullptr, partial_pm, hdr, -1.f, 0,
                                      d0, 0, 0, st);
            if (rc != LVAE_OK) return rc;
        }
    }
    return LVAE_OK;
}
