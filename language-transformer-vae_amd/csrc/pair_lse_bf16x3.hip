// EXPERIMENTAL, opt-in: operand preparation and launchers of the split-bf16 form of the pair kernel
// (pair_lse_kernel.h, template parameter BX; LVAE_PATH_MFMA_BF16X3 / lvae_set_pair_math).  Not what BASELINE's
// north star asks for (an fp32 MFMA contraction -- pair_lse_mfma.hip, the default and the path bench.py
// measures); kept next to it because the fp32 kernel sits at the ceiling of its structure (fp32 MFMA shares
// the vector lanes with the epilogue's exponentials, tools/ubench/bf16x3_tile.cpp: 2348 cycles per 32x32 tile
// at D = 32, measured in the kernel: 2349) and the same tile costs 976 cycles in this form.
#include "pair_lse_kernel.h"

// ---- operand preparation ------------------------------------------------------------------------------
// One thread per (row, step of 8 dimensions, k-half): computes the eight fp32 operand values of that half
// (SRC 0, GAUSS rows: a_d or b_d; SRC 1, SAMPLE rows: z_d^2 or z_d; SRC 2, POINT rows: 1 or -2 y_d -- the
// same values, after the same centering shift, as prep_operand_kernel of the fp32 path; SRC 3 / 4, the
// linear-only operands of the kernel sums: -2 y_d / x_d with sixteen dimensions per step, DP = padded D / 2)
// and writes their three limbs as 16-byte vectors.
//   FRAG (reduced axis): out[blk][step][limb][lane = row%32 + 32*half][8]           (16 B per lane)
//   else (kept axis)   : out[row][half][step][limb][8]
// The row bias (GAUSS) and the header statistics are produced by the fp32 prep kernel, which runs anyway
// (conditioning estimate, normaliser maximum); this kernel only adds the limb operands.
template <int SRC, bool FRAG>
__global__ void __launch_bounds__(256)
prep_bf16x3_kernel(const float* __restrict__ x, const float* __restrict__ spread, int spread_kind, int64_t n,
                   int64_t npad, int D, int DP, const PairHeader* __restrict__ hdr, double inv_ngauss,
                   uint4* __restrict__ out) {
    constexpr bool GAUSS = SRC == 0;
    constexpr bool POINT = SRC == 2;
    constexpr bool LIN = SRC == 3 || SRC == 4;           // linear-only operands of the kernel sums: 16 dims per step
    const int NS = DP / 8;                               // MFMA steps
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = npad * NS * 2;
    if (idx >= total) return;
    int64_t row;
    int step, half;
    if (FRAG) {                                          // row fastest: a wave writes whole 1 KiB pieces
        const int64_t blk = idx / (64 * NS);
        const int rem = (int)(idx - blk * 64 * NS);
        step = rem / 64;
        const int lane = rem % 64;
        half = lane >> 5;
        row = blk * 32 + (lane & 31);
    } else {
        row = idx / (2 * NS);
        const int rem = (int)(idx - row * 2 * NS);
        half = rem / NS;
        step = rem % NS;
    }
    unsigned hw[4] = {0, 0, 0, 0}, mw[4] = {0, 0, 0, 0}, lw[4] = {0, 0, 0, 0};
    if (row < n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = LIN ? 16 * step + 8 * half + e : 8 * step + e;
            float v = 0.f;
            if (d < D) {
                const float xs = x[row * D + d] - (LIN ? 0.f : (float)(hdr->colsum[d] * inv_ngauss));
                if (LIN) {      // (the polynomial epilogues' scale is folded into the POINT rows: prep.hip, prep_rows)
                    v = SRC == 3 ? (-2.0f * (hdr->ksum.poly ? -hdr->ksum.nbase : 1.f)) * xs : xs;
                } else if (GAUSS) {
                    const float sp = spread[row * D + d];
                    const float p = spread_kind == LVAE_SPREAD_VAR ? 1.0f / sp : expf(-sp);
                    v = half == 0 ? -0.5f * LVAE_LOG2E * p : LVAE_LOG2E * xs * p;
                } else if (POINT) {
                    v = half == 0 ? 1.0f : -2.0f * xs;
                } else {
                    v = half == 0 ? xs * xs : xs;
                }
            }
            const __bf16 hb = (__bf16)v;
            const float r1 = v - (float)hb;
            const __bf16 mb = (__bf16)r1;
            const __bf16 lb = (__bf16)(r1 - (float)mb);
            const unsigned sh = 16 * (e & 1);
            hw[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, hb) << sh;
            mw[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, mb) << sh;
            lw[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, lb) << sh;
        }
    }
    if (FRAG) {
        const int64_t blk = row / 32;
        const int lane = (int)(row % 32) + 32 * half;
        uint4* o = out + (blk * NS + step) * 3 * 64 + lane;
        o[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        o[64] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
        o[128] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    } else {
        uint4* o = out + ((row * 2 + half) * NS + step) * 3;
        o[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        o[1] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
        o[2] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

// ---- host launchers ----------------------------------------------------------------------------------
bool lvae_bf16x3_supported(int DP) { return DP == 16 || DP == 32 || DP == 64; }

// src: 0 GAUSS, 1 SAMPLE, 2 POINT, 3 / 4 linear-only POINT / SAMPLE (as lvae_launch_prep); frag: reduced-axis
// (fragment-major) layout.
int lvae_launch_prep_bf16x3(int src, bool frag, const float* x, const float* spread, int spread_kind,
                            int64_t n, int64_t npad, int D, int DP, const PairHeader* hdr, int64_t ngauss,
                            void* out, hipStream_t st) {
    const int64_t total = npad * (DP / 8) * 2;
    const dim3 grid((unsigned)lvae_ceil_div(total, 256));
    const double inv = 1.0 / (double)ngauss;
#define LVAE_PBX(S, F)                                                                                       \
    hipLaunchKernelGGL((prep_bf16x3_kernel<S, F>), grid, dim3(256), 0, st, x, spread, spread_kind, n, npad, D, \
                       DP, hdr, inv, (uint4*)out)
    if (src == 0 && frag) LVAE_PBX(0, true);
    else if (src == 2 && frag) LVAE_PBX(2, true);
    else if (src == 1 && !frag) LVAE_PBX(1, false);
    else if (src == 3 && frag) LVAE_PBX(3, true);
    else if (src == 4 && !frag) LVAE_PBX(4, false);
    else return lvae_set_error(LVAE_ERR_INVALID, "bf16x3 prep: unsupported operand role");
#undef LVAE_PBX
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

template <int DP, int NCB, int WPS>
static int launch_bx(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                     float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                     bool excl, int64_t excl_offset, int epi, const int* redo_flags, int sym, const float* colnorm,
                    hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32 * NCB)), (unsigned)p.nsplit);
    const PermArgs pa{};
#define LVAE_PAIR(X, E)                                                                                    \
    hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, NCB, WPS, X, E, false, true>), grid, dim3(256), 0, st, rfrag, \
                       rbias, cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad,  \
                       excl_offset, pa, redo_flags, sym, colnorm)
    switch (epi * 2 + (excl ? 1 : 0)) {
        case 0: LVAE_PAIR(false, 0); break;
        case 1: LVAE_PAIR(true, 0); break;
        case 2: LVAE_PAIR(false, 1); break;
        case 3: LVAE_PAIR(true, 1); break;
        case 4: LVAE_PAIR(false, 2); break;
        case 5: LVAE_PAIR(true, 2); break;
        case 6: LVAE_PAIR(false, 3); break;
        case 7: LVAE_PAIR(true, 3); break;
        case 8: LVAE_PAIR(false, 4); break;
        case 9: LVAE_PAIR(true, 4); break;
        case 10: LVAE_PAIR(false, 5); break;
        case 11: LVAE_PAIR(true, 5); break;
        default: return lvae_set_error(LVAE_ERR_INVALID, "bad epilogue id %d", epi);
    }
#undef LVAE_PAIR
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// Same contract as lvae_launch_pair_mfma; rfrag / cop hold the limb operands written by
// lvae_launch_prep_bf16x3 (the row biases and the header are those of the fp32 prep).
int lvae_launch_pair_bf16x3(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                            float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                            bool excl, int64_t excl_offset, int epi, const int* redo_flags, int sym,
                            const float* colnorm, hipStream_t st) {
    if (((epi >= 1 && epi <= 3) || epi == 5) && !colnorm)
        return lvae_set_error(LVAE_ERR_INVALID, "kernel-sum epilogues need the per-column norms");
    if (p.cpad % (4 * 32 * p.NCB) != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "pair plan: cpad %lld not a multiple of the tile", (long long)p.cpad);
    if (p.NCB != (p.DP <= 32 ? 2 : 1))
        return lvae_set_error(LVAE_ERR_INVALID, "pair plan: the bf16x3 kernel needs NCB 2 at D <= 32, 1 at D = 64");
    switch (p.DP) {
        case 16: return launch_bx<16, 2, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, sym, colnorm, st);
        case 32: return launch_bx<32, 2, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, sym, colnorm, st);
        case 64: return launch_bx<64, 1, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, sym, colnorm, st);
        default:
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the bf16x3 path needs D <= 64 (got D=%d)", p.D);
    }
}

// Kernel-matrix column sums plus label-permutation sums: same contract as lvae_launch_pair_mfma_perm.
template <int DP, int WPS>
static int launch_bx_perm(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                          float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                          int64_t excl_offset, int epi, const PermArgs& pa, int sym, const float* colnorm,
                          hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32)), (unsigned)p.nsplit);
    if (epi == 1)
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, 1, WPS, true, 1, true, true>), grid, dim3(256), 0, st, rfrag,
                           rbias, cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad, excl_offset, pa, nullptr,
                           sym, colnorm);
    else if (epi == 2)
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, 1, WPS, true, 2, true, true>), grid, dim3(256), 0, st, rfrag,
                           rbias, cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad, excl_offset, pa, nullptr,
                           sym, colnorm);
    else
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, 1, WPS, true, 5, true, true>), grid, dim3(256), 0, st, rfrag,
                           rbias, cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad, excl_offset, pa, nullptr,
                           sym, colnorm);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_pair_bf16x3_perm(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                                 float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                                 int64_t excl_offset, int epi, const PermArgs& pa, int sym, const float* colnorm,
                                 hipStream_t st) {
    if (sym && pa.col_offset % 32 != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "symmetric permutation sums need col_offset %% 32 == 0");
    if (p.NCB != 1 || p.cpad % 128 != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "permutation plan needs NCB == 1 and cpad %% 128 == 0");
    if (epi != 1 && epi != 2 && epi != 5)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED,
                              "the permutation test needs bandwidths that are integer multiples (<= 16) of the "
                              "smallest one");
    if (pa.P < 1 || pa.P > 128) return lvae_set_error(LVAE_ERR_INVALID, "need 1 <= labelings <= 128 per pass");
    if (!colnorm) return lvae_set_error(LVAE_ERR_INVALID, "kernel-sum epilogues need the per-column norms");
    switch (p.DP) {
        case 16: return launch_bx_perm<16, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl_offset, epi, pa, sym, colnorm, st);
        case 32: return launch_bx_perm<32, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl_offset, epi, pa, sym, colnorm, st);
        case 64: return launch_bx_perm<64, 1>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl_offset, epi, pa, sym, colnorm, st);
        default:
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the bf16x3 path needs D <= 64 (got D=%d)", p.D);
    }
}
