// Launchers of the fused pair kernel (pair_lse_kernel.h) on the fp32 matrix cores: the default contraction.
#include "pair_lse_kernel.h"

template <int DP, int NCB, int WPS>
static int launch_t(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                    float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                    bool excl, int64_t excl_offset, int epi, const int* redo_flags, int sym, const float* colnorm,
                    hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32 * NCB)), (unsigned)p.nsplit);
    const PermArgs pa{};
#define LVAE_PAIR(X, E)                                                                              \
    hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, NCB, WPS, X, E>), grid, dim3(256), 0, st, rfrag, rbias, \
                       cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad,  \
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

// cond_limit < 0: run unconditionally.  Otherwise run iff (cond > cond_limit) == use_if_above.
// epi 0: logsumexp epilogue with a running maximum; 4: logsumexp against the fixed reference folded into
// the row biases; 1/2/3/5: kernel-matrix column sums (float64 partials; parameters in hdr->ksum: single
// bandwidth / polynomial of degree <= 16 / arbitrary bandwidths / polynomial with paired coefficients).
// redo_flags (may be NULL): one int per workgroup column group; workgroups whose flag is 0 exit.
// colnorm (epi 1/2/3 only, else NULL): |x_c|^2 per kept element, see the kernel.
// sym != 0 (epi 1/2/3 only; the kept elements are rows excl_offset.. of the reduced sample, excl_offset a
// multiple of 32): visit the upper
// triangle of tiles only; the TOTAL over columns is that of the full matrix, the per-column sums are not.
// D in (32, 64] with TWO column blocks per wave (220 registers, two waves per SIMD): the log q(z) plans of the fp32
// path (lvae_log_qz_plan) -- 88.8 -> 89.9 % of peak at 65,536^2, every LDS read feeds two MFMAs.  Logsumexp
// epilogues only; kernel sums, permutation sums and the split-bf16 contraction keep one column block at this width.
static int launch_64x2(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                       float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                       bool excl, int64_t excl_offset, int epi, const int* redo_flags, hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32 * 2)), (unsigned)p.nsplit);
    const PermArgs pa{};
#define LVAE_PAIR(X, E)                                                                              \
    hipLaunchKernelGGL((pair_lse_mfma_kernel<64, 2, 2, X, E>), grid, dim3(256), 0, st, rfrag, rbias, \
                       cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad,  \
                       excl_offset, pa, redo_flags, 0, nullptr)
    // (with the diagonal excluded the masked copies of the biases need 32 more registers and spill: those calls keep
    //  one column block, lvae_log_qz_plan)
    if (excl) return lvae_set_error(LVAE_ERR_INVALID, "two column blocks at D = 64: not with an excluded diagonal");
    if (epi == 0) LVAE_PAIR(false, 0);
    else if (epi == 4) LVAE_PAIR(false, 4);
    else return lvae_set_error(LVAE_ERR_INVALID, "two column blocks at D = 64 serve the logsumexp epilogues only (epi %d)", epi);
#undef LVAE_PAIR
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// D in (64, 128], logsumexp epilogues: two workgroups per CU (202-228 registers, no spills; the kernel-sum epilogues
// need up to 268 and stay at one).  65,536^2 at D = 128: 90.9 -> 91.6 % of peak; the plans of log q(z) count 512 slots.
static int launch_128w2(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                        float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                        bool excl, int64_t excl_offset, int epi, const int* redo_flags, hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32)), (unsigned)p.nsplit);
    const PermArgs pa{};
#define LVAE_PAIR(X, E)                                                                               \
    hipLaunchKernelGGL((pair_lse_mfma_kernel<128, 1, 2, X, E>), grid, dim3(256), 0, st, rfrag, rbias, \
                       cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad,   \
                       excl_offset, pa, redo_flags, 0, nullptr)
    if (epi == 0) { if (excl) LVAE_PAIR(true, 0); else LVAE_PAIR(false, 0); }
    else { if (excl) LVAE_PAIR(true, 4); else LVAE_PAIR(false, 4); }
#undef LVAE_PAIR
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_pair_mfma(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                          float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                          bool excl, int64_t excl_offset, int epi, const int* redo_flags, int sym, const float* colnorm,
                    hipStream_t st) {
    if (p.cpad % (4 * 32 * p.NCB) != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "pair plan: cpad %lld not a multiple of the tile",
                              (long long)p.cpad);
    if (((epi >= 1 && epi <= 3) || epi == 5) && !colnorm)
        return lvae_set_error(LVAE_ERR_INVALID, "kernel-sum epilogues need the per-column norms");
    switch (p.DP) {
        case 16: return launch_t<16, 2, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, sym, colnorm, st);
        case 32: return launch_t<32, 2, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, sym, colnorm, st);
        case 64:
            if (p.NCB == 2) return launch_64x2(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, st);
            return launch_t<64, 1, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, sym, colnorm, st);
        case 128:
            if ((epi == 0 || epi == 4) && p.NCB == 1) return launch_128w2(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, st);
            return launch_t<128, 1, 1>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, epi, redo_flags, sym, colnorm, st);
        default:
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "MFMA path needs D <= 128 (got D=%d)", p.D);
    }
}

int lvae_launch_pair_lse_mfma(const PairPlan& p, const float* rfrag, const float* rbias,
                              const float* cop, float2* partial, const PairHeader* hdr,
                              float cond_limit, int use_if_above, bool excl, int64_t excl_offset,
                              hipStream_t st) {
    return lvae_launch_pair_mfma(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl,
                                 excl_offset, 0, nullptr, 0, nullptr, st);
}

// Kernel-matrix column sums plus label-permutation sums (see PERM above); diagonal pairs (reduced row ==
// kept column + excl_offset) are always excluded.  The plan must have NCB == 1; epi 1, 2 or 5.
template <int DP, int WPS>
static int launch_perm_t(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                         float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                         int64_t excl_offset, int epi, const PermArgs& pa, int sym, const float* colnorm,
                         hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32)), (unsigned)p.nsplit);
    if (epi == 1)
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, 1, WPS, true, 1, true>), grid, dim3(256), 0, st, rfrag, rbias,
                           cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad, excl_offset, pa, nullptr, sym, colnorm);
    else if (epi == 2)
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, 1, WPS, true, 2, true>), grid, dim3(256), 0, st, rfrag, rbias,
                           cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad, excl_offset, pa, nullptr, sym, colnorm);
    else
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, 1, WPS, true, 5, true>), grid, dim3(256), 0, st, rfrag, rbias,
                           cop, partial, hdr, cond_limit, use_if_above, p.nblk, p.blk_per_split, p.cpad, excl_offset, pa, nullptr, sym, colnorm);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_pair_mfma_perm(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
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
        case 16: return launch_perm_t<16, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl_offset, epi, pa, sym, colnorm, st);
        case 32: return launch_perm_t<32, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl_offset, epi, pa, sym, colnorm, st);
        case 64: return launch_perm_t<64, 1>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl_offset, epi, pa, sym, colnorm, st);
        case 128: return launch_perm_t<128, 1>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl_offset, epi, pa, sym, colnorm, st);
        default:
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "MFMA path needs D <= 128 (got D=%d)", p.D);
    }
}

// Three kernel-sum jobs in one launch (pair_lse_multi_kernel).  Served: D <= 128 (plan dimension 16 / 32 / 64) and the
// single-exponential epilogues (1, 2, 5); anything else returns LVAE_ERR_UNSUPPORTED and the caller runs the jobs
// one by one.
template <int DP, int NCB, int WPS>
static int launch_multi_t(const PairJobs& jobs, const PairHeader* hdr, dim3 grid, bool excl, int epi, hipStream_t st) {
#define LVAE_MULTI(X, E) hipLaunchKernelGGL((pair_lse_multi_kernel<DP, NCB, WPS, X, E>), grid, dim3(256), 0, st, jobs, hdr)
    switch (epi * 2 + (excl ? 1 : 0)) {
        case 2: LVAE_MULTI(false, 1); break;
        case 3: LVAE_MULTI(true, 1); break;
        case 4: LVAE_MULTI(false, 2); break;
        case 5: LVAE_MULTI(true, 2); break;
        case 10: LVAE_MULTI(false, 5); break;
        case 11: LVAE_MULTI(true, 5); break;
        default: return lvae_set_error(LVAE_ERR_UNSUPPORTED, "multi-job launch: epilogue %d not instantiated", epi);
    }
#undef LVAE_MULTI
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

bool lvae_pair_multi_supported(int DP, int epi) { return (DP == 16 || DP == 32 || DP == 64) && (epi == 1 || epi == 2 || epi == 5); }

int lvae_launch_pair_multi(int DP, const PairJobs& jobs, int njobs, const PairHeader* hdr, bool excl, int epi,
                           hipStream_t st) {
    int gx = 0, gy = 0;
    for (int k = 0; k < njobs; ++k) {
        const PairJob& J = jobs.j[k];
        const int ncb = DP <= 32 ? 2 : 1;
        if (J.cpad % (128 * ncb) != 0 || J.gx != (int)(J.cpad / (128 * ncb)) || J.gy < 1 || !J.colnorm)
            return lvae_set_error(LVAE_ERR_INVALID, "multi-job launch: job %d does not match its plan", k);
        if (J.gx > gx) gx = J.gx;
        if (J.gy > gy) gy = J.gy;
    }
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)njobs);
    switch (DP) {
        case 16: return launch_multi_t<16, 2, 2>(jobs, hdr, grid, excl, epi, st);
        case 32: return launch_multi_t<32, 2, 2>(jobs, hdr, grid, excl, epi, st);
        case 64: return launch_multi_t<64, 1, 2>(jobs, hdr, grid, excl, epi, st);
        default: return lvae_set_error(LVAE_ERR_UNSUPPORTED, "multi-job launch: plan dimension %d not instantiated", DP);
    }
}
