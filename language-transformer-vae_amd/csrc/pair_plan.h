// Host-side plan for one "pair log-density + logsumexp" job: tile shapes, grid split and the
// layout of the caller-provided workspace.  Shared by the MFMA path, the direct path and the API.
#pragma once
#include <stdlib.h>
#include "lvae_common.h"

// Parameters of the kernel-matrix epilogue: f(v) = sum_k exp(-alpha_k v), v = eps + |d2|.
// poly: c[0..n] are the coefficients of a polynomial in t = exp2(v * nbase); else c[k] = -alpha_k*log2e.
#define LVAE_KSUM_MAX 32
// Fixed-reference logsumexp: row biases are shifted so that every term is <= 2^LSE_HEADROOM; a column
// whose total stays below 2^LSE_FLOOR_LOG2 is redone with the running-maximum kernel.
#define LSE_HEADROOM 100.0f
#define LSE_FLOOR 8.2718061e-25f     // 2^-80
// If the typical normaliser sits more than this many bits below the largest one (a few extremely peaked
// posteriors among broad ones), nearly every column would underflow the fixed reference: skip that pass.
#define LSE_SKIP_SPREAD 130.0f
struct KsumParams {
    int poly;
    int n;
    float eps;
    float nbase;
    float c[LVAE_KSUM_MAX + 1];
};

// Normaliser statistics of the posteriors, spread over LVAE_NORM_SLOTS cache lines: every workgroup of the operand
// preparation adds its share to slot (workgroup index mod slots), lse_shift_bias_kernel folds the slots.  (All
// workgroups on ONE line cost ~35 ns each, device-scope atomics on a single address serialise at the memory side:
// 77 us for the 2,048 workgroups of C2 at one 32-row block each.)
#define LVAE_NORM_SLOTS 64
struct NormSlot {
    unsigned int cmax_key;   // order-preserving key of the largest density normaliser max_j -0.5 log2 prod_d(2 pi var_jd)
    unsigned int pad_;
    double norm_sum;         // sum_j of the density normalisers (their mean locates the typical posterior)
    unsigned int fill_[28];  // one slot = 128 bytes
};

// Lives at the start of the workspace; zeroed (hipMemsetAsync) at the start of every call.
struct PairHeader {
    double colsum[128];      // per-dimension sum of the posterior means (centering shift * R_gauss)
    unsigned int cond_bits;  // float bits of max_j conditioning estimate (non-negative -> uint order)
    unsigned int unused0_;
    float lse_ref;           // reference the row biases were shifted by: every term <= 2^LSE_HEADROOM
    unsigned int lse_skip;   // 1: the fixed-reference pass is pointless (see lse_shift_bias_kernel), go straight to the redo
    double unused1_;
    unsigned int pad_[58];
    KsumParams ksum;         // kernel-matrix epilogue parameters (copied in by the MMD entry point)
    unsigned int pad2_[27];
    NormSlot slot[LVAE_NORM_SLOTS];
};

// Late merge folded into latent_sums_kernel (the statistics call): rows whose fixed-reference total underflowed take
// the redo pass's partials (source a: running-maximum pairs, + the reference the biases were shifted by), and when
// the conditioning gate chose the direct form every row takes the direct kernel's partials (source b) -- the
// arithmetic of finalize_lse_kernel, in the launch that reads log q(z) anyway instead of two launches of its own.
struct LateMerge {
    const float2* partial_a;   // [nsplit_a][cpad]
    const float2* partial_b;   // [nsplit_b][cpad] or NULL
    const PairHeader* hdr;
    const int* flags;          // redo flag per group_cols kept elements
    float* out;                // log q(z) [M]: rewritten for merged rows
    int64_t cpad;
    int nsplit_a, nsplit_b, group_cols, use_ref_a, enabled;
    float cond_limit;          // < 0: no gate (source a only)
    float log_norm;
};

// Arguments of the label-permutation mode of the MFMA pair kernel (MMD permutation test).
struct PermArgs {
    const void* labfrag;             // [nblk][4 perm tiles][2 steps][64 lanes][8 bf16] (prep_labels_kernel)
    const unsigned char* labels;     // [P][n_lab] 0/1, for the kept-axis side
    double* qperm;                   // [256]: q_p partial sums, then (sym) the unweighted sums of Y
    double* diag_partial;            // [nsplit][cpad]: column sums of the diagonal tiles (sym)
    int64_t n_lab;                   // pooled sample size (row stride of labels)
    int64_t col_offset;              // pooled index of kept element 0
    int64_t C;                       // kept elements
    int P;
};

// One kernel-sum job of a multi-job launch (pair_lse_multi_kernel): operands, partials and grid of one kernel matrix.
struct PairJob {
    const float* rfrag;
    const float* rbias;
    const float* cop;
    const float* colnorm;
    float2* partial;
    int64_t nblk, blk_per_split, cpad, excl_offset;
    int sym, gx, gy;
};
struct PairJobs {
    PairJob j[3];
};

// Workspace region and grid of the MFMA backward of log q(z) (pair_grad_mfma.hip).
struct BwdMfmaLayout {
    int DP;                        // 0: unsupported (D > 128)
    int64_t cpad_i, cpad_j;        // samples / posteriors padded to the workgroup tile (128)
    int ns_a, ns_b;                // splits of the streamed axis: posteriors (sweep A), samples (sweep B)
    int64_t bps_a, bps_b;          // 32-row blocks per split
    size_t off_recj, off_reci, off_colj, off_coli, off_cbj, off_cbi, off_pa, off_pb, off_w0, off_w0a, off_gs, off_cgj, total;
};

struct PairPlan {
    int64_t R;       // reduced axis length (posteriors for log q(z); samples for q_z_estimate)
    int64_t C;       // kept axis length (one output per element)
    int D, DP, NCB;  // latent dim, padded dim (0: MFMA path unsupported), column blocks per wave
    int64_t nblk;    // 32-row blocks on the reduced axis
    int64_t cpad;    // kept axis padded to the MFMA workgroup tile (multiple of 256 when DP == 0)
    int nsplit;      // reduced-axis splits of the MFMA grid
    int64_t blk_per_split;
    int dsplit;      // reduced-axis splits of the direct grid
    int64_t rows_per_dsplit;
    bool with_pm;
    int DC;          // padded dim of the direct-path parameters: 16/32/64, or a multiple of 16 above
    // byte offsets into the workspace
    size_t off_hdr, off_rfrag, off_rbias, off_cop, off_colbias, off_partial;
    size_t off_dparams, off_dpartial, off_dpartial_pm, off_dflags, off_pmtmp, off_redo, off_pmbx_a, off_pmbx_b, total;
};

static inline size_t lvae_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// sym: the MFMA grid only does work on the upper triangle of tiles, uneven per workgroup -> cut the reduced
// axis finer so that the hardware dispatcher can balance it.
static inline PairPlan lvae_make_pair_plan(int64_t R, int64_t C, int D, bool with_pm, int force_ncb = 0,
                                           bool sym = false, bool lse = false);

// The plan of log q(z) (samples kept, posteriors reduced): at D in (32, 64] the fp32 contraction runs two column
// blocks per wave (pair_lse_mfma.hip: launch_64x2), the split-bf16 contraction (bx) one.
static inline PairPlan lvae_log_qz_plan(int64_t N, int64_t M, int D, bool with_pm, bool bx, bool excl = false) {
    return lvae_make_pair_plan(N, M, D, with_pm, (!bx && !excl && lvae_pad_dim(D) == 64) ? 2 : 0, false, true);
}

// lse: the plan is for a logsumexp epilogue (two workgroups per CU at every width; kernel sums at D > 128: one)
static inline PairPlan lvae_make_pair_plan(int64_t R, int64_t C, int D, bool with_pm, int force_ncb, bool sym, bool lse) {
    PairPlan p{};
    p.R = R; p.C = C; p.D = D; p.with_pm = with_pm;
    p.DP = lvae_pad_dim(D);
    p.NCB = (p.DP == 16 || p.DP == 32) ? 2 : 1;
    if (force_ncb) p.NCB = force_ncb;
    p.nblk = lvae_ceil_div(R, 32);
    const int64_t wg_cols = p.DP ? 4 * 32 * p.NCB : 256;
    p.cpad = lvae_round_up(C, wg_cols < 256 ? 256 : wg_cols);
    // MFMA grid: (column groups) x (reduced-axis splits).  All workgroups of a launch do the same work, S of
    // them are resident at a time (two per CU; one for the kernel-sum epilogues at D > 128), so the time is
    // rounds * (blocks per split + per-workgroup overhead) with rounds = ceil(groups * splits / S): pick the
    // split count that minimises it (a grid of 543 equal workgroups on 512 slots would take two full rounds).
    // At least five rounds' worth of workgroups (sym: uneven work per workgroup -> eight) when the axis allows.
    const int64_t groups = p.DP ? p.cpad / (4 * 32 * p.NCB) : 1;
    const int64_t ns_forced = lvae_switches().nsplit;
    const int64_t slots = 256 * ((p.DP == 128 && !lse) ? 1 : 2);
    const int64_t max_ns = p.nblk < 64 ? (p.nblk < 1 ? 1 : p.nblk) : 64;
    int64_t ns = 1;
    if (ns_forced) {
        ns = ns_forced;
    } else {
        // (1) The many-rounds rule, measured at C2 (N = M = 65,536): five rounds of smaller workgroups beat one round
        //     of large ones by 2.4 % (sym: eight, the work per workgroup is uneven) -- as long as a split keeps
        //     enough blocks to amortise the per-workgroup prologue (>= 128).
        double best = 1e300;
        int64_t ns_rounds = 1;
        for (int64_t c = 1; c <= max_ns; ++c) {
            if (groups * c < (sym ? 8 : 5) * slots && c < max_ns) continue;
            const int64_t rounds = lvae_ceil_div(groups * c, slots);
            const double cost = (double)rounds * ((double)lvae_ceil_div(p.nblk, c) + 4.0);
            if (cost < best * 0.999) {
                best = cost;
                ns_rounds = c;
            }
        }
        // (2) Smaller problems (measured at 1,024^2 .. 16,384^2, tools/sweep_nsplit.py): a workgroup costs about four
        //     blocks of prologue; a grid that fits one workgroup per CU runs each of them at ~1/0.55 the shared speed;
        //     among near-ties the fewest splits win (the merge of the partials grows with them).
        best = 1e300;
        int64_t ns_model = 1;
        for (int64_t c = 1; c <= max_ns; ++c) {
            // (sym: only the workgroups on and above the diagonal -- about half -- have work, the others exit at once;
            //  with one resident round the fullest workgroup sets the time, so the range of a split must shrink)
            const int64_t wgs = sym ? lvae_ceil_div(groups * c, 2) : groups * c;
            const double bps = (double)lvae_ceil_div(p.nblk, c);
            const double cost = wgs <= 256 ? 0.55 * bps + 4.0 : (double)lvae_ceil_div(wgs, slots) * (bps + 4.0);
            if (cost < best * 0.97) {
                best = cost;
                ns_model = c;
            }
        }
        // (3) Kernel sums below the many-rounds regime (measured on the three-job MMD launch, tools/sweep_mmd_nsplit.py,
        //     n = 2,048 ... 32,768): their tiles carry twice the vector work and the symmetric jobs' workgroups are
        //     uneven, so fine splits pack better than the model above says -- down to four blocks per split, at most 32
        //     splits (16,384: 598 -> 542 us, 8,192: 215 -> 198, 4,096: 96 -> 80).
        if (!lse) {
            ns_model = p.nblk / 4 < 1 ? 1 : p.nblk / 4;
            if (ns_model > 32) ns_model = 32;
        }
        // (4) Logsumexp plans from ~12,000 reduced rows up to the many-rounds regime: sixteen splits at least.  One
        //     round of few large workgroups (when that is what the model above picks) loses to 1.5-3 rounds of smaller ones --
        //     measured, tools/sweep_nsplit.py, D = 32, pair kernel: 12,288^2 177 -> 155 us, 24,576^2 624 -> 578,
        //     32,768^2 1,035 -> 1,017, 16,384^2 unchanged (267 / 265).
        if (lse && p.nblk >= 384 && ns_model < 16 && groups * ns_model <= slots) {
            best = 1e300;
            for (int64_t c = 16; c <= max_ns; ++c) {       // the same model, from sixteen splits on
                const double bps = (double)lvae_ceil_div(p.nblk, c);
                const double cost = (double)lvae_ceil_div(groups * c, slots) * (bps + 4.0);
                if (cost < best * 0.97) {
                    best = cost;
                    ns_model = c;
                }
            }
        }
        ns = lvae_ceil_div(p.nblk, ns_rounds) >= 128 ? ns_rounds : ns_model;
    }
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    p.blk_per_split = lvae_ceil_div(p.nblk, ns);
    p.nsplit = (int)lvae_ceil_div(p.nblk, p.blk_per_split);
    // direct grid: one lane per kept element, 256 per workgroup, (cpad / 256) x dsplit equal workgroups, about
    // six of them resident per CU (1536 on the chip).  Measured with the prod-marginals kernel: three rounds
    // of workgroups beat one by 9 % on large problems, while on small ones every extra split costs more in
    // partials (dsplit x cpad x DC x 8 B, written and merged) than it gains: aim at 4608 / 1536 / 768
    // workgroups by size, never fewer than 128 reduced rows per split (32 for latent rows wider than 64).
    {
        const int64_t dgroups = p.cpad / 256 > 0 ? p.cpad / 256 : 1;
        const int64_t target = dgroups >= 192 ? 4608 : (dgroups >= 96 ? 1536 : 768);
        int64_t ds = lvae_ceil_div(target, dgroups);
        if (lvae_switches().dsplit) ds = lvae_switches().dsplit;
        // (wide latent rows carry D / 32 times the work per pair: a training batch of 256 at D = 768 is worth
        //  eight workgroups, not two)
        const int64_t min_rows = D > 64 ? 32 : 128;
        if (ds > R / min_rows) ds = R / min_rows;
        if (ds > 64) ds = 64;
        if (ds < 1) ds = 1;
        p.rows_per_dsplit = lvae_round_up(lvae_ceil_div(R, ds), 32);    // whole 32-row blocks (the bf16 prod-marginals
                                                                        // kernel streams blocks; same ranges for its redo)
        p.dsplit = (int)lvae_ceil_div(R, p.rows_per_dsplit);
    }
    p.DC = D <= 16 ? 16 : (D <= 32 ? 32 : (D <= 64 ? 64 : (int)lvae_round_up(D, 16)));

    size_t o = 0;
    p.off_hdr = o;      o = lvae_align256(o + sizeof(PairHeader));
    p.off_redo = o;     // redo flags of the fixed-reference logsumexp pass: one int per MFMA workgroup column group
    o = lvae_align256(o + (size_t)(p.cpad / 128 + 1) * sizeof(int));   // (right behind the header: one memset clears both)
    // (x 1.5: the experimental bf16x3 path stores three bf16 limbs per value in the same regions)
    p.off_rfrag = o;    o = lvae_align256(o + (size_t)p.nblk * (p.DP ? p.DP : 0) * 96 * sizeof(float));
    p.off_rbias = o;    o = lvae_align256(o + ((size_t)p.nblk * 32 + 64) * sizeof(float));
    p.off_cop = o;      o = lvae_align256(o + (size_t)p.cpad * 3 * p.DP * sizeof(float));
    p.off_colbias = o;  o = lvae_align256(o + (size_t)p.cpad * sizeof(float));
    p.off_partial = o;  o = lvae_align256(o + (size_t)p.nsplit * p.cpad * 2 * sizeof(float));
    p.off_dparams = o;  o = lvae_align256(o + (size_t)R * 3 * p.DC * sizeof(float));
    p.off_dpartial = o; o = lvae_align256(o + (size_t)p.dsplit * p.cpad * 2 * sizeof(float));
    p.off_dpartial_pm = o;
    if (with_pm) o = lvae_align256(o + (size_t)p.dsplit * p.cpad * p.DC * 2 * sizeof(float));
    p.off_dflags = o;   // redo flags of the prod-marginals fast pass: [dsplit][DC/16][cpad/64]
    if (with_pm) o = lvae_align256(o + (size_t)p.dsplit * (p.DC / 16) * (p.cpad / 64) * sizeof(int));
    p.off_pmtmp = o;    // prod-marginals: per-(dimension, kept element) log-terms before the sum over dimensions
    if (with_pm) o = lvae_align256(o + (size_t)p.DC * p.cpad * sizeof(float));
    // limb operands of the experimental bf16 prod-marginals kernel (pair_pm_bx.hip): 1 KiB per posterior / sample
    p.off_pmbx_a = o;
    if (with_pm && p.DP != 0 && p.DC <= 128) o = lvae_align256(o + (size_t)p.nblk * p.DC * 1024);
    p.off_pmbx_b = o;
    if (with_pm && p.DP != 0 && p.DC <= 128) o = lvae_align256(o + (size_t)p.cpad * p.DC * 32);
    p.total = o;
    return p;
}
