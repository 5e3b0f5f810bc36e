// Fused pair log-density + streaming logsumexp on the fp32 matrix cores (gfx950).
//
//   out[c] = log2 sum_r 2^( sum_k Rop[r][k] * Cop[c][k] + rbias[r] )          (base-2 domain)
//
// r runs over the REDUCED axis (posteriors for log q(z)), c over the KEPT axis (samples).  Operands
// come from prep.hip.  Each wave owns 32*NCB kept elements for its whole life: their operand rows sit
// in VGPRs as the B operand of v_mfma_f32_32x32x2_f32 (lane = column + 32*half).  The wave then streams
// 32-row blocks of the reduced axis through an LDS ring shared by the workgroup's four waves and filled
// by LDS-DMA (one 1 KiB piece per 4 MFMA steps, fragment-major layout), the accumulators start from the
// row bias, DP chained MFMAs per 32x32 tile produce the base-2 log-densities, and the epilogue folds the
// 16 values a lane holds (16 reduced rows of its own column) into that lane's sum.  The reduction axis
// lives on the MFMA ROW index, i.e. in registers, so the logsumexp needs no cross-lane traffic until the
// very end, where the two half-waves (rows +0..3 / +4..7 of every 8) are merged with one wave shuffle.
//
// Two logsumexp epilogues (template parameter EPI):
//   EPI 4  fixed reference: lse_shift_bias_kernel has folded -(largest density normaliser - 100) into the
//          row biases, every term is <= 2^100, and a tile costs 16 v_exp_f32 + 8 v_pk_add_f32.  Columns
//          whose total underflows are flagged by finalize_lse_kernel and redone with
//   EPI 0  running maximum, updated lazily: the tile sum of exponentials relative to the running max
//          proves (all terms are positive) that no term exceeded it by LSE_THR bits; otherwise the wave
//          redoes the tile against the true maximum.  Sums stay below 2^LSE_THR * R.
//
// Bound: fp32 MFMA (157.3 TFLOP/s); algorithmic cost 4*D flop per pair.  HBM traffic is negligible
// (operands are L2/Infinity-Cache resident and re-read by every wave).
//
// BX (template parameter, opt-in, experimental): the same contraction on the bf16 matrix pipe at
// fp32-equivalent accuracy.  Every fp32 operand value is three bf16 limbs v = h + m + l (prep_bf16x3_kernel);
// a product is the sum of the six limb products of weight >= 2^-16 (hh, hm, mh, hl, lh, mm; the dropped
// three are below 2^-24 of the product), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  One MFMA step
// covers eight dimensions: k-half 0 of the A / B operands holds the eight quadratic terms, k-half 1 the
// eight linear ones.  A ring stage is then a whole 32-row block (DP/8 steps x 3 limbs x 1 KiB); everything
// else -- ring, bias as the first C operand, ping-pong accumulators, every epilogue -- is shared.
#ifndef LVAE_PAIR_LSE_KERNEL_H
#define LVAE_PAIR_LSE_KERNEL_H
#include "pair_plan.h"

#define LSE_THR 8.0f
#define LSE_BIG 256.0f      // 2^LSE_THR

// s_waitcnt vmcnt(n) for a small n known after unrolling
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#define LSE_MINIT (-1.0e30f)

// EPI 0 / 4: streaming logsumexp (log q(z), q_z_estimate, MI bounds), see above.
// EPI 1-3: kernel-matrix column sums.  The contraction is -2 x.y (POINT x SAMPLE operands, two dimensions per
//        MFMA step), the squared distance d2 = |x|^2 + |y|^2 - 2 x.y is completed by the row bias |y|^2 and
//        the per-column |x|^2, and every pair contributes f(d2) = sum_k exp(-alpha_k * (eps + |d2|)),
//        accumulated per column in float64 (one tile's 16 terms are first summed in float32).
//        EPI 1: one bandwidth; EPI 2: alpha_k = base * (integer <= 16), f is a polynomial in
//        t = exp(-base*v) evaluated by Horner's rule on the packed fp32 pipe (one v_exp_f32 per pair for
//        any number of bandwidths; EPI 5: the same with paired coefficients, see KC below); EPI 3: one
//        exponential per bandwidth.
//        sym: a sample against itself -- only the tiles on and above the diagonal are visited.
// PERM (with EPI 1/2/5, NCB == 1): label-permutation sums for the MMD permutation test.  Every 32x32
//        tile of kernel values K[r][c] is, straight from the accumulator layout, the A operand
//        (i = column c, k = row r) of v_mfma_f32_32x32x16_bf16 against 0/1 label fragments
//        L[r][p] (128 labelings, four 32-wide tiles), split into two bf16 limbs (hi + lo, ~17 bits):
//        Y[c][p] += sum_r K[r][c] L[r][p].  Every PERM_FLUSH blocks the lane folds
//        sum_c L[c][p] Y[c][p] into a float64 and clears Y; at the end q_p = L_p^T K L_p partials go
//        to `qperm` with float64 atomics.
#ifndef PERM_FLUSH
#define PERM_FLUSH 64
#endif
// Development switches of this template (A/B builds; the defaults are the product):
//   LVAE_EPI_ANCHOR  1: every epilogue micro-slice ends in an empty volatile asm that names its results, so that no IR
//                    pass can sink the slice's arithmetic past the MFMA slots it was placed between (without it the
//                    split-bf16 permutation kernel came out as 12 MFMAs, then the whole epilogue in one lump)
//   LVAE_COND_MASK   1: the diagonal-exclusion mask is only built for blocks that hold the diagonal (wave-uniform branch
//                    between two copies of the block body)
//   LVAE_KS_F32      1: PERM: tile sums are folded in float32 between flushes, float64 only at a flush
#ifndef LVAE_EPI_ANCHOR
#define LVAE_EPI_ANCHOR 1
#endif
#ifndef LVAE_COND_MASK
#define LVAE_COND_MASK 1
#endif
#ifndef LVAE_KS_F32
#define LVAE_KS_F32 1
#endif

typedef __bf16 lvae_bf16x8 __attribute__((ext_vector_type(8)));

// (the body is a device function: pair_lse_mfma_kernel below runs one job per launch, pair_lse_multi_kernel up to
//  three kernel-sum jobs selected by blockIdx.z)
template <int DP, int NCB, int WPS, bool EXCL, int EPI, bool PERM = false, bool BX = false>
__device__ __forceinline__ void
pair_lse_body(const float* __restrict__ rfrag, const float* __restrict__ rbias,
              const float* __restrict__ cop, float2* __restrict__ partial,
              const PairHeader* __restrict__ hdr, float cond_limit, int use_if_above,
              int64_t nblk, int64_t blk_per_split, int64_t cpad, int64_t excl_offset,
              const PermArgs& pa, const int* __restrict__ redo_flags, int sym,
              const float* __restrict__ colnorm) {
    static_assert(!PERM || (NCB == 1 && (EPI == 1 || EPI == 2 || EPI == 5)), "PERM needs NCB == 1 and a polynomial epilogue");
    constexpr bool KS = EPI == 1 || EPI == 2 || EPI == 3 || EPI == 5;      // kernel-sum epilogues
    static_assert(!BX || DP <= 64, "the split-bf16 contraction is built for D <= 64");
    // Path gate: the conditioning estimate written by prep decides between this kernel and the direct
    // one without a host round trip (both are launched; the unselected one exits here).
    if (cond_limit >= 0.f) {
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    // redo pass of the fixed-reference logsumexp: only the flagged workgroups have work
    if (redo_flags && redo_flags[blockIdx.x] == 0) return;
    if (EPI == 4 && hdr->lse_skip) return;       // the fixed reference cannot hold this call's range of normalisers
    constexpr int NQ = DP / 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * NCB);

    // kept-axis operand: resident for the whole kernel
    constexpr int NS = DP / 8;                    // BX: MFMA steps (8 dimensions each) per tile
    float bz[BX ? 1 : NCB][BX ? 1 : DP];
    lvae_bf16x8 bzx[BX ? NCB : 1][BX ? NS : 1][3];        // BX: [column block][step][limb]
    if constexpr (BX) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const uint4* src = reinterpret_cast<const uint4*>(cop) + ((col0 + cb * 32 + r) * 2 + h) * NS * 3;
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int l = 0; l < 3; ++l) bzx[cb][s][l] = __builtin_bit_cast(lvae_bf16x8, src[s * 3 + l]);
        }
    } else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const f32x4* src = reinterpret_cast<const f32x4*>(cop + (col0 + cb * 32 + r) * (2 * DP) + h * DP);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4 v = src[q];
                bz[cb][4 * q + 0] = v[0];
                bz[cb][4 * q + 1] = v[1];
                bz[cb][4 * q + 2] = v[2];
                bz[cb][4 * q + 3] = v[3];
            }
        }
    }

    float m[NCB], s[NCB];
    double ks[NCB];       // EPI 1: float64 column sums
    double ksd = 0.0;     // PERM + sym: column sums of the diagonal tile
    float ksf = 0.f;      // PERM (LVAE_KS_F32): tile sums since the last flush (<= PERM_FLUSH tiles of <= 16 * n_alpha each)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        m[cb] = LSE_MINIT;
        s[cb] = 0.f;
        ks[cb] = 0.0;
    }

    // sym: the splits are walked from the LAST row range to the first.  Work per workgroup grows with the row range
    // (a column block only visits rows from its diagonal on), and the hardware dispatches in blockIdx order: the
    // fullest workgroups must go first, the empty ones last, or the launch ends on a tail of full workgroups.
    const int64_t ysplit = (KS && sym != 0) ? (nblk + blk_per_split - 1) / blk_per_split - 1 - (int64_t)blockIdx.y
                                            : (int64_t)blockIdx.y;
    int64_t b0 = ysplit * blk_per_split;
    int64_t b1 = b0 + blk_per_split;
    if (b1 > nblk) b1 = nblk;
    // sym (kernel sums of a sample against itself, totals only): the matrix is symmetric, so a column
    // block only visits the row blocks from its own diagonal tile on (weight 1) and counts the tiles
    // beyond it twice; the workgroup starts at its first column block's diagonal tile.
    const bool use_sym = KS && sym != 0;
    const int64_t sym_off = excl_offset / 32;    // block index of kept element 0 on the reduced axis (its diagonal)
    if (use_sym) {
        const int64_t gs = sym_off + (int64_t)blockIdx.x * 4 * NCB;
        if (b0 < gs) b0 = gs;
    }
    float wgt[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) wgt[cb] = 1.f;
    auto set_weights = [&](int64_t pblk) {
        if (use_sym) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int64_t cbi = sym_off + col0 / 32 + cb;
                wgt[cb] = pblk > cbi ? 2.f : (pblk == cbi ? 1.f : 0.f);
            }
        }
    };

    // ---- streaming of the reduced-axis operand through LDS -------------------------------------
    // The four waves of a workgroup sweep the same reduced-axis blocks (each for its own columns), so
    // they share one ring of PF + 1 stages in LDS (PF = 3 chunks in flight) and fill it together with LDS-DMA
    // (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR staging; each wave brings a quarter
    // of every chunk).  The fragment-major global layout is lane-linear, which is exactly the image an
    // LDS-DMA writes, and the later ds_read_b128 of lane l at offset 16*l is conflict-free.
    // One barrier per chunk: before it every wave drains its own DMAs (vmcnt) of the chunk issued at
    // the top of this one (two chunks ahead of the one being consumed); after it that chunk is
    // complete for everybody and the stage just consumed may be overwritten.
    // (BX: one chunk = one block = NS steps x 3 limbs of 1 KiB each)
    constexpr int CS = BX ? DP : (DP < 32 ? DP : 32);     // dimensions (fp32: MFMA steps) per ring stage ("chunk")
    constexpr int CPB = DP / CS;              // chunks per 32-row block
    constexpr int CQ = BX ? NS * 3 : CS / 4;  // 1 KiB DMA pieces per chunk
    constexpr int CHUNK_F = CQ * 256;         // floats per chunk
    constexpr int STAGE_F = CHUNK_F + 64;     // + one 256-B piece holding the block's 32 row biases
#ifndef LVAE_PF
#define LVAE_PF 3
#endif
    // chunks the DMA runs ahead of the MFMAs (BX: 24 KiB stages at D = 64, and the label ring of PERM, leave
    // room for three stages only)
    constexpr int PF = BX ? ((DP >= 64 || PERM) ? 2 : 3) : LVAE_PF;
    constexpr int NST = PF + 1;
    __shared__ __attribute__((aligned(16))) float ring[NST * STAGE_F];
    // PERM: label fragments of a block are consumed one block after its operand chunks (with the
    // tile's epilogue), so they live in a ring of their own, deep enough to outlast that delay
    constexpr int NSTL = 2 + (PF + CPB - 1) / CPB;
    constexpr int LAB_F = 4 * 2 * 64 * 4;     // floats (= 8 KiB) of label fragments per block
    __shared__ __attribute__((aligned(16))) float lring[PERM ? NSTL * LAB_F : 4];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // global addresses = wave-uniform base (scalar registers) + a constant 32-bit lane offset, so that the
    // per-chunk address arithmetic stays on the scalar unit
    const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
    auto dma_chunk = [&](int64_t blk, int ch, int stage) {
        const char* g = BX ? reinterpret_cast<const char*>(rfrag) + blk * (int64_t)(CQ * 1024)
                           : reinterpret_cast<const char*>(rfrag + (blk * (DP / 4) + ch * CQ) * 256);
        float* l = ring + stage * STAGE_F;
#pragma unroll
        for (int q0 = 0; q0 < CQ; q0 += 4) {
            int q = q0 + wave_u;
            if (CQ % 4 != 0 && q >= CQ) q = CQ - 1;     // same DMA count for every wave; a duplicate is harmless
            __builtin_amdgcn_global_load_lds((gptr_t)(g + q * 1024 + lane16), (lptr_t)(l + q * 256), 16, 0, 0);
        }
        if (ch == 0) { // 64 lanes x 4 B: this block's 32 biases (+32 of the next block, unused); every wave
                       // writes the same bytes so that all waves have the same DMA count per chunk
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(rbias + blk * 32) + lane4),
                                             (lptr_t)(l + CHUNK_F), 4, 0, 0);
            if (PERM) {
                const float* lg = reinterpret_cast<const float*>(pa.labfrag) + blk * LAB_F + lane * 4;
                float* ll = lring + (int)((blk - b0) % NSTL) * LAB_F;
#pragma unroll
                for (int q0 = 0; q0 < 8; q0 += 4) {
                    const int q = q0 + wave_u;
                    __builtin_amdgcn_global_load_lds((gptr_t)(lg + q * 256), (lptr_t)(ll + q * 256), 16, 0, 0);
                }
            }
        }
    };
    // chunk index t (counted from this workgroup's first block) -> (block, chunk-in-block); the block is clamped
    // to the last one of the range, so that running ahead past the end is a harmless re-read -- into the stage and
    // with the DMA count of the regular pattern, so the counted waits stay exact
    auto dma_linear = [&](int t) {
        int64_t blk = b0 + t / CPB;
        if (blk > b1 - 1) blk = b1 - 1;
        dma_chunk(blk, t % CPB, t % NST);
    };

    // EXCL: drop the pair (reduced row == kept column + offset).  The row biases enter a tile as the C operand
    // of its first MFMA, one register per row and one lane per column -- exactly the (row, column) layout of
    // the pair -- so the pair is dropped by making that one bias entry infinite for the (wave-uniformly
    // chosen) tiles that contain it; the accumulators themselves are never touched by vector instructions.
    auto mask_bias = [&](f32x16& c, int cb, int64_t blk) {
        const int64_t row_lo = blk * 32;
        const int64_t cbase = col0 + cb * 32 + excl_offset;
        if (row_lo < cbase + 32 && row_lo + 32 > cbase) {
            const int want = (int)(cbase - row_lo) + r - 4 * h;   // row index inside the block, minus the half-wave's 4
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (want == 8 * g + e) c[4 * g + e] = KS ? INFINITY : -INFINITY;
        }
    };

    // Epilogue of one 32x32 tile, cut into EPI_MS micro-slices of <= ~10 VALU instructions so that
    // each one issues in the shadow of one MFMA of the NEXT block (64 cycles of matrix-pipe time):
    //   0..7: two exponentials each (packed subtract / add), summed into the tile sum
    //   8: tile sum -> running sum, or the rare redo against a new maximum.
    // (EPI 1-3: eight slices of two kernel values each, the ninth folds the tile sum into float64;
    //  EPI 4: eight slices of two exponentials, nothing else.)
    constexpr int EPI_MS = EPI == 4 ? 8 : 9;
    f32x2 m2[NCB], kt2[NCB];            // EPI 0: minus the running max, twice (packed operand); tile sum in flight
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        m2[cb] = (f32x2){-LSE_MINIT, -LSE_MINIT};
        kt2[cb] = (f32x2){0.f, 0.f};
    }
    float kv[PERM ? 16 : 1];            // PERM: the kernel values of the tile whose epilogue is running
    // Kernel-sum parameters, read once into (scalar) registers: EPI 1 = one bandwidth (c0 + c1*t),
    // EPI 2 = polynomial of degree <= 16 in t, fully unrolled Horner (unused high coefficients are 0),
    // EPI 3 = arbitrary bandwidths, one exponential each.
    // EPI 5 = polynomial of degree <= 14 whose coefficients come in pairs (c0 = 0, c[2j+1] == c[2j+2] = e_j):
    // t (1 + t) Q(t^2), nine packed operations instead of sixteen; the reference's fifteen bandwidths 0, 0.1 .. 1.4
    // are this case.
    constexpr int KC = EPI == 1 ? 2 : (EPI == 2 ? 17 : (EPI == 3 ? LVAE_KSUM_MAX : (EPI == 5 ? 7 : 1)));
    float kc[KC];
    float k_eps = 0.f;
    int k_n = 0;
    if (KS) {
        const KsumParams* kp = &hdr->ksum;
        k_eps = kp->eps;
        k_n = kp->n;
#pragma unroll
        for (int k = 0; k < KC; ++k) kc[k] = EPI == 5 ? kp->c[2 * k + 1] : kp->c[k];
    }
    // Kernel sums contract the LINEAR part of the squared distance only: the operands hold two dimensions per
    // fp32 MFMA step (POINT rows -2y against SAMPLE rows x; DP is the padded HALF dimension), the reduced
    // element's |y|^2 is the row bias and the kept element's |x|^2 (colnorm, one value per lane) is added here.
    float cn[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) cn[cb] = KS ? colnorm[col0 + cb * 32 + r] : 0.f;
    // two kernel values at a time (packed fp32 for the polynomial)
    auto kfun2 = [&](float d2a, float d2b, float cnv) -> f32x2 {
        if (EPI == 1 || EPI == 2 || EPI == 5) {
            // The operands of these epilogues arrive scaled (prep_rows: base log2e folded into the POINT rows, the
            // biases and the column norms, eps into the biases), so the contraction plus the column norm IS the
            // exponent up to sign: one packed add per pair of values, then v_exp_f32 with the negated absolute value
            // as an input modifier -- no multiply-add in front of the exponential
            const f32x2 x = (f32x2){d2a, d2b} + (f32x2){cnv, cnv};
            f32x2 t;
            t.x = lvae_exp2(-__builtin_fabsf(x.x));
            t.y = lvae_exp2(-__builtin_fabsf(x.y));
            if (EPI == 1) return __builtin_elementwise_fma((f32x2){kc[1], kc[1]}, t, (f32x2){kc[0], kc[0]});
            if (EPI == 5) {
                const f32x2 t2 = t * t;
                const f32x2 lin = t2 + t;                 // t (1 + t)
                f32x2 q = {kc[6], kc[6]};
#pragma unroll
                for (int k = 5; k >= 0; --k) q = __builtin_elementwise_fma(q, t2, (f32x2){kc[k], kc[k]});
                return lin * q;
            }
            f32x2 pacc = {kc[16], kc[16]};
#pragma unroll
            for (int k = 15; k >= 0; --k) pacc = __builtin_elementwise_fma(pacc, t, (f32x2){kc[k], kc[k]});
            return pacc;
        } else {
            const f32x2 v = {fabsf(d2a + cnv) + k_eps, fabsf(d2b + cnv) + k_eps};
            f32x2 f = {0.f, 0.f};
#pragma unroll 4
            for (int k = 0; k < KC; ++k)
                if (k < k_n) {
                    f.x += lvae_exp2(v.x * kc[k]);
                    f.y += lvae_exp2(v.y * kc[k]);
                }
            return f;
        }
    };
    auto epi_micro = [&](const f32x16 (&prev)[NCB], int cb, int ms) {
        if (EPI == 4) {
            // fixed reference (folded into the row biases by lse_shift_bias_kernel): every term is at
            // most 2^LSE_HEADROOM, so the tile is two exponentials and one packed add per slice
            f32x2 e;
            e.x = lvae_exp2(prev[cb][2 * ms]);
            e.y = lvae_exp2(prev[cb][2 * ms + 1]);
            kt2[cb] += e;
        } else if (EPI >= 1) {
            if (ms < 8) {
                const f32x2 kval = kfun2(prev[cb][2 * ms], prev[cb][2 * ms + 1], cn[cb]);
                if (ms == 0) kt2[cb] = kval;
                else kt2[cb] += kval;
                if (PERM) {
                    kv[2 * ms] = kval.x;
                    kv[2 * ms + 1] = kval.y;
                }
#if LVAE_EPI_ANCHOR
                if (PERM) asm volatile("" : "+v"(kt2[cb]), "+v"(kv[2 * ms]), "+v"(kv[2 * ms + 1]));
                else asm volatile("" : "+v"(kt2[cb]));
#endif
            } else if (PERM) {
                // strictly-upper (or, without sym, all) tiles and the diagonal tile are kept apart
                const float ts = kt2[cb].x + kt2[cb].y;
                if (use_sym && wgt[cb] == 1.f) ksd += (double)ts;
                else if (!use_sym || wgt[cb] == 2.f) {
                    if (LVAE_KS_F32) ksf += ts;
                    else ks[cb] += (double)ts;
                }
            } else {
                ks[cb] += (double)((kt2[cb].x + kt2[cb].y) * wgt[cb]);
            }
        } else if (ms < 8) {
            // two pairs per slice on the packed fp32 pipe: subtract the running max, exponentiate,
            // accumulate the tile sum (the first slice starts it)
            const f32x2 pr = {prev[cb][2 * ms], prev[cb][2 * ms + 1]};
            f32x2 x;    // (the compiler would split this packed add into two scalar ones)
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(pr), "v"(m2[cb]));
            f32x2 e;
            e.x = lvae_exp2(x.x);
            e.y = lvae_exp2(x.y);
            if (ms == 0) kt2[cb] = e;
            else kt2[cb] += e;
        } else {
            // Lazy running max without a max tree: all terms are positive, so a tile sum below 2^LSE_THR
            // proves that no term exceeded the running max by LSE_THR bits and the sum is simply added.
            // Otherwise (the first tile, or a rare jump of the maximum; inf/NaN land here too) the whole
            // wave redoes the tile against the true maximum.
            const float ts = kt2[cb].x + kt2[cb].y;
            if (__builtin_expect(__any(!(ts < LSE_BIG)), 0)) {
                float t = prev[cb][0];
#pragma unroll
                for (int e = 1; e < 16; ++e) t = fmaxf(t, prev[cb][e]);
                const float mn = fmaxf(m[cb], t);
                float acc = s[cb] * lvae_exp2(m[cb] - mn);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc += lvae_exp2(prev[cb][e] - mn);
                s[cb] = acc;
                m[cb] = mn;
                m2[cb] = (f32x2){-mn, -mn};
            } else {
                s[cb] += ts;
            }
        }
    };
    // ---- PERM: label-permutation accumulators --------------------------------------------------
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    f32x16 Y[PERM ? 4 : 1];
    double qacc[PERM ? 4 : 1];
    double yacc[PERM ? 4 : 1];          // unweighted sums of Y (sym: one half of lr_p)
    unsigned cmask[PERM ? 4 : 1];       // bit e: label of kept element (accumulator row e) under labeling 32*pt + r
    unsigned vmask = 0;                 // bit e: accumulator row e is a real kept element (not padding)
    int pcount = 0;
    if (PERM) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (col0 + (e & 3) + 8 * (e >> 2) + 4 * h < pa.C) vmask |= 1u << e;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            unsigned mk = 0;
            const int perm = 32 * pt + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t c = col0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (perm < pa.P && c < pa.C && pa.labels[(int64_t)perm * pa.n_lab + pa.col_offset + c]) mk |= 1u << e;
            }
            cmask[pt] = mk;
            qacc[pt] = 0.0;
            yacc[pt] = 0.0;
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[pt][e] = 0.f;
        }
    }
    auto perm_flush = [&](float wv) __attribute__((always_inline)) {
        // (the masks are re-read through an opaque asm: left to itself the compiler hoists all 80 per-bit lane masks
        //  out of the loop as scalar register pairs, spills them into vector lanes and pays 130 v_readlane per flush)
        const bool full = __builtin_amdgcn_readfirstlane((int)__all(vmask == 0xFFFFu)) != 0;
#pragma unroll
        for (int pt = 0; pt < (PERM ? 4 : 0); ++pt) {
            unsigned cm = cmask[pt], vm = vmask;
            asm volatile("" : "+v"(cm), "+v"(vm));
            float t = 0.f, u = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) t = __builtin_fmaf((float)((cm >> e) & 1u), Y[pt][e], t);
            if (full) {       // (one wave-uniform branch per labeling tile, not one per element)
#pragma unroll
                for (int e = 0; e < 16; ++e) u += Y[pt][e];
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) u = __builtin_fmaf((float)((vm >> e) & 1u), Y[pt][e], u);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[pt][e] = 0.f;
            qacc[pt] += (double)(t * wv);
            yacc[pt] += (double)u;
        }
        if (LVAE_KS_F32) {
            ks[0] += (double)ksf;
            ksf = 0.f;
        }
    };
    // kv (the finished tile, accumulator layout: lane = column, 16 rows) -> two bf16 limbs -> 16 MFMAs
    // wv: weight of the tile (sym: 0 below the diagonal tile -> nothing to do, 1 on it, 2 beyond; else 1).
    // A flush never mixes weights: the diagonal tile is flushed on its own.
    auto perm_tile = [&](int lstage, float wv) __attribute__((always_inline)) {
        // (a scalar branch, not a test of the per-lane copy of the weight: MFMA ignores EXEC, and the compiler may lower a
        //  short conditional block to EXEC masking without a jump)
        if (__builtin_amdgcn_readfirstlane(__float_as_int(wv)) == 0) return;
        const bf16x8* lb = reinterpret_cast<const bf16x8*>(lring + lstage * LAB_F);
        bf16x8 bcur = lb[lane];            // (the limb split below covers this read's latency)
        bf16x8 ahi[2], alo[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const f32x2 v = {kv[8 * sp + 2 * e2], kv[8 * sp + 2 * e2 + 1]};
                const bf16x2 hh = __builtin_convertvector(v, bf16x2);
                const f32x2 hf = __builtin_convertvector(hh, f32x2);
                const bf16x2 ll = __builtin_convertvector(v - hf, bf16x2);
                ahi[sp][2 * e2] = hh[0];
                ahi[sp][2 * e2 + 1] = hh[1];
                alo[sp][2 * e2] = ll[0];
                alo[sp][2 * e2 + 1] = ll[1];
            }
#pragma unroll
        for (int pt = 0; pt < (PERM ? 4 : 0); ++pt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 b = bcur;
                if (pt * 2 + sp + 1 < 8) bcur = lb[(pt * 2 + sp + 1) * 64 + lane];
                Y[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[sp], b, Y[pt], 0, 0, 0);
                Y[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[sp], b, Y[pt], 0, 0, 0);
            }
        if ((use_sym && wv == 1.f) || ++pcount == PERM_FLUSH) {
            perm_flush(wv);
            pcount = 0;
        }
    };

    constexpr int SLOTS = BX ? NS * 6 * NCB : DP * NCB;   // MFMAs per block
    constexpr int MS_TOTAL = EPI_MS * NCB;             // micro-slices per block
    // Kernel-sum epilogues issue their slices in PAIRS behind one MFMA: a slice is one dependent chain of packed fp32
    // operations (the polynomial), and the hardware wants a wait state between dependent packed operations -- two
    // independent chains in one scheduling region fill each other's wait states (47 s_nop per tile otherwise).
    constexpr int MSG = KS ? 2 : 1;                    // slices per group
    constexpr int NG = (MS_TOTAL + MSG - 1) / MSG;     // groups per block

    // Values carried from the tail of one chunk into the next: the first 1 KiB piece of the next chunk
    // and (at a block boundary) the next block's row biases, both already read from LDS.
    f32x4 a_pre;
    lvae_bf16x8 ax_pre[3];                  // BX: the three limbs of step 0 of the next block
    f32x16 c_pre;
    auto read_piece = [&](int stage, int q) -> f32x4 {
        return reinterpret_cast<const f32x4*>(ring + stage * STAGE_F)[q * 64 + lane];
    };
    auto read_step = [&](int stage, int s, lvae_bf16x8 (&dst)[3]) {
        const uint4* p = reinterpret_cast<const uint4*>(ring + stage * STAGE_F) + s * 3 * 64 + lane;
#pragma unroll
        for (int l = 0; l < 3; ++l) dst[l] = __builtin_bit_cast(lvae_bf16x8, p[l * 64]);
    };
    // BX: limb pairs of the six kept products, smallest first
    constexpr int LA[6] = {1, 2, 0, 1, 0, 0};     // limb of the reduced-axis (A) operand
    constexpr int LB[6] = {1, 0, 2, 0, 1, 0};     // limb of the kept-axis (B) operand
    auto read_bias = [&](int stage) {
        const f32x4* bl = reinterpret_cast<const f32x4*>(ring + stage * STAGE_F + CHUNK_F) + h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 c = bl[2 * g];
#pragma unroll
            for (int e = 0; e < 4; ++e) c_pre[4 * g + e] = c[e];
        }
    };

    // One block: keep the ring one chunk ahead, run the DP*NCB MFMAs into `acc` (first step takes the
    // row biases as its C operand) while the epilogue of `prev` issues in their shadow.  Everything is
    // pinned in program order (sched_barrier) so the compiler cannot bunch the epilogue up.
    int t_lin = 0;                                         // chunks consumed so far by this wave
    auto run_block = [&](f32x16 (&acc)[NCB], const f32x16 (&prev)[NCB], int64_t blk, bool with_epi) __attribute__((always_inline)) {
        const f32x16 cv = c_pre;
        // EXCL: only a block whose rows meet one of the wave's column blocks on the excluded diagonal needs masked
        // biases (16 moves + 16 compare / select per column block).  LVAE_COND_MASK: the first MFMA of the chain is
        // issued from one of two places under a wave-uniform branch -- with the masked copy, built on the spot, or
        // straight from the biases as they were read.
        constexpr bool MASKED = EXCL && !LVAE_COND_MASK;      // every block builds the masked copy up front
        bool hit = false;
        if constexpr (EXCL && LVAE_COND_MASK) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int64_t cbase = col0 + cb * 32 + excl_offset;
                hit = hit || (blk * 32 < cbase + 32 && blk * 32 + 32 > cbase);
            }
            hit = __builtin_amdgcn_readfirstlane((int)hit) != 0;
        }
        f32x16 cvm[MASKED ? NCB : 1];
        if constexpr (MASKED) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                cvm[cb] = cv;
                mask_bias(cvm[cb], cb, blk);
            }
        }
        if (KS) set_weights(blk - 1);
        if constexpr (BX) {
            const int stage = t_lin % NST;
            const int stage_next = (t_lin + 1) % NST;
            dma_linear(t_lin + PF);    // into the stage consumed one block ago (freed by the last barrier)
            lvae_bf16x8 ax[2][3];
#pragma unroll
            for (int l = 0; l < 3; ++l) ax[0][l] = ax_pre[l];
#pragma unroll
            for (int sx = 0; sx < NS; ++sx) {
                __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the step read one step ago has landed
                if (sx + 1 < NS) {
                    read_step(stage, sx + 1, ax[(sx + 1) & 1]);
                } else {
                    // block t_lin+1 was certified complete by the barrier that ended block t_lin-1
                    read_step(stage_next, 0, ax_pre);
                    read_bias(stage_next);
                }
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        if (EXCL && LVAE_COND_MASK && sx == 0 && pr == 0 && hit) {
                            f32x16 cm = cv;
                            mask_bias(cm, cb, blk);
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[0][LA[0]], bzx[cb][0][LB[0]], cm, 0, 0, 0);
                        } else {
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[sx & 1][LA[pr]], bzx[cb][sx][LB[pr]],
                                                                          (sx == 0 && pr == 0) ? (MASKED ? cvm[MASKED ? cb : 0] : cv) : acc[cb],
                                                                          0, 0, 0);
                        }
                        const int u = (sx * 6 + pr) * NCB + cb;
                        if (with_epi) {
#pragma unroll
                            for (int gk = (u * NG) / SLOTS; gk < ((u + 1) * NG) / SLOTS; ++gk)
#pragma unroll
                                for (int k = gk * MSG; k < (gk + 1) * MSG && k < MS_TOTAL; ++k)
                                    epi_micro(prev, k / EPI_MS, k % EPI_MS);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (PERM && with_epi) perm_tile((int)((blk - 1 - b0) % NSTL), wgt[0]);
            // the NEXT block's last step reads step 0 and the biases of block t_lin+2: my DMAs up to that
            // block must have landed (block t_lin+3, if the ring is that deep, may still fly)
            wait_vmcnt((PF - 2) * ((CQ + 3) / 4 + (PERM ? 3 : 1)));
            __builtin_amdgcn_s_barrier();
            ++t_lin;
        } else {
#pragma unroll
        for (int ch = 0; ch < CPB; ++ch) {
            const int stage = t_lin % NST;
            const int stage_next = (t_lin + 1) % NST;
#ifndef LVAE_DBG_NODMA   // (ablation builds for DESIGN.md: -DLVAE_DBG_NODMA / -DLVAE_DBG_NOBAR; results are garbage)
            dma_linear(t_lin + PF);    // into the stage consumed one chunk ago (freed by the last barrier)
#endif
            f32x4 a4[2];
            a4[0] = a_pre;
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                // The piece read one group ago has had four MFMAs to land: wait for it BEFORE issuing the
                // next read, otherwise the compiler's own lgkmcnt(0) lands after the new read and stalls
                // the wave for a full LDS round trip.  (0xC07F = lgkmcnt(0), vmcnt/expcnt untouched.)
                __builtin_amdgcn_s_waitcnt(0xC07F);
                if (q + 1 < CQ) {
                    a4[(q + 1) & 1] = read_piece(stage, q + 1);
                } else {
                    // chunk t_lin+1 was certified complete by the barrier that ended chunk t_lin-1
                    a_pre = read_piece(stage_next, 0);
                    if (ch == CPB - 1) read_bias(stage_next);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int st = ch * CS + q * 4 + e;
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        if (EXCL && LVAE_COND_MASK && st == 0 && hit) {
                            f32x16 cm = cv;
                            mask_bias(cm, cb, blk);
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0][0], bz[cb][0], cm, 0, 0, 0);
                        } else
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q & 1][e], bz[cb][st],
                                                                       st == 0 ? (MASKED ? cvm[MASKED ? cb : 0] : cv) : acc[cb], 0, 0, 0);
                        const int u = st * NCB + cb;
                        if (with_epi) {   // micro-slices (kernel sums: pairs of them) spread evenly over the block's MFMA slots
#pragma unroll
                            for (int gk = (u * NG) / SLOTS; gk < ((u + 1) * NG) / SLOTS; ++gk)
#pragma unroll
                                for (int k = gk * MSG; k < (gk + 1) * MSG && k < MS_TOTAL; ++k)
                                    epi_micro(prev, k / EPI_MS, k % EPI_MS);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (PERM && with_epi && ch == CPB - 1) perm_tile((int)((blk - 1 - b0) % NSTL), wgt[0]);
            // my pieces of chunk t_lin+2 have landed; after the barrier everybody's have, and
            // everybody is done reading chunk t_lin's stage
            // (chunks t_lin+3 .. t_lin+PF, issued later, may stay in flight: counted wait)
            {
                constexpr int per_chunk = CQ / 4;
                int left = 0;
#pragma unroll
                for (int k = 3; k <= PF; ++k) left += per_chunk + (((ch + k) % CPB) == 0 ? (PERM ? 3 : 1) : 0);
                wait_vmcnt(left);
            }
#ifndef LVAE_DBG_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            ++t_lin;
        }
        }
    };
    auto drain = [&](const f32x16 (&prev)[NCB]) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int ms = 0; ms < EPI_MS; ++ms) epi_micro(prev, cb, ms);
    };

    f32x16 accA[NCB], accB[NCB];
    if (b0 < b1) {
#pragma unroll
        for (int k = 0; k < PF; ++k) dma_linear(k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (BX) read_step(0, 0, ax_pre);
        else a_pre = read_piece(0, 0);
        read_bias(0);
        run_block(accA, accB, b0, false);
        int64_t blk = b0 + 1;
        // two blocks per trip so the accumulator sets swap roles without register copies
        for (; blk + 1 < b1; blk += 2) {
            run_block(accB, accA, blk, true);
            run_block(accA, accB, blk + 1, true);
        }
        if (KS) set_weights(b1 - 1);
        if (blk < b1) {
            run_block(accB, accA, blk, true);
            set_weights(b1 - 1);
            drain(accB);
        } else {
            drain(accA);
        }
        if (PERM) {
            perm_tile((int)((b1 - 1 - b0) % NSTL), wgt[0]);
            if (pcount) perm_flush(wgt[0]);
            if (LVAE_KS_F32) ks[0] += (double)ksf;      // (tiles of weight 0 fold nothing into Y but do count here)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const double tot = qacc[pt] + __shfl_xor(qacc[pt], 32, 64);
                const double ytot = yacc[pt] + __shfl_xor(yacc[pt], 32, 64);
                if (h == 0 && 32 * pt + r < pa.P) {
                    atomicAdd(&pa.qperm[32 * pt + r], tot);
                    if (use_sym) atomicAdd(&pa.qperm[128 + 32 * pt + r], ytot);
                }
            }
        }
    }

    // merge the two half-waves (same column, complementary rows) and publish
    if (EPI == 4) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const float sv = kt2[cb].x + kt2[cb].y;
            const float ss = sv + __shfl_xor(sv, 32, 64);
            if (h == 0) partial[ysplit * cpad + col0 + cb * 32 + r] = make_float2(0.f, ss);
        }
        return;
    }
    if (EPI >= 1) {
        double* kpartial = reinterpret_cast<double*>(partial);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const double tot = ks[cb] + __shfl_xor(ks[cb], 32, 64);
            if (h == 0) kpartial[ysplit * cpad + col0 + cb * 32 + r] = tot;
        }
        if (PERM) {
            const double dtot = ksd + __shfl_xor(ksd, 32, 64);
            if (h == 0) pa.diag_partial[ysplit * cpad + col0 + r] = dtot;
        }
        return;
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const float mo = __shfl_xor(m[cb], 32, 64);
        const float so = __shfl_xor(s[cb], 32, 64);
        const float mm = fmaxf(m[cb], mo);
        const float ss = s[cb] * lvae_exp2(m[cb] - mm) + so * lvae_exp2(mo - mm);
        if (h == 0) partial[ysplit * cpad + col0 + cb * 32 + r] = make_float2(mm, ss);
    }
}

template <int DP, int NCB, int WPS, bool EXCL, int EPI, bool PERM = false, bool BX = false>
__global__ void __launch_bounds__(256, WPS)
pair_lse_mfma_kernel(const float* __restrict__ rfrag, const float* __restrict__ rbias,
                     const float* __restrict__ cop, float2* __restrict__ partial,
                     const PairHeader* __restrict__ hdr, float cond_limit, int use_if_above,
                     int64_t nblk, int64_t blk_per_split, int64_t cpad, int64_t excl_offset,
                     PermArgs pa, const int* __restrict__ redo_flags, int sym,
                     const float* __restrict__ colnorm) {
    pair_lse_body<DP, NCB, WPS, EXCL, EPI, PERM, BX>(rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, nblk,
                                                     blk_per_split, cpad, excl_offset, pa, redo_flags, sym, colnorm);
}

// Up to three kernel-sum jobs in ONE launch (the three kernel matrices of an MMD: cross term first, it is the
// fullest): blockIdx.z picks the job, workgroups beyond a job's own grid exit.  One ramp-up and one tail instead
// of three, and the uneven workgroups of the symmetric jobs fill in behind each other.

template <int DP, int NCB, int WPS, bool EXCL, int EPI>
__global__ void __launch_bounds__(256, WPS)
pair_lse_multi_kernel(const PairJobs jobs, const PairHeader* __restrict__ hdr) {
    const PairJob& J = jobs.j[blockIdx.z];
    if ((int)blockIdx.x >= J.gx || (int)blockIdx.y >= J.gy) return;
    const PermArgs pa{};
    pair_lse_body<DP, NCB, WPS, EXCL, EPI, false, false>(J.rfrag, J.rbias, J.cop, J.partial, hdr, -1.f, 0, J.nblk,
                                                         J.blk_per_split, J.cpad, J.excl_offset, pa, nullptr, J.sym,
                                                         J.colnorm);
}
#endif  // LVAE_PAIR_LSE_KERNEL_H
