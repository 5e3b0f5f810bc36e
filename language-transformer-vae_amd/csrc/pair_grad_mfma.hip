// Backward of log q(z) (the joint density of approximate_log_q_z, loss_and_optimisation.py:139-221, differentiated
// by autograd inside the reference's objectives :925-1018) on the fp32 matrix cores, for analysis-scale batches
// (D <= 128; nothing of size M x N is stored).
//
// With p_ij = exp(v_ij - L_i) the softmax weight of posterior j for sample i (v_ij the pair log-density, L_i its
// logsumexp over j = log_qz_i + log_norm) and g_i the upstream gradient on log_qz_i, W_ij = g_i p_ij and
//     d z_id      = - sum_j W_ij (z_id - mu_jd) a_jd                        a_jd = 1 / var_jd
//     d mu_jd     = + sum_i W_ij (z_id - mu_jd) a_jd
//     d logvar_jd = 0.5 sum_i W_ij ((z_id - mu_jd)^2 a_jd - 1).
// Expanded in the centred coordinates z' = z - c, mu' = mu - c (c = mean of the posterior means, as in the forward
// pass) every one of them is a product of W (or its transpose) with a thin feature matrix:
//     sweep A (kept axis = samples):     S1 = W [a],  S2 = W [mu' a]            d z  = g (S2 - z' S1)
//     sweep B (kept axis = posteriors):  T2 = W^T [z'^2],  T1 = W^T [z'],  T0 = W^T 1
//                                        d mu = a (T1 - mu' T0),   d logvar = 0.5 (a (T2 - 2 mu' T1 + mu'^2 T0) - T0).
// Both sweeps are the same kernel.  A wave owns 32 kept elements (their operand rows of the density contraction sit
// in VGPRs as the B operand of v_mfma_f32_32x32x2_f32) and streams 32-row blocks of the other axis through LDS:
//   1. DP chained MFMAs from the row bias give the 32x32 tile of base-2 pair log-densities (the forward's
//      contraction: [z', z'^2] . [2 w^2 mu', -w^2] + bias, w^2 = 0.5 log2e / var),
//   2. 16 v_exp_f32 per lane turn it into weights exp2(v - L) (times the row's g in sweep B),
//   3. the accumulator layout of that tile (lane = column, registers = rows) IS the B operand of the next
//      contraction over the 32 rows -- out[f][c] += sum_r F[r][f] W[r][c] -- so the weights never leave the
//      registers: 16 more MFMAs per 32 feature columns, accumulated over the whole sweep.
// Numerics.  A gradient needs the weights to ~1e-4 RELATIVE, i.e. the exponent to ~1e-4 absolute -- tighter than the
// forward, whose log-densities are O(100).  The expansion's cancelling terms are therefore accumulated in groups of
// eight dimensions, each chain starting from that group's share of -sum_d w^2 mu'^2 (so a chain's partial sums stay
// at the size of eight dimensions' terms instead of D), the groups are added on the vector ALU, and the two large
// scalars (the posterior's normaliser and -L_i) join last.  What remains scales with the largest group, and a
// device-side estimate of it (written by the operand preparation) gates this path against the vector-ALU sweeps of
// pair_grad.hip, which work with exact differences (both are launched, the unselected one exits).
// Bound: fp32 MFMA (157.3 TFLOP/s); 16 * D flop per pair over the two sweeps.
#include "pair_plan.h"

// floats of one 32-row record: density operand | features | tail = group biases [DP/8][32], normaliser [32], g [32]
// (the tail padded to whole 64-float DMA pieces)
// BX (D <= 64): the features are stored as three bf16 limbs each, in the fragment order of v_mfma_f32_32x32x16_bf16
// (see the sweep kernel): (DP / 16) feature tiles x 2 k-groups x 3 limbs pieces of 1 KiB.
#define BWD_TAIL_PIECES(DP) (((DP) / 8 + 2 + 1) / 2)
#ifdef BWD_FORCE_FP32
#define BWD_BX(DP) (false && (DP) > 0)
#else
#ifndef BWD_BX_MAX
#define BWD_BX_MAX 64          // widest padded D on the bf16-limb form (D = 128: two ring stages would need 196 KB of LDS)
#endif
#define BWD_BX(DP) ((DP) <= BWD_BX_MAX)
#endif
#define BWD_FEAT_F(DP) (BWD_BX(DP) ? ((DP) / 16) * 6 * 256 : 64 * (DP))
// ... and so is the density operand: (DP / 8) steps of eight dimensions x 3 limbs pieces of 1 KiB for a streamed block,
// [row][k-half][step][limb][8] bf16 (3 DP floats' worth per row) for a kept element
#define BWD_DENS_F(DP) (BWD_BX(DP) ? ((DP) / 8) * 3 * 256 : 64 * (DP))
#define BWD_COL_F(DP) (BWD_BX(DP) ? 3 * (DP) : 2 * (DP))
#define BWD_TAIL_OFF(DP) (BWD_DENS_F(DP) + BWD_FEAT_F(DP))
#define BWD_REC_F(DP) (BWD_TAIL_OFF(DP) + 64 * BWD_TAIL_PIECES(DP))

typedef __bf16 bwd_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bwd_bf16x2 __attribute__((ext_vector_type(2)));

// ---- operand preparation ---------------------------------------------------------------------------------------
// One workgroup per 32-row block; blocks below nblk_j hold posteriors, the others samples.  Every block is written
// in both roles: as a streamed record (fragment-major images of the two A operands, exactly as the lanes read them
// from LDS, + the 32 row biases + the 32 row gradients) and as kept-axis rows colop [row][2][DP] + colbias.
//   posterior j: density operand (2 w^2 mu', -w^2), group biases -sum_{d in group} w^2 mu'^2, scalar sum_d b_d (the
//                normaliser), features [a | mu' a]
//   sample i:    density operand (z', z'^2), group biases 0, scalar -L_i log2e, features [z'^2 | z'], gradient g_i
__global__ void __launch_bounds__(256)
bwd_prep_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, int64_t N,
                const float* __restrict__ z, const float* __restrict__ log_qz, float log_norm,
                const float* __restrict__ g, int64_t M, int D, int DP, PairHeader* hdr, int64_t nblk_j,
                float* __restrict__ rec_j, float* __restrict__ colop_j, float* __restrict__ colbias_j,
                float* __restrict__ rec_i, float* __restrict__ colop_i, float* __restrict__ colbias_i,
                float* __restrict__ colgb_j, float cond_floor) {
    __shared__ float p0[32 * 129], p1[32 * 129];
    __shared__ float sm_shift[128], sm_gb[16 * 32], sm_bias[32], sm_g[32];
    __shared__ float sm_cond[4];
    const int tid = threadIdx.x;
    const bool post = (int64_t)blockIdx.x < nblk_j;
    const int64_t blk = post ? blockIdx.x : (int64_t)blockIdx.x - nblk_j;
    const int64_t n = post ? N : M;
    const float* __restrict__ x = post ? mu : z;
    const int LD = DP + 1;
    const int G = DP / 8;
    if (tid < DP) sm_shift[tid] = tid < D ? (float)(hdr->colsum[tid] / (double)N) : 0.f;
    __syncthreads();
    for (int idx = tid; idx < 32 * DP; idx += 256) {
        const int row = idx / DP, d = idx - row * DP;
        const int64_t gr = blk * 32 + row;
        const bool valid = gr < n && d < D;
        p0[row * LD + d] = valid ? x[gr * D + d] - sm_shift[d] : 0.f;
        p1[row * LD + d] = (valid && post) ? expf(-logvar[gr * D + d]) : 0.f;
    }
    __syncthreads();
    {   // row scalars: 8 consecutive lanes per row, lane l8 owns dimension 8 gi + l8 of every group gi
        const int row = tid >> 3, l8 = tid & 7;
        const int64_t gr = blk * 32 + row;
        float nrm = 0.f, cond = 0.f;
        for (int gi = 0; gi < G; ++gi) {
            const int d = 8 * gi + l8;
            float gb = 0.f, ct = 0.f;
            if (post && gr < n && d < D) {
                const float m = p0[row * LD + d], a = p1[row * LD + d];
                gb = -0.5f * LVAE_LOG2E * a * m * m;
                nrm += -0.5f * LVAE_LOG2E * (logvar[gr * D + d] + LVAE_LOG_2PI);
                const float t = 2.0f * fabsf(m) * sqrtf(a) + 3.0f;      // the forward's per-dimension estimate (lvae_api.hip)
                ct = 0.5f * t * t;
            }
#pragma unroll
            for (int o = 4; o >= 1; o >>= 1) {
                gb += __shfl_xor(gb, o, 64);
                ct += __shfl_xor(ct, o, 64);
            }
            if (l8 == 0) sm_gb[gi * 32 + row] = gb;
            cond = fmaxf(cond, ct);
        }
#pragma unroll
        for (int o = 4; o >= 1; o >>= 1) nrm += __shfl_xor(nrm, o, 64);
        if (l8 == 0) {
            float b = nrm, gv = 0.f;
            if (!post && gr < n) {
                b = -(log_qz[gr] + log_norm) * LVAE_LOG2E;
                gv = g[gr];
            }
            if (gr >= n) b = -INFINITY;
            sm_bias[row] = b;
            sm_g[row] = gv;
        }
        cond = wave_max(cond);
        if ((tid & 63) == 0) sm_cond[tid >> 6] = cond;
    }
    __syncthreads();
    if (post && tid == 0) {      // error scale of the grouped accumulation: the largest group, sqrt(groups) of them
        const float c = fmaxf(fmaxf(sm_cond[0], sm_cond[1]), fmaxf(sm_cond[2], sm_cond[3])) * sqrtf((float)G);
        // (the estimate is only ever compared with the gate's limit: workgroups at or below it publish nothing --
        //  one device-scope atomic per workgroup on a single word was the larger half of this launch)
        if (c > cond_floor) atomicMax(&hdr->cond_bits, __float_as_uint(c));
    }
    float* __restrict__ rec = (post ? rec_j : rec_i) + blk * (int64_t)BWD_REC_F(DP);
    float* __restrict__ cop = (post ? colop_j : colop_i) + blk * 32 * (int64_t)BWD_COL_F(DP);
    float* __restrict__ cb = (post ? colbias_j : colbias_i) + blk * 32;
    // density operand, term h of dimension s
    auto dens = [&](int row, int h, int s) -> float {
        const float v = p0[row * LD + s];
        if (post) {
            const float a = p1[row * LD + s];
            return h == 0 ? LVAE_LOG2E * a * v : -0.5f * LVAE_LOG2E * a;
        }
        return h == 0 ? v : v * v;
    };
    auto limbs3 = [](float o, unsigned short& h16, unsigned short& m16, unsigned short& l16) {
        const __bf16 hb = (__bf16)o;
        const float r1 = o - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const __bf16 lb = (__bf16)(r1 - (float)mb);
        h16 = __builtin_bit_cast(unsigned short, hb);
        m16 = __builtin_bit_cast(unsigned short, mb);
        l16 = __builtin_bit_cast(unsigned short, lb);
    };
    if (BWD_BX(DP)) {
        // A operand of the density contraction on the bf16 pipe: piece (step * 3 + limb) holds, for lane row + 32 h, the
        // eight dimensions 8 step .. 8 step + 7 of term h (one MFMA step = one group of eight dimensions)
        unsigned short* dr = reinterpret_cast<unsigned short*>(rec);
        for (int idx = tid; idx < (DP / 8) * 512; idx += 256) {
            const int i = idx & 7, lane = (idx >> 3) & 63, step = idx >> 9;
            unsigned short a, b, c;
            limbs3(dens(lane & 31, lane >> 5, 8 * step + i), a, b, c);
            const int pb = (step * 3) * 512 + lane * 8 + i;
            dr[pb] = a;
            dr[pb + 512] = b;
            dr[pb + 1024] = c;
        }
    } else {
    for (int idx = tid; idx < 64 * DP; idx += 256) {       // A operand of the density contraction: [s/4][h][row][s%4] (16 B per lane, lane = row + 32 h)
        const int e = idx & 3, row = (idx >> 2) & 31, h = (idx >> 7) & 1, sg = idx >> 8;     // slot = lane = row + 32 h
        rec[idx] = dens(row, h, 4 * sg + e);
    }
    }
    auto feat = [&](int row, int f) -> float {
        const int half = f >= DP, d = f - half * DP;
        const float v = p0[row * LD + d];
        if (post) return half ? v * p1[row * LD + d] : p1[row * LD + d];
        return half ? v : v * v;
    };
    if (BWD_BX(DP)) {
        // A operand of the feature contraction on the bf16 pipe: piece ((t * 2 + s) * 3 + limb) holds, for lane
        // (f % 32) + 32 h, the eight k-entries i of k-group s -- streamed row 16 s + 8 (i / 4) + 4 h + (i % 4), the row
        // that register 8 s + i of a 32x32 fp32 accumulator holds on half-wave h
        unsigned short* fr = reinterpret_cast<unsigned short*>(rec + BWD_DENS_F(DP));
        for (int idx = tid; idx < (DP / 16) * 2 * 512; idx += 256) {
            const int i = idx & 7, lane = (idx >> 3) & 63, s2 = (idx >> 9) & 1, t = idx >> 10;
            const int fl = lane & 31, h = lane >> 5;
            const int row = 16 * s2 + 8 * (i >> 2) + 4 * h + (i & 3);
            const int pbase = ((t * 2 + s2) * 3) * 512 + lane * 8 + i;
            limbs3(feat(row, 32 * t + fl), fr[pbase], fr[pbase + 512], fr[pbase + 1024]);
        }
    } else {
        for (int idx = tid; idx < 64 * DP; idx += 256) {       // A operand of the feature contraction: [f/32][q][h][f%32][e]
            const int e = idx & 3, fl = (idx >> 2) & 31, h = (idx >> 7) & 1, q = (idx >> 8) & 3, ft = idx >> 10;
            rec[BWD_DENS_F(DP) + idx] = feat(8 * q + 4 * h + e, ft * 32 + fl);
        }
    }
    const int toff = BWD_TAIL_OFF(DP);
    for (int idx = tid; idx < 32 * G; idx += 256) rec[toff + idx] = sm_gb[idx];
    if (tid < 32) rec[toff + 32 * G + tid] = sm_bias[tid];
    else if (tid < 64) rec[toff + 32 * G + tid] = sm_g[tid - 32];
    if (BWD_BX(DP)) {                                       // kept-axis rows [row][h][step][limb][8] bf16
        unsigned short* cr = reinterpret_cast<unsigned short*>(cop);
        const int NSTEP = DP / 8;
        for (int idx = tid; idx < 64 * DP; idx += 256) {
            const int i = idx & 7, step = (idx >> 3) % NSTEP, h = ((idx >> 3) / NSTEP) & 1, row = (idx >> 3) / (2 * NSTEP);
            const int pb = (((row * 2 + h) * NSTEP + step) * 3) * 8 + i;
            limbs3(dens(row, h, 8 * step + i), cr[pb], cr[pb + 8], cr[pb + 16]);
        }
    } else {
    for (int idx = tid; idx < 64 * DP; idx += 256) {       // kept-axis rows [row][h][s]
        const int s = idx % DP, h = (idx / DP) & 1, row = idx / (2 * DP);
        cop[idx] = dens(row, h, s);
    }
    }
    if (tid < 32) cb[tid] = sm_bias[tid];
    if (post)       // as a kept element the posterior brings its group biases itself: [row][G]
        for (int idx = tid; idx < 32 * G; idx += 256) colgb_j[blk * 32 * G + idx] = sm_gb[(idx % G) * 32 + idx / G];
}

// ---- one 32x32 block of a sweep in the bf16-limb form, explicitly pipelined --------------------------------------------
// Both contractions of a tile are chains of v_mfma_f32_32x32x16_bf16 whose operands come from LDS; left to the
// compiler every chain starts with its LDS reads and a full wait for them, and the exponentials and the limb split sit
// between the two contractions as one block of vector work: a wave then keeps the matrix pipe busy 40 % of its time
// and three waves per SIMD reach 56-60 % (PMC).  Here the fragments (and, in sweep A, the row biases) of group g + 1
// are read while the MFMAs of group g run, the sum over groups is folded one group late in the shadow of the next
// chain, and the second half of the weights (exponentials + limb split of k-group 1) is formed behind the feature MFMAs
// of k-group 0.  sched_barrier keeps the phases where they are put.
template <int DP, bool ROWG>
__device__ __forceinline__ void bwd_block_bx(const float* __restrict__ st, const float* __restrict__ tail, int lane, int h,
                                             const bwd_bf16x8 (&bzx)[DP / 8][3], const float (&cg)[DP / 8], float cb,
                                             f32x16 (&oacc)[DP / 16], float& wsum) {
    constexpr int G = DP / 8, NFT = DP / 16;
    constexpr int LA[6] = {1, 2, 0, 1, 0, 0};
    constexpr int LB[6] = {1, 0, 2, 0, 1, 0};
    const uint4* dr = reinterpret_cast<const uint4*>(st) + lane;
    const uint4* fr = reinterpret_cast<const uint4*>(st + BWD_DENS_F(DP)) + lane;
    auto load_d = [&](int g, bwd_bf16x8 (&d)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < 3; ++l) d[l] = __builtin_bit_cast(bwd_bf16x8, dr[(g * 3 + l) * 64]);
    };
    auto init_acc = [&](int g, f32x16& a) __attribute__((always_inline)) {
        if constexpr (ROWG) {
#pragma unroll
            for (int v = 0; v < 16; ++v) a[v] = cg[g];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(tail + g * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[4 * q + e] = b4[e];
            }
        }
    };
    auto load_f = [&](int idx, bwd_bf16x8 (&f)[3]) __attribute__((always_inline)) {      // idx = s * NFT + t
        const int s2 = idx / NFT, t = idx % NFT;
#pragma unroll
        for (int l = 0; l < 3; ++l) f[l] = __builtin_bit_cast(bwd_bf16x8, fr[((t * 2 + s2) * 3 + l) * 64]);
    };
    // ---- density: groups of eight dimensions, fragments and biases one group ahead ----
    // (an explicit lgkmcnt(0) BEFORE each prefetch: the reads of the previous one have had a whole chain's issue time
    //  to land, and the compiler then finds nothing pending when it reaches their first use -- left to itself it puts
    //  its wait after the new reads and the wave sits out a full LDS round trip per group.  0xC07F = lgkmcnt(0) only.)
    constexpr int NACC = ROWG ? 2 : 3;        // sweep A reads the next group's biases into a third accumulator image
    bwd_bf16x8 da[2][3];
    bwd_bf16x8 fa[2][3];
    f32x16 acc[NACC];
    f32x16 tot;
    load_d(0, da[0]);
    init_acc(0, acc[0]);
    f32x4 n4[2];                              // row scalars of the exponent (and the row gradients in sweep B) of a k-group
    f32x4 g4[ROWG ? 2 : 1];
    auto load_rows = [&](int s2) __attribute__((always_inline)) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            n4[qq] = *reinterpret_cast<const f32x4*>(tail + G * 32 + 8 * (2 * s2 + qq) + 4 * h);
            if constexpr (ROWG) g4[qq] = *reinterpret_cast<const f32x4*>(tail + (G + 1) * 32 + 8 * (2 * s2 + qq) + 4 * h);
        }
    };
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int cur = g % NACC, nxt = (g + 1) % NACC, prv = (g + NACC - 1) % NACC;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (g + 1 < G) {
            load_d(g + 1, da[(g + 1) & 1]);
            if constexpr (!ROWG) init_acc(g + 1, acc[nxt]);
        } else {
            load_rows(0);
            load_f(0, fa[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (same (posterior limb, sample limb) order in both sweeps: see the comment in the kernel)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
            acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[g & 1][ROWG ? LB[pr] : LA[pr]], bzx[g][ROWG ? LA[pr] : LB[pr]],
                                                               acc[cur], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g >= 1) {                          // fold the previous group (its chain has had this one's issue time to finish)
            if (g == 1) tot = acc[prv];
            else tot += acc[prv];
        }
        if constexpr (ROWG)
            if (g + 1 < G) init_acc(g + 1, acc[nxt]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (G == 1) tot = acc[0];
    else tot += acc[(G - 1) % NACC];
    // ---- weights of k-group 0, its limbs, then the feature MFMAs with the second half formed behind them ----
    float w[16];
    bwd_bf16x8 wl[2][3];
    auto weights = [&](int s2) __attribute__((always_inline)) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int q = 2 * s2 + qq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#ifdef BWD_DBG_NOEXP              // (ablation build: results are garbage)
                float wv = tot[4 * q + e] + (n4[qq][e] + cb);
#else
                float wv = lvae_exp2(tot[4 * q + e] + (n4[qq][e] + cb));
#endif
                if constexpr (ROWG) wv *= g4[qq][e];
                w[4 * q + e] = wv;
                wsum += wv;
            }
        }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const f32x2 v = {w[8 * s2 + 2 * pr], w[8 * s2 + 2 * pr + 1]};
            const bwd_bf16x2 hh = __builtin_convertvector(v, bwd_bf16x2);
#ifdef BWD_DBG_NOSPLIT            // (ablation build: results are garbage)
            const bwd_bf16x2 mm = hh, ll = hh;
#else
            const f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
            const bwd_bf16x2 mm = __builtin_convertvector(r1, bwd_bf16x2);
            const f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
            const bwd_bf16x2 ll = __builtin_convertvector(r2, bwd_bf16x2);
#endif
            wl[s2][0][2 * pr] = hh[0];
            wl[s2][0][2 * pr + 1] = hh[1];
            wl[s2][1][2 * pr] = mm[0];
            wl[s2][1][2 * pr + 1] = mm[1];
            wl[s2][2][2 * pr] = ll[0];
            wl[s2][2][2 * pr + 1] = ll[1];
        }
    };
    weights(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < 2 * NFT; ++idx) {
        const int cur = idx & 1, s2 = idx / NFT, t = idx % NFT;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (idx + 1 < 2 * NFT) load_f(idx + 1, fa[cur ^ 1]);
        if (idx == 0) load_rows(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][LA[pr]], wl[s2][LB[pr]], oacc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (idx == 0) {                        // the other half of the weights, behind the MFMAs just issued
            weights(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- the sweep ---------------------------------------------------------------------------------------------------
// partial [nsplit][cpad][2 DP]: per kept element the 2 DP feature sums; wsum [nsplit][cpad]: sum of its weights.  ROWG: weights carry the streamed row's gradient (sweep B); otherwise the kept element's own gradient is
// applied by the merge kernel.
template <int DP, bool ROWG, int WPS>
__global__ void __launch_bounds__(256, WPS)
pair_bwd_mfma_kernel(const float* __restrict__ rec, const float* __restrict__ colop, const float* __restrict__ colbias,
                     const float* __restrict__ colgb, int64_t nblk, int64_t blk_per_split, int64_t cpad, float* __restrict__ partial,
                     float* __restrict__ wsum_out, const PairHeader* __restrict__ hdr, float cond_limit) {
    if (cond_limit >= 0.f && __uint_as_float(hdr->cond_bits) > cond_limit) return;
    constexpr bool BX = BWD_BX(DP);
    constexpr int REC = BWD_REC_F(DP);
    constexpr int NFT = DP / 16;                  // 32-wide tiles of the 2 DP feature columns
    constexpr int PIECES = BWD_TAIL_OFF(DP) / 256;    // 1 KiB DMA pieces of a record (without the tail)
    __shared__ __attribute__((aligned(16))) float ring[2 * REC];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t col = ((int64_t)blockIdx.x * 4 + wave) * 32 + r;

    constexpr int NS = DP / 8;                   // BX: MFMA steps (= groups of eight dimensions) per tile
    float bc[BX ? 1 : DP];
    bwd_bf16x8 bzx[BX ? NS : 1][3];              // BX: [step][limb]
    if constexpr (BX) {
        const uint4* src = reinterpret_cast<const uint4*>(colop) + (col * 2 + h) * NS * 3;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx)
#pragma unroll
            for (int l = 0; l < 3; ++l) bzx[sx][l] = __builtin_bit_cast(bwd_bf16x8, src[sx * 3 + l]);
    } else {
        const f32x4* src = reinterpret_cast<const f32x4*>(colop + (col * 2 + h) * DP);
#pragma unroll
        for (int q = 0; q < DP / 4; ++q) {
            const f32x4 v = src[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) bc[4 * q + e] = v[e];
        }
    }
    constexpr int LA[6] = {1, 2, 0, 1, 0, 0};     // limb of the streamed operand (A), smallest products first
    constexpr int LB[6] = {1, 0, 2, 0, 1, 0};     // limb of the resident operand (B)
    const float cb = colbias[col];
    // ROWG: the kept posterior's group biases (the streamed samples have none) enter a chain as one more MFMA step,
    // ones x (bias in k-half 0, zero in k-half 1): no per-group accumulator images in registers, no vector adds
    // (BX: the same bias as a broadcast into the sixteen accumulator registers of every lane -- bit-identical to the
    //  ones x bias MFMA step and off the matrix pipe)
    float cg[DP / 8];
#pragma unroll
    for (int gi = 0; gi < DP / 8; ++gi) cg[gi] = (ROWG && (BX || h == 0)) ? colgb[col * (DP / 8) + gi] : 0.f;
    f32x16 oacc[NFT];
#pragma unroll
    for (int t = 0; t < NFT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) oacc[t][v] = 0.f;
    float wsum = 0.f;

    const int64_t b0 = (int64_t)blockIdx.y * blk_per_split;
    int64_t b1 = b0 + blk_per_split;
    if (b1 > nblk) b1 = nblk;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
    auto dma_block = [&](int64_t blk, int stage) __attribute__((always_inline)) {
        const char* gsrc = reinterpret_cast<const char*>(rec + blk * (int64_t)REC);
        float* l = ring + stage * REC;
#pragma unroll
        for (int q0 = 0; q0 < PIECES; q0 += 4) {
            const int q = q0 + wave;
            if (PIECES % 4 == 0 || q < PIECES)
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + q * 1024 + lane16), (lptr_t)(l + q * 256), 16, 0, 0);
        }
#pragma unroll
        for (int t0 = 0; t0 < BWD_TAIL_PIECES(DP); t0 += 4) {      // 64 lanes x 4 B: two 32-row vectors of the tail
            const int t = t0 + wave;
            if (t < BWD_TAIL_PIECES(DP))
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + PIECES * 1024 + t * 256 + lane4),
                                                 (lptr_t)(l + PIECES * 256 + t * 64), 4, 0, 0);
        }
    };

    if (b0 < b1) dma_block(b0, 0);
    for (int64_t blk = b0; blk < b1; ++blk) {
        const int stage = (int)((blk - b0) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef BWD_DBG_NOBAR             // (ablation build: results are garbage)
        __syncthreads();          // block `blk` is complete for everybody, and everybody is done with the other stage
#endif
        if (blk + 1 < b1) dma_block(blk + 1, stage ^ 1);
        const float* st = ring + stage * REC;
        constexpr int G = DP / 8;
        const float* tail = st + BWD_TAIL_OFF(DP);
        if constexpr (BX) {
            bwd_block_bx<DP, ROWG>(st, tail, lane, h, bzx, cg, cb, oacc, wsum);
            continue;
        }
        f32x16 tot;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            f32x16 acc;
            if constexpr (ROWG) {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, cg[gi], acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(tail + gi * 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * q + e] = b4[e];
                }
            }
            if constexpr (BX) {
                // one step of v_mfma_f32_32x32x16_bf16 = this group's eight dimensions, both terms; six limb products
                const uint4* dr = reinterpret_cast<const uint4*>(st) + lane;
                bwd_bf16x8 da[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) da[l] = __builtin_bit_cast(bwd_bf16x8, dr[(gi * 3 + l) * 64]);
                // The limb products are accumulated in the same (posterior limb, sample limb) order in both sweeps -- the
                // streamed and the resident operand swap roles between them --, so that v_ij comes out bit-identical in
                // A and in B: sweep B's weights are normalised by the sums of sweep A, and the rounding of a far sample's
                // |v| ~ 1e4 only cancels if both sweeps made the same one (fuzz case: 4e-4 instead of 9e-7 otherwise).
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ROWG ? LB[pr] : LA[pr]], bzx[gi][ROWG ? LA[pr] : LB[pr]],
                                                                  acc, 0, 0, 0);
            } else {
#pragma unroll
            for (int sg = 2 * gi; sg < 2 * gi + 2; ++sg) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(st + (sg * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], bc[4 * sg + e], acc, 0, 0, 0);
            }
            }
            if (gi == 0) tot = acc;
            else tot += acc;
        }
        float w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 n4 = *reinterpret_cast<const f32x4*>(tail + G * 32 + 8 * q + 4 * h);
#pragma unroll
#ifdef BWD_DBG_NOEXP              // (ablation build: results are garbage)
            for (int e = 0; e < 4; ++e) w[4 * q + e] = tot[4 * q + e] + (n4[e] + cb);
#else
            for (int e = 0; e < 4; ++e) w[4 * q + e] = lvae_exp2(tot[4 * q + e] + (n4[e] + cb));
#endif
        }
        if constexpr (ROWG) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(tail + (G + 1) * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[4 * q + e] *= g4[e];
            }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) wsum += w[v];
        if constexpr (BX) {
            // The feature contraction on the bf16 matrix pipe at fp32-equivalent accuracy: the tile of weights --
            // lane = column, registers = rows -- is split into three bf16 limbs (w = h + m + l, 8 + 8 + 8 mantissa bits)
            // and is, register for register, the B operand of v_mfma_f32_32x32x16_bf16 over the 16 rows of k-group s
            // (registers 8 s .. 8 s + 7); the features come from the record as limb fragments in the same row order.
            // Six limb products of weight >= 2^-16 per k-group (the dropped three are below 2^-24 of the product):
            // 12 MFMAs of 32 cycles per 32 feature columns instead of 16 of 64, off the fp32 lanes.
            bwd_bf16x8 wl[2][3];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const f32x2 v = {w[8 * s2 + 2 * pr], w[8 * s2 + 2 * pr + 1]};
                    const bwd_bf16x2 hh = __builtin_convertvector(v, bwd_bf16x2);
#ifdef BWD_DBG_NOSPLIT            // (ablation build: results are garbage)
                    const bwd_bf16x2 mm = hh, ll = hh;
#else
                    const f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
                    const bwd_bf16x2 mm = __builtin_convertvector(r1, bwd_bf16x2);
                    const f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
                    const bwd_bf16x2 ll = __builtin_convertvector(r2, bwd_bf16x2);
#endif
                    wl[s2][0][2 * pr] = hh[0];
                    wl[s2][0][2 * pr + 1] = hh[1];
                    wl[s2][1][2 * pr] = mm[0];
                    wl[s2][1][2 * pr + 1] = mm[1];
                    wl[s2][2][2 * pr] = ll[0];
                    wl[s2][2][2 * pr + 1] = ll[1];
                }
            const uint4* fr = reinterpret_cast<const uint4*>(st + BWD_DENS_F(DP)) + lane;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int t = 0; t < NFT; ++t) {
                    bwd_bf16x8 fa[3];
#pragma unroll
                    for (int l = 0; l < 3; ++l) fa[l] = __builtin_bit_cast(bwd_bf16x8, fr[((t * 2 + s2) * 3 + l) * 64]);
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
                        oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[LA[pr]], wl[s2][LB[pr]], oacc[t], 0, 0, 0);
                }
        } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 f4[NFT];
#pragma unroll
            for (int t = 0; t < NFT; ++t)
                f4[t] = *reinterpret_cast<const f32x4*>(st + BWD_DENS_F(DP) + ((t * 4 + q) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NFT; ++t)
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4[t][e], w[4 * q + e], oacc[t], 0, 0, 0);
        }
        }
    }
    // out[f][c]: lane (c, h) holds f = 32 t + 8 q + 4 h + e in register 4 q + e of tile t
    float* __restrict__ o = partial + ((int64_t)blockIdx.y * cpad + col) * (2 * DP);
#pragma unroll
    for (int t = 0; t < NFT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = oacc[t][4 * q + e];
            *reinterpret_cast<f32x4*>(o + 32 * t + 8 * q + 4 * h) = v;
        }
    wsum += __shfl_xor(wsum, 32, 64);
    if (h == 0) wsum_out[(int64_t)blockIdx.y * cpad + col] = wsum;
}

// Between the sweeps: the weights of sample i are renormalised by their own sum s_i = sum_j exp2(v_ij - L_i) from
// sweep A (L_i comes from the forward pass, whose absolute error of a few 1e-4 at D ~ 100 would otherwise be the
// relative error of every weight): g_i / s_i replaces g_i in the streamed sample records and in the d z merge.
__global__ void __launch_bounds__(256)
bwd_rescale_g_kernel(const float* __restrict__ wsum, int nsplit, int64_t cpad, const float* __restrict__ g, int64_t M,
                     int DP, float* __restrict__ rec_i, float* __restrict__ gscaled, const PairHeader* __restrict__ hdr,
                     float cond_limit) {
    if (cond_limit >= 0.f && __uint_as_float(hdr->cond_bits) > cond_limit) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float sum = 0.f;
    for (int k = 0; k < nsplit; ++k) sum += wsum[(int64_t)k * cpad + i];
    const float gs = sum > 0.f ? g[i] / sum : 0.f;
    gscaled[i] = gs;
    rec_i[(i >> 5) * (int64_t)BWD_REC_F(DP) + BWD_TAIL_OFF(DP) + 32 * (DP / 8 + 1) + (i & 31)] = gs;
}

// ---- merges ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bwd_merge_dz_kernel(const float* __restrict__ partial, int nsplit, int64_t cpad, int DP, const float* __restrict__ z,
                    const float* __restrict__ g, int64_t M, int64_t N, int D, const PairHeader* __restrict__ hdr,
                    float cond_limit, float poison_limit, float* __restrict__ d_z) {
    if (cond_limit >= 0.f && __uint_as_float(hdr->cond_bits) > cond_limit) return;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * D) return;
    // no other branch behind this call and the estimate is past the limit: the result would be below the parity bar
    // without anybody noticing -- make it NaN instead (lvae.h, lvae_log_qz_bwd_f32)
    if (poison_limit >= 0.f && __uint_as_float(hdr->cond_bits) > poison_limit) {
        d_z[idx] = __builtin_nanf("");
        return;
    }
    const int64_t i = idx / D;
    const int d = (int)(idx - i * D);
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nsplit; ++k) {
        const float* p = partial + ((int64_t)k * cpad + i) * (2 * DP);
        s1 += p[d];
        s2 += p[DP + d];
    }
    const float zc = z[idx] - (float)(hdr->colsum[d] / (double)N);
    d_z[idx] = g[i] * (s2 - zc * s1);
}

__global__ void __launch_bounds__(256)
bwd_merge_dp_kernel(const float* __restrict__ partial, const float* __restrict__ wsum, int nsplit, int64_t cpad, int DP,
                    const float* __restrict__ mu, const float* __restrict__ logvar, int64_t N, int D,
                    const PairHeader* __restrict__ hdr, float cond_limit, float poison_limit, float* __restrict__ d_mu,
                    float* __restrict__ d_logvar) {
    if (cond_limit >= 0.f && __uint_as_float(hdr->cond_bits) > cond_limit) return;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * D) return;
    if (poison_limit >= 0.f && __uint_as_float(hdr->cond_bits) > poison_limit) {
        if (d_mu) d_mu[idx] = __builtin_nanf("");
        if (d_logvar) d_logvar[idx] = __builtin_nanf("");
        return;
    }
    const int64_t j = idx / D;
    const int d = (int)(idx - j * D);
    float t2 = 0.f, t1 = 0.f, t0 = 0.f;
    for (int k = 0; k < nsplit; ++k) {
        const float* p = partial + ((int64_t)k * cpad + j) * (2 * DP);
        t2 += p[d];
        t1 += p[DP + d];
        t0 += wsum[(int64_t)k * cpad + j];
    }
    const float m = mu[idx] - (float)(hdr->colsum[d] / (double)N);
    const float a = expf(-logvar[idx]);
    if (d_mu) d_mu[idx] = a * (t1 - m * t0);
    if (d_logvar) d_logvar[idx] = 0.5f * (a * (t2 - 2.0f * m * t1 + m * m * t0) - t0);
}

// ---- host side ---------------------------------------------------------------------------------------------------
BwdMfmaLayout lvae_bwd_mfma_layout(int64_t M, int64_t N, int D) {
    BwdMfmaLayout L{};
    L.DP = lvae_pad_dim(D);
    if (L.DP == 0) return L;
    L.cpad_i = lvae_round_up(M, 128);
    L.cpad_j = lvae_round_up(N, 128);
    // workgroups resident per CU: by LDS (two ring stages of a record) and registers
    // (D <= 16: 122-166 registers and 25 KB of ring; D <= 32: 192-256 registers, the pipelined bf16-limb body;
    //  D <= 64 in that form: 330-448 registers and 99 KB of ring, one workgroup = one wave per SIMD -- the explicit
    //  pipelining is what keeps the matrix pipe fed there: 16,384^2 2.36 -> 1.75 ms against the fp32 contractions at two)
    const int64_t slots = 256 * (L.DP <= 16 ? 3 : (L.DP <= 32 || (L.DP == 64 && !BWD_BX(64)) ? 2 : 1));
    auto split = [&](int64_t kept_pad, int64_t rows, int* ns, int64_t* bps) {
        const int64_t groups = kept_pad / 128, nblk = lvae_ceil_div(rows, 32);
        int64_t c = (2 * slots + groups / 2) / groups;
        if (c > 32) c = 32;
        if (c > nblk) c = nblk;
        if (c < 1) c = 1;
        *bps = lvae_ceil_div(nblk, c);
        *ns = (int)lvae_ceil_div(nblk, *bps);
    };
    split(L.cpad_i, N, &L.ns_a, &L.bps_a);
    split(L.cpad_j, M, &L.ns_b, &L.bps_b);
    const size_t rec = (size_t)BWD_REC_F(L.DP) * sizeof(float);
    size_t o = 0;
    L.off_recj = o;  o = lvae_align256(o + (size_t)(L.cpad_j / 32) * rec);
    L.off_reci = o;  o = lvae_align256(o + (size_t)(L.cpad_i / 32) * rec);
    L.off_colj = o;  o = lvae_align256(o + (size_t)L.cpad_j * BWD_COL_F(L.DP) * sizeof(float));
    L.off_coli = o;  o = lvae_align256(o + (size_t)L.cpad_i * BWD_COL_F(L.DP) * sizeof(float));
    L.off_cbj = o;   o = lvae_align256(o + (size_t)L.cpad_j * sizeof(float));
    L.off_cbi = o;   o = lvae_align256(o + (size_t)L.cpad_i * sizeof(float));
    L.off_pa = o;    o = lvae_align256(o + (size_t)L.ns_a * L.cpad_i * 2 * L.DP * sizeof(float));
    L.off_pb = o;    o = lvae_align256(o + (size_t)L.ns_b * L.cpad_j * 2 * L.DP * sizeof(float));
    L.off_w0 = o;    o = lvae_align256(o + (size_t)L.ns_b * L.cpad_j * sizeof(float));
    L.off_w0a = o;   o = lvae_align256(o + (size_t)L.ns_a * L.cpad_i * sizeof(float));
    L.off_gs = o;    o = lvae_align256(o + (size_t)L.cpad_i * sizeof(float));
    L.off_cgj = o;   o = lvae_align256(o + (size_t)L.cpad_j * (L.DP / 8) * sizeof(float));
    L.total = o;
    return L;
}

template <int DP, int WPS>
static void launch_sweeps(const BwdMfmaLayout& L, char* base, int64_t M, int64_t N, const float* g, bool want_p,
                          const PairHeader* hdr, float cond_limit, hipStream_t st) {
    // sweep A always runs: it also yields the weight sums the renormalisation needs
    hipLaunchKernelGGL((pair_bwd_mfma_kernel<DP, false, WPS>), dim3((unsigned)(L.cpad_i / 128), (unsigned)L.ns_a),
                       dim3(256), 0, st, (const float*)(base + L.off_recj), (const float*)(base + L.off_coli),
                       (const float*)(base + L.off_cbi), (const float*)nullptr, lvae_ceil_div(N, 32), L.bps_a, L.cpad_i,
                       (float*)(base + L.off_pa), (float*)(base + L.off_w0a), hdr, cond_limit);
    hipLaunchKernelGGL(bwd_rescale_g_kernel, dim3((unsigned)lvae_ceil_div(M, 256)), dim3(256), 0, st,
                       (const float*)(base + L.off_w0a), L.ns_a, L.cpad_i, g, M, DP, (float*)(base + L.off_reci),
                       (float*)(base + L.off_gs), hdr, cond_limit);
    if (want_p)
        hipLaunchKernelGGL((pair_bwd_mfma_kernel<DP, true, WPS>), dim3((unsigned)(L.cpad_j / 128), (unsigned)L.ns_b),
                           dim3(256), 0, st, (const float*)(base + L.off_reci), (const float*)(base + L.off_colj),
                           (const float*)(base + L.off_cbj), (const float*)(base + L.off_cgj), lvae_ceil_div(M, 32),
                           L.bps_b, L.cpad_j, (float*)(base + L.off_pb), (float*)(base + L.off_w0), hdr, cond_limit);
}

// hdr: zeroed, with the column sums of mu already in it (colsum_kernel).  base: the region lvae_bwd_mfma_layout sizes.
// cond_limit < 0: unconditional; otherwise everything behind the operand preparation runs iff the conditioning
// estimate is <= cond_limit (the caller launches the vector-ALU sweeps with the opposite gate).
// poison_limit >= 0 (with cond_limit < 0: no other branch exists for this shape): the gradients come out as NaN when
// the estimate exceeds it, instead of silently below the parity bar.
int lvae_launch_bwd_mfma(const BwdMfmaLayout& L, void* ws, const float* z, const float* mu, const float* logvar,
                         int64_t M, int64_t N, int D, float log_norm, const float* log_qz, const float* g,
                         PairHeader* hdr, float cond_limit, float* d_z, float* d_mu, float* d_logvar, hipStream_t st,
                         float poison_limit) {
    if (L.DP == 0) return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the MFMA backward needs D <= 128 (got D=%d)", D);
    char* base = (char*)ws;
    const int64_t nbj = L.cpad_j / 32, nbi = L.cpad_i / 32;
    hipLaunchKernelGGL(bwd_prep_kernel, dim3((unsigned)(nbj + nbi)), dim3(256), 0, st, mu, logvar, N, z, log_qz, log_norm,
                       g, M, D, L.DP, hdr, nbj, (float*)(base + L.off_recj), (float*)(base + L.off_colj),
                       (float*)(base + L.off_cbj), (float*)(base + L.off_reci), (float*)(base + L.off_coli),
                       (float*)(base + L.off_cbi), (float*)(base + L.off_cgj),
                       cond_limit >= 0.f ? cond_limit : (poison_limit >= 0.f ? poison_limit : INFINITY));
    const bool want_z = d_z != nullptr, want_p = d_mu || d_logvar;
    switch (L.DP) {
        case 16: launch_sweeps<16, 2>(L, base, M, N, g, want_p, hdr, cond_limit, st); break;
        case 32: launch_sweeps<32, 2>(L, base, M, N, g, want_p, hdr, cond_limit, st); break;
        case 64: launch_sweeps<64, BWD_BX(64) ? 1 : 2>(L, base, M, N, g, want_p, hdr, cond_limit, st); break;
        default: launch_sweeps<128, 1>(L, base, M, N, g, want_p, hdr, cond_limit, st); break;
    }
    if (want_z)
        hipLaunchKernelGGL(bwd_merge_dz_kernel, dim3((unsigned)lvae_ceil_div(M * D, 256)), dim3(256), 0, st,
                           (const float*)(base + L.off_pa), L.ns_a, L.cpad_i, L.DP, z, (const float*)(base + L.off_gs), M, N, D, hdr,
                           cond_limit, poison_limit, d_z);
    if (want_p)
        hipLaunchKernelGGL(bwd_merge_dp_kernel, dim3((unsigned)lvae_ceil_div(N * D, 256)), dim3(256), 0, st,
                           (const float*)(base + L.off_pb), (const float*)(base + L.off_w0), L.ns_b, L.cpad_j, L.DP, mu,
                           logvar, N, D, hdr, cond_limit, poison_limit, d_mu, d_logvar);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
