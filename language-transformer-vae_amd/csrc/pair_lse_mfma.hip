// Fused pair log-density + streaming logsumexp on the fp32 matrix cores (gfx950).
//
//   out[c] = log2 sum_r 2^( sum_k Rop[r][k] * Cop[c][k] + rbias[r] )          (base-2 domain)
//
// r runs over the REDUCED axis (posteriors for log q(z)), c over the KEPT axis (samples).  Operands
// come from prep.hip.  Each wave owns 32*NCB kept elements for its whole life: their operand rows sit
// in VGPRs as the B operand of v_mfma_f32_32x32x2_f32 (lane = column + 32*half).  The wave then streams
// 32-row blocks of the reduced axis: one coalesced 1 KiB load per 4 MFMA steps brings the A operand
// (fragment-major layout), the accumulators start from the row bias, DP chained MFMAs per 32x32 tile
// produce the base-2 log-densities, and the epilogue folds the 16 values a lane holds (16 reduced rows
// of its own column) into that lane's running (max, sum).  The reduction axis lives on the MFMA ROW
// index, i.e. in registers, so the logsumexp needs no cross-lane traffic until the very end, where
// the two half-waves (rows +0..3 / +4..7 of every 8) are merged with one wave shuffle.
//
// The running max is updated lazily: a tile only triggers a rescale when its max exceeds the running
// one by more than LSE_THR (base-2), so the common tile costs one v_exp_f32 per pair and nothing else
// on the transcendental pipe; sums stay below 2^LSE_THR * R, far from fp32 overflow.
//
// Bound: fp32 MFMA (157.3 TFLOP/s); algorithmic cost 4*D flop per pair.  HBM traffic is negligible
// (operands are L2/Infinity-Cache resident and re-read by every wave).
#include "pair_plan.h"

#define LSE_THR 8.0f
#define LSE_MINIT (-1.0e30f)

template <int DP, int NCB, int WPS, bool EXCL>
__global__ void __launch_bounds__(256, WPS)
pair_lse_mfma_kernel(const float* __restrict__ rfrag, const float* __restrict__ rbias,
                     const float* __restrict__ cop, float2* __restrict__ partial,
                     const PairHeader* __restrict__ hdr, float cond_limit, int use_if_above,
                     int64_t nblk, int64_t blk_per_split, int64_t cpad, int64_t excl_offset) {
    // Path gate: the conditioning estimate written by prep decides between this kernel and the direct
    // one without a host round trip (both are launched; the unselected one exits here).
    if (cond_limit >= 0.f) {
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    constexpr int NQ = DP / 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * NCB);

    // kept-axis operand: resident for the whole kernel
    float bz[NCB][DP];
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

    float m[NCB], s[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        m[cb] = LSE_MINIT;
        s[cb] = 0.f;
    }

    const int64_t b0 = (int64_t)blockIdx.y * blk_per_split;
    int64_t b1 = b0 + blk_per_split;
    if (b1 > nblk) b1 = nblk;

    const f32x4* afrag = reinterpret_cast<const f32x4*>(rfrag) + lane;
    const f32x4* bfrag = reinterpret_cast<const f32x4*>(rbias) + h;

    f32x4 a_nxt[NQ];
    f32x4 c_nxt[4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) a_nxt[q] = afrag[(b0 * NQ + q) * 64];
#pragma unroll
    for (int g = 0; g < 4; ++g) c_nxt[g] = bfrag[b0 * 8 + 2 * g];

    for (int64_t blk = b0; blk < b1; ++blk) {
        f32x4 a_cur[NQ];
        f32x4 c_cur[4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) a_cur[q] = a_nxt[q];
#pragma unroll
        for (int g = 0; g < 4; ++g) c_cur[g] = c_nxt[g];
        // prefetch the next block (the last iteration re-reads its own block: harmless)
        const int64_t nb = (blk + 1 < b1) ? blk + 1 : blk;
#pragma unroll
        for (int q = 0; q < NQ; ++q) a_nxt[q] = afrag[(nb * NQ + q) * 64];
#pragma unroll
        for (int g = 0; g < 4; ++g) c_nxt[g] = bfrag[nb * 8 + 2 * g];

        f32x16 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[cb][4 * g + e] = c_cur[g][e];

#pragma unroll
        for (int st = 0; st < DP; ++st) {
            const float a = a_cur[st >> 2][st & 3];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bz[cb][st], acc[cb], 0, 0, 0);
        }

        if (EXCL) {
            // drop the pair (reduced row == kept column + offset): only blocks that can contain it
            const int64_t row_lo = blk * 32;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int64_t cbase = col0 + cb * 32 + excl_offset;
                if (row_lo < cbase + 32 && row_lo + 32 > cbase) {
                    const int64_t want = cbase + r - row_lo;   // row index inside the block
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (want == 8 * g + 4 * h + e) acc[cb][4 * g + e] = -INFINITY;
                }
            }
        }

#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float t = acc[cb][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) t = fmaxf(t, acc[cb][e]);
            if (__any(t > m[cb] + LSE_THR)) {
                const float mn = (t > m[cb] + LSE_THR) ? t : m[cb];
                s[cb] *= lvae_exp2(m[cb] - mn);
                m[cb] = mn;
            }
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) add += lvae_exp2(acc[cb][e] - m[cb]);
            s[cb] += add;
        }
    }

    // merge the two half-waves (same column, complementary rows) and publish
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const float mo = __shfl_xor(m[cb], 32, 64);
        const float so = __shfl_xor(s[cb], 32, 64);
        const float mm = fmaxf(m[cb], mo);
        const float ss = s[cb] * lvae_exp2(m[cb] - mm) + so * lvae_exp2(mo - mm);
        if (h == 0) partial[(int64_t)blockIdx.y * cpad + col0 + cb * 32 + r] = make_float2(mm, ss);
    }
}

template <int DP, int NCB, int WPS>
static int launch_t(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                    float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                    bool excl, int64_t excl_offset, hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32 * NCB)), (unsigned)p.nsplit);
    if (excl)
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, NCB, WPS, true>), grid, dim3(256), 0, st, rfrag,
                           rbias, cop, partial, hdr, cond_limit, use_if_above, p.nblk,
                           p.blk_per_split, p.cpad, excl_offset);
    else
        hipLaunchKernelGGL((pair_lse_mfma_kernel<DP, NCB, WPS, false>), grid, dim3(256), 0, st, rfrag,
                           rbias, cop, partial, hdr, cond_limit, use_if_above, p.nblk,
                           p.blk_per_split, p.cpad, excl_offset);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// cond_limit < 0: run unconditionally.  Otherwise run iff (cond > cond_limit) == use_if_above.
int lvae_launch_pair_lse_mfma(const PairPlan& p, const float* rfrag, const float* rbias,
                              const float* cop, float2* partial, const PairHeader* hdr,
                              float cond_limit, int use_if_above, bool excl, int64_t excl_offset,
                              hipStream_t st) {
    if (p.cpad % (4 * 32 * p.NCB) != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "pair plan: cpad %lld not a multiple of the tile",
                              (long long)p.cpad);
    switch (p.DP) {
        case 16: return launch_t<16, 4, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, st);
        case 32: return launch_t<32, 2, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, st);
        case 64: return launch_t<64, 1, 2>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, st);
        case 128: return launch_t<128, 1, 1>(p, rfrag, rbias, cop, partial, hdr, cond_limit, use_if_above, excl, excl_offset, st);
        default:
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "MFMA path needs D <= 128 (got D=%d)", p.D);
    }
}
