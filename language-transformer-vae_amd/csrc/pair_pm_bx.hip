// The product-of-marginals sweep with its polynomial on the bf16 matrix pipe (D <= 128): the fast pass of
// LVAE_PATH_AUTO for well-conditioned inputs (device-side gate, like the joint density's) and of the opt-in
// LVAE_PATH_MFMA_BF16X3 (there unconditionally).
//
// For one dimension d the base-2 log-density of sample z_i under posterior j is a rank-3 product
//     t_ijd = A_jd z_id^2 + B_jd z_id + C_jd,   A = -w2, B = 2 mu w2, C = b - mu^2 w2,
//     w2 = 0.5 log2e / var, b = -0.5 log2e (logvar + log 2 pi)          (mu, z centred like the joint density)
// and with every value split into three bf16 limbs (h + m + l) its fifteen limb products of weight >= 2^-16
// (six for A z^2, six for B z, three for C * 1) fill the K = 16 of ONE v_mfma_f32_32x32x16_bf16 per 32 x 32
// tile of (posterior, sample) pairs and dimension: 32 matrix-pipe cycles that overlap with the vector unit,
// which is left with the exponential and half a packed add per term (pair_pm_kernel: 2.5 more VALU slots).
//   A operand (reduced axis, per 32-row block and dimension): k = [Ah Ah Am Ah Al Am Bh Bh | Bm Bh Bl Bm Ch Cm Cl 0]
//   B operand (kept axis, per sample and dimension)         : k = [Zh Zm Zh Zl Zh Zm zh zm | zh zl zh zm 1  1  1  0]
// (Z = z^2; lanes 0..31 hold k 0..7, lanes 32..63 k 8..15).  Same contract as pair_pm_kernel: fixed reference 0,
// partials (0, sum) per (split, dimension, sample), one redo flag per (split, 64 samples); pair_pm_online_kernel
// repeats flagged waves.  The expansion cancels like the joint one (each dimension's terms are bounded by the
// joint conditioning estimate): gated on hdr->cond_bits when cond_limit >= 0, else the caller's risk.
#include "pair_plan.h"

typedef __bf16 pmbx_bf16x8 __attribute__((ext_vector_type(8)));

#define PMBX_SUM_MIN 8.6736174e-19f   // 2^-60   (as pair_pm_kernel)
#define PMBX_SUM_MAX 1.2676506e30f    // 2^100

__device__ __forceinline__ void pmbx_split3(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 hb = (__bf16)v;
    const float r1 = v - (float)hb;
    const __bf16 mb = (__bf16)r1;
    const __bf16 lb = (__bf16)(r1 - (float)mb);
    h = __builtin_bit_cast(unsigned short, hb);
    m = __builtin_bit_cast(unsigned short, mb);
    l = __builtin_bit_cast(unsigned short, lb);
}

__device__ __forceinline__ uint4 pmbx_pack(const unsigned short (&k)[8]) {
    return make_uint4((unsigned)k[0] | ((unsigned)k[1] << 16), (unsigned)k[2] | ((unsigned)k[3] << 16),
                      (unsigned)k[4] | ((unsigned)k[5] << 16), (unsigned)k[6] | ((unsigned)k[7] << 16));
}

// one thread per (posterior row, dimension): both k-halves.  out[blk][d][lane = row%32 + 32*khalf] (uint4)
__global__ void __launch_bounds__(256)
prep_pm_bx_a_kernel(const float* __restrict__ mu, const float* __restrict__ spread, int spread_kind, int64_t n,
                    int64_t npad, int D, int DC, const PairHeader* __restrict__ hdr, double inv_n,
                    uint4* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npad * DC) return;
    const int64_t blk = idx / (32 * DC);
    const int rem = (int)(idx - blk * 32 * DC);
    const int d = rem / 32, r = rem % 32;            // row fastest: 512-byte runs per k-half
    const int64_t row = blk * 32 + r;
    unsigned short k0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < n && d < D) {
        const float sp = spread[row * D + d];
        const float logvar = spread_kind == LVAE_SPREAD_VAR ? logf(sp) : sp;
        const float p = spread_kind == LVAE_SPREAD_VAR ? 1.0f / sp : expf(-sp);
        const float ms = mu[row * D + d] - (float)(hdr->colsum[d] * inv_n);
        const float w2 = 0.5f * LVAE_LOG2E * p;
        const float A = -w2, B = 2.0f * ms * w2;
        const float Cc = -0.5f * LVAE_LOG2E * (logvar + LVAE_LOG_2PI) - ms * ms * w2;
        unsigned short ah, am, al, bh, bm, bl, ch, cm, cl;
        pmbx_split3(A, ah, am, al);
        pmbx_split3(B, bh, bm, bl);
        pmbx_split3(Cc, ch, cm, cl);
        k0[0] = ah; k0[1] = ah; k0[2] = am; k0[3] = ah; k0[4] = al; k0[5] = am; k0[6] = bh; k0[7] = bh;
        k1[0] = bm; k1[1] = bh; k1[2] = bl; k1[3] = bm; k1[4] = ch; k1[5] = cm; k1[6] = cl;
    } else if (d < D) {
        k1[4] = 0xff7f;               // padded posterior: C = -3.39e38 (largest finite bf16), every term 2^C = 0
    }
    // (d >= D: all zero -> t = 0, a constant 2^0 per posterior in an unused dimension, never read back)
    uint4* o = out + (blk * DC + d) * 64;
    o[r] = pmbx_pack(k0);
    o[32 + r] = pmbx_pack(k1);
}

// one thread per (sample, dimension).  out[col][d][khalf] (uint4)
__global__ void __launch_bounds__(256)
prep_pm_bx_b_kernel(const float* __restrict__ z, int64_t m, int64_t cpad, int D, int DC,
                    const PairHeader* __restrict__ hdr, double inv_n, uint4* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= cpad * DC) return;
    const int64_t col = idx / DC;
    const int d = (int)(idx - col * DC);
    unsigned short k0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < m && d < D) {
        const float zs = z[col * D + d] - (float)(hdr->colsum[d] * inv_n);
        unsigned short qh, qm, ql, zh, zm, zl;
        pmbx_split3(zs * zs, qh, qm, ql);
        pmbx_split3(zs, zh, zm, zl);
        k0[0] = qh; k0[1] = qm; k0[2] = qh; k0[3] = ql; k0[4] = qh; k0[5] = qm; k0[6] = zh; k0[7] = zm;
        k1[0] = zh; k1[1] = zl; k1[2] = zh; k1[3] = zm;
    }
    k1[4] = k1[5] = k1[6] = 0x3f80;   // 1.0 (also for padded samples: finite sums, dropped by the caller)
    uint4* o = out + (col * DC + d) * 2;
    o[0] = pmbx_pack(k0);
    o[1] = pmbx_pack(k1);
}

template <int DCT>
__global__ void __launch_bounds__(256, 2)
pair_pm_bx_kernel(const uint4* __restrict__ aop, const uint4* __restrict__ bop, float2* __restrict__ partial_pm,
                  int* __restrict__ redo_flags, int64_t C, int64_t cpad, int64_t nblk, int64_t blk_per_split,
                  int DC, int d0,          // this launch covers dimensions d0 .. d0 + DCT - 1 of the DC padded ones
                  const PairHeader* __restrict__ hdr, float cond_limit) {
    // cond_limit >= 0 (AUTO path): run only for well-conditioned inputs; pair_pm_kernel carries the opposite gate
    if (cond_limit >= 0.f && __uint_as_float(hdr->cond_bits) > cond_limit) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t col_base = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int64_t b0 = (int64_t)blockIdx.y * blk_per_split;
    int64_t b1 = b0 + blk_per_split;
    if (b1 > nblk) b1 = nblk;
    const int64_t flag_idx = ((int64_t)blockIdx.y * (DC / 16) + d0 / 16) * (cpad / 64) + col_base / 64;
    if (b0 >= b1) {                          // empty split: zero sums, nothing to redo
        if (lane == 0) redo_flags[flag_idx] = 0;
        for (int d = 0; d < DCT; ++d)
            partial_pm[((int64_t)blockIdx.y * DC + d0 + d) * cpad + col_base + lane] = make_float2(0.f, 0.f);
        return;
    }
    constexpr int STAGE = DCT * 64;          // uint4 per stage: one 32-row block, all dimensions
    __shared__ uint4 ring[2 * STAGE];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nb = (int)(b1 - b0);
    // linear block counter over both passes: t -> block b0 + t % nb, stage t & 1
    auto dma = [&](int t) {
        const uint4* g = aop + ((b0 + t % nb) * (int64_t)DC + d0) * 64 + lane;
        uint4* l = ring + (t & 1) * STAGE;
#pragma unroll
        for (int q0 = 0; q0 < DCT; q0 += 4) {
            const int q = q0 + wave_u;       // DCT 1-KiB pieces per block, dealt round-robin to the four waves
            __builtin_amdgcn_global_load_lds((gptr_t)(g + q * 64), (lptr_t)(l + q * 64), 16, 0, 0);
        }
    };
    dma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    int t = 0;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t col = col_base + pass * 32 + r;
        pmbx_bf16x8 bz[DCT];
#pragma unroll
        for (int d = 0; d < DCT; ++d) bz[d] = __builtin_bit_cast(pmbx_bf16x8, bop[(col * DC + d0 + d) * 2 + h]);
        float S[DCT];
#pragma unroll
        for (int d = 0; d < DCT; ++d) S[d] = 0.f;
        for (int blk = 0; blk < nb; ++blk, ++t) {
            if (t + 1 < 2 * nb) dma(t + 1);  // next block (pass 1 re-streams the same blocks)
            const uint4* st = ring + (t & 1) * STAGE + lane;
            f32x16 acc[2];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pmbx_bf16x8, st[0]), bz[0], zero16, 0, 0, 0);
#pragma unroll
            for (int d = 0; d < DCT; ++d) {
                if (d + 1 < DCT)
                    acc[(d + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(pmbx_bf16x8, st[(d + 1) * 64]), bz[d + 1], zero16, 0, 0, 0);
                const f32x16& a = acc[d & 1];
                f32x2 kt = {lvae_exp2(a[0]), lvae_exp2(a[1])};
#pragma unroll
                for (int e = 2; e < 16; e += 2) {
                    f32x2 ex;
                    ex.x = lvae_exp2(a[e]);
                    ex.y = lvae_exp2(a[e + 1]);
                    kt += ex;
                }
                S[d] += kt.x + kt.y;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int d = 0; d < DCT; ++d) {
            float s = S[d];
            s += __shfl_xor(s, 32, 64);      // the two half-waves hold complementary posterior rows of the same sample
            ok = ok && s >= PMBX_SUM_MIN && s <= PMBX_SUM_MAX;
            if (h == 0 && col < cpad) partial_pm[((int64_t)blockIdx.y * DC + d0 + d) * cpad + col] = make_float2(0.f, s);
        }
    }
    // samples beyond C are padding (finite sums by construction); a flagged wave is redone for all its 64 samples
    const bool redo = !__all(ok);
    if (lane == 0) redo_flags[flag_idx] = redo ? 1 : 0;
}

// ---- host side -----------------------------------------------------------------------------------------
bool lvae_pm_bx_supported(int DC) { return DC >= 16 && DC <= 128 && DC % 16 == 0; }
size_t lvae_pm_bx_a_bytes(int64_t nblk, int DC) { return (size_t)nblk * DC * 64 * sizeof(uint4); }
size_t lvae_pm_bx_b_bytes(int64_t cpad, int DC) { return (size_t)cpad * DC * 2 * sizeof(uint4); }

// aop / bop: workspace regions of lvae_pm_bx_a_bytes / lvae_pm_bx_b_bytes; hdr->colsum must be complete.
int lvae_launch_pm_bx(const PairPlan& p, const float* z, const float* mu, const float* spread, int spread_kind,
                      const PairHeader* hdr, void* aop, void* bop, float2* partial_pm, int* redo_flags,
                      float cond_limit, hipStream_t st) {
    if (!lvae_pm_bx_supported(p.DC) || p.rows_per_dsplit % 32 != 0 || p.cpad % 256 != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "prod-marginals bf16x3 kernel: unsupported plan (DC=%d)", p.DC);
    const double inv = 1.0 / (double)p.R;
    const int64_t npad = p.nblk * 32;
    hipLaunchKernelGGL(prep_pm_bx_a_kernel, dim3((unsigned)lvae_ceil_div(npad * p.DC, 256)), dim3(256), 0, st, mu,
                       spread, spread_kind, p.R, npad, p.D, p.DC, hdr, inv, (uint4*)aop);
    hipLaunchKernelGGL(prep_pm_bx_b_kernel, dim3((unsigned)lvae_ceil_div(p.cpad * p.DC, 256)), dim3(256), 0, st, z,
                       p.C, p.cpad, p.D, p.DC, hdr, inv, (uint4*)bop);
    const dim3 grid((unsigned)(p.cpad / 256), (unsigned)p.dsplit);
    const int64_t bps = p.rows_per_dsplit / 32;
    for (int d0 = 0; d0 < p.DC; d0 += 32) {          // per-dimension states are independent: 32 dimensions per launch
        if (p.DC - d0 >= 32)
            hipLaunchKernelGGL((pair_pm_bx_kernel<32>), grid, dim3(256), 0, st, (const uint4*)aop, (const uint4*)bop,
                               partial_pm, redo_flags, p.C, p.cpad, p.nblk, bps, p.DC, d0, hdr, cond_limit);
        else                                         // (a 16-wide tail: its flag group index d0 / 16 is even, as
            hipLaunchKernelGGL((pair_pm_bx_kernel<16>), grid, dim3(256), 0, st, (const uint4*)aop, (const uint4*)bop,
                               partial_pm, redo_flags, p.C, p.cpad, p.nblk, bps, p.DC, d0, hdr, cond_limit);   // pair_pm_online_kernel expects)
    }
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
