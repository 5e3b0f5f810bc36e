// EXPERIMENTAL, opt-in (LVAE_PATH_MFMA_BF16X3): the pair log-density contraction on the bf16 matrix pipe at
// fp32-equivalent accuracy.  Not what BASELINE's north star asks for (an fp32 MFMA contraction -- that is
// pair_lse_mfma.hip, the default and the path bench.py measures); kept next to it because the fp32 kernel
// sits at the ceiling of its structure (fp32 MFMA shares the vector lanes with the epilogue's exponentials,
// tools/ubench/bf16x3_tile.cpp: 2348 cycles per 32x32 tile at D = 32, measured in the kernel: 2349) and the
// same tile costs 976 cycles in this form.
//
// Every fp32 operand value v is split into three bf16 limbs v = h + m + l (8 + 8 + 8 mantissa bits,
// prep_bf16x3_kernel); a product of two values is the sum of the six limb products of weight >= 2^-16
// (hh, hm, mh, hl, lh, mm; the dropped ml, lm, ll are below 2^-24 of the product, fp32 rounding level), all
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  One MFMA step covers eight latent dimensions: k-half 0
// of the A / B operands holds the eight quadratic terms (a_d ; z_d^2), k-half 1 the eight linear ones
// (b_d ; z_d).  The rest -- reduced axis on MFMA rows, LDS ring filled by LDS-DMA, row biases as the first
// C operand, ping-pong accumulators, epilogue micro-slices pinned between the MFMAs -- is the structure of
// pair_lse_mfma.hip; the epilogues are its EPI 4 (fixed reference) and EPI 0 (running maximum, redo pass).
#include "pair_plan.h"

#define BX_LSE_THR_BIG 256.0f
#define BX_LSE_MINIT (-1.0e30f)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void bx_wait_vmcnt(int n) {
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
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// ---- operand preparation ------------------------------------------------------------------------------
// One thread per (row, step of 8 dimensions, k-half): computes the eight fp32 operand values of that half
// (GAUSS rows: a_d or b_d; SAMPLE rows: z_d^2 or z_d, after the same centering shift as the fp32 path) and
// writes their three limbs as 16-byte vectors.
//   FRAG (reduced axis): out[blk][step][limb][lane = row%32 + 32*half][8]           (16 B per lane)
//   else (kept axis)   : out[row][half][step][limb][8]
// The row bias (GAUSS) and the header statistics are produced by the fp32 prep kernel, which runs anyway
// (conditioning estimate, normaliser maximum); this kernel only adds the limb operands.
template <bool GAUSS, bool FRAG>
__global__ void __launch_bounds__(256)
prep_bf16x3_kernel(const float* __restrict__ x, const float* __restrict__ spread, int spread_kind, int64_t n,
                   int64_t npad, int D, int DP, const PairHeader* __restrict__ hdr, double inv_ngauss,
                   uint4* __restrict__ out) {
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
            const int d = 8 * step + e;
            float v = 0.f;
            if (d < D) {
                const float xs = x[row * D + d] - (float)(hdr->colsum[d] * inv_ngauss);
                if (GAUSS) {
                    const float sp = spread[row * D + d];
                    const float p = spread_kind == LVAE_SPREAD_VAR ? 1.0f / sp : expf(-sp);
                    v = half == 0 ? -0.5f * LVAE_LOG2E * p : LVAE_LOG2E * xs * p;
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

// ---- the pair kernel ------------------------------------------------------------------------------------
// EPI 4: fixed-reference logsumexp; EPI 0: running maximum (redo pass, gated by redo_flags).
template <int DP, int NCB, int WPS, int EPI>
__global__ void __launch_bounds__(256, WPS)
pair_lse_bf16x3_kernel(const uint4* __restrict__ rfrag, const float* __restrict__ rbias,
                       const uint4* __restrict__ cop, float2* __restrict__ partial,
                       const PairHeader* __restrict__ hdr, int64_t nblk, int64_t blk_per_split, int64_t cpad,
                       const int* __restrict__ redo_flags) {
    if (redo_flags && redo_flags[blockIdx.x] == 0) return;
    if (EPI == 4 && hdr->lse_skip) return;
    constexpr int NS = DP / 8;                    // MFMA steps (8 dimensions each) per tile
    constexpr int PIECES = NS * 3;                // 1 KiB pieces (step, limb) per 32-row block
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * NCB);

    // kept-axis operand limbs: resident for the whole kernel
    bf16x8 bz[NCB][NS][3];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const uint4* src = cop + ((col0 + cb * 32 + r) * 2 + h) * NS * 3;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int l = 0; l < 3; ++l) bz[cb][s][l] = __builtin_bit_cast(bf16x8, src[s * 3 + l]);
    }

    float m[NCB], sacc[NCB];
    f32x2 m2[NCB], kt2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        m[cb] = BX_LSE_MINIT;
        sacc[cb] = 0.f;
        m2[cb] = (f32x2){-BX_LSE_MINIT, -BX_LSE_MINIT};
        kt2[cb] = (f32x2){0.f, 0.f};
    }

    const int64_t b0 = (int64_t)blockIdx.y * blk_per_split;
    int64_t b1 = b0 + blk_per_split;
    if (b1 > nblk) b1 = nblk;

    // LDS ring: one stage per block = PIECES KiB + a 256-B piece with the block's 32 row biases
    constexpr int STAGE_V = PIECES * 64 + 16;     // uint4 (16 B) units
    constexpr int PF = DP >= 64 ? 2 : 3, NST = PF + 1;      // D = 64: 24 KiB per stage, three stages (2 x 73 KB per CU)
    __shared__ uint4 ring[NST * STAGE_V];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int DMA_PER_WAVE = (PIECES + 3) / 4;          // pieces are dealt round-robin to the four waves
    auto dma_block = [&](int64_t blk, int stage) {
        const uint4* g = rfrag + blk * PIECES * 64 + lane;
        uint4* l = ring + stage * STAGE_V;
#pragma unroll
        for (int q0 = 0; q0 < DMA_PER_WAVE * 4; q0 += 4) {
            int q = q0 + wave_u;
            if (q >= PIECES) q = PIECES - 1;                // same count for every wave; a duplicate is harmless
            __builtin_amdgcn_global_load_lds((gptr_t)(g + q * 64), (lptr_t)(l + q * 64), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(rbias + blk * 32 + lane), (lptr_t)(l + PIECES * 64), 4, 0, 0);
    };
    const int t_end = (int)(b1 - b0);
    auto dma_linear = [&](int t) {
        if (t >= t_end) t = t_end - 1;
        dma_block(b0 + t, t % NST);
    };

    constexpr int EPI_MS = EPI == 4 ? 8 : 9;
    auto epi_micro = [&](const f32x16 (&prev)[NCB], int cb, int ms) {
        if (EPI == 4) {
            f32x2 e;
            e.x = lvae_exp2(prev[cb][2 * ms]);
            e.y = lvae_exp2(prev[cb][2 * ms + 1]);
            kt2[cb] += e;
        } else if (ms < 8) {
            const f32x2 pr = {prev[cb][2 * ms], prev[cb][2 * ms + 1]};
            f32x2 x;
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(pr), "v"(m2[cb]));
            f32x2 e;
            e.x = lvae_exp2(x.x);
            e.y = lvae_exp2(x.y);
            if (ms == 0) kt2[cb] = e;
            else kt2[cb] += e;
        } else {
            const float ts = kt2[cb].x + kt2[cb].y;
            if (__builtin_expect(__any(!(ts < BX_LSE_THR_BIG)), 0)) {
                float t = prev[cb][0];
#pragma unroll
                for (int e = 1; e < 16; ++e) t = fmaxf(t, prev[cb][e]);
                const float mn = fmaxf(m[cb], t);
                float acc = sacc[cb] * lvae_exp2(m[cb] - mn);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc += lvae_exp2(prev[cb][e] - mn);
                sacc[cb] = acc;
                m[cb] = mn;
                m2[cb] = (f32x2){-mn, -mn};
            } else {
                sacc[cb] += ts;
            }
        }
    };
    constexpr int SLOTS = NS * 6 * NCB;           // MFMAs per block
    constexpr int MS_TOTAL = EPI_MS * NCB;

    // limb pairs of the six kept products, smallest first
    constexpr int LA[6] = {1, 2, 0, 1, 0, 0};     // limb of the reduced-axis (A) operand
    constexpr int LB[6] = {1, 0, 2, 0, 1, 0};     // limb of the kept-axis (B) operand

    bf16x8 a_pre[3];
    f32x16 c_pre;
    auto read_step = [&](int stage, int s, bf16x8 (&dst)[3]) {
        const uint4* p = ring + stage * STAGE_V + s * 3 * 64 + lane;
#pragma unroll
        for (int l = 0; l < 3; ++l) dst[l] = __builtin_bit_cast(bf16x8, p[l * 64]);
    };
    auto read_bias = [&](int stage) {
        const f32x4* bl = reinterpret_cast<const f32x4*>(ring + stage * STAGE_V + PIECES * 64) + h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 c = bl[2 * g];
#pragma unroll
            for (int e = 0; e < 4; ++e) c_pre[4 * g + e] = c[e];
        }
    };

    int t_lin = 0;
    auto run_block = [&](f32x16 (&acc)[NCB], const f32x16 (&prev)[NCB], bool with_epi) {
        const f32x16 cv = c_pre;
        const int stage = t_lin % NST;
        const int stage_next = (t_lin + 1) % NST;
        dma_linear(t_lin + PF);
        bf16x8 a3[2][3];
#pragma unroll
        for (int l = 0; l < 3; ++l) a3[0][l] = a_pre[l];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the step read one step ago has landed
            if (s + 1 < NS) {
                read_step(stage, s + 1, a3[(s + 1) & 1]);
            } else {
                read_step(stage_next, 0, a_pre);
                read_bias(stage_next);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const bool first = s == 0 && pr == 0;
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[s & 1][LA[pr]], bz[cb][s][LB[pr]],
                                                                      first ? cv : acc[cb], 0, 0, 0);
                    const int u = (s * 6 + pr) * NCB + cb;
                    if (with_epi) {
#pragma unroll
                        for (int k = (u * MS_TOTAL) / SLOTS; k < ((u + 1) * MS_TOTAL) / SLOTS; ++k)
                            epi_micro(prev, k / EPI_MS, k % EPI_MS);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // the NEXT block's last step reads step 0 and the biases of block t_lin+2, so my DMAs up to block
        // t_lin+2 must have landed (block t_lin+3, if the ring is that deep, may still fly); after the
        // barrier everybody's have and everybody is done reading block t_lin's stage
        bx_wait_vmcnt((PF - 2) * (DMA_PER_WAVE + 1));
        __builtin_amdgcn_s_barrier();
        ++t_lin;
    };
    auto drain = [&](const f32x16 (&prev)[NCB]) {
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
        read_step(0, 0, a_pre);
        read_bias(0);
        run_block(accA, accB, false);
        int64_t blk = b0 + 1;
        for (; blk + 1 < b1; blk += 2) {
            run_block(accB, accA, true);
            run_block(accA, accB, true);
        }
        if (blk < b1) {
            run_block(accB, accA, true);
            drain(accB);
        } else {
            drain(accA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA in flight when the workgroup retires
    }

#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (EPI == 4) {
            const float sv = kt2[cb].x + kt2[cb].y;
            const float ss = sv + __shfl_xor(sv, 32, 64);
            if (h == 0) partial[(int64_t)blockIdx.y * cpad + col0 + cb * 32 + r] = make_float2(0.f, ss);
        } else {
            const float mo = __shfl_xor(m[cb], 32, 64);
            const float so = __shfl_xor(sacc[cb], 32, 64);
            const float mm = fmaxf(m[cb], mo);
            const float ss = sacc[cb] * lvae_exp2(m[cb] - mm) + so * lvae_exp2(mo - mm);
            if (h == 0) partial[(int64_t)blockIdx.y * cpad + col0 + cb * 32 + r] = make_float2(mm, ss);
        }
    }
}

// ---- host launchers ----------------------------------------------------------------------------------
bool lvae_bf16x3_supported(int DP) { return DP == 16 || DP == 32 || DP == 64; }

// bytes of the limb operands: reduced axis (per 32-row block) and kept axis (per element)
size_t lvae_bf16x3_rfrag_bytes(int64_t nblk, int DP) { return (size_t)nblk * (DP / 8) * 3 * 1024; }
size_t lvae_bf16x3_cop_bytes(int64_t cpad, int DP) { return (size_t)cpad * 2 * (DP / 8) * 3 * 16; }

int lvae_launch_prep_bf16x3(bool gauss, bool frag, const float* x, const float* spread, int spread_kind,
                            int64_t n, int64_t npad, int D, int DP, const PairHeader* hdr, int64_t ngauss,
                            void* out, hipStream_t st) {
    const int64_t total = npad * (DP / 8) * 2;
    const dim3 grid((unsigned)lvae_ceil_div(total, 256));
    const double inv = 1.0 / (double)ngauss;
#define LVAE_PBX(G, F)                                                                                       \
    hipLaunchKernelGGL((prep_bf16x3_kernel<G, F>), grid, dim3(256), 0, st, x, spread, spread_kind, n, npad, D, \
                       DP, hdr, inv, (uint4*)out)
    if (gauss && frag) LVAE_PBX(true, true);
    else if (!gauss && !frag) LVAE_PBX(false, false);
    else return lvae_set_error(LVAE_ERR_INVALID, "bf16x3 prep: unsupported operand role");
#undef LVAE_PBX
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

template <int DP, int NCB, int WPS>
static int launch_bx(const PairPlan& p, const void* rfrag, const float* rbias, const void* cop, float2* partial,
                     const PairHeader* hdr, int epi, const int* redo_flags, hipStream_t st) {
    const dim3 grid((unsigned)(p.cpad / (4 * 32 * NCB)), (unsigned)p.nsplit);
    if (epi == 4)
        hipLaunchKernelGGL((pair_lse_bf16x3_kernel<DP, NCB, WPS, 4>), grid, dim3(256), 0, st, (const uint4*)rfrag,
                           rbias, (const uint4*)cop, partial, hdr, p.nblk, p.blk_per_split, p.cpad, redo_flags);
    else
        hipLaunchKernelGGL((pair_lse_bf16x3_kernel<DP, NCB, WPS, 0>), grid, dim3(256), 0, st, (const uint4*)rfrag,
                           rbias, (const uint4*)cop, partial, hdr, p.nblk, p.blk_per_split, p.cpad, redo_flags);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// epi 4: fixed-reference pass; epi 0: running-maximum pass (redo_flags may gate it per workgroup).
// The plan's NCB must be 2 for DP 16 / 32 and 1 for DP 64 (lvae_make_pair_plan's defaults).
int lvae_launch_pair_bf16x3(const PairPlan& p, const void* rfrag, const float* rbias, const void* cop,
                            float2* partial, const PairHeader* hdr, int epi, const int* redo_flags,
                            hipStream_t st) {
    if (p.cpad % (4 * 32 * p.NCB) != 0)
        return lvae_set_error(LVAE_ERR_INVALID, "pair plan: cpad %lld not a multiple of the tile", (long long)p.cpad);
    if (p.DP <= 32 && p.NCB != 2) return lvae_set_error(LVAE_ERR_INVALID, "pair plan: NCB must be 2 at D <= 32");
    switch (p.DP) {
        case 16: return launch_bx<16, 2, 2>(p, rfrag, rbias, cop, partial, hdr, epi, redo_flags, st);
        case 32: return launch_bx<32, 2, 2>(p, rfrag, rbias, cop, partial, hdr, epi, redo_flags, st);
        case 64:
            if (p.NCB != 1) return lvae_set_error(LVAE_ERR_INVALID, "pair plan: NCB must be 1 at D = 64");
            return launch_bx<64, 1, 2>(p, rfrag, rbias, cop, partial, hdr, epi, redo_flags, st);
        default:
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the bf16x3 path needs D <= 64 (got D=%d)", p.D);
    }
}
