// Operand preparation and finalisation kernels for the pair log-density paths.
//
// The log-density of sample z under posterior (mu, var), scaled to base 2, is a bilinear form
//     log2e * sum_d log N(z_d; mu_d, var_d) = sum_d [ z_d^2 * a_d + z_d * b_d ] + c
//     a_d = -0.5 * log2e / var_d,  b_d = log2e * mu_d / var_d,
//     c   = -0.5 * log2e * sum_d (mu_d^2 / var_d + log var_d + log 2pi)
// so the N x M matrix is a [*, 2D] x [2D, *] contraction (half 0 = quadratic terms, half 1 = linear
// terms) plus a per-posterior bias.  Both z and mu are first shifted by the per-dimension mean of mu
// (the density only depends on z - mu), which keeps |mu| -- and with it the cancellation in the
// expansion -- as small as a global shift can.
//
// Reduced-axis operands are written "fragment-major": for each block of 32 rows, DP/4 chunks of
// [64 lanes][4 floats], exactly the order in which the lanes of a wave consume them as the A operand
// of v_mfma_f32_32x32x2_f32 (lane = row + 32*half, MFMA step s = dimension s).  One wave-wide
// 16-byte load per chunk is then a single contiguous 1 KiB read.
#include "pair_plan.h"

// Per-dimension sum of the posterior means.  Each block sweeps a contiguous slab of rows with fully
// coalesced loads; a thread's dimension is fixed when 256 % D == 0 (register accumulation), otherwise
// the slab is accumulated through LDS atomics.
#define COLSUM_ROWS 512       // rows per workgroup: few enough workgroups that the float64 atomics on the D sums stay short
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ mu, int64_t n, int D,
                                                     PairHeader* hdr, double* __restrict__ zero5) {
    __shared__ float sm[256];
    // (small path: the five float64 statistics a later kernel of this call adds into start from zero)
    if (zero5 && blockIdx.x == 0 && threadIdx.x < 5) zero5[threadIdx.x] = 0.0;
    const int64_t r0 = (int64_t)blockIdx.x * COLSUM_ROWS;
    int64_t r1 = r0 + COLSUM_ROWS;
    if (r1 > n) r1 = n;
    const int64_t e0 = r0 * D, e1 = r1 * D;
    if (D >= 4 && D <= 256 && 256 % D == 0) {
        // 16-byte loads, four in flight; thread t always sees columns (4t .. 4t+3) % D (1024 % D == 0)
        const float4* __restrict__ m4 = reinterpret_cast<const float4*>(mu);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t q0 = e0 / 4, q1 = e1 / 4;
        int64_t i = q0 + threadIdx.x;
        for (; i + 768 < q1; i += 1024) {
            const float4 a = m4[i], b = m4[i + 256], c = m4[i + 512], d = m4[i + 768];
            acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
            acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; i < q1; i += 256) {
            const float4 a = m4[i];
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
        __shared__ float sm4[1024];
        sm4[4 * threadIdx.x] = acc.x; sm4[4 * threadIdx.x + 1] = acc.y;
        sm4[4 * threadIdx.x + 2] = acc.z; sm4[4 * threadIdx.x + 3] = acc.w;
        __syncthreads();
        if (threadIdx.x < D) {
            double t = 0.0;
            for (int k = threadIdx.x; k < 1024; k += D) t += (double)sm4[k];
            atomicAdd(&hdr->colsum[threadIdx.x], t);
        }
    } else if (D <= 256 && 256 % D == 0) {
        float acc = 0.f;
        for (int64_t i = e0 + threadIdx.x; i < e1; i += 256) acc += mu[i];
        sm[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < D) {
            double t = 0.0;
            for (int k = threadIdx.x; k < 256; k += D) t += (double)sm[k];
            atomicAdd(&hdr->colsum[threadIdx.x], t);
        }
    } else {
        for (int d0 = 0; d0 < D; d0 += 256) {      // D > 256 or odd D: LDS atomics, 256 dims at a time
            sm[threadIdx.x] = 0.f;
            __syncthreads();
            for (int64_t i = e0 + threadIdx.x; i < e1; i += 256) {
                const int d = (int)(i % D);
                if (d >= d0 && d < d0 + 256) atomicAdd(&sm[d - d0], mu[i]);
            }
            __syncthreads();
            if (d0 + threadIdx.x < D && d0 + threadIdx.x < 128)
                atomicAdd(&hdr->colsum[d0 + threadIdx.x], (double)sm[threadIdx.x]);
            __syncthreads();
        }
    }
}

// One thread per (row, 4 consecutive dimensions), dimension-fastest: the DP/4 threads of a row are consecutive
// lanes of one wave, so a row is read with coalesced 16-byte loads and reduced with shuffles; a workgroup
// covers PREP_RB consecutive 32-row blocks (= fragment blocks) and issues one set of header atomics.
// SRC 0 (GAUSS): rows from (mu, spread); SRC 1 (SAMPLE): rows from z -> (z^2, z);
// SRC 2 (POINT): rows from y -> (1, -2y) with bias |y|^2, so that the contraction with a SAMPLE operand
// is the squared distance |x - y|^2 (kernel-matrix sums; padded rows get +inf).  FRAG selects the fragment-major
// reduced-axis layout (the 32 rows of a block at one q are 512 contiguous bytes; a wave writes 128-byte runs),
// otherwise row-major [row][2*DP] for the kept axis.  `out` may be NULL (bias / header statistics only).
// SRC 3 / 4 (kernel sums): linear-only operands with TWO dimensions per MFMA step (DP = padded D / 2): POINT
// rows -2y (k-half 0: dimension 2s, k-half 1: dimension 2s + 1) with bias |y|^2 (+inf on padded rows), SAMPLE
// rows x with `bias` receiving |x|^2 (the per-column norm the epilogue adds); no centering shift.
#define PREP_RB 2      // (C2 posteriors, 65,536 x 32: 11.3 us at 1 or 2 blocks per workgroup, 12.8 at 4, 18 at 8)
// (a device function so that one launch can prepare both operands of a small problem, prep_pair_kernel below:
//  `wg` stands for blockIdx.x, `rb` for the 32-row blocks a workgroup covers)
template <int SRC, bool FRAG>
__device__ __forceinline__ void
prep_rows(const int64_t wg, const int rb, const float* __restrict__ x, const float* __restrict__ spread, int spread_kind,
          int64_t n, int64_t npad, int D, int DP, PairHeader* hdr, double inv_ngauss,
          float* __restrict__ out, float* __restrict__ bias, float* __restrict__ dparams,
          int DC, float cond_floor = -1.f, bool want_norm = true) {
    constexpr bool GAUSS = SRC == 0;
    constexpr bool POINT = SRC == 2;
    constexpr bool LIN = SRC == 3 || SRC == 4;
    __shared__ float sm_shift[128];
    __shared__ float sm_w[16][3];
    const int NQ = DP >> 2;                    // 4, 8, 16 or 32 lanes per row
    const int tid = threadIdx.x;
    // Kernel sums with a polynomial epilogue (one exponential per pair, t = exp2(-base log2e (eps + |d2|))): the scale
    // base log2e is folded into the operands and eps into the row bias, so that the contraction delivers the exponent
    // itself and the epilogue is exp2(-|acc + |x|^2|) with no multiply-add in front of it (the header's parameters are
    // set before any operand is prepared)
    const float lin_s = (LIN && hdr->ksum.poly) ? -hdr->ksum.nbase : 1.f;
    const float lin_e = (LIN && hdr->ksum.poly) ? hdr->ksum.eps : 0.f;
    const float lin_m = SRC == 3 ? -2.0f * lin_s : 1.f;
    if (!LIN && tid < DP) sm_shift[tid] = tid < D ? (float)(hdr->colsum[tid] * inv_ngauss) : 0.f;
    __syncthreads();
    const int r = tid / NQ;
    const int q = tid % NQ;
    const bool vec = (D & 3) == 0 && 4 * q + 4 <= D;
    float cond_max = 0.f, norm_max = -INFINITY, norm_acc = 0.f;
    for (int it = 0; it < rb; ++it) {
        const int64_t blk = wg * rb + it;
        if (blk * 32 >= npad) break;
        const int64_t row = blk * 32 + r;
        const bool valid = row < n;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        float csum = 0.f, cond = 0.f, nsum = 0.f;
        if (LIN) {
            if (valid) {      // eight consecutive dimensions of the row: pairs (2s, 2s + 1) for the four steps 4q .. 4q + 3
                float xr[8];
                if ((D & 3) == 0 && 8 * q + 8 <= D) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(x + row * D + 8 * q);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(x + row * D + 8 * q + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { xr[e] = lo[e]; xr[4 + e] = hi[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[e] = 8 * q + e < D ? x[row * D + 8 * q + e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = xr[2 * e];
                    const float x1 = xr[2 * e + 1];
                    v0[e] = lin_m * x0;
                    v1[e] = lin_m * x1;
                    csum += x0 * x0 + x1 * x1;
                }
            }
        } else if (valid) {
            f32x4 xv4 = {0.f, 0.f, 0.f, 0.f}, sp4 = xv4;
            if (vec) {
                xv4 = *reinterpret_cast<const f32x4*>(x + row * D + 4 * q);
                if (GAUSS) sp4 = *reinterpret_cast<const f32x4*>(spread + row * D + 4 * q);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < D) {
                        xv4[e] = x[row * D + 4 * q + e];
                        if (GAUSS) sp4[e] = spread[row * D + 4 * q + e];
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = 4 * q + e;
                if (d < D) {
                    const float xv = xv4[e];
                    const float xs = xv - sm_shift[d];
                    if (GAUSS) {
                        const float sp = sp4[e];
                        const float logvar = spread_kind == LVAE_SPREAD_VAR ? logf(sp) : sp;
                        const float p = spread_kind == LVAE_SPREAD_VAR ? 1.0f / sp : expf(-sp);
                        v0[e] = -0.5f * LVAE_LOG2E * p;
                        v1[e] = LVAE_LOG2E * xs * p;
                        csum += xs * xs * p + logvar + LVAE_LOG_2PI;
                        nsum += logvar + LVAE_LOG_2PI;
                        const float t = 2.0f * fabsf(xs) * sqrtf(p) + 3.0f;
                        cond += 0.5f * t * t;
                        if (dparams) {
                            dparams[(row * 3 + 0) * DC + d] = xv;
                            dparams[(row * 3 + 1) * DC + d] = sqrtf(0.5f * LVAE_LOG2E * p);
                            dparams[(row * 3 + 2) * DC + d] = -0.5f * LVAE_LOG2E * (logvar + LVAE_LOG_2PI);
                        }
                    } else if (POINT) {
                        v0[e] = 1.0f;
                        v1[e] = -2.0f * xs;
                        csum += xs * xs;
                    } else {
                        v0[e] = xs * xs;
                        v1[e] = xs;
                    }
                } else if (GAUSS && dparams && d < DC) {
                    dparams[(row * 3 + 0) * DC + d] = 0.f;
                    dparams[(row * 3 + 1) * DC + d] = 0.f;
                    dparams[(row * 3 + 2) * DC + d] = 0.f;
                }
            }
        }
        if (row < npad && out) {
            if (FRAG) {
                f32x4* o = reinterpret_cast<f32x4*>(out + blk * (int64_t)DP * 64 + q * 256);
                o[r] = v0;
                o[32 + r] = v1;
            } else {
                *reinterpret_cast<f32x4*>(out + row * 2 * DP + 4 * q) = v0;
                *reinterpret_cast<f32x4*>(out + row * 2 * DP + DP + 4 * q) = v1;
            }
        }
        if (bias || GAUSS) {
            for (int off = NQ >> 1; off >= 1; off >>= 1) {       // the row's NQ lanes are consecutive
                csum += __shfl_xor(csum, off, 64);
                cond += __shfl_xor(cond, off, 64);
                nsum += __shfl_xor(nsum, off, 64);
            }
            // kernel sums behind the split-bf16 gate (lvae_api.hip, KsumMode): a squared norm above the gate's limit is
            // published (the rare case; ordinary inputs issue no atomics)
            if (LIN && valid && q == 0 && cond_floor >= 0.f && csum > cond_floor)
                atomicMax(&hdr->cond_bits, __float_as_uint(csum));
            if (bias && q == 0 && row < npad) {
                float bv = GAUSS ? -0.5f * LVAE_LOG2E * csum
                                 : (LIN ? (csum + (SRC == 3 ? lin_e : 0.f)) * lin_s : (POINT ? csum : 0.f));
                if (!valid) bv = FRAG ? ((POINT || LIN) ? INFINITY : -INFINITY) : 0.f;
                bias[row] = bv;
            }
            if (GAUSS && valid) {
                // the density's normaliser bounds every term of the row: v_ij <= -0.5 log2e sum_d log(2 pi var)
                const float nrm = -0.5f * LVAE_LOG2E * nsum;
                cond_max = fmaxf(cond_max, cond);
                norm_max = fmaxf(norm_max, nrm);
                if (q == 0) norm_acc += nrm;
            }
        }
    }
    if (GAUSS) {          // header statistics: wave shuffles, then one set of atomics per workgroup
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            cond_max = fmaxf(cond_max, __shfl_xor(cond_max, off, 64));
            norm_max = fmaxf(norm_max, __shfl_xor(norm_max, off, 64));
            norm_acc += __shfl_xor(norm_acc, off, 64);
        }
        if ((tid & 63) == 0) {
            sm_w[tid >> 6][0] = cond_max;
            sm_w[tid >> 6][1] = norm_max;
            sm_w[tid >> 6][2] = norm_acc;
        }
        __syncthreads();
        if (tid == 0) {
            const int nw = (blockDim.x + 63) >> 6;
            for (int w = 1; w < nw; ++w) {
                cond_max = fmaxf(cond_max, sm_w[w][0]);
                norm_max = fmaxf(norm_max, sm_w[w][1]);
                norm_acc += sm_w[w][2];
            }
            if (norm_max > -INFINITY) {      // at least one valid row
                // (the small path only needs to know WHETHER the conditioning estimate exceeds the gate's limit, and no
                //  normaliser statistics: for well-conditioned inputs its workgroups issue no atomics at all -- 128
                //  workgroups' atomics on these three words were a third of the launch at 4,096 posteriors)
                if (cond_max > cond_floor) atomicMax(&hdr->cond_bits, __float_as_uint(cond_max));
                if (want_norm) {
                    NormSlot* sl = &hdr->slot[wg & (LVAE_NORM_SLOTS - 1)];
                    atomicMax(&sl->cmax_key, lvae_float_key(norm_max));
                    if (FRAG) atomicAdd(&sl->norm_sum, (double)norm_acc);
                }
            }
        }
    }
}

template <int SRC, bool FRAG>
__global__ void __launch_bounds__(1024)
prep_operand_kernel(const float* __restrict__ x, const float* __restrict__ spread, int spread_kind,
                    int64_t n, int64_t npad, int D, int DP, PairHeader* hdr, double inv_ngauss,
                    float* __restrict__ out, float* __restrict__ bias, float* __restrict__ dparams,
                    int DC, int rb, float cond_floor) {
    prep_rows<SRC, FRAG>(blockIdx.x, rb, x, spread, spread_kind, n, npad, D, DP, hdr, inv_ngauss, out, bias,
                         dparams, DC, cond_floor);
}

// Both operands of a small log q(z) problem in one launch (one 32-row block per workgroup: many small workgroups
// instead of a few large ones): workgroups below wg_gauss prepare posterior rows (fragment-major, biases, header
// statistics, direct-form parameters), the others sample rows.
__global__ void __launch_bounds__(1024)
prep_pair_kernel(const float* __restrict__ mu, const float* __restrict__ spread, int spread_kind, int64_t n,
                 int64_t npad_r, const float* __restrict__ z, int64_t m, int64_t npad_c, int D, int DP,
                 PairHeader* hdr, double inv_ngauss, float* __restrict__ rfrag, float* __restrict__ rbias,
                 float* __restrict__ dparams, int DC, float* __restrict__ cop, int64_t wg_gauss, float cond_floor,
                 int want_norm) {
    if ((int64_t)blockIdx.x < wg_gauss)
        prep_rows<0, true>(blockIdx.x, 1, mu, spread, spread_kind, n, npad_r, D, DP, hdr, inv_ngauss, rfrag, rbias,
                           dparams, DC, cond_floor, want_norm != 0);
    else
        prep_rows<1, false>((int64_t)blockIdx.x - wg_gauss, 1, z, nullptr, 0, m, npad_c, D, DP, hdr, inv_ngauss, cop,
                            nullptr, nullptr, DC);
}

// Up to four kernel-sum operands in one launch (the POINT and SAMPLE forms of the two samples of an MMD): a workgroup
// finds its job from the first-workgroup table, one 32-row block per workgroup.
struct PrepJobs {
    const float* x[4];
    float* out[4];
    float* bias[4];
    int64_t n[4], npad[4], wg_begin[5];
    int point[4];                  // 1: reduced-axis POINT rows (fragment-major, -2x, bias |x|^2); 0: kept-axis SAMPLE rows
    int njobs;
};

__global__ void __launch_bounds__(1024)
prep_ksum_multi_kernel(const PrepJobs J, int D, int DP, PairHeader* hdr) {
    int k = 0;
    while (k + 1 < J.njobs && (int64_t)blockIdx.x >= J.wg_begin[k + 1]) ++k;
    const int64_t wg = (int64_t)blockIdx.x - J.wg_begin[k];
    if (J.point[k])
        prep_rows<3, true>(wg, 1, J.x[k], nullptr, 0, J.n[k], J.npad[k], D, DP, hdr, 1.0, J.out[k], J.bias[k], nullptr, 0);
    else
        prep_rows<4, false>(wg, 1, J.x[k], nullptr, 0, J.n[k], J.npad[k], D, DP, hdr, 1.0, J.out[k], J.bias[k], nullptr, 0);
}

int lvae_launch_prep_ksum_multi(const float* const* x, const int64_t* n, const int64_t* npad, float* const* out,
                                float* const* bias, const int* point, int njobs, int D, int DP, PairHeader* hdr,
                                hipStream_t st) {
    PrepJobs J{};
    J.njobs = njobs;
    int64_t wg = 0;
    for (int k = 0; k < njobs; ++k) {
        J.x[k] = x[k]; J.out[k] = out[k]; J.bias[k] = bias[k]; J.n[k] = n[k]; J.npad[k] = npad[k]; J.point[k] = point[k];
        J.wg_begin[k] = wg;
        wg += npad[k] / 32;
    }
    J.wg_begin[njobs] = wg;
    hipLaunchKernelGGL(prep_ksum_multi_kernel, dim3((unsigned)wg), dim3(32 * (DP / 4)), 0, st, J, D, DP, hdr);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// Totals of up to three sets of float64 kernel-sum partials [nsplit][cpad] in one launch (blockIdx.y = set), added into
// out[set] -- which an earlier kernel of the call has zeroed (ksum_init_kernel).
struct KsumSets {
    const double* partial[3];
    int nsplit[3];
    int64_t cpad[3], C[3];
};
__global__ void __launch_bounds__(256) finalize_ksum_multi_kernel(const KsumSets S, double* __restrict__ out) {
    const int y = blockIdx.y;
    const double* __restrict__ partial = S.partial[y];
    const int64_t cpad = S.cpad[y], C = S.C[y];
    double acc = 0.0;
    for (int k = 0; k < S.nsplit[y]; ++k)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < C; i += (int64_t)gridDim.x * 256)
            acc += partial[k * cpad + i];
    acc = wave_sum_d(acc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[y], sm[0] + sm[1] + sm[2] + sm[3]);
}

int lvae_launch_finalize_ksum_multi(const double* const* partial, const int* nsplit, const int64_t* cpad, const int64_t* C,
                                    int nsets, double* out, hipStream_t st) {
    KsumSets S{};
    int64_t cmax = 1;
    for (int k = 0; k < nsets; ++k) {
        S.partial[k] = partial[k]; S.nsplit[k] = nsplit[k]; S.cpad[k] = cpad[k]; S.C[k] = C[k];
        if (C[k] > cmax) cmax = C[k];
    }
    int64_t blocks = lvae_ceil_div(cmax, 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(finalize_ksum_multi_kernel, dim3((unsigned)blocks, (unsigned)nsets), dim3(256), 0, st, S, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// Direct-path parameters only (used when the MFMA path is not available for this D).
// gate_hdr != NULL: only when the conditioning estimate selects the direct path (cond > cond_limit).
__device__ __forceinline__ void dparams_element(const float* __restrict__ mu, const float* __restrict__ spread,
                                                int spread_kind, int64_t n, int D, int DC,
                                                float* __restrict__ dparams, int64_t idx);

__global__ void prep_dparams_kernel(const float* __restrict__ mu, const float* __restrict__ spread,
                                    int spread_kind, int64_t n, int D, int DC,
                                    float* __restrict__ dparams, const PairHeader* __restrict__ gate_hdr,
                                    float cond_limit) {
    if (gate_hdr && !(__uint_as_float(gate_hdr->cond_bits) > cond_limit)) return;
    dparams_element(mu, spread, spread_kind, n, D, DC, dparams, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__device__ __forceinline__ void dparams_element(const float* __restrict__ mu, const float* __restrict__ spread,
                                                int spread_kind, int64_t n, int D, int DC,
                                                float* __restrict__ dparams, int64_t idx) {
    if (idx >= n * DC) return;
    const int64_t row = idx / DC;
    const int d = (int)(idx - row * DC);
    float m = 0.f, w = 0.f, b = 0.f;
    if (d < D) {
        const float sp = spread[row * D + d];
        const float logvar = spread_kind == LVAE_SPREAD_VAR ? logf(sp) : sp;
        const float p = spread_kind == LVAE_SPREAD_VAR ? 1.0f / sp : expf(-sp);
        m = mu[row * D + d];
        w = sqrtf(0.5f * LVAE_LOG2E * p);
        b = -0.5f * LVAE_LOG2E * (logvar + LVAE_LOG_2PI);
    }
    dparams[(row * 3 + 0) * DC + d] = m;
    dparams[(row * 3 + 1) * DC + d] = w;
    dparams[(row * 3 + 2) * DC + d] = b;
}

// Fixed-reference logsumexp: shift the row biases by ref = (largest normaliser) - LSE_HEADROOM, so that every
// term 2^(v_ij - ref) is at most 2^LSE_HEADROOM and the pair kernel needs no running maximum.
// A sample's total is dominated by posteriors with a typical normaliser; if the mean normaliser is more
// than LSE_SKIP_SPREAD bits below the largest one, almost every column would come out below the floor and be
// redone anyway, so the fixed-reference pass is switched off for this call (lse_skip).
// Workgroups from `shift_blocks` on (AUTO path only) write the direct-form parameters iff the conditioning estimate
// selects the direct path -- the gated prep_dparams_kernel riding on this launch instead of one of its own.
__global__ void __launch_bounds__(256) lse_shift_bias_kernel(float* __restrict__ rbias, int64_t n, int64_t n_valid,
                                                             PairHeader* hdr, unsigned shift_blocks,
                                                             const float* __restrict__ dp_mu,
                                                             const float* __restrict__ dp_spread, int spread_kind,
                                                             int64_t dp_n, int D, int DC, float* __restrict__ dparams,
                                                             float cond_limit) {
    if (blockIdx.x >= shift_blocks) {
        if (!(__uint_as_float(hdr->cond_bits) > cond_limit)) return;
        dparams_element(dp_mu, dp_spread, spread_kind, dp_n, D, DC, dparams,
                        (int64_t)(blockIdx.x - shift_blocks) * 256 + threadIdx.x);
        return;
    }
    // every workgroup folds the LVAE_NORM_SLOTS slots itself (one wave, one slot per lane: 8 KB out of L2)
    __shared__ float sm_cmax;
    __shared__ double sm_sum;
    if (threadIdx.x < 64) {
        static_assert(LVAE_NORM_SLOTS == 64, "one slot per lane of the first wave");
        unsigned key = hdr->slot[threadIdx.x].cmax_key;
        double ns = hdr->slot[threadIdx.x].norm_sum;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned ko = (unsigned)__shfl_xor((int)key, o, 64);
            key = ko > key ? ko : key;
        }
        ns = wave_sum_d(ns);
        if (threadIdx.x == 0) {
            sm_cmax = lvae_key_float(key);
            sm_sum = ns;
        }
    }
    __syncthreads();
    const float cmax = sm_cmax;
    const float ref = cmax - LSE_HEADROOM;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rbias[i] -= ref;          // padding rows stay -inf
    if (i == 0) {
        hdr->lse_ref = ref;
        const float mean = (float)(sm_sum / (double)n_valid);
        hdr->lse_skip = (cmax - mean > LSE_SKIP_SPREAD) ? 1u : 0u;
    }
}

// Merge the per-split (max, sum) partials (base-2 domain) of one kept element into its result.
//   out      : ln2 * (M + log2 S + colbias + ref) - log_norm      (ref = hdr->lse_ref if use_ref)
//   out_max / out_sum : un-normalised natural-log pair for cross-device combination
// flags_out: fixed-reference pass -- a column whose total is not >= LSE_FLOOR (underflow, NaN) raises the
//            redo flag of its MFMA workgroup (group_cols columns per flag);
// flags_in : redo pass -- only columns whose flag is set are (re)written.
__global__ void finalize_lse_kernel(const float2* __restrict__ partial, int nsplit, int64_t cpad,
                                    int64_t C, const float* __restrict__ colbias, float log_norm,
                                    float* __restrict__ out, float* __restrict__ out_max,
                                    float* __restrict__ out_sum, const PairHeader* __restrict__ hdr,
                                    float cond_limit, int use_if_above, int use_ref,
                                    int* __restrict__ flags_out, const int* __restrict__ flags_in,
                                    int group_cols) {
    if (cond_limit >= 0.f) {   // same path gate as the pair kernels
        const float cond = __uint_as_float(hdr->cond_bits);
        if ((cond > cond_limit) != (use_if_above != 0)) return;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    if (flags_in && flags_in[i / group_cols] == 0) return;
    if (flags_out && hdr->lse_skip) {        // no fixed-reference pass was run: everything goes to the redo
        flags_out[i / group_cols] = 1;
        return;
    }
    float mx = -INFINITY;
    for (int k = 0; k < nsplit; ++k) mx = fmaxf(mx, partial[k * cpad + i].x);
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) {
        // (float32 exponent: the scale is exact -- 1 -- for the fixed-reference partials, which all carry 0, and
        //  good to 1 ulp of a float32 sum otherwise; the float64 routine was the whole cost of this kernel)
        const float2 p = partial[k * cpad + i];
        if (p.y > 0.f) s += (double)p.y * (double)(p.x == mx ? 1.0f : lvae_exp2(p.x - mx));
    }
    if (flags_out && !(s >= (double)LSE_FLOOR)) flags_out[i / group_cols] = 1;
    const double cb = (colbias ? (double)colbias[i] : 0.0) + (use_ref ? (double)hdr->lse_ref : 0.0);
    if (out) out[i] = (float)(((double)mx + log2(s) + cb) * LVAE_LN2 - (double)log_norm);
    if (out_max) {
        out_max[i] = (float)(((double)mx + cb) * LVAE_LN2);
        out_sum[i] = (float)s;
    }
}

// Same for the product-of-marginals partials [dsplit][DC][cpad], in two steps so that the merge runs on
// D x C threads: (1) one thread per (dimension, kept element) merges the splits in one pass (running
// maximum; the fast pass's partials all carry the reference 0) -> tmp[d][i] = ln2 (M + log2 S) - log_norm;
// (2) out[i] = sum_{d<D} tmp[d][i], in order.
__global__ void __launch_bounds__(256)
finalize_pm_terms_kernel(const float2* __restrict__ partial, int nsplit, int64_t cpad, int64_t C, int DC,
                         float log_norm, float* __restrict__ tmp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int d = blockIdx.y;
    if (i >= C) return;
    float mx = -INFINITY;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) {
        const float2 p = partial[((int64_t)k * DC + d) * cpad + i];
        if (p.y > 0.f) {
            if (p.x > mx) {
                s *= (double)lvae_exp2(mx - p.x);      // exp2(-inf) = 0 the first time
                mx = p.x;
            }
            s += (double)p.y * (double)lvae_exp2(p.x - mx);
        }
    }
    tmp[(int64_t)d * cpad + i] = (float)(((double)mx + log2(s)) * LVAE_LN2 - (double)log_norm);
}

__global__ void __launch_bounds__(256)
finalize_pm_sum_kernel(const float* __restrict__ tmp, int64_t cpad, int64_t C, int D, float* __restrict__ out) {
    // 64 samples per workgroup, the four waves each take every fourth dimension (a thread per sample walking all D
    // dimensions was one workgroup and 768 dependent double additions at a training batch of 256, D = 768: 0.18 ms)
    __shared__ double sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    double total = 0.0;
    if (i < C)
        for (int d = w; d < D; d += 4) total += (double)tmp[(int64_t)d * cpad + i];
    sm[w][lane] = total;
    __syncthreads();
    if (w == 0 && i < C) out[i] = (float)((sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]));
}

// Small inputs (n * D <= COLSUM_SINGLE_MAX, D a multiple of 4 dividing 4096): ONE workgroup clears the header and writes
// the column sums itself -- no hipMemsetAsync ahead of it (a fill packet followed by a dependent kernel costs an
// 11 us gap on the queue) and no atomics.
#define COLSUM_SINGLE_MAX 262144
__global__ void __launch_bounds__(1024)
colsum_single_kernel(const float* __restrict__ mu, int64_t n, int D, PairHeader* hdr, size_t clear_bytes,
                     double* __restrict__ zero5) {
    __shared__ float sm4[4096];
    __shared__ double sm_d[128 * 8];
    const int tid = threadIdx.x;
    unsigned* w = reinterpret_cast<unsigned*>(hdr);
    for (size_t i = tid; i < clear_bytes / 4; i += 1024) w[i] = 0u;
    if (zero5 && tid < 5) zero5[tid] = 0.0;
    const f32x4* __restrict__ m4 = reinterpret_cast<const f32x4*>(mu);
    const int64_t q1 = n * D / 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int64_t i = tid;
    for (; i + 3072 < q1; i += 4096) {
        const f32x4 a = m4[i], b = m4[i + 1024], c = m4[i + 2048], d = m4[i + 3072];
        acc += (a + b) + (c + d);
    }
    for (; i < q1; i += 1024) acc += m4[i];
    *reinterpret_cast<f32x4*>(sm4 + 4 * tid) = acc;
    __syncthreads();               // (also orders the clearing stores above before the sums below)
    // column d is held by the floats d, d + D, d + 2D, ...: eight threads per column, then one adds the eight
    const int per = 4096 / D;
    if (tid < D * 8) {
        const int d = tid >> 3, part = tid & 7;
        double t = 0.0;
        for (int k = part; k < per; k += 8) t += (double)sm4[d + k * D];
        sm_d[tid] = t;
    }
    __syncthreads();
    if (tid < D) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm_d[tid * 8 + k];
        hdr->colsum[tid] = t;
    }
}

bool lvae_colsum_single_ok(int64_t n, int D) {
    return D >= 4 && D <= 128 && (D & 3) == 0 && 4096 % D == 0 && n * D <= COLSUM_SINGLE_MAX;
}

// Clears `clear_bytes` of the workspace from the header on (a multiple of 4, at least sizeof(PairHeader)) and
// accumulates the column sums, in one launch.  Only when lvae_colsum_single_ok.
int lvae_launch_colsum_single(const float* mu, int64_t n, int D, PairHeader* hdr, size_t clear_bytes, hipStream_t st,
                              double* zero5) {
    if (!lvae_colsum_single_ok(n, D) || clear_bytes < sizeof(PairHeader) || (clear_bytes & 3))
        return lvae_set_error(LVAE_ERR_INVALID, "colsum_single: unsupported shape");
    hipLaunchKernelGGL(colsum_single_kernel, dim3(1), dim3(1024), 0, st, mu, n, D, hdr, clear_bytes, zero5);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- host launchers ----------------------------------------------------------------------------
int lvae_launch_colsum(const float* mu, int64_t n, int D, PairHeader* hdr, hipStream_t st, double* zero5) {
    if (D > 128) return lvae_set_error(LVAE_ERR_INVALID, "colsum: D > 128");
    const int64_t blocks = lvae_ceil_div(n, COLSUM_ROWS);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mu, n, D, hdr, zero5);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_prep(int src, bool frag, const float* x, const float* spread, int spread_kind,
                     int64_t n, int64_t npad, int D, int DP, PairHeader* hdr, int64_t ngauss,
                     float* out, float* bias, float* dparams, int DC, hipStream_t st, float cond_floor) {
    const int threads = 32 * (DP / 4);
    int rb = PREP_RB;
    if (lvae_switches().prep_rb > 0) rb = lvae_switches().prep_rb;   // (sweeps: LVAE_PREP_RB)
    const dim3 grid((unsigned)lvae_ceil_div(npad, 32 * rb));
    const double inv = 1.0 / (double)ngauss;
#define LVAE_PREP(S, F)                                                                          \
    hipLaunchKernelGGL((prep_operand_kernel<S, F>), grid, dim3(threads), 0, st, x, spread,        \
                       spread_kind, n, npad, D, DP, hdr, inv, out, bias, dparams, DC, rb, cond_floor)
    if (src == 3 && frag) LVAE_PREP(3, true);
    else if (src == 4 && !frag) LVAE_PREP(4, false);
    else if (src == 0 && frag) LVAE_PREP(0, true);
    else if (src == 0) LVAE_PREP(0, false);
    else if (src == 1 && frag) LVAE_PREP(1, true);
    else if (src == 1) LVAE_PREP(1, false);
    else if (src == 2 && frag) LVAE_PREP(2, true);
    else return lvae_set_error(LVAE_ERR_INVALID, "prep: unsupported operand source");
#undef LVAE_PREP
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_prep_pair(const float* mu, const float* spread, int spread_kind, int64_t n, int64_t npad_r,
                          const float* z, int64_t m, int64_t npad_c, int D, int DP, PairHeader* hdr, float* rfrag,
                          float* rbias, float* dparams, int DC, float* cop, float cond_floor, hipStream_t st,
                          bool want_norm) {
    const int threads = 32 * (DP / 4);
    const int64_t wg_r = npad_r / 32, wg_c = npad_c / 32;
    hipLaunchKernelGGL(prep_pair_kernel, dim3((unsigned)(wg_r + wg_c)), dim3(threads), 0, st, mu, spread, spread_kind, n,
                       npad_r, z, m, npad_c, D, DP, hdr, 1.0 / (double)n, rfrag, rbias, dparams, DC, cop, wg_r, cond_floor,
                       want_norm ? 1 : 0);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_prep_dparams(const float* mu, const float* spread, int spread_kind, int64_t n, int D,
                             int DC, float* dparams, const PairHeader* gate_hdr, float cond_limit,
                             hipStream_t st) {
    const int64_t total = n * DC;
    hipLaunchKernelGGL(prep_dparams_kernel, dim3((unsigned)lvae_ceil_div(total, 256)), dim3(256), 0,
                       st, mu, spread, spread_kind, n, D, DC, dparams, gate_hdr, cond_limit);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_finalize(const float2* partial, int nsplit, int64_t cpad, int64_t C,
                         const float* colbias, float log_norm, float* out, float* out_max,
                         float* out_sum, const PairHeader* hdr, float cond_limit, int use_if_above,
                         int use_ref, int* flags_out, const int* flags_in, int group_cols, hipStream_t st) {
    hipLaunchKernelGGL(finalize_lse_kernel, dim3((unsigned)lvae_ceil_div(C, 256)), dim3(256), 0, st,
                       partial, nsplit, cpad, C, colbias, log_norm, out, out_max, out_sum, hdr,
                       cond_limit, use_if_above, use_ref, flags_out, flags_in, group_cols);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// dp_mu != NULL: also the gated direct-form parameters (dp_n posteriors), see the kernel.
int lvae_launch_lse_shift_bias(float* rbias, int64_t n, int64_t n_valid, PairHeader* hdr, hipStream_t st,
                               const float* dp_mu, const float* dp_spread, int spread_kind, int64_t dp_n, int D, int DC,
                               float* dparams, float cond_limit) {
    const unsigned shift_blocks = (unsigned)lvae_ceil_div(n, 256);
    const unsigned dp_blocks = dp_mu ? (unsigned)lvae_ceil_div(dp_n * DC, 256) : 0u;
    hipLaunchKernelGGL(lse_shift_bias_kernel, dim3(shift_blocks + dp_blocks), dim3(256), 0, st, rbias, n,
                       n_valid, hdr, shift_blocks, dp_mu, dp_spread, spread_kind, dp_n, D, DC, dparams, cond_limit);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_finalize_pm(const float2* partial, int nsplit, int64_t cpad, int64_t C, int D, int DC,
                            float log_norm, float* tmp, float* out, hipStream_t st) {
    hipLaunchKernelGGL(finalize_pm_terms_kernel, dim3((unsigned)lvae_ceil_div(C, 256), (unsigned)D), dim3(256), 0,
                       st, partial, nsplit, cpad, C, DC, log_norm, tmp);
    LVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(finalize_pm_sum_kernel, dim3((unsigned)lvae_ceil_div(C, 64)), dim3(256), 0, st, tmp, cpad,
                       C, D, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// Sum of the per-column kernel sums (float64 partials [nsplit][cpad]) over the valid columns.
__global__ void __launch_bounds__(256) finalize_ksum_kernel(const double* __restrict__ partial, int nsplit,
                                                            int64_t cpad, int64_t C, double* __restrict__ out) {
    double acc = 0.0;
    for (int k = 0; k < nsplit; ++k)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < C; i += (int64_t)gridDim.x * 256)
            acc += partial[k * cpad + i];
    acc = wave_sum_d(acc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sm[0] + sm[1] + sm[2] + sm[3]);
}

int lvae_launch_finalize_ksum(const double* partial, int nsplit, int64_t cpad, int64_t C, double* out,
                              hipStream_t st) {
    int64_t blocks = lvae_ceil_div(C, 256);
    if (blocks > 256) blocks = 256;
    LVAE_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
    hipLaunchKernelGGL(finalize_ksum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, partial, nsplit, cpad,
                       C, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// Per-kept-element kernel sums from the float64 partials [nsplit][cpad], plus the alpha = 0 bandwidths'
// constant (+1 per counted pair): colsum[i] = sum_k partial[k][i] + zero_alphas * (#pairs counted for i).
__global__ void __launch_bounds__(256) ksum_columns_kernel(const double* __restrict__ partial, int nsplit,
                                                           int64_t cpad, int64_t C, int64_t R, int excl,
                                                           int64_t excl_offset, double zero_alphas,
                                                           double* __restrict__ colsum, double* __restrict__ total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    if (i < C) {
        for (int k = 0; k < nsplit; ++k) acc += partial[k * cpad + i];
        const int64_t js = i + excl_offset;
        const int64_t pairs = R - ((excl && js >= 0 && js < R) ? 1 : 0);
        acc += zero_alphas * (double)pairs;
        if (colsum) colsum[i] = acc;
    }
    acc = wave_sum_d(acc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && total) atomicAdd(total, sm[0] + sm[1] + sm[2] + sm[3]);
}

int lvae_launch_ksum_columns(const double* partial, int nsplit, int64_t cpad, int64_t C, int64_t R, int excl,
                             int64_t excl_offset, int zero_alphas, double* colsum, double* total,
                             hipStream_t st) {
    if (total) LVAE_HIP(hipMemsetAsync(total, 0, sizeof(double), st));
    hipLaunchKernelGGL(ksum_columns_kernel, dim3((unsigned)lvae_ceil_div(C, 256)), dim3(256), 0, st, partial,
                       nsplit, cpad, C, R, excl, excl_offset, (double)zero_alphas, colsum, total);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// Merge of two partial logsumexps over disjoint sets of posteriors (natural-log domain, the (row_max,
// row_sum) pairs of lvae_log_qz_partial_f32):  m = max(ma, mb), s = sa e^(ma - m) + sb e^(mb - m).
// b may be absent (finalisation of a single partial).  The rescaled sums are formed in float64: a sum may be
// as large as N * 2^100 when its reference is the fixed one.
__global__ void __launch_bounds__(256)
lse_combine_kernel(const float* max_a, const float* sum_a, const float* max_b, const float* sum_b, int64_t n,
                   float log_norm, float* out, float* out_max, float* out_sum) {      // (outputs may alias inputs)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float ma = max_a[i];
    const float mb = max_b ? max_b[i] : -INFINITY;
    const float m = fmaxf(ma, mb);
    double s = 0.0;
    if (sum_a[i] > 0.f) s += (double)sum_a[i] * exp((double)ma - (double)m);
    if (max_b && sum_b[i] > 0.f) s += (double)sum_b[i] * exp((double)mb - (double)m);
    // keep the pair in float range: renormalise the sum into [1, 2) * 2^k with the exponent folded into the max
    int e = 0;
    const double fr = frexp(s, &e);                  // s = fr * 2^e, fr in [0.5, 1)
    const double mm = (double)m + (double)(e - 1) * 0.6931471805599453;
    if (out) out[i] = (float)((double)m + log(s) - (double)log_norm);
    if (out_max) {
        out_max[i] = (float)mm;
        out_sum[i] = (float)(fr * 2.0);
    }
}

// (max_a, sum_a) re-expressed against a new, not smaller reference: out_sum = sum_a * exp(max_a - new_max), so that
// the sums of several owners add up (all-reduce) once they share the all-reduced maximum.
__global__ void __launch_bounds__(256)
lse_rebase_kernel(const float* max_a, const float* sum_a, const float* new_max, int64_t n, float* out_sum) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s = sum_a[i];
    out_sum[i] = s > 0.f ? (float)((double)s * exp((double)max_a[i] - (double)new_max[i])) : 0.f;
}

int lvae_launch_lse_rebase(const float* max_a, const float* sum_a, const float* new_max, int64_t n, float* out_sum,
                           hipStream_t st) {
    hipLaunchKernelGGL(lse_rebase_kernel, dim3((unsigned)lvae_ceil_div(n, 256)), dim3(256), 0, st, max_a, sum_a,
                       new_max, n, out_sum);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_lse_combine(const float* max_a, const float* sum_a, const float* max_b, const float* sum_b,
                            int64_t n, float log_norm, float* out, float* out_max, float* out_sum, hipStream_t st) {
    hipLaunchKernelGGL(lse_combine_kernel, dim3((unsigned)lvae_ceil_div(n, 256)), dim3(256), 0, st, max_a, sum_a,
                       max_b, sum_b, n, log_norm, out, out_max, out_sum);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
