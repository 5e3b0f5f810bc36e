// Per-sample Gaussian log-densities, the reparameterised draw and the importance-weighted
// log-likelihood reduction.  All HBM-bound streaming passes over [B,K,D] (eps and/or z) with
// 16-byte-per-lane coalesced accesses: G = D/4 adjacent lanes share one sample row and reduce the
// latent dimension with wave shuffles; the K axis is folded into a per-thread online logsumexp that is
// merged across the workgroup through LDS.
//
//   z          = mu + eps * exp(0.5*logvar)                      modules/encoder.py:148-157
//   log_q_z_x  = sum_d -0.5*(logvar + log 2pi + (z-mu)^2/var)    loss_and_optimisation.py:93-96,105
//   log_p_z    = sum_d -0.5*(log 2pi + z^2)                      loss_and_optimisation.py:99-101,105
//   iw_ll[b]   = LSE_k(log_p_x_z + log_p_z - log_q_z_x) - log K  loss_and_optimisation.py:783-794
//
// Algorithmic bytes per (sentence, sample): 4*D (eps) + 4 (log_p_x_z) read, + 4*D when z is written.
#include "pair_plan.h"

// Partner value for a SUM over 2*O adjacent lanes, without the LDS crossbar where the data-parallel-primitive
// modifiers reach: O = 1, 2 quad permutations (lane ^ 1, lane ^ 2), O = 4 the half-row mirror (lane 7 - i of every
// eight: not lane ^ 4, but it also pairs the lower four lanes with the upper four one to one, which is all a
// sum -- or a reduce-scatter keyed on bit 2 -- needs), O = 8 a rotation by eight inside the 16-lane row
// (= lane ^ 8).  Wider distances go through ds_bpermute.  All lanes of the wave must be active.
template <int O>
__device__ __forceinline__ float sum_partner(float v) {
    if constexpr (O == 1 || O == 2 || O == 4 || O == 8) {
        constexpr int ctrl = O == 1 ? 0xB1 : (O == 2 ? 0x4E : (O == 4 ? 0x141 : 0x128));
        const int x = __float_as_int(v);
        return __int_as_float(__builtin_amdgcn_mov_dpp(x, ctrl, 0xF, 0xF, true));
    } else {
        return __shfl_xor(v, O, 64);
    }
}

template <int O>
__device__ __forceinline__ void group_sum_level(float& v, int G) {
    if (O < G) v += sum_partner<O>(v);
}

__device__ __forceinline__ float group_sum(float v, int G) {
    group_sum_level<32>(v, G);
    group_sum_level<16>(v, G);
    group_sum_level<8>(v, G);
    group_sum_level<4>(v, G);
    group_sum_level<2>(v, G);
    group_sum_level<1>(v, G);
    return v;
}

__device__ __forceinline__ void lse_push(float& m, float& s, float w) {
    if (w > m) {
        s = s * __expf(m - w) + 1.0f;
        m = w;
    } else {
        s += __expf(w - m);
    }
}

// FROM_EPS: the row source is eps (z is computed); otherwise the row source is z itself.
// One workgroup per sentence b; G = D/4 lanes per sample row (power of two <= 64).  Every lane group
// keeps U consecutive rows in flight (U independent 16-B loads per lane); after the shuffle reduction
// lane u of the group owns row u's densities, so the [B,K] outputs and log_p_x_z move as runs of
// consecutive floats (U per group, 64*U/G per wave-instruction) instead of one float per group.
// Sum of U per-row partials over the G lanes of a row group, leaving row (gl % U) in lane gl: levels wider than U
// are plain butterflies, the last log2(U) levels a reduce-scatter (each level halves what a lane still carries):
// G = U = 8 takes 7 shuffles per quantity instead of 24.
template <int U, int O>
__device__ __forceinline__ void scatter_level(float (&a)[U], int gl) {      // a lane keeps the half of its rows that bit O selects
    if constexpr (O >= 1) {
        const bool up = (gl & O) != 0;
#pragma unroll
        for (int u = 0; u < O; ++u) {
            const float send = up ? a[u] : a[u + O];
            const float keep = up ? a[u + O] : a[u];
            a[u] = keep + sum_partner<O>(send);
        }
        scatter_level<U, O / 2>(a, gl);
    }
}

template <int U>
__device__ __forceinline__ float reduce_rows(float (&a)[U], int gl, int G) {
    for (int o = G >> 1; o >= U; o >>= 1) {       // plain butterflies: every lane keeps all U rows
        if (o == 8) {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] += sum_partner<8>(a[u]);
        } else if (o == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] += sum_partner<4>(a[u]);
        } else if (o == 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] += sum_partner<2>(a[u]);
        } else if (o == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] += sum_partner<1>(a[u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] += __shfl_xor(a[u], o, 64);
        }
    }
    scatter_level<U, U / 2>(a, gl);
    return a[0];
}

// z = mu + eps * sigma with the product and the sum rounded separately, like the reference's torch.mul / + (no fma)
__device__ __forceinline__ f32x2 mul_then_add(f32x2 m, f32x2 e, f32x2 sd) {
#pragma clang fp contract(off)
    const f32x2 t = e * sd;
    return m + t;
}

// IO (compile time, so that every trip of the stream issues the same memory operations and the compiler's counted
// waits are exact -- with run-time conditions around a load or store it falls back to vmcnt(0), a full drain of the
// loads that were just issued, in every trip): bit 0 = log_p_x_z is read (iw_ll wanted and the tensor given),
// bit 1 = z is written.
template <bool FROM_EPS, int U, int TPB, bool PIPE, int IO>
__global__ void __launch_bounds__(TPB)
sample_iw_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                 const float* __restrict__ src, const float* __restrict__ log_p_x_z, int64_t K, int D,
                 int G, float* __restrict__ z_out, float* __restrict__ log_q_z_x,
                 float* __restrict__ log_p_z, float* __restrict__ iw_ll, int stage) {
    constexpr bool HAS_PXZ = (IO & 1) != 0;
    constexpr bool WRITE_Z = FROM_EPS && (IO & 2) != 0;
    // stage != 0 (K a multiple of 4, 2*K floats of dynamic LDS): the sentence's two [K] output rows are collected in
    // LDS and leave as full 16-byte-per-lane stores at the end, instead of one 128-byte piece per wave and pass
    // (131 MB of such pieces cost the C5 stream 0.06 ms: 0.39 -> 0.33 ms without them); stage & 2: those stores are
    // write-through (sc1), the default: 0.372 against 0.380 ms at C5.  What the outputs still cost is a property of
    // the memory system, not of this kernel (tools/ubench/rw_mix.cpp: a bare read of the same 2.1 GB takes 0.294 ms,
    // 0.377 ms with 8,000 bytes stored after every 128,000 read -- whichever wave stores them, plain, nt or sc1 --
    // and 0.32-0.33 ms only when ALL stores fall together at the end of the kernel, which needs the outputs of a
    // whole launch in LDS: a wave-per-sentence form with one burst per workgroup, and the same as one round of
    // workgroups per launch, were built and measured at 0.377 / 0.370 ms and dropped)
    extern __shared__ __attribute__((aligned(16))) float iw_stage[];
    const int64_t b = blockIdx.x;
    const int gl = threadIdx.x % G;          // lane inside the row group: owns dims 4*gl .. 4*gl+3
    const int rows_per_iter = (TPB / G) * U;
    const bool have_post = mu != nullptr;
    const float* __restrict__ base = src + b * K * D;
    // Row of (group g of the wave, load u) = wave base + u * Gw + g: one wave-instruction reads Gw
    // consecutive rows = 1 KiB of contiguous memory, and the 64 results a wave owns are 64 consecutive
    // floats (lane u of group g owns row u * Gw + g).
    const int Gw = 64 / G;                                // row groups per wave
    const int gw = (threadIdx.x & 63) / G;                // this lane's group inside its wave
    const int wave_rows = Gw * U;
    const int64_t kw0 = (int64_t)(threadIdx.x >> 6) * wave_rows + gw;           // row of load 0 in pass 0
    // (unconditional loads: a row past the end is clamped to the last one -- its results are never owned by a lane
    //  -- so that a pass is straight-line code and the waits on its loads are counted ones; log_p_x_z rides along:
    //  every lane of a row group loads the element of the row that lane gl % U will own)
    const int64_t klast = K - 1;
    auto issue = [&](f32x4 (&v)[U], float& p, int64_t kw) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = kw + u * Gw < klast ? kw + u * Gw : klast;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + row * D + 4 * gl));
        }
        if (HAS_PXZ) {
            const int64_t kr = kw + (int64_t)(gl % U) * Gw;
            p = log_p_x_z[b * K + (kr < klast ? kr : klast)];
        }
    };
    // per-lane constants of the four dimensions, as two packed pairs: mean, sigma, sqrt(0.5 / var); the constant
    // part of the row's log q(z|x), -0.5 (logvar + log 2pi), is summed once.  Their two loads go out FIRST: memory
    // operations complete in order, so they are back before the first pass, whose loads stay in flight while the
    // constants are formed.
    // (unconditional, from an address that is valid either way: a branch around them would put the wait for them at
    //  its end, ahead of the first pass)
    f32x4 m4 = *reinterpret_cast<const f32x4*>((have_post ? mu + b * D : base) + 4 * gl);
    f32x4 lv4 = *reinterpret_cast<const f32x4*>((have_post ? logvar + b * D : base) + 4 * gl);
    // The stream is software-pipelined over two register buffers: the loads of pass p + 1 are in flight while pass
    // p is reduced (7 dependent shuffles per quantity).
    f32x4 va[U], vb[U];
    float pa = 0.f, pb = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    issue(va, pa, kw0);
    __builtin_amdgcn_sched_barrier(0);      // (or the constants' arithmetic, and the wait for their loads, moves ahead of the first pass)
    asm volatile("" : "+v"(m4), "+v"(lv4)); // (both loads stay where they are: left alone, the compiler sinks one into the branch below)
    f32x2 m2[2] = {{0.f, 0.f}, {0.f, 0.f}}, sd2[2] = {{0.f, 0.f}, {0.f, 0.f}}, sh2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float cq = 0.f;
    if (have_post) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m2[e >> 1][e & 1] = m4[e];
            sd2[e >> 1][e & 1] = expf(0.5f * lv4[e]);
            sh2[e >> 1][e & 1] = sqrtf(0.5f / expf(lv4[e]));
            cq += -0.5f * (lv4[e] + LVAE_LOG_2PI);
        }
    }
    // (the means are only renamed above: without this anchor their first use, and with it the wait for their load,
    //  sinks into the loop, where it can only be a vmcnt(0))
    asm volatile("" : "+v"(m2[0]), "+v"(m2[1]));
    const float cp = -0.5f * LVAE_LOG_2PI * 4.0f;
    float lm = -INFINITY, ls = 0.f;
    auto consume = [&](const f32x4 (&v)[U], float pxz, int64_t kw) {
        const int64_t kmine = kw + (int64_t)(gl % U) * Gw;     // the row whose results this lane will own
        const bool mine = gl < U && kmine < K;
        // packed fp32 throughout (two dimensions per instruction): -(z - mu)^2 / (2 var) as the square of
        // (z - mu) * sqrt(0.5 / var), -z^2 / 2 from the plain sum of squares
        float aq[U], ap[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x2 q2 = {0.f, 0.f}, p2 = {0.f, 0.f};
            f32x4 zz4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 x2 = {v[u][2 * h], v[u][2 * h + 1]};
                const f32x2 zz = FROM_EPS ? mul_then_add(m2[h], x2, sd2[h]) : x2;
                if (have_post) {
                    const f32x2 w = (zz - m2[h]) * sh2[h];
                    q2 = __builtin_elementwise_fma(w, w, q2);
                }
                p2 = __builtin_elementwise_fma(zz, zz, p2);
                zz4[2 * h] = zz[0];
                zz4[2 * h + 1] = zz[1];
            }
            if (WRITE_Z) {      // (a row past the end repeats the last row's store: same address, same value)
                const int64_t row = kw + u * Gw < klast ? kw + u * Gw : klast;
                __builtin_nontemporal_store(zz4, reinterpret_cast<f32x4*>(z_out + (b * K + row) * D + 4 * gl));
            }
            aq[u] = cq - (q2[0] + q2[1]);
            ap[u] = cp - 0.5f * (p2[0] + p2[1]);
            __builtin_amdgcn_sched_barrier(0);      // one row at a time: interleaved rows only lengthen live ranges
        }
        const float my_lq = have_post ? reduce_rows<U>(aq, gl, G) : 0.f;
        const float my_lp = reduce_rows<U>(ap, gl, G);
        if (mine) {
            if (stage) {
                if (log_q_z_x) iw_stage[kmine] = my_lq;
                if (log_p_z) iw_stage[K + kmine] = my_lp;
            } else {
                if (log_q_z_x) log_q_z_x[b * K + kmine] = my_lq;
                if (log_p_z) log_p_z[b * K + kmine] = my_lp;
            }
            if (iw_ll) lse_push(lm, ls, pxz + my_lp - my_lq);
        }
    };
    if (PIPE) {
        // every trip of the main loop: two passes consumed, two issued, nothing conditional
        int64_t k0 = 0;
        for (; k0 + 2 * rows_per_iter < K; k0 += 2 * rows_per_iter) {
            issue(vb, pb, kw0 + k0 + rows_per_iter);
            __builtin_amdgcn_sched_barrier(0);      // (keeps the two passes' arithmetic apart: interleaved, their
            consume(va, pa, kw0 + k0);              //  live ranges cost 40 registers and two waves per SIMD)
            __builtin_amdgcn_sched_barrier(0);
            issue(va, pa, kw0 + k0 + 2 * rows_per_iter);
            __builtin_amdgcn_sched_barrier(0);
            consume(vb, pb, kw0 + k0 + rows_per_iter);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the last one or two passes
        const bool second = k0 + rows_per_iter < K;
        if (second) issue(vb, pb, kw0 + k0 + rows_per_iter);
        __builtin_amdgcn_sched_barrier(0);
        consume(va, pa, kw0 + k0);
        __builtin_amdgcn_sched_barrier(0);
        if (second) consume(vb, pb, kw0 + k0 + rows_per_iter);
    } else {
        for (int64_t k0 = 0; k0 < K; k0 += rows_per_iter) {
            if (k0) issue(va, pa, kw0 + k0);
            consume(va, pa, kw0 + k0);
        }
    }
    if (stage) {
        __syncthreads();
        const f32x4* sq = reinterpret_cast<const f32x4*>(iw_stage);
        const f32x4* sp = reinterpret_cast<const f32x4*>(iw_stage + K);
        for (int i = threadIdx.x; i < (int)(K >> 2); i += TPB) {
            if (stage & 2) {
                if (log_q_z_x) {
                    const f32x4 v = sq[i];
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(reinterpret_cast<f32x4*>(log_q_z_x + b * K) + i), "v"(v) : "memory");
                }
                if (log_p_z) {
                    const f32x4 v = sp[i];
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(reinterpret_cast<f32x4*>(log_p_z + b * K) + i), "v"(v) : "memory");
                }
            } else {
                if (log_q_z_x) reinterpret_cast<f32x4*>(log_q_z_x + b * K)[i] = sq[i];
                if (log_p_z) reinterpret_cast<f32x4*>(log_p_z + b * K)[i] = sp[i];
            }
        }
    }
    if (iw_ll) {
        __shared__ float sm_m[TPB], sm_s[TPB];
        sm_m[threadIdx.x] = lm;
        sm_s[threadIdx.x] = ls;
        __syncthreads();
        if (threadIdx.x < 64) {
            float mx = sm_m[threadIdx.x];
#pragma unroll
            for (int t = 1; t < TPB / 64; ++t) mx = fmaxf(mx, sm_m[threadIdx.x + 64 * t]);
            mx = wave_max(mx);
            // (float exponentials: every factor is <= 1 and the per-lane sums are float already; only the
            //  accumulation across the workgroup runs in float64)
            double sacc = 0.0;
#pragma unroll
            for (int t = 0; t < TPB / 64; ++t) {
                const float sv = sm_s[threadIdx.x + 64 * t];
                if (sv > 0.f) sacc += (double)(sv * __expf(sm_m[threadIdx.x + 64 * t] - mx));
            }
            sacc = wave_sum_d(sacc);
            if (threadIdx.x == 0) iw_ll[b] = (float)((double)mx + log(sacc) - log((double)K));
        }
    }
}

// Any D: one lane per sample row, scalar loads.
template <bool FROM_EPS>
__global__ void __launch_bounds__(256)
sample_iw_generic_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                         const float* __restrict__ src, const float* __restrict__ log_p_x_z, int64_t K,
                         int D, float* __restrict__ z_out, float* __restrict__ log_q_z_x,
                         float* __restrict__ log_p_z, float* __restrict__ iw_ll) {
    const int64_t b = blockIdx.x;
    const bool have_post = mu != nullptr;
    float lm = -INFINITY, ls = 0.f;
    for (int64_t k = threadIdx.x; k < K; k += 256) {
        float lq = 0.f, lp = 0.f;
        for (int d = 0; d < D; ++d) {
            const float v = src[(b * K + k) * D + d];
            float zz = v;
            if (have_post) {
                const float mm = mu[b * D + d], lv = logvar[b * D + d];
                if (FROM_EPS) zz = __fadd_rn(mm, __fmul_rn(v, expf(0.5f * lv)));
                const float df = zz - mm;
                lq += -0.5f * (lv + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv)));
            }
            lp += -0.5f * (LVAE_LOG_2PI + zz * zz);
            if (FROM_EPS && z_out) z_out[(b * K + k) * D + d] = zz;
        }
        if (log_q_z_x) log_q_z_x[b * K + k] = lq;
        if (log_p_z) log_p_z[b * K + k] = lp;
        if (iw_ll) lse_push(lm, ls, (log_p_x_z ? log_p_x_z[b * K + k] : 0.f) + lp - lq);
    }
    if (iw_ll) {
        __shared__ float sm_m[256], sm_s[256];
        sm_m[threadIdx.x] = lm;
        sm_s[threadIdx.x] = ls;
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = -INFINITY;
            for (int t = 0; t < 256; ++t) mx = fmaxf(mx, sm_m[t]);
            double s = 0.0;
            for (int t = 0; t < 256; ++t)
                if (sm_s[t] > 0.f) s += (double)sm_s[t] * exp((double)sm_m[t] - (double)mx);
            iw_ll[b] = (float)((double)mx + log(s) - log((double)K));
        }
    }
}

// Latent sizes the vector kernel does not take (D / 4 not a power of two up to 64: 20, 48, 100, 768 ...), up to
// D = 1024: a WAVE per sample row, lanes along the row (coalesced 16-byte loads when D % 4 == 0), the sentence's
// posterior constants in registers, one wave reduction per row; the four waves of a workgroup share a sentence.
// (The one-lane-per-row kernel above walks each row with 4-byte loads a row apart in every lane and evaluates two
//  exponentials and a division per element: 0.57 TB/s at D = 768.)
template <bool FROM_EPS, bool VEC>
__global__ void __launch_bounds__(256)
sample_iw_waverow_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                         const float* __restrict__ src, const float* __restrict__ log_p_x_z, int64_t K,
                         int D, float* __restrict__ z_out, float* __restrict__ log_q_z_x,
                         float* __restrict__ log_p_z, float* __restrict__ iw_ll) {
    constexpr int W = VEC ? 4 : 1;              // elements per lane and step
    constexpr int MAXS = VEC ? 4 : 16;          // steps of 64 lanes: D <= 1024
    const int64_t b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool have_post = mu != nullptr;
    float m[MAXS][W], sd[MAXS][W], var[MAXS][W];
    float cq = 0.f, cp = 0.f;                   // this lane's share of the constants of log q(z|x) and log p(z)
#pragma unroll
    for (int st = 0; st < MAXS; ++st)
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const int d = (st * 64 + lane) * W + e;
            m[st][e] = 0.f;
            sd[st][e] = 1.f;
            var[st][e] = 1.f;
            if (d < D) {
                cp += -0.5f * LVAE_LOG_2PI;
                if (have_post) {
                    const float lv = logvar[b * D + d];
                    m[st][e] = mu[b * D + d];
                    sd[st][e] = expf(0.5f * lv);
                    var[st][e] = expf(lv);
                    cq += -0.5f * (lv + LVAE_LOG_2PI);
                }
            }
        }
    float lm = -INFINITY, ls = 0.f;
    for (int64_t k = wave; k < K; k += 4) {
        const float* __restrict__ row = src + (b * K + k) * D;
        float* __restrict__ zrow = (FROM_EPS && z_out) ? z_out + (b * K + k) * D : nullptr;
        float q = 0.f, p = 0.f;
#pragma unroll
        for (int st = 0; st < MAXS; ++st) {
            const int d0 = (st * 64 + lane) * W;
            if (d0 < D) {
                float v[W], zz[W];
                if (VEC) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(row + d0);
#pragma unroll
                    for (int e = 0; e < W; ++e) v[e] = t[e];
                } else {
                    v[0] = row[d0];
                }
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    zz[e] = (FROM_EPS && have_post) ? __fadd_rn(m[st][e], __fmul_rn(v[e], sd[st][e])) : v[e];
                    if (have_post) {
                        const float df = zz[e] - m[st][e];
                        q += __fdiv_rn(df * df, var[st][e]);
                    }
                    p = fmaf(zz[e], zz[e], p);
                }
                if (zrow) {
                    if (VEC) {
                        f32x4 t;
#pragma unroll
                        for (int e = 0; e < W; ++e) t[e] = zz[e];
                        *reinterpret_cast<f32x4*>(zrow + d0) = t;
                    } else {
                        zrow[d0] = zz[0];
                    }
                }
            }
        }
        const float lq = have_post ? wave_sum(cq - 0.5f * q) : 0.f;
        const float lp = wave_sum(cp - 0.5f * p);
        if (lane == 0) {
            if (log_q_z_x) log_q_z_x[b * K + k] = lq;
            if (log_p_z) log_p_z[b * K + k] = lp;
        }
        if (iw_ll) lse_push(lm, ls, (log_p_x_z ? log_p_x_z[b * K + k] : 0.f) + lp - lq);
    }
    if (iw_ll) {
        __shared__ float sm_m[4], sm_s[4];
        if (lane == 0) {
            sm_m[wave] = lm;
            sm_s[wave] = ls;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = -INFINITY;
            for (int t = 0; t < 4; ++t) mx = fmaxf(mx, sm_m[t]);
            double s2 = 0.0;
            for (int t = 0; t < 4; ++t)
                if (sm_s[t] > 0.f) s2 += (double)sm_s[t] * exp((double)sm_m[t] - (double)mx);
            iw_ll[b] = (float)((double)mx + log(s2) - log((double)K));
        }
    }
}

// Short rows outside the vector kernel's envelope (D / 4 not a power of two: 20, 36, 48, 100 ...): G lanes per row with
// G the next power of two at or above the row's 16-byte pieces (VEC: D % 4 == 0, D <= 256) or elements (D <= 64), the
// lanes past the end of the row idle (5 of 8 at D = 20, 25 of 32 at D = 100); 256 / G rows per workgroup pass, four
// passes in flight, a butterfly over the G lanes per row.  (The wave-per-row kernel above keeps 5 of 64 lanes busy at
// D = 20: 0.29 TB/s.)
template <bool FROM_EPS, bool VEC>
__global__ void __launch_bounds__(256)
sample_iw_group_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                       const float* __restrict__ src, const float* __restrict__ log_p_x_z, int64_t K,
                       int D, int G, float* __restrict__ z_out, float* __restrict__ log_q_z_x,
                       float* __restrict__ log_p_z, float* __restrict__ iw_ll) {
    constexpr int W = VEC ? 4 : 1;
    constexpr int U = 4;                          // row passes in flight
    const int64_t b = blockIdx.x;
    const int gl = threadIdx.x & (G - 1), rsub = threadIdx.x / G, RP = 256 / G;
    const int d0 = gl * W;
    const bool active = d0 < D;
    const bool have_post = mu != nullptr;
    float m[W], sd[W], var[W];
    float cq = 0.f, cp = 0.f;
#pragma unroll
    for (int e = 0; e < W; ++e) {
        m[e] = 0.f;
        sd[e] = 1.f;
        var[e] = 1.f;
        if (active) {
            cp += -0.5f * LVAE_LOG_2PI;
            if (have_post) {
                const float lv = logvar[b * D + d0 + e];
                m[e] = mu[b * D + d0 + e];
                sd[e] = expf(0.5f * lv);
                var[e] = expf(lv);
                cq += -0.5f * (lv + LVAE_LOG_2PI);
            }
        }
    }
    float lm = -INFINITY, ls = 0.f;
    for (int64_t k0 = 0; k0 < K; k0 += (int64_t)U * RP) {
        float v[U][W];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + u * RP + rsub;
#pragma unroll
            for (int e = 0; e < W; ++e) v[u][e] = 0.f;
            if (active && k < K) {
                const float* __restrict__ row = src + (b * K + k) * D + d0;
                if (VEC) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(row);
#pragma unroll
                    for (int e = 0; e < W; ++e) v[u][e] = t[e];
                } else {
                    v[u][0] = row[0];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + u * RP + rsub;
            const bool ok = k < K;                 // (uniform over the G lanes of a row)
            float zz[W], q = 0.f, p = 0.f;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                zz[e] = (FROM_EPS && have_post) ? __fadd_rn(m[e], __fmul_rn(v[u][e], sd[e])) : v[u][e];
                if (have_post) {
                    const float df = zz[e] - m[e];
                    q += __fdiv_rn(df * df, var[e]);
                }
                p = fmaf(zz[e], zz[e], p);
            }
            if (FROM_EPS && z_out && active && ok) {
                float* __restrict__ zrow = z_out + (b * K + k) * D + d0;
                if (VEC) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < W; ++e) t[e] = zz[e];
                    *reinterpret_cast<f32x4*>(zrow) = t;
                } else {
                    zrow[0] = zz[0];
                }
            }
            float lq = active ? cq - 0.5f * q : 0.f;      // (idle lanes: zz = v = 0, their shares are zero)
            float lp = active ? cp - 0.5f * p : 0.f;
            for (int o = G >> 1; o >= 1; o >>= 1) {
                lq += __shfl_xor(lq, o, 64);
                lp += __shfl_xor(lp, o, 64);
            }
            if (!have_post) lq = 0.f;
            if (gl == 0 && ok) {
                if (log_q_z_x) log_q_z_x[b * K + k] = lq;
                if (log_p_z) log_p_z[b * K + k] = lp;
                if (iw_ll) lse_push(lm, ls, (log_p_x_z ? log_p_x_z[b * K + k] : 0.f) + lp - lq);
            }
        }
    }
    if (iw_ll) {
        // merge the (max, sum) pairs of the row-owning lanes: butterfly over the wave, then the four waves
        for (int o = 32; o >= 1; o >>= 1) {
            const float m2 = __shfl_xor(lm, o, 64), s2 = __shfl_xor(ls, o, 64);
            const float mx = fmaxf(lm, m2);
            ls = (ls > 0.f ? ls * __expf(lm - mx) : 0.f) + (s2 > 0.f ? s2 * __expf(m2 - mx) : 0.f);
            lm = mx;
        }
        __shared__ float sm_m[4], sm_s[4];
        if ((threadIdx.x & 63) == 0) {
            sm_m[threadIdx.x >> 6] = lm;
            sm_s[threadIdx.x >> 6] = ls;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = -INFINITY;
            for (int t = 0; t < 4; ++t) mx = fmaxf(mx, sm_m[t]);
            double s2 = 0.0;
            for (int t = 0; t < 4; ++t)
                if (sm_s[t] > 0.f) s2 += (double)sm_s[t] * exp((double)sm_m[t] - (double)mx);
            iw_ll[b] = (float)((double)mx + log(s2) - log((double)K));
        }
    }
}

// Element-wise (no latent reduction): sample_log_likelihood(..., reduce_latent_dim=False).
__global__ void __launch_bounds__(256)
sample_ll_elem_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                      const float* __restrict__ z, int64_t total, int64_t KD, int D,
                      float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float zz = z[i];
    if (mu) {
        const int64_t b = i / KD;
        const int d = (int)(i % D);
        const float mm = mu[b * D + d], lv = logvar[b * D + d];
        const float df = zz - mm;
        out[i] = -0.5f * (lv + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv)));
    } else {
        out[i] = -0.5f * (LVAE_LOG_2PI + zz * zz);
    }
}

// logsumexp over the K axis of (a + b - c), minus log K:  [B,K] x 3 -> [B].  One wave per row.
__global__ void __launch_bounds__(256)
iw_lse_kernel(const float* __restrict__ log_p_x_z, const float* __restrict__ log_p_z,
              const float* __restrict__ log_q_z_x, int64_t B, int64_t K, float* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    float lm = -INFINITY, ls = 0.f;
    for (int64_t k = lane; k < K; k += 64) {
        const float w = log_p_x_z[b * K + k] + log_p_z[b * K + k] - log_q_z_x[b * K + k];
        lse_push(lm, ls, w);
    }
    const float mx = wave_max(lm);
    double s = ls > 0.f ? (double)ls * exp((double)lm - (double)mx) : 0.0;
    s = wave_sum_d(s);
    if (lane == 0) out[b] = (float)((double)mx + log(s) - log((double)K));
}

// Densities only (no per-sentence reduction): rows = B*K flattened, G lanes per row, grid-stride.
__global__ void __launch_bounds__(256)
sample_ll_rows_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                      const float* __restrict__ z, int64_t rows, int64_t K, int D, int G,
                      float* __restrict__ log_q_z_x, float* __restrict__ log_p_z) {
    const int gl = threadIdx.x % G;
    const int64_t rows_per_pass = (int64_t)gridDim.x * (256 / G);
    const bool have_post = mu != nullptr;
    for (int64_t base = (int64_t)blockIdx.x * (256 / G); base < rows; base += rows_per_pass) {
        const int64_t row = base + threadIdx.x / G;
        const bool ok = row < rows;
        float lq = 0.f, lp = 0.f;
        if (ok) {
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + row * D + 4 * gl);
            f32x4 mm = {0.f, 0.f, 0.f, 0.f}, lv = mm;
            if (have_post) {
                const int64_t b = row / K;
                mm = *reinterpret_cast<const f32x4*>(mu + b * D + 4 * gl);
                lv = *reinterpret_cast<const f32x4*>(logvar + b * D + 4 * gl);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (have_post) {
                    const float df = zz[e] - mm[e];
                    lq += -0.5f * (lv[e] + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv[e])));
                }
                lp += -0.5f * (LVAE_LOG_2PI + zz[e] * zz[e]);
            }
        }
        lq = group_sum(lq, G);
        lp = group_sum(lp, G);
        if (ok && gl == 0) {
            if (log_q_z_x) log_q_z_x[row] = lq;
            if (log_p_z) log_p_z[row] = lp;
        }
    }
}

static bool vec_ok(int D, int* G) {
    if (D % 4 != 0) return false;
    const int g = D / 4;
    if (g > 64 || (g & (g - 1)) != 0) return false;
    *G = g;
    return true;
}

int lvae_launch_sample_iw(bool from_eps, const float* mu, const float* logvar, const float* src,
                          const float* log_p_x_z, int64_t B, int64_t K, int D, float* z_out,
                          float* log_q_z_x, float* log_p_z, float* iw_ll, hipStream_t st) {
    int G = 0;
    const dim3 grid((unsigned)B);
    if (!from_eps && !iw_ll && vec_ok(D, &G)) {
        const int64_t rows = B * K;
        int64_t blocks = lvae_ceil_div(rows, 256 / G);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(sample_ll_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mu, logvar, src,
                           rows, K, D, G, log_q_z_x, log_p_z);
        LVAE_LAUNCH_CHECK();
        return LVAE_OK;
    }
    if (vec_ok(D, &G)) {
        // rows per pass of a workgroup = (threads / G) * U: narrower workgroups when a sentence has few samples
        // (K = 100 at 256 threads leaves 61 % of the lanes idle: 0.35 -> 0.21 ms at B = 65,536); wider ones than
        // 256 only lose (measured at C5: 0.46 / 0.52 / 0.63 ms at 256 / 512 / 1024 threads)
        // U = loads in flight per lane in one pass, software-pipelined over two register buffers.  Measured at C5
        // (tools/bench_iw_variants.py; LVAE_IW_PIPE / LVAE_IW_U override for such sweeps): when z is written, U = 8
        // (0.78 ms, 5.6 TB/s; U = 4: 0.81); the read-only reduction prefers U = 4 -- fewer registers, more waves,
        // 16 KB passes -- at 0.405-0.41 ms = 5.6 TB/s (U = 2: 0.43; the un-pipelined U = 8 kernel of round 1: 0.44).
        // (tools/sweep_iw.py, after the row reduction moved from ds_bpermute to data-parallel-primitive adds)
        int umax = G < 8 ? G : 8;
        if (!(from_eps && z_out) && umax > 4) umax = 4;
        bool pipe = true;
        const LvaeSwitches& sw = lvae_switches();
        if (sw.iw_pipe >= 0) pipe = sw.iw_pipe != 0;
        if (sw.iw_u > 0 && sw.iw_u < umax) umax = sw.iw_u;
        int tpb = K <= (64 / G) * umax ? 64 : (K <= (128 / G) * umax ? 128 : 256);
        const bool read_only = !(from_eps && z_out);
        // Read-only forms: the widest workgroup that still makes four passes over the sentence (a short sentence on
        // a wide workgroup is one load-reduce-merge latency chain per workgroup: B = 65,536 x K = 100, iw_ll only,
        // 0.232 ms at 256 threads, 0.178 at 128, 0.149 at 64; K = 200: 0.167 / 0.145 / 0.141; K = 400: all 0.141-0.145;
        // tools/sweep_iw_geometry.py).  When z is written the wider workgroups win (K = 200: 0.29 against 0.32).
        if (read_only) tpb = K >= (int64_t)4 * (256 / G) * umax ? 256 : (K >= (int64_t)4 * (128 / G) * umax ? 128 : 64);
        if (sw.iw_tpb) tpb = sw.iw_tpb;      // (sweeps: 64, 128 or 256)
        // few passes: the single-buffer form needs fewer registers (more waves)
        if (K <= (int64_t)(tpb / G) * umax * (read_only ? 4 : 2) && sw.iw_pipe < 0) pipe = false;
        // [B,K] outputs through LDS (see the kernel): K a multiple of 4 from 256 (K = 100 measured 2 % slower) up to 2,048
        // (16 KB), 16-byte aligned rows; LVAE_IW_NO_STAGE=1 for comparisons
        const bool al16 = (((uintptr_t)log_q_z_x | (uintptr_t)log_p_z) & 15) == 0;
        const int stage = ((log_q_z_x || log_p_z) && K % 4 == 0 && K >= 256 && K <= 2048 && al16 &&
                           !sw.iw_no_stage) ? (sw.iw_plain_stores ? 1 : 3) : 0;
        const size_t dyn = stage ? (size_t)2 * K * sizeof(float) : 0;
        const int io = ((iw_ll && log_p_x_z) ? 1 : 0) | ((from_eps && z_out) ? 2 : 0);
#define LVAE_SIW3(E, UU, T, P, I)                                                                             \
    hipLaunchKernelGGL((sample_iw_kernel<E, UU, T, P, I>), grid, dim3(T), dyn, st, mu, logvar, src, log_p_x_z, K, D, G, \
                       z_out, log_q_z_x, log_p_z, iw_ll, stage)
#define LVAE_SIW2(E, UU, T, P)                                                     \
    do {                                                                           \
        if (io == 0) LVAE_SIW3(E, UU, T, P, 0);                                    \
        else if (io == 1) LVAE_SIW3(E, UU, T, P, 1);                               \
        else if (io == 2) LVAE_SIW3(E, UU, T, P, (E ? 2 : 0));                     \
        else LVAE_SIW3(E, UU, T, P, (E ? 3 : 1));                                  \
    } while (0)
#define LVAE_SIW(E, UU)                                                                             \
    do {                                                                                            \
        if (tpb == 64) { if (pipe) LVAE_SIW2(E, UU, 64, true); else LVAE_SIW2(E, UU, 64, false); }        \
        else if (tpb == 128) { if (pipe) LVAE_SIW2(E, UU, 128, true); else LVAE_SIW2(E, UU, 128, false); } \
        else { if (pipe) LVAE_SIW2(E, UU, 256, true); else LVAE_SIW2(E, UU, 256, false); }                \
    } while (0)
        const int U = umax;
        if (from_eps) {
            if (U == 8) LVAE_SIW(true, 8); else if (U == 4) LVAE_SIW(true, 4);
            else if (U == 2) LVAE_SIW(true, 2); else LVAE_SIW(true, 1);
        } else {
            if (U == 8) LVAE_SIW(false, 8); else if (U == 4) LVAE_SIW(false, 4);
            else if (U == 2) LVAE_SIW(false, 2); else LVAE_SIW(false, 1);
        }
#undef LVAE_SIW3
#undef LVAE_SIW2
#undef LVAE_SIW
    } else if (D <= 1024 && !lvae_switches().iw_no_stage) {
        // (D % 4 == 0 makes every row 16-byte aligned: the tensors are whole allocations or 16-byte aligned views)
        const bool vec = D % 4 == 0 && (((uintptr_t)src | (uintptr_t)z_out) & 15) == 0;
        const int pieces = vec ? D / 4 : D;
        if (pieces <= 64) {                       // short rows: G lanes per row, G = next power of two
            int Gp = 1;
            while (Gp < pieces) Gp <<= 1;
#define LVAE_SIWG(E, V)                                                                                            \
    hipLaunchKernelGGL((sample_iw_group_kernel<E, V>), grid, dim3(256), 0, st, mu, logvar, src, log_p_x_z, K, D, Gp,   \
                       z_out, log_q_z_x, log_p_z, iw_ll)
            if (from_eps) { if (vec) LVAE_SIWG(true, true); else LVAE_SIWG(true, false); }
            else { if (vec) LVAE_SIWG(false, true); else LVAE_SIWG(false, false); }
#undef LVAE_SIWG
            LVAE_LAUNCH_CHECK();
            return LVAE_OK;
        }
#define LVAE_SIWR(E, V)                                                                                              \
    hipLaunchKernelGGL((sample_iw_waverow_kernel<E, V>), grid, dim3(256), 0, st, mu, logvar, src, log_p_x_z, K, D, z_out, \
                       log_q_z_x, log_p_z, iw_ll)
        if (from_eps) { if (vec) LVAE_SIWR(true, true); else LVAE_SIWR(true, false); }
        else { if (vec) LVAE_SIWR(false, true); else LVAE_SIWR(false, false); }
#undef LVAE_SIWR
    } else {
        if (from_eps)
            hipLaunchKernelGGL(sample_iw_generic_kernel<true>, grid, dim3(256), 0, st, mu, logvar, src,
                               log_p_x_z, K, D, z_out, log_q_z_x, log_p_z, iw_ll);
        else
            hipLaunchKernelGGL(sample_iw_generic_kernel<false>, grid, dim3(256), 0, st, mu, logvar, src,
                               log_p_x_z, K, D, z_out, log_q_z_x, log_p_z, iw_ll);
    }
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_sample_ll_elem(const float* mu, const float* logvar, const float* z, int64_t B,
                               int64_t K, int D, float* out, hipStream_t st) {
    const int64_t total = B * K * D;
    hipLaunchKernelGGL(sample_ll_elem_kernel, dim3((unsigned)lvae_ceil_div(total, 256)), dim3(256), 0,
                       st, mu, logvar, z, total, K * D, D, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

int lvae_launch_iw_lse(const float* a, const float* b, const float* c, int64_t B, int64_t K, float* out,
                       hipStream_t st) {
    hipLaunchKernelGGL(iw_lse_kernel, dim3((unsigned)lvae_ceil_div(B, 4)), dim3(256), 0, st, a, b, c, B,
                       K, out);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

// ---- fused per-sample densities + float64 sums ------------------------------------------------------
// For rows i < M (z_i drawn from posterior i): log p(z_i) and log q(z_i|x_i) are evaluated on the fly
// and summed in float64 together with the given per-sample log q(z_i) (and prod-marginals) vectors:
//   sums = [ sum log q(z), sum log prod-marg, sum log p(z), sum log q(z|x), M ]
// These are the four means that loss_and_optimisation.py:816-824 turns into MI / TC / dim-KL /
// marginal-KL.  One pass over z, mu, logvar (12*D bytes per row): HBM-bound.
__global__ void __launch_bounds__(256)
latent_sums_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                   const float* __restrict__ logvar, const float* __restrict__ log_qz,
                   const float* __restrict__ log_qz_pm, int64_t M, int D, int G,
                   double* __restrict__ sums, const LateMerge lm) {
    double a_lq = 0.0, a_pm = 0.0, a_lp = 0.0, a_lqx = 0.0;
    // late merge (G > 0 only, see LateMerge): which rows take which partials
    const bool lm_b = lm.enabled && lm.partial_b && lm.cond_limit >= 0.f &&
                      __uint_as_float(lm.hdr->cond_bits) > lm.cond_limit;
    auto merged_row = [&](int64_t i) -> float {
        const float2* __restrict__ partial = lm_b ? lm.partial_b : lm.partial_a;
        const int nsplit = lm_b ? lm.nsplit_b : lm.nsplit_a;
        float mx = -INFINITY;
        for (int k = 0; k < nsplit; ++k) mx = fmaxf(mx, partial[k * lm.cpad + i].x);
        double s = 0.0;
        for (int k = 0; k < nsplit; ++k) {
            const float2 p = partial[k * lm.cpad + i];
            if (p.y > 0.f) s += (double)p.y * (double)(p.x == mx ? 1.0f : lvae_exp2(p.x - mx));
        }
        const double cb = (!lm_b && lm.use_ref_a) ? (double)lm.hdr->lse_ref : 0.0;
        return (float)(((double)mx + log2(s) + cb) * LVAE_LN2 - (double)lm.log_norm);
    };
    if (G > 0) {
        const int gl = threadIdx.x % G;
        const int64_t rows_per_pass = (int64_t)gridDim.x * (256 / G);
        for (int64_t base = (int64_t)blockIdx.x * (256 / G); base < M; base += rows_per_pass) {
            const int64_t row = base + threadIdx.x / G;
            const bool ok = row < M;
            float lq = 0.f, lp = 0.f;
            if (ok && sums) {       // (sums == NULL: the late merge alone, lvae_log_qz_f32)
                const f32x4 zz = *reinterpret_cast<const f32x4*>(z + row * D + 4 * gl);
                const f32x4 mm = *reinterpret_cast<const f32x4*>(mu + row * D + 4 * gl);
                const f32x4 lv = *reinterpret_cast<const f32x4*>(logvar + row * D + 4 * gl);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float df = zz[e] - mm[e];
                    lq += -0.5f * (lv[e] + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv[e])));
                    lp += -0.5f * (LVAE_LOG_2PI + zz[e] * zz[e]);
                }
            }
            lq = group_sum(lq, G);
            lp = group_sum(lp, G);
            if (ok && gl == 0) {
                a_lqx += (double)lq;
                a_lp += (double)lp;
                if (lm.enabled && (lm_b || lm.flags[row / lm.group_cols] != 0)) {
                    const float v = merged_row(row);
                    lm.out[row] = v;
                    a_lq += (double)v;
                } else if (log_qz) {
                    a_lq += (double)log_qz[row];
                }
                if (log_qz_pm) a_pm += (double)log_qz_pm[row];
            }
        }
    } else {
        for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < M; row += (int64_t)gridDim.x * 256) {
            float lq = 0.f, lp = 0.f;
            for (int d = 0; d < D; ++d) {
                const float zz = z[row * D + d], mm = mu[row * D + d], lv = logvar[row * D + d];
                const float df = zz - mm;
                lq += -0.5f * (lv + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv)));
                lp += -0.5f * (LVAE_LOG_2PI + zz * zz);
            }
            a_lqx += (double)lq;
            a_lp += (double)lp;
            if (log_qz) a_lq += (double)log_qz[row];
            if (log_qz_pm) a_pm += (double)log_qz_pm[row];
        }
    }
    if (!sums) return;
    __shared__ double sm[4][4];
    a_lq = wave_sum_d(a_lq);
    a_pm = wave_sum_d(a_pm);
    a_lp = wave_sum_d(a_lp);
    a_lqx = wave_sum_d(a_lqx);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[0][w] = a_lq;
        sm[1][w] = a_pm;
        sm[2][w] = a_lp;
        sm[3][w] = a_lqx;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double t = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
        atomicAdd(&sums[threadIdx.x], t);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[4] = (double)M;
}

// Small problems: the merge of the split partials of whichever pair kernel ran -- MFMA (partial_a) or direct form
// (partial_b); the device-side conditioning gate picks -- into log_qz, and optionally the float64 sums of
// latent_sums_kernel in the same launch (one launch instead of six).  A row is served by G = D/4 consecutive lanes:
// they share the row's split partials (online (max, sum) per lane, merged by shuffles) and its D-vector (one 16-byte
// load each).  Partials are base-2 (max, sum) pairs without a fixed reference (running-maximum epilogue).  `sums`
// must have been zeroed earlier on the stream (colsum_kernel does it): float64 atomics, as in latent_sums_kernel.
__global__ void __launch_bounds__(256)
finalize_small_kernel(const float2* __restrict__ partial_a, int nsplit_a, const float2* __restrict__ partial_b,
                      int nsplit_b, int64_t cpad, int64_t C, float log_norm, float* __restrict__ out,
                      float* __restrict__ out_max, float* __restrict__ out_sum,
                      const PairHeader* __restrict__ hdr, float cond_limit, const float* __restrict__ z,
                      const float* __restrict__ mu, const float* __restrict__ logvar, int D, int G, int passes,
                      double* __restrict__ sums) {
    const bool use_b = partial_a == nullptr || (cond_limit >= 0.f && __uint_as_float(hdr->cond_bits) > cond_limit);
    const float2* __restrict__ partial = use_b ? partial_b : partial_a;
    const int nsplit = use_b ? nsplit_b : nsplit_a;
    const int gl = threadIdx.x % G;
    const int rpp = 256 / G;                        // rows per pass
    double a_lq = 0.0, a_lp = 0.0, a_lqx = 0.0;
    // (a workgroup takes `passes` groups of rows: few workgroups keep the float64 atomics on the three sums short)
    for (int ps = 0; ps < passes; ++ps) {
        const int64_t row = ((int64_t)blockIdx.x * passes + ps) * rpp + threadIdx.x / G;
        const bool ok = row < C;
        float m = -INFINITY;
        double sacc = 0.0;
        if (ok) {
#pragma unroll 4
            for (int k = gl; k < nsplit; k += G) {
                const float2 pp = partial[k * cpad + row];
                if (pp.y > 0.f) {
                    if (pp.x > m) {
                        sacc = sacc * (double)lvae_exp2(m - pp.x) + (double)pp.y;      // exp2(-inf) = 0 the first time
                        m = pp.x;
                    } else {
                        sacc += (double)pp.y * (double)lvae_exp2(pp.x - m);
                    }
                }
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) {
            const float mo = __shfl_xor(m, o, 64);
            const double so = __shfl_xor(sacc, o, 64);
            const float mm = fmaxf(m, mo);
            sacc = (sacc > 0.0 ? sacc * (double)lvae_exp2(m - mm) : 0.0) + (so > 0.0 ? so * (double)lvae_exp2(mo - mm) : 0.0);
            m = mm;
        }
        // sacc = f * 2^e with f in [0.5, 1): the float32 logarithm of f carries the full float32 result
        int ex = 0;
        const double fr = frexp(sacc, &ex);
        const float lqz = sacc > 0.0 ? (float)(((double)m + (double)ex + (double)__log2f((float)fr)) * LVAE_LN2 - (double)log_norm)
                                     : -INFINITY;
        if (ok && gl == 0) {
            if (out) out[row] = lqz;
            if (out_max) {
                out_max[row] = (float)((double)m * LVAE_LN2);
                out_sum[row] = (float)sacc;
            }
        }
        if (!sums) continue;
        float lq = 0.f, lp = 0.f;
        if (ok) {
            if (G > 1 || D == 4) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(z + row * D + 4 * gl);
                const f32x4 mm = *reinterpret_cast<const f32x4*>(mu + row * D + 4 * gl);
                const f32x4 lv = *reinterpret_cast<const f32x4*>(logvar + row * D + 4 * gl);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float df = zz[e] - mm[e];
                    lq += -0.5f * (lv[e] + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv[e])));
                    lp += -0.5f * (LVAE_LOG_2PI + zz[e] * zz[e]);
                }
            } else {
                for (int d = 0; d < D; ++d) {
                    const float zz = z[row * D + d], mm = mu[row * D + d], lv = logvar[row * D + d];
                    const float df = zz - mm;
                    lq += -0.5f * (lv + LVAE_LOG_2PI + __fdiv_rn(df * df, expf(lv)));
                    lp += -0.5f * (LVAE_LOG_2PI + zz * zz);
                }
            }
        }
        lq = group_sum(lq, G);
        lp = group_sum(lp, G);
        if (ok && gl == 0) {
            a_lq += (double)lqz;
            a_lp += (double)lp;
            a_lqx += (double)lq;
        }
    }
    if (!sums) return;
    __shared__ double sm[3][4];
    a_lq = wave_sum_d(a_lq);
    a_lp = wave_sum_d(a_lp);
    a_lqx = wave_sum_d(a_lqx);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[0][w] = a_lq;
        sm[1][w] = a_lp;
        sm[2][w] = a_lqx;
    }
    __syncthreads();
    if (threadIdx.x < 3)      // {sum log_qz, (prod-marginals: 0), sum log p(z), sum log q(z|x), M}
        atomicAdd(&sums[threadIdx.x == 0 ? 0 : threadIdx.x + 1],
                  sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3]);
    if (blockIdx.x == 0 && threadIdx.x == 3) sums[4] = (double)C;
}

int lvae_launch_finalize_small(const float2* partial_a, int nsplit_a, const float2* partial_b, int nsplit_b,
                               int64_t cpad, int64_t C, float log_norm, float* out, float* out_max, float* out_sum,
                               const PairHeader* hdr, float cond_limit, const float* z, const float* mu,
                               const float* logvar, int D, double* sums, hipStream_t st) {
    int G = 1;
    if (!vec_ok(D, &G)) G = 1;
    const int64_t groups = lvae_ceil_div(C, 256 / G);
    const int passes = 1;      // (more rows per workgroup were measured slower: the passes serialise on their loads)
    hipLaunchKernelGGL(finalize_small_kernel, dim3((unsigned)lvae_ceil_div(groups, passes)), dim3(256), 0, st, partial_a,
                       nsplit_a, partial_b, nsplit_b, cpad, C, log_norm, out, out_max, out_sum, hdr, cond_limit, z, mu,
                       logvar, D, G, passes, sums);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}

bool lvae_latent_sums_vec(int D) {
    int G = 0;
    return vec_ok(D, &G);
}

int lvae_launch_latent_sums(const float* z, const float* mu, const float* logvar, const float* log_qz,
                            const float* log_qz_pm, int64_t M, int D, double* sums, hipStream_t st,
                            bool sums_are_zero, const LateMerge* late) {
    int G = 0;
    if (!vec_ok(D, &G)) G = 0;
    LateMerge lm{};
    if (late) {
        if (!G) return lvae_set_error(LVAE_ERR_INVALID, "late merge needs the vector layout (D a multiple of 4)");
        lm = *late;
        lm.enabled = 1;
    }
    const int64_t rows_per_block = G ? 256 / G : 256;
    int64_t blocks = lvae_ceil_div(M, rows_per_block);
    if (blocks > 512) blocks = 512;      // four float64 atomics per block land on the same four words (C2: 17.4 us at 1,024 blocks, 12.1 at 512, 12.7 at 256)
    // (sums_are_zero: an earlier kernel of the same call cleared them -- no fill packet ahead of this launch)
    if (!sums && !late) return lvae_set_error(LVAE_ERR_INVALID, "latent sums: nothing to do");
    if (sums && !sums_are_zero) LVAE_HIP(hipMemsetAsync(sums, 0, 5 * sizeof(double), st));
    hipLaunchKernelGGL(latent_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, z, mu, logvar, log_qz,
                       log_qz_pm, M, D, G, sums, lm);
    LVAE_LAUNCH_CHECK();
    return LVAE_OK;
}
