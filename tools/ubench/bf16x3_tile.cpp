// Micro-benchmark for the split-bf16 idea: per 32x32 tile of pairs at D = 32, six limb
// products x (2D / 16) k-steps = 24 v_mfma_f32_32x32x16_bf16, with the tile's logsumexp epilogue (16 v_exp_f32,
// 8 v_pk_add_f32) issued between them -- against the fp32 form: 32 v_mfma_f32_32x32x2_f32 with the same
// epilogue.  Two accumulator chains per wave, one or two waves per SIMD.  Reports cycles per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF16, bool EPI>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    f32x16 acc[2], prev[2];
    for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) { acc[c][e] = 0.f; prev[c][e] = -(float)(threadIdx.x % 7) - e; }
    bf16x8 a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(0.001f * (threadIdx.x + e)); b8[e] = (__bf16)(0.002f * (e + 1)); }
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-4f;
    f32x2 s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    constexpr int NM = BF16 ? 24 : 32;       // MFMAs per tile
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NM; ++u) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
                if (EPI) {
                    // 8 slices per tile spread over the tile's NM * 2 MFMA slots
                    const int slot = u * 2 + c;
                    if (slot % (NM * 2 / 16) == 0) {
                        const int sl = slot / (NM * 2 / 16);          // 0..15: chain sl / 8, slice sl % 8
                        const int cc = sl / 8, ms = sl % 8;
                        f32x2 e;
                        e.x = __builtin_amdgcn_exp2f(prev[cc][2 * ms]);
                        e.y = __builtin_amdgcn_exp2f(prev[cc][2 * ms + 1]);
                        s2[cc] += e;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) prev[c][e] = acc[c][e] * 1e-30f - 3.f;
    }
    float s = s2[0].x + s2[0].y + s2[1].x + s2[1].y;
    for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool BF16, bool EPI>
void run(int threads, float* d) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BF16, EPI>), dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BF16, EPI>), dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = (double)iters * 2 * (threads / 256);
    printf("%s %s waves/SIMD=%d : %.0f cycles per 32x32 tile (D=32) @2.4GHz\n", BF16 ? "bf16x3 (24 MFMA)" : "fp32   (32 MFMA)",
           EPI ? "+epilogue" : "bare     ", threads / 256, ms * 1e-3 * 2.4e9 / tiles_per_simd);
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int t : {256, 512}) {
        run<false, false>(t, d); run<false, true>(t, d); run<true, false>(t, d); run<true, true>(t, d);
    }
    return 0;
}
