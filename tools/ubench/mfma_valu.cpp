// Micro-benchmark: how many VALU / transcendental instructions hide under one v_mfma_f32_32x32x2_f32?
// One wave per SIMD (256 threads per CU block) or two (512), NV plain VALU + NE v_exp per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NE, int CHAINS>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = (float)threadIdx.x * 1e-3f + e + c;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    float x[4] = {a, b, a * 0.5f, b * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i % 8] = v[i % 8] * 1.0001f + 0.5f;
#pragma unroll
                for (int i = 0; i < NE; ++i) x[i % 4] = __builtin_amdgcn_exp2f(x[i % 4]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NE, int CHAINS>
void run(int threads, float* d) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NE, CHAINS>), dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NE, CHAINS>), dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * CHAINS * (threads / 256);
    const double ns_per = ms * 1e6 / mfma_per_simd;
    const double tf = 256.0 * 4 * mfma_per_simd * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD=%d chains=%d NV=%2d NE=%d : %.2f ns per MFMA per SIMD (%.0f cyc @2.4GHz)  %.1f TFLOP/s\n",
           threads / 256, CHAINS, NV, NE, ns_per, ns_per * 2.4, tf);
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int t : {256, 512}) {
        run<0, 0, 2>(t, d); run<2, 0, 2>(t, d); run<4, 0, 2>(t, d); run<6, 0, 2>(t, d); run<8, 0, 2>(t, d);
        run<10, 0, 2>(t, d); run<12, 0, 2>(t, d); run<16, 0, 2>(t, d);
        run<0, 1, 2>(t, d); run<0, 2, 2>(t, d); run<0, 4, 2>(t, d); run<2, 1, 2>(t, d); run<4, 2, 2>(t, d);
        run<6, 2, 2>(t, d); run<0, 0, 1>(t, d); run<4, 0, 1>(t, d);
    }
    return 0;
}
