// Micro-benchmark: what a streaming READ of 2 GiB reaches with the access shapes the IW reduction can use.
//   chunk:  one workgroup per contiguous 128 KB chunk (a sentence's K x D block), U 16-byte loads in flight per lane
//   pers:   a persistent grid (W workgroups per CU) walking chunks with stride gridDim
//   pipe:   chunk shape with the next U loads issued before the current ones are consumed
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/read_bw.cpp -o tools/ubench/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__device__ __forceinline__ void load_u(const f32x4* p, int stride, f32x4 (&v)[U], int left) {   // left: float4s up to the chunk end
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u * stride < left) v[u] = NT ? __builtin_nontemporal_load(p + u * stride) : p[u * stride];
    }
}
template <int U>
__device__ __forceinline__ float fold(const f32x4 (&v)[U]) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    return s;
}

// chunk_f4 = float4s per chunk; TPB threads; each iteration covers TPB * U float4s
template <int U, int TPB, bool NT, bool PIPE, bool PERS>
__global__ void __launch_bounds__(TPB) rd(const f32x4* __restrict__ x, long nchunks, int chunk_f4, float* out) {
    float acc = 0.f;
    for (long c = blockIdx.x; c < nchunks; c += PERS ? gridDim.x : nchunks) {
        const f32x4* base = x + c * chunk_f4 + threadIdx.x;
        const int iters = (chunk_f4 + TPB * U - 1) / (TPB * U);
        const int mine = chunk_f4 - (int)threadIdx.x;          // float4s from this lane's first one to the chunk end
        if (PIPE) {
            f32x4 a[U], b[U];
            load_u<U, NT>(base, TPB, a, mine);
            for (int it = 0; it < iters; it += 2) {
                if (it + 1 < iters) load_u<U, NT>(base + (it + 1) * TPB * U, TPB, b, mine - (it + 1) * TPB * U);
                acc += fold<U>(a);
                if (it + 2 < iters) load_u<U, NT>(base + (it + 2) * TPB * U, TPB, a, mine - (it + 2) * TPB * U);
                if (it + 1 < iters) acc += fold<U>(b);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                f32x4 a[U];
                load_u<U, NT>(base + it * TPB * U, TPB, a, mine - it * TPB * U);
                acc += fold<U>(a);
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int U, int TPB, bool NT, bool PIPE, bool PERS>
void run(const char* name, const f32x4* x, long nbytes, int chunk_bytes, int wg_per_cu, float* out) {
    const long nchunks = nbytes / chunk_bytes;
    const int chunk_f4 = chunk_bytes / 16;
    const int grid = PERS ? 256 * wg_per_cu : (int)nchunks;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rd<U, TPB, NT, PIPE, PERS>), dim3(grid), dim3(TPB), 0, 0, x, nchunks, chunk_f4, out);
    float best = 1e9f;
    for (int i = 0; i < 10; ++i) {
        hipEventRecord(a);
        hipLaunchKernelGGL((rd<U, TPB, NT, PIPE, PERS>), dim3(grid), dim3(TPB), 0, 0, x, nchunks, chunk_f4, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-44s U=%2d TPB=%4d chunk=%4d KB grid=%6d: %.3f ms  %.0f GB/s\n", name, U, TPB, chunk_bytes / 1024, grid, best, nbytes / best / 1e6);
    fflush(stdout);
}

int main() {
    const long nbytes = 16384L * 1000 * 128;      // C5: 16,384 sentences x 1,000 samples x 32 floats
    f32x4* x; float* out;
    hipMalloc(&x, nbytes); hipMalloc(&out, 64);
    {   // non-trivial contents
        float* h = (float*)malloc(1 << 26);
        for (long i = 0; i < (1 << 24); ++i) h[i] = (float)((i * 2654435761u) >> 8) * 1e-7f;
        for (long off = 0; off < nbytes; off += (1 << 26)) hipMemcpy((char*)x + off, h, nbytes - off < (1 << 26) ? nbytes - off : (1 << 26), hipMemcpyHostToDevice);
        free(h);
    }
    const int CH = 128000;                         // one sentence
    run<8, 256, true, false, false>("chunk nt (the IW kernel's shape)", x, nbytes, CH, 0, out);
    run<8, 256, false, false, false>("chunk plain", x, nbytes, CH, 0, out);
    run<4, 256, true, false, false>("chunk nt", x, nbytes, CH, 0, out);
    run<16, 256, true, false, false>("chunk nt", x, nbytes, CH, 0, out);
    run<8, 512, true, false, false>("chunk nt", x, nbytes, 2 * CH, 0, out);
    run<8, 128, true, false, false>("chunk nt", x, nbytes, CH, 0, out);
    run<8, 256, true, true, false>("chunk nt pipelined", x, nbytes, CH, 0, out);
    run<4, 256, true, true, false>("chunk nt pipelined", x, nbytes, CH, 0, out);
    run<8, 256, true, false, true>("persistent nt 8 wg/cu", x, nbytes, CH, 8, out);
    run<8, 256, true, false, true>("persistent nt 4 wg/cu", x, nbytes, CH, 4, out);
    run<8, 256, true, true, true>("persistent nt pipelined 8 wg/cu", x, nbytes, CH, 8, out);
    run<8, 256, true, true, true>("persistent nt pipelined 4 wg/cu", x, nbytes, CH, 4, out);
    run<4, 256, true, true, true>("persistent nt pipelined 8 wg/cu", x, nbytes, CH, 8, out);
    run<8, 256, false, true, true>("persistent plain pipelined 8 wg/cu", x, nbytes, CH, 8, out);
    run<8, 256, true, false, false>("chunk nt 32 KB chunks", x, nbytes, 32768, 0, out);
    run<8, 256, true, false, false>("chunk nt 1 MB chunks", x, nbytes, 1 << 20, 0, out);
    run<8, 256, true, false, true>("persistent nt 8 wg/cu, 32 KB chunks", x, nbytes, 32768, 8, out);
    run<16, 256, true, false, true>("persistent nt 4 wg/cu", x, nbytes, CH, 4, out);
    run<16, 256, true, false, true>("persistent nt 8 wg/cu", x, nbytes, CH, 8, out);
    return 0;
}
