// Micro-benchmark: what a SPARSE write stream costs a streaming read, and whether clustering the writes changes it.
// Shape of the IW reduction with density outputs (C5): every 128,000-byte chunk read (one sentence's eps) produces
// W bytes of output (8,000 B: two [K] float rows).
//   chunk   one workgroup per chunk, its W bytes stored when the chunk has been read (the product kernel's shape)
//   group   one workgroup per S consecutive chunks, the S*W bytes stored at the end (bigger bursts per workgroup)
//   pers    a persistent grid (G workgroups, each a contiguous range of chunks), stores every S chunks: the
//           workgroups run in step, so the stores of the whole chip fall into common phases
//   tail    persistent, ALL stores at the end of the kernel (upper bound of what clustering can give; needs the
//           values to be kept, which a real kernel can only do for LDS-sized outputs)
// Stores are 16 bytes per lane, 1 KiB per wave-instruction; `nt` / `sc1` select the cache policy.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/rw_mix.cpp -o tools/ubench/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TPB 256
#define U 4

template <int POLICY>
__device__ __forceinline__ void store16(f32x4* p, f32x4 v) {
    if (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else *p = v;
}

__device__ __forceinline__ float read_chunk(const f32x4* __restrict__ base, int chunk_f4) {
    float acc = 0.f;
    const int iters = (chunk_f4 + TPB * U - 1) / (TPB * U);
    f32x4 a[U], b[U];
    auto issue = [&](f32x4 (&v)[U], int it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int idx = (it * U + u) * TPB + (int)threadIdx.x;
            if (idx > chunk_f4 - 1) idx = chunk_f4 - 1;
            v[u] = __builtin_nontemporal_load(base + idx);
        }
    };
    auto fold = [&](const f32x4 (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    };
    issue(a, 0);
    for (int it = 0; it < iters; it += 2) {
        if (it + 1 < iters) issue(b, it + 1);
        fold(a);
        if (it + 2 < iters) issue(a, it + 2);
        if (it + 1 < iters) fold(b);
    }
    return acc;
}

// chunks [c0, c1) of this workgroup; a store burst after every S chunks covering those chunks' outputs
template <int POLICY>
__global__ void __launch_bounds__(TPB) rw(const f32x4* __restrict__ x, f32x4* __restrict__ y, long nchunks, int chunk_f4,
                                         int w_f4, int per_wg, int S, float* out) {
    const long c0 = (long)blockIdx.x * per_wg;
    long c1 = c0 + per_wg;
    if (c1 > nchunks) c1 = nchunks;
    float acc = 0.f;
    long pending0 = c0;
    for (long c = c0; c < c1; ++c) {
        acc += read_chunk(x + c * chunk_f4, chunk_f4);
        if (c + 1 - pending0 == S || c + 1 == c1) {
            if (y) {
                const long lo = pending0 * w_f4, hi = (c + 1) * w_f4;
                const f32x4 v = {acc, acc, acc, acc};
                for (long i = lo + threadIdx.x; i < hi; i += TPB) store16<POLICY>(y + i, v);
            }
            pending0 = c + 1;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// A fifth wave does the storing: the four reading waves never have a store in their (in-order) memory counter and end
// -- or go on to the next chunk -- without waiting for a store to be acknowledged.
//   PERS = false: one workgroup per chunk;  PERS = true: a grid-strided persistent workgroup, one barrier per chunk
template <bool PERS>
__global__ void __launch_bounds__(TPB + 64) rw_writer(const f32x4* __restrict__ x, f32x4* __restrict__ y, long nchunks,
                                                     int chunk_f4, int w_f4, float* out) {
    __shared__ float sh[2][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    int it = 0;
    for (long c = blockIdx.x; c < nchunks; c += PERS ? gridDim.x : nchunks, ++it) {
        if (wave < 4) {
            acc = read_chunk(x + c * chunk_f4, chunk_f4);
            if (lane == 0) sh[it & 1][wave] = acc;
        }
        __syncthreads();
        if (wave == 4) {
            const float t = sh[it & 1][0] + sh[it & 1][1] + sh[it & 1][2] + sh[it & 1][3];
            const f32x4 v = {t, t, t, t};
            for (long i = c * w_f4 + lane; i < (c + 1) * w_f4; i += 64) y[i] = v;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <bool PERS>
static float run_writer(const char* name, const f32x4* x, f32x4* y, long nchunks, int chunk_bytes, int w_bytes, int grid,
                        float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((rw_writer<PERS>), dim3(grid), dim3(TPB + 64), 0, 0, x, y, nchunks, chunk_bytes / 16, w_bytes / 16, out);
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a);
        hipLaunchKernelGGL((rw_writer<PERS>), dim3(grid), dim3(TPB + 64), 0, 0, x, y, nchunks, chunk_bytes / 16, w_bytes / 16, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        sum += ms;
    }
    const double bytes = (double)nchunks * (chunk_bytes + w_bytes);
    printf("%-34s grid=%6d W=%5d B: best %.3f ms  mean %.3f ms  %.0f GB/s\n", name, grid, w_bytes, best, sum / reps,
           bytes / best / 1e6);
    fflush(stdout);
    return best;
}

template <int POLICY>
static float run(const char* name, const f32x4* x, f32x4* y, long nchunks, int chunk_bytes, int w_bytes, int per_wg, int S,
                 float* out) {
    const int grid = (int)((nchunks + per_wg - 1) / per_wg);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((rw<POLICY>), dim3(grid), dim3(TPB), 0, 0, x, y, nchunks, chunk_bytes / 16, w_bytes / 16, per_wg, S, out);
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a);
        hipLaunchKernelGGL((rw<POLICY>), dim3(grid), dim3(TPB), 0, 0, x, y, nchunks, chunk_bytes / 16, w_bytes / 16, per_wg, S, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        sum += ms;
    }
    const double bytes = (double)nchunks * (chunk_bytes + (y ? w_bytes : 0));
    printf("%-34s grid=%6d per_wg=%3d S=%3d W=%5d B: best %.3f ms  mean %.3f ms  %.0f GB/s\n", name, grid, per_wg, S,
           y ? w_bytes : 0, best, sum / reps, bytes / best / 1e6);
    fflush(stdout);
    return best;
}

int main() {
    const long nchunks = 16384;
    const int chunk_bytes = 128000;
    const long nbytes = nchunks * chunk_bytes;
    f32x4 *x, *y; float* out;
    if (hipMalloc(&x, nbytes) != hipSuccess || hipMalloc(&y, nchunks * 16384L) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    {
        const size_t hb = 1 << 26;
        float* h = (float*)malloc(hb);
        for (size_t i = 0; i < hb / 4; ++i) h[i] = (float)(i % 977) * 1e-3f;
        for (long o = 0; o < nbytes; o += hb) hipMemcpy((char*)x + o, h, (nbytes - o) < (long)hb ? (nbytes - o) : hb, hipMemcpyHostToDevice);
        free(h);
    }
    hipMemset(y, 0, nchunks * 16384L);
    hipDeviceSynchronize();
    run<0>("read only, chunk", x, nullptr, nchunks, chunk_bytes, 8000, 1, 1, out);
    run<0>("read only, pers 2048", x, nullptr, nchunks, chunk_bytes, 8000, 8, 8, out);
    for (int w : {8000, 4000, 16000}) {
        run<0>("chunk, plain stores", x, y, nchunks, chunk_bytes, w, 1, 1, out);
        run<1>("chunk, nt stores", x, y, nchunks, chunk_bytes, w, 1, 1, out);
        run<2>("chunk, sc1 stores", x, y, nchunks, chunk_bytes, w, 1, 1, out);
    }
    for (int w : {8000, 16000}) run_writer<false>("writer wave, chunk", x, y, nchunks, chunk_bytes, w, (int)nchunks, out);
    for (int g : {1024, 1280, 1536, 2048, 4096})
        run_writer<true>("writer wave, persistent", x, y, nchunks, chunk_bytes, 8000, g, out);
    for (int s : {2, 4, 8, 16}) run<0>("group (S chunks per wg)", x, y, nchunks, chunk_bytes, 8000, s, s, out);
    for (int per : {8, 16, 32, 64}) {
        run<0>("pers, stores every chunk", x, y, nchunks, chunk_bytes, 8000, per, 1, out);
        run<0>("pers, stores every 4", x, y, nchunks, chunk_bytes, 8000, per, 4, out);
        run<0>("tail (all stores at the end)", x, y, nchunks, chunk_bytes, 8000, per, per, out);
        run<1>("tail, nt", x, y, nchunks, chunk_bytes, 8000, per, per, out);
    }
    return 0;
}
