// How fast can ONE compute unit pull data that sits in L2 / Infinity Cache?  (the ceiling behind every few-block kernel
// of the single-frame schedule).  Blocks stream a private slice with U independent 16-byte loads in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void fetch(const f32x4* __restrict__ src, float* out, int per_block /* f32x4 elements */, int shared_slice) {
    const f32x4* p = src + (shared_slice ? 0 : (size_t)blockIdx.x * per_block);
    f32x4 acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < per_block; i += blockDim.x * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int j = i + u * blockDim.x; v[u] = p[j < per_block ? j : i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}
template <int U> void run(const f32x4* src, float* out, int blocks, int threads, int kb, int shared) {
    const int per = kb * 1024 / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fetch<U>, dim3(blocks), dim3(threads), 0, 0, src, out, per, shared);
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fetch<U>, dim3(blocks), dim3(threads), 0, 0, src, out, per, shared);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("blocks %4d x %4d thr  U=%2d  %5d KiB/block %s : %7.2f us  %7.1f GB/s per block  %8.1f GB/s total\n", blocks, threads, U, kb,
           shared ? "(same slice)" : "            ", us, kb * 1024.0 / us / 1e3, (double)blocks * kb * 1024.0 / us / 1e3);
}
int main() {
    const size_t bytes = 256u << 20;
    f32x4* src; float* out;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 0, bytes)); CK(hipMalloc(&out, 64));
    for (int shared = 0; shared < 2; ++shared)
        for (int blocks : {1, 32, 256})
            for (int threads : {256, 1024}) {
                const int kb = 256;
                run<1>(src, out, blocks, threads, kb, shared);
                run<2>(src, out, blocks, threads, kb, shared);
                run<4>(src, out, blocks, threads, kb, shared);
                run<8>(src, out, blocks, threads, kb, shared);
                run<16>(src, out, blocks, threads, kb, shared);
            }
    return 0;
}
