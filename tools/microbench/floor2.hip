// Calibration without a profiler: per-launch cost of a chain of 200 identical dependent kernels in a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_empty(float* p) {}
__global__ void k_ldst(const float* a, float* b) { int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[i] + 1.f; }
__global__ void k_ld2st(const float* a, const int* idx, float* b) { int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[idx[i] + i] + 1.f; }
__global__ void k_ld3st(const float* a, const int* idx, float* b) { int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[idx[idx[i] + i] + i] + 1.f; }
template <int N> __global__ void k_code(const float* a, float* b) {      // N straight-line FMAs in 4 independent chains
    int i = blockIdx.x * blockDim.x + threadIdx.x; float v0 = a[i], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
#pragma unroll
    for (int j = 0; j < N / 4; ++j) { v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0002f, 0.25f); v2 = __builtin_fmaf(v2, 1.0003f, 0.125f); v3 = __builtin_fmaf(v3, 0.9999f, 0.0625f); }
    b[i] = (v0 + v1) + (v2 + v3);
}
template <int N> __global__ void k_loop(const float* a, float* b, int n) {   // same work as a rolled loop
    int i = blockIdx.x * blockDim.x + threadIdx.x; float v0 = a[i], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
#pragma unroll 1
    for (int j = 0; j < n; ++j) { v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0002f, 0.25f); v2 = __builtin_fmaf(v2, 1.0003f, 0.125f); v3 = __builtin_fmaf(v3, 0.9999f, 0.0625f); }
    b[i] = (v0 + v1) + (v2 + v3);
}
float *A, *B; int* IDX; hipStream_t st;
void run(const char* name, int blocks, int th, std::function<void(int)> launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 200; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < 10; ++it) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-12s blocks %5d x %4d : %6.2f us per launch\n", name, blocks, th, ms * 1e3 / 2000);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}
int main() {
    const int n = 1 << 21;
    CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&B, n * 4)); CK(hipMalloc(&IDX, n * 4));
    CK(hipMemset(A, 0, n * 4)); CK(hipMemset(B, 0, n * 4)); CK(hipMemset(IDX, 0, n * 4));
    CK(hipStreamCreate(&st));
    for (int blocks : {8, 128, 1024}) for (int th : {256, 1024}) {
        dim3 g(blocks), b(th);
        run("empty", blocks, th, [&](int i) { hipLaunchKernelGGL(k_empty, g, b, 0, st, A); });
        run("ldst", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_ldst, g, b, 0, st, A, B); else hipLaunchKernelGGL(k_ldst, g, b, 0, st, B, A); });
        run("ld2st", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_ld2st, g, b, 0, st, A, IDX, B); else hipLaunchKernelGGL(k_ld2st, g, b, 0, st, B, IDX, A); });
        run("ld3st", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_ld3st, g, b, 0, st, A, IDX, B); else hipLaunchKernelGGL(k_ld3st, g, b, 0, st, B, IDX, A); });
        run("code256", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL((k_code<256>), g, b, 0, st, A, B); else hipLaunchKernelGGL((k_code<256>), g, b, 0, st, B, A); });
        run("code1024", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL((k_code<1024>), g, b, 0, st, A, B); else hipLaunchKernelGGL((k_code<1024>), g, b, 0, st, B, A); });
        run("code4096", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL((k_code<4096>), g, b, 0, st, A, B); else hipLaunchKernelGGL((k_code<4096>), g, b, 0, st, B, A); });
        run("loop1024", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL((k_loop<0>), g, b, 0, st, A, B, 256); else hipLaunchKernelGGL((k_loop<0>), g, b, 0, st, B, A, 256); });
        run("loop4096", blocks, th, [&](int i) { if (i & 1) hipLaunchKernelGGL((k_loop<0>), g, b, 0, st, A, B, 1024); else hipLaunchKernelGGL((k_loop<0>), g, b, 0, st, B, A, 1024); });
    }
    return 0;
}
