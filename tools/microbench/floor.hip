// Calibration: what does a tiny dependent kernel cost on this box?  (rocprofv3 --kernel-trace on this binary)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* p) {}
__global__ void k_store(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }
__global__ void k_ldst(const float* a, float* b) { int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[i] + 1.f; }
__global__ void k_ldst2(const float* a, const int* idx, float* b) { int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[idx[i]] + 1.f; }
__global__ void k_ldst_sync(const float* a, float* b) {
    __shared__ float s[1024];
    int i = blockIdx.x * blockDim.x + threadIdx.x; s[threadIdx.x] = a[i]; __syncthreads();
    float v = s[(threadIdx.x + 1) % blockDim.x]; b[i] = v + a[(i + 4096) % (gridDim.x * blockDim.x)] * v;   // second load depends on nothing, but after barrier
}
template <int N> __global__ void k_bigcode(const float* a, float* b) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; float v = a[i];
#pragma unroll
    for (int j = 0; j < N; ++j) v = __builtin_fmaf(v, 1.0001f + j, 0.5f - j);
    b[i] = v;
}
int main() {
    const int n = 1 << 20;
    float *a, *b; int* idx;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&idx, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(idx, 0, n * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int rep = 0; rep < 4; ++rep) {
        for (int blocks : {8, 64, 256, 1024}) {
            for (int th : {256, 1024}) {
                hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(th), 0, st, a);
                hipLaunchKernelGGL(k_store, dim3(blocks), dim3(th), 0, st, a);
                hipLaunchKernelGGL(k_ldst, dim3(blocks), dim3(th), 0, st, a, b);
                hipLaunchKernelGGL(k_ldst, dim3(blocks), dim3(th), 0, st, b, a);
                hipLaunchKernelGGL(k_ldst2, dim3(blocks), dim3(th), 0, st, a, idx, b);
                hipLaunchKernelGGL(k_ldst_sync, dim3(blocks), dim3(th), 0, st, b, a);
                hipLaunchKernelGGL((k_bigcode<64>), dim3(blocks), dim3(th), 0, st, a, b);
                hipLaunchKernelGGL((k_bigcode<4096>), dim3(blocks), dim3(th), 0, st, b, a);
            }
        }
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 20; ++it) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    printf("done\n");
    return 0;
}
