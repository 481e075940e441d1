// Calibration: per-block partial rows + a reduce in the consumer, against fixed-point atomic accumulation into G rows
// (squeeze-excite fc1 pre-activations: 256 producer blocks x 240 values per sample).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 240;
__global__ void k_part(float* part, const float* src) {
    const float v = src[(blockIdx.x % 256) * 1024 + threadIdx.x];
    if (threadIdx.x < S) part[blockIdx.x * S + threadIdx.x] = v;
}
template <int G> __global__ void k_atom(unsigned long long* acc, const float* src) {
    const float v = src[(blockIdx.x % 256) * 1024 + threadIdx.x];
    if (threadIdx.x < S) atomicAdd(&acc[(blockIdx.x % G) * S + threadIdx.x], (unsigned long long)(long long)__double2ll_rn((double)v * 1099511627776.0));
}
// consumer: 15 blocks, every one needs the whole vector
__global__ void k_red_part(const float* part, int nparts, float* out) {
    __shared__ float sh[1024];
    const int i = threadIdx.x % S, g = threadIdx.x / S;       // 4 groups
    float s = 0.f;
    if (g < 4) for (int p = g; p < nparts; p += 64) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = part[(size_t)min(p + 4 * j, nparts - 1) * S + i];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += (p + 4 * j < nparts) ? v[j] : 0.f;
    }
    sh[threadIdx.x] = s; __syncthreads();
    if (threadIdx.x < S) out[blockIdx.x * S + threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + S] + sh[threadIdx.x + 2 * S] + sh[threadIdx.x + 3 * S];
}
template <int G> __global__ void k_red_atom(const unsigned long long* acc, float* out) {
    if (threadIdx.x < S) {
        long long s = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) s += (long long)acc[g * S + threadIdx.x];
        out[blockIdx.x * S + threadIdx.x] = (float)((double)s * (1.0 / 1099511627776.0));
    }
}
int main() {
    float *src, *part, *out; unsigned long long* acc;
    CK(hipMalloc(&src, 256 * 1024 * 4)); CK(hipMalloc(&part, 256 * S * 4)); CK(hipMalloc(&out, 128 * S * 4)); CK(hipMalloc(&acc, 64 * S * 8));
    CK(hipMemset(src, 0, 256 * 1024 * 4)); CK(hipMemset(acc, 0, 64 * S * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(k_part, dim3(256), dim3(1024), 0, st, part, src);
        hipLaunchKernelGGL(k_red_part, dim3(15), dim3(1024), 0, st, part, 256, out);
        hipLaunchKernelGGL((k_atom<1>), dim3(256), dim3(1024), 0, st, acc, src);
        hipLaunchKernelGGL((k_red_atom<1>), dim3(15), dim3(1024), 0, st, acc, out);
        hipLaunchKernelGGL((k_atom<8>), dim3(256), dim3(1024), 0, st, acc, src);
        hipLaunchKernelGGL((k_red_atom<8>), dim3(15), dim3(1024), 0, st, acc, out);
        hipLaunchKernelGGL((k_atom<32>), dim3(256), dim3(1024), 0, st, acc, src);
        hipLaunchKernelGGL((k_red_atom<32>), dim3(15), dim3(1024), 0, st, acc, out);
        hipLaunchKernelGGL((k_atom<1>), dim3(2048), dim3(1024), 0, st, acc, src);      // 8 samples' worth on one row (worst case)
        hipLaunchKernelGGL((k_red_atom<1>), dim3(120), dim3(1024), 0, st, acc, out);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 20; ++it) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    printf("done\n");
    return 0;
}
