// What does the access pattern of dgf_apply reach with no arithmetic?  Three fp16 planes in (r, g, b of a 4K frame), four out
// (fgr r, g, b + pha), 16 bytes per lane, one 2048-pixel row segment x R rows per block like the kernel; sources cycled through a
// ring larger than the 256 MiB Infinity Cache.  Prints TB/s for a few block shapes, and for a plain contiguous copy of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int R> __global__ __launch_bounds__(256) void planes(const unsigned short* src, unsigned short* dst, int H, int W) {
    const int x0 = blockIdx.x * 2048 + threadIdx.x * 8, y0 = blockIdx.y * R;
    if (x0 >= W) return;
    const size_t plane = (size_t)H * W;
    u32x4 v[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[r][c] = *(const u32x4*)(src + c * plane + (size_t)(y0 + r) * W + x0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const size_t o = (size_t)(y0 + r) * W + x0;
#pragma unroll
        for (int c = 0; c < 3; ++c) *(u32x4*)(dst + c * plane + o) = v[r][c];
        *(u32x4*)(dst + 3 * plane + o) = v[r][0] ^ v[r][1] ^ v[r][2];
    }
}
__global__ __launch_bounds__(256) void flat(const u32x4* src, u32x4* dst, size_t nin, size_t nout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)gridDim.x * 256;
    u32x4 a = {0, 0, 0, 0};
    for (size_t k = i; k < nin; k += n) a ^= src[k];
    for (size_t k = i; k < nout; k += n) dst[k] = a;
}
int main() {
    const int H = 2160, W = 3840, RING = 8;
    const size_t plane = (size_t)H * W, inb = 3 * plane * 2, outb = 4 * plane * 2;
    std::vector<unsigned short*> src(RING);
    unsigned short* dst;
    for (auto& p : src) { if (hipMalloc(&p, inb) != hipSuccess) return 1; (void)hipMemset(p, 1, inb); }
    if (hipMalloc(&dst, outb) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch(i % RING);
        (void)hipEventRecord(e0);
        const int reps = 40;
        for (int i = 0; i < reps; ++i) launch(i % RING);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.2f us  %.2f TB/s\n", name, ms / reps * 1e3, (inb + outb) / (ms / reps * 1e-3) / 1e12);
    };
    time("planes R=1", [&](int k) { hipLaunchKernelGGL(planes<1>, dim3(2, H), dim3(256), 0, 0, src[k], dst, H, W); });
    time("planes R=2", [&](int k) { hipLaunchKernelGGL(planes<2>, dim3(2, H / 2), dim3(256), 0, 0, src[k], dst, H, W); });
    time("planes R=4", [&](int k) { hipLaunchKernelGGL(planes<4>, dim3(2, H / 4), dim3(256), 0, 0, src[k], dst, H, W); });
    time("flat copy, 2048 blocks", [&](int k) { hipLaunchKernelGGL(flat, dim3(2048), dim3(256), 0, 0, (const u32x4*)src[k], (u32x4*)dst, inb / 16, outb / 16); });
    time("flat copy, 8192 blocks", [&](int k) { hipLaunchKernelGGL(flat, dim3(8192), dim3(256), 0, 0, (const u32x4*)src[k], (u32x4*)dst, inb / 16, outb / 16); });
    return 0;
}
