// Is q = fma(fma(-3, s * c, s), c, s * c) with c = RN(1/3) the correctly rounded s / 3 for every fp32 s?  (dgf_apply's gray)
// Exhaustive over all finite non-negative bit patterns; prints the number of mismatches against the hardware's IEEE division.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void check(unsigned long long* bad, unsigned* first) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (unsigned long long b = i; b < 0x7F800000ull; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float s = __uint_as_float((unsigned)b);
        const float c = 1.f / 3.f;
        const float q0 = s * c;
        const float r = __builtin_fmaf(-3.f, q0, s);
        const float q = __builtin_fmaf(r, c, q0);
        const float want = __fdiv_rn(s, 3.f);
        if (__float_as_uint(q) != __float_as_uint(want)) { ++nb; atomicMin(first, (unsigned)b); }
    }
    if (nb) atomicAdd(bad, nb);
}
int main() {
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long hb = 0; unsigned hf = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("mismatches %llu of %u patterns, first at bits 0x%08x (%g)\n", hb, 0x7F800000u, hf, (double)*(float*)&hf);
    return 0;
}
