// Is the (hi, lo) split of the split-MFMA mode through v_fma_mix{lo,hi}_f16 (device_common.hpp: split4) bit-equal to the plain
// form  hi = (f16)v, lo = (f16)(v - (float)hi)  for EVERY fp32 bit pattern?  Prints the number of mismatches (NaN payloads
// are compared as NaN = NaN).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned split_lo2(unsigned hi2, float a, float b) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi2), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi2), "v"(b));
    return l;
}
__device__ __forceinline__ bool same16(unsigned short x, unsigned short y) {
    const bool nx = (x & 0x7FFF) > 0x7C00, ny = (y & 0x7FFF) > 0x7C00;
    return (nx && ny) || x == y;
}
__global__ void check(unsigned long long* bad, unsigned* first) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (unsigned long long b = 2 * i; b < 0x100000000ull; b += 2ull * gridDim.x * blockDim.x) {
        const float v0 = __uint_as_float((unsigned)b), v1 = __uint_as_float((unsigned)b + 1u);
        const f16x2 h = {(f16)v0, (f16)v1};
        const f16x2 want = {(f16)(v0 - (float)h[0]), (f16)(v1 - (float)h[1])};
        const unsigned got = split_lo2(__builtin_bit_cast(unsigned, h), v0, v1), w = __builtin_bit_cast(unsigned, want);
        if (!same16(got & 0xFFFF, w & 0xFFFF)) { ++nb; atomicMin(first, (unsigned)b); }
        if (!same16(got >> 16, w >> 16)) { ++nb; atomicMin(first, (unsigned)b + 1u); }
    }
    if (nb) atomicAdd(bad, nb);
}
int main() {
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4);
    hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, bad, first);
    unsigned long long hb = 0; unsigned hf = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("mismatches %llu of 2^32 patterns, first at bits 0x%08x (%g)\n", hb, hf, (double)*(float*)&hf);
    return 0;
}
