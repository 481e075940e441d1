// Device-side helpers shared by the kernel translation units (kernels.hip, mbconv_fused.hip): in-kernel timeline,
// vector load / store helpers, MFMA operand fragments of the three arithmetic modes, activations.
#pragma once
#include "common.hpp"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// Optional in-kernel timeline (build with -DRVM_STAMPS -> librvm_mi355x_stamps.so, tools/stamp_table.py):
// thread 0 of block 0 of every launch takes the next slot of g_stamp and records the 100 MHz wall
// clock at kernel start plus shader-clock stamps at phase boundaries.  Compiled out in the product.
// The stamps build is ONE translation unit (kernels.hip includes the other kernel files), so the tables below exist once.
// ------------------------------------------------------------------------------------------
#ifdef RVM_STAMPS
// table 0: block 0 of every launch; table 1: the LAST block of the grid (dispatched last, sees the loaded machine)
__device__ unsigned long long g_stamp[2 * 512 * 16];
__device__ unsigned int g_nlaunch[2];
struct Stamper {
    int base;
    __device__ __forceinline__ Stamper() : base(-1) {
        if ((threadIdx.x | threadIdx.y) != 0) return;
        int tab = -1;
        if ((blockIdx.x | blockIdx.y | blockIdx.z) == 0) tab = 0;
        else if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) tab = 1;
        if (tab < 0) return;
        base = tab * 512 * 16 + (int)(atomicAdd(&g_nlaunch[tab], 1u) & 511u) * 16;
        for (int k = 2; k < 16; ++k) g_stamp[base + k] = 0;
        g_stamp[base] = wall_clock64();
        g_stamp[base + 1] = clock64();
    }
    __device__ __forceinline__ void operator()(int k) { if (base >= 0) g_stamp[base + 1 + k] = clock64(); }
};
#else
struct Stamper { __device__ __forceinline__ void operator()(int) const {} };
#endif

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
struct F8 { f32x4 lo, hi; };

__device__ __forceinline__ f32x4 ld4f(const f16* p) { return __builtin_convertvector(*(const f16x4*)p, f32x4); }
__device__ __forceinline__ f32x4 ld4f(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ void st4f(f16* p, f32x4 v) { *(f16x4*)p = __builtin_convertvector(v, f16x4); }
__device__ __forceinline__ void st4f(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ F8 ld8f(const f16* p) {
    f16x8 v = *(const f16x8*)p; F8 r;
    r.lo = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    r.hi = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    return r;
}
__device__ __forceinline__ F8 ld8f(const float* p) { F8 r; r.lo = *(const f32x4*)p; r.hi = *(const f32x4*)(p + 4); return r; }
__device__ __forceinline__ void st8f(f16* p, F8 v) {
    f16x8 o = {(f16)v.lo[0], (f16)v.lo[1], (f16)v.lo[2], (f16)v.lo[3], (f16)v.hi[0], (f16)v.hi[1], (f16)v.hi[2], (f16)v.hi[3]};
    *(f16x8*)p = o;
}
__device__ __forceinline__ void st8f(float* p, F8 v) { *(f32x4*)p = v.lo; *(f32x4*)(p + 4) = v.hi; }

// split type: plain fp32 in global memory
__device__ __forceinline__ f32x4 ld4f(const f32s* p) { return *(const f32x4*)p; }
__device__ __forceinline__ void st4f(f32s* p, f32x4 v) { *(f32x4*)p = v; }
// hi = RN16(v), lo = RN16(v - hi).  v - hi is exact in fp32, so lo is ONE rounding of fma(hi, -1, v): v_fma_mix{lo,hi}_f16
// takes the f16 half and the fp32 value as they are - 2 + 4 instructions per 4 values instead of 2 + 4 conversions back +
// 4 subtractions + 2 packed conversions (the compiler does not form the mix instructions from the subtraction; bit-equal
// over all 2^32 inputs, tools/microbench/split_check.hip).  Every split-mode operand goes through here.
__device__ __forceinline__ unsigned split_lo2(unsigned hi2, float a, float b) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi2), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi2), "v"(b));
    return l;
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(f32x4 v, f16x4& hi, f16x4& lo) {
    hi = __builtin_convertvector(v, f16x4);
    const u32x2_t h = __builtin_bit_cast(u32x2_t, hi);
    lo = __builtin_bit_cast(f16x4, u32x2_t{split_lo2(h[0], v[0], v[1]), split_lo2(h[1], v[2], v[3])});
}
// LDS halo-patch stores: 4 channels c..c+3 (c % 4 == 0) of the pixel whose row starts at `row`.
// f16 / float: channel-major row.  split: per 8-channel group [8 x hi f16][8 x lo f16] (32 B).
__device__ __forceinline__ void stp4(f16* row, int c, f32x4 v) { st4f(row + c, v); }
__device__ __forceinline__ void stp4(float* row, int c, f32x4 v) { st4f(row + c, v); }
__device__ __forceinline__ void stp4(f32s* row, int c, f32x4 v) {
    f16x4 hi, lo;
    split4(v, hi, lo);
    f16* g = (f16*)(row + (c & ~7)) + (c & 7);
    *(f16x4*)g = hi;
    *(f16x4*)(g + 8) = lo;
}

// bf16 <-> fp32, four at a time (round-to-nearest-even like bf16's constructor)
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4f(const bf16* p) {
    const u16x4 u = *(const u16x4*)p;
    return f32x4{(float)bf16_bits(u[0]), (float)bf16_bits(u[1]), (float)bf16_bits(u[2]), (float)bf16_bits(u[3])};
}
__device__ __forceinline__ void st4f(bf16* p, f32x4 v) {
    *(u16x4*)p = u16x4{bf16(v[0]).v, bf16(v[1]).v, bf16(v[2]).v, bf16(v[3]).v};
}
// element-typed access through a runtime mode (16-bit tensors at the API boundary of the mixed modes):
// 0 / false = fp32, 1 / true = fp16, 2 = bf16 (IoMode)
__device__ __forceinline__ f32x4 ld4x(const void* p, size_t i, int mode) {
    return mode == 1 ? ld4f((const f16*)p + i) : (mode == 2 ? ld4f((const bf16*)p + i) : ld4f((const float*)p + i));
}
__device__ __forceinline__ void st4x(void* p, size_t i, int mode, f32x4 v) {
    if (mode == 1) st4f((f16*)p + i, v); else if (mode == 2) st4f((bf16*)p + i, v); else st4f((float*)p + i, v);
}

// MFMA operand fragments: 8 K-elements per lane.
template <typename T> struct Frag;
template <> struct Frag<f16> {
    f16x8 v;
    __device__ __forceinline__ void load(const f16* p) { v = *(const f16x8*)p; }
    __device__ __forceinline__ void load_lds(const f16* p) { v = *(const f16x8*)p; }
    // packed weight fragment (512 elements): lane l owns elements [8l, 8l+8)
    __device__ __forceinline__ void loadw(const f16* frag, int lane) { v = *(const f16x8*)(frag + lane * 8); }
    __device__ __forceinline__ void zero() { v = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    __device__ __forceinline__ void scale(const f16* s) { v = v * *(const f16x8*)s; }
};
template <> struct Frag<float> {
    f32x4 lo, hi;
    __device__ __forceinline__ void load(const float* p) { lo = *(const f32x4*)p; hi = *(const f32x4*)(p + 4); }
    __device__ __forceinline__ void load_lds(const float* p) { load(p); }
    // packed weight fragment (512 elements) is stored as two planes of 64 x 4 floats so that both the
    // global load and the LDS read are 16 contiguous bytes per lane (no 32-byte lane stride)
    __device__ __forceinline__ void loadw(const float* frag, int lane) {
        lo = *(const f32x4*)(frag + lane * 4); hi = *(const f32x4*)(frag + 256 + lane * 4);
    }
    __device__ __forceinline__ void zero() { lo = f32x4{0, 0, 0, 0}; hi = lo; }
    __device__ __forceinline__ void scale(const float* s) { lo = lo * *(const f32x4*)s; hi = hi * *(const f32x4*)(s + 4); }
};

template <> struct Frag<f32s> {
    f16x8 hi, lo;
    // from global memory: 8 fp32 values, split in registers
    __device__ __forceinline__ void load(const f32s* p) { set(*(const f32x4*)p, *(const f32x4*)((const float*)p + 4)); }
    __device__ __forceinline__ void set(f32x4 a, f32x4 b) {
        f16x4 h0, l0, h1, l1;
        split4(a, h0, l0); split4(b, h1, l1);
        hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // from the LDS halo patch: already split (stp4)
    __device__ __forceinline__ void load_lds(const f32s* p) { hi = *(const f16x8*)p; lo = *(const f16x8*)((const f16*)p + 8); }
    // packed weight fragment (512 x 4 bytes): [64 lanes][8] hi halves, then [64][8] lo halves
    __device__ __forceinline__ void loadw(const f32s* frag, int lane) {
        hi = *(const f16x8*)((const f16*)frag + lane * 8); lo = *(const f16x8*)((const f16*)frag + 512 + lane * 8);
    }
    __device__ __forceinline__ void zero() { hi = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; lo = hi; }
};
template <typename T> __device__ __forceinline__ void load_scaled(Frag<T>& f, const T* p, const T* s) { f.load(p); f.scale(s); }
template <> __device__ __forceinline__ void load_scaled<f32s>(Frag<f32s>& f, const f32s* p, const f32s* s) {
    const float* pf = (const float*)p; const float* sf = (const float*)s;
    f.set(*(const f32x4*)pf * *(const f32x4*)sf, *(const f32x4*)(pf + 4) * *(const f32x4*)(sf + 4));
}

// Activation operand exactly as loaded from global memory.  Nothing is computed at the load (the hi/lo split of
// the f32s type and the SE rescale both need the data, i.e. a wait for the load): the conversion to an MFMA
// fragment happens where the fragment is used, K steps later, so the loads of a prefetch window stay in flight.
template <typename T> struct XRaw;
template <> struct XRaw<f16> {
    f16x8 v;
    __device__ __forceinline__ void load(const f16* p) { v = *(const f16x8*)p; }
    __device__ __forceinline__ void zero() { v = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    __device__ __forceinline__ Frag<f16> frag() const { Frag<f16> f; f.v = v; return f; }
    __device__ __forceinline__ Frag<f16> frag(const f16* s) const { Frag<f16> f; f.v = v * *(const f16x8*)s; return f; }
};
template <> struct XRaw<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    __device__ __forceinline__ void zero() { a = f32x4{0, 0, 0, 0}; b = a; }
    __device__ __forceinline__ Frag<float> frag() const { Frag<float> f; f.lo = a; f.hi = b; return f; }
    __device__ __forceinline__ Frag<float> frag(const float* s) const {
        Frag<float> f; f.lo = a * *(const f32x4*)s; f.hi = b * *(const f32x4*)(s + 4); return f;
    }
};
template <> struct XRaw<f32s> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const f32s* p) { a = *(const f32x4*)p; b = *(const f32x4*)((const float*)p + 4); }
    __device__ __forceinline__ void zero() { a = f32x4{0, 0, 0, 0}; b = a; }
    __device__ __forceinline__ Frag<f32s> frag() const { Frag<f32s> f; f.set(a, b); return f; }
    __device__ __forceinline__ Frag<f32s> frag(const f32s* s) const {
        const float* sf = (const float*)s;
        Frag<f32s> f; f.set(a * *(const f32x4*)sf, b * *(const f32x4*)(sf + 4)); return f;
    }
};

// acc(16 out-channels x 16 pixels) += W(16 x 32) * X(32 x 16)
__device__ __forceinline__ void mma(f32x4& acc, const Frag<f32s>& w, const Frag<f32s>& x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.lo, x.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi, x.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi, x.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const Frag<f16>& w, const Frag<f16>& x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.v, x.v, acc, 0, 0, 0);
}
// fp32: 8 MFMAs of K=4; in MFMA i lane group g=(l>>4) supplies k = 8g+i for BOTH operands, so
// the 32-wide K step is covered exactly once (the order inside the step is immaterial).
__device__ __forceinline__ void mma(f32x4& acc, const Frag<float>& w, const Frag<float>& x) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo[i], x.lo[i], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi[i], x.hi[i], acc, 0, 0, 0);
}

// MW x NW accumulators against NW weight and MW pixel fragments for the split type.  The three MFMAs of one accumulator
// are a dependent chain; issued accumulator by accumulator they stall on each other and on the fragment reads (what the
// compiler schedules when left alone: read one pixel fragment, wait, three dependent MFMAs, next fragment ...).  Here the
// accumulators are taken in groups of >= 4 (GM pixel fragments x NW), each group term by term, and the scheduler is
// told to put the fragment reads of group i + 1 in front of the MFMAs of group i (sched_group_barrier: 0x100 = DS read,
// 0x008 = MFMA), so reads overlap MFMAs inside one wave and no MFMA waits for its predecessor.
template <int MW, int NW> struct MmaGroups { static constexpr int GM = NW >= 4 ? 1 : (NW >= 2 ? 2 : MW); };
template <int MW, int NW>
__device__ __forceinline__ void mma_tile(f32x4 (&acc)[MW][NW], const Frag<f32s> (&w)[NW], const Frag<f32s> (&x)[MW]) {
    constexpr int GM = MmaGroups<MW, NW>::GM;
#pragma unroll
    for (int m0 = 0; m0 < MW; m0 += GM) {
#pragma unroll
        for (int mi = m0; mi < m0 + GM && mi < MW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ni].lo, x[mi].hi, acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = m0; mi < m0 + GM && mi < MW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ni].hi, x[mi].lo, acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = m0; mi < m0 + GM && mi < MW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ni].hi, x[mi].hi, acc[mi][ni], 0, 0, 0);
    }
}
// the schedule that goes with it: call right after mma_tile in a K step whose fragments come from LDS (two ds_read_b128 each)
template <typename T, int MW, int NW> __device__ __forceinline__ void mma_tile_schedule() {
    if constexpr (std::is_same<T, f32s>::value) {
        constexpr int GM = MmaGroups<MW, NW>::GM;
        constexpr int g0 = GM < MW ? GM : MW;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NW + 2 * g0, 0);            // all weight fragments + the first group's pixels
#pragma unroll
        for (int m0 = 0; m0 < MW; m0 += GM) {
            const int cur = (MW - m0) < GM ? (MW - m0) : GM;
            const int nxt = m0 + GM < MW ? ((MW - m0 - GM) < GM ? (MW - m0 - GM) : GM) : 0;
            if (nxt == 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (nxt == 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            else if (nxt > 2) __builtin_amdgcn_sched_group_barrier(0x100, 2 * 5, 0);
            if (cur == 1) __builtin_amdgcn_sched_group_barrier(0x008, 3 * NW, 0);
            else if (cur == 2) __builtin_amdgcn_sched_group_barrier(0x008, 6 * NW, 0);
            else __builtin_amdgcn_sched_group_barrier(0x008, 3 * NW * MW, 0);
        }
    }
}
template <typename T, int MW, int NW>
__device__ __forceinline__ void mma_tile(f32x4 (&acc)[MW][NW], const Frag<T> (&w)[NW], const Frag<T> (&x)[MW]) {
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) mma(acc[mi][ni], w[ni], x[mi]);
}

// N independent accumulators against one weight fragment; the split type's three terms go term by term so that no MFMA
// waits for its predecessor
template <int N> __device__ __forceinline__ void mma_n(f32x4* acc, const Frag<f32s>& w, const Frag<f32s>* x) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.lo, x[i].hi, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi, x[i].lo, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi, x[i].hi, acc[i], 0, 0, 0);
}
template <int N, typename T> __device__ __forceinline__ void mma_n(f32x4* acc, const Frag<T>& w, const Frag<T>* x) {
#pragma unroll
    for (int i = 0; i < N; ++i) mma(acc[i], w, x[i]);
}
__device__ __forceinline__ float act_f(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_HSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
        case ACT_HSIGMOID: return fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }

// ---- tensors at the API boundary, 8 values at a time (dgf_apply, project_out, compose) ----------------------------------
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
template <typename TIO> struct IoV8;
template <> struct IoV8<f16> {
    typedef f16x8 type;
    static __device__ __forceinline__ float get(const f16x8& v, int j) { return (float)v[j]; }
    static __device__ __forceinline__ void set(f16x8& v, int j, f16 e) { v[j] = e; }
};
template <> struct IoV8<float> {
    typedef f32x8 type;
    static __device__ __forceinline__ float get(const f32x8& v, int j) { return v[j]; }
    static __device__ __forceinline__ void set(f32x8& v, int j, float e) { v[j] = e; }
};
template <> struct IoV8<bf16> {
    typedef u16x8 type;
    static __device__ __forceinline__ float get(const u16x8& v, int j) { return (float)bf16_bits(v[j]); }
    static __device__ __forceinline__ void set(u16x8& v, int j, bf16 e) { v[j] = e.v; }
};
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 8 output values of one channel, packed to the I/O type pair by pair as they are produced
template <typename TIO> struct Out8;
template <> struct Out8<f16> {
    unsigned r[4];
    __device__ __forceinline__ void set(int jp, float a, float b) { f16x2 p = {(f16)a, (f16)b}; r[jp] = __builtin_bit_cast(unsigned, p); }
    __device__ __forceinline__ void store(f16* p) const { *(u32x4*)p = u32x4{r[0], r[1], r[2], r[3]}; }
    __device__ __forceinline__ f16 get(int j) const { return __builtin_bit_cast(f16x2, r[j >> 1])[j & 1]; }
};
template <> struct Out8<bf16> {
    unsigned r[4];
    __device__ __forceinline__ void set(int jp, float a, float b) { r[jp] = (unsigned)bf16(a).v | ((unsigned)bf16(b).v << 16); }
    __device__ __forceinline__ void store(bf16* p) const { *(u32x4*)p = u32x4{r[0], r[1], r[2], r[3]}; }
    __device__ __forceinline__ bf16 get(int j) const { return bf16_bits((unsigned short)(r[j >> 1] >> (16 * (j & 1)))); }
};
template <> struct Out8<float> {
    float r[8];
    __device__ __forceinline__ void set(int jp, float a, float b) { r[2 * jp] = a; r[2 * jp + 1] = b; }
    __device__ __forceinline__ void store(float* p) const { *(f32x4*)p = f32x4{r[0], r[1], r[2], r[3]}; *(f32x4*)(p + 4) = f32x4{r[4], r[5], r[6], r[7]}; }
    __device__ __forceinline__ float get(int j) const { return r[j]; }
};
// Output step of the reference driver (inference.py:116,137-141; inference_utils.py:44), one pixel: every tensor op of the
// reference rounds to the tensor dtype, so the arithmetic is written in that type (TIO) op by op and the bytes come out
// identical.  Used by compose_u8 (its own pass over fgr / pha) and by dgf_apply's fused uint8 output.
template <typename TIO> __device__ __forceinline__ unsigned char to_byte(TIO v) {
    const TIO m = v * (TIO)255.0f;                                   // .mul(255) in the tensor dtype
    return (unsigned char)(int)(float)m;                            // .byte(): truncation (values are in [0, 255])
}
template <typename TIO>
__device__ __forceinline__ unsigned compose_px(int mode, TIO f0, TIO f1, TIO f2, TIO al, const float* bgr) {
    if (mode == COMPOSE_ALPHA) { const unsigned b = to_byte<TIO>(al); return b | (b << 8) | (b << 16); }
    const TIO one = (TIO)1.0f, zero = (TIO)0.0f;
    TIO f[3] = {f0, f1, f2};
    unsigned out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        TIO v = f[c];
        if (mode == COMPOSE_GREEN) {
#pragma clang fp contract(off)                                       // no fma: the reference rounds after every op
            const TIO t1 = v * al;                                   // each op rounds to TIO like the torch ops do
            const TIO t2 = one - al;
            const TIO t3 = (TIO)bgr[c] * t2;
            v = t1 + t3;
        } else if (mode == COMPOSE_RGBA) {
            v = v * (al > zero ? one : zero);
        }
        out |= (unsigned)to_byte<TIO>(v) << (8 * c);
    }
    if (mode == COMPOSE_RGBA) out |= (unsigned)to_byte<TIO>(al) << 24;
    return out;
}
