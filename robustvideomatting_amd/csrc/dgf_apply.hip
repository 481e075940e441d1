// ------------------------------------------------------------------------------------------
// Refiner, full-res half: bilinear-upsample A|b (size-driven coords, deep_guided_filter.py:
// 37-39), out = A*[src,gray]+b (:41), fgr = clamp(out+src), pha = clamp(out) (model.py:65-67).
// One pass: src read once, fgr/pha written once; a thread owns 8 consecutive x.
// ------------------------------------------------------------------------------------------
#include "device_common.hpp"

// (1 - l) * a + l * b and the output step, with the fused multiply-adds written out: both kernels below must round alike
// (left to the compiler, the contraction of a * b + c * d into an fma picks either product, and differs between kernels)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 dgf_lerp(f32x4 a, f32x4 b, float l) {
#pragma clang fp contract(off)
    // two lanes of fp32 per instruction (v_pk_mul_f32 / v_pk_fma_f32): the same IEEE operations, half the issue slots
    const float m = 1.f - l;
    const f32x2 l2 = {l, l}, m2 = {m, m};
    const f32x2 r0 = __builtin_elementwise_fma(l2, f32x2{b[0], b[1]}, m2 * f32x2{a[0], a[1]});
    const f32x2 r1 = __builtin_elementwise_fma(l2, f32x2{b[2], b[3]}, m2 * f32x2{a[2], a[3]});
    return f32x4{r0[0], r0[1], r1[0], r1[1]};
}
__device__ __forceinline__ void dgf_pixel(f32x4 A, f32x4 B, float sr, float sg, float sb, float* o) {
#pragma clang fp contract(off)
    // (r + g + b) / 3 as q0 = s * c, q = fma(fma(-3, q0, s), c, q0) with c = RN(1 / 3): equal to the IEEE quotient for every
    // finite s (exhaustive check: tools/microbench/div3_check.hip), 3 instructions instead of the ~11 of a true division
    const float ssum = (sr + sg + sb), c3 = 1.f / 3.f, q0 = ssum * c3;
    const float gray = __builtin_fmaf(__builtin_fmaf(-3.f, q0, ssum), c3, q0);
    o[0] = fminf(fmaxf(__builtin_fmaf(A[0], sr, B[0]) + sr, 0.f), 1.f);
    o[1] = fminf(fmaxf(__builtin_fmaf(A[1], sg, B[1]) + sg, 0.f), 1.f);
    o[2] = fminf(fmaxf(__builtin_fmaf(A[2], sb, B[2]) + sb, 0.f), 1.f);
    o[3] = fminf(fmaxf(__builtin_fmaf(A[3], gray, B[3]), 0.f), 1.f);
}

// Output step of one thread's 8 pixels of row y: fgr / pha in the I/O type, or the driver's composed uint8 frame.
template <typename TIO>
__device__ __forceinline__ void dgf_emit(const DgfApplyArgs& a, int n, int y, int x0, bool full, size_t plane,
                                         const Out8<TIO> (&of)[3], const Out8<TIO>& op) {
    if (a.u8) {
        // fused driver output: composed uint8 NHWC instead of fgr / pha (the values are first rounded to TIO exactly as
        // the fgr / pha tensors would hold them)
        unsigned px[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) px[j] = compose_px<TIO>(a.u8mode, of[0].get(j), of[1].get(j), of[2].get(j), op.get(j), a.bgr);
        const size_t p0 = (size_t)n * plane + (size_t)y * a.W + x0;
        if (a.u8mode == COMPOSE_RGBA) {
            unsigned* o = (unsigned*)(a.u8 + p0 * 4);
            if (full) { *(u32x4*)o = u32x4{px[0], px[1], px[2], px[3]}; *(u32x4*)(o + 4) = u32x4{px[4], px[5], px[6], px[7]}; }
            else for (int j = 0; j < 8; ++j) if (x0 + j < a.W) o[j] = px[j];
        } else if (full) {
            // 8 px x 3 bytes = 24 bytes at an 8-byte aligned address (x0 % 8 == 0, W % 8 == 0)
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2* o = (u32x2*)(a.u8 + p0 * 3);
            const unsigned w0 = (px[0] & 0xFFFFFF) | (px[1] << 24), w1 = ((px[1] >> 8) & 0xFFFF) | (px[2] << 16);
            const unsigned w2 = ((px[2] >> 16) & 0xFF) | (px[3] << 8), w3 = (px[4] & 0xFFFFFF) | (px[5] << 24);
            const unsigned w4 = ((px[5] >> 8) & 0xFFFF) | (px[6] << 16), w5 = ((px[6] >> 16) & 0xFF) | (px[7] << 8);
            o[0] = u32x2{w0, w1}; o[1] = u32x2{w2, w3}; o[2] = u32x2{w4, w5};
        } else {
            unsigned char* o = a.u8 + p0 * 3;
            for (int j = 0; j < 8; ++j)
                if (x0 + j < a.W) { o[3 * j] = px[j] & 0xFF; o[3 * j + 1] = (px[j] >> 8) & 0xFF; o[3 * j + 2] = (px[j] >> 16) & 0xFF; }
        }
        return;
    }
    TIO* fp = (TIO*)a.fgr + (size_t)n * 3 * plane + (size_t)y * a.W + x0;
    TIO* pp = (TIO*)a.pha + (size_t)n * plane + (size_t)y * a.W + x0;
    if (full) {
#pragma unroll
        for (int c = 0; c < 3; ++c) of[c].store(fp + c * plane);
        op.store(pp);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (x0 + j < a.W) {
#pragma unroll
                for (int c = 0; c < 3; ++c) fp[c * plane + j] = of[c].get(j);
                pp[j] = op.get(j);
            }
        }
    }
}

// Block = one segment of 2048 consecutive x of one row (256 threads x 8 px).  The A|b rows are first
// interpolated vertically for the low-res columns the segment touches and parked in LDS
// ([column][8] fp32), then every thread interpolates horizontally out of LDS, branch-free, and packs
// its results to the I/O type pair by pair (few registers -> many rows in flight per CU).
template <typename TIO>
__global__ __launch_bounds__(256, 6) void dgf_apply(DgfApplyArgs a) {      // <= 85 VGPRs: 6 rows in flight per CU
    [[maybe_unused]] Stamper stamp;
    typedef typename IoV8<TIO>::type V8;
    extern __shared__ __attribute__((aligned(16))) f32x4 colAB[];       // [ncols][2]
    const int tid = threadIdx.x;
    const int seg0 = blockIdx.x * 2048;
    const int x0 = seg0 + tid * 8;
    const int y = blockIdx.y, n = blockIdx.z;
    const size_t plane = (size_t)a.H * a.W;
    const TIO* sp = (const TIO*)a.src + (size_t)n * 3 * plane + (size_t)y * a.W + x0;
    const bool active = x0 < a.W;
    const bool full = (x0 + 8 <= a.W) && ((a.W & 7) == 0);
    V8 s[3];
    if (full) {                                                         // issued first: in flight during the LDS phase
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] = *(const V8*)(sp + c * plane);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                TIO e = (TIO)0.f;
                if (active && x0 + j < a.W) e = sp[c * plane + j];
                IoV8<TIO>::set(s[c], j, e);
            }
    }
    // ---- vertical interpolation of the columns this segment needs
    float sy = a.scale_y * (y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    int iy0 = (int)sy; if (iy0 > a.h - 1) iy0 = a.h - 1;
    const int iy1 = iy0 + (iy0 < a.h - 1);
    const float ly = sy - iy0;
    float sxb = a.scale_x * (seg0 + 0.5f) - 0.5f; if (sxb < 0.f) sxb = 0.f;
    const int cbase = min((int)sxb, a.w - 1);
    const f32x4* r0 = (const f32x4*)(a.ab + ((size_t)n * a.h + iy0) * a.w * 8);
    const f32x4* r1 = (const f32x4*)(a.ab + ((size_t)n * a.h + iy1) * a.w * 8);
    // item = (column, A or b half); the first 5 x 256 items (all of them up to a 0.31 ratio) are requested
    // together - one round trip to L2 instead of one per 256 items
    {
        constexpr int SU = 5;
        f32x4 t[SU], u[SU];
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            const int i = tid + 256 * k;
            if (256 * k < 2 * a.ncols) {                                // uniform
                const int col = min(cbase + (i >> 1), a.w - 1);
                t[k] = r0[col * 2 + (i & 1)]; u[k] = r1[col * 2 + (i & 1)];
            }
        }
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            const int i = tid + 256 * k;
            if (i < 2 * a.ncols) colAB[i] = dgf_lerp(t[k], u[k], ly);
        }
        for (int i = tid + 256 * SU; i < 2 * a.ncols; i += 256) {
            const int col = min(cbase + (i >> 1), a.w - 1);
            const f32x4 tt = r0[col * 2 + (i & 1)], uu = r1[col * 2 + (i & 1)];
            colAB[i] = dgf_lerp(tt, uu, ly);
        }
    }
    __syncthreads();
    if (!active) return;
    Out8<TIO> of[3], op;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        float o[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * jp + e;
            float sx = a.scale_x * (x0 + j + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            int ix0 = (int)sx; if (ix0 > a.w - 1) ix0 = a.w - 1;
            const int ix1 = ix0 + (ix0 < a.w - 1);
            const float lx = sx - ix0;
            const int k0 = min(ix0 - cbase, a.ncols - 1), k1 = min(ix1 - cbase, a.ncols - 1);
            const f32x4 A = dgf_lerp(colAB[2 * k0], colAB[2 * k1], lx);
            const f32x4 B = dgf_lerp(colAB[2 * k0 + 1], colAB[2 * k1 + 1], lx);
            dgf_pixel(A, B, IoV8<TIO>::get(s[0], j), IoV8<TIO>::get(s[1], j), IoV8<TIO>::get(s[2], j), o[e]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) of[c].set(jp, o[0][c], o[1][c]);
        op.set(jp, o[0][3], o[1][3]);
    }
    dgf_emit<TIO>(a, n, y, x0, full, plane, of, op);
}

// The usual ratios: W = KX w exactly with KX = 4 or 8 (downsample_ratio 0.25 / 0.125) and W % 8 == 0.  A thread's 8 pixels
// then always sit on the same 4 (3) low-res columns x0 / KX - 1 ... with the same fractions, all known at compile time: the
// thread reads those columns once per row (8 or 6 LDS reads instead of 32 gathered ones, which ran 4-way bank-conflicted
// and were what bounded the general kernel) and spends nothing on per-pixel geometry.  R = 1 or 2 rows per block; with 2, the
// (column, half) items of THREE low-res rows are fetched once and both vertically interpolated rows written to LDS.  The
// arithmetic per value is the general kernel's, in the same order (the fractions are exact binary numbers either way), so
// the outputs are bit-identical to it.
template <int KX> struct DgfTaps {
    static constexpr int NC = KX == 4 ? 4 : 3;
    // pixel j of the thread: sx = (x0 + j + 0.5) / KX - 0.5 = x0 / KX + u / KX with u = j + 0.5 - KX / 2
    static constexpr int fl(int j) { return (2 * j + 1 - KX) >= 0 ? (2 * j + 1 - KX) / (2 * KX) : -((KX - 2 * j - 1 + 2 * KX - 1) / (2 * KX)); }   // floor(u / KX)
    static constexpr int i0(int j) { return fl(j) + 1; }                                      // index into the thread's columns
    static constexpr float lx(int j) { return (float)(2 * j + 1 - KX) / (float)(2 * KX) - (float)fl(j); }
};
static_assert(DgfTaps<4>::i0(0) == 0 && DgfTaps<4>::i0(2) == 1 && DgfTaps<4>::i0(5) == 1 && DgfTaps<4>::i0(7) == 2, "taps");
static_assert(DgfTaps<4>::lx(0) == 0.625f && DgfTaps<4>::lx(3) == 0.375f && DgfTaps<4>::lx(6) == 0.125f, "taps");
static_assert(DgfTaps<8>::i0(3) == 0 && DgfTaps<8>::i0(4) == 1 && DgfTaps<8>::lx(0) == 0.5625f && DgfTaps<8>::lx(7) == 0.4375f, "taps");

template <typename TIO, int R, int KX>
__global__ __launch_bounds__(256) void dgf_apply_pow2(DgfApplyArgs a) {
    [[maybe_unused]] Stamper stamp;
    typedef typename IoV8<TIO>::type V8;
    typedef DgfTaps<KX> X;
    constexpr int NR = R == 1 ? 2 : 3;                                  // low-res rows the block's R rows can touch
    extern __shared__ __attribute__((aligned(16))) f32x4 colAB[];       // [R][ncols][2]
    const int tid = threadIdx.x;
    const int seg0 = blockIdx.x * 2048;
    const int x0 = seg0 + tid * 8;
    const int y0 = blockIdx.y * R, n = blockIdx.z;
    const size_t plane = (size_t)a.H * a.W;
    const bool active = x0 < a.W;                                       // (W % 8 == 0: an active thread has all 8 pixels)
    V8 s[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = min(y0 + r, a.H - 1);                             // (rows past the image: loaded again, never stored)
        const TIO* sp = (const TIO*)a.src + (size_t)n * 3 * plane + (size_t)y * a.W + min(x0, a.W - 8);
#pragma unroll
        for (int c = 0; c < 3; ++c) s[r][c] = *(const V8*)(sp + c * plane);    // issued first: in flight during the LDS phase
    }
    // ---- vertical geometry of the R rows (uniform); low-res rows L0 .. L0 + NR - 1 cover them (host-checked for R = 2)
    int d0[R], d1[R];
    float ly[R];
    int L0 = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = min(y0 + r, a.H - 1);
        float sy = a.scale_y * (y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        int iy0 = (int)sy; if (iy0 > a.h - 1) iy0 = a.h - 1;
        const int iy1 = iy0 + (iy0 < a.h - 1);
        if (r == 0) L0 = iy0;
        d0[r] = iy0 - L0; d1[r] = iy1 - L0; ly[r] = sy - iy0;
    }
    const int cbase = max(seg0 / KX - 1, 0);                            // first column of the segment's LDS tile
    const f32x4* rows[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) rows[m] = (const f32x4*)(a.ab + ((size_t)n * a.h + min(L0 + m, a.h - 1)) * a.w * 8);
    const int rstride = 2 * a.ncols;                                    // f32x4 per interpolated row
    auto vrow = [&](int i, const f32x4 (&raw)[NR]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f32x4 t = raw[0], u = raw[1];
            if constexpr (NR == 3) {
                t = d0[r] == 0 ? raw[0] : (d0[r] == 1 ? raw[1] : raw[2]);
                u = d1[r] == 0 ? raw[0] : (d1[r] == 1 ? raw[1] : raw[2]);
            } else {
                if (d1[r] == 0) u = raw[0];                             // (bottom border: iy1 == iy0)
            }
            colAB[r * rstride + i] = dgf_lerp(t, u, ly[r]);
        }
    };
    {
        constexpr int SU = 2048 / KX * 2 / 256 + 1;                     // (column, half) items per thread: 2 (2048 / KX + 3) / 256
        f32x4 raw[SU][NR];
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            const int i = tid + 256 * k;
            if (256 * k < rstride) {                                    // uniform
                const int col = min(cbase + (i >> 1), a.w - 1);
#pragma unroll
                for (int m = 0; m < NR; ++m) raw[k][m] = rows[m][col * 2 + (i & 1)];
            }
        }
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            const int i = tid + 256 * k;
            if (i < rstride) vrow(i, raw[k]);
        }
    }
    __syncthreads();
    if (!active) return;
    // the thread's columns x0 / KX - 1 ... (clamped to the map, like the general kernel's ix0 / ix1)
    int kc[X::NC];
#pragma unroll
    for (int m = 0; m < X::NC; ++m) kc[m] = 2 * (min(max(x0 / KX - 1 + m, 0), a.w - 1) - cbase);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (y0 + r < a.H) {                                             // (uniform)
            const f32x4* cr = colAB + r * rstride;
            f32x4 CA[X::NC], CB[X::NC];
#pragma unroll
            for (int m = 0; m < X::NC; ++m) { CA[m] = cr[kc[m]]; CB[m] = cr[kc[m] + 1]; }
            Out8<TIO> of[3], op;
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                float o[2][4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * jp + e;
                    constexpr int dummy = 0; (void)dummy;
                    const int i0 = X::i0(j);
                    // left border: sx clamps to 0 there, i.e. fraction 0 on column 0
                    const float lx = (X::i0(j) == 0 && x0 == 0) ? 0.f : X::lx(j);
                    const f32x4 A = dgf_lerp(CA[i0], CA[i0 + 1], lx);
                    const f32x4 B = dgf_lerp(CB[i0], CB[i0 + 1], lx);
                    dgf_pixel(A, B, IoV8<TIO>::get(s[r][0], j), IoV8<TIO>::get(s[r][1], j), IoV8<TIO>::get(s[r][2], j), o[e]);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) of[c].set(jp, o[0][c], o[1][c]);
                op.set(jp, o[0][3], o[1][3]);
            }
            dgf_emit<TIO>(a, n, y0 + r, x0, true, plane, of, op);
        }
    }
}

// Which kernel a geometry gets: DgfApplyArgs.kx = 4 / 8 where the power-of-two kernel applies (else 0: the general kernel),
// .rows = its rows per block.  RVM_DGF_POW2=0 forces the general kernel, RVM_DGF_ROWS=1|2 the rows (test hooks).
void dgf_apply_pick(DgfApplyArgs& a) {
    a.kx = 0; a.rows = 1;
    if (const char* e = getenv("RVM_DGF_POW2")) { if (atoi(e) == 0) return; }
    for (int kx : {4, 8}) {
        if (a.W != kx * a.w || a.H != kx * a.h || (a.W & 7) || a.scale_x != 1.f / kx || a.scale_y != 1.f / kx) continue;
        if (a.ncols < std::min(a.w, std::min(2048, a.W) / kx + 3)) continue;
        a.kx = kx;
        a.rows = 2;                                                     // H = kx h: rows 2i, 2i + 1 never span more than three low-res rows
        if (const char* e = getenv("RVM_DGF_ROWS")) { if (atoi(e) == 1) a.rows = 1; }
        return;
    }
}
template <typename TIO, int R, int KX> static void dgf_apply_launch(const DgfApplyArgs& a, int N, hipStream_t s) {
    const dim3 grid((a.W + 2047) / 2048, (a.H + R - 1) / R, N);
    hipLaunchKernelGGL((dgf_apply_pow2<TIO, R, KX>), grid, dim3(256), (size_t)R * a.ncols * 32, s, a);
}
template <typename TIO> static void dgf_apply_dispatch(const DgfApplyArgs& a, int N, hipStream_t s) {
    if (a.kx == 4) { if (a.rows == 2) dgf_apply_launch<TIO, 2, 4>(a, N, s); else dgf_apply_launch<TIO, 1, 4>(a, N, s); return; }
    if (a.kx == 8) { if (a.rows == 2) dgf_apply_launch<TIO, 2, 8>(a, N, s); else dgf_apply_launch<TIO, 1, 8>(a, N, s); return; }
    const dim3 grid((a.W + 2047) / 2048, a.H, N);
    hipLaunchKernelGGL((dgf_apply<TIO>), grid, dim3(256), (size_t)a.ncols * 32, s, a);
}
template <typename T> void launch_dgf_apply(const DgfApplyArgs& a, int N, int io, hipStream_t s) {
    if (sizeof(T) == 2 || io == IO_F16) dgf_apply_dispatch<f16>(a, N, s);
    else if (io == IO_BF16) dgf_apply_dispatch<bf16>(a, N, s);
    else dgf_apply_dispatch<float>(a, N, s);
}
template void launch_dgf_apply<f16>(const DgfApplyArgs&, int, int, hipStream_t);
template void launch_dgf_apply<float>(const DgfApplyArgs&, int, int, hipStream_t);

template <typename TIO> static void dgf_apply_attr() {
    (void)hipFuncSetAttribute((const void*)dgf_apply<TIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}
void dgf_apply_init() { dgf_apply_attr<f16>(); dgf_apply_attr<float>(); dgf_apply_attr<bf16>(); }
