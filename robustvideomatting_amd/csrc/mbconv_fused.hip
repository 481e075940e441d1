// A whole MobileNetV3 inverted-residual block WITHOUT squeeze-excite in one launch (gfx950):
//   1x1 expand + BN + act  ->  depthwise 3x3 (stride 1 / 2) + BN + act  ->  1x1 project + BN  [+ block input]
// Replaces torchvision InvertedResidual as configured by the reference (model/mobilenetv3.py:11-12,16-19: features[2],
// [3], [7], [8], [9], [10]; forward model/mobilenetv3.py:39-58).  The two-launch form (mbconv_ed + pw_gemm) sends the
// depthwise output through memory and pays a second dependent launch; here a block of 8 waves owns a tile of
// THo x TWo output pixels with ALL channels:
//   X  the tile's input halo ((THo-1)s+3) x ((TWo-1)s+3) pixels x Cin goes to LDS as MFMA pixel fragments (the split
//      mode stores (hi, lo) f16 pairs), one round of loads; every weight the block needs is requested in the same
//      round, straight into registers (wave w: its expand n-tile, its project n-tile / K part), so the block pays
//      ONE round trip to L2 for all its operands
//   E  wave w = (n-tile, m-group) multiplies its weight fragments against the halo's m-tiles out of LDS; bias + act;
//      pixels outside the image become ZERO (the depthwise conv pads the EXPANDED tensor); result [quad][pixel] fp32
//   D  a thread owns (channel quad, output pixel): nine taps out of LDS, bias + act, written as the project GEMM's
//      pixel fragments
//   P  wave w = (n-tile, K part): split-K over the waves, fixed-order sum through LDS, bias (+ residual), NHWC store
// Halo recomputation costs MFMA time the launch has to spare; what it buys is one launch and one trip through L2 less
// per block.
#include "device_common.hpp"
#include <algorithm>

// Eight waves per block, two per SIMD: measured on the 16-wave form of this kernel, the block was bound by INSTRUCTION ISSUE -
// four waves share a SIMD's issue slots, so the ~500 instructions of address arithmetic and staging every wave runs before
// the first barrier alone took 2.3 us; half the waves, each with twice the n-tiles, halve that overhead and the LDS
// fragment reads (one pixel fragment now feeds two n-tiles) at unchanged MFMA work.
constexpr int MBF_NWV = 8, MBF_NTH = 64 * MBF_NWV;
constexpr int MBF_NPW = 2;           // expand n-tiles per wave, at most (NTe <= 16)
constexpr int MBF_XU = 4;            // staging loads per thread: one round covers the padded halo (host-checked)
constexpr int MBF_MTP = 2;           // m-tiles (16 output pixels) of a tile, at most
constexpr int MBF_NTP = 5;           // project n-tiles (Cout <= 80), at most
constexpr int MBF_PREW = 144 + 16 + 256 + 16;   // floats of the folded features[1]: depthwise [9][16] + bias, project [16][16] + bias

// Geometry and LDS map of a tile, derived once on the host: every instruction of address arithmetic in the kernel is paid
// by all the waves that share a SIMD.
bool mbf_prepare(MbfArgs& a, int es) {
    if (a.NTe < 1 || a.NTe > MBF_NWV * MBF_NPW || a.KSe < 1 || a.KSe > 3 || a.NTp < 1 || a.NTp > MBF_NTP || a.KSp < 1 || a.KSp > MBF_NWV) return false;
    if ((a.C & 3) || (a.Cin & 3) || (a.Cout & 3) || (a.stride != 1 && a.stride != 2)) return false;
    if (!mbf_supported(a.stride, a.KSe, a.act)) return false;
    if (a.NTp > (a.KSe == 1 ? 2 : MBF_NTP)) return false;               // n-tiles of the instantiation (mbf_dispatch)
    if (a.THo < 2 || a.TWo < 4 || (a.TWo & (a.TWo - 1)) || (a.THo & (a.THo - 1))) return false;
    const int pad = es == 2 ? 8 : 4;
    a.THi = (a.THo - 1) * a.stride + 3; a.TWi = (a.TWo - 1) * a.stride + 3;
    a.NPXi = a.THi * a.TWi; a.MTe = (a.NPXi + 15) >> 4; a.npx = a.MTe * 16;
    a.TP = a.THo * a.TWo; a.MTp = a.TP >> 4;
    if (a.TP != 16 && a.TP != 32) return false;                         // MBF_MTP m-tiles at most
    if (a.MTe > 32) return false;                                       // one bit per m-tile in the lane's in-image mask
    a.CSLx = a.KSe * 32 + pad; a.CSLd = a.KSp * 32 + pad;
    a.Cq = a.NTe * 16;
    a.KP = a.KSp;                                                       // project: one K step per wave, all n-tiles
    a.ngE = a.NTe <= MBF_NWV ? MBF_NWV / a.NTe : 1;
    a.padq = (a.KSp * 32 - a.C) >> 2;
    a.twl = 31 - __builtin_clz((unsigned)a.TWo);
    a.hpl = 31 - __builtin_clz((unsigned)(a.TP >> 1));
    a.invTWi = 0xFFFFFFFFu / (unsigned)a.TWi + 1u;                      // px / TWi == umulhi(px, invTWi) for px < 2^16
    a.invCq = 0xFFFFFFFFu / (unsigned)(a.Cq >> 2) + 1u;                 // i / (Cq / 4)
    a.mulE = 65536 / a.NTe + 1; a.mulP = 0;                             // wave / NTe == (wave * mulE) >> 16 for wave < 16
    const size_t off_e = ((size_t)a.npx * a.CSLx * es + 15) & ~(size_t)15;
    const size_t off_d = off_e + (size_t)a.NTe * 4 * a.npx * 16;
    const size_t off_w = off_d + (((size_t)a.TP * a.CSLd * es + 15) & ~(size_t)15);
    const size_t total = off_w + (size_t)10 * a.Cq * 4;
    const size_t red = (size_t)a.KP * a.NTp * a.MTp * 1024;
    if (a.npx * a.KSe * 8 > MBF_NTH * MBF_XU) return false;              // one staging round
    if (10 * (a.Cq >> 2) > MBF_NTH * 2) return false;                    // depthwise weights: two 16-byte loads per thread
    if (red > off_d) return false;                                       // the split-K sums reuse the dead X / E areas
    if (total > 160 * 1024) return false;
    if (a.res && (a.stride != 1 || a.Cin != a.Cout)) return false;
    a.off_e = (int)off_e; a.off_d = (int)off_d; a.off_w = (int)off_w; a.lds_total = (int)total; a.red_bytes = (int)red;
    if (a.pre_x) {
        if (a.stride != 2 || a.KSe != 1 || a.Cin != 16 || a.act != ACT_RELU) return false;
        a.pre_pw = a.TWi + 2; a.pre_npx = (a.THi + 2) * a.pre_pw;
        a.inv_pre_pw = 0xFFFFFFFFu / (unsigned)a.pre_pw + 1u;
        if (a.pre_npx * 4 > MBF_NTH * MBF_XU) return false;              // one staging round
        if ((size_t)a.pre_npx * 64 + MBF_PREW * 4 > off_d - off_e) return false;    // the stem patch borrows the expanded halo's area
    }
    return true;
}

template <int ACT> __device__ __forceinline__ float mbf_act(float v) {
    if constexpr (ACT == ACT_RELU) return fmaxf(v, 0.f);
    // hardswish x * relu6(x + 3) / 6 with the division as a multiplication by 1/6 (a true fp32 division is ~10
    // instructions; twelve of them per lane and m-tile were a third of the expand phase) - one ulp, far inside every bound
    else return v * __builtin_amdgcn_fmed3f(v + 3.f, 0.f, 6.f) * (1.f / 6.f);
}

template <typename T, int S, int KSE, int ACT, int NTP, bool PRE = false>
__global__ __launch_bounds__(MBF_NTH) void mbconv_fused(MbfArgs a) {
    [[maybe_unused]] Stamper stamp;
    extern __shared__ __attribute__((aligned(16))) char mbf_smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
    T* xbuf = (T*)mbf_smem;                                           // [npx][CSLx]   pixel fragments of the input halo
    f32x4* ebuf = (f32x4*)(mbf_smem + a.off_e);                       // [NTe * 4][npx] expanded halo, fp32
    T* dbuf = (T*)(mbf_smem + a.off_d);                               // [TP][CSLd]    pixel fragments of the depthwise output
    float* wd = (float*)(mbf_smem + a.off_w);                         // [9][Cq] depthwise weights, [Cq] bias
    f32x4* red = (f32x4*)mbf_smem;                                    // [KP][NTp][MTp][64], over X / E once they are dead
    const int oy0 = ty * a.THo, ox0 = tx * a.TWo;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;                   // image position of halo pixel (0, 0)
    // ---- X: every global load the expand phase needs is requested here (clamped addresses, masked where used)
    constexpr int QX = KSE * 8;                                       // channel quads of a padded pixel row
    const int nitems = a.npx * QX, cinq = a.Cin >> 2;
    const T* xn = (const T*)a.x + (size_t)n * a.Hi * a.Wi * a.Cin;
    f32x4 xv[MBF_XU];
    bool xok[MBF_XU];
    [[maybe_unused]] f32x4 prew = f32x4{0, 0, 0, 0};
    if constexpr (PRE) {
        // the stem patch one pixel around the halo (features[1]'s depthwise taps), and features[1]'s weights
        const T* sn = (const T*)a.pre_x + (size_t)n * a.Hi * a.Wi * 16;
#pragma unroll
        for (int u = 0; u < MBF_XU; ++u) {
            const int it = tid + u * MBF_NTH, p = it >> 2, q = it & 3;
            const int py = (int)__umulhi((unsigned)p, a.inv_pre_pw), px = p - py * a.pre_pw;
            const int iy = iy0 - 1 + py, ix = ix0 - 1 + px;
            xok[u] = p < a.pre_npx && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            xv[u] = ld4f(sn + (xok[u] ? (iy * a.Wi + ix) * 16 + q * 4 : 0));
        }
        if (tid < MBF_PREW / 4) {
            const float* src = tid < 36 ? a.pre_dw_w + tid * 4 : (tid < 40 ? a.pre_dw_b + (tid - 36) * 4 : (tid < 104 ? a.pre_pj_w + (tid - 40) * 4 : a.pre_pj_b + (tid - 104) * 4));
            prew = *(const f32x4*)src;
        }
    } else {
#pragma unroll
        for (int u = 0; u < MBF_XU; ++u) {
            const int it = tid + u * MBF_NTH, row = it / QX, q = it - row * QX;
            const int ry = (int)__umulhi((unsigned)row, a.invTWi), rx = row - ry * a.TWi;
            const int iy = iy0 + ry, ix = ix0 + rx;
            xok[u] = row < a.NPXi && q < cinq && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            xv[u] = ld4f(xn + (xok[u] ? (iy * a.Wi + ix) * a.Cin + q * 4 : 0));
        }
    }
    // expand weights.  NTe <= 8: wave = (n-tile, m-group), one n-tile; else wave w owns n-tiles w and w + 8, all m-tiles
    const bool two = a.NTe > MBF_NWV;
    const int mgE = two ? 0 : (wave * a.mulE) >> 16;
    const int nt0 = two ? wave : wave - mgE * a.NTe;
    const bool actE = mgE < a.ngE;
    const bool act1 = two && wave + MBF_NWV < a.NTe;                   // the second n-tile exists
    Frag<T> wfe[MBF_NPW][KSE];
    f32x4 eb[MBF_NPW];
#pragma unroll
    for (int j = 0; j < MBF_NPW; ++j) {
        const int nt = min(nt0 + j * MBF_NWV, a.NTe - 1);             // (an absent tile repeats the last one: never used)
#pragma unroll
        for (int ks = 0; ks < KSE; ++ks) wfe[j][ks].loadw((const T*)a.we + (nt * KSE + ks) * 512, lane);
        eb[j] = *(const f32x4*)(a.ebias + nt * 16 + g * 4);
    }
    // depthwise weights + bias: [10][Cq] floats through registers into LDS, 16 bytes at a time
    f32x4 dwr[2];
    const int cq4 = a.Cq >> 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = min(tid + u * MBF_NTH, 10 * cq4 - 1), tap = (int)__umulhi((unsigned)i, a.invCq), c = (i - tap * cq4) * 4;
        const float* src = tap < 9 ? a.dw_w + tap * a.C : a.dw_b;
        const f32x4 v = *(const f32x4*)(src + min(c, a.C - 4));
        dwr[u] = c < a.C ? v : f32x4{0, 0, 0, 0};
    }
    // which of this lane's halo pixels (one per m-tile: mt * 16 + l15) lie inside the image
    unsigned inbmask;
    if (iy0 >= 0 && ix0 >= 0 && iy0 + a.THi <= a.Hi && ix0 + a.TWi <= a.Wi) {      // interior tile (uniform): only the tail is masked
        const int tail = a.NPXi - (a.MTe - 1) * 16;
        inbmask = (l15 < tail ? 1u << (a.MTe - 1) : 0u) | ((1u << (a.MTe - 1)) - 1u);
    } else {
        inbmask = 0;
        for (int mt = 0; mt < a.MTe; ++mt) {
            const int px = mt * 16 + l15;
            const int ry = (int)__umulhi((unsigned)px, a.invTWi), rx = px - ry * a.TWi;
            const bool inb = px < a.NPXi && (unsigned)(iy0 + ry) < (unsigned)a.Hi && (unsigned)(ix0 + rx) < (unsigned)a.Wi;
            inbmask |= (inb ? 1u : 0u) << mt;
        }
    }
    // project: wave w < KSp multiplies K step w against ALL n-tiles (one pixel fragment read feeds NTP independent
    // chains); the partial sums meet in LDS and wave nt < NTp finishes n-tile nt
    const bool actP = wave < a.KSp;
    // ---- halo to LDS; depthwise weights to LDS; zero the K padding of the depthwise output's fragments
    if constexpr (PRE) {
        f32x4* sbuf = (f32x4*)(mbf_smem + a.off_e);                   // [pre_npx][4 quads] stem patch, fp32; then the weights
#pragma unroll
        for (int u = 0; u < MBF_XU; ++u) {
            const int it = tid + u * MBF_NTH;
            if (it < a.pre_npx * 4) sbuf[it] = xok[u] ? xv[u] : f32x4{0, 0, 0, 0};     // zero outside the image: the depthwise conv's padding
        }
        if (tid < MBF_PREW / 4) sbuf[a.pre_npx * 4 + tid] = prew;
        // K padding of the input fragments (channels 16..31) and the m-tile tail
        for (int it = tid; it < nitems; it += MBF_NTH) {
            const int row = it >> 3, q = it & 7;
            if (q >= 4 || row >= a.NPXi) stp4(xbuf + row * a.CSLx, q * 4, f32x4{0, 0, 0, 0});
        }
    } else {
#pragma unroll
        for (int u = 0; u < MBF_XU; ++u) {
            const int it = tid + u * MBF_NTH, row = it / QX, q = it - row * QX;
            if (it < nitems) stp4(xbuf + row * a.CSLx, q * 4, xok[u] ? xv[u] : f32x4{0, 0, 0, 0});
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int i = tid + u * MBF_NTH; if (i < 10 * cq4) *(f32x4*)(wd + i * 4) = dwr[u]; }
    if (a.padq) {
        const unsigned invp = 0xFFFFFFFFu / (unsigned)a.padq + 1u;      // padq in 1..7
        for (int i = tid; i < a.TP * a.padq; i += MBF_NTH) {
            const int p = a.padq > 1 ? (int)__umulhi((unsigned)i, invp) : i, q = i - p * a.padq;
            stp4(dbuf + p * a.CSLd, a.C + q * 4, f32x4{0, 0, 0, 0});
        }
    }
    if constexpr (PRE) {
        // ---- features[1] on the halo: item = (pixel, half of the 16 output channels); depthwise 3x3 + ReLU on all 16
        //      channels, then the 16 x 16 project in fp32 + bias + the stem value (mobilenetv3.py: InvertedResidual without
        //      expand, residual) -> the expand's input fragments, and to memory for the pixels this tile owns (all of the halo
        //      but its first row and column, which the neighbouring tiles own)
        __syncthreads();
        const f32x4* sbuf = (const f32x4*)(mbf_smem + a.off_e);
        const f32x4* pw4 = sbuf + a.pre_npx * 4;                       // [36] dw, [4] dw bias, [64] project, [4] bias
        T* f1n = (T*)a.pre_y + (size_t)n * a.Hi * a.Wi * 16;
        for (int it = tid; it < a.NPXi * 2; it += MBF_NTH) {
            const int p = it >> 1, h = it & 1;
            const int ry = (int)__umulhi((unsigned)p, a.invTWi), rx = p - ry * a.TWi;
            const f32x4* s0 = sbuf + (ry * a.pre_pw + rx) * 4;         // top-left tap
            float d[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 acc = pw4[36 + q];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc += s0[(ky * a.pre_pw + kx) * 4 + q] * pw4[(ky * 3 + kx) * 4 + q];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d[q * 4 + r] = fmaxf(acc[r], 0.f);
                    if constexpr (sizeof(T) == 2) d[q * 4 + r] = (float)(f16)d[q * 4 + r];      // as the depthwise output would be stored
                }
            }
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool inb = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int qo = 2 * h + e;
                f32x4 o = pw4[104 + qo];
#pragma unroll
                for (int ci = 0; ci < 16; ++ci) o += d[ci] * pw4[40 + ci * 4 + qo];
                o += s0[(a.pre_pw + 1) * 4 + qo];                       // residual: the block's input
                stp4(xbuf + p * a.CSLx, qo * 4, inb ? o : f32x4{0, 0, 0, 0});
                if (inb && ry >= 1 && rx >= 1) st4f(f1n + (iy * a.Wi + ix) * 16 + qo * 4, o);
            }
        }
    }
    __syncthreads();
    stamp(1);
    // project weights, requested only now: they queue BEHIND every wave's halo pixels and expand weights in the CU's
    // memory pipeline (what the expand phase waits for) and have the expand and depthwise phases to arrive.  Always
    // NTP loads (absent tiles / steps repeat the last one): a load count known at compile time lets the expand phase
    // wait for its own weights only.
    Frag<T> wfp[NTP];
#pragma unroll
    for (int j = 0; j < NTP; ++j)
        wfp[j].loadw((const T*)a.wp + (min(j, a.NTp - 1) * a.KSp + min(wave, a.KSp - 1)) * 512, lane);
    // ---- E: expand the halo
    if (actE) {
        auto finish = [&](int mt, int nt, f32x4 acc, f32x4 bias) {
            const bool inb = (inbmask >> mt) & 1u;
            f32x4 v = acc + bias;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float t = mbf_act<ACT>(v[r]); v[r] = inb ? t : 0.f; }
            ebuf[(nt * 4 + g) * a.npx + mt * 16 + l15] = v;
        };
        auto tile = [&](int mt, const Frag<T> (&xf)[KSE]) {
            f32x4 acc0 = f32x4{0, 0, 0, 0}, acc1 = acc0;
            if (act1) {
#pragma unroll
                for (int ks = 0; ks < KSE; ++ks) { mma(acc0, wfe[0][ks], xf[ks]); mma(acc1, wfe[1][ks], xf[ks]); }
                finish(mt, nt0, acc0, eb[0]); finish(mt, nt0 + MBF_NWV, acc1, eb[1]);
            } else {
#pragma unroll
                for (int ks = 0; ks < KSE; ++ks) mma(acc0, wfe[0][ks], xf[ks]);
                finish(mt, nt0, acc0, eb[0]);
            }
        };
        const T* xr = xbuf + l15 * a.CSLx + g * 8;
        const int mstep = 16 * a.CSLx;
        // one m-tile at a time, its fragments read while the previous tile is multiplied (two register sets)
        Frag<T> xa[KSE], xb[KSE];
        int mt = mgE;
#pragma unroll
        for (int ks = 0; ks < KSE; ++ks) xa[ks].load_lds(xr + mt * mstep + ks * 32);
        for (; mt < a.MTe; mt += 2 * a.ngE) {
            const bool has1 = mt + a.ngE < a.MTe;
            if (has1) {
#pragma unroll
                for (int ks = 0; ks < KSE; ++ks) xb[ks].load_lds(xr + (mt + a.ngE) * mstep + ks * 32);
            }
            tile(mt, xa);
            if (mt == mgE) stamp(2);
            if (!has1) break;
            if (mt + 2 * a.ngE < a.MTe) {
#pragma unroll
                for (int ks = 0; ks < KSE; ++ks) xa[ks].load_lds(xr + (mt + 2 * a.ngE) * mstep + ks * 32);
            }
            tile(mt + a.ngE, xb);
        }
    }
    // project epilogue operands (bias, residual) of the waves that finish an n-tile: in flight during the depthwise pass
    const bool fin = wave < a.NTp;
    const int chP = wave * 16 + g * 4;
    f32x4 pb = f32x4{0, 0, 0, 0}, e_res[MBF_MTP];
    int opix[MBF_MTP];                                                // output element index, -1: outside the image / tile
#pragma unroll
    for (int mi = 0; mi < MBF_MTP; ++mi) { e_res[mi] = f32x4{0, 0, 0, 0}; opix[mi] = -1; }
    if (fin) {
        pb = *(const f32x4*)(a.pbias + chP);
#pragma unroll
        for (int mi = 0; mi < MBF_MTP; ++mi) {
            const int p = mi * 16 + l15, oy = oy0 + (p >> a.twl), ox = ox0 + (p & (a.TWo - 1));
            if (mi < a.MTp && oy < a.Ho && ox < a.Wo && chP < a.Cout) opix[mi] = ((n * a.Ho + oy) * a.Wo + ox) * a.Cout + chP;
            if (a.res) {
                const f32x4 t = ld4f((const T*)a.res + max(opix[mi], 0));
                if (opix[mi] >= 0) e_res[mi] = t;
            }
        }
    }
    stamp(3);
    __syncthreads();
    stamp(4);
    // ---- D: depthwise 3x3 out of the expanded halo; a thread owns one channel quad of two vertically adjacent output
    //      pixels (rows 2yp, 2yp + 1: they share 2 of 3 input rows at stride 1, 1 at stride 2, and all nine weights)
    {
        const int items = (a.C >> 2) << a.hpl, HP = a.TP >> 1;
        for (int it = tid; it < items; it += MBF_NTH) {
            const int q = it >> a.hpl, pr = it & (HP - 1), ox = pr & (a.TWo - 1), yp = pr >> a.twl;
            const f32x4* e = ebuf + q * a.npx + (2 * yp * S) * a.TWi + ox * S;
            const float* w = wd + q * 4;
            f32x4 wv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wv[t] = *(const f32x4*)(w + t * a.Cq);
            f32x4 acc0 = *(const f32x4*)(w + 9 * a.Cq), acc1 = acc0;
#pragma unroll
            for (int r = 0; r < 3 + S; ++r) {
                const f32x4 e0 = e[r * a.TWi], e1 = e[r * a.TWi + 1], e2 = e[r * a.TWi + 2];
                if (r < 3) acc0 += e0 * wv[r * 3] + e1 * wv[r * 3 + 1] + e2 * wv[r * 3 + 2];
                if (r >= S) acc1 += e0 * wv[(r - S) * 3] + e1 * wv[(r - S) * 3 + 1] + e2 * wv[(r - S) * 3 + 2];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[r] = mbf_act<ACT>(acc0[r]); acc1[r] = mbf_act<ACT>(acc1[r]); }
            T* d0 = dbuf + ((2 * yp) * a.TWo + ox) * a.CSLd;
            stp4(d0, q * 4, acc0);
            stp4(d0 + a.TWo * a.CSLd, q * 4, acc1);
        }
    }
    stamp(5);
    __syncthreads();
    stamp(6);
    // ---- P: project
    if (actP) {
        const T* dr = dbuf + l15 * a.CSLd + g * 8 + wave * 32;
#pragma unroll
        for (int mi = 0; mi < MBF_MTP; ++mi) {
            if (mi < a.MTp) {
                Frag<T> xf; xf.load_lds(dr + mi * 16 * a.CSLd);
#pragma unroll
                for (int j = 0; j < NTP; ++j) {
                    if (j < a.NTp) {
                        f32x4 pacc = f32x4{0, 0, 0, 0};
                        mma(pacc, wfp[j], xf);
                        red[((wave * a.NTp + j) * a.MTp + mi) * 64 + lane] = pacc;
                    }
                }
            }
        }
    }
    __syncthreads();
    f32x4 acc[MBF_MTP];
    if (fin) {
#pragma unroll
        for (int mi = 0; mi < MBF_MTP; ++mi) {
            acc[mi] = f32x4{0, 0, 0, 0};
            if (mi < a.MTp) {
                // all K parts requested together (clamped, masked), added in a fixed order
                f32x4 part[MBF_NWV];
#pragma unroll
                for (int k = 0; k < MBF_NWV; ++k) part[k] = red[((min(k, a.KSp - 1) * a.NTp + wave) * a.MTp + mi) * 64 + lane];
                f32x4 sum = part[0];
#pragma unroll
                for (int k = 1; k < MBF_NWV; ++k) sum += k < a.KSp ? part[k] : f32x4{0, 0, 0, 0};
                acc[mi] = sum;
            }
        }
    }
    stamp(7);
    if (!fin) return;
#pragma unroll
    for (int mi = 0; mi < MBF_MTP; ++mi)
        if (opix[mi] >= 0) st4f((T*)a.y + opix[mi], acc[mi] + pb + e_res[mi]);
    stamp(8);
}

// the (stride, expand K steps, activation) combinations of the reference's table (model/mobilenetv3.py:11-12,16-19):
// features[2]: s2, 16 in (1 step), ReLU; [3]: s1, 24 in (1), ReLU; [7]: s2, 40 in (2), hardswish; [8..10]: s1, 80 in (3), hardswish.
// Last template argument: project n-tiles at most (24 / 80 output channels)
template <typename T> static bool mbf_dispatch(const MbfArgs& a, const dim3& grid, hipStream_t s, bool launch) {
#define RVM_MBF_CASE(S_, K_, A_, P_) \
    if (a.stride == S_ && a.KSe == K_ && a.act == A_) { \
        if (launch) hipLaunchKernelGGL((mbconv_fused<T, S_, K_, A_, P_>), grid, dim3(MBF_NTH), (size_t)a.lds_total, s, a); \
        else (void)hipFuncSetAttribute((const void*)mbconv_fused<T, S_, K_, A_, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        return true; }
    if (a.stride == 2 && a.KSe == 1 && a.act == ACT_RELU && (a.pre_x || !launch)) {
        if (launch) hipLaunchKernelGGL((mbconv_fused<T, 2, 1, ACT_RELU, 2, true>), grid, dim3(MBF_NTH), (size_t)a.lds_total, s, a);
        else (void)hipFuncSetAttribute((const void*)mbconv_fused<T, 2, 1, ACT_RELU, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (launch) return true;
    }
    RVM_MBF_CASE(2, 1, ACT_RELU, 2) RVM_MBF_CASE(1, 1, ACT_RELU, 2) RVM_MBF_CASE(2, 2, ACT_HSWISH, 5) RVM_MBF_CASE(1, 3, ACT_HSWISH, 5)
#undef RVM_MBF_CASE
    return false;
}
bool mbf_supported(int stride, int KSe, int act) {
    return (stride == 2 && KSe == 1 && act == ACT_RELU) || (stride == 1 && KSe == 1 && act == ACT_RELU) ||
           (stride == 2 && KSe == 2 && act == ACT_HSWISH) || (stride == 1 && KSe == 3 && act == ACT_HSWISH);
}
template <typename T> void launch_mbf(const MbfArgs& a, int N, hipStream_t s) {
    mbf_dispatch<T>(a, dim3(a.tiles_x, a.tiles_y, N), s, true);
}
template void launch_mbf<f16>(const MbfArgs&, int, hipStream_t);
template void launch_mbf<float>(const MbfArgs&, int, hipStream_t);
template void launch_mbf<f32s>(const MbfArgs&, int, hipStream_t);

template <typename T> static void mbf_attr() {
    MbfArgs a = MbfArgs();
    const int combos[4][3] = {{2, 1, ACT_RELU}, {1, 1, ACT_RELU}, {2, 2, ACT_HSWISH}, {1, 3, ACT_HSWISH}};
    for (auto& c : combos) { a.stride = c[0]; a.KSe = c[1]; a.act = c[2]; mbf_dispatch<T>(a, dim3(1), nullptr, false); }
}
void mbf_init() { mbf_attr<f16>(); mbf_attr<float>(); mbf_attr<f32s>(); }
