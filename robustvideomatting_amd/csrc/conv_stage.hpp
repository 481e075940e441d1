// Staging of the concatenated conv input (up to four sources, any of which a x2 bilinear upsample) into an LDS halo patch,
// shared by conv_kxk (kernels.hip) and conv_cat (conv_cat.hip).
#pragma once
#include "device_common.hpp"

// Stage channel quads [q_lo, q_lo+q_cnt) of one source of the concatenated conv input into the
// LDS halo patch (origin gy0,gx0 in source-resolution coordinates), by the `nthr` threads (whole waves)
// the launcher gave this source; `tl` is the thread's index among them.  The sources of a block are
// staged by different waves at the same time, so the patch costs the L2 round trips of its slowest wave
// group instead of the sum over the sources.  Items are walked as one flat index, UN per thread per round,
// with a branch-free body (clamped address + select), so 8 independent loads are in flight per thread.
//   direct source:  item = (pixel, quad), 1 load.
//   x2 bilinear (align_corners=False, decoder.py:74), patch with even PH, PW on an odd origin (3x3 stride-1
//                   tiles: origin = even tile corner - 1): item = (2x2 output block, quad); the outputs
//                   (2i+1, 2i+2) x (2j+1, 2j+2) interpolate between the SAME source pixels (i, i+1) x (j, j+1),
//                   4 loads for 4 outputs.  Weights, clamps and zero padding as in the per-pixel form.
template <typename T, bool UP, bool HF, int PH, int PW, int UND = 4, int UNU = 2>
__device__ __forceinline__ void stage_source(T* patch, const ConvSrc S, int q_lo, int q_cnt, int dst_c, int CSL, int n,
                                             int gy0, int gx0, int H, int W, int tl, int nthr) {
    const int QN = q_cnt;
    const unsigned inv = 0xFFFFFFFFu / (unsigned)QN + 1u;   // it / QN == umulhi(it, inv) for it < 2^16 .. see below
    constexpr bool hf = HF;                                  // compile-time: keeps the loads of a round branch-free
    const bool live = S.p != nullptr;
    if (!live) {                                             // no state yet (first frame): zeros
        const int total = PH * PW * QN;
        for (int it = tl; it < total; it += nthr) {
            const int pp = QN == 1 ? it : (int)__umulhi((unsigned)it, inv), q = it - pp * QN;
            stp4(patch + pp * CSL, dst_c + q * 4, f32x4{0, 0, 0, 0});
        }
        return;
    }
    const void* base = S.p;
    if constexpr (UP && PH % 2 == 0 && PW % 2 == 0) {
        if ((gy0 & 1) && (gx0 & 1)) {
            constexpr int BH = PH / 2, BW = PW / 2, UN = UNU;
            const int total = BH * BW * QN;
            const int i0 = (gy0 - 1) >> 1, j0 = (gx0 - 1) >> 1;
            const size_t nb = (size_t)n * S.Hs * S.Ws * S.Cs;
            const bool edge = gy0 < 0 || gx0 < 0 || gy0 + PH > H || gx0 + PW > W;     // (uniform per block)
            for (int it0 = tl; it0 < total; it0 += nthr * UN) {
                f32x4 s[UN][4];
                int dst[UN], dch[UN], by_[UN], bx_[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int it = min(it0 + u * nthr, total - 1);
                    const int bb = QN == 1 ? it : (int)__umulhi((unsigned)it, inv), q = it - bb * QN;
                    const int by = bb / BW, bx = bb - by * BW;
                    const int r0 = min(max(i0 + by, 0), S.Hs - 1), r1 = min(max(i0 + by + 1, 0), S.Hs - 1);
                    const int c0 = min(max(j0 + bx, 0), S.Ws - 1), c1 = min(max(j0 + bx + 1, 0), S.Ws - 1);
                    const size_t b = nb + (q_lo + q) * 4;
                    s[u][0] = ld4x(base, b + ((size_t)r0 * S.Ws + c0) * S.Cs, hf);
                    s[u][1] = ld4x(base, b + ((size_t)r0 * S.Ws + c1) * S.Cs, hf);
                    s[u][2] = ld4x(base, b + ((size_t)r1 * S.Ws + c0) * S.Cs, hf);
                    s[u][3] = ld4x(base, b + ((size_t)r1 * S.Ws + c1) * S.Cs, hf);
                    dst[u] = (it0 + u * nthr < total) ? (2 * by * PW + 2 * bx) * CSL : -1;
                    dch[u] = dst_c + q * 4;
                    by_[u] = by; bx_[u] = bx;
                }
                // The block's outputs sit at fractions (1/4, 3/4) between its source rows / columns (odd output 2i+1 -> i + 1/4,
                // even 2i+2 -> i + 3/4; where the output lies outside the source's span the two clamped rows / columns are the
                // same pixel, so the fraction is immaterial): separable, a + f * (b - a) with compile-time f - 24 packed
                // fp32 operations per 16 values instead of 48 plus the per-output coordinate arithmetic; only tiles that
                // stick out of the image test their outputs against it.
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (dst[u] < 0) continue;
                    const f32x4 d0 = s[u][1] - s[u][0], d1 = s[u][3] - s[u][2];
                    f32x4 h0[2], e[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const float lx = m ? 0.75f : 0.25f;
                        h0[m] = s[u][0] + lx * d0;
                        e[m] = (s[u][2] + lx * d1) - h0[m];
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float ly = k ? 0.75f : 0.25f;
                        const int gy = gy0 + 2 * by_[u] + k;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int gx = gx0 + 2 * bx_[u] + m;
                            f32x4 o = h0[m] + ly * e[m];
                            if (edge && !(gy >= 0 && gy < H && gx >= 0 && gx < W)) o = f32x4{0, 0, 0, 0};
                            stp4(patch + dst[u] + (k * PW + m) * CSL, dch[u], o);
                        }
                    }
                }
            }
            return;
        }
    }
    constexpr int UN = UP ? 2 : UND;                         // loads in flight per thread (UND: direct sources)
    const int total = PH * PW * QN;
    for (int it0 = tl; it0 < total; it0 += nthr * UN) {
        f32x4 v[UN];
        int dst[UN], dch[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int it = min(it0 + u * nthr, total - 1);
            const int pp = QN == 1 ? it : (int)__umulhi((unsigned)it, inv), q = it - pp * QN;
            const int py = pp / PW, px = pp - py * PW;
            const int gy = gy0 + py, gx = gx0 + px;
            const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            dst[u] = (it0 + u * nthr < total) ? pp * CSL : -1;
            dch[u] = dst_c + q * 4;
            if (!UP) {
                v[u] = ld4x(base, (((size_t)n * H + cy) * W + cx) * S.Cs + (q_lo + q) * 4, hf);
            } else {
                // x2 bilinear, align_corners=False (decoder.py:74): src = (dst+0.5)/2-0.5, clamped at 0
                const float sy = fmaxf(cy * 0.5f - 0.25f, 0.f), sx = fmaxf(cx * 0.5f - 0.25f, 0.f);
                const int iy0 = (int)sy, ix0 = (int)sx;
                const float ly = sy - iy0, lx = sx - ix0;
                const int iy1 = iy0 + (iy0 < S.Hs - 1), ix1 = ix0 + (ix0 < S.Ws - 1);
                const size_t b = (size_t)n * S.Hs * S.Ws * S.Cs + (q_lo + q) * 4;
                const f32x4 v00 = ld4x(base, b + ((size_t)iy0 * S.Ws + ix0) * S.Cs, hf);
                const f32x4 v01 = ld4x(base, b + ((size_t)iy0 * S.Ws + ix1) * S.Cs, hf);
                const f32x4 v10 = ld4x(base, b + ((size_t)iy1 * S.Ws + ix0) * S.Cs, hf);
                const f32x4 v11 = ld4x(base, b + ((size_t)iy1 * S.Ws + ix1) * S.Cs, hf);
                v[u] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            }
            if (!inb) v[u] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (dst[u] >= 0) stp4(patch + dst[u], dch[u], v[u]);
    }
}

// Waves of a conv_kxk block per source, in proportion to the loads the source's part of the halo patch takes
// (the sources lie along a [0, 8) wave axis; a wave on a boundary stages both neighbours, one after the other).
inline void conv_assign_waves(ConvArgs& a, int PH, int PW, int und, int unu = 2) {
    if (a.nchunks > 1) {                                    // channel chunks cut across the sources: every wave stages every source
        for (int k = 0; k < a.nsrc; ++k) { a.src[k].w0 = 0; a.src[k].nw = 8; }
        return;
    }
    // Whole waves per source, no wave shared between two sources (a shared wave stages them one after the other: two
    // dependent round trips on the block's critical path).  A source's staging loop takes ceil(items / (threads x UN))
    // rounds of loads (stage_source: UN = 2 items of 4 loads for a 2x2-block upsampled source, UN = `und` single loads for a
    // direct one); of all ways to give the 8 waves to the sources take the one with the fewest rounds on the slowest
    // source, then the fewest in total.
    int items[4], un[4];
    for (int k = 0; k < a.nsrc; ++k) {
        const bool blocks = a.src[k].up && PH % 2 == 0 && PW % 2 == 0;
        items[k] = (a.src[k].C / 4) * (blocks ? (PH / 2) * (PW / 2) : PH * PW);
        un[k] = a.src[k].up ? unu : und;
    }
    int best[4] = {8, 0, 0, 0}, bestmax = 1 << 30, bestsum = 1 << 30, cur[4];
    const int n = a.nsrc;
    // compositions of 8 into n positive parts
    for (int c0 = 1; c0 <= 8; ++c0)
        for (int c1 = (n > 1 ? 1 : 0); c1 <= (n > 1 ? 8 : 0); ++c1)
            for (int c2 = (n > 2 ? 1 : 0); c2 <= (n > 2 ? 8 : 0); ++c2) {
                const int c3 = 8 - c0 - c1 - c2;
                if (n > 3 ? c3 < 1 : c3 != 0) continue;
                cur[0] = c0; cur[1] = c1; cur[2] = c2; cur[3] = c3;
                int mx = 0, sum = 0;
                for (int k = 0; k < n; ++k) {
                    const int per = cur[k] * 64 * un[k];
                    const int r = (items[k] + per - 1) / per;
                    mx = r > mx ? r : mx; sum += r;
                }
                if (mx < bestmax || (mx == bestmax && sum < bestsum)) { bestmax = mx; bestsum = sum; for (int k = 0; k < 4; ++k) best[k] = cur[k]; }
            }
    int at = 0;
    for (int k = 0; k < n; ++k) { a.src[k].w0 = at; a.src[k].nw = best[k]; at += best[k]; }
}
