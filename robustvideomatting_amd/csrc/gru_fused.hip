// A whole ConvGRU step in ONE launch (gfx950), for the two large-map GRUs of the decoder (decode1: 16 channels at 1/2 scale,
// decode2: 20 channels at 1/4 scale).  Replaces, in the reference's model/decoder.py:168-172 (ConvGRU.forward_single_frame):
//     r, z = sigmoid(conv3x3(cat[x, h])).split(C)        (ih)
//     c    = tanh(conv3x3(cat[x, r * h]))                (hh)
//     h'   = (1 - z) * h + z * c
// The two-launch form (conv_kxk with the gate / candidate epilogues) stages the [x | h] halo patch twice and sends r * h and z
// through memory between two dependent launches of 8-9 us each.  Here a block of 8 waves owns a tile of TH x TW pixels:
//   X   the (TH + 4) x (TW + 4) halo patch of [x | h] goes to LDS as MFMA pixel fragments (one round of loads; every weight
//       fragment is requested in the same round, straight into registers: wave = (n-tile, m-group))
//   IH  the gate conv on the (TH + 2) x (TW + 2) region the candidate conv will read (the one-pixel ring is recomputed by the
//       neighbouring tiles), m-tiles in flat order over that region
//   RH  r * h overwrites h in the patch (zero outside the image, like the reference's padding of cat[x, r * h]);
//       z goes to an fp32 LDS tile
//   HH  the candidate conv on the tile's own pixels out of the same patch, tanh, blend with the previous state (fetched
//       again from memory at the start: the patch copy is overwritten), store (+ the caller's 16-bit copy)
// K index of both convs = tap * 2C + channel, as packed for conv_kxk (engine.cpp pack_gru).
#include "device_common.hpp"
#include <type_traits>

constexpr int GRU_NTH = 512;

// The [x | h] halo patch in LDS, PLANE-MAJOR: one plane of 16-byte slots per 8-channel group (and per half of it: the split
// type's hi / lo f16 halves, the fp32 type's first / last four floats), a pixel = one slot of each plane.  The 16 lanes of a
// ds_read_b128 lane group read 16 consecutive pixels of ONE plane, or 8 + 8 pixels of two planes whose distance is a multiple of
// 16 slots - 16 different bank sets either way (pixel-major rows put the two 8-channel groups of such a lane group 32 bytes
// apart: two-way conflicts on a quarter of the banks, and the fragment reads bounded both convs).  Row pitch P = region width
// (mod 16), so that the flat m-tiles stay on consecutive slots (mod 16) across a row wrap.
template <typename T> struct GruPatch;
template <> struct GruPatch<f16> {
    static constexpr int NH = 1;
    static __device__ __forceinline__ void store4(char* p, int plane, int slot, int q, f32x4 v) {      // quad q of the pixel at `slot` (its group's plane 0)
        *(f16x4*)(p + ((q >> 1) * plane + slot) * 16 + (q & 1) * 8) = __builtin_convertvector(v, f16x4);
    }
    static __device__ __forceinline__ f32x4 load4(const char* p, int plane, int slot, int q) {
        return __builtin_convertvector(*(const f16x4*)(p + ((q >> 1) * plane + slot) * 16 + (q & 1) * 8), f32x4);
    }
    static __device__ __forceinline__ void frag(Frag<f16>& f, const char* p, int plane, int slot) { f.v = *(const f16x8*)(p + slot * 16); }
};
template <> struct GruPatch<f32s> {
    static constexpr int NH = 2;
    static __device__ __forceinline__ void store4(char* p, int plane, int slot, int q, f32x4 v) {
        f16x4 hi, lo;
        split4(v, hi, lo);
        char* b = p + ((q >> 1) * 2 * plane + slot) * 16 + (q & 1) * 8;
        *(f16x4*)b = hi;
        *(f16x4*)(b + plane * 16) = lo;
    }
    static __device__ __forceinline__ f32x4 load4(const char* p, int plane, int slot, int q) {             // hi + lo
        const char* b = p + ((q >> 1) * 2 * plane + slot) * 16 + (q & 1) * 8;
        return __builtin_convertvector(*(const f16x4*)b, f32x4) + __builtin_convertvector(*(const f16x4*)(b + plane * 16), f32x4);
    }
    static __device__ __forceinline__ void frag(Frag<f32s>& f, const char* p, int plane, int slot) {
        f.hi = *(const f16x8*)(p + slot * 16); f.lo = *(const f16x8*)(p + (slot + plane) * 16);
    }
};
template <> struct GruPatch<float> {
    static constexpr int NH = 2;
    static __device__ __forceinline__ void store4(char* p, int plane, int slot, int q, f32x4 v) {
        *(f32x4*)(p + (q * plane + slot) * 16) = v;                      // quad q = group q / 2, half q % 2
    }
    static __device__ __forceinline__ f32x4 load4(const char* p, int plane, int slot, int q) { return *(const f32x4*)(p + (q * plane + slot) * 16); }
    static __device__ __forceinline__ void frag(Frag<float>& f, const char* p, int plane, int slot) {
        f.lo = *(const f32x4*)(p + slot * 16); f.hi = *(const f32x4*)(p + (slot + plane) * 16);
    }
};
// gate activations on the hardware's exp2 / rcp (1 ulp each): the libm forms cost ~30 instructions per value, and a tile has
// 24 of them per lane
__device__ __forceinline__ float gru_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v)); }
__device__ __forceinline__ float gru_tanh(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177792681f * v)); }

template <int C, int TH, int TW, int ES> struct GruGeo {
    static constexpr int CP = 2 * C, NQ8 = CP / 8, QN = CP / 4, KS = (9 * CP + 31) / 32;
    static constexpr int NTG = (C + 15) / 16;                  // n-tiles per gate
    static constexpr int NTIH = 2 * NTG, NTHH = NTG;
    // a wave owns one n-tile and one slice of K of each conv (its weight fragments live in registers, every fragment is
    // fetched by exactly one wave of the block) and runs ALL m-tiles over that slice; the slices meet in LDS
    static constexpr int WP1 = 8 / NTIH, KSP1 = (KS + WP1 - 1) / WP1, NP1K = (KS + KSP1 - 1) / KSP1;   // waves per n-tile, K steps per wave, slices in use
    static constexpr int WP2 = 8 / NTHH, KSP2 = (KS + WP2 - 1) / WP2, NP2K = (KS + KSP2 - 1) / KSP2;
    static constexpr int ZC = NTG * 16;                        // channels of the z tile
    static constexpr int PH = TH + 4, PW = TW + 4, RH = TH + 2, RW = TW + 2;
    // m-tiles of the region: TW = 8 walks it flat (10-pixel rows); TW = 16 gives each row's first 16 pixels a tile and packs
    // the two leftover columns of eight rows into one (both orders touch 16 different slots (mod 16) per tile, see P)
    static constexpr bool SPLIT = TW == 16;
    static constexpr int NP1 = RH * RW, MT1 = SPLIT ? RH + (RH * 2 + 15) / 16 : (NP1 + 15) / 16, M1 = (MT1 + WP1 - 1) / WP1;
    static constexpr int MT2 = TH * TW / 16, M2 = (MT2 + WP2 - 1) / WP2;
    static constexpr int NITEMS = PH * PW * QN, XU = (NITEMS + GRU_NTH - 1) / GRU_NTH;
    static constexpr int NH = ES == 2 ? 1 : 2;                 // planes per 8-channel group
    // patch row pitch in slots, >= PW. Flat order: RW (mod 16), so a tile that wraps a row continues at the next slot
    // (mod 16). Split order: 22, whose multiples (mod 16) 0 6 12 2 8 14 4 10 leave room for the column pair beside each
    static constexpr int P = SPLIT ? 22 : RW + 16;
    static constexpr int PLANE = NH == 2 ? PH * P + (24 - (PH * P) % 16) % 16 : PH * P + (16 - (PH * P) % 16) % 16;   // slots per plane: 8 / 0 (mod 16)
    static constexpr int RED = (NP1K * NTIH * MT1 > NP2K * NTHH * MT2 ? NP1K * NTIH * MT1 : NP2K * NTHH * MT2) * 1024;   // bytes of K-slice partials
    static constexpr int OFF_Z = NQ8 * NH * PLANE * 16, OFF_RED = OFF_Z + NP1 * ZC * 4, LDS = OFF_RED + RED;
    // region pixel of lane-index i of the m-tile order (false: past the region), and the order's index of a pixel (< NP1)
    static __device__ __forceinline__ bool rpix(int i, int& ry, int& rx) {
        if constexpr (SPLIT) {
            const int j = i - RH * 16;
            ry = j < 0 ? i >> 4 : j >> 1; rx = j < 0 ? i & 15 : 16 + (j & 1);
            return ry < RH;
        } else {
            ry = i / RW; rx = i - ry * RW;
            return i < NP1;
        }
    }
    static __device__ __forceinline__ int ridx(int ry, int rx) {
        if constexpr (SPLIT) return rx < 16 ? ry * 16 + rx : RH * 16 + ry * 2 + rx - 16;
        else return ry * RW + rx;
    }
};
bool gru_fused_supported(int C, int es) { return (C == 16 || C == 20) && (es == 2 || es == 4); }
template <int C, int ES> static constexpr size_t gru_lds2(int TH, int TW) {
    return TH == 4 ? (TW == 8 ? GruGeo<C, 4, 8, ES>::LDS : GruGeo<C, 4, 16, ES>::LDS) : (TW == 8 ? GruGeo<C, 8, 8, ES>::LDS : GruGeo<C, 8, 16, ES>::LDS);
}
size_t gru_fused_lds(int C, int TH, int TW, int es) {
    return C == 16 ? (es == 2 ? gru_lds2<16, 2>(TH, TW) : gru_lds2<16, 4>(TH, TW)) : (es == 2 ? gru_lds2<20, 2>(TH, TW) : gru_lds2<20, 4>(TH, TW));
}

// acc[0..MT) += W(ks) * X(m-tile, ks) for the wave's K steps, pixel fragments read four m-tiles at a time; base[i] = slot of
// m-tile i's pixel (this lane's) at tap (0, 0)
template <typename T, int MT, int KSP, int KS, int NQ8, int P, int PLANE>
__device__ __forceinline__ void gru_conv(f32x4* acc, const Frag<T>* w, const char* patch, const int* base, int ks0, int g) {
#pragma unroll
    for (int j = 0; j < KSP; ++j) {
        const int ks = ks0 + j;
        if (ks < KS) {                                                 // (uniform per wave)
            // 8-channel group q8 = 4 ks + g sits on tap q8 / NQ8, group q8 % NQ8; K tail (tap 9): zero weights, any finite operand
            const int q8 = ks * 4 + g, t9 = q8 / NQ8, tap = min(t9, 8), c8 = q8 - t9 * NQ8;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int ko = c8 * GruPatch<T>::NH * PLANE + dy * P + dx;
#pragma unroll
            for (int m0 = 0; m0 < MT; m0 += 4) {
                Frag<T> xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) if (m0 + i < MT) GruPatch<T>::frag(xf[i], patch, PLANE, base[m0 + i] + ko);
                if (MT - m0 >= 4) mma_n<4>(acc + m0, w[j], xf);
                else if (MT - m0 == 3) mma_n<3>(acc + m0, w[j], xf);
                else if (MT - m0 == 2) mma_n<2>(acc + m0, w[j], xf);
                else mma_n<1>(acc + m0, w[j], xf);
            }
        }
    }
}

template <typename T, int C, int TH, int TW>
__global__ __launch_bounds__(GRU_NTH) void gru_fused(GruArgs a) {
    [[maybe_unused]] Stamper stamp;
    typedef GruGeo<C, TH, TW, (int)sizeof(T)> G;
    typedef typename std::conditional<sizeof(T) == 2, f16, float>::type TS;      // element type of x / h / hout in memory
    constexpr int KS = G::KS, PW = G::PW, P = G::P, PLANE = G::PLANE, MT1 = G::MT1, MT2 = G::MT2;
    extern __shared__ __attribute__((aligned(16))) char gru_smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
    char* patch = gru_smem;                                        // [8-channel group][half][PLANE slots of 16 bytes]
    float* zbuf = (float*)(gru_smem + G::OFF_Z);                   // [NP1][ZC]
    f32x4* red = (f32x4*)(gru_smem + G::OFF_RED);                  // [K slice][n-tile][m-tile][lane]
    const int y0 = ty * TH, x0 = tx * TW;
    const size_t nb = (size_t)n * a.H * a.W;
    const TS* xg = (const TS*)a.x;
    const TS* hg = (const TS*)a.h;
    // ---- X: the halo patch, one round of loads (clamped addresses, masked where used)
    f32x4 xv[G::XU];
    bool xin[G::XU];
#pragma unroll
    for (int u = 0; u < G::XU; ++u) {
        const int it = min(tid + u * GRU_NTH, G::NITEMS - 1), pp = it / G::QN, q = it - pp * G::QN;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        const bool in = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W, isx = q < C / 4;
        const size_t pix = nb + (in ? (size_t)gy * a.W + gx : 0);
        const TS* p = (isx || !hg) ? xg + pix * a.xs + (isx ? q * 4 : 0) : hg + pix * C + (q - C / 4) * 4;
        xin[u] = in && (isx || hg);
        xv[u] = ld4f(p);
    }
    // the wave's weight fragments of both convs
    const int nt1 = wave % G::NTIH, kp1 = wave / G::NTIH, nt2 = wave % G::NTHH, kp2 = wave / G::NTHH;
    Frag<T> wih[G::KSP1], whh[G::KSP2];
#pragma unroll
    for (int j = 0; j < G::KSP1; ++j) wih[j].loadw((const T*)a.wih + (nt1 * KS + min(kp1 * G::KSP1 + j, KS - 1)) * 512, lane);
#pragma unroll
    for (int j = 0; j < G::KSP2; ++j) whh[j].loadw((const T*)a.whh + (nt2 * KS + min(kp2 * G::KSP2 + j, KS - 1)) * 512, lane);
    const f32x4 b1 = *(const f32x4*)(a.bih + nt1 * 16 + g * 4);
    const f32x4 b2 = *(const f32x4*)(a.bhh + nt2 * 16 + g * 4);
    // output pixels this wave finishes (n-tile nt2, m-tiles kp2 + i WP2): their previous-state values for the blend are requested now
    const int cb2 = nt2 * 16 + g * 4;
    f32x4 hprev[G::M2];
    long long opix[G::M2];
#pragma unroll
    for (int i = 0; i < G::M2; ++i) {
        const int mt = kp2 + i * G::WP2, p2 = mt * 16 + l15, r = p2 / TW, cx = p2 - r * TW, gy = y0 + r, gx = x0 + cx;
        opix[i] = (mt < MT2 && gy < a.H && gx < a.W && cb2 < C) ? (long long)((nb + (size_t)gy * a.W + gx) * C + cb2) : -1;
        // (unconditional load from a valid address, selected afterwards: a load under a branch makes the compiler wait for
        //  every load in flight at the join; x has at least C channels per pixel, so the index is inside it too)
        const f32x4 t = ld4f((hg ? hg : xg) + max(opix[i], 0ll));
        hprev[i] = (hg && opix[i] >= 0) ? t : f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < G::XU; ++u) {
        const int it = tid + u * GRU_NTH, pp = it / G::QN, q = it - pp * G::QN;
        const int py = pp / PW, px = pp - py * PW;
        if (it < G::NITEMS) GruPatch<T>::store4(patch, PLANE, py * P + px, q, xin[u] ? xv[u] : f32x4{0, 0, 0, 0});
    }
    __syncthreads();
    stamp(1);
    // ---- IH: gate conv on the (TH + 2) x (TW + 2) region, the wave's slice of K
    {
        f32x4 acc[MT1];
        int base[MT1];
#pragma unroll
        for (int i = 0; i < MT1; ++i) {
            acc[i] = f32x4{0, 0, 0, 0};
            int ry, rx;
            const bool ok = G::rpix(i * 16 + l15, ry, rx);
            base[i] = ok ? ry * P + rx : 0;
        }
        gru_conv<T, MT1, G::KSP1, KS, G::NQ8, P, PLANE>(acc, wih, patch, base, kp1 * G::KSP1, g);
        if (kp1 < G::NP1K) {
#pragma unroll
            for (int i = 0; i < MT1; ++i) red[((kp1 * G::NTIH + nt1) * MT1 + i) * 64 + lane] = acc[i];
        }
    }
    stamp(2);
    __syncthreads();                                                   // slices complete; everyone has read h out of the patch
    // ---- RH: sum the slices; r * h back into the patch, z into its tile (n-tile nt1, m-tiles kp1 + i WP1)
    {
        const bool is_r = (nt1 & 1) == 0;
        const int cb = (nt1 >> 1) * 16 + g * 4;
#pragma unroll
        for (int i = 0; i < G::M1; ++i) {
            const int mt = kp1 + i * G::WP1, p1 = mt * 16 + l15;
            if (mt < MT1) {                                            // (uniform per wave)
                f32x4 v = b1;
#pragma unroll
                for (int p = 0; p < G::NP1K; ++p) v += red[((p * G::NTIH + nt1) * MT1 + mt) * 64 + lane];
                int ry, rx;
                if (G::rpix(p1, ry, rx) && cb < C) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gru_sigmoid(v[r]);
                    if (is_r) {
                        const int slot = (ry + 1) * P + rx + 1, q = (C + cb) >> 2;
                        GruPatch<T>::store4(patch, PLANE, slot, q, v * GruPatch<T>::load4(patch, PLANE, slot, q));
                    } else {
                        *(f32x4*)(zbuf + p1 * G::ZC + cb) = v;
                    }
                }
            }
        }
    }
    __syncthreads();
    stamp(3);
    // ---- HH: candidate conv on the tile's own pixels, the wave's slice of K
    {
        f32x4 acc[MT2];
        int base[MT2];
#pragma unroll
        for (int i = 0; i < MT2; ++i) {
            acc[i] = f32x4{0, 0, 0, 0};
            const int p2 = i * 16 + l15, r = p2 / TW, cx = p2 - r * TW;
            base[i] = (r + 1) * P + cx + 1;
        }
        gru_conv<T, MT2, G::KSP2, KS, G::NQ8, P, PLANE>(acc, whh, patch, base, kp2 * G::KSP2, g);
        if (kp2 < G::NP2K) {
#pragma unroll
            for (int i = 0; i < MT2; ++i) red[((kp2 * G::NTHH + nt2) * MT2 + i) * 64 + lane] = acc[i];
        }
    }
    __syncthreads();
    stamp(4);
    // ---- sum the slices, tanh, blend with the previous state
#pragma unroll
    for (int i = 0; i < G::M2; ++i) {
        const int mt = kp2 + i * G::WP2;
        if (mt < MT2) {
            f32x4 cv = b2;
#pragma unroll
            for (int p = 0; p < G::NP2K; ++p) cv += red[((p * G::NTHH + nt2) * MT2 + mt) * 64 + lane];
            if (opix[i] >= 0) {
                const int p2 = mt * 16 + l15, r = p2 / TW, cx = p2 - r * TW;
                const f32x4 z = *(const f32x4*)(zbuf + G::ridx(r + 1, cx + 1) * G::ZC + cb2);
#pragma unroll
                for (int k = 0; k < 4; ++k) cv[k] = (1.f - z[k]) * hprev[i][k] + z[k] * gru_tanh(cv[k]);
                st4f((TS*)a.hout + opix[i], cv);
                if (a.hout2) st4x(a.hout2, (size_t)opix[i], a.omode2, cv);
            }
        }
    }
    stamp(5);
}

template <typename T, int C, int TH, int TW> static void gru_launch1(const GruArgs& a, int N, hipStream_t s) {
    const dim3 grid(a.tiles_x, a.tiles_y, N);
    constexpr size_t lds = GruGeo<C, TH, TW, (int)sizeof(T)>::LDS;
    hipLaunchKernelGGL((gru_fused<T, C, TH, TW>), grid, dim3(GRU_NTH), lds, s, a);
}
template <typename T, int C> static void gru_launch2(const GruArgs& a, int N, hipStream_t s) {
    if (a.TH == 4 && a.TW == 8) gru_launch1<T, C, 4, 8>(a, N, s);
    else if (a.TH == 8 && a.TW == 8) gru_launch1<T, C, 8, 8>(a, N, s);
    else if (a.TH == 4 && a.TW == 16) gru_launch1<T, C, 4, 16>(a, N, s);
    else gru_launch1<T, C, 8, 16>(a, N, s);
}
template <typename T> void launch_gru_fused(const GruArgs& a, int N, hipStream_t s) {
    if (a.C == 16) gru_launch2<T, 16>(a, N, s); else gru_launch2<T, 20>(a, N, s);
}
template void launch_gru_fused<f16>(const GruArgs&, int, hipStream_t);
template void launch_gru_fused<float>(const GruArgs&, int, hipStream_t);
template void launch_gru_fused<f32s>(const GruArgs&, int, hipStream_t);

// blocks of one tile shape a CU holds at a time (registers and LDS), for the plan's tile choice
template <typename T, int C> static int gru_occ2(int TH, int TW) {
    int n = 0;
    const size_t lds = gru_fused_lds(C, TH, TW, (int)sizeof(T));
    hipError_t e;
    if (TH == 4 && TW == 8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_fused<T, C, 4, 8>, GRU_NTH, lds);
    else if (TH == 8 && TW == 8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_fused<T, C, 8, 8>, GRU_NTH, lds);
    else if (TH == 4 && TW == 16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_fused<T, C, 4, 16>, GRU_NTH, lds);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_fused<T, C, 8, 16>, GRU_NTH, lds);
    return (e == hipSuccess && n > 0) ? n : 1;
}
template <typename T> int gru_fused_occupancy(int C, int TH, int TW) { return C == 16 ? gru_occ2<T, 16>(TH, TW) : gru_occ2<T, 20>(TH, TW); }
template int gru_fused_occupancy<f16>(int, int, int);
template int gru_fused_occupancy<float>(int, int, int);
template int gru_fused_occupancy<f32s>(int, int, int);

template <typename T, int C> static void gru_attr2() {
    (void)hipFuncSetAttribute((const void*)gru_fused<T, C, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gru_fused<T, C, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gru_fused<T, C, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gru_fused<T, C, 8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T> static void gru_attr() { gru_attr2<T, 16>(); gru_attr2<T, 20>(); }
void gru_fused_init() { gru_attr<f16>(); gru_attr<float>(); gru_attr<f32s>(); }
