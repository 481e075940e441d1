// 3x3 conv (pad 1) + folded BN + ReLU over concatenated, partly x2-upsampled sources - the decoder's UpsamplingBlock and
// OutputBlock convs (model/decoder.py:82-86, :127-131) - with the weights in REGISTERS (gfx950).
// conv_kxk streams its weight fragments through double-buffered LDS stages; one frame has so little MFMA work per stage
// that every stage costs a full global round trip (decode3.conv: 7 stages x ~1 us of a 13 us launch).  Here a block of
// 8 waves owns a tile of TH x TW output pixels and NTB n-tiles; a wave owns ONE n-tile and ONE slice of K: its weight
// fragments (<= KSP of them) are requested before the halo patch is staged and sit in registers when the patch is ready;
// the wave then runs ALL m-tiles of the tile over its slice out of LDS, the slices meet in LDS and are added in slice order.
// Every weight fragment is fetched by exactly one wave of the block; the K loop has no barrier and no memory access.
// Staging (stage_source, conv_stage.hpp), K order (tap * Cpad + channel) and packed weights are conv_kxk's.
#include "conv_stage.hpp"

constexpr int CCAT_NT = 512;

template <typename T, int NTB, int MG, int KSP, int TH, int TW>
__global__ __launch_bounds__(CCAT_NT, ((KSP <= 5 && TH * TW < 256) ? 4 : 2)) void conv_cat(ConvArgs a) {     // short slices: <= 128 VGPRs, two blocks per CU
    [[maybe_unused]] Stamper stamp;
    extern __shared__ __attribute__((aligned(16))) char ccat_smem[];
    T* patch = (T*)ccat_smem;
    constexpr int KP = 8 / (NTB * MG), MT = TH * TW / 16, MTW = MT / MG, PH = TH + 2, PW = TW + 2;
    const int CSL = a.CC + Tr<T>::kLdsPad;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x, n = blockIdx.z;
    const int y0 = ty * TH, x0 = tx * TW, H = a.H, W = a.W;
    const int gy0 = y0 - 1, gx0 = x0 - 1;                    // patch origin (odd: the 2x2-block upsample path of stage_source)
    const int nt = wave % NTB, mg = (wave / NTB) % MG, kp = wave / (NTB * MG);
    const int ntg = min((int)blockIdx.y * NTB + nt, a.NT - 1);
    const int ksp = (a.KS + KP - 1) / KP;                   // K steps per slice (<= KSP, host-checked)
    const int ks0 = kp * ksp, ks1 = min(ks0 + ksp, a.KS);
    // ---- the wave's weight fragments: in flight while the patch is staged
    Frag<T> wf[KSP];
#pragma unroll
    for (int j = 0; j < KSP; ++j) wf[j].loadw((const T*)a.w + ((size_t)ntg * a.KS + min(ks0 + j, a.KS - 1)) * 512, lane);
    // the epilogue's bias quad, requested with the weights (a load issued after the K loop is a round trip on the block's critical path)
    const int ch = ((int)blockIdx.y * NTB + nt) * 16 + g * 4;
    const bool chok = ((int)blockIdx.y * NTB + nt) < a.NT && ch < a.Cout;
    const f32x4 bias = *(const f32x4*)(a.bias + min(ch, a.NT * 16 - 4));
    // ---- the halo patch (every source by its own waves, conv_assign_waves)
    {
        int coff = 0;
        for (int s = 0; s < a.nsrc; ++s) {
            const int C = a.src[s].C;
            const bool mine = wave >= a.src[s].w0 && wave < a.src[s].w0 + a.src[s].nw;
            if (mine) {
                const bool hf = sizeof(T) == 2 || a.src[s].f16;
                const int tl = tid - a.src[s].w0 * 64, nthr = a.src[s].nw * 64;
                if (a.src[s].up) {
                    if (hf) stage_source<T, true, true, PH, PW>(patch, a.src[s], 0, C >> 2, coff, CSL, n, gy0, gx0, H, W, tl, nthr);
                    else stage_source<T, true, false, PH, PW>(patch, a.src[s], 0, C >> 2, coff, CSL, n, gy0, gx0, H, W, tl, nthr);
                } else {
                    if (hf) stage_source<T, false, true, PH, PW, 4>(patch, a.src[s], 0, C >> 2, coff, CSL, n, gy0, gx0, a.Hi, a.Wi, tl, nthr);
                    else stage_source<T, false, false, PH, PW, 4>(patch, a.src[s], 0, C >> 2, coff, CSL, n, gy0, gx0, a.Hi, a.Wi, tl, nthr);
                }
            }
            coff += C;
        }
        if (coff < a.CC) {                                   // pad channels: zero weights, keep the operands finite
            for (int pp = tid; pp < PH * PW; pp += CCAT_NT)
                for (int cch = coff; cch < a.CC; cch += 4) stp4(patch + pp * CSL, cch, f32x4{0, 0, 0, 0});
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the weights were requested first and are in as well
    __syncthreads();
    stamp(1);
    // ---- implicit GEMM: all m-tiles (flat over the tile) x the wave's K slice
    f32x4 acc[MTW];
    int base[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        acc[i] = f32x4{0, 0, 0, 0};
        const int p = (mg * MTW + i) * 16 + l15, r = p / TW, cx = p - r * TW;
        base[i] = (r * PW + cx) * CSL;
    }
    {
        const int NQ8 = a.CC >> 3;
        int q8 = ks0 * 4 + g, tap = q8 / NQ8, c8 = q8 - tap * NQ8;
#pragma unroll
        for (int j = 0; j < KSP; ++j) {
            if (ks0 + j < ks1) {                             // (uniform per wave)
                const int tp = min(tap, 8);                  // K tail: weights are zero there, keep the read in-bounds
                const int dy = tp / 3, dx = tp - dy * 3;
                const int ko = (dy * PW + dx) * CSL + c8 * 8;
#pragma unroll
                for (int m0 = 0; m0 < MTW; m0 += 4) {
                    Frag<T> xf[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (m0 + i < MTW) xf[i].load_lds(patch + base[m0 + i] + ko);
                    if (MTW - m0 >= 4) mma_n<4>(acc + m0, wf[j], xf);
                    else if (MTW - m0 == 3) mma_n<3>(acc + m0, wf[j], xf);
                    else if (MTW - m0 == 2) mma_n<2>(acc + m0, wf[j], xf);
                    else mma_n<1>(acc + m0, wf[j], xf);
                }
            }
            c8 += 4;
            while (c8 >= NQ8) { c8 -= NQ8; ++tap; }
        }
    }
    stamp(2);
    __syncthreads();                                         // the patch area is reused for the slices' partial sums
    f32x4* red = (f32x4*)ccat_smem;                          // [slice][n-tile][m-tile][lane]
    const int kpu = (a.KS + ksp - 1) / ksp;                  // slices in use
    if (kp < kpu) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) red[((kp * NTB + nt) * MT + mg * MTW + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    // ---- add the slices in order, bias, ReLU, store: wave = (n-tile, m-tiles wave / NTB + i (8 / NTB))
    constexpr int WPN = 8 / NTB;
#pragma unroll
    for (int i = 0; i < (MT + WPN - 1) / WPN; ++i) {
        const int mt = wave / NTB + i * WPN;
        if (mt < MT) {
            f32x4 part[KP];
#pragma unroll
            for (int s = 0; s < KP; ++s) part[s] = red[((min(s, kpu - 1) * NTB + nt) * MT + mt) * 64 + lane];
            f32x4 v = bias;
#pragma unroll
            for (int s = 0; s < KP; ++s) if (s < kpu) v += part[s];
            const int p = mt * 16 + l15, r = p / TW, cx = p - r * TW, gy = y0 + r, gx = x0 + cx;
            if (chok && gy < H && gx < W) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                st4f((T*)a.y + (((size_t)n * H + gy) * W + gx) * a.ys + ch, v);
            }
        }
    }
    stamp(3);
}

// variant v = tile * 3 + cfg; tile: 0 = 4 x 8, 1 = 4 x 16, 2 = 8 x 16, 3 = 16 x 16 output pixels;
// cfg: 0 = two n-tiles per block x 4 K slices, 1 = one n-tile x 8 K slices (more blocks, the patch staged once per n-tile),
//      2 = one n-tile x 2 m-groups x 4 K slices (single n-tile layers: half the partial sums in LDS)
static const int ccat_tiles[4][2] = {{4, 8}, {4, 16}, {8, 16}, {16, 16}};
static int ccat_bucket(int ksp) { return ksp <= 2 ? 2 : (ksp <= 5 ? 5 : (ksp <= 8 ? 8 : (ksp <= 13 ? 13 : 0))); }
struct CcatCfg { int ntb, mg, kp, bucket; };
static bool ccat_cfg(const ConvArgs& a, int v, CcatCfg& c) {
    if (v < 0 || v >= CCAT_NVAR) return false;
    const int cfg = v % 3;
    c.ntb = cfg == 0 ? 2 : 1; c.mg = cfg == 2 ? 2 : 1; c.kp = 8 / (c.ntb * c.mg);
    if (cfg == 0 && a.NT < 2) return false;
    if (cfg == 2 && a.NT != 1) return false;
    c.bucket = ccat_bucket((a.KS + c.kp - 1) / c.kp);
    if (!c.bucket) return false;
    // instantiated: (2, 1, 5 / 8 / 13), (1, 1, 2 / 5 / 8), (1, 2, 5)
    if (cfg == 0 && c.bucket < 5) c.bucket = 5;
    if (cfg == 1 && c.bucket > 8) return false;
    if (cfg == 2) { if (c.bucket > 5) return false; c.bucket = 5; }
    return true;
}
bool conv_cat_supported(const ConvArgs& a, int v, int es, size_t* lds_out) {
    if (a.nchunks != 1 || a.epi != EPI_RELU || a.CC != a.Cpad || a.nsrc < 1) return false;
    int ctot = 0;
    for (int k = 0; k < a.nsrc; ++k) { if (a.src[k].C % 4) return false; ctot += a.src[k].C; }
    if (ctot > a.CC || (a.CC & 7)) return false;
    CcatCfg c;
    if (!ccat_cfg(a, v, c)) return false;
    const int TH = ccat_tiles[v / 3][0], TW = ccat_tiles[v / 3][1], MT = TH * TW / 16;
    if (MT % c.mg) return false;
    size_t lds = (size_t)(TH + 2) * (TW + 2) * (a.CC + (es == 2 ? 8 : 4)) * es;
    lds = std::max(lds, (size_t)c.kp * c.ntb * MT * 1024);
    if (lds_out) *lds_out = lds;
    return lds <= 160 * 1024;
}
void conv_cat_shape(const ConvArgs& a, int v, int* th, int* tw, int* ntb, int* ksp, int* mtw) {
    CcatCfg c; ccat_cfg(a, v, c);
    *th = ccat_tiles[v / 3][0]; *tw = ccat_tiles[v / 3][1]; *ntb = c.ntb; *ksp = (a.KS + c.kp - 1) / c.kp; *mtw = *th * *tw / 16 / c.mg;
}
template <typename T, int NTB, int MG, int KSP, int TH, int TW> static void ccat_launch3(ConvArgs& a, int N, size_t lds, hipStream_t s) {
    conv_assign_waves(a, TH + 2, TW + 2, 4);
    a.tiles_x = (a.W + TW - 1) / TW;
    const dim3 grid(a.tiles_x * ((a.H + TH - 1) / TH), (a.NT + NTB - 1) / NTB, N);
    hipLaunchKernelGGL((conv_cat<T, NTB, MG, KSP, TH, TW>), grid, dim3(CCAT_NT), lds, s, a);
}
template <typename T, int NTB, int MG, int KSP> static void ccat_launch2(ConvArgs& a, int N, int tile, size_t lds, hipStream_t s) {
    if (tile == 0) { if constexpr (MG == 1) ccat_launch3<T, NTB, MG, KSP, 4, 8>(a, N, lds, s); else ccat_launch3<T, NTB, MG, KSP, 4, 8>(a, N, lds, s); }
    else if (tile == 1) ccat_launch3<T, NTB, MG, KSP, 4, 16>(a, N, lds, s);
    else if (tile == 2) ccat_launch3<T, NTB, MG, KSP, 8, 16>(a, N, lds, s);
    else ccat_launch3<T, NTB, MG, KSP, 16, 16>(a, N, lds, s);
}
template <typename T> void launch_conv_cat(const ConvArgs& a_, int N, int v, hipStream_t s) {
    ConvArgs a = a_;
    size_t lds = 0;
    CcatCfg c;
    if (!conv_cat_supported(a, v, (int)sizeof(T), &lds) || !ccat_cfg(a, v, c)) return;     // (the plan checked)
    const int tile = v / 3;
    if (c.ntb == 2) {
        if (c.bucket == 5) ccat_launch2<T, 2, 1, 5>(a, N, tile, lds, s);
        else if (c.bucket == 8) ccat_launch2<T, 2, 1, 8>(a, N, tile, lds, s);
        else ccat_launch2<T, 2, 1, 13>(a, N, tile, lds, s);
    } else if (c.mg == 2) {
        ccat_launch2<T, 1, 2, 5>(a, N, tile, lds, s);
    } else {
        if (c.bucket == 2) ccat_launch2<T, 1, 1, 2>(a, N, tile, lds, s);
        else if (c.bucket == 5) ccat_launch2<T, 1, 1, 5>(a, N, tile, lds, s);
        else ccat_launch2<T, 1, 1, 8>(a, N, tile, lds, s);
    }
}
template void launch_conv_cat<f16>(const ConvArgs&, int, int, hipStream_t);
template void launch_conv_cat<float>(const ConvArgs&, int, int, hipStream_t);
template void launch_conv_cat<f32s>(const ConvArgs&, int, int, hipStream_t);

template <typename T, int NTB, int MG, int KSP> static void ccat_attr2() {
    (void)hipFuncSetAttribute((const void*)conv_cat<T, NTB, MG, KSP, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_cat<T, NTB, MG, KSP, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_cat<T, NTB, MG, KSP, 8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_cat<T, NTB, MG, KSP, 16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T> static void ccat_attr() {
    ccat_attr2<T, 2, 1, 5>(); ccat_attr2<T, 2, 1, 8>(); ccat_attr2<T, 2, 1, 13>();
    ccat_attr2<T, 1, 1, 2>(); ccat_attr2<T, 1, 1, 5>(); ccat_attr2<T, 1, 1, 8>(); ccat_attr2<T, 1, 2, 5>();
}
void conv_cat_init() { ccat_attr<f16>(); ccat_attr<float>(); ccat_attr<f32s>(); }
