// Hand-written CDNA4 (gfx950) kernels for the RVM per-frame inference path.
//
// Layout convention: every internal activation is NHWC ("pixels x channels") so that a
// pixel's channels are contiguous - the K dimension of the MFMA implicit GEMMs - and the
// 64-lane fragment loads are 16-byte vectors.  Matrix work uses
//   v_mfma_f32_16x16x32_f16   (fp16 mode, fp32 accumulate)
//   v_mfma_f32_16x16x4_f32    (fp32 mode, exact fp32 fma chain)
// with the WEIGHTS as the MFMA A operand (rows = output channels) and the PIXELS as the B
// operand (columns = pixels), so lane l of the result holds 4 consecutive output channels
// (l>>4)*4.. of pixel (l&15): one 8/16-byte NHWC store per lane per tile.
//
// Reference arithmetic each kernel replaces is cited at the kernel (paths under
// /root/reference).
#include "common.hpp"
#include <mutex>
#include <type_traits>

#include "device_common.hpp"
#include "conv_stage.hpp"

#ifdef RVM_STAMPS
int stamps_read(unsigned long long* out, unsigned int* nlaunch) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 2 * 512 * 16) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(nlaunch, HIP_SYMBOL(g_nlaunch), 2 * sizeof(unsigned int)) != hipSuccess) return 1;
    return 0;
}
#else
int stamps_read(unsigned long long*, unsigned int* nlaunch) { nlaunch[0] = nlaunch[1] = 0; return 0; }
#endif

// ------------------------------------------------------------------------------------------
// Pointwise (1x1) conv as an MFMA GEMM, operands straight from L2 into registers.
// Replaces: torchvision ConvBNActivation 1x1 (expand / project / features[16]) used by
// model/mobilenetv3.py:39-58, SE rescale (x * scale), residual add, and LR-ASPP aspp1 * aspp2
// (model/lraspp.py:6-10,18).  BN is folded into w/bias on the host.
// ------------------------------------------------------------------------------------------
// KSP = 1: the block's 4 waves own WM x WN different output tiles.  KSP = 4 (small pixel counts, long K):
// all 4 waves own the SAME MW x NW tile group, each takes every 4th K step, and the partial
// accumulators are summed through LDS in a fixed order before the epilogue (short dependent chains
// instead of a 30-step one per wave).
template <typename T, int MW, int NW, int WM, int WN, int KSP>
__global__ __launch_bounds__(256) void pw_gemm(PwArgs a) {
    [[maybe_unused]] Stamper stamp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = KSP > 1 ? 0 : wave % WM, wn = KSP > 1 ? 0 : wave / WM;
    const int mt0 = (blockIdx.x * (KSP > 1 ? 1 : WM) + wm) * MW;
    const int nt0 = (blockIdx.y * (KSP > 1 ? 1 : WN) + wn) * NW;
    const int n = blockIdx.z;
    // SE rescale of the input channels (x * scale): the sample's scale vector (<= 1024 channels, host-checked) is
    // requested first, parked in LDS behind the operand prefetch and applied where a fragment is used
    extern __shared__ __attribute__((aligned(16))) char pw_smem[];
    T* scl = (T*)pw_smem;
    const bool se_tail = a.se_hpart != nullptr;
    const bool has_sc = a.in_scale != nullptr || se_tail;
    const int sci = threadIdx.x * 4;
    f32x4 scv = f32x4{0, 0, 0, 0};
    if (a.in_scale && sci < a.Cin) scv = ld4f((const T*)a.in_scale + (size_t)n * a.Cin + sci);   // Cin % 4 == 0 for SE widths
    // SE tail in the prologue (small layers): thread t holds partial rows pg, pg + 8, ... of fc1 output t % 32 and row t of
    // fc2; everything is requested here, in front of the operand prefetch, and reduced in stage_scale()
    constexpr int SEP = 32;                                  // partial rows per thread at most (8 groups x 32 = 256 rows)
    typedef typename std::conditional<sizeof(T) == 2, f16, float>::type TS;   // storage type of fc2 / the scale vector
    float se_h[SEP];
    f32x4 se_w[8];
    float se_b1v = 0.f, se_b2v = 0.f;
    if (se_tail) {
        const int sidx = threadIdx.x & 31, pg = threadIdx.x >> 5;
        const float* hp = a.se_hpart + (size_t)n * a.se_nparts * a.se_S + min(sidx, a.se_S - 1);
#pragma unroll
        for (int j = 0; j < SEP; ++j) se_h[j] = hp[(size_t)min(pg + 8 * j, a.se_nparts - 1) * a.se_S];
        const TS* wr = (const TS*)a.se_w2 + (size_t)min((int)threadIdx.x, a.Cin - 1) * a.se_S;
#pragma unroll
        for (int k = 0; k < 8; ++k) se_w[k] = ld4f(wr + min(4 * k, a.se_S - 4));       // S % 4 == 0 (host-checked)
        se_b1v = a.se_b1[min((int)threadIdx.x, a.se_S - 1)];
        se_b2v = a.se_b2[min((int)threadIdx.x, a.Cin - 1)];
    }
    auto stage_scale = [&]() {
        if (se_tail) {
            float* part = (float*)(pw_smem + (size_t)a.KS * 32 * sizeof(T));      // [8][32]
            float* hs = part + 256;                                                // [32]
            const int t = threadIdx.x, sidx = t & 31, pg = t >> 5;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < SEP; ++j) s += (pg + 8 * j < a.se_nparts) ? se_h[j] : 0.f;
            part[pg * 32 + sidx] = s;
            __syncthreads();
            if (t < 32) {
                float h = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) h += part[q * 32 + t];
                hs[t] = t < a.se_S ? fmaxf(h * a.se_xscale + se_b1v, 0.f) : 0.f;
            }
            __syncthreads();
            if (t < a.KS * 32) {
                float acc = se_b2v;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (4 * k < a.se_S) {
                        const f32x4 hv = *(const f32x4*)(hs + 4 * k);
                        acc += (se_w[k][0] * hv[0] + se_w[k][1] * hv[1]) + (se_w[k][2] * hv[2] + se_w[k][3] * hv[3]);
                    }
                }
                const float v = t < a.Cin ? act_f(acc, ACT_HSIGMOID) : 0.f;      // zeros beyond Cin
                ((TS*)scl)[t] = (TS)v;
            }
            __syncthreads();
        } else if (has_sc) {
            if (sci < a.KS * 32) st4f(scl + sci, scv);       // zeros beyond Cin
            __syncthreads();
        }
    };
    if (mt0 >= a.MT || nt0 >= a.NT) { stage_scale(); return; }   // uniform per block when KSP > 1
    const int l15 = lane & 15, g = lane >> 4;

    const T* xp[MW];
    bool pv[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        int pix = (mt0 + mi) * 16 + l15;
        pv[mi] = (mt0 + mi) < a.MT && pix < a.HW;
        size_t ipix = (size_t)n * a.HW + (pv[mi] ? pix : 0);
        if (a.sstride > 1) {                                 // 1x1 conv with stride (resnet downsample branch)
            const int oy = (pv[mi] ? pix : 0) / a.sWo, ox = (pv[mi] ? pix : 0) - oy * a.sWo;
            ipix = (size_t)n * a.sHi * a.sWi + (size_t)oy * a.sstride * a.sWi + ox * a.sstride;
        }
        xp[mi] = (const T*)a.x + ipix * a.xs + g * 8;
    }
    const T* wp[NW];
    bool nv[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        nv[ni] = (nt0 + ni) < a.NT;
        wp[ni] = (const T*)a.w + (size_t)(nv[ni] ? nt0 + ni : nt0) * a.KS * 512;
    }
    f32x4 acc[MW][NW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};

    constexpr int PF = 3;                                   // operand prefetch depth (K steps)
    Frag<T> wf[PF][NW];
    XRaw<T> xf[PF][MW];
    auto load_step = [&](int ks, Frag<T>* wo, XRaw<T>* xo) {
        const int c = ks * 32 + g * 8;
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) wo[ni].loadw(wp[ni] + (size_t)ks * 512, lane);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            // always a load from a valid address (masked where the fragment is used): nothing here waits for data
            const bool ok = pv[mi] && c < a.Cin;
            xo[mi].load(ok ? xp[mi] + ks * 32 : (const T*)a.x);
        }
    };
    // this wave's K steps: k0, k0 + KSP, ...
    const int k0 = KSP > 1 ? wave : 0;
    const int nst = KSP > 1 ? (a.KS - k0 + KSP - 1) / KSP : a.KS;     // may be 0 for a wave when KS < KSP
#pragma unroll
    for (int p = 0; p < PF; ++p) load_step(min(k0 + KSP * p, a.KS - 1), wf[p], xf[p]);
    stage_scale();
    // epilogue operands (bias, residual, output gate) requested now: the epilogue would otherwise start with
    // one more exposed round trip to L2
    f32x4 e_bias[NW], e_osc[NW], e_res[MW][NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        const int ch = (nt0 + (nv[ni] ? ni : 0)) * 16 + g * 4;
        e_bias[ni] = *(const f32x4*)(a.bias + ch);
        e_osc[ni] = f32x4{1, 1, 1, 1};
        if (a.out_scale) {
            const f32x4 t = *(const f32x4*)(a.out_scale + (size_t)n * a.oss + min(ch, a.Cout - 4));
            if (ch < a.Cout) e_osc[ni] = t;
        }
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            e_res[mi][ni] = f32x4{0, 0, 0, 0};
            if (a.res && (KSP == 1 || ((mi * NW + ni) & 3) == wave)) {
                const int pix = (mt0 + mi) * 16 + l15;
                const bool ok = pv[mi] && nv[ni] && ch < a.Cout;
                const f32x4 t = ld4f((const T*)a.res + (ok ? ((size_t)n * a.HW + pix) * a.rs + ch : 0));
                if (ok) e_res[mi][ni] = t;
            }
        }
    }
    for (int j0 = 0; j0 < nst; j0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int j = j0 + p;
            if (j < nst) {
                const int c = (k0 + KSP * j) * 32 + g * 8;
#pragma unroll
                for (int mi = 0; mi < MW; ++mi) {
                    Frag<T> xa = has_sc ? xf[p][mi].frag(scl + c) : xf[p][mi].frag();
                    if (!(pv[mi] && c < a.Cin)) xa.zero();
#pragma unroll
                    for (int ni = 0; ni < NW; ++ni) mma(acc[mi][ni], wf[p][ni], xa);
                }
                if (j + PF < nst) load_step(k0 + KSP * (j + PF), wf[p], xf[p]);
            }
        }
    }

    stamp(1);
    if constexpr (KSP > 1) {
        // fixed-order reduction of the 4 partial accumulators; tile t of the group is finished by wave t % 4
        __shared__ f32x4 red[KSP][MW * NW][64];
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) red[wave][mi * NW + ni][lane] = acc[mi][ni];
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) {
                const int t = mi * NW + ni;
                if ((t & 3) == wave) acc[mi][ni] = (red[0][t][lane] + red[1][t][lane]) + (red[2][t][lane] + red[3][t][lane]);
            }
    }

    // epilogue: lane holds channels ch..ch+3 of pixel pix
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        if (!nv[ni]) continue;
        const int ch = (nt0 + ni) * 16 + g * 4;
        const f32x4 bias = e_bias[ni];
        const f32x4 osc = e_osc[ni];
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            if (KSP > 1 && ((mi * NW + ni) & 3) != wave) continue;
            const int pix = (mt0 + mi) * 16 + l15;
            f32x4 v = acc[mi][ni] + bias;
            if (!a.act_post) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], a.act);
            }
            const bool ok = pv[mi] && ch < a.Cout;
            if (ok) {
                if (a.res) v += e_res[mi][ni];
                if (a.act_post) {                            // resnet bottleneck: relu(bn(conv) + identity)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], a.act);
                }
                v *= osc;
                st4f((T*)a.y + ((size_t)n * a.HW + pix) * a.ys + ch, v);
            }
            if (a.pool) {
                f32x4 s = ok ? v : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += __shfl_xor(s[r], m);
                }
                if (l15 == 0 && (mt0 + mi) < a.MT)
                    *(f32x4*)(a.pool + ((size_t)n * a.MT + mt0 + mi) * (a.NT * 16) + ch) = s;
            }
        }
    }
    stamp(2);
}

// ------------------------------------------------------------------------------------------
// The same 1x1 conv as an LDS-tiled, software-pipelined GEMM (fp32-stored activations: the float and f32s types):
// block tile (64 MW) pixels x 64 channels, 8 waves as 4 (pixels) x 2 (channels), each MW x 2 MFMA tiles, K in steps of
// 32.  BOTH operands reach LDS by LDS-DMA (global_load_lds, 16 bytes per lane) into a ring of PWL_NS stages, PWL_NS - 1
// of them in flight, one barrier per K step.  The pixels land raw and are turned into MFMA fragments when read (the
// (hi, lo) split of the f32s type happens in registers); the 16-byte pieces of a pixel row are rotated by a
// row-dependent amount so that fragment reads (one pixel row per lane, 128-byte row pitch) spread over all banks
// (pwl_rot; the DMA's LDS side is contiguous per wave-instruction, so the permutation is applied to the SOURCE address
// of each lane - still one 128-byte line per 8 lanes).  Weight fragments land in their lane-linear order.
// Inside a wave the K loop is skewed by one step: the fragment reads of step k are issued, then the MFMAs of step k - 1
// run while they are in flight, then the new fragments are converted.  (Measured on the unskewed loop: 0.29 us per K
// step with the operand loads switched off against 0.08 us for the loads alone - LDS reads, conversions and MFMAs of
// the two waves of a SIMD ran one after the other, in step behind the barrier.)
// All LDS reads of the K loop are inline asm with their own s_waitcnt: left to the compiler, every LDS read after an
// LDS-DMA gets a vmcnt(0) in front (it cannot tell the ring's stages apart), which would drain the pipeline at every step.
// Blocks are mapped to tiles XCD-aware (see below); optional split-K with a fixed-order reduce pass.  pw_gemm loads
// fragments straight into registers, each wave its own: fine where the layer is bound by launch latency, but for
// resnet50's wide 1x1 layers at 1/16 scale (M = 510, K / N = 512 ... 2048) every wave pulls its operands through L2
// separately and only three K steps deep.  Candidate beside the pw_gemm configurations; picked by plan-time measurement.
// No SE input scale, no SE tail.
// ------------------------------------------------------------------------------------------
constexpr int PWL_NS = 6, PWL_NTH = 512;
template <int MW> constexpr int pwl_stage_bytes() { return 64 * MW * 128 + 4 * 2048; }
template <int MW> constexpr size_t pwl_lds_bytes() { return (size_t)PWL_NS * pwl_stage_bytes<MW>(); }
// rotation of a row's eight pieces: over the 16 rows x 2 lane groups one ds_read_b128 serves together, the pieces then
// fall into 16 different 16-byte bank sets (rows r and r + 2 share a bank set at a 128-byte pitch)
__device__ __forceinline__ int pwl_rot(int row) { return (row >> 1) & 5; }
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {           // result valid after lds_wait() on it
    f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); return v;
}
template <int OFF> __device__ __forceinline__ f32x4 lds_read16_at(unsigned addr) {   // the same with an immediate byte offset
    f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory"); return v;
}
__device__ __forceinline__ void lds_wait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ void lds_wait(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory"); }
__device__ __forceinline__ void pwl_frag(Frag<f32s>& f, f32x4 p0, f32x4 p1) { f.hi = __builtin_bit_cast(f16x8, p0); f.lo = __builtin_bit_cast(f16x8, p1); }
__device__ __forceinline__ void pwl_frag(Frag<float>& f, f32x4 p0, f32x4 p1) { f.lo = p0; f.hi = p1; }

template <typename T, int MW>
__global__ __launch_bounds__(PWL_NTH) void pw_gemm_lds(PwArgs a) {
    static_assert(sizeof(T) == 4, "fp32-stored activations only");
    [[maybe_unused]] Stamper stamp;
    extern __shared__ __attribute__((aligned(16))) char pwl_smem[];
    constexpr int NS = PWL_NS, BM = 64 * MW, ABYTES = BM * 128, STAGE = pwl_stage_bytes<MW>(), NI = MW + 1;   // NI: LDS-DMA instructions per thread and stage
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
    // Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share an L2): XCD class c = b % 8 takes the c-th eighth of
    // the tiles in an order in which neighbours share operands - (sample, K slice, channel block, pixel block) with the
    // pixel blocks fastest where the weights are the larger operand (an XCD then pulls 1/8 of the weights + the pixels
    // instead of all the weights), channel blocks fastest otherwise.  1-D grid of 8 * ceil(tiles / 8) blocks.
    const int mblks = (a.MT + 4 * MW - 1) / (4 * MW), nblks = (a.NT + 3) / 4;
    const int tiles = mblks * nblks * nsplit * a.nsamp, per_xcd = (tiles + 7) >> 3;
    int t = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (t >= tiles) return;
    int mb, nb;
    if (a.w_major) { mb = t % mblks; t /= mblks; nb = t % nblks; t /= nblks; }
    else { nb = t % nblks; t /= nblks; mb = t % mblks; t /= mblks; }
    const int zs = t % nsplit, n = t / nsplit;
    const int mt0 = mb * (4 * MW), nt0 = nb * 4;
    const int per = (a.KS + nsplit - 1) / nsplit, ks_lo = zs * per, ks_hi = min(a.KS, ks_lo + per);
    const int wm = wave & 3, wn = wave >> 2;
    // ---- epilogue operands requested first (oldest in the vector-memory queue: the K loop's counted waits cover them)
    f32x4 e_bias[2], e_res[MW][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int ch = min(nt0 + 2 * wn + ni, a.NT - 1) * 16 + g * 4;
        e_bias[ni] = *(const f32x4*)(a.bias + ch);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            e_res[mi][ni] = f32x4{0, 0, 0, 0};
            if (a.res && !a.part) {
                const int pix = min((mt0 + MW * wm + mi) * 16 + l15, a.HW - 1);
                e_res[mi][ni] = ld4f((const T*)a.res + ((size_t)n * a.HW + pix) * a.rs + min(ch, a.Cout - 4));
            }
        }
    }
    // ---- DMA sources.  Pixel operand: piece i = tid + 512 j sits in LDS slot i (row i / 8, rotated piece i % 8)
    const char* asrc[MW]; int ach[MW];
#pragma unroll
    for (int j = 0; j < MW; ++j) {
        const int i = tid + PWL_NTH * j, row = i >> 3, p = ((i & 7) - pwl_rot(row)) & 7;
        const int pix = min(mt0 * 16 + row, a.HW - 1);                  // rows past the map repeat the last pixel (never stored)
        size_t ipix = (size_t)n * a.HW + pix;
        if (a.sstride > 1) {
            const int oy = pix / a.sWo, ox = pix - oy * a.sWo;
            ipix = (size_t)n * a.sHi * a.sWi + (size_t)oy * a.sstride * a.sWi + ox * a.sstride;
        }
        ach[j] = p * 4;
        asrc[j] = (const char*)((const float*)a.x + ipix * a.xs + ach[j]);
    }
    // weight operand: piece tid of the stage's 4 fragments (an absent n-tile repeats the last one and is never stored)
    const char* wsrc = (const char*)a.w + (size_t)min(nt0 + (tid >> 7), a.NT - 1) * a.KS * 2048 + (tid & 127) * 16;
    auto issue = [&](int ks, int st) {
        char* sb = pwl_smem + st * STAGE;
#pragma unroll
        for (int j = 0; j < MW; ++j) {
            const bool okc = ks * 32 + ach[j] < a.Cin;                  // pieces past Cin: any valid address (zeroed / masked where used)
            const char* src = okc ? asrc[j] + (size_t)ks * 128 : (const char*)a.x;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(sb + (PWL_NTH * j + wave * 64) * 16), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wsrc + (size_t)ks * 2048),
                                         (void __attribute__((address_space(3)))*)(sb + ABYTES + wave * 1024), 16, 0, 0);
    };
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)pwl_smem;
    // fragment read addresses inside stage 0
    unsigned aoff[MW][2], woff[2];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int row = (MW * wm + mi) * 16 + l15, rot = pwl_rot(row);
        aoff[mi][0] = lbase + row * 128 + ((2 * g + rot) & 7) * 16;
        aoff[mi][1] = lbase + row * 128 + ((2 * g + 1 + rot) & 7) * 16;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) woff[ni] = lbase + ABYTES + (2 * wn + ni) * 2048 + lane * 16;
    f32x4 acc[MW][2];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (ks_lo + s < ks_hi) issue(ks_lo + s, s);
    int st = 0;
    Frag<T> xf[2][MW], wf[2][2];                                     // [set]: the fragments being multiplied / being read
    // one K step: reads of step ks into set `in`, MFMAs of step ks - 1 out of set `out` while they fly, conversion
    auto step = [&](int ks, Frag<T> (&xi)[MW], Frag<T> (&wi)[2], const Frag<T> (&xo)[MW], const Frag<T> (&wo)[2]) {
        // stage ks has landed once at most the NS - 2 stages issued after it are outstanding (fewer exist near the end)
        if (ks + NS - 2 < ks_hi) wait_vm<NI * (NS - 2)>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                                // everyone's pieces are in; everyone has read stage ks - 1
        if (ks + NS - 1 < ks_hi) issue(ks + NS - 1, st == 0 ? NS - 1 : st - 1);
        const unsigned so = st * STAGE;
        f32x4 wr[2][2], xr[MW][2];
        const unsigned wa = woff[0] + so;                            // the wave's two weight fragments are 2048 bytes apart
        wr[0][0] = lds_read16_at<0>(wa); wr[0][1] = lds_read16_at<1024>(wa);
        wr[1][0] = lds_read16_at<2048>(wa); wr[1][1] = lds_read16_at<3072>(wa);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) { xr[mi][0] = lds_read16(aoff[mi][0] + so); xr[mi][1] = lds_read16(aoff[mi][1] + so); }
        if (ks > ks_lo) mma_tile(acc, wo, xo);
        lds_wait(wr[0][0], wr[0][1], wr[1][0], wr[1][1]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) pwl_frag(wi[ni], wr[ni][0], wr[ni][1]);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            lds_wait(xr[mi][0], xr[mi][1]);
            XRaw<T> x; x.a = xr[mi][0]; x.b = xr[mi][1];
            xi[mi] = x.frag();
        }
        if (ks * 32 + 32 > a.Cin) {                                  // the last, partial K step (uniform branch, kept a branch by the asm):
            asm volatile("; partial K step" ::: "memory");           // lanes past Cin hold whatever was there
            if (ks * 32 + g * 8 >= a.Cin) {
#pragma unroll
                for (int mi = 0; mi < MW; ++mi) xi[mi].zero();
            }
        }
        st = st + 1 == NS ? 0 : st + 1;
    };
    int ks = ks_lo;
    for (; ks + 1 < ks_hi; ks += 2) {
        step(ks, xf[0], wf[0], xf[1], wf[1]);
        step(ks + 1, xf[1], wf[1], xf[0], wf[0]);
    }
    if (ks < ks_hi) { step(ks, xf[0], wf[0], xf[1], wf[1]); mma_tile(acc, wf[0], xf[0]); }
    else if (ks_lo < ks_hi) mma_tile(acc, wf[1], xf[1]);
    stamp(1);
    // ---- epilogue: lane holds channels ch..ch+3 of pixel pix (as in pw_gemm)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int nt = nt0 + 2 * wn + ni;
        if (nt >= a.NT) continue;
        const int ch = nt * 16 + g * 4;
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            const int mt = mt0 + MW * wm + mi;
            const int pix = mt * 16 + l15;
            const bool ok = mt < a.MT && pix < a.HW && ch < a.Cout;
            if (a.part) {                                            // split-K: raw partial sums [slice][N][HW][NT * 16]
                if (mt < a.MT && pix < a.HW)
                    *(f32x4*)(a.part + (((size_t)zs * a.nsamp + n) * a.HW + pix) * (a.NT * 16) + ch) = acc[mi][ni];
                continue;
            }
            f32x4 v = acc[mi][ni] + e_bias[ni];
            if (!a.act_post) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], a.act);
            }
            if (ok) {
                if (a.res) v += e_res[mi][ni];
                if (a.act_post) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], a.act);
                }
                if (a.out_scale) v *= *(const f32x4*)(a.out_scale + (size_t)n * a.oss + min(ch, a.Cout - 4));
                st4f((T*)a.y + ((size_t)n * a.HW + pix) * a.ys + ch, v);
            }
            if (a.pool) {
                f32x4 sm = ok ? v : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sm[r] += __shfl_xor(sm[r], m);
                }
                if (l15 == 0 && mt < a.MT) *(f32x4*)(a.pool + ((size_t)n * a.MT + mt) * (a.NT * 16) + ch) = sm;
            }
        }
    }
    stamp(2);
}
// second pass of a split-K pw_gemm_lds launch: partial sums added in slice order, then pw_gemm's epilogue
template <typename T>
__global__ __launch_bounds__(256) void pw_split_reduce(PwArgs a, long quads) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const int CQ = a.Cout >> 2, NC = a.NT * 16;
    const long pix = i / CQ;                                         // n * HW + pixel
    const int ch = (int)(i - pix * CQ) * 4;
    const size_t per = (size_t)(quads / CQ) * NC;
    const float* p = a.part + (size_t)pix * NC + ch;
    f32x4 v = *(const f32x4*)(a.bias + ch), t[4];
    for (int c0 = 0; c0 < a.nsplit; c0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *(const f32x4*)(p + (size_t)min(c0 + k, a.nsplit - 1) * per);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (c0 + k < a.nsplit) v += t[k];
    }
    if (!a.act_post) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], a.act);
    }
    if (a.res) v += ld4f((const T*)a.res + (size_t)pix * a.rs + ch);
    if (a.act_post) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], a.act);
    }
    st4f((T*)a.y + (size_t)pix * a.ys + ch, v);
}

template <typename T> void launch_pw(const PwArgs& a, int N, int cfg, hipStream_t s) {
    auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
    if constexpr (sizeof(T) == 2) { if (cfg >= 7) cfg = 0; }         // the pipelined form exists for fp32-stored activations only
    if constexpr (sizeof(T) == 4) if (cfg >= 7) {                    // LDS-tiled pipelined form (a.nsplit, a.part set by the host for split-K)
        PwArgs b = a;
        const int ns = a.nsplit > 1 ? a.nsplit : 1, mw = cfg >= 11 ? 2 : 1;                     // 11..13: 128-pixel tiles
        const int tiles = cdiv(a.MT, 4 * mw) * cdiv(a.NT, 4) * ns * N;
        b.nsamp = N;
        b.w_major = (double)a.NT * 16 >= (double)a.HW * N;           // weights the larger operand (per K element)
        const dim3 grid(8 * cdiv(tiles, 8));
        if (mw == 2) hipLaunchKernelGGL((pw_gemm_lds<T, 2>), grid, dim3(PWL_NTH), pwl_lds_bytes<2>(), s, b);
        else hipLaunchKernelGGL((pw_gemm_lds<T, 1>), grid, dim3(PWL_NTH), pwl_lds_bytes<1>(), s, b);
        if (a.part) {
            const long quads = (long)N * a.HW * (a.Cout / 4);
            hipLaunchKernelGGL((pw_split_reduce<T>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a, quads);
        }
        return;
    }
    const size_t lds = (a.in_scale || a.se_hpart) ? (size_t)a.KS * 32 * sizeof(T) + (a.se_hpart ? (256 + 32) * 4 : 0) : 0;   // the SE scale vector (+ SE tail scratch)
    switch (cfg) {
        case 0: hipLaunchKernelGGL((pw_gemm<T, 2, 2, 2, 2, 1>), dim3(cdiv(a.MT, 4), cdiv(a.NT, 4), N), dim3(256), lds, s, a); break;
        case 1: hipLaunchKernelGGL((pw_gemm<T, 1, 1, 2, 2, 1>), dim3(cdiv(a.MT, 2), cdiv(a.NT, 2), N), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL((pw_gemm<T, 2, 1, 4, 1, 1>), dim3(cdiv(a.MT, 8), cdiv(a.NT, 1), N), dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL((pw_gemm<T, 1, 2, 2, 2, 1>), dim3(cdiv(a.MT, 2), cdiv(a.NT, 4), N), dim3(256), lds, s, a); break;
        // split-K over the 4 waves: block tile = MW x NW
        case 4: hipLaunchKernelGGL((pw_gemm<T, 2, 2, 1, 1, 4>), dim3(cdiv(a.MT, 2), cdiv(a.NT, 2), N), dim3(256), lds, s, a); break;
        case 5: hipLaunchKernelGGL((pw_gemm<T, 1, 2, 1, 1, 4>), dim3(cdiv(a.MT, 1), cdiv(a.NT, 2), N), dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL((pw_gemm<T, 1, 1, 1, 1, 4>), dim3(cdiv(a.MT, 1), cdiv(a.NT, 1), N), dim3(256), lds, s, a); break;
    }
}
template void launch_pw<f16>(const PwArgs&, int, int, hipStream_t);
template void launch_pw<float>(const PwArgs&, int, int, hipStream_t);
template void launch_pw<f32s>(const PwArgs&, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// Fused inverted-residual front half: 1x1 expand (+BN+act) on the MFMA straight into an LDS tile,
// depthwise k x k (+BN+act) out of that tile, and - for squeeze-excite blocks - the block's share of
// the SE average pool pushed through fc1 (fc1 is linear before its ReLU, so per-block partial
// pre-activations simply add up).  Replaces torchvision InvertedResidual's first two
// ConvBNActivation units and the pooling half of SqueezeExcitation (reference
// model/mobilenetv3.py:39-58).  The expanded tensor never goes to memory.
// Block = 16 expanded channels x a band of output rows (full width), 1024 threads:
//   E: the band's input rows form one contiguous pixel range; wave w runs the 16-pixel m-tiles
//      w, w+16, ... with all K steps of a tile loaded at once; results go to LDS as
//      [channel quad][pixel][4] so the depthwise reads are conflict-free.
//   D: a wave owns one channel quad (uniform weights, LDS broadcast) x 64 consecutive output pixels.
// ------------------------------------------------------------------------------------------
[[maybe_unused]] constexpr int ED_KSMAX = 4;                                  // fused expand: Cin <= 128 (wider layers expand in their own GEMM; 5 steps spill)
size_t ed_lds_bytes(int K, int KS, int npxs, int es) {
    return (size_t)4 * npxs * 16 + (size_t)KS * 512 * es + (size_t)(K * K * 16 + 32 + 16) * 4 + 4096;
}
// sum over the 64 lanes of a wave, same value in every lane: DPP adds inside the rows of 16, then the four
// row totals through readlane (fixed order)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1) {
        const int t = sh == 1 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)
                    : sh == 2 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true)
                    : sh == 4 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)
                              : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true);
        v += __builtin_bit_cast(float, t);
    }
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 15)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 31)))
         + (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 47)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 63)));
}
// The LDS tile is ZERO-PADDED like the convolution's input: tile row tr <-> input row r0*stride - pad + tr,
// tile column tc <-> input column tc - pad, so a tap is a constant offset from the output's base address
// (no clamps, no masks in the depthwise loop).
// KSE = K steps of the fused expand (0: the input is already expanded), compile-time so that the number of loads
// in flight is known where a tile is consumed (a run-time count makes every wait a wait for ALL loads)
template <typename T, int K, int KSE, bool PJ = false, bool COLT = false>
__global__ __launch_bounds__(1024) void mbconv_ed(EdArgs a) {
    [[maybe_unused]] Stamper stamp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    // COLT: blockIdx.y = band x column tile (frames too wide for a full-width LDS tile); else full-width bands (compile-time,
    // so the common form keeps its contiguous-row addressing)
    const int ncol = COLT ? a.ncol : 1;
    const int slice = blockIdx.x, band = COLT ? (int)blockIdx.y / ncol : (int)blockIdx.y, ctile = COLT ? (int)blockIdx.y - band * ncol : 0, n = blockIdx.z;
    const int ox0 = COLT ? ctile * a.cwt : 0, ox1 = COLT ? min(ox0 + a.cwt, a.Wo) : a.Wo;   // output columns of this tile
    const int TWo = ox1 - ox0;
    // a block owns a.cw channels: 16, or 8 for the already expanded layers whose grid would leave CUs idle
    const int c0 = slice * a.cw, cend = min(a.C, c0 + a.cw), nq = a.cw >> 2, nqlog = a.cw == 8 ? 1 : 2;
    const int r0 = band * a.rb, r1 = min(r0 + a.rb, a.Ho);
    const int t_lo = r0 * a.stride - a.pad;                          // input row of tile row 0 (may be negative)
    const int trows = (r1 - r0 - 1) * a.stride + (K - 1) * a.dil + 1;
    const int in_lo = max(t_lo, 0), in_hi = min(t_lo + trows - 1, a.Hi - 1);
    const int tc_lo = ox0 * a.stride - a.pad;                        // input column of tile column 0 (may be negative)
    const int PWt = COLT ? (TWo - 1) * a.stride + (K - 1) * a.dil + 1 : a.Wi + 2 * a.pad;     // padded tile width
    const int ic_lo = COLT ? max(tc_lo, 0) : 0, ic_hi = COLT ? min(tc_lo + PWt - 1, a.Wi - 1) : a.Wi - 1;
    const int TWi = ic_hi - ic_lo + 1;                               // real input columns of the tile
    const int NPX = (in_hi - in_lo + 1) * TWi;                       // real input pixels of the tile (rows of TWi)
    f32x4* ebuf = (f32x4*)smem;                                      // [4][npxs]
    T* we = (T*)(ebuf + 4 * a.npxs);                                 // [KS][512]
    float* wd = (float*)(we + (size_t)a.KS * 512);                   // [K*K][16]
    float* bd = wd + K * K * 16;                                     // [16]
    float* be = bd + 16;                                             // [16]
    float* psum = be + 16;                                           // [16]
    f32x4* red = (f32x4*)(psum + 16);                                // [<= 256] per-wave-item partial sums
    const T* xb = (const T*)a.x + (((size_t)n * a.Hi + in_lo) * a.Wi + ic_lo) * a.xs;
    const unsigned invW = TWi > 1 ? 0xFFFFFFFFu / (unsigned)TWi + 1u : 0u;    // px / TWi == umulhi(px, invW) for px < 2^16 (TWi == 1: see fdiv)
    auto fdiv = [](int p, unsigned inv) { return inv ? (int)__umulhi((unsigned)p, inv) : p; };   // inv == 0 <=> divisor 1
    const bool fullw = !COLT || TWi == a.Wi;                         // the tile's input rows are one contiguous pixel range
    auto tile_index = [&](int px) {                                  // flat tile pixel -> padded tile index
        const int iy = fdiv(px, invW), ix = px - iy * TWi;
        return (in_lo - t_lo + iy) * PWt + ix + (ic_lo - tc_lo);
    };
    auto src_index = [&](int px) {                                   // flat tile pixel -> pixel offset from xb
        if (fullw) return px;
        const int iy = fdiv(px, invW), ix = px - iy * TWi;
        return iy * a.Wi + ix;
    };

    // ---- E: expand into the LDS tile.  The first m-tile's pixel loads are issued before the weights are
    //      staged so that both round trips to L2 overlap.
    const int MT = (NPX + 15) >> 4;
    constexpr int KSN = KSE > 0 ? KSE : 1;
    XRaw<T> xfa[KSN], xfb[KSN];                                      // two m-tiles in flight; as loaded, split where used
    auto load_x = [&](int mt, XRaw<T>* xf) {
        const int px = mt * 16 + l15;
        const bool pv = px < NPX;
        const T* xp = xb + (size_t)(pv ? src_index(px) : 0) * a.xs + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {                           // branch-free loads
            const bool ok = pv && ks * 32 + g * 8 < a.Cin;
            xf[ks].load(ok ? xp + ks * 32 : xb);                   // masked where the fragment is used
        }
    };
    if (KSE > 0 && wave < MT && !(a.dbg & 1)) load_x(wave, xfa);
    // SE: this block's 16 columns of the fc1 rows, requested now, used after the depthwise pass
    f32x4 w1[4];
    if (a.hpart && tid < a.S && c0 + a.cw <= a.C) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w1[q] = f32x4{0, 0, 0, 0};
            if (q < nq) w1[q] = *(const f32x4*)(a.fc1_w + (size_t)tid * a.C + c0 + q * 4);
        }
    }
    // weights of this channel slice into LDS, zero the tile
    {
        f32x4 wreg[2];
        const int pieces = a.w ? a.KS * 512 * (int)sizeof(T) / 16 : 0;         // <= 5 * 128 = 640 < 1024
        const char* src = (const char*)a.w + (size_t)slice * a.KS * 512 * sizeof(T);
        if (tid < pieces) wreg[0] = *(const f32x4*)(src + (size_t)tid * 16);
        float dwv = 0.f, bdv = 0.f, bev = 0.f;
        if (tid < K * K * 16) { const int c = c0 + (tid & 15); if (c < cend) dwv = a.dw_w[(size_t)(tid >> 4) * a.C + c]; }
        if (tid >= 1024 - 16) {
            const int c = c0 + (tid & 15);
            if (c < cend) bdv = a.dw_b[c];
            if (a.w) bev = a.ebias[c];
        }
        const int tcells = trows * PWt;
        for (int i = tid; i < 4 * tcells; i += 1024) {
            const int q = i / tcells;
            ebuf[q * a.npxs + (i - q * tcells)] = f32x4{0, 0, 0, 0};
        }
        if (tid < pieces) *(f32x4*)((char*)we + (size_t)tid * 16) = wreg[0];
        if constexpr (PJ) {                                          // project weights + bias share the (unused) SE scratch
            if (tid < 256) ((float*)red)[tid] = a.pj_w[tid];
            else if (tid < 272) ((float*)red)[tid] = a.pj_b[tid - 256];
        }
        if (tid < K * K * 16) wd[tid] = dwv;
        if (tid >= 1024 - 16) { bd[tid & 15] = bdv; be[tid & 15] = bev; }
    }
    __syncthreads();
    stamp(1);
    // every load requested so far has landed except the first tile's pixels and (conditionally, so the compiler
    // cannot count them) the fc1 columns: wait for all of them here, once, or every tile of the loop below does
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
    if (KSE > 0 && !(a.dbg & 1)) {
        auto expand_tile = [&](int mt, const XRaw<T>* xf) {
            const int px = mt * 16 + l15;
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                Frag<T> wf;
                wf.loadw(we + (size_t)ks * 512, lane);
                Frag<T> xa = xf[ks].frag();
                if (!(px < NPX && ks * 32 + g * 8 < a.Cin)) xa.zero();
                mma(acc, wf, xa);
            }
            f32x4 v = acc + *(const f32x4*)(be + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], a.eact);
            if (px < NPX) ebuf[g * a.npxs + tile_index(px)] = v;
        };
        // the next tile's pixels are requested before the current tile is multiplied
        // (requests past the last tile read a dummy address: a load count without conditions lets the compiler
        //  wait for one tile while the other stays in flight)
        for (int mt = wave; mt < MT; mt += 32) {
            load_x(mt + 16, xfb);
            expand_tile(mt, xfa);
            load_x(mt + 32, xfa);
            if (mt + 16 < MT) expand_tile(mt + 16, xfb);
        }
    } else {
        // already expanded input: copy the slice into the tile, 4 loads in flight per thread
        for (int it0 = tid; it0 < NPX * nq; it0 += 4096) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = it0 + u * 1024, px = it >> nqlog, q = it & (nq - 1);
                const bool ok = it < NPX * nq && c0 + q * 4 < cend;
                v[u] = ld4f(xb + (ok ? (size_t)src_index(px) * a.xs + c0 + q * 4 : 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = it0 + u * 1024, px = it >> nqlog, q = it & (nq - 1);
                if (it < NPX * nq && c0 + q * 4 < cend) ebuf[q * a.npxs + tile_index(px)] = v[u];
            }
        }
    }
    __syncthreads();

    stamp(2);
    // ---- D: depthwise conv out of the tile
    const int npo = (r1 - r0) * TWo;
    const int npo_pad = (npo + 63) & ~63;
    const int wpq = npo_pad >> 6;                                    // wave-items per channel quad
    T* yb = (T*)a.y + (((size_t)n * a.Ho + r0) * a.Wo + ox0) * a.C;
    const unsigned invWo = TWo > 1 ? 0xFFFFFFFFu / (unsigned)TWo + 1u : 0u;
    const int rowstep = a.dil * PWt;
    if constexpr (PJ) {
        // Block 1 (mobilenetv3.py features[1]: depthwise 3x3 + BN + ReLU -> 1x1 16 -> 16 + BN -> + input): a thread owns
        // one output pixel with all 16 channels, so the 1x1 runs in registers right behind the depthwise pass (weights
        // broadcast from LDS) and the residual is the centre tap of the tile.  Stride 1, dilation 1 (host-checked).
        const float* pjw = (const float*)red;
        for (int p = tid; p < npo; p += 1024) {
            const int oyl = fdiv(p, invWo), ox = p - oyl * TWo;
            const f32x4* e0 = ebuf + oyl * PWt + ox;
            f32x4 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4* eq = e0 + q * a.npxs;
                f32x4 acc = *(const f32x4*)(bd + q * 4);
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) acc += eq[ky * PWt + kx] * *(const f32x4*)(wd + (ky * K + kx) * 16 + q * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d[q][r] = act_f(acc[r], a.dact);
                    if constexpr (sizeof(T) == 2) d[q][r] = (float)(f16)d[q][r];          // as the depthwise output would be stored
                }
            }
            T* yp = yb + ((size_t)oyl * a.Wo + ox) * 16;
#pragma unroll
            for (int qo = 0; qo < 4; ++qo) {
                asm volatile("" ::: "memory");                       // keep the 272 weight floats in LDS, not hoisted into registers
                f32x4 o = *(const f32x4*)(pjw + 256 + qo * 4);
#pragma unroll
                for (int ci = 0; ci < 16; ++ci) o += d[ci >> 2][ci & 3] * *(const f32x4*)(pjw + ci * 16 + qo * 4);
                o += e0[qo * a.npxs + a.pad * PWt + a.pad];                               // residual: the block's input
                st4f(yp + qo * 4, o);
            }
        }
        stamp(3);
        stamp(4);
        return;
    }
    for (int it = tid; it < nq * npo_pad; it += 1024) {
        const int q = it / npo_pad, p = it - q * npo_pad;            // q is wave-uniform
        const bool valid = p < npo;
        const int pp = valid ? p : 0;
        const int oyl = fdiv(pp, invWo), ox = pp - oyl * TWo;
        const f32x4* eq = ebuf + q * a.npxs + (oyl * a.stride) * PWt + ox * a.stride;
        const float* wq = wd + q * 4;
        f32x4 acc = *(const f32x4*)(bd + q * 4);
#pragma unroll 1
        for (int ky = 0; ky < ((a.dbg & 2) ? 0 : K); ++ky) {         // one tap row in flight at a time (registers)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
                acc += eq[kx * a.dil] * *(const f32x4*)(wq + (ky * K + kx) * 16);
            eq += rowstep;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = act_f(acc[r], a.dact);
        const bool cok = c0 + q * 4 < cend;
        if (valid && cok && !(a.dbg & 4)) st4f(yb + ((size_t)oyl * a.Wo + ox) * a.C + c0 + q * 4, acc);
        if (a.hpart) {
            f32x4 s4;
#pragma unroll
            for (int r = 0; r < 4; ++r) s4[r] = wave_sum(valid ? acc[r] : 0.f);
            if (lane == 0) red[it >> 6] = s4;                        // slot = q * wpq + (p >> 6)
        }
    }
    stamp(3);
    if (a.hpart) {
        __syncthreads();
        if (tid < 16) {
            const int q = tid >> 2, r = tid & 3;
            float s = 0.f;
            for (int j0 = 0; j0 < wpq; j0 += 8) {                    // 8 independent LDS reads per round, fixed order
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (j0 + j < wpq) ? red[q * wpq + min(j0 + j, wpq - 1)][r] : 0.f;
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            psum[tid] = (c0 + tid < cend) ? s : 0.f;
        }
        __syncthreads();
        // fc1 partial: h[s] = sum_c W1[s][c0 + c] * psum[c]
        const int nparts = gridDim.x * gridDim.y;
        float* hp = a.hpart + ((size_t)n * nparts + (size_t)slice * gridDim.y + blockIdx.y) * a.S;
        if (tid < a.S) {
            float h = 0.f;
            if (c0 + a.cw <= a.C) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    h += w1[q][0] * psum[q * 4] + w1[q][1] * psum[q * 4 + 1] + w1[q][2] * psum[q * 4 + 2] + w1[q][3] * psum[q * 4 + 3];
            } else {
                const float* wr = a.fc1_w + (size_t)tid * a.C + c0;
                for (int c = 0; c0 + c < cend; ++c) h += wr[c] * psum[c];
            }
            hp[tid] = h;
        }
    }
    stamp(4);
}
template <typename T> void launch_ed(const EdArgs& a, int N, int K, hipStream_t s) {
    dim3 grid((a.C + a.cw - 1) / a.cw, a.nband * (a.ncol > 1 ? a.ncol : 1), N);
    const size_t lds = ed_lds_bytes(K, a.KS, a.npxs, sizeof(T));
    const int kse = a.w ? a.KS : 0;                                  // 0..ED_KSMAX (host-checked)
    const bool colt = a.ncol > 1;
    if (a.pj_w) {                                                    // block 1
        if (colt) hipLaunchKernelGGL((mbconv_ed<T, 3, 0, true, true>), grid, dim3(1024), lds, s, a);
        else hipLaunchKernelGGL((mbconv_ed<T, 3, 0, true>), grid, dim3(1024), lds, s, a);
        return;
    }
#define RVM_ED_CASE(KK, KS_) if (K == KK && kse == KS_) { \
        if (colt) hipLaunchKernelGGL((mbconv_ed<T, KK, KS_, false, true>), grid, dim3(1024), lds, s, a); \
        else hipLaunchKernelGGL((mbconv_ed<T, KK, KS_>), grid, dim3(1024), lds, s, a); \
        return; }
    RVM_ED_CASE(3, 0) RVM_ED_CASE(3, 1) RVM_ED_CASE(3, 2) RVM_ED_CASE(3, 3) RVM_ED_CASE(3, 4)
    RVM_ED_CASE(5, 0) RVM_ED_CASE(5, 1) RVM_ED_CASE(5, 2) RVM_ED_CASE(5, 3) RVM_ED_CASE(5, 4)
#undef RVM_ED_CASE
}
template void launch_ed<f16>(const EdArgs&, int, int, hipStream_t);
template void launch_ed<float>(const EdArgs&, int, int, hipStream_t);
template void launch_ed<f32s>(const EdArgs&, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// Dense layer on a pooled vector: SE fc1 (+ReLU), fc2 (+hardsigmoid) and the LR-ASPP gate
// (global mean -> 1x1 conv -> sigmoid, lraspp.py:11-15).  The input vector is first reduced
// from per-block partial sums (fixed order => deterministic).
// ------------------------------------------------------------------------------------------
// v + (value of the lane `n` places below in the same row of 16 lanes, 0 if none): DPP row_shr
template <int N> __device__ __forceinline__ float row_shr_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}
// Block = NTH/64 waves; a wave owns 4 outputs, one per row of 16 lanes; lane `sub` of a row takes the
// 4-element chunks sub, sub+16, ... of the weight row (coalesced 256 B per row and round).  The weight
// chunks and the bias are requested before the input vector is reduced, so the two global round trips
// overlap; the dot products finish with 4 DPP adds.
// LPO = lanes per output: 16 (a wave owns 4 outputs: the SE layers, rows of <= 256 inputs) or 64 (one output per
// wave: long rows such as the 960-input ASPP gate, spread over 4x as many blocks).  TCH = 4-element chunks of a
// weight row per lane held in registers (LPO * TCH * 4 inputs; longer rows continue from memory).
template <typename T, int TCH, int LPO>
__global__ __launch_bounds__(1024) void gemv_act(GemvArgs a) {
    [[maybe_unused]] Stamper stamp;
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ __attribute__((aligned(16))) float part[4096];                  // [row group][I] partial sums (<= 1024 threads x 4 values)
    const int n = blockIdx.y, tid = threadIdx.x, NTH = blockDim.x;
    constexpr int OPW = 64 / LPO;                                    // outputs per wave
    const int wave = tid >> 6, lane = tid & 63, row = lane / LPO, sub = lane % LPO;
    const int o = (blockIdx.x * (NTH >> 6) + wave) * OPW + row;
    const bool ov = o < a.O;
    const int nchunk = a.I >> 2;                                     // I % 4 == 0 (host-checked)
    const T* wrow = (const T*)a.w + (size_t)(ov ? o : 0) * a.I;
    f32x4 w[TCH];
#pragma unroll
    for (int t = 0; t < TCH; ++t) {
        const int c = sub + LPO * t;
        w[t] = ld4f(wrow + 4 * min(c, nchunk - 1));
        if (!(ov && c < nchunk)) w[t] = f32x4{0, 0, 0, 0};
    }
    const float bias = (a.bias && ov) ? a.bias[o] : 0.f;
    const float xb = (a.xbias && tid < a.I) ? a.xbias[tid] : 0.f;      // needed after the partial-row reduction (I < NTH case)

    // x[i] = act(xscale * sum_p xin[p][i] + xbias[i]); the partial rows are summed in a fixed order with
    // 16 loads in flight; short vectors use G = NTH / I thread groups that each take every G-th row.
    auto psum = [&](int i, int p0, int G) {
        const float* xp = a.xin + (size_t)n * a.nparts * a.part_stride + i;
        float s = 0.f;
        int p = p0;
        for (; p < a.nparts; p += 16 * G) {                  // all 16 loads of a round in flight (clamped + masked tail)
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int pj = p + j * G;
                const float t = xp[(size_t)min(pj, a.nparts - 1) * a.part_stride];
                v[j] = pj < a.nparts ? t : 0.f;
            }
            s += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])))
               + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        }
        return s;
    };
    auto fin = [&](float s, int i) {
        s *= a.xscale;
        if (a.xbias) s = act_f(s + a.xbias[i], a.xact);
        return s;
    };
    const int C4 = a.I >> 2, RG = C4 > 0 ? NTH / C4 : 0;
    if (a.I < NTH && a.nparts > 16 && (a.part_stride & 3) == 0 && a.nparts <= 16 * RG) {
        // many partial rows of a short vector (the squeeze-excite layers: <= 256 rows of 120 .. 240 values): thread =
        // (4-value column c4, row group rg) takes rows rg, rg + RG, ... - at most 16 - as 16-byte loads, ALL in flight at
        // once, so the reduction costs one round trip instead of one per 16 rows; the groups meet in LDS in group order.
        const int c4 = tid % C4, rg = tid / C4;
        if (rg < RG) {
            const float* xp = a.xin + (size_t)n * a.nparts * a.part_stride + 4 * c4;
            f32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int pj = rg + j * RG;
                v[j] = *(const f32x4*)(xp + (size_t)min(pj, a.nparts - 1) * a.part_stride);
                if (pj >= a.nparts) v[j] = f32x4{0, 0, 0, 0};
            }
            *(f32x4*)(part + rg * a.I + 4 * c4) = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])))
                                               + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        }
        __syncthreads();
        if (tid < a.I) {
            float s = 0.f;
#pragma unroll 8
            for (int gq = 0; gq < RG; ++gq) s += part[gq * a.I + tid];
            s *= a.xscale;
            xs[tid] = a.xbias ? act_f(s + xb, a.xact) : s;
        }
    } else if (a.I >= NTH || a.nparts == 1) {
        for (int i = tid; i < a.I; i += NTH) xs[i] = fin(psum(i, 0, 1), i);
    } else {
        const int G = min(NTH / a.I, a.nparts);
        if (tid < G * a.I) part[tid] = psum(tid % a.I, tid / a.I, G);
        __syncthreads();
        if (tid < a.I) {
            float s = 0.f;
            for (int gq = 0; gq < G; ++gq) s += part[gq * a.I + tid];
            s *= a.xscale;
            xs[tid] = a.xbias ? act_f(s + xb, a.xact) : s;
        }
    }
    __syncthreads();
    stamp(1);
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < TCH; ++t) {
        const int c = min(sub + LPO * t, nchunk - 1);
        const f32x4 x = *(const f32x4*)(xs + 4 * c);
        acc += (w[t][0] * x[0] + w[t][1] * x[1]) + (w[t][2] * x[2] + w[t][3] * x[3]);
    }
    for (int base = LPO * TCH; base < nchunk; base += LPO * 16) {    // the rest of a longer row, from memory
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int c = base + sub + LPO * t;
            if (ov && c < nchunk) {
                const f32x4 wv = ld4f(wrow + 4 * c), x = *(const f32x4*)(xs + 4 * c);
                acc += (wv[0] * x[0] + wv[1] * x[1]) + (wv[2] * x[2] + wv[3] * x[3]);
            }
        }
    }
    if constexpr (LPO == 16) {
        acc = row_shr_add<1>(acc);
        acc = row_shr_add<2>(acc);
        acc = row_shr_add<4>(acc);
        acc = row_shr_add<8>(acc);                                   // lane 15 of the row holds the sum
    } else {
        acc = wave_sum(acc);                                         // every lane holds the sum
    }
    if (sub == LPO - 1 && ov) {
        const float r = act_f(acc + bias, a.act);
        if (a.y_is_T) ((T*)a.y)[(size_t)n * a.O + o] = (T)r;
        else ((float*)a.y)[(size_t)n * a.O + o] = r;
    }
    stamp(2);
}
template <typename T> void launch_gemv(const GemvArgs& a, int N, hipStream_t s) {
    // many partial rows (SE fc1 partials, pooled ASPP input): 1024 threads so that one round of loads covers them
    const int nth = a.nparts > 16 ? 1024 : 256;
    if (a.I <= 256) {
        const int opb = (nth / 64) * 4;
        hipLaunchKernelGGL((gemv_act<T, 4, 16>), dim3((a.O + opb - 1) / opb, N), dim3(nth), (size_t)a.I * sizeof(float), s, a);
    } else {
        const int opb = nth / 64;
        hipLaunchKernelGGL((gemv_act<T, 4, 64>), dim3((a.O + opb - 1) / opb, N), dim3(nth), (size_t)a.I * sizeof(float), s, a);
    }
}
template void launch_gemv<f16>(const GemvArgs&, int, hipStream_t);
template void launch_gemv<float>(const GemvArgs&, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// 3x3 conv (pad 1) as an LDS-tiled implicit GEMM over up to four concatenated sources, any of
// which may be a x2 bilinear upsample computed while the halo patch is staged.
// Replaces, in model/decoder.py: UpsamplingBlock upsample+crop+cat+conv+BN+ReLU (:82-86),
// OutputBlock (:127-131) and both ConvGRU convolutions with their sigmoid / tanh / state
// blend epilogues (:168-172).  K index = tap * Cpad + channel; each 8-channel group of a K
// step may sit on a different tap.
// ------------------------------------------------------------------------------------------
// Geometry of one block of conv_kxk: TH x 16 output pixels, input halo patch PH x PW.
template <int MW, int KD, int S, int D> struct ConvGeo {
    static constexpr int TH = 4 * MW;
    static constexpr int PH = (TH - 1) * S + (KD - 1) * D + 1;
    static constexpr int PW = 15 * S + (KD - 1) * D + 1;
    static constexpr int PAD = (KD - 1) / 2 * D;
};
// LDS of one conv_kxk block: halo patch + double-buffered weight stages (SK K-steps x NW n-tiles)
constexpr int CONV_NT = 512;                                // threads per conv_kxk block
constexpr int CONV_SKMAX = 8;                               // K steps per weight stage: 2, 4 or 8 (ConvArgs::SK)
// SK > 0: double-buffered weight stages of SK steps; SK < 0: all -SK steps resident (conv_kxk WRES)
template <typename T> size_t conv_lds_bytes(int MW, int NW, int KD, int S, int D, int CC, int SK) {
    const int PH = (4 * MW - 1) * S + (KD - 1) * D + 1, PW = 15 * S + (KD - 1) * D + 1;
    size_t patch = (size_t)PH * PW * (CC + Tr<T>::kLdsPad) * sizeof(T);
    if (patch < (size_t)MW * NW * 4096) patch = (size_t)MW * NW * 4096;     // the K-group reduction reuses the patch area
    return patch + (size_t)(SK < 0 ? -SK : 2 * SK) * NW * 512 * sizeof(T);
}
template size_t conv_lds_bytes<f16>(int, int, int, int, int, int, int);
template size_t conv_lds_bytes<float>(int, int, int, int, int, int, int);
template size_t conv_lds_bytes<f32s>(int, int, int, int, int, int, int);

// KD x KD conv, stride S, dilation D, padding (KD-1)/2*D, as an LDS-tiled implicit GEMM.
// The concatenated input channels are processed in chunks of a.CC channels (one chunk for the
// mobilenetv3 decoder; several for the resnet50 layers whose halo patch would not fit LDS);
// K index inside a chunk = tap * chunk_channels + channel.
// WRES: ALL of the block's weight fragments (KS steps x NW n-tiles, host-checked to be small) are copied into LDS once, by
// LDS-DMA (global_load_lds, no staging registers), while the patch is staged; the K loop then runs without weight stages
// and without barriers.  For the short-K decoder convs, whose barrier-paced stages each cost a global round trip.
template <typename T, int MW, int NW, int KD, int S, int D, bool SPLIT = false, bool WRES = false>
__global__ __launch_bounds__(CONV_NT, (MW >= 5 ? 2 : 4)) void conv_kxk(ConvArgs a) {   // 20-row tiles: one block per CU (LDS), registers to spare
    [[maybe_unused]] Stamper stamp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* patch = (T*)smem;
    typedef ConvGeo<MW, KD, S, D> G;
    constexpr int TH = G::TH, PH = G::PH, PW = G::PW, KK = KD * KD;
    const int SK = a.SK;                                    // K steps per weight stage (2, 4 or 8; multiple of 2)
    const int WSTAGE = SK * NW * 512;                       // elements per stage
    constexpr int EPP = 16 / sizeof(T);                     // elements per 16-byte piece
    constexpr int NPMAX = CONV_SKMAX * NW * 512 / EPP / CONV_NT;   // pieces per thread per stage, at most
    const int NP = WSTAGE / EPP / CONV_NT;                  // >= 1: the host keeps SK * NW * 512 >= EPP * CONV_NT
    const int npieces = WSTAGE / EPP;
    constexpr int PPS = 512 / EPP;                          // pieces per (step, tile) fragment
    const int CSL = a.CC + Tr<T>::kLdsPad;
    T* wbuf = patch + (size_t)PH * PW * CSL;                // [2][NW][SK][512]
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    // split mode (a.part != null): blockIdx.z = sample x channel chunk; the block runs ONE chunk of the K loop and leaves
    // its raw partial sums in a.part[chunk] - the chunks of a long-K layer run side by side on different CUs instead of
    // one after the other in few blocks; conv_split_reduce adds them in chunk order and applies the epilogue
    constexpr bool splitk = SPLIT;                          // compile-time: the one-block-per-tile form compiles as it always did
    const int n = splitk ? (int)blockIdx.z / a.nchunks : (int)blockIdx.z;
    const int chunk0 = splitk ? (int)blockIdx.z - n * a.nchunks : 0;
    const int y0 = ty * TH, x0 = tx * 16;
    const int H = a.H, W = a.W;                             // output size
    const int gy0 = y0 * S - G::PAD, gx0 = x0 * S - G::PAD; // patch origin in input coordinates

    // 8 waves: wave & 3 = output row of the tile (m-tile), wave >> 2 = K group (even / odd K steps);
    // the two groups' partial accumulators are added through LDS before the epilogue
    const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3), kgrp = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7: which sources this wave stages (ConvSrc::w0, nw)
    const int nt0 = blockIdx.y * NW;
    bool nv[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) nv[ni] = (nt0 + ni) < a.NT;

    // Weight stages: the block's NW n-tiles x SK K-steps are fetched ONCE per block (16-byte pieces,
    // all 256 threads), parked in registers while the previous stage is consumed, then written to
    // the other LDS buffer.  Every wave reads its fragments from LDS, so the L2 sees each weight
    // line once per block instead of once per wave.
    f32x4 wr[NPMAX];
    int wsx[NPMAX];                                         // K step of piece j inside a stage
    unsigned wsrc[NPMAX];                                   // its source offset at step 0 (the weight blob is < 4 GiB)
#pragma unroll
    for (int j = 0; j < NPMAX; ++j) {
        const int i = min(tid + CONV_NT * j, npieces - 1);
        const int shf = a.SKlog2 + (EPP == 4 ? 7 : 6);      // log2(SK * PPS)
        const int ni = i >> shf, rem = i - (ni << shf);
        const int sidx = rem / PPS, r = rem - sidx * PPS;
        const int tile = min(nt0 + ni, a.NT - 1);
        wsx[j] = sidx;
        wsrc[j] = (unsigned)(((size_t)tile * a.KS * 512) * sizeof(T) + (size_t)r * 16);
    }
    auto wload = [&](int gs) {
#pragma unroll
        for (int j = 0; j < NPMAX; ++j)      // always NPMAX loads (pieces beyond NP repeat the last one): a count known at
                                             // compile time lets the compiler wait for exactly these and nothing else
            wr[j] = *(const f32x4*)((const char*)a.w + wsrc[j] + (size_t)min(gs + wsx[j], a.KS - 1) * (512 * sizeof(T)));
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NPMAX; ++j)
            if (j < NP) *(f32x4*)((char*)(wbuf + (size_t)buf * WSTAGE) + (size_t)(tid + CONV_NT * j) * 16) = wr[j];
    };
    const int KSfull = ((KK * a.CC + 31) / 32 + 7) / 8 * 8;   // K steps of a full channel chunk (several chunks: padded to 8)
    if constexpr (WRES) {
        // fragments of n-tiles nt0 .. nt0 + NW - 1 are contiguous in the weight blob ([NT][KS][512]) and keep that order in
        // LDS; a wave-instruction moves 64 consecutive 16-byte pieces (an odd last tile repeats tile NT - 1: never used)
        const int per_tile = a.KS * PPS, npc = NW * per_tile;
        const int swv = __builtin_amdgcn_readfirstlane(tid >> 6);
        for (int i0 = swv * 64; i0 < npc; i0 += CONV_NT) {
            const int i = i0 + (tid & 63);
            const int ni = i / per_tile, rem = i - ni * per_tile;
            const int tile = min(nt0 + ni, a.NT - 1);
            const char* g = (const char*)a.w + ((size_t)tile * per_tile + rem) * 16;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                             (void __attribute__((address_space(3)))*)((char*)wbuf + (size_t)i0 * 16), 16, 0, 0);
        }
    } else {
        wload(chunk0 * KSfull);                             // in flight while the patch is staged
    }

    f32x4 acc[MW][NW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};
    // the epilogues' bias quads (all three index them (n-tile) * 16 + g * 4), requested here: a load issued after the K loop
    // is a round trip on the block's critical path
    f32x4 biasq[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) biasq[ni] = *(const f32x4*)(a.bias + min(nt0 + ni, a.NT - 1) * 16 + g * 4);

    int gs = chunk0 * KSfull, buf = 0;                       // global K step at the start of the stage
    const int chunk_end = splitk ? chunk0 + 1 : a.nchunks;
    for (int chunk = chunk0; chunk < chunk_end; ++chunk) {
        const int c_lo = chunk * a.CC;
        const int CCc = min(a.CC, a.Cpad - c_lo);
        const int KSc = a.nchunks > 1 ? ((KK * CCc + 31) / 32 + 7) / 8 * 8 : a.KS;
        // ---- stage the halo patch of this channel chunk (the stage loop below ends with a barrier,
        //      so nobody still reads the previous chunk's patch)
        {
            int coff = 0;
            for (int s = 0; s < a.nsrc; ++s) {
                const int C = a.src[s].C;
                const int lo = max(c_lo, coff), hi = min(c_lo + CCc, coff + C);
                const bool mine = swave >= a.src[s].w0 && swave < a.src[s].w0 + a.src[s].nw;
                if (hi > lo && mine && !(a.dbg & 1)) {
                    const int q_lo = (lo - coff) >> 2, q_cnt = (hi - lo) >> 2, dst_c = lo - c_lo;
                    const bool hf = sizeof(T) == 2 || a.src[s].f16;
                    const int tl = tid - a.src[s].w0 * 64, nthr = a.src[s].nw * 64;
                    if (a.src[s].up) {
                        if (hf) stage_source<T, true, true, PH, PW>(patch, a.src[s], q_lo, q_cnt, dst_c, CSL, n, gy0, gx0, H, W, tl, nthr);
                        else stage_source<T, true, false, PH, PW>(patch, a.src[s], q_lo, q_cnt, dst_c, CSL, n, gy0, gx0, H, W, tl, nthr);
                    } else {
                        constexpr int UND = (SPLIT && MW >= 5) ? 8 : 4;      // loads in flight per thread (the big tiles have registers to spare)
                        if (hf) stage_source<T, false, true, PH, PW, UND>(patch, a.src[s], q_lo, q_cnt, dst_c, CSL, n, gy0, gx0, a.Hi, a.Wi, tl, nthr);
                        else stage_source<T, false, false, PH, PW, UND>(patch, a.src[s], q_lo, q_cnt, dst_c, CSL, n, gy0, gx0, a.Hi, a.Wi, tl, nthr);
                    }
                }
                coff += C;
            }
            const int used = min(max(coff - c_lo, 0), CCc);  // zero the pad channels (weights there are 0, keep them finite)
            if (used < CCc) {
                for (int pp = tid; pp < PH * PW; pp += CONV_NT)
                    for (int cch = used; cch < CCc; cch += 4) stp4(patch + pp * CSL, cch, f32x4{0, 0, 0, 0});
            }
        }
        if constexpr (!WRES) { if (chunk == chunk0) wstore(0); }
        // Every load so far has landed (the LDS stores above used them) except, as far as the compiler can tell, the
        // padding loads of partly filled staging rounds; say so, or it guards the first register reuse in the K loop
        // with a vmcnt(0) that also waits for the weight stage in flight, on every K step.
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        __syncthreads();
        if (chunk == chunk0) stamp(1);

        // ---- implicit GEMM over this chunk, SK steps per weight stage ----
        const int NQ8 = CCc >> 3;
        int tap = 0, c8 = g + 4 * kgrp;                          // K group 1 starts one step in
        while (c8 >= NQ8) { c8 -= NQ8; ++tap; }
        if constexpr (WRES) {
            for (int p = kgrp; p < KSc; p += 2) {
                const int tp = min(tap, KK - 1);            // K tail: weights are zero there, keep the read in-bounds
                const int dy = tp / KD, dx = tp - dy * KD;
                const T* pb = patch + ((wave * S + dy * D) * PW + l15 * S + dx * D) * CSL + c8 * 8;
                Frag<T> xf[MW], wf[NW];
#pragma unroll
                for (int mi = 0; mi < MW; ++mi) xf[mi].load_lds(pb + (4 * mi * S) * PW * CSL);
#pragma unroll
                for (int ni = 0; ni < NW; ++ni) wf[ni].loadw(wbuf + (size_t)(ni * a.KS + p) * 512, lane);
                mma_tile(acc, wf, xf);
                mma_tile_schedule<T, MW, NW>();
                c8 += 8;                                    // two K steps on
                while (c8 >= NQ8) { c8 -= NQ8; ++tap; }
            }
            __syncthreads();                                // the patch area is reused below
        } else
        for (int ks0 = 0; ks0 < ((a.dbg & 2) ? 0 : KSc); ks0 += SK) {
            const bool more = (splitk ? ks0 + SK < KSc : gs + SK < a.KS) && !(a.dbg & 8);
            if (more) wload(gs + SK);
            const T* wst = wbuf + (size_t)buf * WSTAGE;
            for (int p = kgrp; p < SK; p += 2) {
                if (ks0 + p < KSc && !(a.dbg & 4)) {
                    const int tp = min(tap, KK - 1);        // K tail: weights are zero there, keep the read in-bounds
                    const int dy = tp / KD, dx = tp - dy * KD;
                    const T* pb = patch + ((wave * S + dy * D) * PW + l15 * S + dx * D) * CSL + c8 * 8;
                    Frag<T> xf[MW], wf[NW];
#pragma unroll
                    for (int mi = 0; mi < MW; ++mi) xf[mi].load_lds(pb + (4 * mi * S) * PW * CSL);
#pragma unroll
                    for (int ni = 0; ni < NW; ++ni) wf[ni].loadw(wst + (ni * SK + p) * 512, lane);
                    mma_tile(acc, wf, xf);
                    mma_tile_schedule<T, MW, NW>();
                }
                c8 += 8;                                    // two K steps on
                while (c8 >= NQ8) { c8 -= NQ8; ++tap; }
            }
            if (more) wstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            gs += SK;
        }
    }

    // ---- add the two K groups' accumulators (the patch area is free after the last barrier)
    {
        f32x4* red = (f32x4*)smem;                               // [4 rows][MW * NW][64]
        if (kgrp == 1) {
#pragma unroll
            for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                for (int ni = 0; ni < NW; ++ni) red[(wave * MW * NW + mi * NW + ni) * 64 + lane] = acc[mi][ni];
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) acc[mi][ni] += red[(wave * MW * NW + mi * NW + ni) * 64 + lane];
    }

    // ---- epilogue ----
    stamp(2);
    const int gx = x0 + l15;
    if constexpr (splitk) {                                  // raw partial sums [chunk][sample][H][W][NT * 16]
        const int NC = a.NT * 16;
        float* part = a.part + (size_t)chunk0 * (gridDim.z / a.nchunks) * H * W * NC;
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            const int gy = y0 + wave + 4 * mi;
            if (gy >= H || gx >= W) continue;
            const size_t pix = ((size_t)n * H + gy) * W + gx;
#pragma unroll
            for (int ni = 0; ni < NW; ++ni)
                if (nv[ni]) *(f32x4*)(part + pix * NC + (nt0 + ni) * 16 + g * 4) = acc[mi][ni];
        }
        stamp(3);
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int gy = y0 + wave + 4 * mi;
        if (gy >= H || gx >= W) continue;
        const size_t pix = ((size_t)n * H + gy) * W + gx;
        if (a.epi == EPI_RELU) {
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) {
                const int ch = (nt0 + ni) * 16 + g * 4;
                if (!nv[ni] || ch >= a.Cout) continue;
                f32x4 v = acc[mi][ni] + biasq[ni];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                st4f((T*)a.y + pix * a.ys + ch, v);
            }
        } else if (a.epi == EPI_GRU_IH) {
            // tiles come in (r, z) pairs: packed tile 2j = r[16j..], 2j+1 = z[16j..]  (decoder.py:169)
            if constexpr (NW >= 2) {
#pragma unroll
                for (int ni = 0; ni + 1 < NW; ni += 2) {
                    const int t = nt0 + ni;
                    const int c = (t >> 1) * 16 + g * 4;
                    if (!nv[ni] || c >= a.hC) continue;
                    f32x4 rv = acc[mi][ni] + biasq[ni];
                    f32x4 zv = acc[mi][ni + 1] + biasq[ni + 1];
                    f32x4 h = f32x4{0, 0, 0, 0};
                    if (a.hprev) h = ld4x(a.hprev, pix * a.hC + c, sizeof(T) == 2 || a.state16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { rv[r] = sigmoid_f(rv[r]) * h[r]; zv[r] = sigmoid_f(zv[r]); }
                    st4f((T*)a.rh + pix * a.hC + c, rv);
                    *(f32x4*)(a.z + pix * a.hC + c) = zv;
                }
            }
        } else {
            // candidate + blend: h' = (1-z) h + z tanh(.)   (decoder.py:170-171)
#pragma unroll
            for (int ni = 0; ni < NW; ++ni) {
                const int c = (nt0 + ni) * 16 + g * 4;
                if (!nv[ni] || c >= a.hC) continue;
                f32x4 cv = acc[mi][ni] + biasq[ni];
                f32x4 h = f32x4{0, 0, 0, 0};
                if (a.hprev) h = ld4x(a.hprev, pix * a.hC + c, sizeof(T) == 2 || a.state16);
                const f32x4 z = *(const f32x4*)(a.z + pix * a.hC + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[r] = (1.f - z[r]) * h[r] + z[r] * tanhf(cv[r]);
                st4x(a.hout, pix * a.hC + c, sizeof(T) == 2 || a.state16, cv);
                if (a.hout2) st4x(a.hout2, pix * a.hC + c, a.out2_mode ? a.out2_mode : (int)(sizeof(T) == 2 || a.state16), cv);
            }
        }
    }
    stamp(3);
}

// Second pass of a split-K conv_kxk launch: y = relu(bias + sum of the partial sums, in (chunk, K group) order).
template <typename T>
__global__ __launch_bounds__(256) void conv_split_reduce(ConvArgs a, long quads) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const int CQ = a.Cout >> 2, NC = a.NT * 16;
    const long pix = i / CQ;
    const int ch = (int)(i - pix * CQ) * 4;
    const size_t per = (size_t)(quads / CQ) * NC;            // elements of one chunk's partial tensor
    const float* p = a.part + (size_t)pix * NC + ch;
    f32x4 v[4], s = *(const f32x4*)(a.bias + ch);
    const int nparts = a.nchunks;
    for (int c0 = 0; c0 < nparts; c0 += 4) {                 // four partial sums in flight, added in chunk order
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *(const f32x4*)(p + (size_t)min(c0 + k, nparts - 1) * per);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (c0 + k < nparts) s += v[k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = fmaxf(s[r], 0.f);
    st4f((T*)a.y + (size_t)pix * a.ys + ch, s);
}

template <typename T, int MW, int NW, int KD, int S, int D, bool SPLITOK = false, bool WRESOK = false>
static void launch_conv_t(const ConvArgs& a_, int N, hipStream_t s) {
    ConvArgs a = a_;
    conv_assign_waves(a, ConvGeo<MW, KD, S, D>::PH, ConvGeo<MW, KD, S, D>::PW, 4);
    const int TH = 4 * MW;
    const int tiles_y = (a.H + TH - 1) / TH;
    dim3 grid(a.tiles_x * tiles_y, (a.NT + NW - 1) / NW, a.part ? N * a.nchunks : N);
    const size_t lds = conv_lds_bytes<T>(MW, NW, KD, S, D, a.CC, a.wres ? -a.KS : a.SK);
    if constexpr (WRESOK) {
        if (a.wres) { hipLaunchKernelGGL((conv_kxk<T, MW, NW, KD, S, D, false, true>), grid, dim3(CONV_NT), lds, s, a); return; }
    }
    if constexpr (SPLITOK) {
        if (a.part) {
            hipLaunchKernelGGL((conv_kxk<T, MW, NW, KD, S, D, true>), grid, dim3(CONV_NT), lds, s, a);
            const long quads = (long)N * a.H * a.W * (a.Cout / 4);
            hipLaunchKernelGGL((conv_split_reduce<T>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a, quads);
            return;
        }
    }
    hipLaunchKernelGGL((conv_kxk<T, MW, NW, KD, S, D>), grid, dim3(CONV_NT), lds, s, a);
}
// geo: 0 = 3x3 s1 d1, 1 = 3x3 s2 d1, 2 = 3x3 s1 d2, 3 = 7x7 s2 d1 (geometries 1..3 exist for MW=1, NW=2 only)
// ------------------------------------------------------------------------------------------
// 3x3 conv (stride 1 or 2, dilation 1 or 2, padding = dilation) for SMALL maps: the same implicit GEMM as conv_kxk, but shaped like the split-K
// pw_gemm - a block owns ONE 16-pixel row segment x NW n-tiles, its 4 waves take every 4th K step, operands come
// straight from L2 into registers (4 K steps of prefetch, no LDS staging, no barriers in the K loop) and the four
// partial accumulators are added through LDS in a fixed order.  On the 17x30 / 34x60 ConvGRUs conv_kxk has only
// 40..108 blocks, each streaming its ~150 KB of weights through one CU at the ~30 GB/s a barrier-paced stage
// pipeline reaches; this form spreads the same stream over 3-4x as many CUs.  Sources are direct (not upsampled)
// and 8-channel aligned; fp16 state tensors feed the split MFMA as (hi = value, lo = 0).
// ------------------------------------------------------------------------------------------
template <typename T> struct SkOperand;                              // one 8-channel K group of one pixel, as loaded
template <> struct SkOperand<f16> {
    f32x4 a; bool ok;
    __device__ __forceinline__ void load(const void* p, bool) { a = *(const f32x4*)p; }
    __device__ __forceinline__ Frag<f16> frag(bool) const { Frag<f16> f; f.v = __builtin_bit_cast(f16x8, a); if (!ok) f.zero(); return f; }
};
template <> struct SkOperand<float> {
    f32x4 a, b; bool ok;
    __device__ __forceinline__ void load(const void* p, bool half) { a = *(const f32x4*)p; b = *(const f32x4*)((const char*)p + (half ? 0 : 16)); }
    __device__ __forceinline__ Frag<float> frag(bool half) const {
        Frag<float> f;
        if (half) { const f16x8 h = __builtin_bit_cast(f16x8, a); f.lo = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; f.hi = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]}; }
        else { f.lo = a; f.hi = b; }
        if (!ok) f.zero();
        return f;
    }
};
template <> struct SkOperand<f32s> {
    f32x4 a, b; bool ok;
    __device__ __forceinline__ void load(const void* p, bool half) { a = *(const f32x4*)p; b = *(const f32x4*)((const char*)p + (half ? 0 : 16)); }
    __device__ __forceinline__ Frag<f32s> frag(bool half) const {
        Frag<f32s> f;
        if (half) { f.hi = __builtin_bit_cast(f16x8, a); f.lo = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }     // an fp16 value is its own hi part
        else f.set(a, b);
        if (!ok) f.zero();
        return f;
    }
};
// The same operand fetched as two 4-channel halves, each from the source that holds it: for concatenations whose
// channel counts are multiples of 4 but not of 8 (the 20-channel ConvGRU: x[16..19] | h[0..3] share a K group).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct SkQuads {
    f32x4 q[2]; bool ok[2], half[2];
    __device__ __forceinline__ void load(int i, const void* p, bool h) {
        half[i] = h;
        if (h) { const f32x2 t = *(const f32x2*)p; q[i] = f32x4{t[0], t[1], 0.f, 0.f}; }
        else q[i] = *(const f32x4*)p;
    }
    __device__ __forceinline__ f32x4 value(int i) const {
        f32x4 v = q[i];
        if (half[i]) { const f16x4 h = __builtin_bit_cast(f16x4, f32x2{q[i][0], q[i][1]}); v = __builtin_convertvector(h, f32x4); }
        return ok[i] ? v : f32x4{0, 0, 0, 0};
    }
    __device__ __forceinline__ Frag<T> frag() const;
};
template <> __device__ __forceinline__ Frag<f32s> SkQuads<f32s>::frag() const { Frag<f32s> f; f.set(value(0), value(1)); return f; }
template <> __device__ __forceinline__ Frag<float> SkQuads<float>::frag() const { Frag<float> f; f.lo = value(0); f.hi = value(1); return f; }
template <> __device__ __forceinline__ Frag<f16> SkQuads<f16>::frag() const {
    const f16x4 a = __builtin_convertvector(value(0), f16x4), b = __builtin_convertvector(value(1), f16x4);
    Frag<f16> f; f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); return f;
}
// WS = waves per block = ways the K steps are split (4, or 8 for long K on grids that leave CUs idle)
template <typename T, int NW, bool QUAD, int WS>
__global__ __launch_bounds__(64 * WS) void conv_sk(ConvArgs a) {
    [[maybe_unused]] Stamper stamp;
    __shared__ f32x4 red[WS][NW][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int tx = blockIdx.x % a.tiles_x, y = blockIdx.x / a.tiles_x, n = blockIdx.z;
    const int H = a.H, W = a.W;
    const int px = tx * 16 + l15;
    const int nt0 = blockIdx.y * NW;
    bool nv[NW];
    const T* wp[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        nv[ni] = (nt0 + ni) < a.NT;
        wp[ni] = (const T*)a.w + (size_t)(nv[ni] ? nt0 + ni : nt0) * a.KS * 512;
    }
    const int NQ8 = a.Cpad >> 3;
    const unsigned invQ = 0xFFFFFFFFu / (unsigned)NQ8 + 1u;          // q8 / NQ8 == umulhi(q8, invQ) for these sizes
    const int C0 = a.src[0].C, Ctot = C0 + (a.nsrc > 1 ? a.src[1].C : 0);
    const bool kernel16 = sizeof(T) == 2;
    const int cS = a.cs > 1 ? a.cs : 1, cD = a.cd > 1 ? a.cd : 1;
    const int KSc0 = ((9 * a.CC + 31) / 32 + 7) / 8 * 8;           // K steps of a full channel chunk (conv_kxk's packing)
    constexpr int PF = 4;
    Frag<T> wf[PF][NW];
    SkOperand<T> xo[PF];
    SkQuads<T> xq[PF];
    bool xh[PF];
    auto load_step = [&](int ks, int p) {
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) wf[p][ni].loadw(wp[ni] + (size_t)ks * 512, lane);
        int tap, c;
        if (a.nchunks <= 1) {
            const int q8 = ks * 4 + g;
            tap = (int)__umulhi((unsigned)q8, invQ); c = (q8 - tap * NQ8) * 8;
        } else {
            // weights packed for conv_kxk's channel chunks: [chunk][tap][channel of the chunk], every chunk padded to
            // a multiple of 8 K steps (K steps beyond the 9 taps are zero there)
            const int chunk = min(ks / KSc0, a.nchunks - 1), ksl = ks - chunk * KSc0;
            const int nq = min(a.CC, a.Cpad - chunk * a.CC) >> 3;
            const int q8 = ksl * 4 + g;
            tap = q8 / nq; c = chunk * a.CC + (q8 - tap * nq) * 8;
        }
        const int dy = tap / 3, dx = tap - dy * 3;
        const int yy = y * cS + (dy - 1) * cD, xx = px * cS + (dx - 1) * cD;      // padding = dilation
        const bool pos = tap < 9 && yy >= 0 && yy < a.Hi && xx >= 0 && xx < a.Wi;
        const size_t pixel = pos ? ((size_t)n * a.Hi + yy) * a.Wi + xx : 0;
        if constexpr (!QUAD) {
            const bool second = c >= C0;
            const ConvSrc& S = second ? a.src[1] : a.src[0];
            const bool half = kernel16 || S.f16;
            xh[p] = half;
            xo[p].ok = pos && c < Ctot && S.p != nullptr;
            const size_t off = xo[p].ok ? pixel * S.Cs + (c - (second ? C0 : 0)) : 0;
            const void* base = S.p ? S.p : a.src[0].p;               // a missing state (first frame) reads as zeros
            xo[p].load((const char*)base + off * (half ? 2 : 4), half);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int cq = c + 4 * i;
                const bool second = cq >= C0;
                const ConvSrc& S = second ? a.src[1] : a.src[0];
                const bool half = kernel16 || S.f16;
                xq[p].ok[i] = pos && cq < Ctot && S.p != nullptr;
                const size_t off = xq[p].ok[i] ? pixel * S.Cs + (cq - (second ? C0 : 0)) : 0;
                const void* base = S.p ? S.p : a.src[0].p;
                xq[p].load(i, (const char*)base + off * (half ? 2 : 4), half);
            }
        }
    };
    f32x4 acc[NW];
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) acc[ni] = f32x4{0, 0, 0, 0};
    // Wave 0's epilogue operands (bias quads; previous state and z of the ConvGRU epilogues) are requested HERE, in front
    // of the K loop: asked for after the reduction they were one more L2 round trip on the block's critical path
    // (~1 us of a 6-8 us launch).  Clamped addresses, unconditional within the wave.
    const bool s16 = sizeof(T) == 2 || a.state16;
    const size_t pix = ((size_t)n * H + y) * W + min(px, W - 1);
    f32x4 e_bias[NW], e_h[NW], e_z[NW];
    if (wave == 0) {
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            e_bias[ni] = *(const f32x4*)(a.bias + min(nt0 + ni, a.NT - 1) * 16 + g * 4);
            e_h[ni] = f32x4{0, 0, 0, 0}; e_z[ni] = e_h[ni];
            if (a.epi == EPI_GRU_IH) {
                if (NW >= 2 && !(ni & 1) && a.hprev) e_h[ni] = ld4x(a.hprev, pix * a.hC + min(((nt0 + ni) >> 1) * 16 + g * 4, a.hC - 4), s16);
            } else if (a.epi != EPI_RELU) {
                const int c = min((nt0 + ni) * 16 + g * 4, a.hC - 4);
                if (a.hprev) e_h[ni] = ld4x(a.hprev, pix * a.hC + c, s16);
                e_z[ni] = *(const f32x4*)(a.z + pix * a.hC + c);
            }
        }
    }
    const int nst = (a.KS - wave + WS - 1) / WS;                     // this wave's K steps: wave, wave + WS, ...
#pragma unroll
    for (int p = 0; p < PF; ++p) load_step(min(wave + WS * p, a.KS - 1), p);
    auto consume = [&](int p) {
        Frag<T> xa;
        if constexpr (QUAD) xa = xq[p].frag(); else xa = xo[p].frag(xh[p]);
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) mma(acc[ni], wf[p][ni], xa);
    };
    // Groups of PF steps.  Every group but the last refills each slot right after using it, unconditionally (steps
    // beyond the end repeat the last one and are never used): with a load count that does not depend on run-time
    // conditions the compiler waits for exactly the slot it needs and leaves the PF - 1 younger ones in flight.
    // The last group only drains.
    int j0 = 0;
    for (; j0 + PF < nst; j0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            consume(p);
            load_step(min(wave + WS * (j0 + p + PF), a.KS - 1), p);
        }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (j0 + p < nst) consume(p);
    stamp(1);
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) red[wave][ni][lane] = acc[ni];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        acc[ni] = (red[0][ni][lane] + red[1][ni][lane]) + (red[2][ni][lane] + red[3][ni][lane]);
        if constexpr (WS == 8) acc[ni] += (red[4][ni][lane] + red[5][ni][lane]) + (red[6][ni][lane] + red[7][ni][lane]);
    }

    // ---- epilogue (wave 0): lane holds channels g*4..g*4+3 of pixel px for every n-tile; same arithmetic as conv_kxk
    if (px >= W) return;
    if (a.epi == EPI_RELU) {
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            const int ch = (nt0 + ni) * 16 + g * 4;
            if (!nv[ni] || ch >= a.Cout) continue;
            f32x4 v = acc[ni] + e_bias[ni];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            st4f((T*)a.y + pix * a.ys + ch, v);
        }
    } else if (a.epi == EPI_GRU_IH) {
        if constexpr (NW >= 2) {
#pragma unroll
            for (int ni = 0; ni + 1 < NW; ni += 2) {
                const int t = nt0 + ni;
                const int c = (t >> 1) * 16 + g * 4;
                if (!nv[ni] || c >= a.hC) continue;
                f32x4 rv = acc[ni] + e_bias[ni];
                f32x4 zv = acc[ni + 1] + e_bias[ni + 1];
                const f32x4 h = e_h[ni];
#pragma unroll
                for (int r = 0; r < 4; ++r) { rv[r] = sigmoid_f(rv[r]) * h[r]; zv[r] = sigmoid_f(zv[r]); }
                st4f((T*)a.rh + pix * a.hC + c, rv);
                *(f32x4*)(a.z + pix * a.hC + c) = zv;
            }
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < NW; ++ni) {
            const int c = (nt0 + ni) * 16 + g * 4;
            if (!nv[ni] || c >= a.hC) continue;
            f32x4 cv = acc[ni] + e_bias[ni];
            const f32x4 h = e_h[ni], z = e_z[ni];
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[r] = (1.f - z[r]) * h[r] + z[r] * tanhf(cv[r]);
            st4x(a.hout, pix * a.hC + c, s16, cv);
            if (a.hout2) st4x(a.hout2, pix * a.hC + c, a.out2_mode ? a.out2_mode : (int)s16, cv);
        }
    }
    stamp(2);
}
template <typename T, int NW> static void launch_conv_sk(const ConvArgs& a, int N, hipStream_t s) {
    dim3 grid(a.tiles_x * a.H, (a.NT + NW - 1) / NW, N);
    bool quad = false;                                               // a source whose channel count is not a multiple of 8
    for (int k = 0; k < a.nsrc; ++k) quad |= (a.src[k].C % 8) != 0;
    // 8 waves per block (each takes every 8th K step) when the grid is small and K is long enough to give every wave a
    // few steps: twice the loads in flight per CU (measured: 17x30 ConvGRU 8.0 / 8.3 -> 7.0 / 6.5 us)
    const long nblk = (long)grid.x * grid.y * grid.z;
    const bool ws8 = a.KS >= 20 && (nblk <= 160 || (NW == 1 && nblk <= 512));   // (two n-tiles: the 512-thread blocks of a larger grid no longer fit at once)
    if (ws8) {
        if (quad) hipLaunchKernelGGL((conv_sk<T, NW, true, 8>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv_sk<T, NW, false, 8>), grid, dim3(512), 0, s, a);
    } else {
        if (quad) hipLaunchKernelGGL((conv_sk<T, NW, true, 4>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_sk<T, NW, false, 4>), grid, dim3(256), 0, s, a);
    }
}
template <typename T> void launch_conv3(const ConvArgs& a, int N, int MW, int NW, int geo, hipStream_t s) {
    if (MW == 0) { if (NW == 2) launch_conv_sk<T, 2>(a, N, s); else launch_conv_sk<T, 1>(a, N, s); return; }   // small-map form
    if (geo == 1) { launch_conv_t<T, 1, 2, 3, 2, 1, true>(a, N, s); return; }
    if (geo == 2) {
        if (MW == 5 && NW == 4) launch_conv_t<T, 5, 4, 3, 1, 2, true>(a, N, s);
        else if (MW == 5 && NW == 1) launch_conv_t<T, 5, 1, 3, 1, 2, true>(a, N, s);
        else if (MW == 5) launch_conv_t<T, 5, 2, 3, 1, 2, true>(a, N, s);
        else launch_conv_t<T, 1, 2, 3, 1, 2, true>(a, N, s);
        return;
    }
    if (geo == 3) { launch_conv_t<T, 1, 2, 7, 2, 1>(a, N, s); return; }
    if (MW == 5 && NW == 4) launch_conv_t<T, 5, 4, 3, 1, 1, true>(a, N, s);
    else if (MW == 5 && NW == 1) launch_conv_t<T, 5, 1, 3, 1, 1, true>(a, N, s);
    else if (MW == 5 && NW == 2) launch_conv_t<T, 5, 2, 3, 1, 1, true>(a, N, s);
    else if (MW == 4 && NW == 1) launch_conv_t<T, 4, 1, 3, 1, 1, false, true>(a, N, s);
    else if (MW == 2 && NW == 2) launch_conv_t<T, 2, 2, 3, 1, 1, true, true>(a, N, s);
    else if (MW == 2 && NW == 1) launch_conv_t<T, 2, 1, 3, 1, 1, true, true>(a, N, s);
    else if (MW == 1 && NW == 2) launch_conv_t<T, 1, 2, 3, 1, 1, true, true>(a, N, s);
    else launch_conv_t<T, 1, 1, 3, 1, 1, false, true>(a, N, s);
}
template void launch_conv3<f16>(const ConvArgs&, int, int, int, int, hipStream_t);
template void launch_conv3<float>(const ConvArgs&, int, int, int, int, hipStream_t);
template void launch_conv3<f32s>(const ConvArgs&, int, int, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// Downsample src (NCHW, full res) -> s0 (NHWC4: r,g,b,gray) with scale-factor-driven
// bilinear coordinates (model/model.py:80-81: align_corners=False,
// recompute_scale_factor=False => src = (dst+0.5)/ds - 0.5, clamped at 0).
// ------------------------------------------------------------------------------------------
// One wave = one 8x8 block of s0 = one s3 pixel, lanes in Morton order: 4 adjacent lanes are the 2x2 block under one
// s1 pixel, 16 the 4x4 block under one s2 pixel.  Every lane interpolates ONE s0 pixel (12 source taps, one round
// trip to memory); the three levels of the decoder's AvgPool pyramid (decoder.py:33-39: 2x2, ceil_mode,
// count_include_pad=False) are then formed in registers through DPP / bpermute / readlane, each level from the
// previous one rounded to the storage type, children summed in the order (0,0) (0,1) (1,0) (1,1).
__device__ __forceinline__ float quad_lane(float v, int k) {        // value of lane k of this lane's group of 4
    const int i = __builtin_bit_cast(int, v);
    int r;
    switch (k) {
        case 0: r = __builtin_amdgcn_update_dpp(0, i, 0x00, 0xf, 0xf, true); break;      // quad_perm [0,0,0,0]
        case 1: r = __builtin_amdgcn_update_dpp(0, i, 0x55, 0xf, 0xf, true); break;      // [1,1,1,1]
        case 2: r = __builtin_amdgcn_update_dpp(0, i, 0xaa, 0xf, 0xf, true); break;      // [2,2,2,2]
        default: r = __builtin_amdgcn_update_dpp(0, i, 0xff, 0xf, 0xf, true); break;     // [3,3,3,3]
    }
    return __builtin_bit_cast(float, r);
}
// StatePrep (common.hpp): blocks beyond the downsample grid turn the caller's 16-bit states into this frame's fp32
// states - element by element the kept fp32 value where it rounds to what the caller passed, else the caller's value.
template <typename T16>
__device__ __forceinline__ void state_prep(const StatePrep& sp, int blk, int tid) {
    // a block = 4096 elements of one level: four independent (caller, kept) load pairs per thread in flight
    int l = 0, b0 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int nb = (sp.n[k] + 4095) >> 12;
        if (l == k && blk >= b0 + nb) { l = k + 1; b0 += nb; }
    }
    const int n = l == 0 ? sp.n[0] : (l == 1 ? sp.n[1] : (l == 2 ? sp.n[2] : sp.n[3]));
    const void* in = l == 0 ? sp.in[0] : (l == 1 ? sp.in[1] : (l == 2 ? sp.in[2] : sp.in[3]));
    const float* sh = l == 0 ? sp.shadow[0] : (l == 1 ? sp.shadow[1] : (l == 2 ? sp.shadow[2] : sp.shadow[3]));
    float* out = l == 0 ? sp.out[0] : (l == 1 ? sp.out[1] : (l == 2 ? sp.out[2] : sp.out[3]));
    const int i0 = ((blk - b0) << 12) + tid * 4;
    f32x4 h[4], k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = min(i0 + j * 1024, n - 4);                    // (levels hold multiples of 4 elements)
        if (in) { h[j] = ld4f((const T16*)in + i); k[j] = *(const f32x4*)(sh + i); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * 1024;
        if (i >= n) continue;
        f32x4 v = f32x4{0, 0, 0, 0};
        if (in) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((float)(T16)k[j][r] == h[j][r]) ? k[j][r] : h[j][r];
        }
        *(f32x4*)(out + i) = v;
    }
}
// two neighbouring elements of a source row, loaded as one (element-aligned) access
template <typename TIO> struct TIO2 { float a, b; };
typedef unsigned int u32_a2 __attribute__((aligned(2)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ TIO2<f16> ld_pair(const f16* p) {
    const unsigned u = *(const u32_a2*)p;
    return TIO2<f16>{(float)__builtin_bit_cast(f16, (unsigned short)(u & 0xFFFFu)), (float)__builtin_bit_cast(f16, (unsigned short)(u >> 16))};
}
__device__ __forceinline__ TIO2<bf16> ld_pair(const bf16* p) {
    const unsigned u = *(const u32_a2*)p;
    return TIO2<bf16>{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xFFFF0000u)};
}
__device__ __forceinline__ TIO2<float> ld_pair(const float* p) { const f32x2_a4 v = *(const f32x2_a4*)p; return TIO2<float>{v[0], v[1]}; }

template <typename T, typename TIO, bool STEM>
__global__ __launch_bounds__(256) void downsample(DownArgs a) {
    [[maybe_unused]] Stamper stamp;
    if constexpr (sizeof(TIO) == 2 && sizeof(T) == 4) {
        if ((int)blockIdx.x >= a.nblk) {                             // whole block: the state blocks are spread over the samples' rows of the grid
            const int sb = ((int)blockIdx.x - a.nblk) * (int)gridDim.y + (int)blockIdx.y;
            if (sb < a.sp.blocks) state_prep<TIO>(a.sp, sb, threadIdx.x);
            return;
        }
    }
    // stem weights (27 x 16) and bias through LDS: read per (tap, channel) in the conv loop below, 27 dependent global loads
    // per wave otherwise
    [[maybe_unused]] __shared__ __attribute__((aligned(16))) float stem_lds[27 * 16 + 16];
    if constexpr (STEM) {
        const int t = threadIdx.x;
        const float w0 = a.stem_w[t], w1 = a.stem_w[min(t + 256, 431)], b0 = a.stem_b[t & 15];
        stem_lds[t] = w0;
        if (t + 256 < 432) stem_lds[t + 256] = w1;
        if (t < 16) stem_lds[432 + t] = b0;
        __syncthreads();
    }
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, n = blockIdx.y;
    if (wv >= a.h3 * a.w3) return;                                   // whole wave
    const int y3 = wv / a.w3, x3 = wv - y3 * a.w3;
    const int dx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4);
    const int dy = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int oy = 8 * y3 + dy, ox = 8 * x3 + dx;
    const bool inb = oy < a.h && ox < a.w;
    const TIO* sp = (const TIO*)a.src + (size_t)n * 3 * a.H * a.W;
    const size_t plane = (size_t)a.H * a.W;
    // one s0 pixel (r, g, b rounded to the storage type; [3] unset)
    auto interp = [&](int py, int px) {
        const int cy = min(max(py, 0), a.h - 1), cx = min(max(px, 0), a.w - 1);
        float sy = a.scale_y * (cy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        float sx = a.scale_x * (cx + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
        int y0 = (int)sy, x0 = (int)sx;
        if (y0 > a.H - 1) y0 = a.H - 1;
        if (x0 > a.W - 1) x0 = a.W - 1;
        const int y1 = y0 + (y0 < a.H - 1), x1 = x0 + (x0 < a.W - 1);
        const float ly = sy - y0, lx = sx - x0;
        f32x4 r;
        // the two taps of a row are neighbours (x1 = x0 + 1, or x1 = x0 on the last column): one load of the pair
        // [xb, xb + 1], xb = min(x0, W - 2) - a 4-byte load at a 2-byte aligned address for the 16-bit types - instead of
        // two element loads: 6 instead of 12 vector-memory instructions per pixel
        const int xb = min(x0, a.W - 2);
        const bool first_hi = x0 != xb, second_lo = x1 == xb;      // (x0 == W - 1: both taps are the pair's second element)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const TIO* pl = sp + c * plane;
            float v00, v01, v10, v11;
            if (a.W >= 2) {                                          // uniform
                const TIO2<TIO> p0 = ld_pair(pl + (size_t)y0 * a.W + xb), p1 = ld_pair(pl + (size_t)y1 * a.W + xb);
                v00 = first_hi ? p0.b : p0.a; v01 = second_lo ? p0.a : p0.b;
                v10 = first_hi ? p1.b : p1.a; v11 = second_lo ? p1.a : p1.b;
            } else {
                v00 = (float)pl[(size_t)y0 * a.W + x0]; v01 = (float)pl[(size_t)y0 * a.W + x1];
                v10 = (float)pl[(size_t)y1 * a.W + x0]; v11 = (float)pl[(size_t)y1 * a.W + x1];
            }
            r[c] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            r[c] = (float)(T)r[c];
        }
        r[3] = 0.f;
        return r;
    };
    f32x4 o = interp(oy, ox);
    // stem: the conv also needs the s0 row above and the column left of the block (lanes 0..8 / 9..16; the other
    // lanes repeat their own pixel) - requested together with the block so that it stays one round trip
    [[maybe_unused]] int hy = oy, hx = ox;
    [[maybe_unused]] f32x4 hv;
    if constexpr (STEM) {
        if (lane < 9) { hy = 8 * y3 - 1; hx = 8 * x3 - 1 + lane; }
        else if (lane < 17) { hy = 8 * y3 + (lane - 9); hx = 8 * x3 - 1; }
        hv = f32x4{0, 0, 0, 0};
        if (lane < 17) hv = interp(hy, hx);                          // 17 of 64 lanes: the other lanes issue no halo loads
    }
    o[3] = (float)(T)((o[0] + o[1] + o[2]) / 3.f);
    if (inb) st4f((T*)a.s0 + (((size_t)n * a.h + oy) * a.w + ox) * 4, o);
    // ---- s1: the quad's valid pixels
    f32x4 v1 = f32x4{0, 0, 0, 0};
    float c1 = 0.f;
    {
        const float fin = inb ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float vj = quad_lane(fin, j);
            c1 += vj;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = quad_lane(o[c], j); if (vj != 0.f) v1[c] += t; }
        }
    }
    const bool ex1 = c1 > 0.f;
    if (ex1) {
        v1 = v1 / c1;
        if ((lane & 3) == 0) st4f((T*)a.s1 + (((size_t)n * a.h1 + (oy >> 1)) * a.w1 + (ox >> 1)) * 4, v1);
#pragma unroll
        for (int c = 0; c < 4; ++c) v1[c] = (float)(T)v1[c];         // the next level averages what was stored
    }
    // ---- s2: the four s1 pixels of this 16-lane group (lanes base, base+4, base+8, base+12)
    f32x4 v2 = f32x4{0, 0, 0, 0};
    float c2 = 0.f;
    {
        const float f1 = ex1 ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int srcl = (lane & 48) + 4 * j;
            const float vj = __shfl(f1, srcl);
            c2 += vj;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = __shfl(v1[c], srcl); if (vj != 0.f) v2[c] += t; }
        }
    }
    const bool ex2 = c2 > 0.f;
    if (ex2) {
        v2 = v2 / c2;
#pragma unroll
        for (int c = 0; c < 4; ++c) v2[c] = (float)(T)v2[c];
        if ((lane & 15) == 0) st4f((T*)a.s2 + (((size_t)n * a.h2 + (oy >> 2)) * a.w2 + (ox >> 2)) * 4, v2);
    }
    // ---- s3: the four s2 pixels of the wave (lanes 0, 16, 32, 48)
    f32x4 v3 = f32x4{0, 0, 0, 0};
    float c3 = 0.f;
    {
        const float f2 = ex2 ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float vj = __shfl(f2, 16 * j);
            c3 += vj;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = __shfl(v2[c], 16 * j); if (vj != 0.f) v3[c] += t; }
        }
    }
    if (lane == 0) {
        v3 = v3 / c3;
        st4f((T*)a.s3 + (((size_t)n * a.h3 + y3) * a.w3 + x3) * 4, v3);
    }
    // ---- encoder stem on the block: ImageNet normalise + 3x3 stride-2 conv 3->16 + BN + hardswish (mobilenetv3.py:37,39;
    //      torchvision features[0]): 4 x 4 output pixels x 16 channels = 64 lanes x 4 channels.  Normalisation is applied
    //      to in-bounds pixels only, so the conv's zero padding pads the *normalised* image like the reference; the
    //      9 x 9 normalised s0 patch goes through the wave's own LDS tile.
    if constexpr (STEM) {
        __shared__ f32x4 tiles[4][81];
        f32x4* tile = tiles[threadIdx.x >> 6];
        auto norm = [&](f32x4 v, bool ok) {
            f32x4 r;
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = ok ? (v[c] - a.mean[c]) / a.std_[c] : 0.f;
            r[3] = 0.f;
            return r;
        };
        tile[(dy + 1) * 9 + dx + 1] = norm(o, inb);
        if (lane < 17) tile[lane < 9 ? lane : (lane - 8) * 9] = norm(hv, hy >= 0 && hx >= 0 && hy < a.h && hx < a.w);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int p16 = lane >> 2, oq = lane & 3, py = p16 >> 2, px = p16 & 3;
        float acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = stem_lds[432 + oq * 4 + q];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 v = tile[(2 * py + ky) * 9 + 2 * px + kx];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 w = *(const f32x4*)(stem_lds + ((ky * 3 + kx) * 3 + c) * 16 + oq * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] += v[c] * w[q];
                }
            }
        const int r = 4 * y3 + py, cc = 4 * x3 + px;
        if (r < a.Ho && cc < a.Wo) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = act_f(acc[q], ACT_HSWISH);
            st4f((T*)a.stem_y + (((size_t)n * a.Ho + r) * a.Wo + cc) * 16 + oq * 4, v);
        }
    }
}
template <typename T> void launch_down(const DownArgs& a_, int N, int io, hipStream_t s) {
    DownArgs a = a_;
    a.nblk = (a.h3 * a.w3 + 3) / 4;
    if (sizeof(T) != 4 || !io) a.sp.blocks = 0;
    const dim3 grid(a.nblk + (a.sp.blocks + N - 1) / N, N);
    if constexpr (sizeof(T) == 4) {
        if (io == IO_F16) {
            if (a.stem_y) hipLaunchKernelGGL((downsample<T, f16, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((downsample<T, f16, false>), grid, dim3(256), 0, s, a);
            return;
        }
        if (io == IO_BF16) {
            if (a.stem_y) hipLaunchKernelGGL((downsample<T, bf16, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((downsample<T, bf16, false>), grid, dim3(256), 0, s, a);
            return;
        }
    }
    if (a.stem_y) hipLaunchKernelGGL((downsample<T, T, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((downsample<T, T, false>), grid, dim3(256), 0, s, a);
}
// the state preparation as a launch of its own (frame pipeline: it belongs to the decoder phase there)
template <typename T16> __global__ __launch_bounds__(256) void state_prep_k(StatePrep sp) { state_prep<T16>(sp, blockIdx.x, threadIdx.x); }
void launch_state_prep(const StatePrep& sp, int io, hipStream_t s) {
    if (sp.blocks <= 0) return;
    if (io == IO_BF16) hipLaunchKernelGGL((state_prep_k<bf16>), dim3(sp.blocks), dim3(256), 0, s, sp);
    else hipLaunchKernelGGL((state_prep_k<f16>), dim3(sp.blocks), dim3(256), 0, s, sp);
}
// Frame pipeline: ordering between its two streams by a counter in device memory.  flag_set (one thread) publishes a frame
// number behind a phase; flag_wait (one wave) holds its stream until the number is there.  It sleeps between polls and
// gives up after ~2 s of wall clock, so no wave can wait for ever.
__global__ __launch_bounds__(64) void flag_set(unsigned* flag, unsigned k) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(64) void flag_wait(const unsigned* flag, unsigned k) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - k) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 200000000ull) break;                // 100 MHz counter
    }
}
// Frame pipeline: which streams really run side by side?  A stream is bound to one of the device's few hardware queues
// (GPU_MAX_HW_QUEUES = 4, handed out in turn), and two streams on one queue take turns.  queue_probe occupies its stream for
// `ticks` of the 100 MHz wall clock and reports when it ran; the host launches one on each stream in question and compares.
__global__ __launch_bounds__(64) void queue_probe(unsigned long long* out, unsigned ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    out[0] = t0; out[1] = wall_clock64();
}
void launch_queue_probe(unsigned long long* out, unsigned ticks, hipStream_t s) { hipLaunchKernelGGL(queue_probe, dim3(1), dim3(64), 0, s, out, ticks); }
void launch_flag_set(unsigned* flag, unsigned k, hipStream_t s) { hipLaunchKernelGGL(flag_set, dim3(1), dim3(64), 0, s, flag, k); }
void launch_flag_wait(const unsigned* flag, unsigned k, hipStream_t s) { hipLaunchKernelGGL(flag_wait, dim3(1), dim3(64), 0, s, flag, k); }
template void launch_down<f16>(const DownArgs&, int, int, hipStream_t);
template void launch_down<float>(const DownArgs&, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// Refiner, low-res half: project_mat 1x1 (model.py:62) + guided-filter statistics with the
// 3x3 box filter on LDS + the 1x1 MLP (deep_guided_filter.py:26-35) or cov/(var+eps)
// (fast_guided_filter.py:51-56).  Emits A|b as [h][w][8] fp32.  All arithmetic in fp32.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dgf_lowres(DgfLowArgs a) {
    [[maybe_unused]] Stamper stamp;
    __shared__ f32x4 sx[18 * 18], sy[18 * 18];
    const DgfWeights& wt = *a.wt;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x, n = blockIdx.y;
    const int y0 = ty * 16, x0 = tx * 16;
    const T* hid = (const T*)a.hid + (size_t)n * a.h * a.w * 16;
    const T* s0 = (const T*)a.s0 + (size_t)n * a.h * a.w * 4;
    for (int pp = threadIdx.x; pp < 324; pp += 256) {
        const int py = pp / 18, px = pp - py * 18;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        f32x4 xv = f32x4{0, 0, 0, 0}, yv = xv;
        if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) {
            const size_t pix = (size_t)gy * a.w + gx;
            xv = ld4f(s0 + pix * 4);
            xv[3] = (xv[0] + xv[1] + xv[2]) / 3.f;
            F8 h0 = ld8f(hid + pix * 16), h1 = ld8f(hid + pix * 16 + 8);
            float hv[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) { hv[k] = h0.lo[k]; hv[4 + k] = h0.hi[k]; hv[8 + k] = h1.lo[k]; hv[12 + k] = h1.hi[k]; }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float s = wt.proj_b[o];
#pragma unroll
                for (int k = 0; k < 16; ++k) s += wt.proj_w[o][k] * hv[k];
                yv[o] = s;
            }
        }
        sx[pp] = xv; sy[pp] = yv;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy >= a.h || gx >= a.w) return;
    f32x4 mx = f32x4{0, 0, 0, 0}, my = mx, exy = mx, exx = mx;
    if (a.fast) {
        // separable 1/3 box, horizontal then vertical, like fast_guided_filter.py:71-76
        const float t3 = 1.0f / 3.0f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            f32x4 rx = mx * 0.f, ry = rx, rxy = rx, rxx = rx;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int pp = (ly + dy) * 18 + lx + dx;
                const f32x4 xv = sx[pp], yv = sy[pp];
                rx += xv * t3; ry += yv * t3; rxy += (xv * yv) * t3; rxx += (xv * xv) * t3;
            }
            mx += rx * t3; my += ry * t3; exy += rxy * t3; exx += rxx * t3;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pp = (ly + t / 3) * 18 + lx + t % 3;
            const f32x4 xv = sx[pp], yv = sy[pp];
            f32x4 bw;
#pragma unroll
            for (int c = 0; c < 4; ++c) bw[c] = wt.box[c][t];
            mx += bw * xv; my += bw * yv; exy += bw * (xv * yv); exx += bw * (xv * xv);
        }
    }
    // no fma contraction here: E[xy]-E[x]E[y] cancels, keep the reference's rounding sequence
    f32x4 cov, var;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cov[c] = __fsub_rn(exy[c], __fmul_rn(mx[c], my[c]));
        var[c] = __fsub_rn(exx[c], __fmul_rn(mx[c], mx[c]));
    }
    f32x4 A;
    if (a.fast) {
        A = cov / (var + 1e-5f);
    } else {
        const size_t pix = (size_t)gy * a.w + gx;
        float in[24];
#pragma unroll
        for (int k = 0; k < 4; ++k) { in[k] = cov[k]; in[4 + k] = var[k]; }
        F8 h0 = ld8f(hid + pix * 16), h1 = ld8f(hid + pix * 16 + 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) { in[8 + k] = h0.lo[k]; in[12 + k] = h0.hi[k]; in[16 + k] = h1.lo[k]; in[20 + k] = h1.hi[k]; }
        float t1[16], t2[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            float s = wt.b1[o];
#pragma unroll
            for (int k = 0; k < 24; ++k) s += wt.w1[o][k] * in[k];
            t1[o] = fmaxf(s, 0.f);
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            float s = wt.b2[o];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += wt.w2[o][k] * t1[k];
            t2[o] = fmaxf(s, 0.f);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float s = wt.b3[o];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += wt.w3[o][k] * t2[k];
            A[o] = s;
        }
    }
    const f32x4 b = my - A * mx;
    float* o = a.ab + (((size_t)n * a.h + gy) * a.w + gx) * 8;
    *(f32x4*)o = A;
    *(f32x4*)(o + 4) = b;
}
// Deep refiner, MFMA form of the same arithmetic: the per-pixel MLP (24 -> 16 -> 16 -> 4) runs as nine
// 16x16x32 f16 MFMA tiles per 16 pixels with (hi, lo) split operands (fp32-accurate, see f32s).
// Thread (x = tid & 15, row = tid >> 4) computes the box statistics of its pixel as before; wave w owns the
// four 16-pixel rows 4w..4w+3 (= MFMA tiles t).  For tile t the K slots of lane group g hold
//   (g - t) mod 4 = 0: [cov, var] of pixel (x, row t)  - exactly what lane (x, g = t) has in registers,
//                  1: hid[0..7], 2: hid[8..15] of that pixel (loaded in this layout), 3: zeros,
// layer outputs (4 channels per lane) feed the next layer as K slots [4 channels, 4 zeros], and the
// last layer's 4 real rows sit at rows 4t..4t+3 so A lands in lane (x, g = t) again: no cross-lane
// traffic.  The host packs one weight fragment per (layer, t) accordingly (engine.cpp pack_dgf_mlp).
struct HL8 { f16x8 hi, lo; };
__device__ __forceinline__ HL8 split8(f32x4 a, f32x4 b) {
    f16x4 h0, l0, h1, l1;
    split4(a, h0, l0); split4(b, h1, l1);
    HL8 r;
    r.hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    r.lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    return r;
}
__device__ __forceinline__ f32x4 mma_hl(f32x4 acc, const f16* wfrag, int lane, const HL8& x) {
    const f16x8 wh = *(const f16x8*)(wfrag + lane * 8), wl = *(const f16x8*)(wfrag + 512 + lane * 8);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, x.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, x.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, x.hi, acc, 0, 0, 0);
    return acc;
}
// The OutputBlock's second conv (decoder.py:129-131: conv3x3 16 -> 16 + BN + ReLU) runs inside this kernel: the block
// computes `hid` on its 18 x 18 halo region out of a 20 x 20 patch of the first conv's output (the same implicit GEMM
// and weight fragments as conv_kxk: K index = tap * 16 + channel, 5 K steps, one n-tile; 21 m-tiles of 16 pixels over
// the 4 waves), keeps it in LDS for the projection / the MLP and writes the interior to memory (segmentation pass,
// debug taps).  One launch and one trip of `hid` through memory less.  TC = arithmetic type of the conv.
// LDS of a block, kept under a third of the CU's 160 KB (three blocks per CU: the 510 blocks of an HD / 4K frame then
// start at once with room to spare - at two per CU a few of them waited for a second round, and a batched launch runs three
// rows of blocks per CU instead of two):
//   [0, PATCHB)  the 20 x 20 patch of the first conv's output; once the conv is done, the MLP weight fragments
//                (9 x 1024 f16 = 18 KB) and the guided filter's inputs sx, sy (2 x 324 x 16 B) move into the same bytes
//   hid tile     [18 * 18][16] fp32 without padding; the four 16-byte pieces of a pixel are XOR-swizzled by (pixel >> 2) & 3
//                so that 16 consecutive pixels reading the SAME piece hit 16 different bank sets
//   project_mat  weights [4][16] + bias [4]
constexpr int DGF_HS = 16;                                              // floats per pixel of the hid tile in LDS
__device__ __forceinline__ int dgf_hsw(int pp, int piece) { return pp * DGF_HS + ((piece ^ ((pp >> 2) & 3)) << 2); }
template <typename TC> constexpr int dgf_patch_csl() { return 16 + Tr<TC>::kLdsPad; }
template <typename TC> constexpr size_t dgf_patch_bytes() {
    size_t patch = (size_t)20 * 20 * dgf_patch_csl<TC>() * sizeof(TC);
    const size_t later = (size_t)9 * 1024 * 2 + (size_t)2 * 18 * 18 * 16;   // what moves in once the patch is dead
    return patch < later ? later : patch;
}
template <typename TC> constexpr size_t dgf_lds_bytes() { return dgf_patch_bytes<TC>() + (size_t)18 * 18 * DGF_HS * 4 + 68 * 4 + 16; }
template <typename T, typename TC>
__global__ __launch_bounds__(256) void dgf_lowres_mfma(DgfLowArgs a) {
    [[maybe_unused]] Stamper stamp;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr int CSL = dgf_patch_csl<TC>();
    constexpr size_t PATCHB = dgf_patch_bytes<TC>();
    TC* patch = (TC*)dsm;                                               // [20 * 20][CSL], later wl | sx | sy
    f16* wl = (f16*)dsm;                                                // W1[t] x4, W2, W3[t] x4 (9 x 1024 f16)
    f32x4* sx = (f32x4*)(dsm + 9 * 1024 * 2);
    f32x4* sy = sx + 18 * 18;
    float* hidt = (float*)(dsm + PATCHB);                               // [18 * 18][DGF_HS], pieces swizzled (dgf_hsw)
    const DgfWeights& wt = *a.wt;
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x, n = blockIdx.y;
    const int y0 = ty * 16, x0 = tx * 16;
    const T* s0 = (const T*)a.s0 + (size_t)n * a.h * a.w * 4;
    const int lane = tid & 63, lx = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int gx = x0 + lx;
    // Everything this block reads from memory is requested up front, branch-free (clamped addresses, masked
    // afterwards): MLP weight fragments (18 KiB, 16-byte pieces), the conv's weight fragments and bias, the 18 x 18
    // halo of s0 (two rounds of 256 threads); then the 20 x 20 x 16 patch of the first conv's output is staged.
    f32x4 wreg[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { const int i = min(tid + 256 * j, 1151); wreg[j] = ((const f32x4*)a.mlp)[i]; }
    Frag<TC> cwf[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) cwf[ks].loadw((const TC*)a.cw + (size_t)ks * 512, lane);
    const f32x4 cb = *(const f32x4*)(a.cbias + g * 4);
    // (the MLP's biases too: asked for where they are used they were a round trip in front of the first layer)
    const f32x4 b1 = *(const f32x4*)(wt.b1 + 4 * g), b2 = *(const f32x4*)(wt.b2 + 4 * g), b3 = *(const f32x4*)wt.b3;
    f32x4 xr[2]; bool inb[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pp = min(tid + 256 * r, 323);
        const int py = pp / 18, px = pp - py * 18;
        const int gy = y0 + py - 1, gxx = x0 + px - 1;
        inb[r] = gy >= 0 && gy < a.h && gxx >= 0 && gxx < a.w;
        xr[r] = ld4f(s0 + (inb[r] ? (size_t)gy * a.w + gxx : 0) * 4);
    }
    float* pjl = hidt + 18 * 18 * DGF_HS;                               // project_mat weights [4][16] + bias [4], via LDS
    const float pjv = ((const float*)&wt.proj_w[0][0])[min(tid, 67)];   // proj_w and proj_b are adjacent in DgfWeights
    {
        const ConvSrc S = ConvSrc{a.hin, 16, 16, 0, 0, 0, 0, 0, 0};
        // 1600 items over 256 threads: 7 per thread, all in flight at once
        stage_source<TC, false, sizeof(T) == 2, 20, 20, 7>(patch, S, 0, 4, 0, CSL, n, y0 - 2, x0 - 2, a.h, a.w, tid, 256);
    }
    if (tid < 68) pjl[tid] = pjv;
    __syncthreads();
    stamp(1);
    // ---- hid = relu(conv3x3(patch) + bias) on the 18 x 18 region
    // three m-tiles at a time per wave: independent accumulators keep the MFMA pipe busy (one tile alone is a chain of
    // 15 dependent MFMAs); two such rounds cover the wave's 5-6 tiles.  The results go to the LDS tile at once, to
    // memory (the block's own 16 x 16 pixels: segmentation pass, debug taps) only after both rounds - a store between
    // the rounds would have to complete before its registers are reused.
    T* hout = (T*)a.hid + (size_t)n * a.h * a.w * 16;
    f32x4 hv_[2][3];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int mt0 = wave + 12 * rd;
        int pp[3], py[3], px[3];
        f32x4 acc[3][1];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            pp[u] = min((mt0 + 4 * u) * 16 + lx, 323);
            py[u] = pp[u] / 18; px[u] = pp[u] - py[u] * 18;
            acc[u][0] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int q8 = ks * 4 + g, tap = min(q8 >> 1, 8), c8 = q8 & 1;      // q8 >= 18: zero weights, any finite operand
            const int dy = tap / 3, dx = tap - dy * 3;
            Frag<TC> xf[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) xf[u].load_lds(patch + ((py[u] + dy) * 20 + px[u] + dx) * CSL + c8 * 8);
            const Frag<TC> w1[1] = {cwf[ks]};
            mma_tile(acc, w1, xf);                                              // term-major over the three tiles
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            f32x4 v = acc[u][0] + cb;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (float)(T)fmaxf(v[r], 0.f);       // as it would come back from memory
            hv_[rd][u] = v;
            if ((mt0 + 4 * u) * 16 + lx < 324) *(f32x4*)(hidt + dgf_hsw(pp[u], g)) = v;
        }
    }
    stamp(2);
    if (a.hid)                                                                  // (null: nobody reads hid from memory, it stays in LDS)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int p = (wave + 12 * rd + 4 * u) * 16 + lx;
            const int py = p / 18, px = p - py * 18;
            const int gy = y0 + py - 1, gxx = x0 + px - 1;
            if (p < 324 && py >= 1 && py <= 16 && px >= 1 && px <= 16 && gy < a.h && gxx < a.w)
                st4f(hout + ((size_t)gy * a.w + gxx) * 16 + g * 4, hv_[rd][u]);
        }
    stamp(3);
    __syncthreads();
    stamp(4);
    // ---- guided-filter inputs of the halo: x = [s0 rgb, gray], y = project_mat(hid) (model.py:62); MLP weights into LDS
    f32x4 hs[4][2];
    {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pp = tid + 256 * r;
            if (pp < 324) {
                f32x4 xv = f32x4{0, 0, 0, 0}, yv = xv;
                if (inb[r]) {
                    xv = xr[r];
                    xv[3] = (xv[0] + xv[1] + xv[2]) / 3.f;
                    float hv[16];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 t = *(const f32x4*)(hidt + dgf_hsw(pp, k));
                        hv[4 * k] = t[0]; hv[4 * k + 1] = t[1]; hv[4 * k + 2] = t[2]; hv[4 * k + 3] = t[3];
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        float sacc = pjl[64 + o];
#pragma unroll
                        for (int k = 0; k < 16; ++k) sacc += pjl[o * 16 + k] * hv[k];
                        yv[o] = sacc;
                    }
                }
                sx[pp] = xv; sy[pp] = yv;
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) { const int i = tid + 256 * j; if (i < 1152) ((f32x4*)wl)[i] = wreg[j]; }
        // the hid slices this lane supplies to the other three tiles of its wave
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int rg = (g - t) & 3;
            const int gyt = y0 + 4 * wave + t;
            const bool ok = (rg == 1 || rg == 2) && gyt < a.h && gx < a.w;
            const int hpp = (4 * wave + t + 1) * 18 + lx + 1, hpc = ok ? (rg - 1) * 2 : 0;
            const f32x4 h0 = *(const f32x4*)(hidt + dgf_hsw(hpp, hpc)), h1 = *(const f32x4*)(hidt + dgf_hsw(hpp, hpc + 1));
            hs[t][0] = ok ? h0 : f32x4{0, 0, 0, 0};
            hs[t][1] = ok ? h1 : f32x4{0, 0, 0, 0};
        }
    }
    __syncthreads();
    stamp(5);
    // ---- box statistics of this thread's pixel (x = lx, row = 4 * wave + g)
    const int ly = 4 * wave + g;
    f32x4 mx = f32x4{0, 0, 0, 0}, my = mx, exy = mx, exx = mx;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int pp = (ly + t / 3) * 18 + lx + t % 3;
        const f32x4 xv = sx[pp], yv = sy[pp];
        f32x4 bw;
#pragma unroll
        for (int c = 0; c < 4; ++c) bw[c] = wt.box[c][t];
        mx += bw * xv; my += bw * yv; exy += bw * (xv * yv); exx += bw * (xv * xv);
    }
    // no fma contraction here: E[xy]-E[x]E[y] cancels, keep the reference's rounding sequence
    f32x4 cov, var;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cov[c] = __fsub_rn(exy[c], __fmul_rn(mx[c], my[c]));
        var[c] = __fsub_rn(exx[c], __fmul_rn(mx[c], mx[c]));
    }
    stamp(6);
    // ---- MLP on the MFMA, tile by tile
    const f32x4 z4 = f32x4{0, 0, 0, 0};
    f32x4 A = z4;
    {
        // the four tiles layer by layer, each layer's three MFMA terms across the four tiles: a tile's nine MFMAs are one
        // dependent chain, the tiles are independent of each other
        f32x4 m[4];
        HL8 in[4];
        auto layer = [&](const f16* w, int wstride) {
            f16x8 wh[4], wlo[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f16* wf = w + t * wstride;
                wh[t] = *(const f16x8*)(wf + lane * 8); wlo[t] = *(const f16x8*)(wf + 512 + lane * 8);
                m[t] = z4;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[t], in[t].hi, m[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], in[t].lo, m[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], in[t].hi, m[t], 0, 0, 0);
        };
#pragma unroll
        for (int t = 0; t < 4; ++t) in[t] = (g == t) ? split8(cov, var) : split8(hs[t][0], hs[t][1]);
        layer(wl, 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 v = m[t] + b1;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            in[t] = split8(v, z4);
        }
        layer(wl + 4 * 1024, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 v = m[t] + b2;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            in[t] = split8(v, z4);
        }
        layer(wl + 5 * 1024, 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) if (g == t) A = m[t];
    }
    A += b3;
    const int gy = y0 + ly;
    if (gy >= a.h || gx >= a.w) return;
    const f32x4 bq = my - A * mx;
    float* o = a.ab + (((size_t)n * a.h + gy) * a.w + gx) * 8;
    *(f32x4*)o = A;
    *(f32x4*)(o + 4) = bq;
    stamp(7);
}
template <typename T> void launch_dgf_low(const DgfLowArgs& a, int N, hipStream_t s) {
    const int tiles_y = (a.h + 15) / 16;
    const dim3 grid(a.tiles_x * tiles_y, N);
    if (a.cw) {                                                           // deep guided filter with the OutputBlock's second conv inside
        if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((dgf_lowres_mfma<T, f16>), grid, dim3(256), dgf_lds_bytes<f16>(), s, a);
        else if (a.csplit) hipLaunchKernelGGL((dgf_lowres_mfma<T, f32s>), grid, dim3(256), dgf_lds_bytes<f32s>(), s, a);
        else hipLaunchKernelGGL((dgf_lowres_mfma<T, float>), grid, dim3(256), dgf_lds_bytes<float>(), s, a);
    } else hipLaunchKernelGGL((dgf_lowres<T>), grid, dim3(256), 0, s, a);
}
template void launch_dgf_low<f16>(const DgfLowArgs&, int, hipStream_t);
template void launch_dgf_low<float>(const DgfLowArgs&, int, hipStream_t);


// ------------------------------------------------------------------------------------------
// downsample_ratio == 1 tail (no refiner): project_mat + residual + clamp (model.py:62,65-67)
// or project_seg (model.py:70) when seg != 0 (written to `pha`, no clamp).
// ------------------------------------------------------------------------------------------
template <typename T, typename TIO>
__global__ __launch_bounds__(256) void proj_out(ProjOutArgs a) {
    [[maybe_unused]] Stamper stamp;
    const int p = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    const int HW = a.H * a.W;
    if (p >= HW) return;
    const DgfWeights& wt = *a.wt;
    const T* hp = (const T*)a.hid + ((size_t)n * HW + p) * 16;
    F8 h0 = ld8f(hp), h1 = ld8f(hp + 8);
    float hv[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) { hv[k] = h0.lo[k]; hv[4 + k] = h0.hi[k]; hv[8 + k] = h1.lo[k]; hv[12 + k] = h1.hi[k]; }
    if (a.seg) {
        float s = wt.seg_b;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += wt.seg_w[k] * hv[k];
        ((TIO*)a.pha)[(size_t)n * HW + p] = (TIO)s;
        return;
    }
    float y[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float s = wt.proj_b[o];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += wt.proj_w[o][k] * hv[k];
        y[o] = s;
    }
    const TIO* sp = (const TIO*)a.src + (size_t)n * 3 * HW + p;
    TIO* fp = (TIO*)a.fgr + (size_t)n * 3 * HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) fp[(size_t)c * HW] = (TIO)fminf(fmaxf(y[c] + (float)sp[(size_t)c * HW], 0.f), 1.f);
    ((TIO*)a.pha)[(size_t)n * HW + p] = (TIO)fminf(fmaxf(y[3], 0.f), 1.f);
}
template <typename T> void launch_proj_out(const ProjOutArgs& a, int N, int io, hipStream_t s) {
    const dim3 grid((a.H * a.W + 255) / 256, N);
    if (sizeof(T) == 4 && io == IO_F16) hipLaunchKernelGGL((proj_out<T, f16>), grid, dim3(256), 0, s, a);
    else if (sizeof(T) == 4 && io == IO_BF16) hipLaunchKernelGGL((proj_out<T, bf16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((proj_out<T, T>), grid, dim3(256), 0, s, a);
}
template void launch_proj_out<f16>(const ProjOutArgs&, int, int, hipStream_t);
template void launch_proj_out<float>(const ProjOutArgs&, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// Output step of the reference driver on the device (inference.py:116,137-141 and the writers'
// `frames.mul(255).byte()` + NCHW->NHWC, inference_utils.py:44): from fgr [N,3,H,W] / pha [N,1,H,W] to
// uint8 [N,H,W,C] frames in one pass.  Every tensor op of the reference rounds to the tensor dtype, so
// the arithmetic below is written in that type (TIO) op by op: the bytes are identical.
//   GREEN      com = fgr * pha + bgr * (1 - pha)            C = 3   (video composition, green screen)
//   RGBA       cat[fgr * (pha > 0), pha]                   C = 4   (png_sequence composition)
//   ALPHA      pha repeated to 3 channels                  C = 3   (VideoWriter grayscale -> RGB)
//   FOREGROUND fgr                                         C = 3
// ------------------------------------------------------------------------------------------
template <typename TIO>
__global__ __launch_bounds__(256) void compose_u8(ComposeArgs a) {
    [[maybe_unused]] Stamper stamp;
    const size_t plane = (size_t)a.H * a.W;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= plane) return;
    const TIO al = ((const TIO*)a.pha)[(size_t)n * plane + p];
    const int C = a.mode == COMPOSE_RGBA ? 4 : 3;
    unsigned char* o = a.out + ((size_t)n * plane + p) * C;
    TIO f0 = (TIO)0.f, f1 = (TIO)0.f, f2 = (TIO)0.f;
    if (a.mode != COMPOSE_ALPHA) {
        const TIO* f = (const TIO*)a.fgr + (size_t)n * 3 * plane + p;
        f0 = f[0]; f1 = f[plane]; f2 = f[2 * plane];
    }
    const unsigned px = compose_px<TIO>(a.mode, f0, f1, f2, al, a.bgr);
    o[0] = px & 0xFF; o[1] = (px >> 8) & 0xFF; o[2] = (px >> 16) & 0xFF;
    if (C == 4) o[3] = px >> 24;
}
void launch_compose(const ComposeArgs& a, int N, int io, hipStream_t s) {
    const dim3 grid((unsigned)(((size_t)a.H * a.W + 255) / 256), N);
    if (io == IO_F16) hipLaunchKernelGGL((compose_u8<f16>), grid, dim3(256), 0, s, a);
    else if (io == IO_BF16) hipLaunchKernelGGL((compose_u8<bf16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((compose_u8<float>), grid, dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// Matting quality metrics of the reference's evaluator (evaluation/evaluate_hr.py:176-235), one pass over a
// clip: MAD (:176-178), MSE (:181-183), GRAD (:186-227: 9x9 Gaussian-derivative filters, sigma 1.4,
// cross-correlation with replicate border as kornia.filters.filter2d does it, magnitude, squared
// difference summed), dtSSD (:230-235) and the foreground MSE over true_pha > 0 (:169-171).
// Block = one 32 x 32 tile of one frame: the 40 x 40 halo of both mattes goes to LDS, the filter runs
// separably (rows then columns) out of LDS, and the block leaves six partial sums; a second launch adds the
// partial sums of a frame in a fixed order in double precision.  HBM-bound: every input is read once.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void matte_metrics(MetricArgs a) {
    constexpr int TS = METRIC_TILE, R = 4, PS = TS + 2 * R;             // 40
    __shared__ float im[2][PS][PS + 1];                                 // pred, true with halo
    __shared__ float hx[2][PS][TS + 1], hg[2][PS][TS + 1];              // row pass: derivative / smoothing along x
    __shared__ float red[4][METRIC_NSUM];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const size_t plane = (size_t)a.H * a.W;
    const float* P = a.pred + (size_t)n * plane;
    const float* T = a.tru + (size_t)n * plane;
    for (int i = tid; i < PS * PS; i += 256) {
        const int py = i / PS, px = i - py * PS;
        const int gy = min(max(y0 + py - R, 0), a.H - 1), gx = min(max(x0 + px - R, 0), a.W - 1);   // replicate border
        im[0][py][px] = P[(size_t)gy * a.W + gx];
        im[1][py][px] = T[(size_t)gy * a.W + gx];
    }
    __syncthreads();
    float s[METRIC_NSUM] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- pointwise sums over the tile's own pixels
    const float* Pp = n > 0 ? P - plane : a.pred_prev;
    const float* Tp = n > 0 ? T - plane : a.tru_prev;
    for (int i = tid; i < TS * TS; i += 256) {
        const int ly = i / TS, lx = i - ly * TS;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= a.H || gx >= a.W) continue;
        const float p = im[0][ly + R][lx + R], t = im[1][ly + R][lx + R];
        const float d = p - t;
        s[0] += fabsf(d);
        s[1] += d * d;
        if (Pp) {
            const size_t o = (size_t)gy * a.W + gx;
            const float dt = (p - Pp[o]) - (t - Tp[o]);
            s[3] += dt * dt;
        }
        if (a.pred_fgr && t > 0.f) {
            const size_t o = (size_t)n * 3 * plane + (size_t)gy * a.W + gx;
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float e = a.pred_fgr[o + c * plane] - a.tru_fgr[o + c * plane]; s[4] += e * e; }
            s[5] += 3.f;
        }
    }
    // ---- row pass of the separable filter
    for (int i = tid; i < 2 * PS * TS; i += 256) {
        const int m = i / (PS * TS), r = (i - m * PS * TS) / TS, c = i % TS;
        float ax = 0.f, ag = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) { const float v = im[m][r][c + j]; ax += a.dg[j] * v; ag += a.g[j] * v; }
        hx[m][r][c] = ax; hg[m][r][c] = ag;
    }
    __syncthreads();
    // ---- column pass, gradient magnitude, squared difference
    for (int i = tid; i < TS * TS; i += 256) {
        const int ly = i / TS, lx = i - ly * TS;
        if (y0 + ly >= a.H || x0 + lx >= a.W) continue;
        float mag[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float fx = 0.f, fy = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) { fx += a.g[k] * hx[m][ly + k][lx]; fy += a.dg[k] * hg[m][ly + k][lx]; }
            mag[m] = sqrtf(fx * fx + fy * fy);
        }
        const float e = mag[1] - mag[0];
        s[2] += e * e;
    }
    // ---- block sums (fixed order)
#pragma unroll
    for (int k = 0; k < METRIC_NSUM; ++k) s[k] = wave_sum(s[k]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < METRIC_NSUM; ++k) red[tid >> 6][k] = s[k];
    }
    __syncthreads();
    if (tid < METRIC_NSUM) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
        a.partial[((size_t)n * nblk + blk) * METRIC_NSUM + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}
__global__ __launch_bounds__(64) void matte_metrics_sum(MetricArgs a, int nblk) {
    const int n = blockIdx.x, k = threadIdx.x;
    if (k >= METRIC_NSUM) return;
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) acc += (double)a.partial[((size_t)n * nblk + b) * METRIC_NSUM + k];
    a.out[(size_t)n * METRIC_NSUM + k] = acc;
}
void launch_metrics(const MetricArgs& a, int N, hipStream_t s) {
    hipLaunchKernelGGL(matte_metrics, dim3(a.tiles_x, a.tiles_y, N), dim3(256), 0, s, a);
    hipLaunchKernelGGL(matte_metrics_sum, dim3(N), dim3(64), 0, s, a, a.tiles_x * a.tiles_y);
}

// Scratch of the allocation-free entry point rvm_metrics_static (include/rvm_mi355x.h): the per-tile partial sums of
// matte_metrics live in this file-scope device array instead of a caller-provided workspace.  Every GPU the library
// runs on holds its own copy (METRIC_SCRATCH_BYTES, until the process ends) at its own address; the array's NAME is
// only a host-side handle, so the device address is looked up once per device with hipGetSymbolAddress (with that
// device current) and every kernel argument uses that pointer.  Nothing in it has to start at zero: matte_metrics
// writes every slot matte_metrics_sum reads.  Calls that overlap on one GPU share the array and race.
static __device__ __align__(256) float scratch[METRIC_SCRATCH_BYTES / sizeof(float)];
static float* g_scratch_ptr[64];                       // per HIP device ordinal
static std::mutex g_scratch_mu;
int metrics_scratch(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return 2;
        g_scratch_ptr[dev] = (float*)p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

// ------------------------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max pool on NHWC (torchvision ResNet.maxpool, reference resnet.py:25).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s2(MaxPoolArgs a) {
    [[maybe_unused]] Stamper stamp;
    const int CG = a.C >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (idx >= a.Ho * a.Wo * CG) return;
    const int p = idx / CG, cg = idx - p * CG;
    const int oy = p / a.Wo, ox = p - oy * a.Wo;
    const T* xb = (const T*)a.x + (size_t)n * a.Hi * a.Wi * a.C + cg * 8;
    f32x4 m0 = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, m1 = m0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if (iy < 0 || iy >= a.Hi) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (ix < 0 || ix >= a.Wi) continue;
            const F8 v = ld8f(xb + ((size_t)iy * a.Wi + ix) * a.C);
#pragma unroll
            for (int r = 0; r < 4; ++r) { m0[r] = fmaxf(m0[r], v.lo[r]); m1[r] = fmaxf(m1[r], v.hi[r]); }
        }
    }
    F8 o; o.lo = m0; o.hi = m1;
    st8f((T*)a.y + ((size_t)n * a.Ho * a.Wo + p) * a.C + cg * 8, o);
}
template <typename T> void launch_maxpool(const MaxPoolArgs& a, int N, hipStream_t s) {
    const int total = a.Ho * a.Wo * (a.C >> 3);
    hipLaunchKernelGGL((maxpool3s2<T>), dim3((total + 255) / 256, N), dim3(256), 0, s, a);
}
template void launch_maxpool<f16>(const MaxPoolArgs&, int, hipStream_t);
template void launch_maxpool<float>(const MaxPoolArgs&, int, hipStream_t);

// ------------------------------------------------------------------------------------------
template <typename T, int MW, int NW, int KD, int S, int D, bool SPLIT = false, bool WRES = false> static void conv_attr() {
    (void)hipFuncSetAttribute((const void*)conv_kxk<T, MW, NW, KD, S, D, SPLIT, WRES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T> static void conv_attr_all() {
    conv_attr<T, 4, 1, 3, 1, 1>(); conv_attr<T, 2, 2, 3, 1, 1>(); conv_attr<T, 2, 1, 3, 1, 1>();
    conv_attr<T, 1, 2, 3, 1, 1>(); conv_attr<T, 1, 1, 3, 1, 1>();
    conv_attr<T, 4, 1, 3, 1, 1, false, true>(); conv_attr<T, 2, 2, 3, 1, 1, false, true>(); conv_attr<T, 2, 1, 3, 1, 1, false, true>();
    conv_attr<T, 1, 2, 3, 1, 1, false, true>(); conv_attr<T, 1, 1, 3, 1, 1, false, true>();
    conv_attr<T, 1, 2, 3, 2, 1>(); conv_attr<T, 1, 2, 3, 1, 2>(); conv_attr<T, 1, 2, 7, 2, 1>();
    conv_attr<T, 5, 1, 3, 1, 1>(); conv_attr<T, 5, 2, 3, 1, 1>(); conv_attr<T, 5, 1, 3, 1, 2>(); conv_attr<T, 5, 2, 3, 1, 2>();
    conv_attr<T, 5, 1, 3, 1, 1, true>(); conv_attr<T, 5, 2, 3, 1, 1, true>(); conv_attr<T, 5, 1, 3, 1, 2, true>(); conv_attr<T, 5, 2, 3, 1, 2, true>();
    conv_attr<T, 5, 4, 3, 1, 1>(); conv_attr<T, 5, 4, 3, 1, 2>(); conv_attr<T, 5, 4, 3, 1, 1, true>(); conv_attr<T, 5, 4, 3, 1, 2, true>();
    conv_attr<T, 2, 2, 3, 1, 1, true>(); conv_attr<T, 2, 1, 3, 1, 1, true>(); conv_attr<T, 1, 2, 3, 1, 1, true>();
    conv_attr<T, 1, 2, 3, 2, 1, true>(); conv_attr<T, 1, 2, 3, 1, 2, true>();
}
void kernels_init() {
    mbf_init();
    gru_fused_init();
    dgf_apply_init();
    conv_cat_init();
    conv_attr_all<f16>(); conv_attr_all<float>(); conv_attr_all<f32s>();
    (void)hipFuncSetAttribute((const void*)pw_gemm_lds<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwl_lds_bytes<1>());
    (void)hipFuncSetAttribute((const void*)pw_gemm_lds<f32s, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwl_lds_bytes<1>());
    (void)hipFuncSetAttribute((const void*)pw_gemm_lds<float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwl_lds_bytes<2>());
    (void)hipFuncSetAttribute((const void*)pw_gemm_lds<f32s, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwl_lds_bytes<2>());
    (void)hipFuncSetAttribute((const void*)dgf_lowres_mfma<f16, f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dgf_lds_bytes<f16>());
    (void)hipFuncSetAttribute((const void*)dgf_lowres_mfma<float, f32s>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dgf_lds_bytes<f32s>());
    (void)hipFuncSetAttribute((const void*)dgf_lowres_mfma<float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dgf_lds_bytes<float>());
#define RVM_ED_ATTR1(TT, KK, KS_) \
    (void)hipFuncSetAttribute((const void*)mbconv_ed<TT, KK, KS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)hipFuncSetAttribute((const void*)mbconv_ed<TT, KK, KS_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define RVM_ED_ATTR(TT, KK) RVM_ED_ATTR1(TT, KK, 0) RVM_ED_ATTR1(TT, KK, 1) RVM_ED_ATTR1(TT, KK, 2) RVM_ED_ATTR1(TT, KK, 3) RVM_ED_ATTR1(TT, KK, 4)
    RVM_ED_ATTR(f16, 3) RVM_ED_ATTR(f16, 5) RVM_ED_ATTR(float, 3) RVM_ED_ATTR(float, 5) RVM_ED_ATTR(f32s, 3) RVM_ED_ATTR(f32s, 5)
#undef RVM_ED_ATTR
#undef RVM_ED_ATTR1
    (void)hipFuncSetAttribute((const void*)mbconv_ed<f16, 3, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mbconv_ed<float, 3, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mbconv_ed<f32s, 3, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mbconv_ed<f16, 3, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mbconv_ed<float, 3, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mbconv_ed<f32s, 3, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

#ifdef RVM_STAMPS
#include "mbconv_fused.hip"   // the stamps build is one translation unit (device_common.hpp)
#include "gru_fused.hip"
#include "dgf_apply.hip"
#include "conv_cat.hip"
#endif
