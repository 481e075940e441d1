// Shared device/host definitions for the RVM MI355X (gfx950) engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
// bfloat16 at the API boundary only (src, fgr, pha, r1..r4 of a .bfloat16() module): 1-8-7 bits, converted to / from
// fp32 with round-to-nearest-even; arithmetic on it is done in fp32 and rounded per operation, like torch does.
struct bf16 {
    unsigned short v;
    bf16() = default;
    __host__ __device__ bf16(float f) {
        unsigned u = __builtin_bit_cast(unsigned, f);
        if ((u & 0x7fffffffu) > 0x7f800000u) { v = (unsigned short)((u >> 16) | 0x40u); return; }   // NaN stays NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        v = (unsigned short)(u >> 16);
    }
    __host__ __device__ operator float() const { return __builtin_bit_cast(float, (unsigned)v << 16); }
};
__host__ __device__ inline bf16 bf16_bits(unsigned short u) { bf16 b; b.v = u; return b; }
__host__ __device__ inline bf16 operator*(bf16 a, bf16 b) { return bf16((float)a * (float)b); }
__host__ __device__ inline bf16 operator+(bf16 a, bf16 b) { return bf16((float)a + (float)b); }
__host__ __device__ inline bf16 operator-(bf16 a, bf16 b) { return bf16((float)a - (float)b); }
__host__ __device__ inline bool operator>(bf16 a, bf16 b) { return (float)a > (float)b; }
// element type of the tensors at the API boundary when it differs from the kernels' storage type
enum IoMode { IO_SAME = 0, IO_F16 = 1, IO_BF16 = 2 };
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Tr;
template <> struct Tr<f16> {
    typedef f16x4 v4;
    typedef f16x8 v8;
    static constexpr int kLdsPad = 8;   // elements of padding per LDS pixel row (16 B)
};
template <> struct Tr<float> {
    typedef f32x4 v4;
    typedef f32x8 v8;
    static constexpr int kLdsPad = 4;
};
// "split" element type of the GEMM kernels: fp32 in memory, but every MFMA operand is the pair
// (hi, lo) = (fp16(x), fp16(x - hi)) so that  w*x ~= wh*xh + wh*xl + wl*xh  runs on
// v_mfma_f32_16x16x32_f16 (3 instructions of 16 cycles per K=32 step instead of 8 of 32 cycles on
// the fp32 MFMA) with ~2^-21 relative error per product and fp32 accumulation.
struct f32s { float v; };
template <> struct Tr<f32s> {
    static constexpr int kLdsPad = 4;
};

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2, ACT_HSIGMOID = 3, ACT_SIGMOID = 4 };

// ---- pointwise (1x1) GEMM ----------------------------------------------------------------
struct PwArgs {
    const void* x;  int Cin; int xs;        // activations [N][HW][xs], Cin % 8 == 0
    const void* w;  int KS;                 // weight fragments [NT][KS][64][8]
    const float* bias;                      // [NT*16]
    const void* in_scale;                   // T [N][Cin] (squeeze-excite) or null
    const void* res; int rs;                // residual T [N][HW][rs] or null
    const float* out_scale; int oss;        // float [N][oss] (LR-ASPP gate) or null
    void* y; int Cout; int ys;              // out [N][HW][ys]
    float* pool;                            // [N][MT][NT*16] per-m-tile column sums or null
    int HW; int MT; int NT; int act;
    int act_post;                           // apply `act` after the residual add (resnet bottleneck)
    int sstride, sHi, sWi, sWo;             // sstride > 1: strided 1x1 (input HxW = sHi x sWi, output width sWo)
    // Small squeeze-excite layers (S <= 32, Cin <= 256, <= 256 partial rows): the SE tail runs in this kernel's prologue
    // instead of a gemv launch: scale = hardsigmoid(fc2(relu(xscale * sum of fc1 partials + b1)) + b2); in_scale == null
    const float* se_hpart; int se_nparts, se_S; float se_xscale;    // partial fc1 pre-activations [N][nparts][S]
    const float* se_b1; const void* se_w2 /* T [Cin][S] */; const float* se_b2;
    // pw_gemm_lds split-K: K steps in `nsplit` slices over blockIdx.z, raw partial sums [nsplit][N][HW][NT*16], reduce pass
    float* part; int nsplit;
    int nsamp, w_major;                // pw_gemm_lds (set by launch_pw): samples in the launch; tile order for the XCD-aware mapping
};

// ---- fused expand (1x1, MFMA) -> depthwise k x k (LDS) [-> squeeze-excite fc1 partial] -----------
struct EdArgs {
    const void* x; int Cin; int xs;         // block input [N][Hi*Wi][xs]
    const void* w; int KS; int NT;          // expand weight fragments [NT][KS][512]; null: no expand (C == Cin)
    const float* ebias; int eact;           // [NT*16]
    const float* dw_w; const float* dw_b;   // [K*K][C], [C] (BN folded)
    int dact;
    void* y;                                // depthwise output [N][Ho*Wo][C]
    const float* fc1_w; int S; float* hpart;// SE: fc1 weights [S][C] (fp32), partial pre-activations [N][nparts][S]; or null
    int C, Hi, Wi, Ho, Wo, stride, dil, pad;
    int rb, nband;                          // output rows per band, number of bands
    int cwt, ncol;                          // output columns per column tile and their number (ncol <= 1: full-width bands)
    int cw;                                 // channels per block: 16, or 8 (already expanded input only)
    int npxs;                               // pixels per channel-quad plane of the LDS tile (>= rows*Wi of any band, multiple of 16)
    int dbg;                                // ablation switches for timing experiments (RVM_DBG env), 0 in production
    // block 1 (16 -> 16 channels, no expand, no SE): the 1x1 project + BN + residual run in the same launch
    const float* pj_w; const float* pj_b;   // [16 in][16 out] fp32 (BN folded), [16]; null: none.  y is then the block's output
};

// ---- whole inverted-residual block without squeeze-excite in one launch (mbconv_fused.hip) ---------------
// 1x1 expand (MFMA) -> depthwise 3x3 (LDS) -> 1x1 project (MFMA) [+ residual] per tile of output pixels; the expanded
// tensor and the depthwise output never leave the CU.
struct MbfArgs {
    const void* x; int Cin;                 // block input [N][Hi*Wi][Cin]
    const void* we; int KSe, NTe;           // expand weight fragments [NTe][KSe][512]
    const float* ebias; int act;            // [NTe*16]; activation of the expand and the depthwise conv (ReLU / hardswish)
    const float* dw_w; const float* dw_b;   // [9][C], [C] (BN folded)
    const void* wp; int KSp, NTp;           // project weight fragments [NTp][KSp][512]
    const float* pbias;                     // [NTp*16]
    const void* res;                        // residual [N][Ho*Wo][Cout] (the block input) or null
    void* y; int Cout;                      // block output [N][Ho*Wo][Cout]
    int C;                                  // expanded channels (multiple of 4)
    int Hi, Wi, Ho, Wo, stride;
    int THo, TWo, tiles_x, tiles_y;         // output tile (powers of two, THo even, THo * TWo in {16, 32})
    // derived by mbf_prepare (host): halo geometry and the LDS map, so that the kernel computes none of it
    int THi, TWi, NPXi, MTe, npx, TP, MTp, CSLx, CSLd, Cq, KP, ngE, padq, twl, hpl;
    unsigned invTWi, invCq;
    int mulE, mulP;
    int off_e, off_d, off_w, lds_total, red_bytes;
    // features[1] folded in (stride-2 block with 16 input channels only): the block's input is computed in place from the
    // stem output - depthwise 3x3 + ReLU, 16 x 16 project, residual - and also written out (the decoder's f1 skip)
    const void* pre_x;                      // stem output [N][Hi*Wi][16], or null: x is read as it is
    const float* pre_dw_w; const float* pre_dw_b;   // [9][16], [16]
    const float* pre_pj_w; const float* pre_pj_b;   // [16 in][16 out], [16]
    void* pre_y;                            // features[1] output [N][Hi*Wi][16]
    int pre_pw, pre_npx; unsigned inv_pre_pw;       // stem patch: (THi + 2) x (TWi + 2) pixels (derived)
};
bool mbf_supported(int stride, int KSe, int act);    // a kernel instantiation exists for the combination
bool mbf_prepare(MbfArgs& a, int es);                // fills the derived fields; false: the tile does not fit the kernel
template <typename T> void launch_mbf(const MbfArgs& a, int N, hipStream_t s);
void mbf_init();                                     // one-time function attributes

// ---- small dense layer on pooled vectors (SE fc1/fc2, LR-ASPP gate) ------------------------
struct GemvArgs {
    const float* xin; int nparts; int part_stride; float xscale;   // x[i] = xscale * sum_p xin[(n*nparts+p)*part_stride+i]
    const float* xbias; int xact;           // then x[i] = act(x[i] + xbias[i]) when xbias != null (SE fc1 tail)
    const void* w;                          // T [O][I]
    const float* bias;                      // [O] or null
    void* y; int y_is_T;                    // [N][O] float or T
    int I, O, act;
};

// ---- 3x3 implicit-GEMM conv (decoder, ConvGRU) ----------------------------------------------
struct ConvSrc {
    const void* p; int C; int Cs; int up; int Hs; int Ws;
    int f16;                                // fp16 tensor feeding an fp32 kernel
    int w0, nw;                             // conv_kxk: the waves [w0, w0 + nw) of a block stage this source (set by the launcher)
};
// 16-bit state tensors at the API (fp16 / bf16 module, fp32 arithmetic inside).  The context keeps the fp32 value of
// every state it hands out (`shadow`); at the start of a frame each element of the incoming state is replaced by the
// shadow value when that rounds to exactly what the caller passed (the caller handed our state straight back, the
// normal recurrence) and taken as is otherwise (foreign / edited state).  Without this the state is re-rounded to
// 16 bits every frame and stalls wherever the GRU's update is below half an ulp: a slow drift against the fp32
// reference (measured: r2 4e-3 off after 140 HD frames).
struct StatePrep {
    const void* in[4]; const float* shadow[4]; float* out[4];   // per level: caller's state or null (zeros), kept fp32 state, fp32 state of this frame
    int n[4];                                                   // elements per level (multiples of 4)
    int blocks;                                                 // blocks of 1024 elements over the four levels; 0: none
    int bf;                                                     // 16-bit type: 0 fp16, 1 bf16
};
enum ConvEpi { EPI_RELU = 0, EPI_GRU_IH = 1, EPI_GRU_HH = 2 };
struct ConvArgs {
    ConvSrc src[4]; int nsrc; int Cpad;
    const void* w; int KS; int NT; const float* bias;
    int H, W, tiles_x, epi;                 // H, W: OUTPUT size
    int Hi, Wi;                             // input size of the direct (non-upsampled) sources
    int CC, nchunks;                        // channel chunk staged per pass (multiple of 8), number of passes
    int SK, SKlog2;                         // K steps per weight stage in LDS (2, 4 or 8)
    int wres;                               // conv_kxk: all KS x NW weight fragments resident in LDS (no stages)
    int cs, cd;                             // conv_sk only: stride and dilation (0 = 1)
    void* y; int Cout; int ys;              // EPI_RELU
    const void* hprev; int hC;              // GRU: previous state [N][H][W][hC] or null
    void* rh; float* z;                     // GRU scratch [N][H][W][hC]
    void* hout;                             // GRU new state [N][H][W][hC]
    void* hout2;                            // optional second copy of it (the caller's r_out)
    int state16;                            // hprev / hout are fp16 although the kernel computes in fp32
    int out2_mode;                          // IoMode of hout2 (IO_SAME: like hout)
    float* part;                            // conv_kxk split-K: partial sums [nchunks][N][H][W][NT*16] (EPI_RELU only); null: one block runs all chunks
    int dbg;                                // ablation switches for timing experiments (RVM_DBG env), 0 in production
};

// ---- decoder cat-convs with the weights in registers (conv_cat.hip); variant = tile * 3 + wave layout (see there)
constexpr int CCAT_NVAR = 12;
bool conv_cat_supported(const ConvArgs& a, int variant, int es, size_t* lds_out);
void conv_cat_shape(const ConvArgs& a, int variant, int* th, int* tw, int* ntb, int* ksp, int* mtw);
template <typename T> void launch_conv_cat(const ConvArgs& a, int N, int variant, hipStream_t s);
void conv_cat_init();

// ---- whole ConvGRU step in one launch (gru_fused.hip): gate conv, r * h, candidate conv, state blend -----------
struct GruArgs {
    const void* x; int xs;                  // the GRU's input: second half of the block's features [N][H*W][xs] (pointer at the half)
    const void* h; int hmode;               // previous state [N][H*W][C] (0 fp32 / 1 fp16 / 2 bf16 elements) or null (zeros)
    const void* wih; const float* bih;      // gate conv fragments [2 * ceil(C/16)][KS][512], tiles (r, z) interleaved; bias
    const void* whh; const float* bhh;      // candidate conv fragments [ceil(C/16)][KS][512]; bias
    void* hout; int omode;                  // new state [N][H*W][C]
    void* hout2; int omode2;                // optional second copy (the caller's r_out in the 16-bit I/O modes)
    int C, H, W, TH, TW, tiles_x, tiles_y;  // tiles of TH x TW pixels: 4 x 8, 8 x 8, 4 x 16 or 8 x 16
};
bool gru_fused_supported(int C, int es);
size_t gru_fused_lds(int C, int TH, int TW, int es);
template <typename T> void launch_gru_fused(const GruArgs& a, int N, hipStream_t s);
template <typename T> int gru_fused_occupancy(int C, int TH, int TW);     // blocks per CU (needs the device)
void gru_fused_init();

// ---- resize / pyramid (+ encoder stem) ----------------------------------------------------------------------
struct DownArgs {
    const void* src; void* s0; void* s1; void* s2; void* s3;
    int H, W, h, w, h1, w1, h2, w2, h3, w3; float scale_y, scale_x;
    // fused encoder stem (mobilenetv3): normalize + conv3x3 s2 + BN + hardswish of the s0 block just formed; null: none
    void* stem_y; const float* stem_w /*[27][16]*/; const float* stem_b /*[16]*/;
    int Ho, Wo; float mean[3]; float std_[3];
    int nblk;                               // blocks (x) of the downsample grid proper; the StatePrep blocks follow
    StatePrep sp;
};

struct MaxPoolArgs { const void* x; void* y; int C, Hi, Wi, Ho, Wo; };

// ---- refiner ------------------------------------------------------------------------------
struct DgfWeights {                          // device blob layout (floats)
    float proj_w[4][16]; float proj_b[4];
    float box[4][9];
    float w1[16][24]; float b1[16];
    float w2[16][16]; float b2[16];
    float w3[4][16];  float b3[4];
    float seg_w[16];  float seg_b;
};
struct DgfLowArgs {
    void* hid;                              // [h][w][16]: read (dgf_lowres); dgf_lowres_mfma computes it and writes it here unless null
    const void* s0; const DgfWeights* wt; float* ab; int h, w, tiles_x, fast;
    const void* mlp;                        // deep refiner: 9 MFMA weight fragments (hi/lo fp16), see pack in engine.cpp
    // dgf_lowres_mfma: the OutputBlock's second conv runs inside it
    const void* hin;                        // first conv's output [h][w][16]
    const void* cw; const float* cbias;     // conv3x3 16 -> 16 weight fragments [5][512] and bias (null: dgf_lowres on a given hid)
    int csplit;                             // fp32 tensors: conv on the split-fp16 MFMA (else fp32 MFMA)
};
struct DgfApplyArgs {
    const void* src; const float* ab; void* fgr; void* pha;
    int H, W, h, w; float scale_y, scale_x;
    int ncols;                              // low-res columns one 2048-px row segment can touch (LDS tile)
    int kx, rows;                           // dgf_apply_pick: 4 / 8 = the power-of-two kernel (W = kx w) with `rows` rows per block; 0 = general
    // fused driver output (rvm_forward_u8): composed uint8 [N][H][W][C] instead of fgr / pha; null: fgr / pha
    unsigned char* u8; int u8mode; float bgr[3];
};
struct ProjOutArgs { const void* hid; const void* src; const DgfWeights* wt; void* fgr; void* pha; int H, W, seg; };

// ---- output composition (driver step after forward) --------------------------------------------
enum ComposeMode { COMPOSE_GREEN = 0, COMPOSE_RGBA = 1, COMPOSE_ALPHA = 2, COMPOSE_FOREGROUND = 3 };
struct ComposeArgs { const void* fgr; const void* pha; unsigned char* out; int H, W, mode; float bgr[3]; };
void launch_compose(const ComposeArgs& a, int N, int io, hipStream_t s);     // io: IO_SAME = fp32 tensors

// ---- matting quality metrics (reference evaluation/evaluate_hr.py:176-235) ------------------------
constexpr int METRIC_TILE = 32;             // output tile edge of matte_metrics
constexpr int METRIC_NSUM = 6;              // |d|, d^2, grad^2, dt^2, masked fgr d^2, masked element count
struct MetricArgs {
    const float* pred; const float* tru;            // [N][H][W] alpha in 0..1
    const float* pred_prev; const float* tru_prev;  // [H][W] frame in front of frame 0 (nullptr: none)
    const float* pred_fgr; const float* tru_fgr;    // [N][3][H][W] (nullptr: no foreground metric)
    float* partial;                                 // [N][blocks][METRIC_NSUM]
    double* out;                                    // [N][METRIC_NSUM]
    int H, W, tiles_x, tiles_y;
    float g[9], dg[9];                              // separable 9x9 Gaussian-derivative filter: K[i][j] = g[i] * dg[j]
};
void launch_metrics(const MetricArgs& a, int N, hipStream_t s);
constexpr size_t METRIC_SCRATCH_BYTES = 4u << 20;      // static device scratch of rvm_metrics_static (per GPU)
int metrics_scratch(float** out);                       // its address on the CURRENT device (0 ok, 1 bad device, 2 symbol lookup failed)

// launchers (kernels.hip)
template <typename T> void launch_pw(const PwArgs& a, int N, int cfg, hipStream_t s);
template <typename T> void launch_gemv(const GemvArgs& a, int N, hipStream_t s);
template <typename T> void launch_conv3(const ConvArgs& a, int N, int MW, int NW, int geo, hipStream_t s);
template <typename T> void launch_maxpool(const MaxPoolArgs& a, int N, hipStream_t s);
template <typename T> void launch_down(const DownArgs& a, int N, int io, hipStream_t s);
void launch_state_prep(const StatePrep& sp, int io, hipStream_t s);   // io: IO_F16 / IO_BF16
void launch_flag_set(unsigned* flag, unsigned k, hipStream_t s);          // frame pipeline: counter-in-memory ordering between its streams
void launch_flag_wait(const unsigned* flag, unsigned k, hipStream_t s);
void launch_queue_probe(unsigned long long* out, unsigned ticks, hipStream_t s);   // out[0], out[1]: begin / end on the 100 MHz wall clock
template <typename T> void launch_dgf_low(const DgfLowArgs& a, int N, hipStream_t s);
template <typename T> void launch_dgf_apply(const DgfApplyArgs& a, int N, int io, hipStream_t s);
void dgf_apply_pick(DgfApplyArgs& a);                           // sets kx / rows from the geometry
void dgf_apply_init();
template <typename T> void launch_proj_out(const ProjOutArgs& a, int N, int io, hipStream_t s);
template <typename T> size_t conv_lds_bytes(int MW, int NW, int KD, int S, int D, int CC, int SK);
template <typename T> void launch_ed(const EdArgs& a, int N, int K, hipStream_t s);
size_t ed_lds_bytes(int K, int KS, int npxs, int es);
int stamps_read(unsigned long long* out, unsigned int* nlaunch);   // in-kernel timeline (RVM_STAMPS builds)
void kernels_init();   // one-time function attributes (dynamic LDS limits)
