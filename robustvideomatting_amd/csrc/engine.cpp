// Host side of librvm_mi355x.so: parameter store, BatchNorm folding, MFMA fragment packing,
// activation arena, the fixed per-frame kernel schedule and the C ABI (include/rvm_mi355x.h).
//
// The schedule mirrors MattingNetwork.forward (reference model/model.py:40-71):
//   downsample -> encoder -> LR-ASPP -> recurrent decoder -> projection -> refiner -> clamp
// but as 56 fused launches per frame (mobilenetv3) on NHWC buffers instead of ~110 ATen ops on NCHW.
#include "common.hpp"
#include "../../include/rvm_mi355x.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// Every ABI entry point that touches the device runs with the context's device current and puts the caller's
// device back on exit (a forward on a cuda:1 tensor must not change the calling thread's current device).
struct DeviceGuard {
    int prev = -1; hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DEVGUARD(dev) DeviceGuard guard_(dev); if (guard_.err != hipSuccess) return fail(std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int rup(int a, int b) { return cdiv(a, b) * b; }

// ---------------------------------------------------------------------------------------------
// architecture tables (reference model/mobilenetv3.py:9-27, model/model.py:23-33)
// ---------------------------------------------------------------------------------------------
struct MbRow { int cin, k, exp, cout; bool se; int act; int stride, dil; };
const MbRow kMbRows[15] = {
    {16, 3, 16, 16, false, ACT_RELU, 1, 1},   {16, 3, 64, 24, false, ACT_RELU, 2, 1},
    {24, 3, 72, 24, false, ACT_RELU, 1, 1},   {24, 5, 72, 40, true, ACT_RELU, 2, 1},
    {40, 5, 120, 40, true, ACT_RELU, 1, 1},   {40, 5, 120, 40, true, ACT_RELU, 1, 1},
    {40, 3, 240, 80, false, ACT_HSWISH, 2, 1}, {80, 3, 200, 80, false, ACT_HSWISH, 1, 1},
    {80, 3, 184, 80, false, ACT_HSWISH, 1, 1}, {80, 3, 184, 80, false, ACT_HSWISH, 1, 1},
    {80, 3, 480, 112, true, ACT_HSWISH, 1, 1}, {112, 3, 672, 112, true, ACT_HSWISH, 1, 1},
    {112, 5, 672, 160, true, ACT_HSWISH, 2, 2}, {160, 5, 960, 160, true, ACT_HSWISH, 1, 2},
    {160, 5, 960, 160, true, ACT_HSWISH, 1, 2},
};
int make_divisible(int v, int d) {
    int nv = std::max(d, (v + d / 2) / d * d);
    if (nv < 0.9 * v) nv += d;
    return nv;
}
int se_squeeze(int exp) { return make_divisible(exp / 4, 8); }

struct ParamSpec { std::string name; std::vector<int64_t> shape; };

void add_bn(std::vector<ParamSpec>& v, const std::string& p, int c) {
    v.push_back({p + ".weight", {c}});
    v.push_back({p + ".bias", {c}});
    v.push_back({p + ".running_mean", {c}});
    v.push_back({p + ".running_var", {c}});
    v.push_back({p + ".num_batches_tracked", {}});
}

std::vector<ParamSpec> param_specs(int variant, int refiner) {
    std::vector<ParamSpec> v;
    int enc_out, aspp, feat[4], dec[4];
    if (variant == RVM_VARIANT_MOBILENETV3) {
        enc_out = 960; aspp = 128;
        int f[4] = {16, 24, 40, 128}, d[4] = {80, 40, 32, 16};
        memcpy(feat, f, sizeof f); memcpy(dec, d, sizeof d);
        v.push_back({"backbone.features.0.0.weight", {16, 3, 3, 3}});
        add_bn(v, "backbone.features.0.1", 16);
        for (int i = 0; i < 15; ++i) {
            const MbRow& r = kMbRows[i];
            std::string p = "backbone.features." + std::to_string(i + 1) + ".block.";
            int j = 0;
            if (r.exp != r.cin) {
                v.push_back({p + std::to_string(j) + ".0.weight", {r.exp, r.cin, 1, 1}});
                add_bn(v, p + std::to_string(j) + ".1", r.exp); ++j;
            }
            v.push_back({p + std::to_string(j) + ".0.weight", {r.exp, 1, r.k, r.k}});
            add_bn(v, p + std::to_string(j) + ".1", r.exp); ++j;
            if (r.se) {
                int sq = se_squeeze(r.exp);
                v.push_back({p + std::to_string(j) + ".fc1.weight", {sq, r.exp, 1, 1}});
                v.push_back({p + std::to_string(j) + ".fc1.bias", {sq}});
                v.push_back({p + std::to_string(j) + ".fc2.weight", {r.exp, sq, 1, 1}});
                v.push_back({p + std::to_string(j) + ".fc2.bias", {r.exp}});
                ++j;
            }
            v.push_back({p + std::to_string(j) + ".0.weight", {r.cout, r.exp, 1, 1}});
            add_bn(v, p + std::to_string(j) + ".1", r.cout);
        }
        v.push_back({"backbone.features.16.0.weight", {960, 160, 1, 1}});
        add_bn(v, "backbone.features.16.1", 960);
    } else {
        enc_out = 2048; aspp = 256;
        int f[4] = {64, 256, 512, 256}, d[4] = {128, 64, 32, 16};
        memcpy(feat, f, sizeof f); memcpy(dec, d, sizeof d);
        v.push_back({"backbone.conv1.weight", {64, 3, 7, 7}});
        add_bn(v, "backbone.bn1", 64);
        const int nblk[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
        int inpl = 64;
        for (int l = 0; l < 4; ++l)
            for (int b = 0; b < nblk[l]; ++b) {
                std::string p = "backbone.layer" + std::to_string(l + 1) + "." + std::to_string(b);
                v.push_back({p + ".conv1.weight", {planes[l], inpl, 1, 1}});
                add_bn(v, p + ".bn1", planes[l]);
                v.push_back({p + ".conv2.weight", {planes[l], planes[l], 3, 3}});
                add_bn(v, p + ".bn2", planes[l]);
                v.push_back({p + ".conv3.weight", {planes[l] * 4, planes[l], 1, 1}});
                add_bn(v, p + ".bn3", planes[l] * 4);
                if (b == 0) {
                    v.push_back({p + ".downsample.0.weight", {planes[l] * 4, inpl, 1, 1}});
                    add_bn(v, p + ".downsample.1", planes[l] * 4);
                }
                inpl = planes[l] * 4;
            }
    }
    v.push_back({"aspp.aspp1.0.weight", {aspp, enc_out, 1, 1}});
    add_bn(v, "aspp.aspp1.1", aspp);
    v.push_back({"aspp.aspp2.1.weight", {aspp, enc_out, 1, 1}});
    auto gru = [&](const std::string& p, int c) {
        v.push_back({p + ".ih.0.weight", {2 * c, 2 * c, 3, 3}});
        v.push_back({p + ".ih.0.bias", {2 * c}});
        v.push_back({p + ".hh.0.weight", {c, 2 * c, 3, 3}});
        v.push_back({p + ".hh.0.bias", {c}});
    };
    gru("decoder.decode4.gru", feat[3] / 2);
    const char* un[3] = {"decode3", "decode2", "decode1"};
    int uin[3] = {feat[3], dec[0], dec[1]}, usk[3] = {feat[2], feat[1], feat[0]}, uout[3] = {dec[0], dec[1], dec[2]};
    for (int i = 0; i < 3; ++i) {
        std::string p = std::string("decoder.") + un[i];
        v.push_back({p + ".conv.0.weight", {uout[i], uin[i] + usk[i] + 3, 3, 3}});
        add_bn(v, p + ".conv.1", uout[i]);
        gru(p + ".gru", uout[i] / 2);
    }
    v.push_back({"decoder.decode0.conv.0.weight", {dec[3], dec[2] + 3, 3, 3}});
    add_bn(v, "decoder.decode0.conv.1", dec[3]);
    v.push_back({"decoder.decode0.conv.3.weight", {dec[3], dec[3], 3, 3}});
    add_bn(v, "decoder.decode0.conv.4", dec[3]);
    v.push_back({"project_mat.conv.weight", {4, 16, 1, 1}});
    v.push_back({"project_mat.conv.bias", {4}});
    v.push_back({"project_seg.conv.weight", {1, 16, 1, 1}});
    v.push_back({"project_seg.conv.bias", {1}});
    if (refiner == RVM_REFINER_DEEP_GUIDED_FILTER) {
        v.push_back({"refiner.box_filter.weight", {4, 1, 3, 3}});
        v.push_back({"refiner.conv.0.weight", {16, 24, 1, 1}});
        add_bn(v, "refiner.conv.1", 16);
        v.push_back({"refiner.conv.3.weight", {16, 16, 1, 1}});
        add_bn(v, "refiner.conv.4", 16);
        v.push_back({"refiner.conv.6.weight", {4, 16, 1, 1}});
        v.push_back({"refiner.conv.6.bias", {4}});
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// packed layers (offsets into the device weight blob)
// ---------------------------------------------------------------------------------------------
struct PwLayer { size_t w = 0, bias = 0; int Cin = 0, Cout = 0, KS = 0, NT = 0, act = 0; };
struct DwLayer { size_t w = 0, bias = 0; int C = 0, K = 3, stride = 1, dil = 1, pad = 1, act = 0; };
struct SeLayer { size_t w1f = 0, b1 = 0, w2 = 0, b2 = 0; int C = 0, S = 0; };   // w1f: fc1 weights [S][C] fp32
struct Conv3Layer { size_t w = 0, bias = 0; int Cpad = 0, KS = 0, NT = 0, Cout = 0, Cin = 0; int KD = 3, geo = 0, CC = 0, nchunks = 1; };
struct ResBlock { PwLayer c1, c3, ds; Conv3Layer c2; bool has_ds = false; int stride = 1, planes = 0, inpl = 0; };
struct GruLayer { Conv3Layer ih, hh; int C = 0; };
struct MbBlock {
    bool has_expand = false, has_se = false, res = false; PwLayer expand, project; DwLayer dw; SeLayer se; MbRow row;
    size_t pj_w = 0, pj_b = 0; bool pj = false;     // block 1: project weights as plain fp32 [Cin][Cout] + bias (run inside mbconv_ed)
};

// fill: the tap's buffer is not written by the frame's own launches (the kernel that produces the tensor keeps it on chip);
// rvm_debug_tap runs this first to compute it from what the last forward left in the arena
struct Tap { const void* p; int B, H, W, C, Cs; bool is_float; std::function<void(hipStream_t)> fill; };

struct FrameIO {
    const void* src; const void* r_in[4]; void* fgr; void* pha; void* r_out[4]; int64_t seg;
    unsigned char* u8; int64_t u8mode; float bgr[4];       // rvm_forward_u8: composed uint8 frames instead of fgr / pha
    bool operator==(const FrameIO& o) const { return memcmp(this, &o, sizeof(FrameIO)) == 0; }
};

}  // namespace

struct rvm_ctx {
    int variant, refiner, dtype, device;
    bool split = false;             // F16IO / BF16IO: GEMM operands as (hi, lo) fp16 pairs on the f16 MFMA (common.hpp f32s)
    int iomode() const { return dtype == RVM_DTYPE_F16IO ? IO_F16 : (dtype == RVM_DTYPE_BF16IO ? IO_BF16 : IO_SAME); }
    std::vector<ParamSpec> specs;
    std::map<std::string, std::vector<float>> params;
    bool finalized = false;

    // packed weights
    std::vector<char> blob;
    char* wdev = nullptr;
    size_t stem_w = 0, stem_b = 0;
    MbBlock mb[15];
    Conv3Layer res_stem;            // resnet50: conv1 7x7 s2
    std::vector<ResBlock> res;      // resnet50: 16 bottlenecks
    PwLayer last, aspp1;
    size_t aspp2_w = 0;
    int enc_out = 0, aspp_c = 0, feat[4], dec[4];
    GruLayer gru[4];            // decode4, decode3, decode2, decode1
    Conv3Layer up[3];           // decode3, decode2, decode1
    Conv3Layer up_ch[3];        // the same weights packed in 56-channel chunks where up[i] is not (KS == 0: none): tall tiles for batched frames
    Conv3Layer out0, out1;      // decode0 conv.0, conv.3
    size_t dgf = 0, dgf_mlp = 0;

    // plan
    bool planned = false;
    int B = 0, T = 1, H = 0, W = 0, h = 0, w = 0;    // B streams x T frames of each (T > 1: a chunk of a clip)
    double ds = 1;
    int lh[5], lw[5];
    char* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    struct Launch {
        std::string name; double flops, bytes;     // algorithmic work of the launch
        std::function<void(hipStream_t, const FrameIO&)> fn;
        int edge;                                  // 1 / 2: first / last launches, the only ones that touch src / fgr / pha
        int nk = 1;                                // kernels the launch enqueues (2: a split-K conv and its reduce pass)
    };
    std::vector<Launch> launches;
    std::map<std::string, Tap> taps;
    // plan-time tile selection by measurement (B*T > 1): launch key -> index into that launch's candidate list
    std::map<std::string, int> tuned;
    struct Tunable { std::string key; int ncand, model; bool always; };   // always: measured at plan time for single frames too
    std::vector<Tunable> tunable;                       // filled by build_plan: candidates and the model's choice
    float* split_ws = nullptr;                          // conv_kxk split-K partial sums, shared by the launches that use it
    size_t split_need = 0;

    // Frame pipeline (rvm_set_frame_pipeline).  The encoder phase of a frame (everything up to LR-ASPP) does not depend on
    // the recurrent state: it is cut into `enc_stages` stages of consecutive launches, each on a stream of the context's
    // own, and the decoder phase stays on the caller's stream - so stage j works on frame n + (enc_stages - j) while the
    // decoder works on frame n.  enc_stages + 1 frames are in flight, so the tensors the encoder stages hand on (and
    // their scratch) exist that many times in the arena - [set 0][set 1]...[decoder region] - and the plan is built once
    // per set: launches / taps are set 0's, launch_sets / tap_sets the others'.
    bool pipe_req = false;                              // what rvm_set_frame_pipeline asked for (plans built from now on)
    bool pipe = false;                                  // the active plan is (being) built for the pipeline
    int enc_stages = 1;                                 // of the active plan (RVM_PIPE_STAGES, default 3; 1..4)
    int stage_end[5] = {0, 0, 0, 0, 0};                 // encoder stage j = launches [stage_end[j - 1], stage_end[j])
    bool enc_phase = false;                             // build_plan: allocations and launches so far belong to the encoder phase
    size_t enc_shift = 0, enc_bytes = 0;                // build_plan: offset of the set being built; size of one set
    int n_enc = 0;                                      // launches [0, n_enc) are the encoder phase
    std::vector<std::vector<Launch>> launch_sets;
    std::vector<std::map<std::string, Tap>> tap_sets;
    float* split_ws_set[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // the encoder stages' split-K workspace, one per buffer set (frames in flight overlap)
    int frame_set = 0, last_set = 0;                    // set of the next frame / of the last one (debug taps)
    hipStream_t enc_streams[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipStream_t> probed_streams;            // caller's streams the stage streams were probed against (pick_stage_streams)
    int stream_picks = 0, stream_prio = 0;              // probing rounds so far; priority of the stage streams (the caller's)
    hipEvent_t ev_stage[5][5] = {};                     // [stage: encoder stages, then the decoder phase][set]
    bool ev_dec_valid[5] = {false, false, false, false, false};
    hipEvent_t src_event = nullptr;                     // one-shot: what the next frame's src waits for (null: src is complete)
    unsigned* sig_stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // RVM_PIPE_SYNC=kernel: frame counters per stage
    uint32_t pipe_seq = 0;                              // frames sent through the pipeline; frame k's stages publish k
    int pipe_sync = 0;                                  // 0: undecided, 2: events, 3: flag kernels
    bool pipe_adapt = true;                             // a frame that finds the device idle runs on the caller's stream (RVM_PIPE_ADAPT=0: never)
    hipStream_t last_dec_stream = nullptr;
    // An encoder stage behind the launch that reads src touches neither src nor the states nor the outputs: one HIP graph
    // per (buffer set, stage), captured at the set's first frame.  With two encoder stages the host is the bound (46
    // launches, ~280 us of enqueueing per HD frame for ~210 us of device time); a frame then costs it ~16 launches.
    hipGraphExec_t stage_exec[5][4] = {};               // [set][stage]
    bool stage_graph_off = false;                       // capture failed once / RVM_PIPE_GRAPH=0: eager launches

    // graph mode
    bool graph_mode = false;
    hipStream_t cap_stream = nullptr;
    struct GraphEntry { FrameIO key; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;

    // Plans built earlier and not in use: a driver that alternates shapes (convert_video's short last chunk, 4-D and
    // 5-D calls in turn) switches between them instead of re-allocating and re-measuring every time.  LRU, <= 4 parked.
    struct Parked {
        int B, T, H, W, h, w; double ds; int lh[5], lw[5];
        char* arena; size_t arena_bytes, arena_used;
        std::vector<Launch> launches; std::map<std::string, Tap> taps; std::map<std::string, int> tuned; std::vector<Tunable> tunable;
        float* split_ws; size_t split_need; std::vector<GraphEntry> graphs;
        bool pipe = false; size_t enc_bytes = 0; int n_enc = 0, enc_stages = 1, stage_end[5] = {0, 0, 0, 0, 0};
        std::vector<std::vector<Launch>> launch_sets; std::vector<std::map<std::string, Tap>> tap_sets;
        float* split_ws_set[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        hipGraphExec_t stage_exec[5][4] = {};
    };
    std::vector<Parked> parked;                       // back = most recently used
    void swap_plan(Parked& p) {
        std::swap(B, p.B); std::swap(T, p.T); std::swap(H, p.H); std::swap(W, p.W); std::swap(h, p.h); std::swap(w, p.w);
        std::swap(ds, p.ds);
        for (int l = 0; l < 5; ++l) { std::swap(lh[l], p.lh[l]); std::swap(lw[l], p.lw[l]); }
        std::swap(arena, p.arena); std::swap(arena_bytes, p.arena_bytes); std::swap(arena_used, p.arena_used);
        launches.swap(p.launches); taps.swap(p.taps); tuned.swap(p.tuned); tunable.swap(p.tunable);
        std::swap(split_ws, p.split_ws); std::swap(split_need, p.split_need); graphs.swap(p.graphs);
        std::swap(pipe, p.pipe); std::swap(enc_bytes, p.enc_bytes); std::swap(n_enc, p.n_enc);
        launch_sets.swap(p.launch_sets); tap_sets.swap(p.tap_sets);
        for (int j = 0; j < 5; ++j) std::swap(split_ws_set[j], p.split_ws_set[j]);
        std::swap(enc_stages, p.enc_stages);
        for (int j = 0; j < 5; ++j) std::swap(stage_end[j], p.stage_end[j]);
        for (int a = 0; a < 5; ++a) for (int b = 0; b < 4; ++b) std::swap(stage_exec[a][b], p.stage_exec[a][b]);
    }
    void free_parked() {
        for (auto& p : parked) {
            for (auto& g : p.graphs) (void)hipGraphExecDestroy(g.exec);
            for (auto& row : p.stage_exec) for (auto& e : row) if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
            if (p.arena) (void)hipFree(p.arena);
        }
        parked.clear();
    }

    size_t esize() const { return dtype == RVM_DTYPE_F16 ? 2 : 4; }          // internal element size
    const std::vector<float>* P(const std::string& n) const {
        auto it = params.find(n);
        return it == params.end() ? nullptr : &it->second;
    }
    size_t blob_alloc(size_t bytes) {
        size_t off = (blob.size() + 255) & ~size_t(255);
        blob.resize(off + bytes, 0);
        return off;
    }
    void* arena_alloc(size_t bytes) {
        size_t off = (arena_used + 255) & ~size_t(255);
        arena_used = off + bytes;
        return arena + off + (enc_phase ? enc_shift : 0);     // arena == nullptr during the sizing pass
    }
    void drop_graphs() {
        bool any = !graphs.empty();
        for (auto& row : stage_exec) for (auto& e : row) any |= e != nullptr;
        if (!any) return;
        (void)hipDeviceSynchronize();              // an exec may still be in flight on the caller's stream
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
        for (auto& row : stage_exec) for (auto& e : row) if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
    }
};

namespace {

// ---------------------------------------------------------------------------------------------
// folding / packing
// ---------------------------------------------------------------------------------------------
struct Affine { std::vector<float> scale, shift; };

bool bn_fold(const rvm_ctx* c, const std::string& p, int n, float eps, Affine& a) {
    auto g = c->P(p + ".weight"), b = c->P(p + ".bias"), m = c->P(p + ".running_mean"), v = c->P(p + ".running_var");
    if (!g || !b || !m || !v) { g_err = "missing BatchNorm parameters: " + p; return false; }
    a.scale.resize(n); a.shift.resize(n);
    for (int i = 0; i < n; ++i) {
        float s = (*g)[i] / std::sqrt((*v)[i] + eps);
        a.scale[i] = s;
        a.shift[i] = (*b)[i] - (*m)[i] * s;
    }
    return true;
}
Affine identity_affine(int n, const std::vector<float>* bias) {
    Affine a; a.scale.assign(n, 1.f); a.shift.assign(n, 0.f);
    if (bias) for (int i = 0; i < n; ++i) a.shift[i] = (*bias)[i];
    return a;
}

void put_elems(rvm_ctx* c, size_t off, const std::vector<float>& v) {
    if (c->dtype == RVM_DTYPE_F16) {
        f16* d = (f16*)(c->blob.data() + off);
        for (size_t i = 0; i < v.size(); ++i) d[i] = (f16)v[i];
    } else {
        memcpy(c->blob.data() + off, v.data(), v.size() * 4);
    }
}
size_t put_floats(rvm_ctx* c, const std::vector<float>& v) {
    size_t off = c->blob_alloc(v.size() * 4);
    memcpy(c->blob.data() + off, v.data(), v.size() * 4);
    return off;
}

// W[out][in][KK] -> MFMA fragments [NT][KS][64 lanes][8].  kmap[k] = {tap, packed channel} of K
// index k (tap < 0: zero column); out_map[po] / in_map[pc] give the original index or -1.
struct KEntry { int tap, pc; };
void pack_frags(rvm_ctx* c, const std::vector<float>& W, int Cin, int KK, const std::vector<int>& out_map,
                const std::vector<int>& in_map, const std::vector<KEntry>& kmap, const Affine& af,
                size_t& w_off, size_t& b_off, int& KS, int& NT) {
    KS = (int)kmap.size() / 32;
    NT = (int)out_map.size() / 16;
    const bool planar = c->dtype != RVM_DTYPE_F16 && !c->split;
    std::vector<float> fr((size_t)NT * KS * 512, 0.f), bias((size_t)NT * 16, 0.f);
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = out_map[nt * 16 + (lane & 15)];
                if (o < 0) continue;
                for (int j = 0; j < 8; ++j) {
                    const KEntry e = kmap[ks * 32 + 8 * (lane >> 4) + j];
                    if (e.tap < 0) continue;
                    const int i = in_map[e.pc];
                    if (i < 0) continue;
                    // fp16: lane-major [64][8]; fp32: two planes [2][64][4] (Frag<float>::loadw)
                    const size_t fo = planar ? (size_t)(j >> 2) * 256 + lane * 4 + (j & 3) : (size_t)lane * 8 + j;
                    fr[((size_t)nt * KS + ks) * 512 + fo] = W[((size_t)o * Cin + i) * KK + e.tap] * af.scale[o];
                }
            }
    for (size_t po = 0; po < out_map.size(); ++po)
        if (out_map[po] >= 0) bias[po] = af.shift[out_map[po]];
    w_off = c->blob_alloc(fr.size() * c->esize());
    if (c->split) {
        // per fragment: 512 hi halves (lane-major [64][8]) then 512 lo halves, lo = fp16(w - hi)
        f16* d = (f16*)(c->blob.data() + w_off);
        for (size_t fi = 0; fi < fr.size() / 512; ++fi)
            for (int e = 0; e < 512; ++e) {
                const float w = fr[fi * 512 + e];
                const f16 hi = (f16)w;
                d[fi * 1024 + e] = hi;
                d[fi * 1024 + 512 + e] = (f16)(w - (float)hi);
            }
    } else {
        put_elems(c, w_off, fr);
    }
    b_off = put_floats(c, bias);
}
// K order of conv_kxk: channel chunks of CC; inside a chunk k = tap * chunk_channels + channel; a
// chunk's step count is padded to a multiple of 8 when there are several chunks (kernels.hip).
std::vector<KEntry> conv_kmap(int KK, int Cpad, int CC, int nchunks) {
    std::vector<KEntry> m;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c_lo = ch * CC, CCc = std::min(CC, Cpad - c_lo);
        int KSc = cdiv(KK * CCc, 32);
        if (nchunks > 1) KSc = rup(KSc, 8);
        for (int k = 0; k < KSc * 32; ++k) {
            if (k < KK * CCc) m.push_back(KEntry{k / CCc, c_lo + k % CCc});
            else m.push_back(KEntry{-1, 0});
        }
    }
    return m;
}

std::vector<int> iota_map(int n, int padded) {
    std::vector<int> m(padded, -1);
    for (int i = 0; i < n; ++i) m[i] = i;
    return m;
}

bool pack_pw(rvm_ctx* c, const std::string& wname, const Affine& af, int Cin, int Cout, int act, PwLayer& L) {
    auto W = c->P(wname);
    if (!W) { g_err = "missing parameter " + wname; return false; }
    L.Cin = Cin; L.Cout = Cout; L.act = act;
    const int Cp = rup(Cin, 8);
    pack_frags(c, *W, Cin, 1, iota_map(Cout, rup(Cout, 16)), iota_map(Cin, Cp), conv_kmap(1, Cp, Cp, 1), af, L.w, L.bias, L.KS, L.NT);
    return true;
}

// srcC: channels of each concatenated source as the kernel sees them (multiples of 4);
// srcOrig: how many of them are real reference channels (the rest, e.g. the gray lane of the
// NHWC4 image sources, get zero weights).
size_t conv_lds(const rvm_ctx* c, int MW, int geo, int CC) {
    const int KD = geo == 3 ? 7 : 3, S = (geo == 1 || geo == 3) ? 2 : 1, D = geo == 2 ? 2 : 1;
    const int SK = 4;
    return c->dtype == RVM_DTYPE_F16 ? conv_lds_bytes<f16>(MW, 2, KD, S, D, CC, SK) : conv_lds_bytes<float>(MW, 2, KD, S, D, CC, SK);   // f32s == float
}
bool pack_conv3(rvm_ctx* c, const std::string& wname, const Affine& af, int Cout, const std::vector<int>& out_map,
                const std::vector<int>& srcC, const std::vector<int>& srcOrig, Conv3Layer& L, int geo = 0, bool splittable = false,
                bool force_chunks = false) {
    auto W = c->P(wname);
    if (!W) { g_err = "missing parameter " + wname; return false; }
    int tot = 0, cin = 0;
    for (int x : srcC) tot += x;
    for (int x : srcOrig) cin += x;
    std::vector<int> in_map(rup(tot, 8), -1);
    int pc = 0, oc = 0;
    for (size_t s = 0; s < srcC.size(); ++s) {
        for (int k = 0; k < srcOrig[s]; ++k) in_map[pc + k] = oc + k;
        pc += srcC[s]; oc += srcOrig[s];
    }
    L.Cpad = (int)in_map.size(); L.Cout = Cout; L.Cin = cin; L.geo = geo; L.KD = geo == 3 ? 7 : 3;
    // one pass if the whole-channel halo patch fits comfortably, else channel chunks of <= 64 KiB
    L.CC = L.Cpad; L.nchunks = 1;
    if (splittable && geo != 3 && (L.Cpad >= 224 || force_chunks)) {
        // long-K layers with a plain bias + ReLU epilogue (resnet50 bottleneck 3x3s, the decoder's cat convs): chunks of
        // 56 channels (9 * 56 / 32 = 15.75 -> 16 K steps, no padding to speak of) that conv_kxk's split mode runs side by
        // side on different CUs; 56 channels keep a 20-row tile's halo patch inside LDS
        L.CC = 56; L.nchunks = cdiv(L.Cpad, 56);
    } else if (conv_lds(c, 1, geo, L.Cpad) > 150 * 1024) {
        int cc = L.Cpad;
        while (cc > 8 && conv_lds(c, 1, geo, cc) > 64 * 1024) cc -= 8;
        L.CC = cc; L.nchunks = cdiv(L.Cpad, cc);
    }
    pack_frags(c, *W, cin, L.KD * L.KD, out_map, in_map, conv_kmap(L.KD * L.KD, L.Cpad, L.CC, L.nchunks), af,
               L.w, L.bias, L.KS, L.NT);
    return true;
}

bool pack_gru(rvm_ctx* c, const std::string& p, int C, GruLayer& G) {
    G.C = C;
    auto bih = c->P(p + ".ih.0.bias"), bhh = c->P(p + ".hh.0.bias");
    if (!bih || !bhh) { g_err = "missing parameter " + p + ".{ih,hh}.0.bias"; return false; }
    // ih: packed tile 2j = r[16j..16j+15], 2j+1 = z[16j..]
    const int ct = cdiv(C, 16);
    std::vector<int> om(ct * 32, -1);
    for (int j = 0; j < ct; ++j)
        for (int i = 0; i < 16; ++i) {
            int ch = j * 16 + i;
            if (ch < C) { om[(2 * j) * 16 + i] = ch; om[(2 * j + 1) * 16 + i] = C + ch; }
        }
    if (!pack_conv3(c, p + ".ih.0.weight", identity_affine(2 * C, bih), 2 * C, om, {C, C}, {C, C}, G.ih)) return false;
    if (!pack_conv3(c, p + ".hh.0.weight", identity_affine(C, bhh), C, iota_map(C, rup(C, 16)), {C, C}, {C, C}, G.hh)) return false;
    return true;
}

int finalize_impl(rvm_ctx* c) {
    for (auto& s : c->specs) {
        if (s.shape.empty()) continue;
        if (!c->P(s.name)) return fail("rvm_finalize: parameter not set: " + s.name);
    }
    c->blob.clear();
    Affine af;
    if (c->variant == RVM_VARIANT_MOBILENETV3) {
        c->enc_out = 960; c->aspp_c = 128;
        { int f[4] = {16, 24, 40, 128}, d[4] = {80, 40, 32, 16}; memcpy(c->feat, f, sizeof f); memcpy(c->dec, d, sizeof d); }
        // stem
        {
            auto W = c->P("backbone.features.0.0.weight");
            if (!bn_fold(c, "backbone.features.0.1", 16, 1e-3f, af)) return 1;
            std::vector<float> w(27 * 16);
            for (int o = 0; o < 16; ++o)
                for (int ci = 0; ci < 3; ++ci)
                    for (int t = 0; t < 9; ++t) w[(t * 3 + ci) * 16 + o] = (*W)[(o * 3 + ci) * 9 + t] * af.scale[o];
            c->stem_w = put_floats(c, w);
            c->stem_b = put_floats(c, af.shift);
        }
        for (int i = 0; i < 15; ++i) {
            MbBlock& b = c->mb[i];
            const MbRow& r = kMbRows[i];
            b.row = r; b.has_expand = r.exp != r.cin; b.has_se = r.se; b.res = (r.stride == 1 && r.cin == r.cout);
            std::string p = "backbone.features." + std::to_string(i + 1) + ".block.";
            int j = 0;
            if (b.has_expand) {
                if (!bn_fold(c, p + std::to_string(j) + ".1", r.exp, 1e-3f, af)) return 1;
                if (!pack_pw(c, p + std::to_string(j) + ".0.weight", af, r.cin, r.exp, r.act, b.expand)) return 1;
                ++j;
            }
            {
                auto W = c->P(p + std::to_string(j) + ".0.weight");
                if (!bn_fold(c, p + std::to_string(j) + ".1", r.exp, 1e-3f, af)) return 1;
                const int KK = r.k * r.k;
                std::vector<float> w((size_t)KK * r.exp);
                for (int ch = 0; ch < r.exp; ++ch)
                    for (int t = 0; t < KK; ++t) w[(size_t)t * r.exp + ch] = (*W)[(size_t)ch * KK + t] * af.scale[ch];
                b.dw.w = put_floats(c, w);
                b.dw.bias = put_floats(c, af.shift);
                b.dw.C = r.exp; b.dw.K = r.k; b.dw.dil = r.dil; b.dw.stride = r.dil > 1 ? 1 : r.stride;
                b.dw.pad = (r.k - 1) / 2 * r.dil; b.dw.act = r.act;
                ++j;
            }
            if (b.has_se) {
                const int S = se_squeeze(r.exp);
                std::string q = p + std::to_string(j);
                b.se.C = r.exp; b.se.S = S;
                b.se.w1f = put_floats(c, *c->P(q + ".fc1.weight"));
                b.se.b1 = put_floats(c, *c->P(q + ".fc1.bias"));
                b.se.w2 = c->blob_alloc((size_t)S * r.exp * c->esize()); put_elems(c, b.se.w2, *c->P(q + ".fc2.weight"));
                b.se.b2 = put_floats(c, *c->P(q + ".fc2.bias"));
                ++j;
            }
            if (!bn_fold(c, p + std::to_string(j) + ".1", r.cout, 1e-3f, af)) return 1;
            if (!pack_pw(c, p + std::to_string(j) + ".0.weight", af, r.exp, r.cout, ACT_NONE, b.project)) return 1;
            b.pj = !b.has_expand && !b.has_se && b.res && r.exp == 16 && r.cout == 16 && r.k == 3 && r.dil == 1 && r.stride == 1;
            if (b.pj) {
                auto W = c->P(p + std::to_string(j) + ".0.weight");
                std::vector<float> w(256);
                for (int co = 0; co < 16; ++co)
                    for (int ci = 0; ci < 16; ++ci) w[(size_t)ci * 16 + co] = (*W)[(size_t)co * 16 + ci] * af.scale[co];
                b.pj_w = put_floats(c, w);
                b.pj_b = put_floats(c, af.shift);
            }
        }
        if (!bn_fold(c, "backbone.features.16.1", 960, 1e-3f, af)) return 1;
        if (!pack_pw(c, "backbone.features.16.0.weight", af, 160, 960, ACT_HSWISH, c->last)) return 1;
    } else {
        // ResNet50 (reference model/resnet.py:5-33; torchvision Bottleneck v1.5, stride on the 3x3,
        // replace_stride_with_dilation=[F,F,T]); BN eps 1e-5, no input normalisation.
        c->enc_out = 2048; c->aspp_c = 256;
        { int f[4] = {64, 256, 512, 256}, d[4] = {128, 64, 32, 16}; memcpy(c->feat, f, sizeof f); memcpy(c->dec, d, sizeof d); }
        if (!bn_fold(c, "backbone.bn1", 64, 1e-5f, af)) return 1;
        if (!pack_conv3(c, "backbone.conv1.weight", af, 64, iota_map(64, 64), {4}, {3}, c->res_stem, 3)) return 1;
        c->res.clear();
        const int nblk[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
        int inpl = 64, dilation = 1;
        for (int l = 0; l < 4; ++l) {
            int stride = l == 0 ? 1 : 2;
            const int prev_dil = dilation;
            if (l == 3) { dilation *= stride; stride = 1; }
            for (int b = 0; b < nblk[l]; ++b) {
                ResBlock R;
                R.planes = planes[l]; R.inpl = inpl; R.stride = b == 0 ? stride : 1; R.has_ds = b == 0;
                const int d = b == 0 ? prev_dil : dilation;
                const std::string p = "backbone.layer" + std::to_string(l + 1) + "." + std::to_string(b);
                if (!bn_fold(c, p + ".bn1", planes[l], 1e-5f, af)) return 1;
                if (!pack_pw(c, p + ".conv1.weight", af, inpl, planes[l], ACT_RELU, R.c1)) return 1;
                if (!bn_fold(c, p + ".bn2", planes[l], 1e-5f, af)) return 1;
                const int geo = R.stride == 2 ? 1 : (d == 2 ? 2 : 0);
                if (!pack_conv3(c, p + ".conv2.weight", af, planes[l], iota_map(planes[l], planes[l]), {planes[l]}, {planes[l]}, R.c2, geo, true)) return 1;
                if (!bn_fold(c, p + ".bn3", planes[l] * 4, 1e-5f, af)) return 1;
                if (!pack_pw(c, p + ".conv3.weight", af, planes[l], planes[l] * 4, ACT_RELU, R.c3)) return 1;
                if (R.has_ds) {
                    if (!bn_fold(c, p + ".downsample.1", planes[l] * 4, 1e-5f, af)) return 1;
                    if (!pack_pw(c, p + ".downsample.0.weight", af, inpl, planes[l] * 4, ACT_NONE, R.ds)) return 1;
                }
                c->res.push_back(R);
                inpl = planes[l] * 4;
            }
        }
    }
    // LR-ASPP
    if (!bn_fold(c, "aspp.aspp1.1", c->aspp_c, 1e-5f, af)) return 1;
    if (!pack_pw(c, "aspp.aspp1.0.weight", af, c->enc_out, c->aspp_c, ACT_RELU, c->aspp1)) return 1;
    c->aspp2_w = c->blob_alloc((size_t)c->aspp_c * c->enc_out * c->esize());
    put_elems(c, c->aspp2_w, *c->P("aspp.aspp2.1.weight"));
    // decoder
    const int* feat = c->feat; const int* dec = c->dec;
    if (!pack_gru(c, "decoder.decode4.gru", feat[3] / 2, c->gru[0])) return 1;
    const char* un[3] = {"decode3", "decode2", "decode1"};
    int uin[3] = {feat[3], dec[0], dec[1]}, usk[3] = {feat[2], feat[1], feat[0]}, uout[3] = {dec[0], dec[1], dec[2]};
    for (int i = 0; i < 3; ++i) {
        std::string p = std::string("decoder.") + un[i];
        if (!bn_fold(c, p + ".conv.1", uout[i], 1e-5f, af)) return 1;
        // sources: up(a) | up(h) | f | s(NHWC4)   == reference cat[x, f, s] with x = cat[a, h]
        if (!pack_conv3(c, p + ".conv.0.weight", af, uout[i], iota_map(uout[i], rup(uout[i], 16)),
                        {uin[i] / 2, uin[i] / 2, usk[i], 4}, {uin[i] / 2, uin[i] / 2, usk[i], 3}, c->up[i], 0, true)) return 1;
        c->up_ch[i] = Conv3Layer();
        if (c->up[i].nchunks == 1 && c->up[i].Cpad >= 96 &&
            !pack_conv3(c, p + ".conv.0.weight", af, uout[i], iota_map(uout[i], rup(uout[i], 16)),
                        {uin[i] / 2, uin[i] / 2, usk[i], 4}, {uin[i] / 2, uin[i] / 2, usk[i], 3}, c->up_ch[i], 0, true, true)) return 1;
        if (!pack_gru(c, p + ".gru", uout[i] / 2, c->gru[i + 1])) return 1;
    }
    if (!bn_fold(c, "decoder.decode0.conv.1", dec[3], 1e-5f, af)) return 1;
    if (!pack_conv3(c, "decoder.decode0.conv.0.weight", af, dec[3], iota_map(dec[3], 16),
                    {dec[2] / 2, dec[2] / 2, 4}, {dec[2] / 2, dec[2] / 2, 3}, c->out0)) return 1;
    if (!bn_fold(c, "decoder.decode0.conv.4", dec[3], 1e-5f, af)) return 1;
    if (!pack_conv3(c, "decoder.decode0.conv.3.weight", af, dec[3], iota_map(dec[3], 16), {dec[3]}, {dec[3]}, c->out1)) return 1;
    // projection + refiner constants
    {
        DgfWeights d; memset(&d, 0, sizeof d);
        auto pw = c->P("project_mat.conv.weight"), pb = c->P("project_mat.conv.bias");
        auto sw = c->P("project_seg.conv.weight"), sb = c->P("project_seg.conv.bias");
        memcpy(d.proj_w, pw->data(), 64 * 4); memcpy(d.proj_b, pb->data(), 16);
        memcpy(d.seg_w, sw->data(), 64); d.seg_b = (*sb)[0];
        if (c->refiner == RVM_REFINER_DEEP_GUIDED_FILTER) {
            memcpy(d.box, c->P("refiner.box_filter.weight")->data(), 36 * 4);
            Affine a1, a2;
            if (!bn_fold(c, "refiner.conv.1", 16, 1e-5f, a1) || !bn_fold(c, "refiner.conv.4", 16, 1e-5f, a2)) return 1;
            auto w1 = c->P("refiner.conv.0.weight"), w2 = c->P("refiner.conv.3.weight"), w3 = c->P("refiner.conv.6.weight");
            for (int o = 0; o < 16; ++o) {
                for (int k = 0; k < 24; ++k) d.w1[o][k] = (*w1)[o * 24 + k] * a1.scale[o];
                for (int k = 0; k < 16; ++k) d.w2[o][k] = (*w2)[o * 16 + k] * a2.scale[o];
                d.b1[o] = a1.shift[o]; d.b2[o] = a2.shift[o];
            }
            memcpy(d.w3, w3->data(), 64 * 4);
            memcpy(d.b3, c->P("refiner.conv.6.bias")->data(), 16);
        } else {
            const float t = 1.0f / 3.0f;
            for (int ch = 0; ch < 4; ++ch) for (int k = 0; k < 9; ++k) d.box[ch][k] = t * t;
        }
        c->dgf = c->blob_alloc(sizeof d);
        memcpy(c->blob.data() + c->dgf, &d, sizeof d);
        c->dgf_mlp = 0;
        if (c->refiner == RVM_REFINER_DEEP_GUIDED_FILTER) {
            // MFMA fragments of the refiner MLP for dgf_lowres_mfma (kernels.hip): 9 fragments of 512 (hi) + 512 (lo)
            // halves; lane (row r = lane & 15, group g = lane >> 4), element j <-> K slot 8g + j.
            std::vector<float> fr(9 * 512, 0.f);
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4;
                for (int j = 0; j < 8; ++j) {
                    for (int t = 0; t < 4; ++t) {
                        const int rg = (g - t) & 3;                       // role of this lane group for tile t
                        if (rg < 3) fr[(size_t)t * 512 + lane * 8 + j] = d.w1[r][rg * 8 + j];      // [cov|var], hid[0..7], hid[8..15]
                        const int o = r - 4 * t;                          // last layer: real rows at 4t..4t+3
                        if (o >= 0 && o < 4 && j < 4) fr[(size_t)(5 + t) * 512 + lane * 8 + j] = d.w3[o][4 * g + j];
                    }
                    if (j < 4) fr[(size_t)4 * 512 + lane * 8 + j] = d.w2[r][4 * g + j];
                }
            }
            c->dgf_mlp = c->blob_alloc(9 * 1024 * sizeof(f16));
            f16* o16 = (f16*)(c->blob.data() + c->dgf_mlp);
            for (int f = 0; f < 9; ++f)
                for (int e = 0; e < 512; ++e) {
                    const float w = fr[(size_t)f * 512 + e];
                    const f16 hi = (f16)w;
                    o16[f * 1024 + e] = hi;
                    o16[f * 1024 + 512 + e] = (f16)(w - (float)hi);
                }
        }
    }
    if (c->wdev) { HIPCHK(hipFree(c->wdev)); c->wdev = nullptr; }
    HIPCHK(hipMalloc((void**)&c->wdev, c->blob.size()));
    HIPCHK(hipMemcpy(c->wdev, c->blob.data(), c->blob.size(), hipMemcpyHostToDevice));
    c->finalized = true;
    c->planned = false;
    c->launches.clear();
    c->drop_graphs();
    (void)hipDeviceSynchronize();
    c->free_parked();                                   // plans hold pointers into the old weight blob
    return 0;
}

// ---------------------------------------------------------------------------------------------
// schedule
// ---------------------------------------------------------------------------------------------
int pick_pw_cfg(int MT, int NT, int KS, int N) {
    // one n-tile (mobilenetv3 block 1): bandwidth-bound, most blocks win (config 1 runs 2 of its 4 waves)
    if (NT == 1 && KS == 1) return 1;
    // Long K on few tiles (the 1/16-scale project / ASPP GEMMs): split K over the block's 4 waves.
    if (KS >= 6) {
        const int sm[3] = {2, 1, 1}, sn[3] = {2, 2, 1};      // configs 4, 5, 6
        for (int k = 0; k < 3; ++k) {
            const int blocks = cdiv(MT, sm[k]) * cdiv(NT, sn[k]) * N;
            if (blocks >= 160 || k == 2) {
                // only when the un-split grid could not fill the chip with 4-wave blocks anyway
                if (cdiv(MT, 2) * cdiv(NT, 2) * N <= 384) return 4 + k;
                break;
            }
        }
    }
    // block tile (m-tiles x n-tiles) of pw_gemm configs 0..3, in order of preference
    const int bm[4] = {4, 2, 8, 2}, bn[4] = {4, 2, 1, 4};
    const int pref[4] = {0, 3, 2, 1};
    int best = 2, best_blocks = -1;
    for (int k = 0; k < 4; ++k) {
        const int cfg = pref[k];
        if (bn[cfg] > 1 && cdiv(NT, bn[cfg]) * bn[cfg] >= 2 * NT) continue;   // more than half the tile idle
        const int blocks = cdiv(MT, bm[cfg]) * cdiv(NT, bn[cfg]) * N;
        if (blocks >= 200) return cfg;
        if (blocks > best_blocks) { best_blocks = blocks; best = cfg; }
    }
    return best;
}

// GEMM launchers: the fp32-storage plan runs them on the split type when ctx->split is set
template <typename T> void launch_pw_m(const PwArgs& a, int N, int cfg, hipStream_t s, bool split) {
    if constexpr (std::is_same<T, float>::value) { if (split) { launch_pw<f32s>(a, N, cfg, s); return; } }
    launch_pw<T>(a, N, cfg, s);
}
template <typename T> void launch_conv3_m(const ConvArgs& a, int N, int MW, int NW, int geo, hipStream_t s, bool split) {
    if constexpr (std::is_same<T, float>::value) { if (split) { launch_conv3<f32s>(a, N, MW, NW, geo, s); return; } }
    launch_conv3<T>(a, N, MW, NW, geo, s);
}

// a launch repeated per time step ("decode3.gru[5].ih") shares one tuning entry ("decode3.gru.ih")
std::string tune_key(const std::string& name) {
    const size_t l = name.find('['), r = name.find(']');
    return (l == std::string::npos || r == std::string::npos) ? name : name.substr(0, l) + name.substr(r + 1);
}

template <typename T>
int build_plan(rvm_ctx* c) {
    c->launches.clear();
    c->taps.clear();
    c->tunable.clear();
    c->arena_used = 0;
    c->split_need = 0;
    c->enc_phase = c->pipe;
    c->n_enc = 0;
    auto tune_pick = [&](const std::string& name, int ncand, int model_choice, bool always = false) {
        const std::string key = tune_key(name);
        bool seen = false;
        for (auto& t : c->tunable) seen |= t.key == key;
        if (!seen) c->tunable.push_back({key, ncand, model_choice, always});
        auto it = c->tuned.find(key);
        return (it != c->tuned.end() && it->second >= 0 && it->second < ncand) ? it->second : model_choice;
    };
    // A chunk of T frames per stream (decoder.py:92-105,174-180): every stage outside the ConvGRUs runs on all
    // T*Bs frames at once (frame t of stream b is sample t*Bs + b); the GRUs step through t on Bs samples.
    const int Bs = c->B, TT = c->T;
    const int B = Bs * TT;
    const bool split = c->split;
    const size_t es = sizeof(T);
    const int io16 = c->iomode();                           // IO_F16 / IO_BF16: 16-bit tensors at the boundary, fp32 inside
    const size_t ies = c->dtype == RVM_DTYPE_F32 ? 4 : 2;   // element size of src / fgr / pha
    char* wb = c->wdev;
    const int* lh = c->lh; const int* lw = c->lw;
    auto abuf = [&](size_t elems) { return (T*)c->arena_alloc(elems * es); };
    auto fbuf = [&](size_t elems) { return (float*)c->arena_alloc(elems * 4); };
    auto add = [&](const std::string& name, double flops, double bytes, std::function<void(hipStream_t, const FrameIO&)> f, int edge = 0) {
        c->launches.push_back(rvm_ctx::Launch{name, flops, bytes, std::move(f), edge, 1});
    };

    // ---- recurrent state buffers.  16-bit I/O modes: the GRUs never read the caller's 16-bit tensors; the first launch
    // turns them into fp32 (StatePrep, common.hpp: the fp32 value kept from the previous call where it rounds to what
    // the caller passed) and every GRU output stays fp32 in the arena (hseq; the caller gets a rounded copy).  The kept
    // value is simply the last time step's slot of hseq.
    const int rch[4] = {c->dec[2] / 2, c->dec[1] / 2, c->dec[0] / 2, c->feat[3] / 2};     // channels of r1..r4
    float* hin32[4] = {nullptr, nullptr, nullptr, nullptr};
    float* hseq32[4] = {nullptr, nullptr, nullptr, nullptr};
    StatePrep sprep; memset(&sprep, 0, sizeof sprep);
    auto state_buffers = [&]() {
        for (int r = 0; r < 4; ++r) {
            const size_t per = (size_t)Bs * lh[r + 1] * lw[r + 1] * rch[r];
            hin32[r] = fbuf(per);
            hseq32[r] = fbuf(per * TT);
            sprep.shadow[r] = hseq32[r] + per * (TT - 1);
            sprep.out[r] = hin32[r];
            sprep.n[r] = (int)per;
            sprep.blocks += (int)((per + 4095) >> 12);
        }
        sprep.bf = io16 == IO_BF16;
    };
    if (io16 && !c->pipe) state_buffers();                  // (frame pipeline: with the decoder phase, below)

    // ---- downsample + pyramid (model.py:52-55, decoder.py:35-39)
    T* s[4];
    for (int l = 0; l < 4; ++l) s[l] = abuf((size_t)B * lh[l] * lw[l] * 4);
    T* stem_out = nullptr;                                   // mobilenetv3: the encoder stem runs inside the downsample launch
    {
        DownArgs a; memset(&a, 0, sizeof a);
        a.s0 = s[0]; a.s1 = s[1]; a.s2 = s[2]; a.s3 = s[3];
        a.H = c->H; a.W = c->W; a.h = c->h; a.w = c->w; a.h1 = lh[1]; a.w1 = lw[1]; a.h2 = lh[2]; a.w2 = lw[2]; a.h3 = lh[3]; a.w3 = lw[3];
        a.scale_y = a.scale_x = (c->ds == 1.0) ? 1.f : (float)(1.0 / c->ds);
        // algorithmic bytes: the 2x2 source taps each s0 pixel needs (all of src when ds == 1) + the s0..s3 writes
        const double taps = (c->ds == 1.0) ? 1.0 : 4.0;
        const double pyr = (double)(lh[1] * lw[1] + lh[2] * lw[2] + lh[3] * lw[3]);
        double sflops = 0, sbytes = 0;
        if (c->variant == RVM_VARIANT_MOBILENETV3) {
            stem_out = abuf((size_t)B * lh[1] * lw[1] * 16);
            a.stem_y = stem_out; a.stem_w = (const float*)(wb + c->stem_w); a.stem_b = (const float*)(wb + c->stem_b);
            a.Ho = lh[1]; a.Wo = lw[1];
            const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
            for (int k = 0; k < 3; ++k) { a.mean[k] = mean[k]; a.std_[k] = sd[k]; }
            sflops = 2.0 * B * a.Ho * a.Wo * 27 * 16; sbytes = (double)B * a.Ho * a.Wo * 16 * es;
        }
        add(stem_out ? "downsample_pyramid_stem" : "downsample_pyramid", 8.0 * B * c->h * c->w * 3 + 8.0 * B * pyr + sflops,
            (double)B * (c->h * c->w * (3 * taps * ies + 4 * es) + pyr * 4 * es) + sbytes,
            [=](hipStream_t st, const FrameIO& io) {
                DownArgs b = a; b.src = io.src; b.sp = sprep;
                for (int r = 0; r < 4; ++r) b.sp.in[r] = io.r_in[r];
                launch_down<T>(b, B, io16, st);
            }, 1);
        for (int l = 0; l < 4; ++l) c->taps["s" + std::to_string(l)] = Tap{s[l], B, lh[l], lw[l], 4, 4, false};
    }

    bool plan_bad = false;
    struct PwX {
        int act_post = 0, sstride = 1, sHi = 0, sWi = 0, sWo = 0;
        // squeeze-excite tail inside the GEMM's prologue (see PwArgs)
        const float* se_hpart = nullptr; int se_nparts = 0, se_S = 0; float se_xscale = 0.f;
        const float* se_b1 = nullptr; const void* se_w2 = nullptr; const float* se_b2 = nullptr;
    };
    auto pw = [&](const std::string& name, const PwLayer& L, const T* x, int xs, int HW, T* y, int ys, const T* in_scale, const T* res, int rs,
                  const float* out_scale, float* pool, PwX ex = PwX()) {
        PwArgs a; memset(&a, 0, sizeof a);
        a.act_post = ex.act_post; a.sstride = ex.sstride; a.sHi = ex.sHi; a.sWi = ex.sWi; a.sWo = ex.sWo;
        a.se_hpart = ex.se_hpart; a.se_nparts = ex.se_nparts; a.se_S = ex.se_S; a.se_xscale = ex.se_xscale; a.se_b1 = ex.se_b1; a.se_w2 = ex.se_w2; a.se_b2 = ex.se_b2;
        a.x = x; a.Cin = L.Cin; a.xs = xs; a.w = wb + L.w; a.KS = L.KS; a.bias = (const float*)(wb + L.bias);
        a.in_scale = in_scale; a.res = res; a.rs = rs; a.out_scale = out_scale; a.oss = L.Cout;
        a.y = y; a.Cout = L.Cout; a.ys = ys; a.pool = pool; a.HW = HW; a.MT = cdiv(HW, 16); a.NT = L.NT; a.act = L.act;
        if (in_scale && (L.KS * 32 > 1024 || L.Cin % 4)) { fail("pw_gemm: input scale vector too long"); plan_bad = true; return; }
        if (ex.se_hpart && (in_scale || L.Cin > 256 || ex.se_S > 32 || ex.se_S % 4 || ex.se_nparts > 256)) { fail("pw_gemm: squeeze-excite tail too large for the prologue"); plan_bad = true; return; }
        // candidates: pw_gemm configurations 0..6; for layers without a squeeze-excite input scale also the LDS-tiled
        // pipelined form with 64-pixel block tiles (7) and, where the epilogue is plain, its split-K variants in 2 / 4 / 8
        // slices (8, 9, 10), then the same with 128-pixel tiles (11; split 2 / 4: 12, 13).  The latency model only knows
        // 0..6; the others are taken where plan-time measurement finds them faster.
        const bool lds_ok = !in_scale && !ex.se_hpart;
        // (split-K only where the unsplit grid leaves CUs idle: <= 512 block tiles; the sizing pass then reserves the
        //  partial-sum workspace of the widest split on offer, so that plan-time measurement can try every candidate)
        const bool split_ok = lds_ok && !pool && !out_scale && (int64_t)B * 8 <= 65535 && (int64_t)cdiv(a.MT, 4) * cdiv(a.NT, 4) * B <= 512;
        const int ncand = lds_ok ? (split_ok ? 14 : 8) : 7;
        if (split_ok) {
            int widest = 0;
            for (int ns = 8; ns >= 2; ns >>= 1) if (a.KS >= 2 * ns) { widest = ns; break; }
            if (widest) c->split_need = std::max(c->split_need, (size_t)widest * B * HW * L.NT * 16 * sizeof(float));
        }
        int cfg = tune_pick(name, ncand, pick_pw_cfg(a.MT, a.NT, a.KS, B));
        if (const char* e = getenv("RVM_PW_CFG")) {               // test hook: force one configuration wherever it is a candidate
            const int v = atoi(e);
            if (v >= 0 && v < ncand) cfg = v;
        }
        if (cfg >= 8 && cfg != 11) {
            a.nsplit = cfg >= 12 ? 1 << (cfg - 11) : 1 << (cfg - 7);
            if (a.KS < 2 * a.nsplit) { cfg = cfg >= 12 ? 11 : 7; a.nsplit = 0; }      // too few K steps to slice: the unsplit form
            else {
                a.part = (float*)1;                                 // marker: the launch takes c->split_ws
                c->split_need = std::max(c->split_need, (size_t)a.nsplit * B * HW * L.NT * 16 * sizeof(float));
            }
        }
        const bool in_enc = c->enc_phase;                   // (frame pipeline: every buffer set has a split-K workspace of its own)
        const int ws_set = c->enc_bytes ? (int)(c->enc_shift / c->enc_bytes) : 0;
        add(name, 2.0 * B * HW * L.Cin * L.Cout,
            ((double)B * HW * (L.Cin + L.Cout + (res ? L.Cout : 0)) + (double)L.Cin * L.Cout) * es,
            [=](hipStream_t st, const FrameIO&) {
                PwArgs b = a;
                if (a.part) b.part = in_enc ? c->split_ws_set[ws_set] : c->split_ws;
                launch_pw_m<T>(b, B, cfg, st, split);
            });
        if (a.part) c->launches.back().nk = 2;
    };

    // KxK conv launcher: (Ho, Wo) output, (Hi, Wi) input of the direct sources
    // (`keep`: hand the launch back instead of adding it to the frame - the caller schedules it itself)
    auto conv3 = [&](const std::string& name, const Conv3Layer& L0, int Ho, int Wo, int Hi, int Wi, int epi,
                     std::function<void(ConvArgs&, const FrameIO&)> fill, int nb = 0,
                     std::function<void(hipStream_t, const FrameIO&)>* keep = nullptr, const Conv3Layer* alt = nullptr) -> int {
        // `alt`: the same layer packed in 56-channel chunks.  Its halo patch holds one chunk at a time, so 20-row tiles
        // fit LDS where the unchunked patch allows 4 or 8 rows: with many frames per launch the weight stream per output
        // pixel is what the launch costs.  Offered as extra candidates when several frames are planned (measured then).
        const bool alt_ok = alt && alt->KS > 0 && alt->nchunks > 1 && alt->CC == 56 && (nb ? nb : Bs * TT) > 1 && L0.geo == 0 && L0.nchunks == 1;
        int alt_first = -1, alt_force = -1;                 // index of the first alt candidate among the valid ones
        static const int alt_cand[4][2] = {{5, 2}, {5, 1}, {2, 2}, {5, 4}};
        constexpr int NALT = 4;
        const Conv3Layer* Lp = &L0;
        {
            // the candidate list is rebuilt the same way on every build_plan pass, so the tuned index decides here already
            // which packing the rest of this function sees
            if (alt_ok) {
                const int cand[6][2] = {{4, 1}, {2, 2}, {2, 1}, {1, 2}, {1, 1}, {0, epi == EPI_GRU_IH ? 2 : 1}};
                int nvalid = 0;
                for (int k = 0; k < 5; ++k) {                  // (conv_sk never applies: these layers have upsampled sources)
                    if (epi == EPI_GRU_IH && cand[k][1] != 2) continue;
                    if (conv_lds_bytes<T>(cand[k][0], cand[k][1], 3, 1, 1, L0.CC, 4) > 150 * 1024) continue;
                    ++nvalid;
                }
                alt_first = nvalid;
                auto it = c->tuned.find(tune_key(name));
                if (it != c->tuned.end() && it->second >= alt_first && it->second < alt_first + NALT) Lp = alt;
                if (const char* e = getenv("RVM_CONV_ALT")) {           // test hook: force chunked candidate 0..3
                    alt_force = atoi(e);
                    if (alt_force >= 0 && alt_force < NALT) Lp = alt; else alt_force = -1;
                }
            }
        }
        const Conv3Layer& L = *Lp;
        const int B = nb ? nb : Bs * TT;                    // samples of this launch (a GRU step: Bs)
        // conv_cat (weights in registers, K split over the waves): extra candidates after the others for the decoder's
        // conv + BN + ReLU layers whose K fits the wave slices
        int nreg = 0, reg_var[CCAT_NVAR];
        ConvArgs p0; memset(&p0, 0, sizeof p0);
        if (epi == EPI_RELU && L0.geo == 0 && L0.nchunks == 1 && !keep && !(getenv("RVM_CONV_REG") && atoi(getenv("RVM_CONV_REG")) == 0)) {
            p0.Cpad = L0.Cpad; p0.KS = L0.KS; p0.NT = L0.NT; p0.CC = L0.CC; p0.nchunks = 1; p0.epi = epi; p0.Cout = L0.Cout;
            FrameIO none; memset(&none, 0, sizeof none);
            fill(p0, none);
            for (int v = 0; v < CCAT_NVAR; ++v) if (conv_cat_supported(p0, v, (int)es, nullptr)) reg_var[nreg++] = v;
        }
        ConvArgs a; memset(&a, 0, sizeof a);
        a.Cpad = L.Cpad; a.w = wb + L.w; a.KS = L.KS; a.NT = L.NT; a.bias = (const float*)(wb + L.bias);
        a.H = Ho; a.W = Wo; a.Hi = Hi; a.Wi = Wi; a.tiles_x = cdiv(Wo, 16); a.epi = epi; a.Cout = L.Cout;
        a.CC = L.CC; a.nchunks = L.nchunks;
        { const char* e = getenv("RVM_DBG"); a.dbg = e ? atoi(e) : 0; }
        int SK0 = 4;                                        // minimum weight stage depth: one 16-byte piece per thread at least
        // tile shape: biggest rows-per-block that still yields >= 256 blocks and fits LDS
        int NW = (L.NT >= 2 || epi == EPI_GRU_IH) ? 2 : 1;
        int MW = 1;
        const int geo = L.geo;
        if (geo == 0 && L.nchunks == 1) {
            // pick the (MW, NW) tile that minimises blocks-per-CU x time-per-block (ties: the larger tile, less halo)
            // candidate 5 = conv_sk, the split-K direct form for small maps (MW = 0): direct (not upsampled) sources only
            const int cand[6][2] = {{4, 1}, {2, 2}, {2, 1}, {1, 2}, {1, 1}, {0, epi == EPI_GRU_IH ? 2 : 1}};
            bool sk_ok = true;
            {
                ConvArgs probe = a; FrameIO none; memset(&none, 0, sizeof none);
                fill(probe, none);
                if (probe.nsrc < 1 || probe.nsrc > 2) sk_ok = false;
                for (int k = 0; k < probe.nsrc && k < 2; ++k) if (probe.src[k].up || probe.src[k].C % 4) sk_ok = false;
            }
            int valid[6], nvalid = 0, model = -1;
            double best = 1e30;
            for (int k = 0; k < 6; ++k) {
                const int m = cand[k][0], nw = cand[k][1];
                double cost;
                if (m == 0) {
                    if (!sk_ok) continue;
                    // measured (17x30 and 34x60 ConvGRUs): about 3.2 us + 0.055 us per K step and n-tile of a block
                    // ... and never less than its weight stream (every block reads its n-tiles' whole K) at ~4.5 TB/s of L2
                    const int blocks = a.tiles_x * Ho * cdiv(L.NT, nw) * B;
                    cost = std::max((double)cdiv(blocks, 512) * (3.2 + 0.055 * L.KS * nw), (double)blocks * L.KS * nw * 512 * es / 4.5e6);
                } else {
                    if (epi == EPI_GRU_IH && nw != 2) continue;
                    const size_t lds = conv_lds_bytes<T>(m, nw, 3, 1, 1, L.CC, SK0);
                    if (lds > 150 * 1024) continue;
                    // measured on the HD frame (every candidate of the 13 decoder convs timed alone): a block takes about
                    // 2.7 us + 0.128 us per (K step x 16x16 output tile), and the blocks that share a CU take turns
                    // (they are bound by what one CU can pull from L2), so the launch costs that times the blocks per CU
                    const int blocks = a.tiles_x * cdiv(Ho, 4 * m) * cdiv(L.NT, nw) * B;
                    cost = (double)cdiv(blocks, 256) * (2.7 + 0.128 * L.KS * m * nw);
                }
                if (cost < best - 1e-9) { best = cost; model = nvalid; }
                valid[nvalid++] = k;
            }
            if (nvalid == 0) return fail("conv: input patch does not fit LDS");
            if (alt_ok && nvalid != alt_first) return fail("conv: candidate lists of the two packings disagree");
            const int reg_first = nvalid + (alt_ok ? NALT : 0);
            for (int r = 0; r < nreg; ++r) {
                // a block: launch + staging of its patch, then its waves' slice of K over their m-tiles (two waves per SIMD,
                // 16 cycles per MFMA at ~2.1 GHz), then the slices' sum
                int th, tw, ntb, ksp, mtw;
                conv_cat_shape(p0, reg_var[r], &th, &tw, &ntb, &ksp, &mtw);
                const int blocks = cdiv(Ho, th) * cdiv(Wo, tw) * cdiv(L.NT, ntb) * B;
                size_t ldsb = 0;
                conv_cat_supported(p0, reg_var[r], (int)es, &ldsb);
                const int per_cu = (ldsb > 80 * 1024 || ksp > 5 || th * tw == 256) ? 1 : 2, nper = cdiv(blocks, 256);
                // fitted to the in-kernel timelines of the four HD-frame layers: launch, staging (patch pixels x channels),
                // the wave's MFMAs (two waves per SIMD, three per step on the split type), the slices' sum and the store
                const int mt = th * tw / 16;
                const double blk = 1.6 + 2.5 + 1e-4 * (th + 2) * (tw + 2) * L.CC + 0.9 + 0.015 * (split ? 3 : 1) * ksp * mtw + 0.6 + 1.0 * cdiv(mt * ntb, 8);
                const double cost = blk * (per_cu == 2 && nper > 1 ? 0.75 * nper : nper) * 0.62;     // (0.62: on the scale of the conv_kxk model above, which reads low)
                if (cost < best - 1e-9) { best = cost; model = reg_first + r; }
            }
            int pidx = tune_pick(name, reg_first + nreg, model);
            if (const char* e = getenv("RVM_CONV_REG_VAR")) {        // test hook: force that conv_cat variant where it applies
                for (int r = 0; r < nreg; ++r) if (reg_var[r] == atoi(e)) pidx = reg_first + r;
            }
            if (pidx >= reg_first) {
                const int tile = reg_var[pidx - reg_first];
                if (getenv("RVM_VERBOSE") && c->arena) fprintf(stderr, "[rvm] %-16s conv_cat variant %d (tile %d, layout %d) KS %d NT %d\n", name.c_str(), tile, tile / 3, tile % 3, L.KS, L.NT);
                auto fn = [=](hipStream_t st, const FrameIO& io) {
                    ConvArgs b = a; fill(b, io);
                    if constexpr (std::is_same<T, float>::value) { if (split) { launch_conv_cat<f32s>(b, B, tile, st); return; } }
                    launch_conv_cat<T>(b, B, tile, st);
                };
                double px_bytes = 0;
                {
                    ConvArgs probe = a; FrameIO none; memset(&none, 0, sizeof none);
                    fill(probe, none);
                    for (int k = 0; k < probe.nsrc; ++k) {
                        const ConvSrc& S = probe.src[k];
                        px_bytes += (double)(S.up ? S.Hs * S.Ws : Hi * Wi) * S.C * (S.f16 ? 2 : es);
                    }
                    px_bytes += (double)Ho * Wo * L.Cout * es;
                }
                add(name, 2.0 * B * Ho * Wo * 9.0 * L.Cin * L.Cout, (double)B * px_bytes + 9.0 * L.Cin * L.Cout * es, fn);
                return 0;
            }
            const int pick = valid[pidx];      // (an alt index never reaches here: Lp == alt then)
            MW = cand[pick][0]; NW = cand[pick][1];
        } else if (Lp == alt && alt_ok) {
            // the chunked packing, one block per tile looping over the chunks
            int k = tune_pick(name, alt_first + NALT + nreg, 0) - alt_first;
            if (alt_force >= 0) k = alt_force;
            if (k < 0 || k >= NALT) k = 0;
            MW = alt_cand[k][0]; NW = alt_cand[k][1];
            if (NW > L.NT) NW = L.NT >= 2 ? 2 : 1;
        } else if (L.nchunks > 1 && L.KD == 3 && epi == EPI_RELU && L.CC == 56) {
            // Long-K layer packed in 56-channel chunks: conv_kxk's split mode (the chunks side by side on different CUs,
            // partial sums added by a second pass) in several tile shapes, the one-block-per-tile chunk loop, and conv_sk.
            // No latency model ranks these; the candidates are timed at plan time (always, a few ms per layer).
            const int cS0 = geo == 1 ? 2 : 1, cD0 = geo == 2 ? 2 : 1;
            struct Cand { int mw, nw, split; };               // split: 1 = chunks side by side (split-K), 0 = the one-block chunk loop
            std::vector<Cand> cands;
            if (geo != 1) { cands.push_back({5, 2, 1}); cands.push_back({5, 4, 1}); cands.push_back({5, 1, 1}); }
            if (geo == 0) { cands.push_back({2, 2, 1}); cands.push_back({2, 1, 1}); }
            cands.push_back({1, 2, 1});
            cands.push_back({1, 2, 0});
            bool sk_ok = true;
            {
                ConvArgs probe = a; FrameIO none; memset(&none, 0, sizeof none);
                fill(probe, none);
                if (probe.nsrc < 1 || probe.nsrc > 2) sk_ok = false;
                for (int k = 0; k < probe.nsrc && k < 2; ++k) if (probe.src[k].up || probe.src[k].C % 4) sk_ok = false;
            }
            if (sk_ok) cands.push_back({0, 1, 0});
            std::vector<Cand> valid;
            int model = 0;
            for (const Cand& k : cands) {
                if (k.mw && conv_lds_bytes<T>(k.mw, k.nw, 3, cS0, cD0, L.CC, k.nw == 4 ? 2 : 4) > 158 * 1024) continue;
                if (k.nw > 1 && L.NT < k.nw) continue;
                if (k.split > 1 && L.NT < k.nw * k.split) continue;
                if (k.split && (int64_t)B * L.nchunks > 65535) continue;
                valid.push_back(k);
            }
            if (valid.empty()) return fail("conv: input patch does not fit LDS");
            // model choice (RVM_AUTOTUNE=0): split with the tallest tile that is not mostly empty
            for (size_t k = 0; k < valid.size(); ++k)
                if (valid[k].split && (valid[k].mw == 5 ? Ho >= 12 : true)) { model = (int)k; break; }
            const Cand pick = valid[tune_pick(name, (int)valid.size(), model, true)];
            MW = pick.mw; NW = pick.nw;
            if (MW == 0) { a.cs = cS0; a.cd = cD0; }
            if (pick.split) {
                a.part = (float*)1;                                   // marker: the launch takes c->split_ws
                c->split_need = std::max(c->split_need, (size_t)L.nchunks * B * Ho * Wo * L.NT * 16 * sizeof(float));
            }
        } else {
            NW = 2;     // strided / dilated / 7x7 / chunked geometries are instantiated for MW=1, NW=2
            if (L.NT < 2 && epi != EPI_RELU) return fail("conv: unsupported tile shape");
            // 3x3 layers of this kind on small maps (resnet50 layer3 / layer4 at 1/16 scale) have conv_sk as the alternative
            if (L.KD == 3 && epi == EPI_RELU) {
                bool sk_ok = true;
                ConvArgs probe = a; FrameIO none; memset(&none, 0, sizeof none);
                fill(probe, none);
                if (probe.nsrc < 1 || probe.nsrc > 2) sk_ok = false;
                for (int k = 0; k < probe.nsrc && k < 2; ++k) if (probe.src[k].up || probe.src[k].C % 4) sk_ok = false;
                if (sk_ok) {
                    const int bk = a.tiles_x * cdiv(Ho, 4) * cdiv(L.NT, 2) * B, bs = a.tiles_x * Ho * L.NT * B;
                    const double ck = (double)cdiv(bk, 256) * (2.7 * L.nchunks + 0.128 * L.KS * 2);
                    const double cs_ = std::max((double)cdiv(bs, 512) * (3.2 + 0.055 * L.KS), (double)bs * L.KS * 512 * es / 4.5e6);
                    if (tune_pick(name, 2, cs_ < ck ? 1 : 0) == 1) { MW = 0; NW = 1; a.cs = (geo == 1) ? 2 : 1; a.cd = (geo == 2) ? 2 : 1; }
                }
            }
        }
        const int cS = (geo == 1 || geo == 3) ? 2 : 1, cD = geo == 2 ? 2 : 1;
        if (NW == 4) SK0 = 2;                                   // four n-tiles per stage step: two steps fill a 16-byte piece per thread
        if (conv_lds_bytes<T>(MW, NW, L.KD, cS, cD, L.CC, SK0) > 160 * 1024) return fail("conv: input patch does not fit LDS");
        // weight stage depth: each stage costs one global round trip, so use the deepest that keeps the
        // block count per CU the grid needs (one block per CU when the grid is small)
        if (SK0 * NW * 512 * (int)sizeof(T) < 16 * 512) SK0 = 8;                   // fp16, one n-tile
        if (conv_lds_bytes<T>(MW, NW, L.KD, cS, cD, L.CC, SK0) > 160 * 1024) return fail("conv: input patch does not fit LDS");
        a.SK = SK0;
        if (MW == 0) { a.SK = 8; a.SKlog2 = 3; }                // conv_sk: no weight stages
        else {
            const int blocks = a.tiles_x * cdiv(Ho, 4 * MW) * cdiv(L.NT, NW) * B * (a.part ? L.nchunks : 1);
            const size_t budget = blocks <= 256 ? 150 * 1024 : (blocks <= 512 ? 78 * 1024 : 52 * 1024);
            const char* e = getenv("RVM_CONV_SK");
            const int skmax = e ? atoi(e) : 8;
            for (int sk = SK0 * 2; sk <= skmax; sk *= 2)
                if (conv_lds_bytes<T>(MW, NW, L.KD, cS, cD, L.CC, sk) <= budget && (L.nchunks == 1 || sk <= 8)) a.SK = sk;
            a.SKlog2 = a.SK == 8 ? 3 : (a.SK == 4 ? 2 : 1);
            if (getenv("RVM_VERBOSE") && c->arena)
                fprintf(stderr, "[rvm] %-16s MW %d NW %d SK %d KS %d blocks %d lds %zu\n", name.c_str(), MW, NW, a.SK, L.KS, blocks,
                        conv_lds_bytes<T>(MW, NW, L.KD, cS, cD, L.CC, a.SK));
        }
        // short-K layers: all weight fragments resident in LDS (conv_kxk WRES) when that keeps the blocks per CU the staged
        // form has (two below 78 KB of LDS, one otherwise)
        if (MW > 0 && geo == 0 && L.nchunks == 1 && !a.part && L.KS * NW <= 24 && !(getenv("RVM_WRES") && atoi(getenv("RVM_WRES")) == 0)) {
            const size_t lw = conv_lds_bytes<T>(MW, NW, 3, 1, 1, L.CC, -L.KS), ls = conv_lds_bytes<T>(MW, NW, 3, 1, 1, L.CC, a.SK);
            const bool same_occupancy = lw <= 52 * 1024 ? true : (lw <= 80 * 1024 ? ls > 52 * 1024 : (lw <= 150 * 1024 && ls > 80 * 1024));   // 2 x 80 KiB = the CU
            if (same_occupancy) a.wres = 1;
            if (getenv("RVM_VERBOSE") && c->arena) fprintf(stderr, "[rvm] %-16s wres %d (resident %zu B, staged %zu B)\n", name.c_str(), a.wres, lw, ls);
        }
        const double KK = (double)L.KD * L.KD;
        const bool in_enc = c->enc_phase;
        const int ws_set = c->enc_bytes ? (int)(c->enc_shift / c->enc_bytes) : 0;
        auto fn = [=](hipStream_t st, const FrameIO& io) {
            ConvArgs b = a; fill(b, io);
            if (a.part) b.part = in_enc ? c->split_ws_set[ws_set] : c->split_ws;
            launch_conv3_m<T>(b, B, MW, NW, geo, st, split);
        };
        if (keep) { *keep = fn; return 0; }
        // algorithmic bytes: every source at the resolution and element size it is READ at (a x2-upsampled source is
        // read at half size, an fp16 state as 2 bytes), the epilogue's own operands, the output(s), the weights once
        double px_bytes = 0;
        {
            ConvArgs probe = a; FrameIO none; memset(&none, 0, sizeof none);
            fill(probe, none);
            for (int k = 0; k < probe.nsrc; ++k) {
                const ConvSrc& S = probe.src[k];
                px_bytes += (double)(S.up ? S.Hs * S.Ws : Hi * Wi) * S.C * (S.f16 ? 2 : es);
            }
            const double opx = (double)Ho * Wo;
            if (epi == EPI_RELU) px_bytes += opx * L.Cout * es;
            else if (epi == EPI_GRU_IH) px_bytes += opx * probe.hC * (es + 4);                    // r*h and z out (h is a source)
            else px_bytes += opx * probe.hC * (4 + es + es + (probe.out2_mode ? 2 : 0));          // z and h in, h' out (+ the caller's 16-bit copy)
        }
        add(name, 2.0 * B * Ho * Wo * KK * L.Cin * L.Cout, (double)B * px_bytes + KK * L.Cin * L.Cout * es, fn);
        if (a.part) c->launches.back().nk = 2;
        return 0;
    };

    T* f[4] = {nullptr, nullptr, nullptr, nullptr};
    const int HW4 = lh[4] * lw[4];
    const int MT4 = cdiv(HW4, 16);
    T* f4 = nullptr;
    float* f4pool = nullptr;
    int pool_stride = 0;
    if (c->variant == RVM_VARIANT_MOBILENETV3) {
        // ---- encoder (mobilenetv3.py:36-60)
        T* cur = stem_out;
        int curC = 16, lvl = 1;
        for (int i = 0; i < 15; ++i) {
            const MbBlock& b = c->mb[i];
            const int Hi = lh[lvl], Wi = lw[lvl];
            const int olvl = b.dw.stride == 2 ? lvl + 1 : lvl;
            const int Ho = lh[olvl], Wo = lw[olvl];
            const int exp = b.row.exp;
            const int K = b.dw.K;
            const int npix = Ho * Wo;
            // ---- blocks without squeeze-excite (features[2], [3], [7]..[10]): the whole block in one launch (mbconv_fused.hip)
            //      - expand, depthwise and project per tile of output pixels, nothing but the block's input and output in memory
            if (b.has_expand && !b.has_se && !b.pj && K == 3 && b.dw.dil == 1 && b.expand.act == b.dw.act &&
                mbf_supported(b.dw.stride, b.expand.KS, b.expand.act) && !(getenv("RVM_MB_FUSE") && atoi(getenv("RVM_MB_FUSE")) == 0)) {
                MbfArgs a; memset(&a, 0, sizeof a);
                a.x = cur; a.Cin = curC;
                a.we = wb + b.expand.w; a.KSe = b.expand.KS; a.NTe = b.expand.NT; a.ebias = (const float*)(wb + b.expand.bias); a.act = b.expand.act;
                a.dw_w = (const float*)(wb + b.dw.w); a.dw_b = (const float*)(wb + b.dw.bias);
                a.wp = wb + b.project.w; a.KSp = b.project.KS; a.NTp = b.project.NT; a.pbias = (const float*)(wb + b.project.bias);
                a.res = b.res ? cur : nullptr; a.Cout = b.row.cout; a.C = exp;
                a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.stride = b.dw.stride;
                // features[1] (depthwise 3x3 + 16 x 16 project + residual on the stem output) folded into features[2]: its
                // launch, just planned, is dropped again; its output buffer stays (the f1 skip, written by the fused kernel)
                const bool pre = i == 1 && c->mb[0].pj && c->mb[0].dw.act == ACT_RELU && stem_out && !c->launches.empty() &&
                                 c->launches.back().name == "mb1.dw3_project" && !(getenv("RVM_MB1_FOLD") && atoi(getenv("RVM_MB1_FOLD")) == 0);
                if (pre) {
                    a.pre_x = stem_out; a.pre_y = cur;
                    a.pre_dw_w = (const float*)(wb + c->mb[0].dw.w); a.pre_dw_b = (const float*)(wb + c->mb[0].dw.bias);
                    a.pre_pj_w = (const float*)(wb + c->mb[0].pj_w); a.pre_pj_b = (const float*)(wb + c->mb[0].pj_b);
                }
                // Tile candidates (rows x columns of output pixels).  One frame: the launch is a chain of round trips, so
                // the smallest per-block work that still keeps the grid within one round of CUs wins; many frames: plan-time
                // measurement decides (larger tiles recompute less halo).
                static const int cand[9][2] = {{2, 8}, {4, 4}, {4, 8}, {2, 16}, {8, 4}, {4, 16}, {8, 8}, {2, 32}, {16, 4}};
                int valid[9], nvalid = 0, model = -1;
                double best = 1e30;
                for (int k = 0; k < 9; ++k) {
                    a.THo = cand[k][0]; a.TWo = cand[k][1];
                    if (!mbf_prepare(a, (int)es)) continue;
                    const int thi = (a.THo - 1) * a.stride + 3, twi = (a.TWo - 1) * a.stride + 3, mte = cdiv(thi * twi, 16), tp = a.THo * a.TWo;
                    const int blocks = cdiv(Ho, a.THo) * cdiv(Wo, a.TWo) * B;
                    const int nge = a.ngE, npw = a.NTe > 8 ? 2 : 1;
                    // cycles of the three phases of one block (MFMA issue + LDS traffic, fitted by eye; refined by measurement on the box)
                    const double cyc = (double)cdiv(mte, nge) * npw * (48.0 * a.KSe * (split ? 3 : 1) + 120.0) + (double)cdiv((exp / 4) * tp / 2, 512) * 500.0 +
                                       (double)cdiv(a.KSp, a.KP) * (tp / 16) * (48.0 * (split ? 3 : 1) + 40.0) + 600.0;
                    const double cost = (double)cdiv(blocks, 256) * (3.2 + cyc / 2000.0);
                    if (cost < best - 1e-9) { best = cost; model = nvalid; }
                    valid[nvalid++] = k;
                }
                if (nvalid > 0) {
                    if (pre) c->launches.pop_back();
                    int pick = valid[tune_pick("mb" + std::to_string(i + 1) + ".fused", nvalid, model)];
                    if (const char* e = getenv("RVM_MBF_TILE")) {       // test hook: "rows,cols" forces that tile where it is valid
                        int th = 0, tw = 0;
                        if (sscanf(e, "%d,%d", &th, &tw) == 2)
                            for (int k = 0; k < nvalid; ++k) if (cand[valid[k]][0] == th && cand[valid[k]][1] == tw) pick = valid[k];
                    }
                    a.THo = cand[pick][0]; a.TWo = cand[pick][1];
                    a.tiles_x = cdiv(Wo, a.TWo); a.tiles_y = cdiv(Ho, a.THo);
                    if (!mbf_prepare(a, (int)es)) return fail("mbconv_fused: tile preparation failed");
                    T* out = abuf((size_t)B * npix * b.row.cout);
                    a.y = out;
                    if (getenv("RVM_VERBOSE") && c->arena)
                        fprintf(stderr, "[rvm] mb%d.fused tile %dx%d blocks %d lds %zu\n", i + 1, a.THo, a.TWo, a.tiles_x * a.tiles_y * B, (size_t)a.lds_total);
                    add("mb" + std::to_string(i + 1) + ".fused",
                        2.0 * B * ((double)Hi * Wi * curC * exp + (double)npix * exp * 9 + (double)npix * exp * b.row.cout + (pre ? (double)Hi * Wi * (16 * 9 + 256) : 0.0)),
                        ((double)B * (Hi * Wi * curC * (pre ? 2 : 1) + npix * b.row.cout * (b.res ? 2 : 1)) + (double)curC * exp + 10.0 * exp + (double)exp * b.row.cout) * es,
                        [=](hipStream_t st, const FrameIO&) {
                            if constexpr (std::is_same<T, float>::value) { if (split) { launch_mbf<f32s>(a, B, st); return; } }
                            launch_mbf<T>(a, B, st);
                        });
                    cur = out; curC = b.row.cout; lvl = olvl;
                    if (i == 2) f[1] = cur;
                    continue;
                }
            }
            T* dwo = b.pj ? nullptr : abuf((size_t)B * npix * exp);
            T* pj_done = nullptr;                              // block 1: the fused launch already wrote the block's output
            PwX se_x;                                          // squeeze-excite tail handed to the project GEMM (small SE layers)
            const T* scale = nullptr;
            // ---- expand -> depthwise (-> SE fc1 partials) in one launch: 16 channels x a band of rows per block
            {
                const int nslice = cdiv(exp, 16);
                auto rows_for = [&](int rb) { return std::min(Hi, (rb - 1) * b.dw.stride + (K - 1) * b.dw.dil + 1); };
                const int KS = b.has_expand ? b.expand.KS : 0;
                auto trows_for = [&](int rb) { return (rb - 1) * b.dw.stride + (K - 1) * b.dw.dil + 1; };      // zero-padded tile rows
                // Column tiles: a full-width LDS tile of the widest layers does not fit for low-res frames wider than
                // ~1800 px (1080p at ds = 1, 4K at ds = 0.5); the tile then covers `cwt` output columns plus the halo.
                int cwt = Wo, ncol = 1;
                if (const char* e = getenv("RVM_ED_CWT")) {                 // test hook: force column tiles on small frames
                    const int v = atoi(e);
                    if (v >= 4 && v < Wo) { cwt = v; ncol = cdiv(Wo, cwt); }
                }
                auto tcols_for = [&](int cw_) { return cw_ >= Wo ? Wi + 2 * b.dw.pad : (cw_ - 1) * b.dw.stride + (K - 1) * b.dw.dil + 1; };
                auto tin_for = [&](int cw_) { return std::min(Wi, tcols_for(cw_)); };                           // real input columns of a tile
                auto lds_for = [&](int rb) { return ed_lds_bytes(K, KS, rup(trows_for(rb) * tcols_for(cwt), 16), (int)es); };
                // Rows per band: the 1024-thread blocks run one per CU, so the launch takes
                // ceil(blocks / 256) rounds of (tile rows) work each; pick the band height that minimises
                // rounds x tile rows with the tile inside LDS.  SE layers keep <= 256 partial rows for the
                // squeeze-excite reduction that follows.
                int rb = Ho;
                const int se_parts = 256;       // partial rows the squeeze-excite gemv sums per sample (measured: 64 / 128 / 256 -> 2275 / 2283 / 2290 fps)
                {
                    int valid[16], nvalid = 0, model = -1;
                    for (;;) {
                        double best = 1e30;
                        for (int cand = Ho; cand >= 1 && nvalid < 16; cand = cand > 1 ? cdiv(cand, 2) : 0) {
                            const int bands = cdiv(Ho, cand);
                            if (lds_for(cand) > 150 * 1024 || rup(cand * std::min(cwt, Wo), 64) > 4096 || rows_for(cand) * tin_for(cwt) >= 65536) continue;
                            if (b.has_se && ncol == 1 && nslice * bands > se_parts && cand != Ho) continue;
                            const double cost = (double)cdiv(nslice * bands * ncol * B, 256) * (trows_for(cand) + 2);   // + fixed per-block part
                            if (cost < best - 1e-9) { best = cost; model = nvalid; }
                            valid[nvalid++] = cand;
                        }
                        if (nvalid > 0 || cwt <= 8) break;
                        cwt = cdiv(cwt, 2); ncol = cdiv(Wo, cwt);         // nothing fits at this width: halve the column tile
                    }
                    if (nvalid == 0) return fail("mbconv_ed: tile does not fit LDS");
                    if ((int64_t)cdiv(Ho, 1) * ncol > 65535) return fail("mbconv_ed: frame too large for the grid");
                    rb = valid[tune_pick("mb" + std::to_string(i + 1) + ".ed", nvalid, model)];
                }
                EdArgs a; memset(&a, 0, sizeof a);
                a.x = cur; a.Cin = curC; a.xs = curC;
                // Every block of the fused kernel reads the band's whole input (all Cin channels); a CU pulls
                // only ~30-70 GB/s, so when that is more than ~160 KB per block (the 5x5 dilated layers at
                // 1/16 scale, whose halo forbids bands) the expand runs as its own full-chip GEMM instead and
                // the fused kernel starts from the expanded tensor (16-channel slices, no redundancy).
                const size_t xbytes = (size_t)rows_for(rb) * tin_for(cwt) * curC * es;
                bool fuse_e = b.has_expand && b.expand.KS <= 4 && !(xbytes > 160 * 1024 && !getenv("RVM_ED_ALWAYS_FUSE"));
                if (b.has_expand && !fuse_e) {
                    T* e = abuf((size_t)B * Hi * Wi * exp);
                    pw("mb" + std::to_string(i + 1) + ".expand", b.expand, cur, curC, Hi * Wi, e, exp, nullptr, nullptr, 0, nullptr, nullptr);
                    a.x = e; a.Cin = exp; a.xs = exp;
                }
                if (fuse_e) { a.w = wb + b.expand.w; a.KS = b.expand.KS; a.NT = b.expand.NT; a.ebias = (const float*)(wb + b.expand.bias); a.eact = b.expand.act; }
                a.dw_w = (const float*)(wb + b.dw.w); a.dw_b = (const float*)(wb + b.dw.bias); a.dact = b.dw.act;
                T* pj_out = b.pj ? abuf((size_t)B * npix * b.row.cout) : nullptr;
                pj_done = pj_out;
                if (b.pj) { a.pj_w = (const float*)(wb + b.pj_w); a.pj_b = (const float*)(wb + b.pj_b); }
                a.y = b.pj ? pj_out : dwo; a.C = exp; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.stride = b.dw.stride; a.dil = b.dw.dil; a.pad = b.dw.pad;
                a.rb = rb; a.nband = cdiv(Ho, rb); a.npxs = rup(trows_for(rb) * tcols_for(cwt), 16);
                a.cwt = cwt; a.ncol = ncol;
                { const char* e = getenv("RVM_DBG"); a.dbg = e ? atoi(e) >> 4 : 0; }
                if (a.KS > 4 || rup(rb * std::min(cwt, Wo), 64) > 4096 || lds_for(rb) > 160 * 1024 || rows_for(rb) * tin_for(cwt) >= 65536 || (b.has_se && b.se.S > 1024))
                    return fail("mbconv_ed: unsupported block shape");
                // an already expanded layer with few 16-channel blocks (the 1/16-scale layers: <= 60 on 256 CUs) splits
                // into 8-channel blocks: half the depthwise work per block, twice the CUs
                a.cw = (!fuse_e && b.has_expand && exp % 8 == 0 && nslice * a.nband * ncol * B <= 128) ? 8 : 16;
                const int nparts = cdiv(exp, a.cw) * a.nband * ncol;
                float* hpart = nullptr;
                if (b.has_se) {
                    hpart = fbuf((size_t)B * nparts * b.se.S);
                    a.fc1_w = (const float*)(wb + b.se.w1f); a.S = b.se.S; a.hpart = hpart;
                }
                const double eflops = fuse_e ? 2.0 * B * Hi * Wi * curC * exp : 0.0;
                add("mb" + std::to_string(i + 1) + (b.pj ? ".dw" : (fuse_e || !b.has_expand ? ".expand_dw" : ".dw")) + std::to_string(K) + (b.pj ? "_project" : ""),
                    eflops + 2.0 * B * npix * exp * K * K + (b.pj ? 2.0 * B * npix * exp * b.row.cout : 0.0), ((double)B * (Hi * Wi * curC + npix * exp) + (double)curC * exp) * es,
                    [=](hipStream_t st, const FrameIO&) {
                        if constexpr (std::is_same<T, float>::value) { if (split) { launch_ed<f32s>(a, B, K, st); return; } }
                        launch_ed<T>(a, B, K, st);
                    });
                // Small SE layers (the 1/8-scale blocks: S = 24 / 32, <= 120 channels): the tail below fits the prologue
                // of the project GEMM, one launch less each
                if (b.has_se && b.se.S <= 32 && b.se.S % 4 == 0 && exp <= 256 && nparts <= 256 && !getenv("RVM_SE_GEMV")) {
                    se_x.se_hpart = hpart; se_x.se_nparts = nparts; se_x.se_S = b.se.S; se_x.se_xscale = 1.f / (float)npix;
                    se_x.se_b1 = (const float*)(wb + b.se.b1); se_x.se_w2 = wb + b.se.w2; se_x.se_b2 = (const float*)(wb + b.se.b2);
                } else if (b.has_se) {
                    // scale = hardsigmoid(fc2(relu(fc1(mean)))): the fc1 pre-activations arrive as per-block partials
                    T* sc = abuf((size_t)B * exp);
                    GemvArgs g2; memset(&g2, 0, sizeof g2);
                    g2.xin = hpart; g2.nparts = nparts; g2.part_stride = b.se.S; g2.xscale = 1.f / (float)npix;
                    g2.xbias = (const float*)(wb + b.se.b1); g2.xact = ACT_RELU;
                    g2.w = wb + b.se.w2; g2.bias = (const float*)(wb + b.se.b2); g2.y = sc; g2.y_is_T = 1;
                    g2.I = b.se.S; g2.O = exp; g2.act = ACT_HSIGMOID;
                    add("mb" + std::to_string(i + 1) + ".se_fc", 4.0 * B * exp * b.se.S, 2.0 * exp * b.se.S * es,
                        [=](hipStream_t st, const FrameIO&) { launch_gemv<T>(g2, B, st); });
                    scale = sc;
                }
            }
            T* out = nullptr;
            if (b.pj) out = pj_done;
            else {
                out = abuf((size_t)B * npix * b.row.cout);
                pw("mb" + std::to_string(i + 1) + ".project", b.project, dwo, exp, npix, out, b.row.cout, scale, b.res ? cur : nullptr, curC, nullptr, nullptr, se_x);
            }
            cur = out; curC = b.row.cout; lvl = olvl;
            if (i == 0) f[0] = cur;
            if (i == 2) f[1] = cur;
            if (i == 5) f[2] = cur;
        }
        f4 = abuf((size_t)B * HW4 * c->enc_out);
        pool_stride = c->last.NT * 16;
        f4pool = fbuf((size_t)B * MT4 * pool_stride);
        pw("features16_1x1", c->last, cur, curC, HW4, f4, c->enc_out, nullptr, nullptr, 0, nullptr, f4pool);
    } else {
        // ---- ResNet50 encoder (resnet.py:20-33)
        T* x1 = abuf((size_t)B * lh[1] * lw[1] * 64);
        {
            const T* s0 = s[0];
            if (conv3("conv1_7x7s2", c->res_stem, lh[1], lw[1], lh[0], lw[0], EPI_RELU, [=](ConvArgs& a, const FrameIO&) {
                    a.src[0] = ConvSrc{s0, 4, 4, 0, 0, 0, 0}; a.nsrc = 1; a.y = x1; a.ys = 64;
                })) return 1;
        }
        f[0] = x1;
        T* cur = abuf((size_t)B * lh[2] * lw[2] * 64);
        {
            MaxPoolArgs m; m.x = x1; m.y = cur; m.C = 64; m.Hi = lh[1]; m.Wi = lw[1]; m.Ho = lh[2]; m.Wo = lw[2];
            add("maxpool3x3s2", 9.0 * B * lh[2] * lw[2] * 64, (double)B * (lh[1] * lw[1] + lh[2] * lw[2]) * 64 * es,
                [=](hipStream_t st, const FrameIO&) { launch_maxpool<T>(m, B, st); });
        }
        int lvl = 2, curC = 64;
        const int nres = (int)c->res.size();
        for (int i = 0; i < nres; ++i) {
            const ResBlock& R = c->res[i];
            const std::string nm = "res" + std::to_string(i);
            const int Hi = lh[lvl], Wi = lw[lvl];
            const int olvl = R.stride == 2 ? lvl + 1 : lvl;
            const int Ho = lh[olvl], Wo = lw[olvl];
            T* t1 = abuf((size_t)B * Hi * Wi * R.planes);
            pw(nm + ".conv1", R.c1, cur, curC, Hi * Wi, t1, R.planes, nullptr, nullptr, 0, nullptr, nullptr);
            T* t2 = abuf((size_t)B * Ho * Wo * R.planes);
            {
                const int P = R.planes;
                if (conv3(nm + ".conv2", R.c2, Ho, Wo, Hi, Wi, EPI_RELU, [=](ConvArgs& a, const FrameIO&) {
                        a.src[0] = ConvSrc{t1, P, P, 0, 0, 0, 0}; a.nsrc = 1; a.y = t2; a.ys = P;
                    })) return 1;
            }
            const int OC = R.planes * 4;
            const T* idt = cur; int idtC = curC;
            if (R.has_ds) {
                T* d = abuf((size_t)B * Ho * Wo * OC);
                PwX ex; ex.sstride = R.stride; ex.sHi = Hi; ex.sWi = Wi; ex.sWo = Wo;
                pw(nm + ".downsample", R.ds, cur, curC, Ho * Wo, d, OC, nullptr, nullptr, 0, nullptr, nullptr, ex);
                idt = d; idtC = OC;
            }
            const bool last = i == nres - 1;
            T* out = abuf((size_t)B * Ho * Wo * OC);
            if (last) { pool_stride = R.c3.NT * 16; f4pool = fbuf((size_t)B * MT4 * pool_stride); }
            PwX ex; ex.act_post = 1;
            pw(nm + ".conv3", R.c3, t2, R.planes, Ho * Wo, out, OC, nullptr, idt, idtC, nullptr, last ? f4pool : nullptr, ex);
            cur = out; curC = OC; lvl = olvl;
            if (i == 2) f[1] = cur;          // layer1 -> f2
            if (i == 6) f[2] = cur;          // layer2 -> f3
        }
        f4 = cur;
    }
    c->taps["f1"] = Tap{f[0], B, lh[1], lw[1], c->feat[0], c->feat[0], false};
    c->taps["f2"] = Tap{f[1], B, lh[2], lw[2], c->feat[1], c->feat[1], false};
    c->taps["f3"] = Tap{f[2], B, lh[3], lw[3], c->feat[2], c->feat[2], false};
    c->taps["f4"] = Tap{f4, B, lh[4], lw[4], c->enc_out, c->enc_out, false};

    // ---- LR-ASPP (lraspp.py:17-18)
    float* gate = fbuf((size_t)B * c->aspp_c);
    {
        GemvArgs g; memset(&g, 0, sizeof g);
        g.xin = f4pool; g.nparts = MT4; g.part_stride = pool_stride; g.xscale = 1.f / (float)HW4;
        g.w = wb + c->aspp2_w; g.bias = nullptr; g.y = gate; g.y_is_T = 0; g.I = c->enc_out; g.O = c->aspp_c;
        g.act = ACT_SIGMOID;
        add("aspp2_gate", 2.0 * B * g.I * g.O, (double)g.I * g.O * es, [=](hipStream_t st, const FrameIO&) { launch_gemv<T>(g, B, st); });
    }
    const int AC = c->aspp_c;
    T* aspp = abuf((size_t)B * HW4 * AC);
    pw("aspp1_1x1", c->aspp1, f4, c->enc_out, HW4, aspp, AC, nullptr, nullptr, 0, gate, nullptr);
    c->taps["aspp"] = Tap{aspp, B, lh[4], lw[4], AC, AC, false};

    // ---- phase boundary of the frame pipeline: everything above is independent of the recurrent state.  One set of the
    // encoder's buffers ends here; the decoder region follows the second set.  The state preparation, part of the first
    // launch otherwise, becomes the decoder phase's first launch (it reads the states the frame before produced).
    c->n_enc = (int)c->launches.size();
    if (c->pipe) {
        const size_t E = (c->arena_used + 255) & ~size_t(255);
        if (c->enc_shift && E != c->enc_bytes) return fail("rvm_plan: the two buffer sets of the frame pipeline disagree in size");
        c->enc_bytes = E;
        c->arena_used = (size_t)(c->enc_stages + 1) * E;
        c->enc_phase = false;
        if (io16) {
            state_buffers();
            const StatePrep sp0 = sprep;
            double sbytes = 0;
            for (int r = 0; r < 4; ++r) sbytes += (double)sp0.n[r] * (2 + 4 + 4);
            add("state_prep", 0.0, sbytes, [=](hipStream_t st, const FrameIO& io) {
                StatePrep sp = sp0;
                for (int r = 0; r < 4; ++r) sp.in[r] = io.r_in[r];
                launch_state_prep(sp, io16, st);
            });
        }
    }

    // ---- recurrent decoder (decoder.py:17-27)
    // GRU on the second half `xb` (channel stride xs) of a block's features; state level idx 3..0 -> r4..r1
    const size_t ses = es;                          // element size of a state inside the plan (16-bit API states never get here)
    char* hseq[4] = {nullptr, nullptr, nullptr, nullptr};   // T > 1: the T outputs of a level's GRU, [T][Bs][H][W][C]
    auto gru = [&](const std::string& name, const GruLayer& G, const T* xb, int xs, int Hh, int Ww, int ridx) -> int {
        const int C = G.C;
        const size_t per = (size_t)Bs * Hh * Ww;    // pixels of one time step
        // io16: fp32 states in the arena for every step (allocated above); else for chunks only, in the state's type
        char* hs = io16 ? (char*)hseq32[ridx] : (TT > 1 ? (char*)c->arena_alloc((size_t)TT * per * C * ses) : nullptr);
        hseq[ridx] = hs;
        const float* hin = hin32[ridx];
        // ---- the two large-map GRUs (decode1, decode2): the whole step in one launch (gru_fused.hip)
        int fusedTH = 0, fusedTW = 0;
        if (gru_fused_supported(C, (int)es) && G.ih.nchunks == 1 && G.hh.nchunks == 1 && G.ih.Cpad == 2 * C && G.hh.Cpad == 2 * C &&
            G.ih.NT == 2 * cdiv(C, 16) && G.hh.NT == cdiv(C, 16) && !(getenv("RVM_GRU_FUSE") && atoi(getenv("RVM_GRU_FUSE")) == 0)) {
            // Tile candidates.  Small tiles recompute more of the gate conv's one-pixel ring ((TH + 2)(TW + 2) / (TH TW): 1.9x at
            // 4 x 8, 1.4x at 8 x 16) but one frame has too few large tiles for 256 CUs; many frames: large tiles, measured at
            // plan time where the plan is tuned.
            static const int cand[4][2] = {{4, 8}, {8, 8}, {4, 16}, {8, 16}};
            int model = 0, valid[4], nvalid = 0; double best = 1e30;
            for (int k = 0; k < 4; ++k) {
                const int th = cand[k][0], tw = cand[k][1], blocks = cdiv(Hh, th) * cdiv(Ww, tw) * Bs;
                if (gru_fused_lds(C, th, tw, (int)es) > 160 * 1024) continue;
                const int wp1 = 8 / G.ih.NT, wp2 = 8 / G.hh.NT, mt1 = cdiv((th + 2) * (tw + 2), 16), mt2 = th * tw / 16;
                // MFMA cycles of a block per SIMD (two of its waves each: all m-tiles over a slice of K) + the phases that do not
                // overlap inside one block
                const double mf = 2.0 * (mt1 * cdiv(G.ih.KS, wp1) + mt2 * cdiv(G.hh.KS, wp2)) * 16.0 * (split ? 3 : 1);
                int occ = 1;
                if constexpr (std::is_same<T, float>::value) occ = split ? gru_fused_occupancy<f32s>(C, th, tw) : gru_fused_occupancy<float>(C, th, tw);
                else occ = gru_fused_occupancy<T>(C, th, tw);
                occ = std::min(occ, 2);
                const double cost = (double)cdiv(blocks, 256 * occ) * (occ * mf + 3000.0);
                if (cost < best - 1e-9) { best = cost; model = nvalid; }
                valid[nvalid++] = k;
            }
            int pick = valid[tune_pick(name + ".fused", nvalid, model)];
            if (const char* e = getenv("RVM_GRU_TILE")) {                   // test hook: "rows,cols"
                int th = 0, tw = 0;
                if (sscanf(e, "%d,%d", &th, &tw) == 2)
                    for (int k = 0; k < nvalid; ++k) if (cand[valid[k]][0] == th && cand[valid[k]][1] == tw) pick = valid[k];
            }
            fusedTH = cand[pick][0]; fusedTW = cand[pick][1];
            if (getenv("RVM_VERBOSE") && c->arena)
                fprintf(stderr, "[rvm] %s.fused tile %dx%d blocks %d\n", name.c_str(), fusedTH, fusedTW, cdiv(Hh, fusedTH) * cdiv(Ww, fusedTW) * Bs);
        }
        T* rh = fusedTH ? nullptr : abuf(per * C);
        float* z = fusedTH ? nullptr : fbuf(per * C);
        for (int t = 0; t < TT; ++t) {
            const std::string nm = TT > 1 ? name + "[" + std::to_string(t) + "]" : name;
            const T* xt = xb + (size_t)t * per * xs;
            const void* hp = t ? hs + (size_t)(t - 1) * per * C * ses : nullptr;      // nullptr: the state coming in
            void* ho = hs ? hs + (size_t)t * per * C * ses : nullptr;                 // nullptr: the caller's r_out
            const bool last = t == TT - 1;
            if (fusedTH) {
                GruArgs a; memset(&a, 0, sizeof a);
                a.x = xt; a.xs = xs; a.hmode = es == 2 ? 1 : 0; a.omode = a.hmode; a.omode2 = io16 ? io16 : a.hmode;
                a.wih = wb + G.ih.w; a.bih = (const float*)(wb + G.ih.bias); a.whh = wb + G.hh.w; a.bhh = (const float*)(wb + G.hh.bias);
                a.C = C; a.H = Hh; a.W = Ww; a.TH = fusedTH; a.TW = fusedTW; a.tiles_x = cdiv(Ww, fusedTW); a.tiles_y = cdiv(Hh, fusedTH);
                const double px = (double)per;
                add(nm + ".fused", 2.0 * px * 9.0 * 2 * C * 3 * C, px * C * 3.0 * es + (double)(G.ih.KS * G.ih.NT + G.hh.KS * G.hh.NT) * 512 * es,
                    [=](hipStream_t st, const FrameIO& io) {
                        GruArgs b = a;
                        b.h = t ? hp : (io16 ? (const void*)hin : io.r_in[ridx]);
                        b.hout = ho ? ho : io.r_out[ridx];
                        b.hout2 = (ho && last) ? io.r_out[ridx] : nullptr;            // the state handed back
                        if constexpr (std::is_same<T, float>::value) { if (split) { launch_gru_fused<f32s>(b, Bs, st); return; } }
                        launch_gru_fused<T>(b, Bs, st);
                    });
                continue;
            }
            if (conv3(nm + ".ih", G.ih, Hh, Ww, Hh, Ww, EPI_GRU_IH, [=](ConvArgs& a, const FrameIO& io) {
                    const void* h = t ? hp : (io16 ? (const void*)hin : io.r_in[ridx]);
                    a.src[0] = ConvSrc{xt, C, xs, 0, 0, 0, 0}; a.src[1] = ConvSrc{h, C, C, 0, 0, 0, 0}; a.nsrc = 2;
                    a.hprev = h; a.hC = C; a.rh = rh; a.z = z;
                }, Bs)) return 1;
            if (conv3(nm + ".hh", G.hh, Hh, Ww, Hh, Ww, EPI_GRU_HH, [=](ConvArgs& a, const FrameIO& io) {
                    a.src[0] = ConvSrc{xt, C, xs, 0, 0, 0, 0}; a.src[1] = ConvSrc{rh, C, C, 0, 0, 0, 0}; a.nsrc = 2;
                    a.hprev = t ? hp : (io16 ? (const void*)hin : io.r_in[ridx]); a.hC = C; a.z = z;
                    a.hout = ho ? ho : io.r_out[ridx];
                    a.hout2 = (ho && last) ? io.r_out[ridx] : nullptr;                 // the state handed back
                    a.out2_mode = io16;
                }, Bs)) return 1;
        }
        return 0;
    };
        // decode4 (BottleneckBlock, decoder.py:63-67): a = aspp[:, :AC/2], b = aspp[:, AC/2:]
    if (gru("decode4.gru", c->gru[0], aspp + AC / 2, AC, lh[4], lw[4], 3)) return 1;
    const T* xa = aspp; int xaC = AC / 2, xaS = AC;   // first half of the previous block output
    const T* skip[3] = {f[2], f[1], f[0]};
    const int skipC[3] = {c->feat[2], c->feat[1], c->feat[0]};
    for (int i = 0; i < 3; ++i) {
        const int l = 3 - i;                        // output level 3,2,1
        const int Cout = c->dec[i];
        T* X = abuf((size_t)B * lh[l] * lw[l] * Cout);
        const int hidx = 3 - i;                     // state produced by the previous block: r4, r3, r2
        const T* sk = skip[i]; const int skC = skipC[i];
        const T* sl = s[l];
        const int Hs = lh[l + 1], Ws = lw[l + 1];
        const T* xa_ = xa; const int xaC_ = xaC, xaS_ = xaS;
        const char* hs_ = hseq[hidx];
        if (conv3("decode" + std::to_string(l) + ".conv", c->up[i], lh[l], lw[l], lh[l], lw[l], EPI_RELU, [=](ConvArgs& a, const FrameIO& io) {
                a.src[0] = ConvSrc{xa_, xaC_, xaS_, 1, Hs, Ws, 0};
                a.src[1] = ConvSrc{hs_ ? (const void*)hs_ : io.r_out[hidx], xaC_, xaC_, 1, Hs, Ws, 0};
                a.src[2] = ConvSrc{sk, skC, skC, 0, 0, 0, 0};
                a.src[3] = ConvSrc{sl, 4, 4, 0, 0, 0, 0};
                a.nsrc = 4; a.y = X; a.ys = Cout;
            }, 0, nullptr, &c->up_ch[i])) return 1;
        if (gru("decode" + std::to_string(l) + ".gru", c->gru[i + 1], X + Cout / 2, Cout, lh[l], lw[l], hidx - 1)) return 1;
        c->taps["x" + std::to_string(l)] = Tap{X, B, lh[l], lw[l], Cout, Cout, false};
        xa = X; xaC = Cout / 2; xaS = Cout;
    }
    // decode0 (OutputBlock, decoder.py:127-131)
    const int HC = c->dec[3];
    const bool conv1_in_dgf = c->ds != 1.0 && c->refiner != RVM_REFINER_FAST_GUIDED_FILTER && !getenv("RVM_DGF_VALU") &&
                              HC == 16 && c->out1.KS == 5 && c->out1.NT == 1 && c->out1.nchunks == 1;
    std::function<void(hipStream_t, const FrameIO&)> conv1_fn;
    T* hidA = abuf((size_t)B * c->h * c->w * HC);
    T* hid = abuf((size_t)B * c->h * c->w * HC);
    {
        const T* xa_ = xa; const int xaC_ = xaC, xaS_ = xaS;
        const T* s0 = s[0];
        const int Hs = lh[1], Ws = lw[1];
        const char* hs_ = hseq[0];
        if (conv3("decode0.conv0", c->out0, c->h, c->w, c->h, c->w, EPI_RELU, [=](ConvArgs& a, const FrameIO& io) {
                a.src[0] = ConvSrc{xa_, xaC_, xaS_, 1, Hs, Ws, 0};
                a.src[1] = ConvSrc{hs_ ? (const void*)hs_ : io.r_out[0], xaC_, xaC_, 1, Hs, Ws, 0};
                a.src[2] = ConvSrc{s0, 4, 4, 0, 0, 0, 0};
                a.nsrc = 3; a.y = hidA; a.ys = HC;
            })) return 1;
        // The second conv runs inside the deep guided filter's low-res kernel when there is one (ds != 1, MFMA form); a
        // segmentation pass on such a plan, which skips the filter, still launches it on its own.
        if (conv3("decode0.conv1", c->out1, c->h, c->w, c->h, c->w, EPI_RELU, [=](ConvArgs& a, const FrameIO&) {
                a.src[0] = ConvSrc{hidA, HC, HC, 0, 0, 0, 0}; a.nsrc = 1; a.y = hid; a.ys = HC;
            }, 0, conv1_in_dgf ? &conv1_fn : nullptr)) return 1;
    }
    c->taps["hid"] = Tap{hid, B, c->h, c->w, HC, HC, false};
    if (conv1_in_dgf) {
        // hid stays in the refiner kernel's LDS on a refining pass (B * h * w * 64 bytes of stores nobody reads); the debug
        // tap recomputes it from hidA with the stand-alone conv that a segmentation pass launches
        const std::function<void(hipStream_t, const FrameIO&)> conv1 = conv1_fn;
        c->taps["hid"].fill = [=](hipStream_t st) { FrameIO none; memset(&none, 0, sizeof none); conv1(st, none); };
    }

    // ---- projection + refiner + clamp (model.py:61-68)
    const DgfWeights* dw = (const DgfWeights*)(wb + c->dgf);
    float* ab = fbuf((size_t)B * c->h * c->w * 8);
    c->taps["ab"] = Tap{ab, B, c->h, c->w, 8, 8, true};
    {
        DgfLowArgs lo; memset(&lo, 0, sizeof lo);
        lo.hid = conv1_in_dgf ? nullptr : hid; lo.s0 = s[0]; lo.wt = dw; lo.ab = ab; lo.h = c->h; lo.w = c->w; lo.tiles_x = cdiv(c->w, 16);
        lo.fast = c->refiner == RVM_REFINER_FAST_GUIDED_FILTER;
        if (conv1_in_dgf) {
            lo.mlp = wb + c->dgf_mlp;
            lo.hin = hidA; lo.cw = wb + c->out1.w; lo.cbias = (const float*)(wb + c->out1.bias); lo.csplit = split ? 1 : 0;
        }
        DgfApplyArgs ap; memset(&ap, 0, sizeof ap); ap.ab = ab; ap.H = c->H; ap.W = c->W; ap.h = c->h; ap.w = c->w;
        ap.scale_y = (float)c->h / (float)c->H; ap.scale_x = (float)c->w / (float)c->W;
        ap.ncols = std::min(c->w, (int)std::ceil(std::min(2048, c->W) * (double)ap.scale_x) + 3);
        dgf_apply_pick(ap);
        ProjOutArgs po; po.hid = hid; po.wt = dw; po.H = c->h; po.W = c->w;
        const bool refine = c->ds != 1.0;
        const double lowpx = (double)B * c->h * c->w, fullpx = (double)B * c->H * c->W;
        if (refine) {
            const double dflops = lowpx * 2.0 * (64 + 4 * 9 * 4 + 24 * 16 + 16 * 16 + 16 * 4);
            if (conv1_in_dgf) {
                const std::function<void(hipStream_t, const FrameIO&)> conv1 = conv1_fn;
                add("decode0.conv1_dgf_lowres", dflops + lowpx * 2.0 * 9 * HC * HC, lowpx * (HC * es + 4 * es + 32) + 9.0 * HC * HC * es,     // hidA in, s0 in, A|b out (hid stays in LDS)
                    [=](hipStream_t st, const FrameIO& io) { if (io.seg) conv1(st, io); else launch_dgf_low<T>(lo, B, st); });
            } else {
                add("dgf_lowres", dflops, lowpx * (20 * es + 32),
                    [=](hipStream_t st, const FrameIO& io) { if (!io.seg) launch_dgf_low<T>(lo, B, st); });
            }
            add("dgf_apply_fullres", fullpx * 80.0, fullpx * 7 * ies + lowpx * 32,
                [=](hipStream_t st, const FrameIO& io) {
                    if (io.seg) {
                        ProjOutArgs q = po; q.src = io.src; q.fgr = io.fgr; q.pha = io.pha; q.seg = 1;
                        launch_proj_out<T>(q, B, io16, st);
                    } else {
                        DgfApplyArgs q = ap; q.src = io.src; q.fgr = io.fgr; q.pha = io.pha;
                        q.u8 = io.u8; q.u8mode = (int)io.u8mode; q.bgr[0] = io.bgr[0]; q.bgr[1] = io.bgr[1]; q.bgr[2] = io.bgr[2];
                        launch_dgf_apply<T>(q, B, io16, st);
                    }
                }, 2);
        } else {
            add("project_out", lowpx * 2.0 * 64, lowpx * (16 * es + 7 * ies),
                [=](hipStream_t st, const FrameIO& io) {
                    ProjOutArgs q = po; q.src = io.src; q.fgr = io.fgr; q.pha = io.pha; q.seg = (int)io.seg;
                    launch_proj_out<T>(q, B, io16, st);
                }, 2);
        }
    }
    c->split_ws = c->split_need ? (float*)c->arena_alloc(c->split_need) : nullptr;
    for (int k = 0; k < 5; ++k)
        c->split_ws_set[k] = (c->pipe && c->split_need && k <= c->enc_stages) ? (float*)c->arena_alloc(c->split_need) : c->split_ws;
    return plan_bad ? 1 : 0;
}

// Tile shapes by measurement.  With many frames per call (B*T > 1) a launch has blocks to spare and the
// latency model build_plan uses for one frame (rounds of blocks x work per block) no longer ranks the
// tiles; here every candidate of every GEMM / conv launch is timed on the device (hipEvents, a few
// repetitions on dummy states) and the fastest kept.  Runs inside rvm_plan, never on the frame path.
int autotune(rvm_ctx* c, bool only_always) {
    if (only_always) {
        bool any = false;
        for (const auto& t : c->tunable) any |= t.always && t.ncand > 1;
        if (!any) return 0;
    }
    auto rebuild = [&]() { return c->dtype == RVM_DTYPE_F16 ? build_plan<f16>(c) : build_plan<float>(c); };
    const size_t ses = c->dtype == RVM_DTYPE_F32 ? 4 : 2;
    const int ch[4] = {c->dec[2] / 2, c->dec[1] / 2, c->dec[0] / 2, c->feat[3] / 2};
    FrameIO io; memset(&io, 0, sizeof io);
    void* bufs[8] = {nullptr};
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        for (void* b : bufs) if (b) (void)hipFree(b);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (st) (void)hipStreamDestroy(st);
    };
#define TUNECHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(std::string("autotune: ") + hipGetErrorString(e_)); } } while (0)
    for (int l = 0; l < 4; ++l) {
        const size_t bytes = (size_t)c->B * ch[l] * c->lh[l + 1] * c->lw[l + 1] * ses;
        TUNECHK(hipMalloc(&bufs[l], bytes)); TUNECHK(hipMemset(bufs[l], 0, bytes));
        TUNECHK(hipMalloc(&bufs[4 + l], bytes)); TUNECHK(hipMemset(bufs[4 + l], 0, bytes));
        io.r_in[l] = bufs[l]; io.r_out[l] = bufs[4 + l];
    }
    TUNECHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    TUNECHK(hipEventCreate(&e0)); TUNECHK(hipEventCreate(&e1));
    TUNECHK(hipDeviceSynchronize());                         // the null-stream memsets (arena, dummy states) before the non-blocking stream runs
    const auto tunable = c->tunable;                      // build_plan refills the list
    const bool verbose = getenv("RVM_VERBOSE") != nullptr;
    for (const auto& t : tunable) {
        if (t.ncand < 2 || (only_always && !t.always)) continue;
        // the launches a key stands for: the one of that name, or - for "mbN.ed", the band height of the fused
        // expand -> depthwise kernel - that kernel together with the separate expand GEMM some heights imply
        const bool ed = t.key.size() > 3 && t.key.compare(t.key.size() - 3, 3, ".ed") == 0;
        const std::string stem = ed ? t.key.substr(0, t.key.size() - 2) : std::string();      // "mbN."
        auto members = [&]() {
            std::vector<int> m;
            for (size_t i = 0; i < c->launches.size(); ++i) {
                const std::string& nm = c->launches[i].name;
                if (ed ? (nm.compare(0, stem.size(), stem) == 0 && (nm.find("expand") != std::string::npos || nm.find(".dw") != std::string::npos))
                       : tune_key(nm) == t.key) { m.push_back((int)i); if (!ed) break; }
            }
            return m;
        };
        if (members().empty()) continue;
        float best = 1e30f; int bestk = t.model;
        for (int k = 0; k < t.ncand; ++k) {
            c->tuned[t.key] = k;
            if ((rc = rebuild())) { cleanup(); return rc; }
            if (c->arena_used + 256 > c->arena_bytes) continue;          // this candidate needs a bigger arena than the model's plan
            const std::vector<int> mem = members();
            const int warm = 3, reps = 10;
            for (int r = 0; r < warm; ++r) for (int i : mem) c->launches[i].fn(st, io);
            TUNECHK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) for (int i : mem) c->launches[i].fn(st, io);
            TUNECHK(hipEventRecord(e1, st));
            TUNECHK(hipStreamSynchronize(st));
            TUNECHK(hipGetLastError());
            float ms = 0.f;
            TUNECHK(hipEventElapsedTime(&ms, e0, e1));
            if (verbose) fprintf(stderr, "[rvm tune] %-20s cand %d  %.2f us%s\n", t.key.c_str(), k, ms / reps * 1e3f, k == t.model ? "  (model)" : "");
            if (ms < best) { best = ms; bestk = k; }
        }
        c->tuned[t.key] = bestk;
    }
#undef TUNECHK
    cleanup();
    if ((rc = rebuild())) return rc;
    if (c->arena_used + 256 > c->arena_bytes) return fail("autotune: chosen plan outgrew the arena");
    return 0;
}

int plan_impl(rvm_ctx* c, int B, int T, int H, int W, double ds) {
    if (!c->finalized) return fail("rvm_plan: call rvm_finalize first");
    if (B < 1 || T < 1 || H < 1 || W < 1) return fail("rvm_plan: bad shape");
    if ((int64_t)B * T > 65535) return fail("rvm_plan: B*T exceeds the grid limit (65535)");
    if (!(ds > 0.0 && ds <= 1.0)) return fail("rvm_plan: downsample_ratio must be in (0, 1]");
    if (c->planned && c->B == B && c->T == T && c->H == H && c->W == W && c->ds == ds && c->pipe == c->pipe_req) return 0;      // the active plan
    // park the active plan, then look for the requested one among the parked
    if (c->planned) {
        rvm_ctx::Parked p{};
        p.arena = nullptr; p.split_ws = nullptr;
        c->swap_plan(p);                                // p <- active plan; active <- empty
        c->parked.push_back(std::move(p));
        c->planned = false;
    }
    for (size_t i = 0; i < c->parked.size(); ++i) {
        rvm_ctx::Parked& p = c->parked[i];
        if (p.B == B && p.T == T && p.H == H && p.W == W && p.ds == ds && p.pipe == c->pipe_req) {
            c->swap_plan(p);
            c->parked.erase(c->parked.begin() + (long)i);
            c->planned = true;
            return 0;
        }
    }
    size_t keep = 4;                                    // parked plans kept beside the active one (RVM_KEEP_PLANS=0..8)
    if (const char* e = getenv("RVM_KEEP_PLANS")) keep = (size_t)std::min(8, std::max(0, atoi(e)));
    while (c->parked.size() > keep) {                   // least recently used first
        HIPCHK(hipDeviceSynchronize());
        for (auto& g : c->parked.front().graphs) (void)hipGraphExecDestroy(g.exec);
        for (auto& row : c->parked.front().stage_exec) for (auto& e : row) if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
        if (c->parked.front().arena) HIPCHK(hipFree(c->parked.front().arena));
        c->parked.erase(c->parked.begin());
    }
    c->B = B; c->T = T; c->H = H; c->W = W; c->ds = ds;
    c->pipe = c->pipe_req;
    for (bool& v : c->ev_dec_valid) v = false;          // (a new arena: nothing in flight reads it)
    c->enc_stages = 1;
    if (c->pipe) {
        const char* e = getenv("RVM_PIPE_STAGES");      // encoder stages (streams of the context's own): 1..4
        c->enc_stages = e ? std::min(4, std::max(1, atoi(e))) : 3;
    }
    if (ds == 1.0) { c->h = H; c->w = W; }
    else { c->h = (int)std::floor((double)H * ds); c->w = (int)std::floor((double)W * ds); }
    if (c->h < 1 || c->w < 1) return fail("rvm_plan: downsampled frame is empty");
    c->lh[0] = c->h; c->lw[0] = c->w;
    for (int l = 1; l < 5; ++l) { c->lh[l] = (c->lh[l - 1] + 1) / 2; c->lw[l] = (c->lw[l - 1] + 1) / 2; }
    c->drop_graphs();
    if (c->arena) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->arena)); c->arena = nullptr; }      // (a failed earlier plan)
    c->planned = false;
    c->tuned.clear();
    // a plan of the other kind (frame pipeline on / off) for this very shape: the same launches in another arena layout -
    // take over its tile choices instead of measuring again (and the two kinds then give the same bits)
    bool inherited = false;
    for (const auto& p : c->parked)
        if (p.B == B && p.T == T && p.H == H && p.W == W && p.ds == ds) { c->tuned = p.tuned; inherited = true; }
    // sizing pass (arena == nullptr), then the real pass
    c->enc_shift = 0;
    int rc = c->dtype == RVM_DTYPE_F16 ? build_plan<f16>(c) : build_plan<float>(c);
    if (rc) return rc;
    c->arena_bytes = c->arena_used + 256;
    HIPCHK(hipMalloc((void**)&c->arena, c->arena_bytes));
    HIPCHK(hipMemset(c->arena, 0, c->arena_bytes));
    HIPCHK(hipDeviceSynchronize());                           // the caller's (possibly non-blocking) stream must see the zeros
    c->enc_shift = 0;
    rc = c->dtype == RVM_DTYPE_F16 ? build_plan<f16>(c) : build_plan<float>(c);
    if (rc) return rc;
    {
        const char* e = getenv("RVM_AUTOTUNE");           // "0": never, "1": also for a single frame
        // measured tile selection: whenever a call carries several frames; always for resnet50 (the latency models that
        // rank the candidates for a single frame were fitted to the mobilenetv3 network: measured +7 % on the HD frame)
        const bool tune = e ? atoi(e) != 0 && ((int64_t)B * T > 1 || atoi(e) == 1)
                            : ((int64_t)B * T > 1 || c->variant == RVM_VARIANT_RESNET50);
        const bool off = e && atoi(e) == 0;               // the reproducible mode: model choices only
        if (!off && !inherited && (rc = autotune(c, !tune))) return rc;
    }
    c->launch_sets.clear(); c->tap_sets.clear();
    if (c->pipe) {
        // the other buffer sets: the same plan (same tile choices) with the encoder region one set further on each time
        for (int k = 1; k <= c->enc_stages; ++k) {
            c->enc_shift = (size_t)k * c->enc_bytes;
            rc = c->dtype == RVM_DTYPE_F16 ? build_plan<f16>(c) : build_plan<float>(c);
            if (!rc && c->arena_used + 256 > c->arena_bytes) rc = fail("rvm_plan: a buffer set of the frame pipeline outgrew the arena");
            c->launch_sets.push_back(std::move(c->launches)); c->tap_sets.push_back(std::move(c->taps));
            c->launches.clear(); c->taps.clear();
            if (rc) { c->enc_shift = 0; return rc; }
        }
        c->enc_shift = 0;
        rc = c->dtype == RVM_DTYPE_F16 ? build_plan<f16>(c) : build_plan<float>(c);
        if (rc) return rc;
        for (const auto& L : c->launch_sets)
            if (L.size() != c->launches.size()) return fail("rvm_plan: the buffer sets of the frame pipeline disagree");
        // stages of equal launch counts (the encoder's launches are all of the same order: 5-10 us)
        for (int j = 0; j < c->enc_stages; ++j) c->stage_end[j] = (int)((long)c->n_enc * (j + 1) / c->enc_stages);
    }
    c->frame_set = 0; c->last_set = 0;
    c->planned = true;
    return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* rvm_last_error(void) { return g_err.c_str(); }
const char* rvm_version(void) { return "rvm_mi355x 0.1 (gfx950)"; }

int rvm_create(int variant, int refiner, int dtype, int device, rvm_ctx** out) {
    if (!out) return fail("rvm_create: null out");
    *out = nullptr;
    if (variant != RVM_VARIANT_MOBILENETV3 && variant != RVM_VARIANT_RESNET50) return fail("rvm_create: bad variant");
    if (refiner != RVM_REFINER_DEEP_GUIDED_FILTER && refiner != RVM_REFINER_FAST_GUIDED_FILTER) return fail("rvm_create: bad refiner");
    if (dtype != RVM_DTYPE_F32 && dtype != RVM_DTYPE_F16 && dtype != RVM_DTYPE_F16IO && dtype != RVM_DTYPE_BF16IO) return fail("rvm_create: bad dtype");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("rvm_create: no such HIP device");
    DEVGUARD(device);
    {
        // once per device, with that device current: function attributes (dynamic LDS limits) and the address of the
        // allocation-free entry point's static scratch array (so that its first call on this device can run where nothing
        // may be looked up any more, e.g. under graph capture).  Contexts may be created from several host threads.
        static std::mutex init_mu;
        static bool inited[64] = {false};
        std::lock_guard<std::mutex> lock(init_mu);
        if (device < 64 && !inited[device]) {
            kernels_init();
            float* scratch_ptr = nullptr;
            if (metrics_scratch(&scratch_ptr)) return fail("rvm_create: hipGetSymbolAddress(scratch) failed");
            inited[device] = true;
        }
    }
    rvm_ctx* c = new rvm_ctx();
    c->variant = variant; c->refiner = refiner; c->dtype = dtype; c->device = device;
    if (dtype == RVM_DTYPE_F16IO || dtype == RVM_DTYPE_BF16IO) {
        const char* e = getenv("RVM_F16IO_MFMA");               // "f32": exact fp32 MFMA instead of the f16 split
        c->split = !(e && std::string(e) == "f32");
    }
    c->specs = param_specs(variant, refiner);
    *out = c;
    return 0;
}

void rvm_destroy(rvm_ctx* c) {
    if (!c) return;
    DeviceGuard guard_(c->device);
    (void)hipDeviceSynchronize();          // graphs / arena may still be in flight on the caller's streams
    c->drop_graphs();
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    for (int j = 0; j < 4; ++j) {
        bool dup = false;
        for (int i = 0; i < j; ++i) dup |= c->enc_streams[i] == c->enc_streams[j];
        if (c->enc_streams[j] && !dup) (void)hipStreamDestroy(c->enc_streams[j]);
    }
    for (auto& g : c->sig_stage) if (g) (void)hipFree(g);
    for (auto& row : c->ev_stage) for (auto& e : row) if (e) (void)hipEventDestroy(e);
    if (c->wdev) (void)hipFree(c->wdev);
    if (c->arena) (void)hipFree(c->arena);
    c->free_parked();
    delete c;
}

int rvm_param_count(const rvm_ctx* c) { return c ? (int)c->specs.size() : 0; }
const char* rvm_param_name(const rvm_ctx* c, int i) {
    if (!c || i < 0 || i >= (int)c->specs.size()) return nullptr;
    return c->specs[i].name.c_str();
}
int rvm_param_shape(const rvm_ctx* c, int i, int64_t shape[4]) {
    if (!c || i < 0 || i >= (int)c->specs.size()) { g_err = "rvm_param_shape: bad index"; return -1; }
    const auto& s = c->specs[i].shape;
    for (size_t k = 0; k < s.size(); ++k) shape[k] = s[k];
    return (int)s.size();
}

int rvm_set_param(rvm_ctx* c, const char* name, const float* data, int64_t numel) {
    if (!c || !name || !data) return fail("rvm_set_param: null argument");
    for (auto& s : c->specs) {
        if (s.name != name) continue;
        int64_t n = 1;
        for (auto d : s.shape) n *= d;
        if (n != numel) return fail(std::string("rvm_set_param: size mismatch for ") + name);
        c->params[name].assign(data, data + numel);
        c->finalized = false;
        return 0;
    }
    return fail(std::string("rvm_set_param: unexpected key ") + name);
}

int rvm_finalize(rvm_ctx* c) {
    if (!c) return fail("rvm_finalize: null ctx");
    DEVGUARD(c->device);
    return finalize_impl(c);
}

int rvm_plan(rvm_ctx* c, int B, int H, int W, double ds) {
    if (!c) return fail("rvm_plan: null ctx");
    DEVGUARD(c->device);
    return plan_impl(c, B, 1, H, W, ds);
}

int rvm_plan_seq(rvm_ctx* c, int B, int T, int H, int W, double ds) {
    if (!c) return fail("rvm_plan_seq: null ctx");
    DEVGUARD(c->device);
    return plan_impl(c, B, T, H, W, ds);
}

int rvm_state_shape(const rvm_ctx* c, int level, int64_t shape[4]) {
    if (!c || !c->planned) return fail("rvm_state_shape: no plan");
    if (level < 1 || level > 4) return fail("rvm_state_shape: level must be 1..4");
    const int ch[4] = {c->dec[2] / 2, c->dec[1] / 2, c->dec[0] / 2, c->feat[3] / 2};
    shape[0] = c->B; shape[1] = ch[level - 1]; shape[2] = c->lh[level]; shape[3] = c->lw[level];
    return 0;
}

int rvm_trim(rvm_ctx* c) {
    if (!c) return fail("rvm_trim: null ctx");
    DEVGUARD(c->device);
    if (c->parked.empty()) return 0;
    HIPCHK(hipDeviceSynchronize());                       // a parked plan's graphs / arena may still be in flight
    c->free_parked();
    return 0;
}

int64_t rvm_memory_bytes(const rvm_ctx* c, int64_t* weights, int64_t* active_plan, int64_t* parked_plans) {
    if (!c) return -1;
    const int64_t w = c->wdev ? (int64_t)c->blob.size() : 0, a = c->arena ? (int64_t)c->arena_bytes : 0;
    int64_t p = 0;
    for (const auto& k : c->parked) p += k.arena ? (int64_t)k.arena_bytes : 0;
    if (weights) *weights = w;
    if (active_plan) *active_plan = a;
    if (parked_plans) *parked_plans = p;
    return w + a + p;
}

int rvm_launch_count(const rvm_ctx* c) { return c ? (int)c->launches.size() : 0; }
int rvm_launch_kernels(const rvm_ctx* c, int i) { return (c && i >= 0 && i < (int)c->launches.size()) ? c->launches[i].nk : 0; }

int rvm_set_graph_mode(rvm_ctx* c, int enabled) {
    if (!c) return fail("rvm_set_graph_mode: null ctx");
    DEVGUARD(c->device);
    c->graph_mode = enabled != 0;
    if (!c->graph_mode) c->drop_graphs();
    return 0;
}

int rvm_set_frame_pipeline(rvm_ctx* c, int enabled) {
    if (!c) return fail("rvm_set_frame_pipeline: null ctx");
    DEVGUARD(c->device);
    const bool on = enabled != 0;
    if (on == c->pipe_req) return 0;
    // the arena layout differs between the two kinds of plan (the plan cache tells them apart): frames in flight finish
    // here, and the next rvm_plan / rvm_plan_seq builds - or finds - a plan of the kind asked for
    HIPCHK(hipDeviceSynchronize());
    c->pipe_req = on;
    for (bool& v : c->ev_dec_valid) v = false;
    return 0;
}
int rvm_frame_pipeline_src_event(rvm_ctx* c, void* hip_event) {
    if (!c) return fail("rvm_frame_pipeline_src_event: null ctx");
    c->src_event = (hipEvent_t)hip_event;
    return 0;
}

static int forward_impl(rvm_ctx* c, FrameIO& io, const void* const r_in[4], void* const r_out[4], void* hip_stream);

int rvm_forward(rvm_ctx* c, const void* src, const void* const r_in[4], void* fgr, void* pha,
                void* const r_out[4], int seg_pass, void* hip_stream) {
    if (!c || !c->planned) return fail("rvm_forward: no plan (call rvm_plan)");
    if (!src || !pha || !r_out || (!seg_pass && !fgr)) return fail("rvm_forward: null buffer");
    FrameIO io; memset(&io, 0, sizeof io);
    io.src = src; io.fgr = fgr; io.pha = pha; io.seg = seg_pass ? 1 : 0;
    return forward_impl(c, io, r_in, r_out, hip_stream);
}

int rvm_forward_u8(rvm_ctx* c, const void* src, const void* const r_in[4], uint8_t* out, void* const r_out[4],
                   int mode, const float bgr[3], void* hip_stream) {
    if (!c || !c->planned) return fail("rvm_forward_u8: no plan (call rvm_plan)");
    if (!src || !out || !r_out) return fail("rvm_forward_u8: null buffer");
    if (mode < 0 || mode > 3) return fail("rvm_forward_u8: bad mode");
    if (mode == RVM_COMPOSE_GREEN && !bgr) return fail("rvm_forward_u8: background colour missing");
    if (c->ds == 1.0) return fail("rvm_forward_u8: the fused uint8 output belongs to the refiner's full-resolution pass; with "
                                  "downsample_ratio == 1 use rvm_forward + rvm_compose_u8");
    FrameIO io; memset(&io, 0, sizeof io);
    io.src = src; io.u8 = out; io.u8mode = mode;
    for (int k = 0; k < 3; ++k) io.bgr[k] = bgr ? bgr[k] : 0.f;
    return forward_impl(c, io, r_in, r_out, hip_stream);
}

// Streams for the encoder stages of the frame pipeline.  A stream is bound to one of the device's hardware queues
// (GPU_MAX_HW_QUEUES, 4 here) in the order the process creates its streams, and two streams on one queue take turns: measured
// on the HD frame, two stages 4750 frames/s on queues of their own and 3005 when one of them shares the caller's queue; three
// stages 5100-5380 against 3650-4000.  Which queue a new stream gets depends on how many streams the process made before, so
// it is found out: candidates are created one by one and each is kept only if a probe kernel on it overlaps in time with
// probes on the caller's stream and on every stream kept so far.  ~1 ms; the probes on the caller's stream are ordinary
// stream-ordered work.  If fewer than `need` independent queues turn up, the stages share the streams found.
// A caller's stream the kept ones were not probed against (the caller went over to another stream) gets the same treatment:
// the kept streams that share ITS queue are given up (the device is idle at that point) and replaced; measured with the
// module moved from the default stream to a torch.cuda.Stream(): 273 us per HD frame before this, against 197.  At most
// eight such rounds per context (caller's streams in turn, more streams than queues: there is no arrangement to find);
// a change of the caller's priority is followed every time.
static int pick_stage_streams(rvm_ctx* c, hipStream_t st, int need) {
    const bool seen = std::find(c->probed_streams.begin(), c->probed_streams.end(), st) != c->probed_streams.end();
    if (c->enc_streams[need - 1] && seen) return 0;          // (every frame comes through here)
    // The stage streams take the priority of the caller's stream: the decoder phase on a high-priority stream waiting for
    // stages on normal-priority ones hung the device-side queues (a caller on torch.cuda.Stream(priority=-1) after three
    // callers on ordinary streams: the frames never completed; tools/pipe_stream_switch.py).
    int prio = 0;
    if (hipStreamGetPriority(st, &prio) != hipSuccess) { prio = 0; (void)hipGetLastError(); }
    if (c->enc_streams[need - 1] && c->stream_picks >= 8 && prio == c->stream_prio) return 0;
    const bool probe = !(getenv("RVM_PIPE_PROBE") && atoi(getenv("RVM_PIPE_PROBE")) == 0);
    std::vector<hipStream_t> own;                            // distinct streams of the context's own, in stage order
    for (int j = 0; j < 4; ++j)
        if (c->enc_streams[j] && std::find(own.begin(), own.end(), c->enc_streams[j]) == own.end()) own.push_back(c->enc_streams[j]);
    const size_t had = own.size();
    if (const char* e = getenv("RVM_PIPE_DUMMY")) {          // measurement hook: n streams made (and used) first, as an application might have
        static std::vector<hipStream_t> dummies;
        for (int k = (int)dummies.size(); k < atoi(e) && k < 8; ++k) {
            hipStream_t d = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
            HIPCHK(hipMemsetAsync(c->arena, 0, 4, d));
            HIPCHK(hipStreamSynchronize(d));
            dummies.push_back(d);
        }
    }
    unsigned long long* ts = nullptr;
    HIPCHK(hipMalloc((void**)&ts, 16 * sizeof(unsigned long long)));
    const unsigned ticks = 4000;                             // 40 us
    std::vector<hipStream_t> rejected;
    unsigned long long h[16];
    // one round of probes: the caller's stream (slot 0), the kept streams, the candidate if there is one (last slot)
    auto round = [&](hipStream_t cand) -> int {
        HIPCHK(hipDeviceSynchronize());
        launch_queue_probe(ts, ticks, st);
        for (size_t j = 0; j < own.size(); ++j) launch_queue_probe(ts + 2 * (j + 1), ticks, own[j]);
        if (cand) launch_queue_probe(ts + 2 * (own.size() + 1), ticks, cand);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(h, ts, sizeof h, hipMemcpyDeviceToHost));
        return 0;
    };
    auto overlap = [&](size_t x, size_t y) {                 // by at least a quarter of a probe's length?
        const unsigned long long b = std::max(h[2 * x], h[2 * y]), e = std::min(h[2 * x + 1], h[2 * y + 1]);
        return e >= b + ticks / 4;
    };
    if (!own.empty() && prio != c->stream_prio) {           // all of them are replaced
        HIPCHK(hipDeviceSynchronize());
        rejected.swap(own);
    }
    c->stream_prio = prio;
    if (probe && !own.empty() && !seen) {                    // a new caller's stream: which of the kept ones sit on its queue?
        if (round(nullptr)) { (void)hipFree(ts); return -1; }
        std::vector<hipStream_t> keep;
        for (size_t j = 0; j < own.size(); ++j) (overlap(0, j + 1) ? keep : rejected).push_back(own[j]);
        own.swap(keep);
        if (getenv("RVM_VERBOSE")) fprintf(stderr, "[rvm] frame pipeline: caller's stream %p: %zu of %zu stage streams share its queue\n", (void*)st, had - own.size(), had);
    }
    const size_t dropped = had - own.size();
    for (int tries = 0; (int)own.size() < need && tries < 8; ++tries) {
        hipStream_t cand = nullptr;
        HIPCHK(hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, prio));
        bool ok = true;
        if (probe) {
            if (round(cand)) { (void)hipFree(ts); return -1; }
            for (size_t j = 0; j <= own.size(); ++j) ok &= overlap(own.size() + 1, j);
        }
        if (ok) own.push_back(cand); else rejected.push_back(cand);
        if (getenv("RVM_VERBOSE")) fprintf(stderr, "[rvm] frame pipeline: candidate %d %s\n", tries, ok ? "kept" : "shares a queue");
    }
    (void)hipFree(ts);
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    if (getenv("RVM_VERBOSE")) fprintf(stderr, "[rvm] frame pipeline: %zu streams given up\n", rejected.size());
    if (getenv("RVM_VERBOSE")) fprintf(stderr, "[rvm] frame pipeline: %zu of %d stage streams on queues of their own (%zu streams shared a queue)\n", own.size(), need, rejected.size());
    if (own.empty()) { hipStream_t s0 = nullptr; HIPCHK(hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, prio)); own.push_back(s0); }
    for (int j = 0; j < 4; ++j) c->enc_streams[j] = j < need ? own[j % own.size()] : ((size_t)j < own.size() ? own[j] : nullptr);   // shared: those stages take turns
    if (!seen) {
        // the caller's streams probed before stay good only if every stream kept then is still in use
        if (dropped) c->probed_streams.clear();
        if (c->probed_streams.size() >= 16) c->probed_streams.erase(c->probed_streams.begin());
        c->probed_streams.push_back(st);
        ++c->stream_picks;
    }
    return 0;
}

static int forward_impl(rvm_ctx* c, FrameIO& io, const void* const r_in[4], void* const r_out[4], void* hip_stream) {
    for (int i = 0; i < 4; ++i) {
        io.r_in[i] = r_in ? r_in[i] : nullptr;
        io.r_out[i] = r_out[i];
        if (!io.r_out[i]) return fail("rvm_forward: null r_out");
        if (io.r_out[i] == io.r_in[i]) return fail("rvm_forward: r_out must not alias r_in");
    }
    hipStream_t st = (hipStream_t)hip_stream;
    DEVGUARD(c->device);                                     // launches, capture stream and graphs belong to the context's device
    hipEvent_t src_ev = c->src_event;
    c->src_event = nullptr;                                  // one-shot
    if (c->pipe && !c->launch_sets.empty()) {
        // Frame pipeline: encoder stage 0 waits for the decoder phase that last read this buffer set (sets = stages in
        // flight) and for src's event if one was given; every later stage waits for the one before it on this frame; the
        // decoder phase, on the caller's stream, for the last encoder stage.
        const int NE = c->enc_stages, S = NE + 1, p = c->frame_set % S;      // (% S: the plan before this one may have had more sets)
        auto& L = p ? c->launch_sets[p - 1] : c->launches;
        { const int rc = pick_stage_streams(c, st, NE); if (rc) return rc; }
        if (!c->pipe_sync) {
            // (stream priorities make no difference: encoder streams at the lowest / highest priority measured 211.7 / 209.9 us
            //  per HD frame against 210.0)
            // device-scope release is all the ordering needs (the default, release to system, also writes the L2 back)
            for (auto& row : c->ev_stage) for (auto& e : row) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice));
            // How the streams order each other.  Default: events.  RVM_PIPE_SYNC=kernel: a one-thread kernel publishes the
            // frame number behind a stage, a one-wave kernel polls for it in front of the next (sleeping between polls;
            // gives up after 2 s) - measured 304 against 310 us per HD frame with one encoder stage, but it assumes that the
            // streams sit on different hardware queues, which nothing guarantees; stream memory operations
            // (hipStreamWaitValue32) measured slower than either (328 us).
            c->pipe_sync = 2;
            if (const char* a = getenv("RVM_PIPE_ADAPT")) c->pipe_adapt = atoi(a) != 0;
            const char* e = getenv("RVM_PIPE_SYNC");
            if (e && std::string(e) == "kernel") {
                for (auto& g : c->sig_stage) { HIPCHK(hipMalloc((void**)&g, 8)); HIPCHK(hipMemset(g, 0, 8)); }
                HIPCHK(hipDeviceSynchronize());
                c->pipe_sync = 3;
            }
        }
        const bool kern = c->pipe_sync == 3;
        if (kern && c->pipe_seq >= 0x7FFFFF00u) {            // (the counters compare as signed differences: start over long before they wrap)
            HIPCHK(hipDeviceSynchronize());
            for (auto& g : c->sig_stage) HIPCHK(hipMemset(g, 0, 8));
            HIPCHK(hipDeviceSynchronize());
            c->pipe_seq = 0;
        }
        const uint32_t k = ++c->pipe_seq;
        // Nothing in flight (the decoder phase of the frame before has completed: a caller that waits for every result, or
        // the first frame): the stages would run one after the other anyway, so they go to the caller's stream and skip the
        // cross-stream hand-offs (~20 us each; sync-every-frame callers measured 2040 frames/s pipelined against 2440 on one stream).
        bool idle = false;
        if (!kern && c->pipe_adapt) {
            const int q = (p + S - 1) % S;
            idle = !c->ev_dec_valid[q] || hipEventQuery(c->ev_stage[NE][q]) == hipSuccess;
            (void)hipGetLastError();                         // (hipErrorNotReady is not an error here)
        }
        int i0 = 0;
        for (int j = 0; j < NE; ++j) {
            hipStream_t es = idle ? st : c->enc_streams[j];
            if (idle) {
                if (j == 0 && src_ev) HIPCHK(hipStreamWaitEvent(es, src_ev, 0));
            } else if (j == 0) {
                if (kern) { if (k > (uint32_t)S) launch_flag_wait(c->sig_stage[NE], k - S, es); }
                else if (c->ev_dec_valid[p]) HIPCHK(hipStreamWaitEvent(es, c->ev_stage[NE][p], 0));
                if (src_ev) HIPCHK(hipStreamWaitEvent(es, src_ev, 0));
            } else {
                if (kern) launch_flag_wait(c->sig_stage[j - 1], k, es);
                else HIPCHK(hipStreamWaitEvent(es, c->ev_stage[j - 1][p], 0));
            }
            while (i0 < c->stage_end[j] && L[i0].edge == 1) L[i0++].fn(es, io);         // the launch(es) that read src
            hipGraphExec_t& ge = c->stage_exec[p][j];
            if (!ge && !idle && !c->stage_graph_off && i0 < c->stage_end[j]) {
                const char* e = getenv("RVM_PIPE_GRAPH");
                if (e && atoi(e) == 0) c->stage_graph_off = true;
                else {
                    // captured on the stage's own stream (no further stream: see above); what is already enqueued on it stays
                    // outside the capture
                    hipGraph_t graph = nullptr;
                    HIPCHK(hipStreamBeginCapture(es, hipStreamCaptureModeThreadLocal));
                    for (int i = i0; i < c->stage_end[j]; ++i) L[i].fn(es, io);
                    const hipError_t launch_err = hipGetLastError();
                    const hipError_t end_err = hipStreamEndCapture(es, &graph);     // always ended: a stream left capturing poisons later calls
                    hipError_t inst_err = hipErrorUnknown;
                    if (launch_err == hipSuccess && end_err == hipSuccess && graph) inst_err = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
                    if (graph) (void)hipGraphDestroy(graph);
                    if (inst_err != hipSuccess) { ge = nullptr; c->stage_graph_off = true; (void)hipGetLastError(); }
                }
            }
            if (ge) { HIPCHK(hipGraphLaunch(ge, es)); i0 = c->stage_end[j]; }
            else for (; i0 < c->stage_end[j]; ++i0) L[i0].fn(es, io);
            if (kern) launch_flag_set(c->sig_stage[j], k, es);
            else if (!idle) HIPCHK(hipEventRecord(c->ev_stage[j][p], es));
        }
        if (idle) {
        } else if (kern) {
            launch_flag_wait(c->sig_stage[NE - 1], k, st);
            if (k > 1 && st != c->last_dec_stream) launch_flag_wait(c->sig_stage[NE], k - 1, st);      // (a caller that changes streams between frames)
        } else {
            HIPCHK(hipStreamWaitEvent(st, c->ev_stage[NE - 1][p], 0));
            const int q = (p + S - 1) % S;
            if (c->ev_dec_valid[q] && st != c->last_dec_stream) HIPCHK(hipStreamWaitEvent(st, c->ev_stage[NE][q], 0));
        }
        c->last_dec_stream = st;
        if (c->graph_mode) {
            // graph mode: the decoder phase up to the launch(es) that write fgr / pha replays as a HIP graph too, one per
            // (state pointers, buffer set) - the caller keeps the states in a ring of static buffers (see rvm_set_graph_mode)
            size_t hi = L.size();
            while (hi > (size_t)c->n_enc && L[hi - 1].edge == 2) --hi;
            FrameIO key = io; key.src = nullptr; key.fgr = nullptr; key.pha = nullptr;
            key.u8 = nullptr; key.u8mode = 0; memset(key.bgr, 0, sizeof key.bgr);
            key.seg |= (int64_t)(p + 1) << 32;
            hipGraphExec_t exec = nullptr;
            for (auto& g : c->graphs) if (g.key == key) { exec = g.exec; break; }
            if (!exec) {
                if (!c->cap_stream) HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
                for (size_t i = (size_t)c->n_enc; i < hi; ++i) L[i].fn(c->cap_stream, io);
                const hipError_t launch_err = hipGetLastError();
                const hipError_t end_err = hipStreamEndCapture(c->cap_stream, &graph);
                if (launch_err != hipSuccess || end_err != hipSuccess || !graph) {
                    if (graph) (void)hipGraphDestroy(graph);
                    (void)hipGetLastError();
                    return fail(std::string("rvm_forward: graph capture failed: ") + hipGetErrorString(launch_err != hipSuccess ? launch_err : end_err));
                }
                const hipError_t inst_err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (inst_err != hipSuccess) return fail(std::string("rvm_forward: hipGraphInstantiate: ") + hipGetErrorString(inst_err));
                if (c->graphs.size() >= 16) {
                    const hipError_t sync_err = hipDeviceSynchronize();      // the evicted exec may still be running
                    if (sync_err != hipSuccess) { (void)hipGraphExecDestroy(exec); return fail(std::string("rvm_forward: ") + hipGetErrorString(sync_err)); }
                    (void)hipGraphExecDestroy(c->graphs.front().exec);
                    c->graphs.erase(c->graphs.begin());
                }
                c->graphs.push_back({key, exec});
            }
            HIPCHK(hipGraphLaunch(exec, st));
            for (size_t i = hi; i < L.size(); ++i) L[i].fn(st, io);
        } else
        for (size_t i = (size_t)c->n_enc; i < L.size(); ++i) L[i].fn(st, io);
        if (kern) launch_flag_set(c->sig_stage[NE], k, st);
        else { HIPCHK(hipEventRecord(c->ev_stage[NE][p], st)); c->ev_dec_valid[p] = true; }
        c->last_set = p; c->frame_set = (p + 1) % S;
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (src_ev) HIPCHK(hipStreamWaitEvent(st, src_ev, 0));
    c->last_set = 0;
    if (!c->graph_mode) {
        for (auto& f : c->launches) f.fn(st, io);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // Graph mode: the launches that touch src / fgr / pha (the first and the last of the schedule) run eagerly around a
    // HIP graph of everything in between, so a replay only needs the STATE pointers to repeat (the caller's ring of
    // state buffers) - a driver may hand in a new src tensor and take fresh fgr / pha tensors every call.
    const size_t nl = c->launches.size();
    size_t lo = 0, hi = nl;
    while (lo < nl && c->launches[lo].edge == 1) ++lo;
    while (hi > lo && c->launches[hi - 1].edge == 2) --hi;
    for (size_t i = 0; i < lo; ++i) c->launches[i].fn(st, io);
    FrameIO key = io; key.src = nullptr; key.fgr = nullptr; key.pha = nullptr;
    key.u8 = nullptr; key.u8mode = 0; memset(key.bgr, 0, sizeof key.bgr);          // only the last (eager) launch sees these
    hipGraphExec_t exec = nullptr;
    for (auto& g : c->graphs)
        if (g.key == key) { exec = g.exec; break; }
    if (!exec) {
        if (!c->cap_stream) HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
        for (size_t i = lo; i < hi; ++i) c->launches[i].fn(c->cap_stream, io);
        // whatever happened in between, the capture is ended here: a stream left capturing would poison every later call
        const hipError_t launch_err = hipGetLastError();
        const hipError_t end_err = hipStreamEndCapture(c->cap_stream, &graph);
        if (launch_err != hipSuccess || end_err != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            return fail(std::string("rvm_forward: graph capture failed: ") + hipGetErrorString(launch_err != hipSuccess ? launch_err : end_err));
        }
        const hipError_t inst_err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (inst_err != hipSuccess) return fail(std::string("rvm_forward: hipGraphInstantiate: ") + hipGetErrorString(inst_err));
        if (c->graphs.size() >= 8) {
            // the evicted exec may still be running on whichever stream it was last launched on
            const hipError_t sync_err = hipDeviceSynchronize();
            if (sync_err != hipSuccess) { (void)hipGraphExecDestroy(exec); return fail(std::string("rvm_forward: ") + hipGetErrorString(sync_err)); }
            (void)hipGraphExecDestroy(c->graphs.front().exec);
            c->graphs.erase(c->graphs.begin());
        }
        c->graphs.push_back({key, exec});
    }
    HIPCHK(hipGraphLaunch(exec, st));
    for (size_t i = hi; i < nl; ++i) c->launches[i].fn(st, io);
    HIPCHK(hipGetLastError());
    return 0;
}

int rvm_launch_info(const rvm_ctx* c, int i, char* name, int name_cap, double* flops, double* bytes) {
    if (!c || i < 0 || i >= (int)c->launches.size()) return fail("rvm_launch_info: bad index");
    const auto& L = c->launches[i];
    if (name && name_cap > 0) { strncpy(name, L.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (flops) *flops = L.flops;
    if (bytes) *bytes = L.bytes;
    return 0;
}

// Eager run of one frame with a hipEvent pair around every launch on `hip_stream`; ms_out[i] is
// the device time of launch i.  Synchronises.  Used by bench.py for the live roofline figure.
int rvm_profile_frame(rvm_ctx* c, const void* src, const void* const r_in[4], void* fgr, void* pha,
                      void* const r_out[4], void* hip_stream, float* ms_out, int cap) {
    if (!c || !c->planned) return fail("rvm_profile_frame: no plan");
    const int n = (int)c->launches.size();
    if (!ms_out || cap < n) return fail("rvm_profile_frame: ms_out too small");
    DEVGUARD(c->device);
    FrameIO io; memset(&io, 0, sizeof io);
    io.src = src; io.fgr = fgr; io.pha = pha; io.seg = 0;
    for (int i = 0; i < 4; ++i) { io.r_in[i] = r_in ? r_in[i] : nullptr; io.r_out[i] = r_out[i]; }
    hipStream_t st = (hipStream_t)hip_stream;
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventRecord(ev[0], st));
    for (int i = 0; i < n; ++i) { c->launches[i].fn(st, io); HIPCHK(hipEventRecord(ev[i + 1], st)); }
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) HIPCHK(hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]));
    for (auto& e : ev) (void)hipEventDestroy(e);
    return 0;
}

int rvm_compose_u8(int dtype, int mode, const void* fgr, const void* pha, uint8_t* out, int N, int H, int W,
                   const float bgr[3], void* hip_stream) {
    if (!pha || !out || (mode != RVM_COMPOSE_ALPHA && !fgr)) return fail("rvm_compose_u8: null buffer");
    if (mode < 0 || mode > 3) return fail("rvm_compose_u8: bad mode");
    if (N < 1 || H < 1 || W < 1) return fail("rvm_compose_u8: bad shape");
    if (dtype != RVM_DTYPE_F32 && dtype != RVM_DTYPE_F16 && dtype != RVM_DTYPE_F16IO && dtype != RVM_DTYPE_BF16IO) return fail("rvm_compose_u8: bad dtype");
    if (mode == RVM_COMPOSE_GREEN && !bgr) return fail("rvm_compose_u8: background colour missing");
    ComposeArgs a; memset(&a, 0, sizeof a);
    a.fgr = fgr; a.pha = pha; a.out = out; a.H = H; a.W = W; a.mode = mode;
    for (int k = 0; k < 3; ++k) a.bgr[k] = bgr ? bgr[k] : 0.f;
    launch_compose(a, N, dtype == RVM_DTYPE_F32 ? IO_SAME : (dtype == RVM_DTYPE_BF16IO ? IO_BF16 : IO_F16), (hipStream_t)hip_stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int64_t rvm_metrics_workspace(int N, int H, int W) {
    if (N < 1 || H < 1 || W < 1) return -1;
    return (int64_t)N * cdiv(W, METRIC_TILE) * cdiv(H, METRIC_TILE) * METRIC_NSUM * (int64_t)sizeof(float);
}

static int metrics_impl(const float* pred_pha, const float* true_pha, const float* pred_prev, const float* true_prev,
                        const float* pred_fgr, const float* true_fgr, int N, int H, int W, double sigma,
                        void* workspace, double* sums, void* hip_stream) {
    if (!pred_pha || !true_pha || !workspace || !sums) return fail("rvm_metrics: null buffer");
    if (N < 1 || N > 65535 || H < 1 || W < 1) return fail("rvm_metrics: bad shape");
    if ((pred_prev == nullptr) != (true_prev == nullptr)) return fail("rvm_metrics: previous frames come in pairs");
    if ((pred_fgr == nullptr) != (true_fgr == nullptr)) return fail("rvm_metrics: foregrounds come in pairs");
    // Gaussian-derivative filter of evaluate_hr.py:203-227 (sigma 1.4 -> 9 x 9), factored: K[i][j] = g[i] * dg[j]
    const double eps = 1e-2;
    const int half = (int)std::ceil(sigma * std::sqrt(-2.0 * std::log(std::sqrt(2.0 * M_PI) * sigma * eps)));
    if (half != 4) return fail("rvm_metrics: the kernel is built for the 9 x 9 filter (sigma 1.4 of the reference)");
    double g[9], dg[9], sg = 0.0, sd = 0.0;
    for (int i = 0; i < 9; ++i) {
        const double x = i - half;
        g[i] = std::exp(-x * x / (2 * sigma * sigma)) / (sigma * std::sqrt(2.0 * M_PI));
        dg[i] = -x * g[i] / (sigma * sigma);
        sg += g[i] * g[i]; sd += dg[i] * dg[i];
    }
    const double norm = std::sqrt(sg * sd);                 // sqrt(sum K^2), :215
    MetricArgs a; memset(&a, 0, sizeof a);
    for (int i = 0; i < 9; ++i) { a.g[i] = (float)(g[i] / std::sqrt(norm)); a.dg[i] = (float)(dg[i] / std::sqrt(norm)); }
    a.pred = pred_pha; a.tru = true_pha; a.pred_prev = pred_prev; a.tru_prev = true_prev;
    a.pred_fgr = pred_fgr; a.tru_fgr = true_fgr; a.partial = (float*)workspace; a.out = sums;
    a.H = H; a.W = W; a.tiles_x = cdiv(W, METRIC_TILE); a.tiles_y = cdiv(H, METRIC_TILE);
    if (a.tiles_y > 65535) return fail("rvm_metrics: frame too tall");
    launch_metrics(a, N, (hipStream_t)hip_stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int rvm_metrics(const float* pred_pha, const float* true_pha, const float* pred_prev, const float* true_prev,
                const float* pred_fgr, const float* true_fgr, int N, int H, int W, double sigma,
                void* workspace, double* sums, void* hip_stream) {
    return metrics_impl(pred_pha, true_pha, pred_prev, true_prev, pred_fgr, true_fgr, N, H, W, sigma, workspace, sums, hip_stream);
}

// Allocation-free variant: same kernels, the per-tile partial sums go to the library's static device array
// (kernels.hip `scratch`) instead of a caller workspace.  No hipMalloc / hipFree / allocating routine on this path.
int64_t rvm_metrics_static_capacity(void) { return (int64_t)METRIC_SCRATCH_BYTES; }
int rvm_metrics_static(const float* pred_pha, const float* true_pha, const float* pred_prev, const float* true_prev,
                       const float* pred_fgr, const float* true_fgr, int N, int H, int W, double sigma,
                       double* sums, void* hip_stream) {
    const int64_t need = rvm_metrics_workspace(N, H, W);
    if (need < 0) return fail("rvm_metrics_static: bad shape");
    if (need > (int64_t)METRIC_SCRATCH_BYTES)
        return fail("rvm_metrics_static: N x tiles needs " + std::to_string(need) + " bytes of scratch, the static array holds " +
                    std::to_string(METRIC_SCRATCH_BYTES) + " (use rvm_metrics with a workspace, or fewer frames per call)");
    float* ws = nullptr;
    const int rc = metrics_scratch(&ws);
    if (rc) return fail(rc == 1 ? "rvm_metrics_static: no usable current HIP device" : "rvm_metrics_static: hipGetSymbolAddress(scratch) failed");
    return metrics_impl(pred_pha, true_pha, pred_prev, true_prev, pred_fgr, true_fgr, N, H, W, sigma, ws, sums, hip_stream);
}

int rvm_debug_stamps(uint64_t* out, int capacity, int* nlaunch) {
    if (!out || capacity < 2 * 512 * 16 || !nlaunch) return fail("rvm_debug_stamps: need room for 2*512*16 values");
    unsigned int n[2] = {0, 0};
    if (stamps_read((unsigned long long*)out, n)) return fail("rvm_debug_stamps: copy failed");
    nlaunch[0] = (int)n[0]; nlaunch[1] = (int)n[1];
    return 0;
}

int rvm_debug_tap(rvm_ctx* c, const char* name, float* out, int64_t cap, int64_t shape[4]) {
    if (!c || !c->planned || !name) return fail("rvm_debug_tap: no plan");
    auto& taps = (c->last_set > 0 && c->last_set <= (int)c->tap_sets.size()) ? c->tap_sets[c->last_set - 1] : c->taps;     // the buffer set of the last forward
    auto it = taps.find(name);
    if (it == taps.end()) return fail(std::string("rvm_debug_tap: unknown tap ") + name);
    const Tap& t = it->second;
    shape[0] = t.B; shape[1] = t.C; shape[2] = t.H; shape[3] = t.W;
    const int64_t n = (int64_t)t.B * t.C * t.H * t.W;
    if (!out) return 0;
    if (cap < n) return fail("rvm_debug_tap: output too small");
    DEVGUARD(c->device);
    HIPCHK(hipDeviceSynchronize());
    if (t.fill) { t.fill(nullptr); HIPCHK(hipDeviceSynchronize()); }
    const size_t es = t.is_float ? 4 : c->esize();
    const size_t pixels = (size_t)t.B * t.H * t.W;
    std::vector<char> raw(pixels * t.Cs * es);
    HIPCHK(hipMemcpy(raw.data(), t.p, raw.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < t.B; ++b)
        for (int y = 0; y < t.H; ++y)
            for (int x = 0; x < t.W; ++x) {
                const size_t pix = ((size_t)b * t.H + y) * t.W + x;
                for (int ch = 0; ch < t.C; ++ch) {
                    float v = es == 4 ? ((const float*)raw.data())[pix * t.Cs + ch] : (float)((const f16*)raw.data())[pix * t.Cs + ch];
                    out[(((size_t)b * t.C + ch) * t.H + y) * t.W + x] = v;
                }
            }
    return 0;
}

}  // extern "C"
