/* rvm_mi355x.h - C ABI of librvm_mi355x.so: RVM per-frame matting inference on AMD MI355X (gfx950).
 *
 * The reference (caixin98/RobustVideoMatting) has no FFI / plugin interface: its hot path is the
 * Python method MattingNetwork.forward (model/model.py:40-71).  This header is the boundary a
 * binding for that path would use; the Python class robustvideomatting_amd.MattingNetwork binds it
 * with ctypes.  Plain pointers and sizes only - no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; rvm_last_error() gives the text
 *     (thread-local).  No exception or abort crosses this boundary.
 *   - src / fgr / pha are dense NCHW device buffers of the context's dtype.
 *   - recurrent states r1..r4 are device buffers with LOGICAL shape [B, C, h, w] stored
 *     channels-last (memory order B, h, w, C; torch.channels_last).  rvm_state_shape gives B,C,h,w.
 *   - all work is enqueued on the given hipStream_t (passed as void*); no host sync inside
 *     rvm_forward.  A context is not thread-safe; distinct contexts may be driven concurrently.
 */
#ifndef RVM_MI355X_H
#define RVM_MI355X_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rvm_ctx rvm_ctx;

/* RVM_DTYPE_F32   : fp32 tensors, fp32 arithmetic (v_mfma_f32_16x16x4_f32)
 * RVM_DTYPE_F16   : fp16 tensors, fp16 storage and v_mfma_f32_16x16x32_f16 inside (fastest, fp16 accuracy)
 * RVM_DTYPE_F16IO : fp16 tensors at the boundary (src, fgr, pha, r1..r4), fp32 storage and arithmetic
 *                   inside - the mode that meets the 2e-3 fp16 parity bound on any weights
 * RVM_DTYPE_BF16IO: bf16 tensors at the boundary, the same fp32-accurate interior as F16IO (decoder.py:184-185: states and
 *                   outputs follow src.dtype for any dtype)
 * In the two 16-bit I/O modes a context also keeps the fp32 value of every recurrent state it hands out and uses it, element
 * by element, wherever the state that comes back rounds to exactly that value (i.e. the caller returned our state, the normal
 * recurrence, inference.py:129) - the state is then never re-rounded to 16 bits from frame to frame (no drift against the fp32
 * reference over long clips); a foreign or edited state is taken at its 16-bit value.  One context = one stream (or one batch
 * of streams): interleaving unrelated streams through one context is correct but falls back to 16-bit state precision. */
enum { RVM_DTYPE_F32 = 0, RVM_DTYPE_F16 = 1, RVM_DTYPE_F16IO = 2, RVM_DTYPE_BF16IO = 3 };
enum { RVM_VARIANT_MOBILENETV3 = 0, RVM_VARIANT_RESNET50 = 1 };
enum { RVM_REFINER_DEEP_GUIDED_FILTER = 0, RVM_REFINER_FAST_GUIDED_FILTER = 1 };

/* MattingNetwork.__init__(variant, refiner)  - model/model.py:15-38.  `device` is the HIP ordinal. */
int rvm_create(int variant, int refiner, int dtype, int device, rvm_ctx** out);
void rvm_destroy(rvm_ctx* ctx);

/* The state_dict contract (inference.py:165 load_state_dict): names and shapes are exactly the
 * reference's keys.  num_batches_tracked entries are listed with ndim 0 and may be
 * left unset. */
int rvm_param_count(const rvm_ctx* ctx);
const char* rvm_param_name(const rvm_ctx* ctx, int index);
int rvm_param_shape(const rvm_ctx* ctx, int index, int64_t shape[4]);   /* returns ndim, -1 on error */
/* Copy one fp32 HOST tensor (contiguous, PyTorch layout) into the context. */
int rvm_set_param(rvm_ctx* ctx, const char* name, const float* host_data, int64_t numel);
/* Fold BatchNorm (eval), pack MFMA weight fragments in the context dtype, upload.  Call after all
 * parameters are set and again whenever they change (module.to()/half()/load_state_dict). */
int rvm_finalize(rvm_ctx* ctx);

/* Fix the problem shape: src is [B,3,H,W]; downsample_ratio as in forward().  Allocates the
 * activation arena (hipMalloc) - call outside any timed / captured region.  Re-planning is allowed. */
int rvm_plan(rvm_ctx* ctx, int B, int H, int W, double downsample_ratio);
/* The same for a chunk of a clip: src [B,T,3,H,W] (model/model.py:49-51 with 5-D input; what
 * convert_video sends with seq_chunk = T, inference.py:127-129).  Like the reference's time-series path
 * (decoder.py:92-105,134-143: convs on the flattened B*T frames; decoder.py:174-180: the ConvGRU loops
 * over t), every stage outside the ConvGRUs runs once on all B*T frames and only the GRU launches are
 * repeated per time step.  rvm_forward then takes src / fgr / pha in TIME-MAJOR order
 * ([T][B][C][H][W], frame t of stream b at index t*B + b) and r_in / r_out as [B,C,h,w]: the state
 * before frame 0 and after frame T-1.  rvm_plan(B, ...) == rvm_plan_seq(B, 1, ...). */
int rvm_plan_seq(rvm_ctx* ctx, int B, int T, int H, int W, double downsample_ratio);
/* Memory.  The arena of a plan, the weight blob and the scratch of plan-time measurement are hipMalloc'd by the library (a
 * torch-free C ABI cannot borrow torch's caching allocator), never on the frame path.  A context keeps the active plan plus up
 * to RVM_KEEP_PLANS (environment, 0..8, default 4) parked ones so that a driver alternating call shapes (convert_video's short
 * last chunk; 4-D and 5-D calls in turn) neither re-allocates nor re-measures; each holds its own arena (HD, one frame:
 * ~45 MB; 8 batched HD streams: ~0.36 GB; an HD chunk of 12: ~0.5 GB).  rvm_memory_bytes reports the device bytes a context
 * holds (total; the three parts through the optional pointers), rvm_trim frees the parked plans (synchronises the device).
 * No counterpart in the reference (PyTorch's allocator owns its activations). */
int64_t rvm_memory_bytes(const rvm_ctx* ctx, int64_t* weights, int64_t* active_plan, int64_t* parked_plans);
int rvm_trim(rvm_ctx* ctx);

/* level = 1..4 -> {B, C, h, w} of r<level> for the current plan. */
int rvm_state_shape(const rvm_ctx* ctx, int level, int64_t shape[4]);

/* MattingNetwork.forward on one frame batch (model/model.py:40-71).
 *   r_in[i]  : previous state r<i+1> or NULL (= zeros, decoder.py:183-185)
 *   r_out[i] : receives the new state; must not alias r_in[i]
 *   seg_pass : 0 -> fgr [B,3,H,W] and pha [B,1,H,W] are written
 *              1 -> `pha` receives project_seg(hid) [B,1,h,w] (model.py:69-71), fgr is ignored
 * After rvm_plan_seq with T>1, src / fgr / pha hold T*B frames in time-major order (see rvm_plan_seq). */
int rvm_forward(rvm_ctx* ctx, const void* src, const void* const r_in[4], void* fgr, void* pha,
                void* const r_out[4], int seg_pass, void* hip_stream);

/* rvm_forward with the driver's output step fused into the refiner's full-resolution pass: instead of
 * fgr / pha the last kernel writes the composed uint8 frames the reference's writers receive - convert_video's
 * composition (inference.py:116,137-141) followed by `frames.mul(255).byte()` and the NCHW->NHWC permute
 * (inference_utils.py:44) - `out` uint8 [B,H,W,C] (C = 4 for RVM_COMPOSE_RGBA, else 3; [T*B,...] time-major after
 * rvm_plan_seq).  The bytes equal rvm_compose_u8 applied to what rvm_forward would have written (fgr / pha are rounded
 * to the tensor dtype first, every op of the composition rounds like the torch ops).  Half the output bytes of the
 * refiner pass and no second pass over fgr / pha.  Needs a refining plan (downsample_ratio != 1); modes and `bgr` as for
 * rvm_compose_u8 below. */
int rvm_forward_u8(rvm_ctx* ctx, const void* src, const void* const r_in[4], uint8_t* out, void* const r_out[4],
                   int mode, const float bgr[3], void* hip_stream);

/* Output step of the reference driver, on the device: convert_video's composition
 * (inference.py:116,137-141: com = fgr*pha + bgr*(1-pha) with bgr = (120,255,155)/255 for video output;
 * cat[fgr*(pha>0), pha] for png_sequence output) fused with the writers' `frames.mul(255).byte()` and
 * NCHW->NHWC permute (inference_utils.py:44).  fgr [N,3,H,W], pha [N,1,H,W] in the given dtype
 * (RVM_DTYPE_F32: fp32 tensors, RVM_DTYPE_BF16IO: bf16, otherwise fp16) -> out uint8 [N,H,W,C], C = 4 for RGBA else 3.
 * Arithmetic is rounded to the tensor dtype op by op like the torch ops it replaces (bit-identical bytes). */
enum { RVM_COMPOSE_GREEN = 0, RVM_COMPOSE_RGBA = 1, RVM_COMPOSE_ALPHA = 2, RVM_COMPOSE_FOREGROUND = 3 };
int rvm_compose_u8(int dtype, int mode, const void* fgr, const void* pha, uint8_t* out, int N, int H, int W,
                   const float bgr[3], void* hip_stream);

/* Matting quality metrics of the reference's evaluator (evaluation/evaluate_hr.py:176-235) on device tensors,
 * one pass over a clip of N consecutive frames (fp32, values 0..1, as the evaluator feeds them: image / 255):
 *   sums[n][0] = sum |pred - true|                     MetricMAD  (:176-178)  = sums0 / (H*W) * 1e3
 *   sums[n][1] = sum (pred - true)^2                   MetricMSE  (:181-183)  = sums1 / (H*W) * 1e3
 *   sums[n][2] = sum (|grad true| - |grad pred|)^2     MetricGRAD (:186-227)  = sums2 / 1000
 *                (9 x 9 Gaussian-derivative filters for `sigma` = 1.4, cross-correlation, replicate border)
 *   sums[n][3] = sum ((pred_n - pred_n-1) - (true_n - true_n-1))^2   MetricDTSSD (:230-235) = sqrt(sums3 / (H*W)) * 1e2;
 *                frame 0 uses pred_prev / true_prev ([H,W]; both NULL: 0, the evaluator's first frame :151-152)
 *   sums[n][4], sums[n][5] = sum of squared foreground error over true_pha > 0 and the number of elements in it
 *                (fgr_mse :169-171 = sums4 / sums5 * 1e3); pred_fgr / true_fgr are [N,3,H,W] or both NULL.
 * `sums` is a DEVICE array of N*6 doubles; `workspace` a device buffer of rvm_metrics_workspace(N,H,W) bytes.
 * Asynchronous on `hip_stream`; deterministic (fixed summation order). */
int64_t rvm_metrics_workspace(int N, int H, int W);
int rvm_metrics(const float* pred_pha, const float* true_pha, const float* pred_prev, const float* true_prev,
                const float* pred_fgr, const float* true_fgr, int N, int H, int W, double sigma,
                void* workspace, double* sums, void* hip_stream);

/* rvm_metrics without any allocation and without a workspace argument.  NO COUNTERPART IN THE REFERENCE: it exists
 * for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  Same kernels and bit-identical sums as rvm_metrics; the per-tile partial sums live in a
 * file-scope device array of rvm_metrics_static_capacity() bytes (4 MiB, held by every GPU the library runs on
 * until the process ends).  Requests with rvm_metrics_workspace(N, H, W) above that capacity are rejected with an
 * error (e.g. more than 21 frames of 3840x2160 per call).  Runs on the calling thread's CURRENT device, which must
 * own all the buffers.  Calls that overlap on one GPU (two streams / threads) share the array and race: serialise
 * them or use rvm_metrics.  Calls no hipMalloc / hipFree / allocating library routine. */
int64_t rvm_metrics_static_capacity(void);
int rvm_metrics_static(const float* pred_pha, const float* true_pha, const float* pred_prev, const float* true_prev,
                       const float* pred_fgr, const float* true_fgr, int N, int H, int W, double sigma,
                       double* sums, void* hip_stream);

/* Debug / test taps: copy an intermediate activation of the LAST forward to host as fp32 NCHW.
 * Names: s0 s1 s2 s3 f1 f2 f3 f4 aspp x3 x2 x1 hid ab.  Synchronises the device. */
int rvm_debug_tap(rvm_ctx* ctx, const char* name, float* host_out, int64_t capacity, int64_t shape[4]);

/* In-kernel timeline of the last <= 512 launches, two tables of 512 x 16 values (block 0 of each launch,
 * then the last block of each launch; per row: 100 MHz wall clock at block start, then shader-clock
 * stamps at phase boundaries); nlaunch[2] = launches recorded per table.  Only the RVM_STAMPS build of the library records them;
 * the product build reports *nlaunch = 0.  Synchronises the device.  Tuning aid, tools/stamp_table.py. */
int rvm_debug_stamps(uint64_t* out, int capacity, int nlaunch[2]);

/* Number of schedule entries (launches) one rvm_forward enqueues for the current plan. */
int rvm_launch_count(const rvm_ctx* ctx);
/* Kernels launch `index` of the schedule enqueues: 1, or 2 for a split-K convolution followed by its reduce pass. */
int rvm_launch_kernels(const rvm_ctx* ctx, int index);

/* Per-launch metadata (name, algorithmic FLOPs and bytes as defined in DESIGN.md) and a profiled
 * eager run that times every launch with hipEvents on `hip_stream` (synchronises). */
int rvm_launch_info(const rvm_ctx* ctx, int index, char* name, int name_cap, double* flops, double* bytes);
int rvm_profile_frame(rvm_ctx* ctx, const void* src, const void* const r_in[4], void* fgr, void* pha,
                      void* const r_out[4], void* hip_stream, float* ms_out, int ms_capacity);

/* HIP-graph replay of the fixed per-frame schedule (pointers must stay the same between calls
 * for a replay to be valid; the graph is re-captured when they change). 0 = eager (default). */
int rvm_set_graph_mode(rvm_ctx* ctx, int enabled);

/* Frame pipeline (no counterpart in the reference, whose ATen launches all go to one stream; for callers that send frames
 * without waiting for the previous one, e.g. reference inference.py:122-141 with a device-side writer).
 * The encoder phase of a frame - downsample, backbone, LR-ASPP (model/model.py:52-58) - does not depend on the recurrent
 * state, so with the pipeline on it is cut into stages (RVM_PIPE_STAGES, default 3) that run on streams of the context's
 * own, BESIDE the decoder / refiner phase (model/model.py:59-67) of the frames before, which stays on the caller's stream:
 * stages + 1 frames are in flight.  Results are bit-identical; fgr / pha / r_out are ordered on the caller's stream as
 * always, and so must r_in be.  What changes is the contract for `src`: the first stage reads it on the context's stream,
 * so it must be complete when rvm_forward is called - or ordered behind the HIP event handed to
 * rvm_frame_pipeline_src_event just before that call (one-shot; e.g. the event of the upload that fills src).  src must
 * stay untouched until work enqueued after the call on `hip_stream` could touch it (stream order on `hip_stream` is
 * enough: the decoder phase waits for the encoder stages).  Not for callers that capture rvm_forward into a graph of their
 * own (the call enqueues on several streams).  The setting applies to plans made from then on: call
 * rvm_plan / rvm_plan_seq again after changing it (a context keeps plans of both kinds).  Costs one more set of the
 * encoder's buffers per stage in the arena (rvm_memory_bytes) and one more launch per frame in the 16-bit modes.  In graph
 * mode (rvm_set_graph_mode) the decoder phase replays as a HIP graph as well, one per (state pointers, buffer set).
 * A call that finds the frame before complete (nothing in flight) runs all of its launches on `hip_stream`.
 * The stage streams are chosen against `hip_stream` - on hardware queues of their own, with its priority - at the first
 * pipelined call on it: a call on a stream the context has not seen before synchronises the device twice (~1 ms, at most
 * eight times per context unless the priority changes).
 * 0 = off (default). */
int rvm_set_frame_pipeline(rvm_ctx* ctx, int enabled);
int rvm_frame_pipeline_src_event(rvm_ctx* ctx, void* hip_event);

const char* rvm_last_error(void);
const char* rvm_version(void);

#ifdef __cplusplus
}
#endif
#endif
