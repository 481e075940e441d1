"""ctypes binding of librvm_mi355x.so (include/rvm_mi355x.h).

There is deliberately no fallback: if the shared library is missing or a symbol
cannot be resolved, importing the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RVM_LIB_VARIANT=stamps selects the instrumented build (in-kernel timeline, tuning only)
_VARIANT = os.environ.get("RVM_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "lib", "librvm_mi355x%s.so" % ("_" + _VARIANT if _VARIANT else ""))

DTYPE_F32, DTYPE_F16, DTYPE_F16IO, DTYPE_BF16IO = 0, 1, 2, 3
VARIANTS = {"mobilenetv3": 0, "resnet50": 1}
REFINERS = {"deep_guided_filter": 0, "fast_guided_filter": 1}

# every symbol include/rvm_mi355x.h declares
SYMBOLS = [
    "rvm_create", "rvm_destroy", "rvm_param_count", "rvm_param_name", "rvm_param_shape",
    "rvm_set_param", "rvm_finalize", "rvm_plan", "rvm_plan_seq", "rvm_memory_bytes", "rvm_trim", "rvm_state_shape", "rvm_forward", "rvm_forward_u8",
    "rvm_debug_tap", "rvm_debug_stamps", "rvm_compose_u8", "rvm_metrics", "rvm_metrics_workspace", "rvm_metrics_static", "rvm_metrics_static_capacity", "rvm_launch_count", "rvm_launch_kernels", "rvm_launch_info", "rvm_profile_frame", "rvm_set_graph_mode", "rvm_set_frame_pipeline", "rvm_frame_pipeline_src_event", "rvm_last_error", "rvm_version",
]

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "librvm_mi355x.so not found at %s - build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C robustvideomatting_amd/csrc`"
            % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cpp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)
    i64p = ctypes.POINTER(ctypes.c_int64)
    L.rvm_create.argtypes = [ci, ci, ci, ci, ctypes.POINTER(vp)]
    L.rvm_create.restype = ci
    L.rvm_destroy.argtypes = [vp]
    L.rvm_destroy.restype = None
    L.rvm_param_count.argtypes = [vp]
    L.rvm_param_count.restype = ci
    L.rvm_param_name.argtypes = [vp, ci]
    L.rvm_param_name.restype = ctypes.c_char_p
    L.rvm_param_shape.argtypes = [vp, ci, i64p]
    L.rvm_param_shape.restype = ci
    L.rvm_set_param.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_int64]
    L.rvm_set_param.restype = ci
    L.rvm_finalize.argtypes = [vp]
    L.rvm_finalize.restype = ci
    L.rvm_plan.argtypes = [vp, ci, ci, ci, ctypes.c_double]
    L.rvm_plan.restype = ci
    L.rvm_plan_seq.argtypes = [vp, ci, ci, ci, ci, ctypes.c_double]
    L.rvm_plan_seq.restype = ci
    L.rvm_memory_bytes.argtypes = [vp, i64p, i64p, i64p]
    L.rvm_memory_bytes.restype = ctypes.c_int64
    L.rvm_trim.argtypes = [vp]
    L.rvm_trim.restype = ci
    L.rvm_state_shape.argtypes = [vp, ci, i64p]
    L.rvm_state_shape.restype = ci
    L.rvm_forward.argtypes = [vp, vp, cpp, vp, vp, cpp, ci, vp]
    L.rvm_forward.restype = ci
    L.rvm_forward_u8.argtypes = [vp, vp, cpp, vp, cpp, ci, ctypes.POINTER(ctypes.c_float), vp]
    L.rvm_forward_u8.restype = ci
    L.rvm_debug_tap.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_int64, i64p]
    L.rvm_debug_tap.restype = ci
    L.rvm_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ci, ctypes.POINTER(ci)]
    L.rvm_debug_stamps.restype = ci
    L.rvm_compose_u8.argtypes = [ci, ci, vp, vp, vp, ci, ci, ci, ctypes.POINTER(ctypes.c_float), vp]
    L.rvm_compose_u8.restype = ci
    fp = ctypes.c_void_p
    L.rvm_metrics_workspace.argtypes = [ci, ci, ci]
    L.rvm_metrics_workspace.restype = ctypes.c_int64
    L.rvm_metrics.argtypes = [fp, fp, fp, fp, fp, fp, ci, ci, ci, ctypes.c_double, vp, vp, vp]
    L.rvm_metrics.restype = ci
    L.rvm_metrics_static_capacity.argtypes = []
    L.rvm_metrics_static_capacity.restype = ctypes.c_int64
    L.rvm_metrics_static.argtypes = [fp, fp, fp, fp, fp, fp, ci, ci, ci, ctypes.c_double, vp, vp]
    L.rvm_metrics_static.restype = ci
    L.rvm_launch_count.argtypes = [vp]
    L.rvm_launch_count.restype = ci
    L.rvm_launch_kernels.argtypes = [vp, ci]
    L.rvm_launch_kernels.restype = ci
    L.rvm_launch_info.argtypes = [vp, ci, ctypes.c_char_p, ci, ctypes.POINTER(ctypes.c_double),
                                  ctypes.POINTER(ctypes.c_double)]
    L.rvm_launch_info.restype = ci
    L.rvm_profile_frame.argtypes = [vp, vp, cpp, vp, vp, cpp, vp, ctypes.POINTER(ctypes.c_float), ci]
    L.rvm_profile_frame.restype = ci
    L.rvm_set_graph_mode.argtypes = [vp, ci]
    L.rvm_set_graph_mode.restype = ci
    L.rvm_set_frame_pipeline.argtypes = [vp, ci]
    L.rvm_set_frame_pipeline.restype = ci
    L.rvm_frame_pipeline_src_event.argtypes = [vp, vp]
    L.rvm_frame_pipeline_src_event.restype = ci
    L.rvm_last_error.argtypes = []
    L.rvm_last_error.restype = ctypes.c_char_p
    L.rvm_version.argtypes = []
    L.rvm_version.restype = ctypes.c_char_p
    _lib = L
    return L


COMPOSE_GREEN, COMPOSE_RGBA, COMPOSE_ALPHA, COMPOSE_FOREGROUND = 0, 1, 2, 3


def compose_u8(dtype_code, mode, fgr_ptr, pha_ptr, out_ptr, N, H, W, bgr, stream):
    """rvm_compose_u8 on raw device pointers (see include/rvm_mi355x.h)."""
    c3 = (ctypes.c_float * 3)(*(bgr if bgr is not None else (0.0, 0.0, 0.0)))
    check(lib().rvm_compose_u8(dtype_code, mode, ctypes.c_void_p(fgr_ptr) if fgr_ptr else None, ctypes.c_void_p(pha_ptr),
                               ctypes.c_void_p(out_ptr), N, H, W, c3, ctypes.c_void_p(stream)), "rvm_compose_u8")


def metrics_workspace(N, H, W):
    n = lib().rvm_metrics_workspace(N, H, W)
    if n < 0:
        raise NativeError("rvm_metrics_workspace: bad shape")
    return n


def metrics(pred, true, pred_prev, true_prev, pred_fgr, true_fgr, N, H, W, sigma, workspace, sums, stream):
    """rvm_metrics on raw device pointers (0 = absent)."""
    P = lambda p: ctypes.c_void_p(p) if p else None
    check(lib().rvm_metrics(P(pred), P(true), P(pred_prev), P(true_prev), P(pred_fgr), P(true_fgr), N, H, W, float(sigma),
                            P(workspace), P(sums), P(stream)), "rvm_metrics")


def metrics_static(pred, true, pred_prev, true_prev, pred_fgr, true_fgr, N, H, W, sigma, sums, stream):
    """rvm_metrics_static: no workspace, no allocation (static device scratch); runs on the current device."""
    P = lambda p: ctypes.c_void_p(p) if p else None
    check(lib().rvm_metrics_static(P(pred), P(true), P(pred_prev), P(true_prev), P(pred_fgr), P(true_fgr), N, H, W,
                                   float(sigma), P(sums), P(stream)), "rvm_metrics_static")


def debug_stamps():
    """((n0, n1), table0, table1) from the RVM_STAMPS build: [512][16] uint64 rows of block 0 / the last
    block of the most recent launches; counts are 0 in the product build."""
    buf = (ctypes.c_uint64 * (2 * 512 * 16))()
    n = (ctypes.c_int * 2)()
    check(lib().rvm_debug_stamps(buf, 2 * 512 * 16, n), "rvm_debug_stamps")
    tabs = [[list(buf[(t * 512 + i) * 16:(t * 512 + i + 1) * 16]) for i in range(512)] for t in range(2)]
    return (n[0], n[1]), tabs[0], tabs[1]


def check(rc, what=""):
    if rc != 0:
        msg = lib().rvm_last_error()
        raise NativeError("%s failed: %s" % (what or "native call", msg.decode() if msg else "?"))


class Context:
    """Thin RAII wrapper over rvm_ctx*."""

    def __init__(self, variant, refiner, dtype_code, device_index):
        L = lib()
        self._h = ctypes.c_void_p()
        check(L.rvm_create(VARIANTS[variant], REFINERS[refiner], dtype_code, device_index,
                           ctypes.byref(self._h)), "rvm_create")
        self.plan_key = None
        self._shapes, self._shapes_key = None, ()
        self._rin, self._rout = (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().rvm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def param_specs(self):
        L = lib()
        out = []
        shape = (ctypes.c_int64 * 4)()
        for i in range(L.rvm_param_count(self._h)):
            nd = L.rvm_param_shape(self._h, i, shape)
            out.append((L.rvm_param_name(self._h, i).decode(), tuple(shape[k] for k in range(nd))))
        return out

    def set_param(self, name, host_f32):
        """host_f32: contiguous fp32 CPU torch tensor."""
        check(lib().rvm_set_param(self._h, name.encode(), ctypes.c_void_p(host_f32.data_ptr()),
                                  host_f32.numel()), "rvm_set_param(%s)" % name)

    def finalize(self):
        check(lib().rvm_finalize(self._h), "rvm_finalize")
        self.plan_key = None

    def plan(self, B, H, W, ds, T=1):
        """T > 1: a chunk of T frames per stream in one call (rvm_plan_seq; tensors time-major)."""
        key = (B, H, W, float(ds), T, getattr(self, "_pipe", False))
        if key != self.plan_key:
            check(lib().rvm_plan_seq(self._h, B, T, H, W, float(ds)), "rvm_plan_seq")
            self.plan_key = key

    def memory_bytes(self):
        """Device bytes the context holds: dict(total, weights, active_plan, parked_plans)."""
        w, a, p = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        tot = lib().rvm_memory_bytes(self._h, ctypes.byref(w), ctypes.byref(a), ctypes.byref(p))
        return dict(total=tot, weights=w.value, active_plan=a.value, parked_plans=p.value)

    def trim(self):
        """Free the plans kept for shapes not in use (rvm_trim)."""
        check(lib().rvm_trim(self._h), "rvm_trim")

    def state_shape(self, level):
        shape = (ctypes.c_int64 * 4)()
        check(lib().rvm_state_shape(self._h, level, shape), "rvm_state_shape")
        return tuple(shape[k] for k in range(4))

    def state_shapes(self):
        """Shapes of r1..r4 for the current plan (cached per plan: four ABI calls otherwise, on every frame)."""
        if self._shapes_key != self.plan_key:
            self._shapes = [self.state_shape(l) for l in (1, 2, 3, 4)]
            self._shapes_key = self.plan_key
        return self._shapes

    def forward(self, src_ptr, r_in_ptrs, fgr_ptr, pha_ptr, r_out_ptrs, seg, stream):
        rin, rout = self._rin, self._rout                   # (reused: building two ctypes arrays costs ~6 us a frame)
        for i in range(4):
            rin[i] = r_in_ptrs[i] or None
            rout[i] = r_out_ptrs[i]
        check(lib().rvm_forward(self._h, ctypes.c_void_p(src_ptr), rin,
                                ctypes.c_void_p(fgr_ptr) if fgr_ptr else None, ctypes.c_void_p(pha_ptr),
                                rout, 1 if seg else 0, ctypes.c_void_p(stream)), "rvm_forward")

    def forward_u8(self, src_ptr, r_in_ptrs, out_ptr, r_out_ptrs, mode, bgr, stream):
        rin = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(p) if p else None for p in r_in_ptrs])
        rout = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(p) for p in r_out_ptrs])
        c3 = (ctypes.c_float * 3)(*(bgr if bgr is not None else (0.0, 0.0, 0.0)))
        check(lib().rvm_forward_u8(self._h, ctypes.c_void_p(src_ptr), rin, ctypes.c_void_p(out_ptr), rout, mode, c3,
                                   ctypes.c_void_p(stream)), "rvm_forward_u8")

    def launch_count(self):
        return lib().rvm_launch_count(self._h)

    def launch_kernels(self):
        """Kernels per schedule entry (2 for a split-K convolution + its reduce pass)."""
        return [lib().rvm_launch_kernels(self._h, i) for i in range(self.launch_count())]

    def launch_info(self):
        """[(name, algorithmic_flops, algorithmic_bytes)] for the current plan."""
        out = []
        buf = ctypes.create_string_buffer(128)
        fl, by = ctypes.c_double(), ctypes.c_double()
        for i in range(self.launch_count()):
            check(lib().rvm_launch_info(self._h, i, buf, 128, ctypes.byref(fl), ctypes.byref(by)), "rvm_launch_info")
            out.append((buf.value.decode(), fl.value, by.value))
        return out

    def profile_frame(self, src_ptr, r_in_ptrs, fgr_ptr, pha_ptr, r_out_ptrs, stream):
        n = self.launch_count()
        ms = (ctypes.c_float * n)()
        rin = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(p) if p else None for p in r_in_ptrs])
        rout = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(p) for p in r_out_ptrs])
        check(lib().rvm_profile_frame(self._h, ctypes.c_void_p(src_ptr), rin, ctypes.c_void_p(fgr_ptr),
                                      ctypes.c_void_p(pha_ptr), rout, ctypes.c_void_p(stream), ms, n),
              "rvm_profile_frame")
        return list(ms)

    def set_graph_mode(self, on):
        check(lib().rvm_set_graph_mode(self._h, 1 if on else 0), "rvm_set_graph_mode")

    def set_frame_pipeline(self, on):
        """Encoder phase of frame n+1 beside the decoder phase of frame n (include/rvm_mi355x.h); plans made from now on."""
        check(lib().rvm_set_frame_pipeline(self._h, 1 if on else 0), "rvm_set_frame_pipeline")
        self._pipe = bool(on)

    def src_event(self, hip_event):
        """One-shot: the next forward's src is ordered behind this HIP event (frame pipeline only)."""
        check(lib().rvm_frame_pipeline_src_event(self._h, ctypes.c_void_p(hip_event)), "rvm_frame_pipeline_src_event")

    def debug_tap(self, name):
        import torch
        shape = (ctypes.c_int64 * 4)()
        check(lib().rvm_debug_tap(self._h, name.encode(), None, 0, shape), "rvm_debug_tap")
        out = torch.empty(tuple(shape[k] for k in range(4)), dtype=torch.float32)
        check(lib().rvm_debug_tap(self._h, name.encode(), ctypes.c_void_p(out.data_ptr()), out.numel(), shape),
              "rvm_debug_tap")
        return out
