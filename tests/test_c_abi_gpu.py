"""GPU: the C ABI driven WITHOUT the Python module - ctypes + torch.empty buffers only, exactly the stub
INTEGRATION.md section 2 shows (rvm_create -> rvm_set_param -> rvm_finalize -> rvm_plan -> rvm_forward) - against
the committed golden fixture; the native parameter table against spec.param_shapes; device handling; two contexts
driven from two host threads.
"""
import ctypes
import os
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    lib = ctypes.CDLL(os.path.join(ROOT, "robustvideomatting_amd", "lib", "librvm_mi355x.so"))
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.rvm_create.argtypes = [ci, ci, ci, ci, ctypes.POINTER(vp)]
    lib.rvm_destroy.argtypes = [vp]
    lib.rvm_destroy.restype = None
    lib.rvm_param_count.argtypes = [vp]
    lib.rvm_param_name.argtypes = [vp, ci]
    lib.rvm_param_name.restype = ctypes.c_char_p
    lib.rvm_param_shape.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_int64)]
    lib.rvm_set_param.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_int64]
    lib.rvm_finalize.argtypes = [vp]
    lib.rvm_plan.argtypes = [vp, ci, ci, ci, ctypes.c_double]
    lib.rvm_state_shape.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_int64)]
    lib.rvm_forward.argtypes = [vp, vp, ctypes.POINTER(vp), vp, vp, ctypes.POINTER(vp), ci, vp]
    lib.rvm_last_error.restype = ctypes.c_char_p
    return lib


def _ok(L, rc):
    assert rc == 0, L.rvm_last_error()


def _native_params(L, h):
    out, shape = [], (ctypes.c_int64 * 4)()
    for i in range(L.rvm_param_count(h)):
        nd = L.rvm_param_shape(h, i, shape)
        out.append((L.rvm_param_name(h, i).decode(), tuple(shape[k] for k in range(nd))))
    return out


@pytest.mark.parametrize("variant,refiner", [("mobilenetv3", "deep_guided_filter"), ("mobilenetv3", "fast_guided_filter"),
                                             ("resnet50", "deep_guided_filter")])
def test_native_param_table_equals_python_spec(L, variant, refiner):
    """rvm_param_name / rvm_param_shape == spec.param_shapes: the state_dict contract, key by key."""
    from robustvideomatting_amd import spec
    h = ctypes.c_void_p()
    _ok(L, L.rvm_create({"mobilenetv3": 0, "resnet50": 1}[variant], {"deep_guided_filter": 0, "fast_guided_filter": 1}[refiner],
                        0, 0, ctypes.byref(h)))
    try:
        native = _native_params(L, h)
    finally:
        L.rvm_destroy(h)
    want = [(k, tuple(v)) for k, v in spec.param_shapes(variant, refiner).items()]
    assert sorted(native) == sorted(want)
    assert len(native) == len(set(n for n, _ in native))


def _make_ctx(L, params, device=0, dtype=0):
    h = ctypes.c_void_p()
    _ok(L, L.rvm_create(0, 0, dtype, device, ctypes.byref(h)))
    for name, shape in _native_params(L, h):
        if len(shape) == 0:
            continue
        t = params[name].to(torch.float32).contiguous()
        assert tuple(t.shape) == shape
        _ok(L, L.rvm_set_param(h, name.encode(), ctypes.c_void_p(t.data_ptr()), t.numel()))
    _ok(L, L.rvm_finalize(h))
    return h


def _run_clip(L, h, clip, ds, device="cuda:0"):
    """clip [T,3,H,W] fp32 on `device`, one stream, frame by frame through rvm_forward; returns pha [T,1,H,W],
    the last fgr and the final states (logical [1,C,h,w], stored channels-last)."""
    T, _, H, W = clip.shape
    _ok(L, L.rvm_plan(h, 1, H, W, ds))
    shp = (ctypes.c_int64 * 4)()
    shapes = []
    for lvl in range(1, 5):
        _ok(L, L.rvm_state_shape(h, lvl, shp))
        shapes.append(tuple(shp[k] for k in range(4)))
    VP4 = ctypes.c_void_p * 4
    fgr = torch.empty((1, 3, H, W), device=device)
    phas = torch.empty((T, 1, H, W), device=device)
    r_in = [None] * 4
    stream = ctypes.c_void_p(torch.cuda.current_stream(torch.device(device)).cuda_stream)
    for t in range(T):
        # NHWC storage of a logical [B,C,h,w] state: allocate as [B,h,w,C]
        r_out = [torch.empty((s[0], s[2], s[3], s[1]), device=device) for s in shapes]
        rin = VP4(*[ctypes.c_void_p(r.data_ptr()) if r is not None else None for r in r_in])
        rout = VP4(*[ctypes.c_void_p(r.data_ptr()) for r in r_out])
        _ok(L, L.rvm_forward(h, ctypes.c_void_p(clip[t].data_ptr()), rin, ctypes.c_void_p(fgr.data_ptr()),
                             ctypes.c_void_p(phas[t].data_ptr()), rout, 0, stream))
        r_in = r_out
    torch.cuda.synchronize(torch.device(device))
    return phas, fgr, [r.permute(0, 3, 1, 2) for r in r_in]


def _golden():
    from robustvideomatting_amd import synthetic
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_mobilenetv3_ds025.npz"))
    H, W, T, seed = (int(x) for x in g["meta"])
    params = synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=seed)
    clip = synthetic.make_clip(T, H, W, seed=seed)[0]
    return g, params, clip, float(g["ds"])


def _check_golden(g, phas, fgr, states, tol=1e-4):
    assert (phas.cpu() - torch.from_numpy(g["pha"])[0]).abs().max() < tol
    assert (fgr.cpu() - torch.from_numpy(g["fgr_last"])).abs().max() < tol
    for i, r in enumerate(states):
        assert (r.cpu() - torch.from_numpy(g["r%d" % (i + 1)])).abs().max() < tol


def test_raw_c_abi_against_golden_fixture(L):
    g, params, clip, ds = _golden()
    h = _make_ctx(L, params)
    try:
        phas, fgr, states = _run_clip(L, h, clip.cuda().contiguous(), ds)
    finally:
        L.rvm_destroy(h)
    _check_golden(g, phas, fgr, states)


def test_raw_c_abi_frame_pipeline_against_golden_fixture(L):
    """rvm_set_frame_pipeline before rvm_plan, frames sent without waiting, the first one's src ordered through
    rvm_frame_pipeline_src_event (an event recorded on the stream that produced the clip): the golden vectors, and the
    same bits as the one-stream run."""
    L.rvm_set_frame_pipeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rvm_frame_pipeline_src_event.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    g, params, clip, ds = _golden()
    h = _make_ctx(L, params)
    try:
        base = _run_clip(L, h, clip.cuda().contiguous(), ds)
        _ok(L, L.rvm_set_frame_pipeline(h, 1))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            dclip = clip.cuda().contiguous()                 # produced on a side stream; no host synchronisation
            ev = torch.cuda.Event()
            ev.record(side)
        _ok(L, L.rvm_plan(h, 1, clip.shape[2], clip.shape[3], ds))       # (the pipeline applies to plans made from now on)
        _ok(L, L.rvm_frame_pipeline_src_event(h, ctypes.c_void_p(ev.cuda_event)))
        phas, fgr, states = _run_clip(L, h, dclip, ds)
    finally:
        L.rvm_destroy(h)
    _check_golden(g, phas, fgr, states)
    assert torch.equal(phas, base[0]) and torch.equal(fgr, base[1])
    for a, b in zip(states, base[2]):
        assert torch.equal(a, b)


def test_two_contexts_on_two_host_threads_equal_single_runs(L):
    """Two real contexts on one device, each driven from its own host thread on its own HIP stream (the '8 contexts per
    GPU' alternative to batching): every stream equals its single-context run bit for bit."""
    from robustvideomatting_amd import synthetic
    g, params, clip0, ds = _golden()
    clip1 = synthetic.make_clip(clip0.shape[0], clip0.shape[2], clip0.shape[3], seed=77)[0]
    clips = [clip0.cuda().contiguous(), clip1.cuda().contiguous()]
    single = []
    for c in clips:
        h = _make_ctx(L, params)
        single.append(_run_clip(L, h, c, ds))
        L.rvm_destroy(h)
    ctxs = [_make_ctx(L, params) for _ in clips]
    results, errors = [None, None], []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):                      # a few passes so the two threads really overlap
                    results[i] = _run_clip(L, ctxs[i], clips[i], ds)
        except Exception as e:                          # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for h in ctxs:
        L.rvm_destroy(h)
    assert not errors, errors
    for (p0, f0, s0), (p1, f1, s1) in zip(single, results):
        assert torch.equal(p0, p1) and torch.equal(f0, f1)
        for a, b in zip(s0, s1):
            assert torch.equal(a, b)
    _check_golden(g, *results[0])


def test_forward_leaves_the_current_device_alone(L):
    """A context on cuda:1 driven while the thread's current device is 0: results equal the golden fixture and
    torch.cuda.current_device() is unchanged by every ABI call (needs two GPUs)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    g, params, clip, ds = _golden()
    torch.cuda.set_device(0)
    h = _make_ctx(L, params, device=1)
    assert torch.cuda.current_device() == 0
    try:
        phas, fgr, states = _run_clip(L, h, clip.to("cuda:1").contiguous(), ds, device="cuda:1")
        assert torch.cuda.current_device() == 0
    finally:
        L.rvm_destroy(h)
    assert torch.cuda.current_device() == 0
    _check_golden(g, phas, fgr, states)


def test_module_on_second_device_and_graph_mode_with_fresh_src(L):
    """MattingNetwork on cuda:1 with current device 0 (skipped on one GPU), and - on any box - graph mode fed a NEW src
    tensor every call: replays must not depend on the src / fgr / pha pointers."""
    from robustvideomatting_amd import MattingNetwork, synthetic
    g, params, clip, ds = _golden()
    dev = "cuda:1" if torch.cuda.device_count() >= 2 else "cuda:0"
    torch.cuda.set_device(0)
    m = MattingNetwork().eval()
    m.load_state_dict(params)
    m = m.to(dev)
    m.enable_graph(True)
    rec, phas, keep = [None] * 4, [], []
    with torch.no_grad():
        for t in range(clip.shape[0]):
            src = clip[t:t + 1].to(dev).clone()         # a fresh tensor per call, like convert_video's chunk.to(device)
            keep.append(src)
            fgr, pha, *rec = m(src, None, *rec, ds)
            phas.append(pha)
    assert torch.cuda.current_device() == 0
    assert len(m._ctx and [1] or []) == 1
    _check_golden(g, torch.cat(phas), fgr, rec)
