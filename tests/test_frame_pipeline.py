"""GPU: the frame pipeline (rvm_set_frame_pipeline) - encoder phase of frame n+1 on the context's own stream beside the
decoder phase of frame n - must change nothing but the timing: every output and every state bit-identical to the
one-stream schedule, frame after frame with no host synchronisation in between (which is when the phases overlap and
a missing dependency would show), in every arithmetic mode, for 4-D, chunked 5-D and uint8 calls, and across switching
the pipeline on and off on one module.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dtype, variant="mobilenetv3", refiner="deep_guided_filter"):
    from robustvideomatting_amd import MattingNetwork, synthetic
    m = MattingNetwork(variant, refiner).eval()
    m.load_state_dict(synthetic.make_params(variant, refiner, seed=3, perturb_box_filter=True))
    return m.to("cuda", dtype)


def _clip(T, H, W, dtype, B=1, seed=0):
    from robustvideomatting_amd import synthetic
    return synthetic.make_clip(T, H, W, seed=seed, B=B).to("cuda", dtype).contiguous()      # [B,T,3,H,W]


def _run(m, clip, ds, sync_every_frame):
    """Frame by frame, states carried; returns every frame's outputs (cloned on the current stream).  The frames are
    made contiguous and complete BEFORE the loop: the pipeline's contract for src."""
    frames = [clip[:, t].contiguous() for t in range(clip.shape[1])]
    torch.cuda.synchronize()
    rec = [None] * 4
    outs = []
    with torch.no_grad():
        for f in frames:
            o = m(f, None, *rec, ds)
            rec = o[2:]
            outs.append([x.clone() for x in o])
            if sync_every_frame:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    return outs


def _same(a, b):
    assert len(a) == len(b)
    for t, (fa, fb) in enumerate(zip(a, b)):
        for k, (x, y) in enumerate(zip(fa, fb)):
            assert torch.equal(x, y), "frame %d output %d differs: max abs %g" % (t, k, (x.float() - y.float()).abs().max().item())


@pytest.mark.parametrize("dtype,variant,H,W,ds,B", [
    (torch.float16, "mobilenetv3", 1080, 1920, 0.25, 1),       # BASELINE configs[1]: full-size launches really overlap
    (torch.float32, "mobilenetv3", 122, 218, 0.25, 2),
    (torch.bfloat16, "mobilenetv3", 150, 201, 0.8, 1),
    (torch.float16, "resnet50", 270, 480, 0.5, 1),             # split-K convs in both phases (a workspace each)
    (torch.float16, "mobilenetv3", 144, 256, 1.0, 2),          # no refiner
])
def test_pipelined_frames_are_bit_identical(dtype, variant, H, W, ds, B):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    T = 40
    clip = _clip(T, H, W, dtype, B=B)
    torch.cuda.synchronize()
    m = _model(dtype, variant)
    want = _run(m, clip, ds, sync_every_frame=True)
    n0 = m._ctx.launch_count()
    m.pipeline_frames(True)
    got = _run(m, clip, ds, sync_every_frame=False)
    assert m._ctx.launch_count() == n0 + (0 if dtype == torch.float32 else 1)      # the state preparation is a launch of its own
    _same(want, got)
    # off again on the same module (the plan cache holds both kinds), then on: still the same
    m.pipeline_frames(False)
    _same(want, _run(m, clip, ds, sync_every_frame=False))
    m.pipeline_frames(True)
    _same(want, _run(m, clip, ds, sync_every_frame=False))


@pytest.mark.parametrize("dtype,H,W,ds,B", [
    (torch.float16, 1080, 1920, 0.25, 1),
    (torch.float32, 122, 218, 0.25, 2),
])
def test_pipelined_graph_mode_is_bit_identical(dtype, H, W, ds, B):
    """enable_graph() with the pipeline: the decoder phase replays as one graph per (buffer set, state-ring position) beside
    the stage graphs - the same bits as eager launches on one stream, through more frames than the graph cache has entries."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    T = 40
    clip = _clip(T, H, W, dtype, B=B)
    torch.cuda.synchronize()
    m = _model(dtype)
    want = _run(m, clip, ds, sync_every_frame=True)
    m.enable_graph(True)
    m.pipeline_frames(True)
    _same(want, _run(m, clip, ds, sync_every_frame=False))
    _same(want, _run(m, clip, ds, sync_every_frame=False))         # replays only
    m.pipeline_frames(False)                                       # graph mode on one stream, same module
    _same(want, _run(m, clip, ds, sync_every_frame=False))
    m.enable_graph(False)
    m.pipeline_frames(True)
    _same(want, _run(m, clip, ds, sync_every_frame=False))


def test_pipelined_debug_taps_follow_the_buffer_set():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    clip = _clip(3, 122, 218, torch.float32)
    torch.cuda.synchronize()
    m = _model(torch.float32)
    taps = {}
    rec = [None] * 4
    with torch.no_grad():
        for t in range(3):
            rec = m(clip[:, t], None, *rec, 0.25)[2:]
            taps[t] = {k: m.debug_tap(k).clone() for k in ("s0", "f3", "aspp", "x1", "hid", "ab")}
    m.pipeline_frames(True)
    rec = [None] * 4
    with torch.no_grad():
        for t in range(3):                                      # frames alternate between the two buffer sets
            rec = m(clip[:, t], None, *rec, 0.25)[2:]
            for k, v in taps[t].items():
                assert torch.equal(m.debug_tap(k), v), (t, k)


def test_pipelined_chunks_and_uint8_frames():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from robustvideomatting_amd import native
    dtype = torch.float16
    clip = _clip(12, 270, 480, dtype, B=2)
    torch.cuda.synchronize()
    m = _model(dtype)

    srcs = [clip[:, t0:t0 + 4].contiguous() for t0 in range(0, 12, 4)]
    torch.cuda.synchronize()

    def chunks(u8):
        rec = [None] * 4
        outs = []
        with torch.no_grad():
            for src in srcs:
                if u8:
                    o = m.forward_u8(src, *rec, downsample_ratio=0.5, mode=native.COMPOSE_GREEN, bgr=(0.47, 1.0, 0.6))
                    rec = o[1:]
                else:
                    o = m(src, None, *rec, 0.5)
                    rec = o[2:]
                outs.append([x.clone() for x in o])
        torch.cuda.synchronize()
        return outs

    want, want8 = chunks(False), chunks(True)
    m.pipeline_frames(True)
    _same(want, chunks(False))
    _same(want8, chunks(True))


def test_src_event_orders_a_source_produced_on_another_stream():
    """src written by a side stream just before the call: with its event handed over the encoder phase waits for it."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dtype = torch.float16
    clip = _clip(8, 540, 960, dtype)
    torch.cuda.synchronize()
    m = _model(dtype)
    want = _run(m, clip, 0.25, sync_every_frame=True)
    m.pipeline_frames(True)
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    ring = [torch.empty_like(clip[:, 0]) for _ in range(3)]
    rec = [None] * 4
    got = []
    with torch.no_grad():
        for t in range(8):
            buf = ring[t % 3]
            with torch.cuda.stream(side):
                side.wait_stream(main)                         # the buffer's previous frame has been consumed
                buf.zero_()
                for _ in range(20):                             # keep the side stream busy so a missing wait would show
                    buf.add_(clip[:, t] * 0.05)
                buf.copy_(clip[:, t])
                ev = torch.cuda.Event()
                ev.record(side)
            m.src_ready = ev
            o = m(buf, None, *rec, 0.25)
            rec = o[2:]
            got.append([x.clone() for x in o])
    torch.cuda.synchronize()
    _same(want, got)


def test_two_shapes_interleaved_through_one_pipelined_module():
    """Two streams of different frame sizes take turns on one module: every call switches the plan (the plan cache parks the
    other one, its frames still in flight on its own arena) - and the buffer-set counter runs on across the switches."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dtype = torch.float16
    a = [f.contiguous() for f in _clip(12, 540, 960, dtype, seed=1).unbind(1)]
    b = [f.contiguous() for f in _clip(12, 270, 482, dtype, B=2, seed=2).unbind(1)]
    torch.cuda.synchronize()
    m = _model(dtype)

    def run():
        ra, rb = [None] * 4, [None] * 4
        outs = []
        with torch.no_grad():
            for fa, fb in zip(a, b):
                oa = m(fa, None, *ra, 0.25)
                ra = oa[2:]
                ob = m(fb, None, *rb, 0.5)
                rb = ob[2:]
                outs.append([x.clone() for x in oa] + [x.clone() for x in ob])
        torch.cuda.synchronize()
        return outs

    want = run()
    m.pipeline_frames(True)
    _same(want, run())
    _same(want, run())


def test_caller_goes_over_to_other_streams():
    """The caller's stream changes between clips - ordinary torch streams, then a high-priority one, then the default stream
    again: the stage streams are probed against each new stream (and take its priority; with normal-priority stage streams
    under a high-priority caller this very sequence never completed), the bits stay the same."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dtype = torch.float16
    clip = _clip(24, 1080, 1920, dtype)
    torch.cuda.synchronize()
    m = _model(dtype)
    want = _run(m, clip, 0.25, sync_every_frame=True)
    m.pipeline_frames(True)
    _same(want, _run(m, clip, 0.25, sync_every_frame=False))
    main = torch.cuda.current_stream()
    for s in [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(priority=-1)]:
        s.wait_stream(main)
        with torch.cuda.stream(s):
            got = _run(m, clip, 0.25, sync_every_frame=False)
        main.wait_stream(s)
        _same(want, got)
    _same(want, _run(m, clip, 0.25, sync_every_frame=False))


def test_frames_that_find_the_device_idle_mixed_with_pipelined_ones():
    """A frame that finds nothing in flight (the caller waited, or was slow) runs on the caller's stream without the
    cross-stream hand-offs; the next one may be pipelined beside it.  Any mix gives the same bits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import random
    import time
    dtype = torch.float16
    clip = _clip(60, 540, 960, dtype, B=2)
    torch.cuda.synchronize()
    m = _model(dtype)
    want = _run(m, clip, 0.25, sync_every_frame=True)
    m.pipeline_frames(True)
    _same(want, _run(m, clip, 0.25, sync_every_frame=True))        # every frame finds the device idle
    frames = [clip[:, t].contiguous() for t in range(clip.shape[1])]
    torch.cuda.synchronize()
    for seed in (1, 2):
        rng = random.Random(seed)
        rec = [None] * 4
        got = []
        with torch.no_grad():
            for f in frames:
                r = rng.random()
                if r < 0.15:
                    torch.cuda.synchronize()
                elif r < 0.4:
                    time.sleep(rng.uniform(1e-4, 1.5e-3))
                o = m(f, None, *rec, 0.25)
                rec = o[2:]
                got.append([x.clone() for x in o])
        torch.cuda.synchronize()
        _same(want, got)
