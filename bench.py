"""Throughput bench for the RVM per-frame inference path on MI355X.

    python bench.py --gpus 1 --steps 300 --warmup 20
    python bench.py --gpus 8 --batch 8              # BASELINE configs[4]: 64 HD streams, 8 per GPU (bare form, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward of one frame batch through MattingNetwork.forward with the recurrent
state carried (inference_speed_test.py:42-47 protocol of the reference: device-resident input,
tensor throughput only).  Default workload = BASELINE.json configs[1]: rvm_mobilenetv3 fp16,
one 1920x1080 stream, downsample_ratio 0.25.  With N GPUs every rank runs its own independent
streams (weak scaling, no collective on the data path); value = all frames / max rank time.

`--gpus N` without a launcher (WORLD_SIZE unset): the parent starts N child processes, one per GPU, BEFORE it
touches the GPU itself (each child sets its device first, rendezvous on 127.0.0.1), and relays rank 0's line.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline definitions).  At N=1 the line also
carries, under "also", the other BASELINE configurations timed in the same process (4K ds=0.125 - the north_star
target -, 8 batched HD streams, resnet50 HD) and "cpu_baseline_small" (configs[0], the reference's CPU case).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, spec, streams, synthetic   # noqa: E402

WORKLOADS = {
    "hd": dict(H=1080, W=1920, ds=0.25, name="rvm_mobilenetv3 fp16 1x1920x1080 ds=0.25 (BASELINE configs[1])"),
    "4k": dict(H=2160, W=3840, ds=0.125, name="rvm_mobilenetv3 fp16 1x3840x2160 ds=0.125 (BASELINE configs[2])"),
    "hd_resnet50": dict(H=1080, W=1920, ds=0.25, variant="resnet50",
                        name="rvm_resnet50 fp16 1x1920x1080 ds=0.25 (BASELINE configs[3])"),
    "small": dict(H=288, W=512, ds=1.0, name="rvm_mobilenetv3 512x288 ds=1 (BASELINE configs[0] shape)"),
}
HBM_PEAK = 8.0e12            # B/s, MI355X_MICROARCH.md
MFMA_PEAK = {"fp16fast": 2.5e15, "fp16": 2.5e15, "bf16": 2.5e15, "fp32": 157.3e12}   # dense peak of the MFMA the mode issues (f16 / f16 split / f32)
PROFILE_DIR = "profiles/r03"
# committed rocprofv3 evidence of the same command, keyed by (workload, streams per GPU, chunk): tools/run_pmc.sh (--pmc passes)
# and tools/run_prof.sh (--kernel-trace --stats)
PROFILE_TAGS = {("hd", 1, 1): "hd", ("4k", 1, 1): "4k", ("hd_resnet50", 1, 1): "rn", ("hd", 8, 1): "hd_b8", ("hd", 1, 12): "hd_c12"}
L3_BYTES = 256 << 20         # Infinity Cache (MI355X_MICROARCH.md): the src ring is sized past it
TORCH_DTYPE = {"fp32": torch.float32, "fp16": torch.float16, "fp16fast": torch.float16, "bf16": torch.bfloat16}


def frame_algorithmic_bytes(H, W, h, w, es, variant="mobilenetv3"):
    """src read + fgr/pha write + r1..r4 read+write + weights, for the variant's state widths and
    parameter count (conv / linear weights and biases + BatchNorm affine and statistics as stored in the state_dict)."""
    hw = [(h, w)]
    for _ in range(4):
        hw.append(((hw[-1][0] + 1) // 2, (hw[-1][1] + 1) // 2))
    S = sum(c * a * b for c, (a, b) in zip(spec.state_channels(variant), hw[1:]))
    P = 0
    for name, shape in spec.param_shapes(variant, "deep_guided_filter").items():
        if name.endswith("num_batches_tracked"):
            continue
        n = 1
        for d in shape:
            n *= d
        P += n
    return es * 7 * H * W + es * 2 * S + es * P


def cpu_baseline(wl, seconds=15.0, T=2):
    """The fp32 CPU oracle (restatement of the reference's PyTorch path) timed on host cores."""
    from oracle import rvm_oracle as O
    variant = wl.get("variant", "mobilenetv3")
    params = synthetic.make_params(variant, "deep_guided_filter", seed=0)
    clip = synthetic.make_clip(T, wl["H"], wl["W"], seed=0)
    rec = [None] * 4
    t0 = time.perf_counter()
    out = O.forward(params, clip[:, 0], *rec, downsample_ratio=wl["ds"], variant=variant)
    rec = out[2:]
    t_first = time.perf_counter() - t0
    n = int(max(3, min(30, seconds / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for i in range(n):
        out = O.forward(params, clip[:, i % T], *rec, downsample_ratio=wl["ds"], variant=variant)
        rec = out[2:]
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample="%d frames of the same %dx%d ds=%g clip, fp32 torch CPU oracle (oracle/rvm_oracle.py), "
                       "1 warm-up frame excluded" % (n, wl["W"], wl["H"], wl["ds"]))


def spawn_ranks(a):
    """`python bench.py --gpus N` with no launcher: N fresh children, one per GPU; this process never touches the GPU.
    All children are polled: when one exits non-zero the others are terminated (a rank that died at start-up would leave
    the rest in the barrier until the caller's time limit) and the exit code is non-zero.  Every rank's stderr is kept
    and replayed on failure; rank 0's stdout is the result line."""
    import socket
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, errs = [], []
    for r in range(a.gpus):
        env = dict(os.environ, WORLD_SIZE=str(a.gpus), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        ef = tempfile.TemporaryFile()
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef))
    rc, failed = 0, None
    live = set(range(a.gpus))
    while live and failed is None:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc, failed = code, r
                break
        else:
            if live:
                time.sleep(0.05)
    if failed is not None:
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    out = procs[0].stdout.read() if procs[0].stdout else b""
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were terminated\n" % (failed, rc))
        for r, ef in enumerate(errs):
            ef.seek(0)
            tail = ef.read().decode(errors="replace")[-4000:]
            if tail.strip():
                sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, tail))
    else:
        errs[0].seek(0)
        sys.stderr.write(errs[0].read().decode(errors="replace"))
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc if rc else 0


PIPE_KEYS = ("frame_pipeline", "warmup_pipeline_extra", "fps_pipeline_off", "fps_pipeline_on", "fps_sync_every_frame_pipelined")


def _ring_frames(H, W, batch, chunk, es, ring):
    """How many distinct step inputs the src ring holds: enough that their total exceeds the 256 MiB Infinity Cache, so a
    frame is not served from cache when its turn comes again (a real stream brings a new frame every step)."""
    if ring > 0:
        return ring
    per_step = 3 * H * W * es * batch * chunk
    return max(2, -(-int(L3_BYTES * 1.1) // per_step))


def measure(a, wl_key, batch, chunk, steps, warmup, dev, rank, world, dist, dump=""):
    """Time `steps` steps of one workload on this rank; rank 0 also gets the live per-launch roofline."""
    wl = WORKLOADS[wl_key]
    tdt = TORCH_DTYPE[a.dtype]
    es = 4 if a.dtype == "fp32" else 2
    variant = wl.get("variant", "mobilenetv3")
    model = MattingNetwork(variant, "deep_guided_filter").eval()
    model.load_state_dict(synthetic.make_params(variant, "deep_guided_filter", seed=0))
    model = model.to(dev, tdt)
    model.fp16_arithmetic = "fp16" if a.dtype == "fp16fast" else "fp32"
    model.enable_graph(a.mode == "graph")
    pipe = a.pipeline == "on"
    model.pipeline_frames(pipe)
    # the ring of step inputs: n_ring distinct frames (batches / chunks) of one synthetic clip per stream
    n_ring = _ring_frames(wl["H"], wl["W"], batch, chunk, es, a.src_ring)
    ring = []
    gen = [synthetic.clip_frames(n_ring * chunk, wl["H"], wl["W"], seed=rank * 64 + b) for b in range(batch)]
    for _ in range(n_ring):
        fr = torch.stack([torch.stack([next(g) for _ in range(chunk)]) for g in gen])      # [B, chunk, 3, H, W]
        fr = fr.to(dev, tdt)
        ring.append(fr[:, 0].contiguous() if chunk == 1 else fr.contiguous())
    del gen
    ds = wl["ds"]

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # rank r owns streams {r, r+world, ...} of the world*batch streams; they advance together as one batch
    ids = streams.shard_streams(world * batch, world, rank)
    runner = streams.StreamRunner(model, ids, ds)

    def timed(n_steps, inputs, sync_each=False, collective=True):
        k = len(inputs)
        with torch.no_grad():
            if collective:
                sync_all()
            else:
                torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(n_steps):
                o = runner.step(inputs[i % k])
                if sync_each:
                    torch.cuda.synchronize(dev)
            if collective:
                sync_all()
            else:
                torch.cuda.synchronize(dev)
            return time.perf_counter() - t0, o

    # W untimed warm-up steps - and, with the frame pipeline, enough further untimed ones that every buffer set has been through
    # its first frames (stream probe, stage graphs) and torch's allocator holds the blocks of all frames in flight: measured on
    # the 20-step window, 3420-4000 frames/s after 5 warm-up frames, 3830-4050 after 30 (the timed region is the K steps either way)
    extra = max(0, 24 - warmup) if pipe else 0
    with torch.no_grad():
        for i in range(warmup + extra):
            out = runner.step(ring[i % n_ring])
    dt_local, out = timed(steps, ring)
    on_gloo = dist is not None and dist.get_backend() == "gloo"
    fps, frames, dt = streams.aggregate_throughput(steps * batch * chunk, dt_local, dist, None if on_gloo else dev)
    assert torch.isfinite(out[0].float()).all() and torch.isfinite(out[1].float()).all()
    res = dict(fps=fps, ms_per_step=dt / steps * 1e3, steps=steps, workload=wl["name"], streams_per_gpu=batch, chunk=chunk,
               src_ring=n_ring, fps_local=steps * batch * chunk / dt_local)
    if dist is not None:                                    # every rank's own rate, for rank 0's line
        import torch as _t
        mine = _t.tensor([res["fps_local"]], dtype=_t.float64, device=None if on_gloo else dev)
        allr = [_t.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        res["fps_per_rank"] = [float(t.item()) for t in allr]
    # the other two protocols, untimed for the headline: the same input every step (what round 1 / 2 measured: the frame
    # then sits in the Infinity Cache) and the reference's loop, which synchronises after every frame
    # (inference_speed_test.py:45-47)
    n2 = max(20, min(steps, 100))

    def schedule(on):
        """Switch between the one-stream schedule and the frame pipeline (the plan cache keeps both kinds of plan)."""
        model.pipeline_frames(on)
        with torch.no_grad():
            for i in range(max(6, warmup // 2)):
                runner.step(ring[i % n_ring])
        torch.cuda.synchronize(dev)

    dt1, _ = timed(n2, ring[:1], collective=False)
    res["fps_single_src"] = n2 * batch * chunk / dt1
    res["frame_pipeline"] = bool(pipe)
    res["warmup_pipeline_extra"] = extra
    if pipe:
        dt2p, _ = timed(n2, ring, sync_each=True, collective=False)
        res["fps_sync_every_frame_pipelined"] = n2 * batch * chunk / dt2p
    # the one-stream schedule (every launch of a frame on the caller's stream), same loop; the reference's protocol - a device
    # synchronisation after every frame - is quoted on it: a caller that waits for each frame gains nothing from the pipeline
    schedule(False)
    if pipe:
        dt3, _ = timed(n2, ring, collective=False)
        res["fps_pipeline_off"] = n2 * batch * chunk / dt3
    dt2, _ = timed(n2, ring, sync_each=True, collective=False)
    res["fps_sync_every_frame"] = n2 * batch * chunk / dt2
    if not pipe and not a.one_schedule:
        schedule(True)
        dt3, _ = timed(n2, ring, collective=False)
        res["fps_pipeline_on"] = n2 * batch * chunk / dt3
        schedule(False)
    # (the per-launch profile below runs on the one-stream plan: its launches are the frame's, in order, on one stream)
    if rank != 0:
        return res
    # ---- live per-kernel timing (hipEvents on the stream the kernels are launched on)
    ctx = model._ctx
    info = ctx.launch_info()
    stream = torch.cuda.current_stream(dev).cuda_stream
    shapes = [ctx.state_shape(l) for l in (1, 2, 3, 4)]
    ra = [torch.zeros(s, device=dev, dtype=tdt).contiguous(memory_format=torch.channels_last) for s in shapes]
    rb = [torch.zeros(s, device=dev, dtype=tdt).contiguous(memory_format=torch.channels_last) for s in shapes]
    fgr = torch.empty((batch * chunk, 3, wl["H"], wl["W"]), device=dev, dtype=tdt)
    pha = torch.empty((batch * chunk, 1, wl["H"], wl["W"]), device=dev, dtype=tdt)
    native_src = [r if chunk == 1 else r.transpose(0, 1).contiguous() for r in ring]      # the native call takes time-major frames
    reps = 21

    def profile(inputs):
        nonlocal ra, rb
        samples = []
        for i in range(reps + 3):
            x = inputs[i % len(inputs)]
            ms = ctx.profile_frame(x.data_ptr(), [t.data_ptr() for t in ra], fgr.data_ptr(), pha.data_ptr(),
                                   [t.data_ptr() for t in rb], stream)
            ra, rb = rb, ra
            if i >= 3:
                samples.append(ms)
        return [sorted(col)[reps // 2] for col in zip(*samples)]      # median per launch: one stalled frame must not name the kernel

    avg = profile(native_src)
    avg_single = profile(native_src[:1])
    k = max(range(len(info)), key=lambda i: avg[i])
    name, flops, nbytes = info[k]
    sec = avg[k] * 1e-3
    intensity = flops / max(nbytes, 1.0)
    ridge = MFMA_PEAK[a.dtype] / HBM_PEAK
    is_gemm = any(s in name for s in (".expand", ".project", "1x1", ".conv", ".gru", ".fused"))
    if is_gemm and intensity > ridge:
        roof = dict(bound="mfma", achieved=flops / sec / 1e12, peak=MFMA_PEAK[a.dtype] / 1e12, unit="TFLOP/s",
                    frac=flops / sec / MFMA_PEAK[a.dtype], traffic=None)
    else:
        roof = dict(bound="hbm", achieved=nbytes / sec / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                    frac=nbytes / sec / HBM_PEAK, traffic=None)
    roof.update(kernel=name, kernel_ms=avg[k], kernel_ms_single_src=avg_single[k], kernel_flops=flops, kernel_bytes=nbytes,
                sum_kernel_ms=sum(avg), launches=len(info), kernels=sum(ctx.launch_kernels()))
    # the full-resolution refiner pass (the one HBM-bound launch of the frame) with the input rotating through the ring
    # and with one input re-used (then src comes out of the Infinity Cache)
    for i, (nm, fl, by) in enumerate(info):
        if nm == "dgf_apply_fullres":
            roof["dgf_apply"] = dict(bytes=by, ms_ring=avg[i], ms_single_src=avg_single[i],
                                     GBps_ring=by / (avg[i] * 1e-3) / 1e9, GBps_single_src=by / (avg_single[i] * 1e-3) / 1e9,
                                     frac_of_hbm_peak_ring=by / (avg[i] * 1e-3) / HBM_PEAK)
    # HBM bytes of that launch from the committed rocprofv3 --pmc passes of this same command
    # (FETCH_SIZE + WRITE_SIZE, separate passes; DESIGN.md "Measurement" has the gfx950 calibration note)
    tag = PROFILE_TAGS.get((wl_key, batch, chunk))
    pmc = os.path.join(ROOT, PROFILE_DIR, "pmc_%s.json" % tag) if tag else ""
    if tag and a.dtype == "fp16" and os.path.isfile(pmc):
        for e in json.load(open(pmc)):
            if e["name"] == name:
                roof["traffic"] = e["fetch_bytes"] + e["write_bytes"]
                roof["traffic_source"] = "%s/pmc_%s.json (%s)" % (PROFILE_DIR, tag, e.get("fetch_note", "raw FETCH_SIZE"))
    # ... and the committed rocprofv3 --kernel-trace mean duration of the same launch: the hipEvent figure above carries the
    # spacing of two event records (~2-3 us more than the kernel's own begin-to-end time), so both are shown
    trace = os.path.join(ROOT, PROFILE_DIR, "%s_kernel_trace_table.txt" % tag) if tag else ""
    if tag and a.dtype == "fp16" and os.path.isfile(trace):
        import re
        for line in open(trace):
            m = re.match(r"\s+(\S+)\s+([\d.]+) us", line)
            if m and m.group(1) == name:
                roof["rocprof_kernel_ms"] = float(m.group(2)) * 1e-3
                roof["rocprof_frac"] = (flops if roof["bound"] == "mfma" else nbytes) / (float(m.group(2)) * 1e-6) / \
                    (MFMA_PEAK[a.dtype] if roof["bound"] == "mfma" else HBM_PEAK)
                roof["rocprof_source"] = "%s/%s_kernel_trace_table.txt" % (PROFILE_DIR, tag)
    if dump:
        os.makedirs(os.path.dirname(os.path.abspath(dump)), exist_ok=True)
        with open(dump, "w") as f:
            json.dump([dict(name=n, ms=m, ms_single_src=m1, flops=fl, bytes=by, kernels=nk)
                       for (n, fl, by), m, m1, nk in zip(info, avg, avg_single, ctx.launch_kernels())], f, indent=1)
    res["roofline"] = roof
    # frame level: SURVEY 8(d)'s algorithmic bytes and the schedule's algorithmic FLOPs per frame against the two peaks
    h, w = spec.lowres_size(wl["H"], wl["W"], ds)
    balg = frame_algorithmic_bytes(wl["H"], wl["W"], h, w, es, variant)
    falg = sum(fl for _, fl, _ in info) / (batch * chunk)
    per_gpu = fps / world
    res["frame_roofline"] = {"algorithmic_bytes_per_frame": balg, "achieved_GBps": balg * per_gpu / 1e9,
                             "frac_of_hbm_peak": balg * per_gpu / HBM_PEAK,
                             "algorithmic_flops_per_frame": falg, "achieved_TFLOPps": falg * per_gpu / 1e12,
                             "frac_of_mfma_peak": falg * per_gpu / MFMA_PEAK[a.dtype]}
    res["memory_bytes"] = ctx.memory_bytes()
    del model, runner
    return res


class _NullWriter:
    """Counts the frames it is handed (a stand-in for an encoder: codecs are out of scope)."""

    def __init__(self):
        self.frames = 0
        self.checksum = 0

    def write(self, frames_u8):
        self.frames += frames_u8.shape[0]
        self.checksum += int(frames_u8[0, 0, 0, 0])

    def close(self):
        pass


def driver_pipeline(a, dev, wl_key="4k", T=64, chunk=4):
    """PCIe-inclusive rate of the driver loop (convert_video: host frames in, composed uint8 frames out) on an
    in-memory clip: the tensor-throughput `value` excludes host traffic like the reference's protocol does; this is what a
    caller with frames in host memory gets.  Two sources: uint8 frames (a decoder's output, 3 B / pixel up) and fp32 frames
    (the reference reader's ToTensor output, 12 B / pixel up); 3 B / pixel down either way.  "pinned": the clip lies in
    page-locked memory and is uploaded from where it is; otherwise every chunk is first copied into a pinned staging buffer
    (host memcpy, ~15 GB/s on one core - the bound of that case)."""
    from robustvideomatting_amd import inference as I
    wl = WORKLOADS[wl_key]
    tdt = TORCH_DTYPE[a.dtype]
    model = MattingNetwork("mobilenetv3", "deep_guided_filter").eval()
    model.load_state_dict(synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=0))
    model = model.to(dev, tdt)
    H, W = wl["H"], wl["W"]
    base = torch.stack(list(synthetic.clip_frames(chunk, H, W, seed=0)))                  # [chunk,3,H,W] fp32
    u8 = base.mul(255).byte().repeat(T // chunk, 1, 1, 1)                                # T frames (the chunk repeated)
    out = {}
    for name, clip, pin in (("uint8_source_pinned", u8, True), ("uint8_source", u8, False),
                            ("fp32_source_pinned", u8[:max(chunk, T // 2)].float().div(255), True)):
        n = clip.shape[0]
        I.convert_video(model, I.TensorSource(clip[:2 * chunk]), downsample_ratio=wl["ds"], output_type="video",
                        output_composition=_NullWriter(), seq_chunk=chunk, progress=False)     # plan + tile measurement
        sink = _NullWriter()
        src = I.TensorSource(clip).pin() if pin else I.TensorSource(clip)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        I.convert_video(model, src, downsample_ratio=wl["ds"], output_type="video",
                        output_composition=sink, seq_chunk=chunk, progress=False)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        assert sink.frames == n
        up = 3 * H * W * (1 if clip.dtype == torch.uint8 else 4)
        out[name] = dict(fps=n / dt, frames=n, seq_chunk=chunk, pcie_bytes_per_frame=up + 3 * H * W,
                         pcie_GBps=(up + 3 * H * W) * n / dt / 1e9)
    out["workload"] = wl["name"] + ", convert_video -> green-screen composition uint8, in-memory clip"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="hd", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp16fast", "fp32", "bf16"],
                    help="fp16: fp16 tensors + fp32-accurate arithmetic inside (parity-gated mode); fp16fast: fp16 storage + fp16 MFMA; "
                         "bf16: bf16 tensors at the boundary, same interior as fp16")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"],
                    help="HIP-graph replay of the state-dependent launches (model.enable_graph(); the default here: with the frame "
                         "pipeline the host thread is next to the bound, and replay takes it out - 5280-5300 frames/s over five "
                         "processes against 4150-5140 with eager launches) or eager launches (what a drop-in user gets by default)")
    ap.add_argument("--batch", type=int, default=1, help="streams batched per GPU (BASELINE configs[4]: --gpus 8 --batch 8)")
    ap.add_argument("--chunk", type=int, default=1,
                    help="frames of each stream per step (T of a [B,T,3,H,W] input, convert_video's seq_chunk); the headline is 1")
    ap.add_argument("--pipeline", default="on", choices=["on", "off"],
                    help="frame pipeline: the encoder phase of frame n+1 on a second stream beside the decoder phase of frame n")
    ap.add_argument("--one-schedule", action="store_true",
                    help="with --pipeline off: do not time the pipelined schedule either (profiler runs: every frame of the process "
                         "is then the same sequence of launches on one stream)")
    ap.add_argument("--src-ring", type=int, default=0,
                    help="distinct step inputs cycled through (0: as many as exceed the 256 MiB Infinity Cache; 1: the same input every step)")
    ap.add_argument("--driver-pipeline", action="store_true", help="only the PCIe-inclusive convert_video rate of --workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configurations reported under 'also' at N=1")
    ap.add_argument("--dump-kernels", default="")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # Rehearsal on a one-GPU box (RVM_BENCH_REHEARSE=1): every rank uses cuda:0 and the timing reduction runs over gloo
    # (RCCL refuses two ranks on one device).  The data path is the same - it has no collective.
    rehearse = os.environ.get("RVM_BENCH_REHEARSE", "") == "1"
    if rehearse:
        local = 0
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if a.driver_pipeline:
        print(json.dumps(driver_pipeline(a, dev, a.workload)))
        return
    head = measure(a, a.workload, a.batch, a.chunk, a.steps, a.warmup, dev, rank, world, dist, a.dump_kernels)

    if rank == 0:
        also, cpu, cpu_small = None, None, None
        default_cfg = a.workload == "hd" and a.batch == 1 and a.chunk == 1 and a.dtype == "fp16"
        if world == 1 and default_cfg and not a.no_also:
            # the other BASELINE configurations, same process, >= 100 steps each (headline fields above are unchanged)
            also = {}
            for key, wlk, b in (("4k", "4k", 1), ("hd_batch8", "hd", 8), ("hd_resnet50", "hd_resnet50", 1)):
                r = measure(a, wlk, b, 1, 100, 10, dev, 0, 1, None)
                also[key] = {k: r[k] for k in ("fps", "ms_per_step", "steps", "workload", "streams_per_gpu", "src_ring", "fps_single_src",
                                               "fps_sync_every_frame", "roofline", "frame_roofline", "memory_bytes") + PIPE_KEYS if k in r}
            also["driver_pipeline_4k"] = driver_pipeline(a, dev)
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(WORKLOADS[a.workload])
            if default_cfg and not a.no_also:
                cpu_small = cpu_baseline(WORKLOADS["small"], seconds=8.0, T=30)
                cpu_small["sample"] += "; BASELINE configs[0] (512x288, ds=1, fp32, T=30 clip cycled)"
        wl = WORKLOADS[a.workload]
        line = {
            "metric": "frames/sec, rvm_mobilenetv3 at HD ds=0.25 and 4K ds=0.125, 1->8 GPUs",
            "value": head["fps"], "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp16": "f16 (MFMA on split hi/lo f16 operands, f32 accumulate; f16 tensors)", "fp16fast": "f16", "fp32": "f32",
                      "bf16": "bf16 tensors (MFMA on split hi/lo f16 operands, f32 accumulate)"}[a.dtype],
            "data": "synthetic",
            "config": {"workload": wl["name"], "streams_per_gpu": a.batch, "frames_per_step": a.batch * a.chunk, "chunk": a.chunk, "mode": a.mode,
                       "frame_pipeline": a.pipeline == "on",
                       "src_ring": head["src_ring"],
                       "parallelism": "stream-sharded, %d independent stream(s), no collective" % (world * a.batch)},
            "roofline": head["roofline"], "cpu_baseline": cpu,
            "frame_roofline": head["frame_roofline"],
            # the same workload under the other two protocols (this rank): one input re-used every step (rounds 1-2; src then
            # stays in the 256 MiB Infinity Cache) and the reference's loop with a device synchronisation after every frame
            "fps_single_src": head["fps_single_src"], "fps_sync_every_frame": head["fps_sync_every_frame"],
            # frame pipeline (encoder stages of the next frames on the context's own streams beside the decoder phase of
            # this one): `value` is measured with it when --pipeline on (default); fps_pipeline_off / _on is the other schedule
            **{k: head[k] for k in PIPE_KEYS if k in head},
            "memory_bytes": head.get("memory_bytes"),
        }
        if "fps_per_rank" in head:
            line["fps_per_rank"] = head["fps_per_rank"]
        if also is not None:
            line["also"] = also
        if cpu_small is not None:
            line["cpu_baseline_small"] = cpu_small
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
