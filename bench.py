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
PMC_FILES = {"hd": "profiles/r02/pmc_hd.json", "4k": "profiles/r02/pmc_4k.json",
             "hd_resnet50": "profiles/r02/pmc_rn.json"}              # tools/run_pmc.sh output of this command
TRACE_FILES = {"hd": "profiles/r02/hd_kernel_trace_table.txt", "4k": "profiles/r02/4k_kernel_trace_table.txt",
               "hd_resnet50": "profiles/r02/rn_kernel_trace_table.txt"}     # tools/run_prof.sh output of this command
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
    """`python bench.py --gpus N` with no launcher: N children, one per GPU; this process never touches the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, WORLD_SIZE=str(a.gpus), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def measure(a, wl_key, batch, chunk, steps, warmup, dev, rank, world, dist, dump=""):
    """Time `steps` steps of one workload on this rank; rank 0 also gets the live per-launch roofline."""
    wl = WORKLOADS[wl_key]
    tdt = TORCH_DTYPE[a.dtype]
    variant = wl.get("variant", "mobilenetv3")
    model = MattingNetwork(variant, "deep_guided_filter").eval()
    model.load_state_dict(synthetic.make_params(variant, "deep_guided_filter", seed=0))
    model = model.to(dev, tdt)
    model.fp16_arithmetic = "fp16" if a.dtype == "fp16fast" else "fp32"
    model.enable_graph(a.mode == "graph")
    clip = synthetic.make_clip(chunk, wl["H"], wl["W"], seed=rank, B=batch).to(dev, tdt).contiguous()
    src = clip[:, 0].contiguous() if chunk == 1 else clip
    ds = wl["ds"]

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # rank r owns streams {r, r+world, ...} of the world*batch streams; they advance together as one batch
    ids = streams.shard_streams(world * batch, world, rank)
    runner = streams.StreamRunner(model, ids, ds)
    with torch.no_grad():
        for _ in range(warmup):
            out = runner.step(src)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = runner.step(src)
        sync_all()
        dt_local = time.perf_counter() - t0
    on_gloo = dist is not None and dist.get_backend() == "gloo"
    fps, frames, dt = streams.aggregate_throughput(steps * batch * chunk, dt_local, dist, None if on_gloo else dev)
    assert torch.isfinite(out[0].float()).all() and torch.isfinite(out[1].float()).all()
    res = dict(fps=fps, ms_per_step=dt / steps * 1e3, steps=steps, workload=wl["name"], streams_per_gpu=batch, chunk=chunk)
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
    if chunk > 1:
        src = clip.transpose(0, 1).contiguous()             # the native call takes time-major frames
    samples = []
    reps = 21
    for i in range(reps + 3):
        ms = ctx.profile_frame(src.data_ptr(), [t.data_ptr() for t in ra], fgr.data_ptr(), pha.data_ptr(),
                               [t.data_ptr() for t in rb], stream)
        ra, rb = rb, ra
        if i >= 3:
            samples.append(ms)
    avg = [sorted(col)[reps // 2] for col in zip(*samples)]      # median per launch: one stalled frame must not name the kernel
    k = max(range(len(info)), key=lambda i: avg[i])
    name, flops, nbytes = info[k]
    sec = avg[k] * 1e-3
    intensity = flops / max(nbytes, 1.0)
    ridge = MFMA_PEAK[a.dtype] / HBM_PEAK
    is_gemm = any(s in name for s in (".expand", ".project", "1x1", ".conv", ".gru"))
    if is_gemm and intensity > ridge:
        roof = dict(bound="mfma", achieved=flops / sec / 1e12, peak=MFMA_PEAK[a.dtype] / 1e12, unit="TFLOP/s",
                    frac=flops / sec / MFMA_PEAK[a.dtype], traffic=None)
    else:
        roof = dict(bound="hbm", achieved=nbytes / sec / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                    frac=nbytes / sec / HBM_PEAK, traffic=None)
    roof.update(kernel=name, kernel_ms=avg[k], kernel_flops=flops, kernel_bytes=nbytes,
                sum_kernel_ms=sum(avg), launches=len(info), kernels=sum(ctx.launch_kernels()))
    # HBM bytes of that launch from the committed rocprofv3 --pmc passes of this same command
    # (FETCH_SIZE + WRITE_SIZE, separate passes; DESIGN.md "Measurement" has the gfx950 calibration note)
    pmc = os.path.join(ROOT, PMC_FILES.get(wl_key, ""))
    if batch == 1 and chunk == 1 and a.dtype == "fp16" and os.path.isfile(pmc):
        for e in json.load(open(pmc)):
            if e["name"] == name:
                roof["traffic"] = e["fetch_bytes"] + e["write_bytes"]
                roof["traffic_source"] = PMC_FILES[wl_key] + " (" + e.get("fetch_note", "raw FETCH_SIZE") + ")"
    # ... and the committed rocprofv3 --kernel-trace mean duration of the same launch: the hipEvent figure above carries the
    # spacing of two event records (~2-3 us more than the kernel's own begin-to-end time), so both are shown
    trace = os.path.join(ROOT, TRACE_FILES.get(wl_key, ""))
    if batch == 1 and chunk == 1 and a.dtype == "fp16" and os.path.isfile(trace):
        import re
        for line in open(trace):
            m = re.match(r"\s+(\S+)\s+([\d.]+) us", line)
            if m and m.group(1) == name:
                roof["rocprof_kernel_ms"] = float(m.group(2)) * 1e-3
                roof["rocprof_frac"] = (flops if roof["bound"] == "mfma" else nbytes) / (float(m.group(2)) * 1e-6) / \
                    (MFMA_PEAK[a.dtype] if roof["bound"] == "mfma" else HBM_PEAK)
                roof["rocprof_source"] = TRACE_FILES[wl_key]
    if dump:
        os.makedirs(os.path.dirname(os.path.abspath(dump)), exist_ok=True)
        with open(dump, "w") as f:
            json.dump([dict(name=n, ms=m, flops=fl, bytes=by, kernels=nk)
                       for (n, fl, by), m, nk in zip(info, avg, ctx.launch_kernels())], f, indent=1)
    res["roofline"] = roof
    h, w = spec.lowres_size(wl["H"], wl["W"], ds)
    es = 4 if a.dtype == "fp32" else 2
    balg = frame_algorithmic_bytes(wl["H"], wl["W"], h, w, es, variant)
    per_gpu = fps / world
    res["frame_roofline"] = {"algorithmic_bytes_per_frame": balg, "achieved_GBps": balg * per_gpu / 1e9,
                             "frac_of_hbm_peak": balg * per_gpu / HBM_PEAK}
    del model, runner
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="hd", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp16fast", "fp32", "bf16"],
                    help="fp16: fp16 tensors + fp32-accurate arithmetic inside (parity-gated mode); fp16fast: fp16 storage + fp16 MFMA; "
                         "bf16: bf16 tensors at the boundary, same interior as fp16")
    ap.add_argument("--mode", default="eager", choices=["graph", "eager"],
                    help="eager (what a drop-in user gets by default) or HIP-graph replay (model.enable_graph())")
    ap.add_argument("--batch", type=int, default=1, help="streams batched per GPU (BASELINE configs[4]: --gpus 8 --batch 8)")
    ap.add_argument("--chunk", type=int, default=1,
                    help="frames of each stream per step (T of a [B,T,3,H,W] input, convert_video's seq_chunk); the headline is 1")
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

    head = measure(a, a.workload, a.batch, a.chunk, a.steps, a.warmup, dev, rank, world, dist, a.dump_kernels)

    if rank == 0:
        also, cpu, cpu_small = None, None, None
        default_cfg = a.workload == "hd" and a.batch == 1 and a.chunk == 1 and a.dtype == "fp16"
        if world == 1 and default_cfg and not a.no_also:
            # the other BASELINE configurations, same process, >= 100 steps each (headline fields above are unchanged)
            also = {}
            for key, wlk, b in (("4k", "4k", 1), ("hd_batch8", "hd", 8), ("hd_resnet50", "hd_resnet50", 1)):
                r = measure(a, wlk, b, 1, 100, 10, dev, 0, 1, None)
                also[key] = dict(fps=r["fps"], ms_per_step=r["ms_per_step"], steps=r["steps"], workload=r["workload"],
                                 streams_per_gpu=b, roofline=r["roofline"], frame_roofline=r["frame_roofline"])
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
                       "parallelism": "stream-sharded, %d independent stream(s), no collective" % (world * a.batch)},
            "roofline": head["roofline"], "cpu_baseline": cpu,
            "frame_roofline": head["frame_roofline"],
        }
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
