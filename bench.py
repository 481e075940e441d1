"""Throughput bench for the RVM per-frame inference path on MI355X.

    python bench.py --gpus 1 --steps 300 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward of one frame batch through MattingNetwork.forward with the recurrent
state carried (inference_speed_test.py:42-47 protocol of the reference: device-resident input,
tensor throughput only).  Default workload = BASELINE.json configs[1]: rvm_mobilenetv3 fp16,
one 1920x1080 stream, downsample_ratio 0.25.  With N GPUs every rank runs its own independent
stream (weak scaling, no collective on the data path); value = all frames / max rank time.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline definitions).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, streams, synthetic   # noqa: E402

WORKLOADS = {
    "hd": dict(H=1080, W=1920, ds=0.25, name="rvm_mobilenetv3 fp16 1x1920x1080 ds=0.25 (BASELINE configs[1])"),
    "4k": dict(H=2160, W=3840, ds=0.125, name="rvm_mobilenetv3 fp16 1x3840x2160 ds=0.125 (BASELINE configs[2])"),
    "hd_resnet50": dict(H=1080, W=1920, ds=0.25, variant="resnet50",
                        name="rvm_resnet50 fp16 1x1920x1080 ds=0.25 (BASELINE configs[3])"),
    "small": dict(H=288, W=512, ds=1.0, name="rvm_mobilenetv3 512x288 ds=1 (BASELINE configs[0] shape)"),
}
HBM_PEAK = 8.0e12            # B/s, MI355X_MICROARCH.md
MFMA_PEAK = {"fp16fast": 2.5e15, "fp16": 2.5e15, "fp32": 157.3e12}   # dense peak of the MFMA the mode issues (f16 / f16 split / f32)
PMC_FILES = {"hd": "profiles/r01/pmc_hd_v8.json", "4k": "profiles/r01/pmc_4k_v8.json"}   # tools/run_pmc.sh output of this command


def frame_algorithmic_bytes(H, W, h, w, es):
    """src read + fgr/pha write + r1..r4 read+write + weights (mobilenetv3)."""
    hw = [(h, w)]
    for _ in range(4):
        hw.append(((hw[-1][0] + 1) // 2, (hw[-1][1] + 1) // 2))
    S = sum(c * a * b for c, (a, b) in zip((16, 20, 40, 64), hw[1:]))
    P = 3.75e6
    return es * 7 * H * W + es * 2 * S + es * P


def cpu_baseline(wl, seconds=15.0):
    """The fp32 CPU oracle (restatement of the reference's PyTorch path) timed on host cores."""
    from oracle import rvm_oracle as O
    variant = wl.get("variant", "mobilenetv3")
    params = synthetic.make_params(variant, "deep_guided_filter", seed=0)
    clip = synthetic.make_clip(2, wl["H"], wl["W"], seed=0)
    rec = [None] * 4
    t0 = time.perf_counter()
    out = O.forward(params, clip[:, 0], *rec, downsample_ratio=wl["ds"], variant=variant)
    rec = out[2:]
    t_first = time.perf_counter() - t0
    n = int(max(3, min(30, seconds / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for i in range(n):
        out = O.forward(params, clip[:, i % 2], *rec, downsample_ratio=wl["ds"], variant=variant)
        rec = out[2:]
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample="%d frames of the same %dx%d ds=%g clip, fp32 torch CPU oracle (oracle/rvm_oracle.py), "
                       "1 warm-up frame excluded" % (n, wl["W"], wl["H"], wl["ds"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="hd", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp16fast", "fp32"],
                    help="fp16: fp16 tensors + fp32 arithmetic inside (parity-gated mode); fp16fast: fp16 storage + fp16 MFMA")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"])
    ap.add_argument("--batch", type=int, default=1, help="streams batched per GPU")
    ap.add_argument("--chunk", type=int, default=1,
                    help="frames of each stream per step (T of a [B,T,3,H,W] input, convert_video's seq_chunk); the headline is 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-kernels", default="")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    wl = WORKLOADS[a.workload]
    tdt = torch.float32 if a.dtype == "fp32" else torch.float16
    es = 4 if a.dtype == "fp32" else 2

    variant = wl.get("variant", "mobilenetv3")
    model = MattingNetwork(variant, "deep_guided_filter").eval()
    model.load_state_dict(synthetic.make_params(variant, "deep_guided_filter", seed=0))
    model = model.to(dev, tdt)
    model.fp16_arithmetic = "fp16" if a.dtype == "fp16fast" else "fp32"
    model.enable_graph(a.mode == "graph")
    clip = synthetic.make_clip(a.chunk, wl["H"], wl["W"], seed=rank, B=a.batch).to(dev, tdt).contiguous()
    src = clip[:, 0].contiguous() if a.chunk == 1 else clip
    ds = wl["ds"]

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # rank r owns streams {r, r+world, ...} of the world*batch streams; they advance together as one batch
    ids = streams.shard_streams(world * a.batch, world, rank)
    runner = streams.StreamRunner(model, ids, ds)
    with torch.no_grad():
        for _ in range(a.warmup):
            out = runner.step(src)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = runner.step(src)
        sync_all()
        dt_local = time.perf_counter() - t0
    fps, frames, dt = streams.aggregate_throughput(a.steps * a.batch * a.chunk, dt_local, dist, dev)
    assert torch.isfinite(out[0].float()).all() and torch.isfinite(out[1].float()).all()

    roof = None
    cpu = None
    if rank == 0:
        # ---- live per-kernel timing (hipEvents on the stream the kernels are launched on)
        ctx = model._ctx
        info = ctx.launch_info()
        stream = torch.cuda.current_stream(dev).cuda_stream
        shapes = [ctx.state_shape(l) for l in (1, 2, 3, 4)]
        ra = [torch.zeros(s, device=dev, dtype=tdt).contiguous(memory_format=torch.channels_last) for s in shapes]
        rb = [torch.zeros(s, device=dev, dtype=tdt).contiguous(memory_format=torch.channels_last) for s in shapes]
        fgr = torch.empty((a.batch * a.chunk, 3, wl["H"], wl["W"]), device=dev, dtype=tdt)
        pha = torch.empty((a.batch * a.chunk, 1, wl["H"], wl["W"]), device=dev, dtype=tdt)
        if a.chunk > 1:
            src = clip.transpose(0, 1).contiguous()             # the native call takes time-major frames
        acc = [0.0] * len(info)
        reps = 20
        for i in range(reps + 3):
            ms = ctx.profile_frame(src.data_ptr(), [t.data_ptr() for t in ra], fgr.data_ptr(), pha.data_ptr(),
                                   [t.data_ptr() for t in rb], stream)
            ra, rb = rb, ra
            if i >= 3:
                acc = [x + y for x, y in zip(acc, ms)]
        avg = [x / reps for x in acc]
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
                    sum_kernel_ms=sum(avg), launches=len(info))
        # HBM bytes of that launch from the committed rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE + WRITE_SIZE, separate passes; DESIGN.md "Measurement" has the gfx950 calibration note)
        pmc = os.path.join(ROOT, PMC_FILES.get(a.workload, ""))
        if a.batch == 1 and a.chunk == 1 and a.dtype == "fp16" and os.path.isfile(pmc):
            for e in json.load(open(pmc)):
                if e["name"] == name:
                    roof["traffic"] = e["fetch_bytes_raw"] + e["write_bytes"]
                    roof["traffic_source"] = PMC_FILES[a.workload]
        if a.dump_kernels:
            os.makedirs(os.path.dirname(os.path.abspath(a.dump_kernels)), exist_ok=True)
            with open(a.dump_kernels, "w") as f:
                json.dump([dict(name=n, ms=m, flops=fl, bytes=by) for (n, fl, by), m in zip(info, avg)], f, indent=1)
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(wl)

    if rank == 0:
        h, w = (wl["H"], wl["W"]) if ds == 1 else (int(wl["H"] * ds), int(wl["W"] * ds))
        balg = frame_algorithmic_bytes(wl["H"], wl["W"], h, w, es)
        line = {
            "metric": "frames/sec, rvm_mobilenetv3 at HD ds=0.25 and 4K ds=0.125, 1->8 GPUs",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp16": "f16 (MFMA on split hi/lo f16 operands, f32 accumulate; f16 tensors)", "fp16fast": "f16", "fp32": "f32"}[a.dtype],
            "data": "synthetic",
            "config": {"workload": wl["name"], "streams_per_gpu": a.batch, "frames_per_step": a.batch * a.chunk, "chunk": a.chunk, "mode": a.mode,
                       "parallelism": "stream-sharded, %d independent stream(s), no collective" % (world * a.batch)},
            "roofline": roof, "cpu_baseline": cpu,
            "frame_roofline": {"algorithmic_bytes_per_frame": balg,
                               "achieved_GBps": balg * fps / world / 1e9, "frac_of_hbm_peak": balg * fps / world / HBM_PEAK},
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
