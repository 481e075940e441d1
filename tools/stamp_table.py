"""True (un-profiled) per-launch timeline of graph-mode frames from the RVM_STAMPS build.

    RVM_LIB_VARIANT=stamps python tools/stamp_table.py [--workload hd] [--dtype fp16] [--frames 40]

For every launch of the frame: time from its block-0 start to the next launch's block-0 start (what the
launch costs the frame, boundary included) and the shader-clock deltas between its phase stamps.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RVM_LIB_VARIANT", "stamps")
from robustvideomatting_amd import MattingNetwork, native, synthetic   # noqa: E402
import bench                                                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hd")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    tdt = torch.float32 if a.dtype == "fp32" else torch.float16
    variant = wl.get("variant", "mobilenetv3")
    model = MattingNetwork(variant, "deep_guided_filter").eval()
    model.load_state_dict(synthetic.make_params(variant, "deep_guided_filter", seed=0))
    model = model.to("cuda", tdt)
    model.fp16_arithmetic = "fp16" if a.dtype == "fp16fast" else "fp32"
    model.enable_graph(True)
    src = synthetic.make_clip(1, wl["H"], wl["W"], seed=0, B=a.batch)[:, 0].to("cuda", tdt).contiguous()
    rec = [None] * 4
    with torch.no_grad():
        for _ in range(a.frames):
            out = model(src, None, *rec, wl["ds"])
            rec = out[2:]
    torch.cuda.synchronize()
    names = [n for n, _, _ in model._ctx.launch_info()]
    n = len(names)
    total, st = native.debug_stamps()
    assert total > 0, "library was not built with RVM_STAMPS"
    assert total % n == 0, (total, n)
    nfr = min(512 // n - 1, a.frames - 2)
    acc = [[0.0, [0.0] * 14] for _ in range(n)]
    for f in range(nfr):
        first = total - (f + 1) * n            # launch counter value of this frame's launch 0
        for i in range(n):
            s = st[(first + i) % 512]
            nxt = st[(first + i + 1) % 512] if (i + 1 < n or f > 0) else None
            if nxt is not None:
                acc[i][0] += (nxt[0] - s[0]) * 0.01 / nfr     # 100 MHz -> us
            for k in range(14):
                if s[2 + k] and s[1 + k]:
                    acc[i][1][k] += (s[2 + k] - s[1 + k]) / nfr
    print("%d frames, %d launches, frame %.1f us" % (nfr, n, sum(x[0] for x in acc)))
    for i in range(n):
        ph = " ".join("%6.0f" % c for c in acc[i][1] if c)
        print("  %-20s %6.2f us   block-0 phase cycles: %s" % (names[i], acc[i][0], ph))


if __name__ == "__main__":
    main()
