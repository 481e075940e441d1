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
    (total, total1), st, st1 = native.debug_stamps()
    assert total > 0, "library was not built with RVM_STAMPS"
    nfr = min(512 // n - 1, a.frames - 2)
    single = [i for i in range(n)]      # launches whose grid has one block never write table 1
    acc = [[0.0, [0.0] * 14, [0.0] * 14, 0.0] for _ in range(n)]
    # table 1 rows are counted separately: only launches with > 1 block appear, in launch order
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
    # match table-1 rows to launches by start time: the last block of launch i starts after block 0 of
    # launch i and before block 0 of launch i+1
    rows1 = sorted((r for r in st1 if r[0]), key=lambda r: r[0])
    starts = sorted(((st[(total - 1 - j) % 512][0], n - 1 - j % n) for j in range(min(total, 512))), key=lambda t: t[0])
    import bisect
    keys = [t[0] for t in starts]
    cnt = [0] * n
    for r in rows1:
        j = bisect.bisect_right(keys, r[0]) - 1
        if j < 0:
            continue
        i = starts[j][1]
        cnt[i] += 1
        acc[i][3] += (r[0] - starts[j][0]) * 0.01
        for k in range(14):
            if r[2 + k] and r[1 + k]:
                acc[i][2][k] += (r[2 + k] - r[1 + k])
    print("RVM_DBG=%s  %d frames, %d launches, frame %.1f us" % (os.environ.get("RVM_DBG", ""), nfr, n, sum(x[0] for x in acc)))
    for i in range(n):
        ph = " ".join("%6.0f" % c for c in acc[i][1] if c)
        ph1 = ""
        if cnt[i]:
            ph1 = "  | last block +%.1f us: " % (acc[i][3] / cnt[i]) + " ".join("%6.0f" % (c / cnt[i]) for c in acc[i][2] if c)
        print("  %-20s %6.2f us   cycles: %s%s" % (names[i], acc[i][0], ph, ph1))


if __name__ == "__main__":
    main()
