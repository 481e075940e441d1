"""Per-launch table from a rocprofv3 --kernel-trace CSV of bench.py.

    python tools/trace_table.py <kernel_trace.csv> <kernels.json from bench.py --dump-kernels> [frames]

Takes the last `frames` complete frames (a frame = len(kernels.json) consecutive dispatches of the
library's kernels), and prints for every launch the mean device duration, the mean idle gap in front
of it, its grid and resources.  This is what the roofline figures in profiles/ are checked against.
"""
import csv
import json
import sys


def main():
    trace, kjson = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 25      # bench.py ends with ~23 event-timed eager frames
    names = [k["name"] for k in json.load(open(kjson))]
    info = {k["name"]: k for k in json.load(open(kjson))}
    nk = [k.get("kernels", 1) for k in json.load(open(kjson))]       # kernels per launch (split-K conv + reduce: 2)
    first = [sum(nk[:i]) for i in range(len(nk))]
    nl = len(names)
    n = sum(nk)                                                      # dispatches per frame
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            kn = r["Kernel_Name"]
            if kn.startswith("void at::") or kn.startswith("__amd_rocclr") or "at::native" in kn:
                continue
            rows.append(r)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[:len(rows) - skip * n]
    nf = min(frames, len(rows) // n - 1)
    rows = rows[len(rows) - nf * n:]
    dur = [0.0] * nl
    gap = [0.0] * nl
    meta = [None] * nl
    for fi in range(nf):
        for i in range(nl):
            for j in range(nk[i]):
                r = rows[fi * n + first[i] + j]
                dur[i] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / nf
                if fi * n + first[i] + j > 0:
                    p = rows[fi * n + first[i] + j - 1]
                    gap[i] += (int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / nf
                if j == 0:
                    meta[i] = r
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / nf
    print("%d frames, %d launches/frame, frame span %.1f us, kernel sum %.1f us, gap sum %.1f us"
          % (nf, nl, span / 1e3, sum(dur) / 1e3, sum(gap) / 1e3))
    for i in range(nl):
        r = meta[i]
        k = info[names[i]]
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        blocks = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
        us = dur[i] / 1e3
        print("  %-20s %6.1f us  gap %5.1f  blk %5d x%4d  lds %6s vgpr %3s  %8.1f GFLOP/s %7.1f GB/s  %s"
              % (names[i], us, gap[i] / 1e3, blocks, wg, r["LDS_Block_Size"], r["VGPR_Count"],
                 k["flops"] / us / 1e3, k["bytes"] / us / 1e3, r["Kernel_Name"][:40]))


if __name__ == "__main__":
    main()
