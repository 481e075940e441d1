"""Soak of the frame pipeline: N frames sent without a host synchronisation, every frame's outputs reduced to checksums on the
device, compared with the one-stream schedule's at the end (must be equal bit for bit), final states compared in full.

    python tools/pipe_soak.py [--variant mobilenetv3] [--frames 2000] [--H 1080 --W 1920 --ds 0.25] [--batch 1]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, synthetic   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="mobilenetv3")
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--H", type=int, default=1080)
    ap.add_argument("--W", type=int, default=1920)
    ap.add_argument("--ds", type=float, default=0.25)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--fast", action="store_true", help="fp16 module with pure-fp16 arithmetic (the host-bound regime)")
    ap.add_argument("--jitter", type=float, default=0.0, help="fraction of frames in front of which the host sleeps 0.1-1 ms or "
                    "synchronises: frames that find the device idle (they run on the caller's stream) mixed with pipelined ones")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype]
    m = MattingNetwork(a.variant, "deep_guided_filter").eval()
    m.load_state_dict(synthetic.make_params(a.variant, "deep_guided_filter", seed=5))
    m = m.to("cuda", dt)
    if a.fast:
        m.fp16_arithmetic = "fp16"
    import random, time
    ring = [f.contiguous() for f in synthetic.make_clip(16, a.H, a.W, seed=2, B=a.batch).to("cuda", dt).unbind(1)]
    torch.cuda.synchronize()

    def run(pipe):
        m.pipeline_frames(pipe)
        rec = [None] * 4
        sums = torch.zeros((a.frames, 2), dtype=torch.float64, device="cuda")
        with torch.no_grad():
            rng = random.Random(7)
            for i in range(a.frames):
                if pipe and a.jitter and rng.random() < a.jitter:
                    if rng.random() < 0.3:
                        torch.cuda.synchronize()
                    else:
                        time.sleep(rng.uniform(1e-4, 1e-3))
                o = m(ring[i % 16], None, *rec, a.ds)
                rec = o[2:]
                sums[i, 0] = o[0].double().sum()
                sums[i, 1] = o[1].double().sum()
        torch.cuda.synchronize()
        return sums.cpu(), [r.clone() for r in rec]

    want, wrec = run(False)
    got, grec = run(True)
    bad = (want != got).any(dim=1).nonzero().flatten()
    ok = bad.numel() == 0 and all(torch.equal(x, y) for x, y in zip(wrec, grec))
    print("%s %s%s %dx%d B=%d: %d frames, %d differ%s, final states %s" % (
        a.variant, a.dtype, (" fast" if a.fast else "") + ((" jitter %.2f" % a.jitter) if a.jitter else ""), a.W, a.H, a.batch, a.frames, bad.numel(), (" (first %d)" % bad[0].item()) if bad.numel() else "",
        "equal" if all(torch.equal(x, y) for x, y in zip(wrec, grec)) else "DIFFER"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
