"""Per-tap parity table: HIP path vs the CPU oracle on one seeded clip (debug / test helper).

    python tools/gpu_check.py [--dtype fp32|fp16] [--H 122 --W 218 --ds 0.25 --T 3 --B 1]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rvm_oracle as O                        # noqa: E402  (checker only)
from robustvideomatting_amd import MattingNetwork, synthetic   # noqa: E402

def fast_gf_bound(params, clip_ref, ds, tol, variant="mobilenetv3"):
    """Parity bound for the FAST guided filter (fast_guided_filter.py:55: A = cov / (var + 1e-5)): the division is
    ill-conditioned where var ~ eps, so the reference's own fp32 arithmetic sits a few 1e-5..1e-3 away from the same
    arithmetic in fp64, depending on the clip.  Bound = the mode's tolerance, or 3x the fp32 oracle's own distance to the
    fp64 oracle on THIS clip if that is larger (clip_ref: [B,T,3,H,W] fp32, what the device sees)."""
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    o32 = O.forward(params, clip_ref, downsample_ratio=ds, variant=variant, refiner="fast_guided_filter")
    o64 = O.forward(p64, clip_ref.double(), downsample_ratio=ds, variant=variant, refiner="fast_guided_filter")
    own = max((o32[0].double() - o64[0]).abs().max().item(), (o32[1].double() - o64[1]).abs().max().item())
    return max(tol, 3.0 * own), own


TAPS = ["s0", "s1", "s2", "s3", "f1", "f2", "f3", "f4", "aspp", "x3", "x2", "x1", "hid", "ab"]


def run(dtype="fp32", H=122, W=218, ds=0.25, T=3, B=1, variant="mobilenetv3", refiner="deep_guided_filter",
        seed=0, verbose=True, graph=False, noise=0.04):
    # dtype: fp32 | fp16 (fp16 tensors, fp32 arithmetic inside) | fp16fast (fp16 storage + fp16 MFMA) | bf16 (bf16 tensors, fp32 inside)
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16}.get(dtype, torch.float16)
    params = synthetic.make_params(variant, refiner, seed=seed, perturb_box_filter=True)
    model = MattingNetwork(variant, refiner).eval()
    model.load_state_dict(params)
    model = model.to("cuda", tdt)
    model.fp16_arithmetic = "fp16" if dtype == "fp16fast" else "fp32"
    if graph:
        model.enable_graph(True)
    clip = synthetic.make_clip(T, H, W, seed=seed, B=B, noise=noise)    # [B,T,3,H,W] fp32
    clip_dev = clip.to("cuda", tdt)
    clip_ref = clip_dev.float().cpu()                                   # what the device actually sees
    rec_g, rec_o = [None] * 4, [None] * 4
    worst = {}
    for t in range(T):
        with torch.no_grad():
            out_g = model(clip_dev[:, t], None, *rec_g, ds)
        taps_o = []
        out_o = O.forward(params, clip_ref[:, t], *rec_o, downsample_ratio=ds, variant=variant,
                          refiner=refiner, taps=taps_o)
        to = taps_o[0]
        rows = []
        for name in TAPS:
            if name == "ab" and ds == 1:
                continue
            g = model.debug_tap(name)
            if name.startswith("s") and name[1:].isdigit():
                ref = {"s0": to["src_sm"]}.get(name)
                if ref is None:
                    ref = to["src_sm"]
                    for _ in range(int(name[1:])):
                        ref = O._avgpool(ref)
                g = g[:, :3]
            elif name == "ab":
                ref = torch.cat([to["A"], to["b"]], 1)
            elif name in ("x3", "x2", "x1"):
                ref = to[name]
                half = ref.shape[1] // 2
                g, ref = g[:, :half], ref[:, :half]
            else:
                ref = to[name]
            rows.append((name, (g - ref).abs().max().item(), ref.abs().max().item()))
        names = ["fgr", "pha", "r1", "r2", "r3", "r4"]
        for n, a, b in zip(names, out_g, out_o):
            rows.append((n, (a.float().cpu() - b).abs().max().item(), b.abs().max().item()))
        if verbose:
            print("frame %d" % t)
            for n, e, m in rows:
                print("   %-5s max|err| %.3e   (max|ref| %.3f)" % (n, e, m))
        for n, e, m in rows:
            worst[n] = max(worst.get(n, 0.0), e)
        rec_g, rec_o = out_g[2:], out_o[2:]
    return worst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--H", type=int, default=122)
    ap.add_argument("--W", type=int, default=218)
    ap.add_argument("--ds", type=float, default=0.25)
    ap.add_argument("--T", type=int, default=3)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--refiner", default="deep_guided_filter")
    ap.add_argument("--variant", default="mobilenetv3")
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    w = run(a.dtype, a.H, a.W, a.ds, a.T, a.B, a.variant, a.refiner, graph=a.graph)
    print("worst:", {k: "%.2e" % v for k, v in w.items()})
