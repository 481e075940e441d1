"""dgf_apply_pow2 (ratios 1/4 and 1/8, 1 or 2 rows per block) against the general kernel: bit-identical outputs (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvideomatting_amd import MattingNetwork, synthetic

def run(rows, tdt, H, W, ds, B, refiner="deep_guided_filter"):
    os.environ.pop("RVM_DGF_ROWS", None); os.environ.pop("RVM_DGF_POW2", None)
    if rows == 0: os.environ["RVM_DGF_POW2"] = "0"
    elif rows is not None: os.environ["RVM_DGF_ROWS"] = str(rows)
    os.environ["RVM_AUTOTUNE"] = "0"
    p = synthetic.make_params("mobilenetv3", refiner, seed=0, perturb_box_filter=True)
    m = MattingNetwork("mobilenetv3", refiner).eval(); m.load_state_dict(p); m = m.to("cuda", tdt)
    src = synthetic.make_clip(1, H, W, seed=1, B=B)[:, 0].to("cuda", tdt)
    with torch.no_grad():
        o = m(src, downsample_ratio=ds)
    return [x.clone() for x in o[:2]]

bad = 0
for tdt in (torch.float16, torch.float32, torch.bfloat16):
    for (H, W, ds, B) in ((1080, 1920, 0.25, 2), (2160, 3840, 0.125, 1), (720, 1280, 0.125, 1), (272, 480, 0.25, 1), (64, 4096, 0.25, 1), (48, 4104, 0.125, 1), (8, 8, 0.25, 3), (270, 479, 0.25, 1)):
        base = run(0, tdt, H, W, ds, B)
        for rows in (None, 1):
            got = run(rows, tdt, H, W, ds, B)
            ok = all(torch.equal(a, b) for a, b in zip(base, got))
            bad += not ok
            print(tdt, (H, W, ds, B), "rows", rows, "equal" if ok else "DIFFERENT %.3e %.3e  n=%d" % ((base[0].float() - got[0].float()).abs().max().item(), (base[1].float() - got[1].float()).abs().max().item(), (base[0] != got[0]).sum().item()), flush=True)
print("FAILED" if bad else "all equal")
sys.exit(1 if bad else 0)
