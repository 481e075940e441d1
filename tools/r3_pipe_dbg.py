import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, synthetic
dtype = torch.float16
m = MattingNetwork("mobilenetv3", "deep_guided_filter").eval()
m.load_state_dict(synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=3, perturb_box_filter=True))
m = m.to("cuda", dtype)
B, T = int(sys.argv[1]), int(sys.argv[2])
clip = synthetic.make_clip(12, 270, 480, seed=0, B=B).to("cuda", dtype).contiguous()
srcs = [clip[:, t0:t0 + T].contiguous() for t0 in range(0, 12, T)]
torch.cuda.synchronize()
def chunks(sync):
    rec = [None] * 4
    outs = []
    with torch.no_grad():
        for src in srcs:
            o = m(src, None, *rec, 0.5)
            rec = o[2:]
            outs.append([x.clone() for x in o])
            if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return outs
want = chunks(True)
m.pipeline_frames(True)
for sync in (True, False):
    got = chunks(sync)
    for t, (a, b) in enumerate(zip(want, got)):
        print("sync", sync, "chunk", t, [("%.3g" % (x.float() - y.float()).abs().max().item()) for x, y in zip(a, b)], [bool(torch.isnan(y.float()).any()) for y in b])
