import sys, os, time, torch, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, synthetic
dtype = torch.float16
m = MattingNetwork("mobilenetv3", "deep_guided_filter").eval()
m.load_state_dict(synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=0))
m = m.to("cuda", dtype)
frames = [f.to("cuda", dtype)[None].contiguous() for f in synthetic.clip_frames(24, 1080, 1920, seed=0)]
torch.cuda.synchronize()
m.pipeline_frames(True)
rec = [None] * 4
def loop(n):
    global rec
    with torch.no_grad():
        for i in range(n):
            rec = m(frames[i % 24], None, *rec, 0.25)[2:]
loop(30); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); loop(400); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
