import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, synthetic
dtype = torch.float16
for (H, W, ds) in ((144, 256, 0.25), (540, 960, 0.25), (1080, 1920, 0.25)):
    m = MattingNetwork("mobilenetv3", "deep_guided_filter").eval()
    m.load_state_dict(synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=0))
    m = m.to("cuda", dtype)
    frames = [f.to("cuda", dtype)[None].contiguous() for f in synthetic.clip_frames(24, H, W, seed=0)]
    torch.cuda.synchronize()
    def loop(n, pipe):
        m.pipeline_frames(pipe)
        rec = [None] * 4
        with torch.no_grad():
            for i in range(30):
                rec = m(frames[i % 24], None, *rec, ds)[2:]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                rec = m(frames[i % 24], None, *rec, ds)[2:]
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
    for pipe in (False, True):
        h, t = loop(400, pipe)
        print("%dx%d pipeline %-5s host enqueue %.1f us/frame, total %.1f us/frame" % (W, H, pipe, h, t))
