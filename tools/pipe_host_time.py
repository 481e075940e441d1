import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, synthetic
dtype = torch.float16
m = MattingNetwork("mobilenetv3", "deep_guided_filter").eval()
m.load_state_dict(synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=0))
m = m.to("cuda", dtype)
frames = [f.to("cuda", dtype)[None].contiguous() for f in synthetic.clip_frames(24, 1080, 1920, seed=0)]
torch.cuda.synchronize()
def loop(n, pipe):
    m.pipeline_frames(pipe)
    rec = [None] * 4
    with torch.no_grad():
        for i in range(20):
            rec = m(frames[i % 24], None, *rec, 0.25)[2:]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            rec = m(frames[i % 24], None, *rec, 0.25)[2:]
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for name, s in (("default stream", None), ("torch.cuda.Stream()", torch.cuda.Stream()), ("high-priority stream", torch.cuda.Stream(priority=-1))):
    for pipe in (False, True):
        if s is None:
            h, t = loop(300, pipe)
        else:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                h, t = loop(300, pipe)
            torch.cuda.current_stream().wait_stream(s)
        print("%-22s pipeline %-5s host enqueue %.1f us/frame, total %.1f us/frame" % (name, pipe, h, t))
