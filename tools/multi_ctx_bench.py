"""Several contexts on one GPU, each driven from its own host thread on its own HIP stream:
total frames/s for C contexts x B batched streams each.

    python tools/multi_ctx_bench.py --contexts 2 --batch 4 [--steps 100] [--graph]
"""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustvideomatting_amd import MattingNetwork, synthetic   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--H", type=int, default=1080)
    ap.add_argument("--W", type=int, default=1920)
    ap.add_argument("--ds", type=float, default=0.25)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    params = synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=0)
    models, srcs, streams = [], [], []
    for c in range(a.contexts):
        m = MattingNetwork().eval()
        m.load_state_dict(params)
        m = m.to(dev, torch.float16)
        m.enable_graph(a.graph)
        models.append(m)
        srcs.append(synthetic.make_clip(1, a.H, a.W, seed=c, B=a.batch)[:, 0].to(dev, torch.float16).contiguous())
        streams.append(torch.cuda.Stream(dev))
    start = threading.Barrier(a.contexts + 1)
    done = threading.Barrier(a.contexts + 1)

    def work(i):
        rec = [None] * 4
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(a.warmup):
                out = models[i](srcs[i], None, *rec, a.ds)
                rec = out[2:]
            streams[i].synchronize()
            start.wait()
            for _ in range(a.steps):
                out = models[i](srcs[i], None, *rec, a.ds)
                rec = out[2:]
            streams[i].synchronize()
            done.wait()

    th = [threading.Thread(target=work, args=(i,)) for i in range(a.contexts)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    done.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    frames = a.contexts * a.batch * a.steps
    print("contexts %d x batch %d  graph %d: %.0f frames/s  (%.3f ms per round of %d frames)"
          % (a.contexts, a.batch, int(a.graph), frames / dt, dt / a.steps * 1e3, a.contexts * a.batch))


if __name__ == "__main__":
    main()
