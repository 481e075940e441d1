"""Where the PCIe-inclusive driver rate goes: clip length sweep (fixed cost vs per-chunk cost) on the 4K workload (GPU box)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvideomatting_amd import MattingNetwork, synthetic, inference as I
import bench
dev = torch.device("cuda:0")
model = MattingNetwork("mobilenetv3", "deep_guided_filter").eval()
model.load_state_dict(synthetic.make_params("mobilenetv3", "deep_guided_filter", seed=0))
model = model.to(dev, torch.float16)
H, W, chunk = 2160, 3840, int(os.environ.get("CHUNK", "4"))
base = torch.stack(list(synthetic.clip_frames(chunk, H, W, seed=0))).mul(255).byte()
for T in (24, 48, 96):
    clip = base.repeat(T // chunk, 1, 1, 1)
    for pin in (True, False):
        I.convert_video(model, I.TensorSource(clip[:2 * chunk]), downsample_ratio=0.125, output_type="video",
                        output_composition=bench._NullWriter(), seq_chunk=chunk, progress=False)
        src = I.TensorSource(clip).pin() if pin else I.TensorSource(clip)
        sink = bench._NullWriter()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        I.convert_video(model, src, downsample_ratio=0.125, output_type="video", output_composition=sink, seq_chunk=chunk, progress=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("T=%d chunk=%d pinned=%s: %.1f ms  %.0f fps" % (T, chunk, pin, dt * 1e3, T / dt), flush=True)
