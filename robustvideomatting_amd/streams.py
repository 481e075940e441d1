"""Stream sharding for multi-GPU inference.

A video stream is an independent recurrence (its r1..r4 live with the caller,
reference inference.py:121-129), so a node scales by pinning stream i to GPU
i mod N for life.  There is no collective on the data path; the only
communication is the host-side reduction of per-rank timings.
"""


def shard_streams(n_streams, world_size, rank):
    """Stream ids owned by `rank`: i with i % world_size == rank."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return [i for i in range(n_streams) if i % world_size == rank]


class StreamRunner:
    """Runs the streams owned by one rank.  Streams of equal shape are batched along B, so one
    forward advances all of them one frame (their states are rows of the same r1..r4)."""

    def __init__(self, model, stream_ids, downsample_ratio):
        self.model = model
        self.stream_ids = list(stream_ids)
        self.ds = downsample_ratio
        self.rec = [None] * 4
        self.frames_done = 0

    def step(self, src):
        """src: [len(stream_ids), 3, H, W] - the next frame of every owned stream, or
        [len(stream_ids), T, 3, H, W] - their next T frames (one chunked call)."""
        assert src.shape[0] == len(self.stream_ids)
        out = self.model(src, None, *self.rec, self.ds)
        self.rec = out[2:]
        self.frames_done += src.shape[0] * (src.shape[1] if src.ndim == 5 else 1)
        return out[0], out[1]


def aggregate_throughput(frames_local, seconds_local, dist=None, device=None):
    """Whole-job frames/s = sum of frames over ranks / max of elapsed time over ranks."""
    import torch
    if dist is None or not dist.is_initialized():
        return frames_local / seconds_local, frames_local, seconds_local
    f = torch.tensor([float(frames_local)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return f.item() / t.item(), f.item(), t.item()
