"""CPU, world_size 2 over gloo: the N>1 path of bench.py - stream sharding and timing reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robustvideomatting_amd import streams


class _FakeModel:
    """Stands in for MattingNetwork: a per-stream recurrence with the same call/return shape."""

    def __call__(self, src, trimap, r1, r2, r3, r4, ds):
        m = src.mean((1, 2, 3)).view(-1, 1, 1, 1)
        rec = [m if r is None else 0.5 * r + m for r in (r1, r2, r3, r4)]
        return [src, m, *rec]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = streams.shard_streams(n_streams, world, rank)
    runner = streams.StreamRunner(_FakeModel(), ids, 0.25)
    g = torch.Generator().manual_seed(0)
    clips = torch.rand(n_streams, T, 3, 8, 8, generator=g)          # same clips on every rank
    for t in range(T):
        runner.step(clips[ids, t])
    secs = 1.0 + rank                                               # rank 1 is the slow one
    fps, frames, tmax = streams.aggregate_throughput(runner.frames_done, secs, dist)
    q.put((rank, ids, runner.rec[0].flatten().tolist(), fps, frames, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    world, n_streams, T = 2, 5, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids0, ids1 = res[0][1], res[1][1]
    assert sorted(ids0 + ids1) == list(range(n_streams)) and not set(ids0) & set(ids1)
    assert ids0 == [0, 2, 4] and ids1 == [1, 3]
    # every rank reports the same whole-job figure: all frames / slowest rank
    for r in res:
        assert r[4] == n_streams * T and r[5] == 2.0 and abs(r[3] - n_streams * T / 2.0) < 1e-9
    # per-stream state equals the single-process recurrence of that stream (nothing leaked across streams)
    g = torch.Generator().manual_seed(0)
    clips = torch.rand(n_streams, T, 3, 8, 8, generator=g)
    for r in res:
        for k, sid in enumerate(r[1]):
            h = None
            for t in range(T):
                m = clips[sid, t].mean()
                h = m if h is None else 0.5 * h + m
            assert abs(r[2][k] - h.item()) < 1e-6


def test_shard_edge_cases():
    assert streams.shard_streams(0, 4, 1) == []
    assert streams.shard_streams(3, 8, 7) == []
    assert streams.shard_streams(64, 8, 3) == list(range(3, 64, 8))
    try:
        streams.shard_streams(4, 2, 2)
        assert False
    except ValueError:
        pass
