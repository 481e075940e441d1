"""The N > 1 path of bench.py as the driver will run it the first time it meets an 8-GPU node.

CPU part (here): a rank that dies at start-up must end the whole run promptly with a non-zero exit code and its stderr
shown - not leave the other ranks in a barrier until the caller's time limit.  GPU part (-m gpu, one-GPU box): the bare
`--gpus 2` form end to end with the real model, both ranks rehearsed on cuda:0 over gloo (RVM_BENCH_REHEARSE=1; the data
path has no collective, so this is the same code the 8-GPU run executes, minus RCCL's barrier)."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU: every rank then fails at start-up")
def test_a_rank_that_dies_at_startup_ends_the_run_with_an_error():
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-also"], capture_output=True, timeout=300)
    assert r.returncode != 0
    assert time.time() - t0 < 240
    err = r.stderr.decode()
    assert "exited with code" in err and "stderr (tail)" in err, err[-2000:]
    assert r.stdout.decode().strip() == ""                      # no result line from a failed run


@pytest.mark.gpu
def test_two_ranks_rehearsed_on_one_gpu_with_the_real_model():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, RVM_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "3",
                        "--no-cpu-baseline", "--no-also", "--src-ring", "2"], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 5
    per = line["fps_per_rank"]
    assert len(per) == 2 and all(p > 0 for p in per)
    # whole-job figure = all frames / slowest rank = 2 x the slower rank's own rate
    assert abs(line["value"] - 2 * min(per)) <= 0.02 * line["value"], (line["value"], per)
    assert line["value"] <= sum(per) * 1.001
