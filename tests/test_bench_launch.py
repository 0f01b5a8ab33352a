"""`python bench.py --gpus N` called plainly (no launcher, no WORLD_SIZE) starts its N ranks itself.

The multi-GPU counterpart of the reference's OpenMP team (reference src/Camera.cpp:69-88) is one process per
GPU; whoever measures the 1 -> 8 curve calls the benchmark without torch.distributed.run.  On a one-GPU box the
N = 2 path is rehearsed with both ranks on GPU 0 and the gather over gloo: same strips, same assembly, and the
assembled frame must hash to the N = 1 frame.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--workload", "c3", "--spp", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"]


def run_bench(extra, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith('{"metric"')]


def test_plain_multi_gpu_call_starts_its_ranks_even_without_a_gpu():
    """Without a GPU the ranks are still started (and each of them then refuses to run: there is no CPU
    fallback); the launcher hands their failure back as its exit code instead of asking for torch.distributed.run."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test below on a GPU box")
    res = run_bench(["--gpus", "2", "--backend", "gloo", "--share-gpu"] + SMALL)
    assert res.returncode != 0
    assert "bench.py needs a GPU" in res.stderr
    assert "must be launched with" not in res.stderr
    assert json_lines(res.stdout) == []


@pytest.mark.gpu
def test_plain_two_rank_call_assembles_the_one_rank_frame():
    one = run_bench(["--gpus", "1"] + SMALL)
    assert one.returncode == 0, one.stderr[-2000:]
    two = run_bench(["--gpus", "2", "--share-gpu", "--backend", "gloo"] + SMALL)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = json_lines(one.stdout), json_lines(two.stdout)
    assert len(a) == 1 and len(b) == 1, (one.stdout, two.stdout)
    a, b = a[0], b[0]
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert b["steps"] == 1 and b["warmup"] == 0
    assert len(b["per_rank"]["kernel_ms"]) == 2 and len(b["per_rank"]["gather_ms"]) == 2
    assert all(k > 0 for k in b["per_rank"]["kernel_ms"])
    assert a["frame_sha256_seed0"] == b["frame_sha256_seed0"]
    assert a["config"]["rays_per_sample"] == b["config"]["rays_per_sample"]
