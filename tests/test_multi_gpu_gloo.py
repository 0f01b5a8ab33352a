"""The N > 1 path on CPU: world_size-2 and -3 gloo process groups run the strip partition and the
gather that bench.py uses over RCCL, with each rank's strips rendered by the CPU oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import scenes
from raytracer_amd.multi_gpu import STRIP_ROWS, StripGather, local_rows, local_strip_count


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.orc import Oracle

        orc = Oracle("f64")
        scene = scenes.small_scene()
        g = StripGather(W, H, rank, world, torch.device("cpu"))
        buf = g.local_buffer()
        assert buf.shape[0] == local_rows(H, rank, world)
        # render this rank's strips (global strip = rank + k*world) into the compact buffer
        for k in range(local_strip_count(H, rank, world)):
            y0 = (rank + k * world) * STRIP_ROWS
            y1 = min(y0 + STRIP_ROWS, H)
            rgba, _, _ = orc.render(scene, W, H, 2, 5, seed=9, y0=y0, y1=y1, want_accum=False)
            buf[k * STRIP_ROWS:k * STRIP_ROWS + (y1 - y0)] = torch.from_numpy(rgba[y0:y1])
        frame = g.gather(buf)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 40, 27), (3, 33, 8), (2, 16, 5)])
def test_strip_gather_reassembles_frame(tmp_path, world, W, H, oracle64):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, out), nprocs=world, join=True)
    frame = np.load(out)
    ref, _, _ = oracle64.render(scenes.small_scene(), W, H, 2, 5, seed=9, want_accum=False)
    assert frame.shape == ref.shape
    assert np.array_equal(frame, ref)


def test_world_one_is_passthrough():
    g = StripGather(10, 9, 0, 1, torch.device("cpu"))
    b = g.local_buffer()
    assert b.shape == (9, 10, 4)
    assert g.gather(b) is b
