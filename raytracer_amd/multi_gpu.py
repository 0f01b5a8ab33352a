"""Multi-GPU assembly: one process per GPU, the image split into 8-row strips dealt
round-robin (strip s belongs to rank s % world), each rank renders its strips into a
compact buffer (see rt_params in include/rt_amd.h) and one gather brings the
finished RGBA8 strips to rank 0 (RCCL over xGMI when the tensors are on GPUs; the
same code runs over gloo on CPU tensors for tests).

The reference has no counterpart: its only parallelism is the OpenMP loop over
pixels (reference src/Camera.cpp:80-87).  Pixels are independent and the random
streams are keyed on the global pixel id, so the partition changes no byte.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

STRIP_ROWS = 8


def n_strips(height: int) -> int:
    return (height + STRIP_ROWS - 1) // STRIP_ROWS


def local_strip_count(height: int, rank: int, world: int) -> int:
    s = n_strips(height)
    return 0 if rank >= s else (s - rank + world - 1) // world


def local_rows(height: int, rank: int, world: int) -> int:
    """Rows of the compact buffer rank `rank` renders (== rt_output_rows)."""
    if world == 1:
        return height
    return local_strip_count(height, rank, world) * STRIP_ROWS


class StripGather:
    def __init__(self, width: int, height: int, rank: int, world: int, device, channels: int = 4,
                 dtype=torch.uint8, stage_on_host: bool = False):
        self.W, self.H, self.rank, self.world = width, height, rank, world
        self.device = device
        self.channels, self.dtype = channels, dtype
        self.max_strips = local_strip_count(height, 0, world)  # rank 0 never has fewer strips than others
        # stage_on_host: the process group cannot move device tensors (gloo): copy through the host
        self.stage_on_host = stage_on_host and world > 1 and torch.device(device).type != "cpu"
        if world > 1:
            shape = (self.max_strips * STRIP_ROWS, width, channels)
            self._send = torch.zeros(shape, dtype=dtype, device=device)
            if rank == 0:
                self._recv = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(world)]
                self._frame = torch.zeros((n_strips(height) * STRIP_ROWS, width, channels), dtype=dtype,
                                          device=device)

    def local_buffer(self) -> torch.Tensor:
        """The buffer this rank renders into (rows = rt_output_rows of its params)."""
        if self.world == 1:
            return torch.zeros((self.H, self.W, self.channels), dtype=self.dtype, device=self.device)
        rows = local_rows(self.H, self.rank, self.world)
        return self._send[:rows] if rows else self._send[:0]

    def gather(self, local: torch.Tensor):
        """Returns the assembled [H, W, C] frame on rank 0, None elsewhere."""
        if self.world == 1:
            return local
        if local.data_ptr() != self._send.data_ptr() and local.numel():
            self._send[: local.shape[0]].copy_(local)
        if self.stage_on_host:
            send = self._send.cpu()
            recv = [torch.empty_like(send) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(send, recv, dst=0)
            if self.rank == 0:
                for dst, src in zip(self._recv, recv):
                    dst.copy_(src)
        else:
            dist.gather(self._send, self._recv if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        total = n_strips(self.H)
        view = self._frame.view(total, STRIP_ROWS, self.W, self.channels)
        for r in range(self.world):
            cnt = local_strip_count(self.H, r, self.world)
            if cnt:
                view[r::self.world][:cnt].copy_(
                    self._recv[r].view(self.max_strips, STRIP_ROWS, self.W, self.channels)[:cnt])
        return self._frame[: self.H]
