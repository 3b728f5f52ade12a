"""portal_amd/parallel.py -- one frame across the GPUs of a node.

The reference is a single-GPU OpenGL program; multi-GPU rendering is new in this build
(SURVEY.md 8e).  Pixels are independent, so the frame shards with no exchange while tracing:
rank g of G renders the 8-row blocks b with b % G == g (interleaved, because cost is spatially
clustered: portal interiors take many bounces, walls one).  The only collective is ONE gather
of the packed RGBA8 shards to the destination rank, followed by a strided copy that puts block
b = k*G + g back in place.  With the "nccl" backend (RCCL) `dist.gather` is a group of direct
send/recv pairs -- 7 point-to-point transfers over 7 xGMI links into rank 0, not a ring.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def blocks_of(height: int) -> int:
    return (height + 7) // 8


def shard_blocks_max(height: int, world: int) -> int:
    """Row blocks of the largest shard; every shard buffer is padded to this many blocks so that
    all ranks contribute the same byte count to the gather."""
    return (blocks_of(height) + world - 1) // world


def alloc_shard(height: int, width: int, world: int, device) -> torch.Tensor:
    return torch.zeros((shard_blocks_max(height, world) * 8, width, 4), dtype=torch.uint8, device=device)


class FrameGatherer:
    """Pre-allocated buffers for gathering one frame per step on `dst`."""

    def __init__(self, height: int, width: int, rank: int, world: int, device, dst: int = 0):
        self.h, self.w, self.rank, self.world, self.dst = height, width, rank, world, dst
        self.kmax = shard_blocks_max(height, world)
        self.parts = None
        self.full = None
        if rank == dst:
            self.full = torch.empty((self.kmax * world * 8, width, 4), dtype=torch.uint8, device=device)
            if world > 1:
                self.parts = [torch.empty((self.kmax * 8, width, 4), dtype=torch.uint8, device=device) for _ in range(world)]

    def gather(self, shard: torch.Tensor) -> Optional[torch.Tensor]:
        """shard: this rank's packed rows, shape (kmax*8, W, 4).  Returns the (H, W, 4) frame on dst."""
        if self.world == 1:
            return shard[: self.h]
        dist.gather(shard, self.parts, dst=self.dst)
        if self.rank != self.dst:
            return None
        # gathered[g][k*8:(k+1)*8] is frame block k*world + g: stack along a new axis 1 and flatten
        _interleave(self.parts, self.full, self.kmax, self.world, self.w)
        return self.full[: self.h]


def _interleave(parts, full, kmax, world, width):
    view = full.view(kmax, world, 8, width, 4)
    for g, p in enumerate(parts):
        view[:, g].copy_(p.view(kmax, 8, width, 4))
    return full
