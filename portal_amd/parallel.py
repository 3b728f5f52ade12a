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


class _Done:
    """A finished collective (the host-staged path is synchronous)."""

    def wait(self):
        return True


class FrameGatherer:
    """Pre-allocated buffers for gathering one frame per step on `dst`.

    `depth` > 1 double-buffers the gather target so that `gather_async` of frame n can overlap the
    tracing of frame n+1 (the collective runs on the process group's own stream)."""

    def __init__(self, height: int, width: int, rank: int, world: int, device, dst: int = 0, depth: int = 1, stage_through_host: bool = False):
        self.h, self.w, self.rank, self.world, self.dst = height, width, rank, world, dst
        # a backend without device-to-device gather (gloo, used to rehearse the multi-rank control flow on one GPU): the shard
        # goes through host memory, synchronously
        self.stage_through_host = stage_through_host
        self.kmax = shard_blocks_max(height, world)
        self.slots = []
        for _ in range(depth):
            full = gathered = None
            if rank == dst:
                full = torch.empty((self.kmax * world * 8, width, 4), dtype=torch.uint8, device=device)
                if world > 1:
                    gathered = torch.empty((world, self.kmax * 8, width, 4), dtype=torch.uint8, device=device)
            self.slots.append((full, gathered))

    def _assemble(self, slot: int) -> torch.Tensor:
        full, gathered = self.slots[slot]
        # gathered[g, k*8:(k+1)*8] is frame block k*world + g: one strided copy puts every block in place
        full.view(self.kmax, self.world, 8, self.w, 4).copy_(gathered.view(self.world, self.kmax, 8, self.w, 4).transpose(0, 1))
        return full[: self.h]

    def gather(self, shard: torch.Tensor, slot: int = 0) -> Optional[torch.Tensor]:
        """shard: this rank's packed rows, shape (kmax*8, W, 4).  Returns the (H, W, 4) frame on dst."""
        if self.world == 1:
            return shard[: self.h]
        gathered = self.slots[slot][1]
        if self.stage_through_host:
            self._gather_through_host(shard, gathered)
        else:
            dist.gather(shard, list(gathered.unbind(0)) if self.rank == self.dst else None, dst=self.dst)
        return self._assemble(slot) if self.rank == self.dst else None

    def _gather_through_host(self, shard: torch.Tensor, gathered: Optional[torch.Tensor]) -> None:
        host = shard.cpu()
        parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
        dist.gather(host, parts, dst=self.dst)
        if self.rank == self.dst:
            gathered.copy_(torch.stack(parts))

    def gather_async(self, shard: torch.Tensor, slot: int = 0):
        """Start the gather; returns a handle for `finish`."""
        if self.world == 1:
            return None
        gathered = self.slots[slot][1]
        if self.stage_through_host:
            self._gather_through_host(shard, gathered)
            return _Done()
        return dist.gather(shard, list(gathered.unbind(0)) if self.rank == self.dst else None, dst=self.dst, async_op=True)

    def finish(self, work, shard: torch.Tensor, slot: int = 0) -> Optional[torch.Tensor]:
        if self.world == 1:
            return shard[: self.h]
        work.wait()
        return self._assemble(slot) if self.rank == self.dst else None
