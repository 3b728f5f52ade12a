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

    def __init__(self, height: int, width: int, rank: int, world: int, device, dst: int = 0, depth: int = 1, stage_through_host: bool = False,
                 collective_at_one_rank: bool = False):
        self.h, self.w, self.rank, self.world, self.dst = height, width, rank, world, dst
        # world == 1 needs no collective and normally gets none.  `collective_at_one_rank` runs it anyway (a gather of one shard to
        # oneself): the way a box with ONE GPU takes the real `dist.gather` over RCCL, buffers, the process group's stream and the
        # de-interleave through the very code the N-rank bench runs (tests/test_gpu_round2.py)
        self.shortcut = world == 1 and not collective_at_one_rank
        # a backend without device-to-device gather (gloo, used to rehearse the multi-rank control flow on one GPU): the shard
        # goes through host memory, synchronously
        self.stage_through_host = stage_through_host
        self.kmax = shard_blocks_max(height, world)
        self.slots = []
        for _ in range(depth):
            full = gathered = None
            if rank == dst:
                full = torch.empty((self.kmax * world * 8, width, 4), dtype=torch.uint8, device=device)
                if not self.shortcut:
                    gathered = torch.empty((world, self.kmax * 8, width, 4), dtype=torch.uint8, device=device)
            self.slots.append((full, gathered))

    def _assemble(self, slot: int) -> torch.Tensor:
        full, gathered = self.slots[slot]
        # gathered[g, k*8:(k+1)*8] is frame block k*world + g: one strided copy puts every block in place
        full.view(self.kmax, self.world, 8, self.w, 4).copy_(gathered.view(self.world, self.kmax, 8, self.w, 4).transpose(0, 1))
        return full[: self.h]

    def gather(self, shard: torch.Tensor, slot: int = 0) -> Optional[torch.Tensor]:
        """shard: this rank's packed rows, shape (kmax*8, W, 4).  Returns the (H, W, 4) frame on dst."""
        if self.shortcut:
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
        if self.shortcut:
            return None
        gathered = self.slots[slot][1]
        if self.stage_through_host:
            self._gather_through_host(shard, gathered)
            return _Done()
        return dist.gather(shard, list(gathered.unbind(0)) if self.rank == self.dst else None, dst=self.dst, async_op=True)

    def finish(self, work, shard: torch.Tensor, slot: int = 0) -> Optional[torch.Tensor]:
        if self.shortcut:
            return shard[: self.h]
        work.wait()
        return self._assemble(slot) if self.rank == self.dst else None


class PeerFrames:
    """Full-frame buffers in the destination GPU's HBM that every rank renders into directly.

    The destination rank allocates `depth` frames (whole hipMalloc allocations), exports them (`ptl_ipc_export`) and every other
    rank maps them (`ptl_ipc_open`); a rank then launches its row blocks with `Frame(..., in_place=1)` and the kernel's 128-byte
    row stores travel over xGMI into the destination's memory -- no gather, no staging shard, no de-interleave copy.  What is
    left of the collective is a *fence*: a one-element all-reduce enqueued behind the launch on every rank, so that once it has
    completed on the destination every rank's kernel (and with it its stores) has completed.

    `host_fence` is for a backend without device collectives (gloo: rehearsing N ranks on one GPU): synchronise + barrier."""

    def __init__(self, height: int, width: int, rank: int, world: int, device, dst: int = 0, depth: int = 2, host_fence: bool = False):
        import portal_amd as pa

        self._pa = pa
        self.h, self.w, self.rank, self.world, self.dst = height, width, rank, world, dst
        self.device = torch.device(device)
        self.host_fence = host_fence
        # whole 8-row blocks: a strided block copy (CopyTransport) of a ragged last block stays inside the allocation
        self.nbytes = blocks_of(height) * 8 * width * 4
        index = self.device.index or 0
        self.owned, self.ptrs = [], []
        handles = [None]
        if rank == dst:
            try:
                self.owned = [pa.device_alloc(self.nbytes, index) for _ in range(depth)]
                self.ptrs = list(self.owned)
                handles = [[pa.ipc_export(p) for p in self.owned]]
            except Exception as e:  # the others are waiting in the broadcast: tell them instead of leaving them there
                handles = [f"export failed: {e}"]
        if world > 1:
            dist.broadcast_object_list(handles, src=dst)
            if isinstance(handles[0], str):
                self._release()
                raise RuntimeError(handles[0])
            if rank != dst:
                try:
                    for h in handles[0]:
                        self.ptrs.append(pa.ipc_open(h, index))
                except Exception:
                    self._release()
                    raise
        elif isinstance(handles[0], str):
            raise RuntimeError(handles[0])
        self.token = torch.zeros(1, dtype=torch.int32, device=self.device)

    def frame(self, rank: Optional[int] = None, world: Optional[int] = None):
        """The ptl_frame a rank launches with: its interleaved row blocks, stored where they belong."""
        return self._pa.Frame(self.w, self.h, self.rank if rank is None else rank, self.world if world is None else world, 1)

    def fence_async(self):
        """Enqueue the completion fence behind what this rank has launched on the current stream."""
        if self.world == 1:
            return None
        if self.host_fence:
            torch.cuda.synchronize(self.device)
            dist.barrier()
            return _Done()
        return dist.all_reduce(self.token, op=dist.ReduceOp.MAX, async_op=True)  # MAX: the token stays 0 however many frames pass

    def finish(self, work) -> None:
        if work is not None:
            work.wait()

    def download(self, slot: int = 0):
        """The assembled (H, W, 4) frame as a numpy array (destination rank only; waits for the current stream)."""
        if self.rank != self.dst:
            return None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        return self._pa.device_download(self.ptrs[slot], self.h * self.w * 4, stream).reshape(self.h, self.w, 4)

    def close(self) -> None:
        if self.world > 1:
            torch.cuda.synchronize(self.device)
            dist.barrier()  # nobody unmaps or frees while a peer may still be storing
        self._release()

    def _release(self) -> None:
        if self.rank != self.dst:
            for p in self.ptrs:
                self._pa.ipc_close(p)
        for p in self.owned:
            self._pa.device_free(p)
        self.ptrs, self.owned = [], []


class GatherTransport:
    """One frame per step through packed shards + ONE gather to `dst` + the de-interleave copy (FrameGatherer)."""

    name = "rccl-gather"

    def __init__(self, height, width, rank, world, device, depth=2, stage_through_host=False, collective_at_one_rank=False):
        import portal_amd as pa

        self.collective = world > 1 or collective_at_one_rank
        self.depth = depth if self.collective else 1
        self.rank, self.world, self.h = rank, world, height
        self.frame = pa.Frame(width, height, rank, world)
        self.shards = [alloc_shard(height, width, world, device) for _ in range(self.depth)]
        self.gatherer = FrameGatherer(height, width, rank, world, device, depth=self.depth, stage_through_host=stage_through_host,
                                      collective_at_one_rank=collective_at_one_rank)

    def out_ptr(self, slot):
        return self.shards[slot].data_ptr()

    def submit(self, slot):
        return self.gatherer.gather_async(self.shards[slot], slot) if self.collective else None

    def finish(self, work, slot):
        """The assembled frame (a device tensor) on the destination rank, None elsewhere."""
        return self.gatherer.finish(work, self.shards[slot], slot)

    def download(self, assembled):
        return assembled.cpu().numpy() if assembled is not None else None

    def close(self):
        pass


class CopyTransport:
    """One frame per step through a packed shard in the rank's OWN memory + ONE strided copy into the destination's frame (mapped
    through HIP IPC like PeerTransport) + the fence.  The copy is a hipMemcpy2DAsync behind the kernel on the same stream: source
    pitch one 8-row block, destination pitch `world` blocks -- the SDMA engine moves the shard over this rank's xGMI link and
    de-interleaves it on the way.  Against `p2p-stores` the kernel writes local HBM; against `rccl-gather` there is no collective
    launch and no second pass over the frame on rank 0."""

    name = "p2p-copy"

    def __init__(self, height, width, rank, world, device, depth=2, host_fence=False):
        import portal_amd as pa

        self._pa = pa
        self.depth = depth
        self.rank, self.world, self.h, self.w = rank, world, height, width
        self.device = torch.device(device)
        self.frames = PeerFrames(height, width, rank, world, device, depth=depth, host_fence=host_fence)
        self.frame = pa.Frame(width, height, rank, world)  # packed shard layout
        self.shards = [alloc_shard(height, width, world, device) for _ in range(depth)]
        blocks = blocks_of(height)
        self.my_blocks = (blocks - rank + world - 1) // world if blocks > rank else 0
        self.failed = None  # the first error of a copy (the fence is still taken: the other ranks are waiting in it)

    def out_ptr(self, slot):
        return self.shards[slot].data_ptr()

    def submit(self, slot):
        pitch = self.w * 4
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.my_blocks and self.failed is None:
            try:
                self._pa.device_copy2d_async(self.frames.ptrs[slot] + self.rank * 8 * pitch, self.world * 8 * pitch, self.shards[slot].data_ptr(), 8 * pitch,
                                             8 * pitch, self.my_blocks, stream)
            except Exception as e:  # e.g. a runtime that refuses strided copies into an IPC mapping: reported by the caller, never a hang
                self.failed = str(e)[:200]
        return self.frames.fence_async()

    def finish(self, work, slot):
        self.frames.finish(work)
        return slot if self.rank == self.frames.dst else None

    def download(self, assembled):
        return self.frames.download(assembled) if assembled is not None else None

    def close(self):
        self.frames.close()

    def abandon(self):
        self.frames._release()


class PeerTransport:
    """One frame per step through stores into the destination's HBM + a fence (PeerFrames)."""

    name = "p2p-stores"

    def __init__(self, height, width, rank, world, device, depth=2, host_fence=False):
        self.depth = depth
        self.rank, self.world = rank, world
        self.frames = PeerFrames(height, width, rank, world, device, depth=depth, host_fence=host_fence)
        self.frame = self.frames.frame()

    def out_ptr(self, slot):
        return self.frames.ptrs[slot]

    def submit(self, slot):
        return self.frames.fence_async()

    def finish(self, work, slot):
        self.frames.finish(work)
        return slot if self.rank == self.frames.dst else None

    def download(self, assembled):
        return self.frames.download(assembled) if assembled is not None else None

    def close(self):
        self.frames.close()

    def abandon(self):
        """Release without the collective shutdown (set-up failed on another rank; nothing has been launched)."""
        self.frames._release()
