"""The N > 1 path on CPU: two `gloo` ranks each render their interleaved row blocks (with the
host build standing in for the GPU launch), ONE gather assembles the frame on rank 0, and the
result equals the single-process frame byte for byte.  Exercises portal_amd/parallel.py, the
code bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, DEPTH = 70, 52, 20  # ragged on purpose: 7 row blocks (last one 4 rows), 70 px wide


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, W=W, H=H, scene_name="monoportal"):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import portal_amd as pa
        from oracle import host_build as hb
        from portal_amd import parallel

        scene = pa.Scene.from_file(pa.scene_path(scene_name))
        r = pa.SceneRenderer(scene, device=-1)
        r.set_option("render_depth", DEPTH)
        hk = hb.host_kernel_for(r, scene, W, H)
        frame = pa.Frame(W, H, rank, world)
        blocks = range(rank, parallel.blocks_of(H), world)
        rows = np.concatenate([np.arange(8 * b, min(H, 8 * b + 8)) for b in blocks])
        assert pa.shard_rows(frame) == len(rows)
        shard = parallel.alloc_shard(H, W, world, "cpu")
        shard[: len(rows)] = torch.from_numpy(hk.render(W, H, rows=rows, threads=1, rgba32f=False)["rgba8"])
        g = parallel.FrameGatherer(H, W, rank, world, "cpu", depth=2)
        # the double-buffered path bench.py uses: start frame 0, start frame 1 (a blank), finish both
        blank = parallel.alloc_shard(H, W, world, "cpu")
        w0 = g.gather_async(shard, 0)
        w1 = g.gather_async(blank, 1)
        full = g.finish(w0, shard, 0)
        other = g.finish(w1, blank, 1)
        if rank == 0:
            assert int(other.sum()) == 0  # (checked now: the returned frame is a view of the slot's buffer)
        again = g.gather(shard, 1)  # synchronous path, other slot (a collective: every rank calls it)
        if rank == 0:
            assert torch.equal(again, full)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_render_and_gather_equals_single_process(pa, tmp_path, world):
    from oracle import host_build as hb

    out_path = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    scene = pa.Scene.from_file(pa.scene_path("monoportal"))
    r = pa.SceneRenderer(scene, device=-1)
    r.set_option("render_depth", DEPTH)
    want = hb.host_kernel_for(r, scene, W, H).render(W, H, rgba32f=False)["rgba8"]
    assert np.array_equal(np.load(out_path), want)


@pytest.mark.parametrize("height, blocks_per_rank", [(2160, [34] * 6 + [33] * 2), (4320, [68] * 4 + [67] * 4), (2156, [34] * 6 + [33] * 2)])
def test_eight_ranks_on_the_geometry_of_the_baseline_frames(pa, tmp_path, height, blocks_per_rank):
    """the 8-rank shape before hardware exists.  BASELINE's 4K frames have 270 row blocks -- 33.75 per rank at G = 8: six ranks
    with 34 blocks, two with 33, every shard padded to 34 for the gather -- C5's 8K frame 540 (68 / 67); a 2156-row frame ends in a ragged block of
    four rows on rank 5.  Eight gloo ranks render their interleaved blocks (host build, a 24-pixel-wide frame of the same height), the
    double-buffered gather assembles them on rank 0, and the frame equals the single-process one byte for byte."""
    from oracle import host_build as hb
    from portal_amd import parallel

    world, width = 8, 24
    counts = [len(range(g, parallel.blocks_of(height), world)) for g in range(world)]
    assert counts == blocks_per_rank and parallel.shard_blocks_max(height, world) == max(blocks_per_rank)
    assert [pa.shard_rows(pa.Frame(width, height, g, world)) for g in range(world)] == [sum(min(8, height - 8 * b) for b in range(g, parallel.blocks_of(height), world)) for g in range(world)]
    out_path = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), out_path, width, height, "portal_in_portal"), nprocs=world, join=True)
    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    r = pa.SceneRenderer(scene, device=-1)
    r.set_option("render_depth", DEPTH)
    want = hb.host_kernel_for(r, scene, width, height).render(width, height, rgba32f=False)["rgba8"]
    got = np.load(out_path)
    assert got.shape == (height, width, 4) and np.array_equal(got, want)


def test_gatherer_single_rank_is_identity():
    from portal_amd import parallel

    shard = parallel.alloc_shard(20, 5, 1, "cpu")
    shard[:] = torch.arange(shard.numel(), dtype=torch.int64).reshape(shard.shape).to(torch.uint8)
    g = parallel.FrameGatherer(20, 5, 0, 1, "cpu")
    assert torch.equal(g.gather(shard), shard[:20])


def _one_rank_worker(rank, world, port):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from portal_amd import parallel

        tr = parallel.GatherTransport(H, W, 0, 1, "cpu", depth=2, collective_at_one_rank=True)
        assert tr.depth == 2 and tr.collective and tr.gatherer.slots[0][1] is not None  # the gather buffers exist although world == 1
        for slot, value in ((0, 5), (1, 9)):
            tr.shards[slot][:] = value
        works = [tr.submit(0), tr.submit(1)]             # dist.gather(..., async_op=True) of one shard to oneself
        for slot, value in ((0, 5), (1, 9)):
            got = tr.download(tr.finish(works[slot], slot))
            assert got.shape == (H, W, 4) and (got == value).all()
        plain = parallel.GatherTransport(H, W, 0, 1, "cpu")
        assert plain.depth == 1 and not plain.collective and plain.submit(0) is None  # the default at one rank: no collective at all
    finally:
        dist.destroy_process_group()


def test_gather_transport_can_run_its_collective_with_one_rank():
    """`collective_at_one_rank` (how a one-GPU box takes the real RCCL gather, tests/test_gpu_round2.py): the same switch over gloo."""
    mp.spawn(_one_rank_worker, args=(1, _free_port()), nprocs=1, join=True)


def _transport_worker(rank, world, port):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from portal_amd import parallel

        # no GPU here: rank 0 cannot allocate the shared frame.  The failure must reach EVERY rank (through the broadcast the
        # others are waiting in) so that bench.py's "did all ranks get it?" all-reduce still lines up -- nobody is left hanging.
        ok = 1
        try:
            parallel.PeerTransport(H, W, rank, world, "cpu", host_fence=True)
        except RuntimeError as e:
            assert "export failed" in str(e)
            ok = 0
        agreed = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        assert int(agreed.item()) == 0
        # the gather transport drives the same buffers as FrameGatherer: two frames through both slots
        tr = parallel.GatherTransport(H, W, rank, world, "cpu")
        assert tr.depth == 2 and tr.frame.in_place == 0 and tr.frame.rb_phase == rank
        for slot, value in ((0, 11), (1, 22)):
            tr.shards[slot][:] = value + rank
        works = [tr.submit(0), tr.submit(1)]
        frames = [tr.finish(works[0], 0), tr.finish(works[1], 1)]
        if rank == 0:
            for f, value in zip(frames, (11, 22)):
                got = tr.download(f)
                assert got.shape == (H, W, 4)
                for b in range(parallel.blocks_of(H)):  # block b came from rank b % world
                    assert (got[8 * b : 8 * b + 8] == value + b % world).all()
        else:
            assert frames == [None, None]
        tr.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.is_available(), reason="the no-device failure path")
def test_transports_on_cpu_ranks_fail_together_and_gather():
    mp.spawn(_transport_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_bench_gpus_n_launches_its_own_ranks_and_never_degrades_to_one():
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (how the driver starts `--gpus 1`): bench.py starts the N ranks itself.
    Here (no GPU) every rank must then refuse loudly; what may never happen is a silent one-rank run that prints n_gpus: 1.
    Over RCCL (the default backend) N > visible GPUs is refused before anything starts."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=dict(env, PTL_BENCH_BACKEND="gloo"))
    import torch

    if not torch.cuda.is_available():
        assert run.returncode != 0
        assert "starting 3 ranks under torch.distributed.run" in run.stderr
        assert run.stderr.count("bench.py needs a GPU") >= 1 and '"n_gpus"' not in run.stdout
    else:  # on a GPU box the same command line is the rehearsal itself
        import json

        assert run.returncode == 0, run.stderr[-2000:]
        assert json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 3
    many = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300, env=dict(env, PTL_BENCH_BACKEND="nccl"))
    assert many.returncode != 0 and "--gpus 64 but this node shows" in many.stderr and '"n_gpus"' not in many.stdout
    # a launcher whose WORLD_SIZE disagrees with --gpus (including WORLD_SIZE=1 for --gpus 8) is refused too
    odd = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300,
                         env=dict(env, PTL_BENCH_BACKEND="gloo", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert odd.returncode != 0 and '"n_gpus"' not in odd.stdout
