#!/usr/bin/env python3
"""tools/fb_store_bench.py -- HBM write bandwidth of the renderer's framebuffer store pattern alone
(portal_amd/csrc/kernels/fb_store.hip), through the same C ABI.  Prints one JSON line per variant."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

SRC = open(os.path.join(pa.REPO_ROOT, "portal_amd", "csrc", "kernels", "fb_store.hip")).read()
NAMES = {0: "rgba8 via 32x8 LDS transpose (renderer)", 1: "rgba32f float4 per lane (parity buffer)", 2: "rgba8 direct 8x8 tile (32 B segments)", 3: "linear 16 B/lane stream",
         4: "rgba8 via 32x8 LDS transpose, 16 B per lane (wave 0 stores the block)", 5: "variant 4 from a persistent grid", 6: "linear 16 B/lane nontemporal, persistent grid",
         7: "empty launch (floor of the timing method)"}
UNIFORMS = [("seed_u", pa.PTL_I32, 0), ("pad_u", pa.PTL_I32, 4), ("real_w_u", pa.PTL_I32, 8), ("real_h_u", pa.PTL_I32, 12)]


def measure(k, frame, buf, stream, torch, n_back_to_back=20):
    """(ms of ONE launch between two events, ms per launch of n launches back to back between two events)"""
    def launch(timed):
        ms = C.c_float()
        rc = pa.lib().ptl_kernel_render(k._h, C.byref(frame), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), None, C.c_void_p(stream.cuda_stream), C.byref(ms) if timed else None)
        assert rc == 0, pa.lib().ptl_last_error()
        return ms.value

    single = [launch(True) for _ in range(14)][2:]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    runs = []
    for _ in range(5):
        e0.record(stream)
        for _ in range(n_back_to_back):
            launch(False)
        e1.record(stream)
        e1.synchronize()
        runs.append(e0.elapsed_time(e1) / n_back_to_back)
    return float(np.median(single)), float(np.median(runs))


def measure_in_flight(k, frame, bufs, streams, torch, n=400):
    """ms per launch of n launches queued round-robin on len(streams) non-blocking streams (a target per stream), wall time between two
    synchronisations -- round 6: what the store pattern sustains with two frames in flight (the renderer's `lane_fence` 0 lanes do this)."""
    import time

    def launch(j):
        rc = pa.lib().ptl_kernel_render(k._h, C.byref(frame), C.c_void_p(bufs[j].data_ptr()), C.c_void_p(bufs[j].data_ptr()), None, C.c_void_p(streams[j]), None)
        assert rc == 0, pa.lib().ptl_last_error()

    best = 1e9
    for _ in range(4):
        for j in range(len(streams)):
            launch(j)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            launch(i % len(streams))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


if __name__ == "__main__":
    import torch

    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
    dev = torch.device("cuda", 0)
    buf = torch.empty(W * H * 16, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    bufs = [buf, torch.empty_like(buf)]
    lanes = []
    pa.lib().ptl_stream_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    for _ in range(2):
        h = C.c_void_p()
        assert pa.lib().ptl_stream_create(0, C.byref(h)) == 0
        lanes.append(int(h.value))
    for variant, grids in ((0, [None]), (1, [None]), (2, [None]), (3, [None]), (4, [None]), (5, [256, 512, 1024, 2048, 4096]), (6, [256, 512, 1024, 2048, 4096]), (7, [None])):
        k = pa.Kernel(SRC, UNIFORMS, 16, device=0, defines=[f"PTL_FB_VARIANT={variant}"])
        k.set_uniform("seed_u", pa.PTL_I32, 12345)
        k.set_uniform("real_w_u", pa.PTL_I32, W)
        k.set_uniform("real_h_u", pa.PTL_I32, H)
        for wgs in grids:
            # persistent variants: the launch geometry of a small fake frame (wgs workgroups: 32 x wgs/32 blocks), the real size in the uniforms
            frame = pa.Frame(W, H, 0, 1) if wgs is None else pa.Frame(32 * 32, 8 * (wgs // 32), 0, 1)
            ms, ms_b2b = measure(k, frame, buf, stream, torch)
            nbytes = 0 if variant == 7 else W * H * (16 if variant == 1 else 4)
            ms_1 = measure_in_flight(k, frame, bufs[:1], lanes[:1], torch)
            ms_2 = measure_in_flight(k, frame, bufs, lanes, torch)
            print(json.dumps({"variant": NAMES[variant], **({"workgroups": wgs} if wgs else {}), "frame": f"{W}x{H}", "bytes": nbytes, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1),
                              "frac_of_8TB/s": round(nbytes / ms / 1e6 / 8000, 4), "ms_back_to_back": round(ms_b2b, 4),
                              "frac_of_8TB/s_back_to_back": round(nbytes / ms_b2b / 1e6 / 8000, 4),
                              "ms_queued_one_stream": round(ms_1, 4), "frac_of_8TB/s_queued_one_stream": round(nbytes / ms_1 / 1e6 / 8000, 4),
                              "ms_two_in_flight": round(ms_2, 4), "frac_of_8TB/s_two_in_flight": round(nbytes / ms_2 / 1e6 / 8000, 4)}), flush=True)
