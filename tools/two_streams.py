#!/usr/bin/env python3
"""tools/two_streams.py -- VERDICT r5 #8: does frame n + 1's ramp overlap frame n's tail when consecutive frames of an unchanged state go to
ALTERNATING streams (one kernel, one uniform block, a target buffer per stream, no event between the streams)?  The upper bound of what any
two-lane scheme can buy a stream of small frames; `concurrent_draws` (clones + cross-stream waits, round 4) measured nothing.
    python tools/two_streams.py            # GPU box; JSON lines: wall time per frame of 400 queued launches on 1, 2, 3 streams, frames compared"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

CASES = [("monoportal", 1920, 1080, 20), ("monoportal", 960, 540, 20), ("portal_in_portal", 3840, 2160, 40), ("portal_in_portal", 1920, 1080, 40), ("triple_portal", 3840, 2160, 40)]


def main():
    lib = pa.lib()
    lib.ptl_stream_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.ptl_stream_destroy.argtypes = [C.c_void_p]
    flags = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    streams = []
    for _ in range(3):
        s = C.c_void_p()
        assert lib.ptl_stream_create(0, C.byref(s)) == 0
        streams.append(int(s.value))
    for name, w, h, depth in CASES:
        r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(name)), device=0, flags=flags)
        r.set_option("render_depth", depth)
        frame = pa.Frame(w, h, 0, 1, 0)
        bufs = [pa.device_alloc(w * h * 4) for _ in range(3)]
        for b in bufs:
            r.draw_device(frame, out_rgba8=b, timed=True)
        want = pa.device_download(bufs[0], w * h * 4)
        out = {"scene": name, "size": f"{w}x{h}"}
        for lanes in (1, 2, 3):
            n = 600
            for k in range(30):
                r.draw_device(frame, out_rgba8=bufs[k % lanes], stream=streams[k % lanes])
            for s in streams:
                pa.device_download(bufs[0], 4, stream=s)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for k in range(n):
                    r.draw_device(frame, out_rgba8=bufs[k % lanes], stream=streams[k % lanes])
                for s in streams[:lanes]:
                    pa.device_download(bufs[0], 4, stream=s)  # waits for that stream
                best = min(best, (time.perf_counter() - t0) / n * 1e3)
            same = all(np.array_equal(pa.device_download(bufs[k], w * h * 4), want) for k in range(lanes))
            out[f"ms_per_frame_{lanes}_stream" + ("s" if lanes > 1 else "")] = round(best, 4)
            out[f"frames_identical_{lanes}"] = bool(same)
        print(json.dumps(out), flush=True)
        for b in bufs:
            pa.device_free(b)


if __name__ == "__main__":
    main()
