#!/usr/bin/env python3
"""tools/small_frames.py -- what does a SMALL frame cost per pixel?

The same baked kernel at 960x540 ... 7680x4320: kernel time (HIP events inside the library), ns per pixel, and the wall time per frame of
a run of launches queued back to back on one stream (no host wait between them) -- the regime of the video pipeline, where a frame's
blur / aa sub-frames follow each other.  If ns/pixel is flat, a launch that batches sub-frames (grid.z) has nothing to win; the
difference between the smallest and the largest frame's ns/pixel is the whole prize.

    python tools/small_frames.py [scene ...]          # on the GPU box; JSON lines
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

SIZES = [(960, 540), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)]
DEPTH = {"monoportal": 20, "portal_in_portal": 40, "triple_portal": 40}


def main():
    scenes = sys.argv[1:] or ["monoportal", "portal_in_portal"]
    flags = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    for name in scenes:
        r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(name)), device=0, flags=flags)
        r.set_option("render_depth", DEPTH.get(name, 40))
        buf = pa.device_alloc(7680 * 4320 * 4)
        try:
            for w, h in SIZES:
                frame = pa.Frame(w, h, 0, 1, 0)
                for _ in range(3):
                    r.draw_device(frame, out_rgba8=buf, timed=True)
                ms = sorted(r.draw_device(frame, out_rgba8=buf, timed=True) for _ in range(15))
                n = 200 if w * h < 4_000_000 else 50
                pa.device_download(buf, 4)
                t0 = time.perf_counter()
                for _ in range(n):
                    r.draw_device(frame, out_rgba8=buf)
                pa.device_download(buf, 4)  # waits for the stream
                wall = (time.perf_counter() - t0) / n * 1e3
                print(json.dumps({"scene": name, "size": f"{w}x{h}", "workgroups": ((w + 31) // 32) * ((h + 7) // 8), "kernel_ms_median": round(ms[7], 4), "kernel_ms_min": round(ms[0], 4),
                                  "ns_per_pixel": round(ms[7] * 1e6 / (w * h), 4), "back_to_back_ms_per_frame": round(wall, 4),
                                  "back_to_back_ns_per_pixel": round(wall * 1e6 / (w * h), 4)}), flush=True)
        finally:
            pa.device_free(buf)


if __name__ == "__main__":
    main()
