#!/usr/bin/env python3
"""tools/average_bench.py -- HBM roofline of the motion-blur averaging kernel (ptl_average_images,
portal_amd/csrc/kernels/average_images.hip) through the C ABI.  Algorithmic bytes per launch = (4 N + 4) W H
(N RGBA8 sub-frames read once, one RGBA8 frame written).  One JSON line per (N, size); HIP-event times per launch."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

if __name__ == "__main__":
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for (w, h) in ((3840, 2160), (7680, 4320)):
        for n in (2, 4, 8, 16):
            g = torch.Generator(device="cuda").manual_seed(n)
            frames = [torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev, generator=g) for _ in range(n)]
            out = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
            ptrs = [f.data_ptr() for f in frames]
            times = [pa.average_images_device(ptrs, out.data_ptr(), w, h, stream=stream, timed=True) for _ in range(30)]
            ms = float(np.median(times[5:]))
            nbytes = (4 * n + 4) * w * h
            print(json.dumps({"kernel": "ptl_average_images_kernel", "frame": f"{w}x{h}", "subframes": n, "bytes": nbytes, "ms": round(ms, 4),
                              "GB/s": round(nbytes / ms / 1e6, 1), "frac_of_8TB/s": round(nbytes / ms / 1e6 / 8000, 4)}), flush=True)
