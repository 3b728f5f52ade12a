#!/usr/bin/env python3
"""tools/average_variants.py -- build-time variants of the averaging kernel (unroll depth, streaming vs plain loads,
vectors per lane, grid cap), timed with tools/average_bench.py's method.  One JSON line per variant."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import portal_amd as pa  # noqa: E402

SRC = os.path.join(ROOT, "portal_amd", "csrc", "kernels", "average_images.hip")
OUT = os.path.join(ROOT, "portal_amd", "kernels")

VARIANTS = [
    ("default", {}, None),
    ("streaming_loads", {"PTL_AVG_NT": 1}, None),
    ("unroll8", {"PTL_AVG_UNROLL": 8}, None),
    ("vpt2", {"PTL_AVG_VPT": 2}, None),
    ("vpt4", {"PTL_AVG_VPT": 4}, None),
    ("vpt2_unroll8", {"PTL_AVG_VPT": 2, "PTL_AVG_UNROLL": 8}, None),
    ("vpt2_cap2048", {"PTL_AVG_VPT": 2}, 2048),
    ("cap2048", {}, 2048),
    ("cap8192", {}, 8192),
]

if __name__ == "__main__":
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    w, h = 3840, 2160
    for name, defs, cap in VARIANTS:
        file = f"average_images_{name}.hsaco"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", "--no-gpu-bundle-output", *[f"-D{k}={v}" for k, v in defs.items()],
                        SRC, "-o", os.path.join(OUT, file)], check=True)
        os.environ["PTL_AVERAGE_IMAGES_HSACO"] = file
        if cap:
            os.environ["PTL_AVERAGE_IMAGES_GRID_CAP"] = str(cap)
        else:
            os.environ.pop("PTL_AVERAGE_IMAGES_GRID_CAP", None)
        row = {"variant": name}
        for n in (2, 4, 8):
            g = torch.Generator(device="cuda").manual_seed(n)
            frames = [torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev, generator=g) for _ in range(n)]
            out = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
            ptrs = [f.data_ptr() for f in frames]
            times = [pa.average_images_device(ptrs, out.data_ptr(), w, h, stream=stream, timed=True) for _ in range(30)]
            ms = float(np.median(times[5:]))
            row[f"n{n}_us"] = round(ms * 1e3, 1)
            row[f"n{n}_GBps"] = round((4 * n + 4) * w * h / ms / 1e6)
        print(json.dumps(row), flush=True)
        os.remove(os.path.join(OUT, file))
