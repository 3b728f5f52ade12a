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
NAMES = {0: "rgba8 via 32x8 LDS transpose (renderer)", 1: "rgba32f float4 per lane (parity buffer)", 2: "rgba8 direct 8x8 tile (32 B segments)", 3: "linear 16 B/lane stream"}

if __name__ == "__main__":
    import torch

    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
    dev = torch.device("cuda", 0)
    buf = torch.empty(W * H * 16, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    for variant in (0, 1, 2, 3):
        k = pa.Kernel(SRC, [("seed_u", pa.PTL_I32, 0), ("pad_u", pa.PTL_I32, 4)], 8, device=0, defines=[f"PTL_FB_VARIANT={variant}"])
        k.set_uniform("seed_u", pa.PTL_I32, 12345)
        frame = pa.Frame(W, H, 0, 1)
        times = []
        for _ in range(12):
            ms = C.c_float()
            rc = pa.lib().ptl_kernel_render(k._h, C.byref(frame), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), None, C.c_void_p(stream.cuda_stream), C.byref(ms))
            assert rc == 0, pa.lib().ptl_last_error()
            times.append(ms.value)
        nbytes = W * H * (16 if variant == 1 else 4)
        ms = float(np.median(times[2:]))
        print(json.dumps({"variant": NAMES[variant], "frame": f"{W}x{H}", "bytes": nbytes, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1),
                          "frac_of_8TB/s": round(nbytes / ms / 1e6 / 8000, 4)}), flush=True)
