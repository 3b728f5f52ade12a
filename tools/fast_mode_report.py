#!/usr/bin/env python3
"""tools/fast_mode_report.py -- the tolerance mode (FLAG_FAST_MATH, `--fast`) against the exact kernel on the BASELINE configs.

Runs ON the GPU box.  Per config and build (dynamic uniforms / all scene uniforms baked): kernel milliseconds of both modes and
how far the fast picture is from the exact one -- pixels beyond the north star's 1e-5 per channel, beyond 1/255 (a visible
change: another path was taken), the median and 99.9th percentile error.  One JSON line per (config, build).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

CONFIGS = [("monoportal", 1920, 1080, 20, 1), ("triple_portal", 3840, 2160, 40, 1), ("portal_in_portal", 3840, 2160, 40, 1), ("mobius_monoportal", 7680, 4320, 64, 4)]

if __name__ == "__main__":
    for scene_name, w, h, depth, aa in CONFIGS:
        for build, spec in (("dynamic", 0), ("baked", pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)):
            frames, ms = {}, {}
            for mode, flags in (("exact", spec), ("fast", spec | pa.FLAG_FAST_MATH)):
                r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(scene_name)), device=0, flags=flags)
                r.set_option("render_depth", depth)
                r.set_option("aa_count", aa)
                out = r.draw(w, h, rgba8=True, rgba32f=True)
                frames[mode] = (out["rgba32f"][..., :3].copy(), out["rgba8"].copy())
                ms[mode] = float(np.median([r.draw(w, h, rgba8=True)["ms"] for _ in range(6)][1:]))
                del r
            err = np.abs(frames["exact"][0] - frames["fast"][0]).max(axis=2)
            print(json.dumps({
                "config": f"{scene_name} {w}x{h} aa{aa} d{depth}", "build": build,
                "exact_ms": round(ms["exact"], 4), "fast_ms": round(ms["fast"], 4), "speedup": round(ms["exact"] / ms["fast"], 3),
                "pixels": w * h, "beyond_1e-5": int((err > 1e-5).sum()), "beyond_1e-5_fraction": float((err > 1e-5).mean()),
                "beyond_1_255": int((err > 1.0 / 255).sum()), "rgba8_bytes_differ": int((frames["exact"][1] != frames["fast"][1]).any(axis=2).sum()),
                "median_abs_error": float(np.median(err)), "p999_abs_error": float(np.quantile(err, 0.999)), "max_abs_error": float(err.max()),
            }), flush=True)
