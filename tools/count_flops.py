#!/usr/bin/env python3
"""count_flops.py -- algorithmic binary32 operations per bounce-loop trip, for bench.py's `roofline`.

MEASUREMENT TOOL (imports oracle/: test infrastructure; nothing under portal_amd/ uses this).  Runs the numpy oracle
on a seeded pixel sample of the FULL-SIZE frame of a BASELINE.json config and writes, per config, the operation
counts per trip (one trip = one pass of the bounce loop for one sample):

  flops           every binary32 operation the arithmetic contract performs (+ - * / sqrt floor cmp min max = 1, fma = 2)
  flops_varying   only those with at least one operand that differs between rays.  This is the arithmetic the TIMED kernel
                  executes: bench.py times the JIT-specialised build (all scene uniforms baked), in which every
                  ray-independent subexpression -- normalize(get_normal(M)), the uniform half of is_collinear, the snippet's
                  normal_b chain ... -- is folded by the compiler.  `roofline.achieved` uses this figure.

Counting is per ACTIVE lane (the oracle masks lanes exactly like the control flow does), so it is work the picture needs,
not issue slots: divergence and the multi-instruction expansions of / and sqrt (11 and 14 VALU instructions) are not in it.

    python tools/count_flops.py                      # the four GPU configs -> profiles/r02/flops_per_segment.json
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)

CONFIGS = {  # name: (scene, width, height, depth, aa, sample fraction of the frame's pixels)
    "monoportal_1920x1080_d20": ("monoportal", 1920, 1080, 20, 1, 0.02),
    "triple_portal_3840x2160_d40": ("triple_portal", 3840, 2160, 40, 1, 0.01),
    "portal_in_portal_3840x2160_d40": ("portal_in_portal", 3840, 2160, 40, 1, 0.0125),
    "mobius_monoportal_7680x4320_d64_aa4": ("mobius_monoportal", 7680, 4320, 64, 4, 0.0002),
}


def count(scene, w, h, depth, aa, frac, seed=20260925, chunk=16384, options=None):
    import portal_amd as pa
    from oracle import glsl_math as M
    from oracle.portal_oracle import Oracle

    M.COUNT_VARYING = True
    o = Oracle(pa.scene_path(scene))
    o.options.update(render_depth=depth, aa_count=aa)
    if options:
        o.options.update(options)
    rng = np.random.default_rng(seed)
    n = int(round(w * h * frac))
    flat = rng.choice(w * h, size=n, replace=False)
    total = {}
    t0 = time.time()
    for i in range(0, n, chunk):
        part = flat[i:i + chunk]
        o.shade_pixels(w, h, part % w, part // w)
        for k, v in o.stats.items():
            total[k] = total.get(k, 0.0) + float(v)
    seg = total.pop("segments")
    return {
        "scene": scene, "width": w, "height": h, "depth": depth, "aa": aa,
        "sampled_pixels": n, "sampled_fraction_of_frame": n / (w * h), "seed": seed,
        "segments_in_sample": int(seg), "segments_per_primary_sample": seg / (n * aa),
        "per_segment": {k: v / seg for k, v in total.items()},
        "oracle_seconds": round(time.time() - t0, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "profiles", "r02", "flops_per_segment.json"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    out = {}
    if os.path.exists(args.out):
        out = json.load(open(args.out))
    for name, cfg in CONFIGS.items():
        if args.only and name != args.only:
            continue
        out[name] = count(*cfg)
        print(name, json.dumps(out[name]["per_segment"]), out[name]["oracle_seconds"], "s", flush=True)
    # SURVEY 8d's Panini variant of the headline (d = 1, fov 140)
    if not args.only or args.only == "portal_in_portal_3840x2160_d40_panini":
        out["portal_in_portal_3840x2160_d40_panini"] = count("portal_in_portal", 3840, 2160, 40, 1, 0.0125,
                                                             options=dict(use_panini=True, panini_param=1.0, view_angle=float(np.radians(140.0))))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
