#!/usr/bin/env python3
"""tools/unspec_anatomy.py -- what separates the un-specialised kernel (valid for every scene state) from the patterns build on the headline frame (GPU box).
Each line: one build, the ingredients it has of {Bool / Int switches compiled in, zero / unit patterns of the run-time matrices, affine rays, no transform dodges},
its kernel time, registers, and the frame hash (every build draws the same frame).
usage: python tools/unspec_anatomy.py [scene [W H depth]]"""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa

name, w, h, depth = (sys.argv[1:] + ["portal_in_portal", "3840", "2160", "40"])[:4]
w, h, depth = int(w), int(h), int(depth)
P, I = pa.FLAG_SPECIALIZE_PATTERNS, pa.FLAG_SPECIALIZE_INTS
NOAFF, NOMASK, KEEP = pa.FLAG_NO_AFFINE_RAYS, pa.FLAG_NO_ZERO_MASKS, pa.FLAG_KEEP_TRANSFORM_DODGES
builds = [("un-specialised", 0), ("un-specialised -O1", 0), ("patterns (switches + masks + affine rays, no dodges)", P), ("patterns, transform dodges kept", P | KEEP),
          ("patterns, general products (switches + masks)", P | NOAFF), ("patterns, no masks (switches only; affine rays need masks)", P | NOMASK),
          ("Bool / Int baked, no masks, general products (switches only, every Int)", I | NOMASK | NOAFF), ("Bool / Int baked", I)]
for waves in (0, 4):
    for label, flags in builds:
        if "-O1" in label:
            os.environ["PTL_JIT_OPT"] = "-O1"
        try:
            r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(name)), device=0, flags=flags | pa.flag_waves(waves))
        finally:
            os.environ.pop("PTL_JIT_OPT", None)
        r.set_option("render_depth", depth)
        outs = [r.draw(w, h, rgba8=True) for _ in range(10)]
        print(json.dumps({"scene": name, "build": label, "waves_hint": waves, "flags": flags, "ms": round(float(np.median([o["ms"] for o in outs[3:]])), 4), "registers": r.resources()["registers"],
                          "scratch": r.resources()["scratch_bytes"], "affine_rays": r.affine_rays(), "sha": hashlib.sha1(outs[-1]["rgba8"].tobytes()).hexdigest()[:10]}), flush=True)
