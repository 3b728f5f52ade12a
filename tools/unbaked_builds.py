#!/usr/bin/env python3
"""tools/unbaked_builds.py -- kernel time of the headline frame on the builds that compile less of the scene in (GPU box).
usage: python tools/unbaked_builds.py [scene [W H depth]]"""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa

name, w, h, depth = (sys.argv[1:] + ["portal_in_portal", "3840", "2160", "40"])[:4]
w, h, depth = int(w), int(h), int(depth)
builds = (("everything baked", pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL | pa.flag_waves(4)), ("Bool / Int baked", pa.FLAG_SPECIALIZE_INTS | pa.flag_waves(4)),
          ("Bool / Int baked, no hint", pa.FLAG_SPECIALIZE_INTS), ("patterns", pa.FLAG_SPECIALIZE_PATTERNS | pa.flag_waves(4)), ("patterns, no hint", pa.FLAG_SPECIALIZE_PATTERNS),
          ("un-specialised", 0), ("Bool / Int baked, full products (A/B)", pa.FLAG_SPECIALIZE_INTS | pa.FLAG_NO_ZERO_MASKS | pa.flag_waves(4)))
for label, flags in builds:
    r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(name)), device=0, flags=flags)
    r.set_option("render_depth", depth)
    outs = [r.draw(w, h, rgba8=True) for _ in range(8)]
    print(json.dumps({"scene": name, "build": label, "flags": flags, "ms": round(float(np.median([o["ms"] for o in outs[2:]])), 4), "registers": r.resources()["registers"],
                      "sha": hashlib.sha1(outs[-1]["rgba8"].tobytes()).hexdigest()[:10]}), flush=True)
