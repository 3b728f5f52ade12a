#!/usr/bin/env python3
"""tools/ab_views.py -- A/B timing of renderer flag sets over several views of a scene (development aid).

    python tools/ab_views.py [--precompile] [--scene portal_in_portal] [--size 3840x2160] [--depth 40] NAME=FLAGS ...

Every NAME=FLAGS (FLAGS: an integer, or `|`-joined portal_amd.FLAG_* names without the prefix, e.g. spec=SPECIALIZE_INTS|SPECIALIZE_ALL) is
drawn at each view: median kernel time of 12 draws and a hash of the frame, which must not depend on the flags.  --precompile only
fills the code-object cache (no GPU)."""
import hashlib, json, os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

VIEWS = {"default": None, "deep": ((0.0, 0.0, 0.0), 0.2, 1.5, 1.6), "deep2": ((0.3, -0.1, 0.2), 2.8, 1.0, 2.4), "side": ((0.1, 0.3, -0.2), 1.2, 1.4, 3.0), "close": ((0.0, 0.1, 0.0), 4.0, 1.7, 0.8)}


def flags_of(text):
    v = 0
    for part in text.split("|"):
        part = part.strip()
        if part.startswith("w") and part[1:].isdigit():
            v |= pa.flag_waves(int(part[1:]))
        else:
            v |= int(part) if part.lstrip("-").isdigit() else getattr(pa, "FLAG_" + part)
    return v


if __name__ == "__main__":
    args = sys.argv[1:]
    pre = "--precompile" in args
    opt = {"--scene": "portal_in_portal", "--size": "3840x2160", "--depth": "40", "--scene-file": ""}
    rest = []
    k = 0
    while k < len(args):
        if args[k] in opt:
            opt[args[k]] = args[k + 1]
            k += 2
        else:
            if args[k] != "--precompile":
                rest.append(args[k])
            k += 1
    w, h = (int(x) for x in opt["--size"].split("x"))
    path = os.path.join(pa.REPO_ROOT, opt["--scene-file"]) if opt["--scene-file"] else pa.scene_path(opt["--scene"])
    extra = {"asset_root": os.path.dirname(os.path.dirname(path))} if opt["--scene-file"] else {}
    for item in rest:
        name, text = item.split("=", 1)
        flags = flags_of(text)
        r = pa.SceneRenderer(pa.Scene.from_file(path), device=-1 if pre else 0, flags=flags, **extra)
        if pre:
            print(json.dumps({"variant": name, "flags": flags, "code_object_bytes": len(r.code_object())}), flush=True)
            continue
        r.set_option("render_depth", int(opt["--depth"]))
        for view, cam in VIEWS.items():
            if cam:
                r.set_camera(*cam)
            else:
                r.use_camera("")
            outs = [r.draw(w, h, rgba8=True) for _ in range(14)]
            print(json.dumps({"scene": os.path.basename(path), "variant": name, "flags": flags, "view": view, "ms": round(float(np.median([o["ms"] for o in outs[2:]])), 4),
                              "sha": hashlib.sha1(outs[-1]["rgba8"].tobytes()).hexdigest()[:10]}), flush=True)
