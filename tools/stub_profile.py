#!/usr/bin/env python3
"""tools/stub_profile.py -- where does the tracer's time go?  (development aid: the stubbed kernels draw WRONG pictures)

Takes the generated, fully baked source of a scene, knocks one stage out by text substitution, and times each variant
through layer 1 of the C ABI.  The differences against the intact kernel apportion the per-frame time to the stages."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

PATCHES = {
    "intact": [],
    # every hit is final with a flat colour: ray generation + ONE scene_intersect per sample, no shading, no second trip
    "no_material": [("m = material_process(r, i);", "m = MaterialProcessing{true, vec3(0.5f), ray_none};"),
                    ("m = material_process(r, i2.scene);", "m = MaterialProcessing{true, vec3(0.5f), ray_none};")],
    # planes never hit: what do the Flat objects cost (transform, early-out, sqrt, divisions, is_inside)?
    "no_planes": [("    r = transform(plane_inv, r);\n    float len = length(r.d);", "    return intersection_none;\n    r = transform(plane_inv, r);\n    float len = length(r.d);")],
    # the snippet of the scene (intersection material) never hits
    "no_snippet": [("hit = intersect_material_0(r);", "hit = SceneIntersectionWithMaterial{scene_intersection_none, material_empty()};")],
    # Complex objects (scene snippets intersect_<k>) skipped
    "no_complex": [("ihit = intersect_", "if (len < 0.0f) ihit = intersect_")],
    # no bounce loop at all: ray generation, AA loop, gamma, store
    "no_trace": [("    bool alive = true;\n    for (int j = 0; j < _ray_tracing_depth; j++) {",
                  "    bool alive = true;\n    if (r.d.x > 2.0f) return RayTraceResult{vec3(r.d.x, r.d.y, r.d.z), 0.0f, false};\n    for (int j = 0; j < 0; j++) {")],
}

if __name__ == "__main__":
    name, w, h, depth = (sys.argv[1:] + ["portal_in_portal", "3840", "2160", "40"])[:4]
    w, h, depth = int(w), int(h), int(depth)
    scene = pa.Scene.from_file(pa.scene_path(name))
    r = pa.SceneRenderer(scene, device=0, flags=pa.FLAG_SPECIALIZE_ALL)
    r.set_option("render_depth", depth)
    source = scene.generate_source(pa.FLAG_SPECIALIZE_ALL)
    layout, size = scene.uniform_layout()
    import ctypes as C

    for variant, subs in PATCHES.items():
        src = source
        for old, new in subs:
            assert old in src, (variant, old)
            src = src.replace(old, new)
        k = pa.Kernel(src, layout, size, device=0)
        for uname, typ, _ in layout:
            if typ == pa.PTL_SAMPLER:
                continue
            v = r.uniform_value(uname, w, h)
            if v is not None:
                k.set_uniform(uname, typ, v)
        times = [k.render(w, h)["ms"] for _ in range(8)]
        print(json.dumps({"scene": name, "variant": variant, "ms": round(float(np.median(times[2:])), 4)}), flush=True)
