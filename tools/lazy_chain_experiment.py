#!/usr/bin/env python3
"""tools/lazy_chain_experiment.py -- what does deferring the snippet's loop-carried ray transforms buy?  (the experiment that led to
glsl_translate's `defer_loop_updates`; kept as a cross-check: hand-patched source vs what the translator now emits)

portal_in_portal's intersection-material snippet advances two rays through a matrix pair on EVERY iteration of its loop
(r_teleport_a / r_teleport_b) but reads them only where a portal hit is recorded.  Deferring the updates (count them, apply the
pending ones right before a read) executes the same operations on the same values -- identical frames -- and skips them for rays that
never record a hit.  This patches the generated source by hand and times it through layer 1; frame hashes must agree."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402

if __name__ == "__main__":
    w, h, depth = 3840, 2160, 40
    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    for label, flags in (("baked", pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL), ("dynamic", 0)):
        r = pa.SceneRenderer(scene, device=0, flags=flags)
        r.set_option("render_depth", depth)
        source = scene.generate_source(flags | pa.FLAG_NO_DEFERRED_UPDATES)  # the snippets as written
        translated = scene.generate_source(flags)                            # what the translator does by itself
        layout, size = scene.uniform_layout()
        lazy = source
        subs = [
            ("Ray r_teleport_a = r;\nRay r_teleport_b = r;\n", "Ray r_teleport_a = r;\nRay r_teleport_b = r;\nint ptl_pend_a = 0, ptl_pend_b = 0;\n"),
            ("\tr_teleport_a = transform(b0_mat, transform(a_mat_inv, r_teleport_a));\n\tr_teleport_b = transform(a_mat, transform(b0_mat_inv, r_teleport_b));\n",
             "\t++ptl_pend_a; ++ptl_pend_b;\n"),
            ("\t\t\t\tresult.material = material_teleport_transformed(offset_ray(r_teleport_a, hit_a.t), vec3(1.f));",
             "\t\t\t\tfor (; ptl_pend_a > 0; --ptl_pend_a) r_teleport_a = transform(b0_mat, transform(a_mat_inv, r_teleport_a));\n"
             "\t\t\t\tresult.material = material_teleport_transformed(offset_ray(r_teleport_a, hit_a.t), vec3(1.f));"),
            ("\t\t\t\tresult.material = material_teleport_transformed(offset_ray(r_teleport_b, hit_b.t), vec3(1.f));",
             "\t\t\t\tfor (; ptl_pend_b > 0; --ptl_pend_b) r_teleport_b = transform(a_mat, transform(b0_mat_inv, r_teleport_b));\n"
             "\t\t\t\tresult.material = material_teleport_transformed(offset_ray(r_teleport_b, hit_b.t), vec3(1.f));"),
        ]
        for old, new in subs:
            assert lazy.count(old) == 1, old
            lazy = lazy.replace(old, new)
        for variant, src in (("eager (snippet as written)", source), ("deferred ray chains, patched by hand", lazy), ("deferred ray chains, translator", translated)):
            k = pa.Kernel(src, layout, size, device=0)
            for uname, typ, _ in layout:
                if typ == pa.PTL_SAMPLER:
                    continue
                v = r.uniform_value(uname, w, h)
                if v is not None:
                    k.set_uniform(uname, typ, v)
            outs = [k.render(w, h) for _ in range(8)]
            print(json.dumps({"build": label, "variant": variant, "ms": round(float(np.median([o["ms"] for o in outs[2:]])), 4),
                              "sha": hashlib.sha1(outs[-1]["rgba8"].tobytes()).hexdigest()[:10]}), flush=True)
