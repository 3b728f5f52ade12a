#!/usr/bin/env python3
"""tests/golden/make_reference_text_fixtures.py -- OUTPUTS of the reference's own shader text.

Runs `oracle.reference_shader.ReferenceShader` -- `/root/reference/src/library.glsl` + `src/frag.glsl`
with the slots filled as `src/gui/scene.rs:693-1075` fills them, executed by oracle/glsl_interp.py --
and stores what it computes:

  reference_text/<case>.npz      frames (float bits + RGBA8) of the five BASELINE scenes and six mode /
                                 camera variants (tests/reftext.py::FRAME_CASES)
  reference_text/functions.npz   every function of library.glsl / frag.glsl on 1024 seeded lanes (random,
                                 scene-range, exact halves, +-0 / inf / NaN / denormals): output bits only,
                                 the inputs are regenerated from the seed (tests/reftext.py::make_args)
  reference_text/teleport.npz    `teleport_external_ray` queries (function result and the 2x3 RGBA8
                                 framebuffer route of src/main.rs:1361-1409)

These are the reference-derived vectors that pin the hand-written oracle (and, through the -m gpu tests,
the HIP kernel): they need /root/reference to be REGENERATED, not to be checked.
Run from the repo root:  python tests/golden/make_reference_text_fixtures.py
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import glsl_values as V  # noqa: E402
from oracle import reference_shader as RS  # noqa: E402
from tests import reftext as T  # noqa: E402

N_FUNCTION_LANES = 1024
TELEPORT_SCENES = ("monoportal", "triple_portal", "portal_in_portal")

if __name__ == "__main__":
    os.makedirs(T.GOLDEN_DIR, exist_ok=True)
    digest = RS.text_digest()
    for case in T.FRAME_CASES:
        out = T.render_case(RS.ReferenceShader, case)
        np.savez_compressed(os.path.join(T.GOLDEN_DIR, case + ".npz"), rgba32f_bits=out["rgba32f"].view(np.uint32), rgba8=out["rgba8"],
                            text_digest=np.array(digest))
        print(case, out["rgba8"].mean(axis=(0, 1)))
    rs = RS.ReferenceShader(os.path.join(ROOT, "scenes", "basics.ron"))
    rs.build(64, 64)
    store = {"text_digest": np.array(digest)}
    for name, ptypes, ret in T.function_table(rs):
        args = T.make_args(name, ptypes, N_FUNCTION_LANES, rs._program.structs)
        got = V.expand(rs.call(name, args, N_FUNCTION_LANES), N_FUNCTION_LANES)
        store[name + "(" + ",".join(ptypes) + ")"] = T.leaves_array(got, N_FUNCTION_LANES)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "functions.npz"), **store)
    print("functions:", len(store) - 1)
    tele = {"text_digest": np.array(digest)}
    for scene in TELEPORT_SCENES:
        rs = RS.ReferenceShader(os.path.join(ROOT, "scenes", scene + ".ron"))
        rows = []
        for a, b in T.teleport_segments(scene):
            p1, h1, s1 = rs.teleport_external_ray(a, b)
            p2, h2, s2 = rs.teleport_external_ray_through_framebuffer(a, b)
            pack = lambda p, h, s: [0 if p is None else 1, int(h), int(s)] + list((np.zeros(3, np.float32) if p is None else p).view(np.uint32))
            rows.append(pack(p1, h1, s1) + pack(p2, h2, s2))
        tele[scene] = np.array(rows, np.uint32)
        print(scene, "teleported:", int(tele[scene][:, 0].sum()), "of", len(rows))
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "teleport.npz"), **tele)
