#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerate the committed golden frames.

The reference publishes no golden image and cannot be run here, so these
vectors come from THIS repo's numpy oracle (oracle/portal_oracle.py); they pin the oracle and
the product against silent drift, they do not pin either against the reference
("parity unpinned").  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.portal_oracle import Oracle  # noqa: E402

CASES = [  # scene, width, height, depth, aa  (BASELINE.json configs, scaled down 10-40x per axis)
    ("basics", 64, 64, 4, 1),
    ("monoportal", 96, 54, 20, 1),
    ("triple_portal", 96, 54, 40, 1),
    ("portal_in_portal", 96, 54, 40, 1),
    ("mobius_monoportal", 64, 36, 64, 2),
]

if __name__ == "__main__":
    for scene, w, h, depth, aa in CASES:
        o = Oracle(os.path.join(ROOT, "scenes", scene + ".ron"))
        o.options.update(render_depth=depth, aa_count=aa)
        out = o.render(w, h)
        path = os.path.join(ROOT, "tests", "golden", f"{scene}_{w}x{h}_d{depth}_aa{aa}.npz")
        np.savez_compressed(path, rgba32f_bits=out["rgba32f"].view(np.uint32), rgba8=out["rgba8"], segments=out["segments"].astype(np.int32),
                            flops_per_segment=np.float64(o.stats["flops"] / max(1, o.stats["segments"])))
        print(path, out["rgba8"].mean(axis=(0, 1)), "segments", int(out["segments"].sum()))
