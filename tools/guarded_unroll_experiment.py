#!/usr/bin/env python3
"""tools/guarded_unroll_experiment.py -- round 5: what would unrolling a snippet loop whose bound is a RUN-TIME Int uniform buy?

    for (int k = 0; k < bound_u; k++) { BODY }
 ->
    _Pragma("unroll") for (int k = 0; k < V; k++) { if (k >= bound_u) break; BODY }      (V = the bound's value when the kernel is generated, <= 16)
    for (int k = V; k < bound_u; k++) { BODY }                                               (whatever a later state adds: rolled)

Valid for every state (the remainder loop), tuned for the state at generation.  The un-specialised and the patterns build of the headline scene,
patched in Python, timed through layer 1 against the unpatched source: kernel ms and a frame hash that must not move.
PTL_VARIANTS_PRECOMPILE=1: no GPU, only fill the code-object cache."""
import hashlib
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa  # noqa: E402


def matching_brace(text, open_at):
    depth, i = 0, open_at
    while i < len(text):
        if text.startswith("//", i):
            i = text.index("\n", i)
            continue
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced")


def guarded_unroll(src, name, value):
    out, at = "", 0
    for m in re.finditer(r"for \(int (\w+) = 0; \1 < " + re.escape(name) + r"; \1\+\+\) \{", src):
        if m.start() < at:
            continue
        close = matching_brace(src, m.end() - 1)
        body = src[m.end():close]
        if re.search(r"\bbreak\b", body):
            continue
        k = m.group(1)
        out += src[at:m.start()]
        out += f'_Pragma("unroll") for (int {k} = 0; {k} < {value}; {k}++) {{ if ({k} >= {name}) break;{body}}}\n'
        out += f"for (int {k} = {value}; {k} < {name}; {k}++) {{{body}}}"
        at = close + 1
    return out + src[at:]


if __name__ == "__main__":
    device = -1 if os.environ.get("PTL_VARIANTS_PRECOMPILE") else 0
    w, h, depth = 3840, 2160, 40
    for label, flags in (("patterns", pa.FLAG_SPECIALIZE_PATTERNS), ("unspecialised", 0)):
        scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
        r = pa.SceneRenderer(scene, device=device, flags=flags)
        r.set_option("render_depth", depth)
        source = r.kernel_source()
        defines = list(scene.generated_defines())
        layout, size = scene.uniform_layout()
        bound = int(r.uniform_value("show_teleported_u", w, h))
        for variant, src in (("as generated", source), ("guarded unroll", guarded_unroll(source, "show_teleported_u", bound))):
            for waves in (0, 4):
                k = pa.Kernel(src, layout, size, device=device, defines=defines + ([f"PTL_WAVES_PER_EU={waves}"] if waves else []))
                if device < 0:
                    continue
                for uname, typ, _ in layout:
                    if typ == pa.PTL_SAMPLER:
                        continue
                    v = r.uniform_value(uname, w, h)
                    if v is not None:
                        k.set_uniform(uname, typ, v)
                outs = [k.render(w, h) for _ in range(8)]
                print(json.dumps({"build": label, "variant": variant, "waves_hint": waves, "ms": round(float(np.median([o["ms"] for o in outs[2:]])), 4),
                                  "sha": hashlib.sha1(outs[-1]["rgba8"].tobytes()).hexdigest()[:10]}), flush=True)
