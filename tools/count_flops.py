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

The oracle evaluates the reference's GLSL as written.  The product's translator defers loop-carried ray transforms of scene snippets
(glsl_translate.h `defer_loop_updates`): an update that no statement ever reads is never executed.  How many that is cannot come from the
oracle; it is measured on the HOST BUILD of the generated source with two counters compiled in (updates scheduled / updates applied) on
rows sampled over the same full-size frame, and `flops_*_executed` = the oracle's count minus (scheduled - applied) per trip x the
operations of one update (56 per `transform`: two mat4 x vec4 of 4 x (1 mul + 3 fma)).
Round 3: the generated plane tests the wave-level cull skips (ptl_library.h::ptl_plane_cull) are subtracted the same way -- tests and
culled tests per trip counted on the host build (per ray; on the GPU the decision is per wave and the 64 rays of a tile agree in 99.9 %
of the cases, so this slightly OVER-subtracts), times what the oracle spends per generated plane test (`plane_intersect` + `nearer`,
measured inside Oracle.scene_intersect).  The z row the cull itself evaluates is not claimed as work.

    python tools/count_flops.py                      # the four GPU configs -> profiles/r03/flops_per_segment.json
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


def _scene_path(pa, scene):
    """`scene`: a name under scenes/, or a repo-relative .ron path (the reference's corpus under tests/corpus/scenes)"""
    return os.path.join(HERE, scene) if scene.endswith(".ron") else pa.scene_path(scene)


def _scene_kw(scene):
    return {"asset_root": os.path.dirname(os.path.dirname(os.path.join(HERE, scene)))} if scene.endswith(".ron") else {}


def count(scene, w, h, depth, aa, frac, seed=20260925, chunk=16384, options=None, camera=None):
    import portal_amd as pa
    from oracle import glsl_math as M
    from oracle.portal_oracle import Oracle

    M.COUNT_VARYING = True
    _vary_the_camera_between_lanes()
    _meter_the_generated_plane_tests()
    PLANE_METER.update(flops=0.0, flops_varying=0.0, lane_tests=0.0)
    o = Oracle(_scene_path(pa, scene), **_scene_kw(scene))
    o.options.update(render_depth=depth, aa_count=aa)
    if options:
        o.options.update(options)
    if camera:
        o.camera = dict(look_at=tuple(camera[:3]), alpha=camera[3], beta=camera[4], r=camera[5])
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
    per_segment = {k: v / seg for k, v in total.items()}
    deferred = deferred_update_counts(scene, w, h, depth, aa, options, camera=camera)
    if deferred:
        skipped = deferred["flops_per_update"] * (deferred["scheduled_per_segment"] - deferred["applied_per_segment"])
        per_segment["flops_executed"] = per_segment["flops"] - skipped
        per_segment["flops_varying_executed"] = per_segment["flops_varying"] - skipped
    culls = plane_cull_counts(scene, w, h, depth, aa, options, camera=camera)
    if culls and PLANE_METER["lane_tests"]:
        per_test = {k: PLANE_METER[k] / PLANE_METER["lane_tests"] for k in ("flops", "flops_varying")}
        culls["oracle_flops_per_plane_test"] = per_test
        culls["oracle_plane_tests_per_segment"] = PLANE_METER["lane_tests"] / seg
        for k in ("flops", "flops_varying"):
            base = per_segment.get(k + "_executed", per_segment[k])
            per_segment[k + "_executed"] = base - culls["culled_per_segment"] * per_test[k]
    # a build with the matrices baked in skips matrix-product terms whose matrix element is zero (ptl_glsl.h `ptl_mterm`)
    zero_terms = per_segment.pop("zero_term_flops_varying", 0.0)
    first = first_trip_origin_flops(scene, w, h, options)
    if first:
        # the share of trips that ARE first trips: one per primary sample
        saved = first["flops_per_first_trip"] * (n * aa) / seg
        first["flops_per_segment"] = saved
        for k in ("flops_executed", "flops_varying_executed"):
            base = per_segment.get(k, per_segment[k.replace("_executed", "")])
            per_segment[k] = base - saved
    if zero_terms:
        # (the deferred updates and first-trip origins subtracted above are whole transforms: their zero terms must not be taken off twice.
        # A deferred update is 2 transforms = 4 products; the share of zero terms in them is the scene's average share)
        share = zero_terms / max(per_segment["flops_varying"], 1.0)
        base = per_segment.get("flops_varying_executed", per_segment["flops_varying"])
        per_segment["zero_term_flops_varying"] = zero_terms
        per_segment["flops_varying_executed_baked"] = base - zero_terms * min(1.0, base / per_segment["flops_varying"]) if share < 1 else base
        # round 5: +-1 matrix elements (the multiplication of fma(+-1, x, acc) is not executed: one add) and, in a kernel with affine rays,
        # the terms that meet a ray's w (a direction's 0: skipped; an origin's 1: `acc + element`) -- scaled like the zero terms
        scale = min(1.0, base / per_segment["flops_varying"]) if share < 1 else 0.0
        unit = per_segment.get("unit_term_flops_varying", 0.0)
        known_w = per_segment.get("known_w_term_flops_varying", 0.0)
        per_segment["flops_varying_executed_baked_units"] = per_segment["flops_varying_executed_baked"] - unit * scale
        # A kernel with affine rays is generated WITHOUT the deferred loop updates and without first-trip snippet copies (codegen.cpp: its
        # transforms are a few additions): it executes every transform the snippet writes, shortened term by term.  So: the oracle's count,
        # minus the culled plane tests, minus zero / unit / known-w terms -- nothing taken off for deferral or first-trip origins.
        affine_base = per_segment["flops_varying"] - (culls["culled_per_segment"] * culls["oracle_flops_per_plane_test"]["flops_varying"] if culls and PLANE_METER["lane_tests"] else 0.0)
        affine_scale = affine_base / per_segment["flops_varying"]
        per_segment["flops_varying_executed_baked_affine"] = affine_base - (zero_terms + unit + known_w) * affine_scale
    return {
        "first_trip_origin_arithmetic": first,
        "scene": scene, "width": w, "height": h, "depth": depth, "aa": aa,
        "sampled_pixels": n, "sampled_fraction_of_frame": n / (w * h), "seed": seed,
        "segments_in_sample": int(seg), "segments_per_primary_sample": seg / (n * aa),
        "per_segment": per_segment,
        "deferred_updates": deferred,
        "plane_cull": culls,
        "oracle_seconds": round(time.time() - t0, 1),
    }


_camera_patched = False


def _vary_the_camera_between_lanes():
    """`flops_varying` is decided by VALUE (an operand whose lanes all hold the same bits counts as ray-independent).  All primary rays
    start at the same point, so everything computed from the ray ORIGIN alone on the first trip -- `plane_inv * r.o` of every plane
    test -- would pass as ray-independent although it depends on the camera, which no build bakes in.  For counting (not for parity)
    every other lane therefore gets a camera moved by a millimetre: camera-dependent operands then differ between lanes, scene-uniform
    ones still do not."""
    global _camera_patched
    if _camera_patched:
        return
    _camera_patched = True
    from oracle import glsl_values as V
    from oracle.portal_oracle import Oracle

    original = Oracle.get_color2

    def get_color2(self, image_position, camera, *args, **kwargs):
        n = len(np.asarray(image_position.c[0]))
        odd = (np.arange(n) % 2).astype(np.float32)
        cols = [V.Vec([np.full(n, np.float32(x), np.float32) for x in col.c]) for col in camera.cols]
        shift = (1e-3, -1e-3, 1e-3, 0.0)
        cols[3] = V.Vec([np.asarray(c + np.float32(s) * odd, np.float32) for c, s in zip(cols[3].c, shift)])
        return original(self, image_position, V.Mat(cols), *args, **kwargs)

    Oracle.get_color2 = get_color2


PLANE_METER = {"flops": 0.0, "flops_varying": 0.0, "lane_tests": 0.0}
_plane_patched = False


def _meter_the_generated_plane_tests():
    """What the oracle spends on ONE generated plane test (scene.rs:912-948: `plane_intersect` + `nearer`), per active lane: the work
    a culled test skips.  Only the calls made from Oracle.scene_intersect are metered -- a scene snippet's own plane_intersect calls
    are never culled."""
    global _plane_patched
    if _plane_patched:
        return
    _plane_patched = True
    from oracle import glsl_math as M
    from oracle.portal_oracle import Natives, Oracle

    inside = {"on": False}
    original_si, original_pi, original_nearer = Oracle.scene_intersect, Natives.plane_intersect, Natives.nearer

    def scene_intersect(self, *a, **k):
        inside["on"] = True
        try:
            return original_si(self, *a, **k)
        finally:
            inside["on"] = False

    def plane_intersect(self, *a, **k):
        if not inside["on"]:
            return original_pi(self, *a, **k)
        before = (M.STATS["flops"], M.STATS["flops_varying"])
        out = original_pi(self, *a, **k)
        PLANE_METER["flops"] += M.STATS["flops"] - before[0] + 3.0 * M._active            # + nearer(): three compares
        PLANE_METER["flops_varying"] += M.STATS["flops_varying"] - before[1] + 3.0 * M._active
        PLANE_METER["lane_tests"] += M._active
        return out

    Oracle.scene_intersect = scene_intersect
    Natives.plane_intersect = plane_intersect


def plane_cull_counts(scene_name, w, h, depth, aa, options=None, row_step=61, camera=None):
    """Generated plane tests and how many of them ptl_plane_cull skips, per bounce-loop trip, counted per ray on the host build of the
    baked source (rows row_step/2, +row_step, ... of the full-size frame).  None when the source has no culled test."""
    import ctypes as C

    import portal_amd as pa
    from oracle import host_build as hb

    scene = pa.Scene.from_file(_scene_path(pa, scene_name))
    source = scene.generate_source(pa.FLAG_COUNT_SEGMENTS)
    needle = "    return ptl_cannot_be_nearer(oz, dz, best_t);\n"  # the host form of ptl_plane_cull / ptl_plane_cull_o (one sign test since round 4)
    if needle not in source or "ptl_plane_cull" not in source:
        return None
    source = source.replace("namespace glsl {\n", "namespace glsl {\nstatic long ptl_cull_stats[2] = {0, 0};\n", 1)
    source = source.replace(needle, "    { const bool ptl_c = ptl_cannot_be_nearer(oz, dz, best_t); __atomic_fetch_add(&ptl_cull_stats[ptl_c ? 1 : 0], 1, __ATOMIC_RELAXED); return ptl_c; }\n")
    source += '\nextern "C" long* ptl_cull_stats_ptr() { return glsl::ptl_cull_stats; }\n'
    renderer = pa.SceneRenderer(scene, device=-1, **_scene_kw(scene_name))
    renderer.set_option("render_depth", depth)
    renderer.set_option("aa_count", aa)
    for k, v in (options or {}).items():
        renderer.set_option({"use_panini": "use_panini_projection"}.get(k, k), float(v))
    if camera:
        renderer.set_camera(camera[:3], camera[3], camera[4], camera[5])
    layout, size = scene.uniform_layout()
    hk = hb.HostKernel(source, layout, size, True)
    for name, typ, _ in layout:
        if typ == pa.PTL_SAMPLER:
            continue
        v = renderer.uniform_value(name, w, h)
        if v is not None:
            hk.set_uniform(name, v)
    from PIL import Image

    paths = scene.textures()
    for name, typ, _ in layout:
        if typ == pa.PTL_SAMPLER and paths.get(name[: -len("_tex")]) and os.path.exists(os.path.join(pa.REPO_ROOT, paths[name[: -len("_tex")]])):
            hk.set_texture(name, np.array(Image.open(os.path.join(pa.REPO_ROOT, paths[name[: -len("_tex")]])).convert("RGBA")))
    hk.lib.ptl_cull_stats_ptr.restype = C.POINTER(C.c_long)
    st = hk.lib.ptl_cull_stats_ptr()
    st[0] = st[1] = 0
    rows = list(range(row_step // 2, h, row_step))
    seg = hk.render(w, h, rows=rows, rgba32f=False)["segments"]
    return {"tests_per_segment": (st[0] + st[1]) / seg, "culled_per_segment": st[1] / seg, "segments_in_sample": seg, "sampled_rows": len(rows),
            "counted_on": "host build of the generated source, per ray (oracle/host_build.py)"}


def first_trip_origin_flops(scene_name, w, h, options=None):
    """The first-trip copies of the intersection-material snippets (ptl_trace.tpl PTL_FIRST_TRIP) take the ORIGIN half of their
    `transform(uniform matrix, ray)` chains from the prologue kernel: a ray that still starts at the camera does not execute it.  Read
    off the generated source: every `ptl_ray_o(<expression>, <table>)` inside the loop of a `_first` function drops one mat4 x vec4
    (28 operations) per `transform(` of its expression, once per loop iteration; iterations = the value of the loop bound.  None when
    the scene has no such copy."""
    import re

    import portal_amd as pa

    scene = pa.Scene.from_file(_scene_path(pa, scene_name))
    source = scene.generate_source(pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)
    total, sites = 0, []
    for m in re.finditer(r"PTL_FN SceneIntersectionWithMaterial intersect_material_\d+_first\(Ray r(?:, float ptl_far)?\) \{", source):
        body = source[m.end():source.index("\n}\n", m.end())]
        bound = re.search(r"const bool ptl_tab_ok_\d+ = \((\w+)\) <= \d+;", body)
        if not bound:
            continue
        value = re.search(r"#define " + re.escape(bound.group(1)) + r" \((-?\d+)\)", source)
        iterations = int(value.group(1)) if value else None
        if iterations is None:
            continue
        for k in [x.start() for x in re.finditer(r"ptl_ray_o\(", body)]:
            depth, i = 0, k + len("ptl_ray_o(")
            start = i
            while i < len(body) and not (body[i] == "," and depth == 0):
                depth += body[i] == "("
                depth -= body[i] == ")"
                i += 1
            transforms = body[start:i].count("transform(")
            sites.append({"expression": " ".join(body[start:i].split()), "transforms": transforms})
            total += 28 * transforms * iterations
    if not sites:
        return None
    return {"flops_per_first_trip": total, "sites": sites, "iterations": iterations, "note": "28 binary32 operations per mat4 x vec4 (4 x (1 mul + 3 fma))"}


def deferred_update_counts(scene_name, w, h, depth, aa, options=None, row_step=61, camera=None):
    """Deferred loop-carried updates of the generated source: scheduled vs applied per bounce-loop trip, counted on the host build
    (rows row_step/2, +row_step, ... of the full-size frame).  None when the scene has no such update."""
    import re

    import portal_amd as pa
    from oracle import host_build as hb

    scene = pa.Scene.from_file(_scene_path(pa, scene_name))
    source = scene.generate_source(pa.FLAG_COUNT_SEGMENTS)
    updates = re.findall(r"for \(; (ptl_pend_\d+) > 0; --\1\) ([^;]*;)", source)
    if not updates:
        return None
    flops_per_update = {name: 56 * stmt.count("transform(") for name, stmt in updates}
    if len(set(flops_per_update.values())) != 1:
        raise SystemExit(f"deferred updates of different sizes in {scene_name}: extend the accounting")
    source = source.replace("namespace glsl {\n", "namespace glsl {\nstatic long ptl_deferred_stats[2] = {0, 0};\n", 1)
    source = re.sub(r"\+\+(ptl_pend_\d+);", r"++\1; __atomic_fetch_add(&ptl_deferred_stats[0], 1, __ATOMIC_RELAXED);", source)
    source = re.sub(r"for \(; (ptl_pend_\d+) > 0; --\1\) ", r"for (; \1 > 0; --\1, __atomic_fetch_add(&ptl_deferred_stats[1], 1, __ATOMIC_RELAXED)) ", source)
    source += '\nextern "C" long* ptl_deferred_stats_ptr() { return glsl::ptl_deferred_stats; }\n'
    renderer = pa.SceneRenderer(scene, device=-1, **_scene_kw(scene_name))
    renderer.set_option("render_depth", depth)
    renderer.set_option("aa_count", aa)
    for k, v in (options or {}).items():
        name = {"use_panini": "use_panini_projection"}.get(k, k)
        renderer.set_option(name, float(v))
    if camera:
        renderer.set_camera(camera[:3], camera[3], camera[4], camera[5])
    layout, size = scene.uniform_layout()
    hk = hb.HostKernel(source, layout, size, True)
    for name, typ, _ in layout:
        if typ == pa.PTL_SAMPLER:
            continue
        v = renderer.uniform_value(name, w, h)
        if v is not None:
            hk.set_uniform(name, v)
    rows = list(range(row_step // 2, h, row_step))
    import ctypes as C

    hk.lib.ptl_deferred_stats_ptr.restype = C.POINTER(C.c_long)
    st = hk.lib.ptl_deferred_stats_ptr()
    st[0] = st[1] = 0  # the library (and its counters) is shared by every HostKernel of the same source in this process
    out = hk.render(w, h, rows=rows, rgba32f=False)
    seg = out["segments"]
    return {"scheduled_per_segment": st[0] / seg, "applied_per_segment": st[1] / seg, "flops_per_update": next(iter(flops_per_update.values())),
            "segments_in_sample": seg, "sampled_rows": len(rows), "counted_on": "host build of the generated source (oracle/host_build.py)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "profiles", "r05", "flops_per_segment.json"))
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
    # round 5: the headline scene seen INTO the nested portals (bench.py --workload c4-deep: the regime with several trips per primary ray)
    if not args.only or args.only == "portal_in_portal_3840x2160_d40_cam0_0_0_0.2_1.5_1.6":
        out["portal_in_portal_3840x2160_d40_cam0_0_0_0.2_1.5_1.6"] = count("portal_in_portal", 3840, 2160, 40, 1, 0.005, camera=(0.0, 0.0, 0.0, 0.2, 1.5, 1.6))
    # ... and the corpus scene whose default view bounces 26 times per primary ray (bench.py --workload recursive-room): "depth = 40" exercised
    if not args.only or args.only == "recursive_room_3840x2160_d40":
        out["recursive_room_3840x2160_d40"] = count("tests/corpus/scenes/recursive_room.ron", 3840, 2160, 40, 1, 0.002)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
