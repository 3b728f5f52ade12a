"""Shared helpers for the reference-text parity tests (tests/test_reference_text.py) and the script
that writes their fixtures (tests/golden/make_reference_text_fixtures.py).

`oracle.reference_shader.ReferenceShader` executes the reference's own GLSL text; this module
generates seeded inputs for every function of that text by parameter type, calls the hand
restatement (`oracle.portal_oracle.Natives` / `Oracle`) with the same inputs and compares bit for bit.
"""
from __future__ import annotations

import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden", "reference_text")

F32, I32 = np.float32, np.int32

# frames rendered from the reference text and committed (scene, w, h, depth, aa, options, overrides, camera)
FRAME_CASES = {
    "basics_64x64_d4_aa1": ("basics", 64, 64, 4, 1, {}, {}, None),
    "monoportal_96x54_d20_aa1": ("monoportal", 96, 54, 20, 1, {}, {}, None),
    "triple_portal_96x54_d40_aa1": ("triple_portal", 96, 54, 40, 1, {}, {}, None),
    "portal_in_portal_96x54_d40_aa1": ("portal_in_portal", 96, 54, 40, 1, {}, {}, None),
    "mobius_monoportal_64x36_d64_aa2": ("mobius_monoportal", 64, 36, 64, 2, {}, {}, None),
    # a view INTO the nested portals (the deep-recursion regime), SURVEY 8d's Panini variant, the other camera modes
    "portal_in_portal_deep_96x54_d40_aa1": ("portal_in_portal", 96, 54, 40, 1, {}, {}, dict(look_at=(0.0, 0.0, 0.0), alpha=0.2, beta=1.5, r=1.6)),
    "portal_in_portal_panini_96x54_d40_aa1": ("portal_in_portal", 96, 54, 40, 1, dict(use_panini=True, panini_param=1.0, view_angle=float(np.radians(140.0))), {}, None),
    "monoportal_360_96x54_d20_aa3": ("monoportal", 96, 54, 20, 3, {}, {"_use_360_camera": I32(1)}, None),
    "triple_portal_180_96x54_d40_aa1": ("triple_portal", 96, 54, 40, 1, {}, {"_use_180_camera": I32(1)}, None),
    "portal_in_portal_depthmap_96x54_d40_aa1": ("portal_in_portal", 96, 54, 40, 1, {}, {"_draw_depth_map": I32(1), "_depth_map_min": F32(1.0), "_depth_map_max": F32(7.5)}, None),
    # side-by-side stereo: the eye matrices come from SceneRenderer::teleport_eye_matrices (src/main.rs:1121-1172), i.e. from
    # teleport_external_ray queries -- `stereo` routes the camera through oracle.portal_oracle.CameraRig / the product's move_camera
    "monoportal_sidebyside_96x54_d20_aa1": ("monoportal", 96, 54, 20, 1, {}, {"_draw_side_by_side": I32(1)}, dict(look_at=(0.2, 0.1, -0.3), alpha=0.9, beta=1.2, r=2.2, stereo=True)),
}

# product option names for the overrides above (portal_amd.SceneRenderer.set_option)
PRODUCT_OPTIONS = {"_use_360_camera": "use_360_camera", "_use_180_camera": "use_180_camera", "_draw_depth_map": "draw_depth_map",
                   "_depth_map_min": "depth_map_min", "_depth_map_max": "depth_map_max", "_draw_side_by_side": "draw_side_by_side"}


def make_tracer(cls, case):
    scene, w, h, depth, aa, options, overrides, camera = FRAME_CASES[case]
    o = cls(os.path.join(ROOT, "scenes", scene + ".ron"))
    o.options.update(render_depth=depth, aa_count=aa, **options)
    o.overrides.update(overrides)
    if camera and camera.get("stereo"):
        from oracle.portal_oracle import CameraRig

        rig = CameraRig(o)
        rig.stereo = True
        rig.move(camera["look_at"], camera["alpha"], camera["beta"], camera["r"])
        o.camera = rig.settings()
    elif camera:
        o.camera = dict(camera)
    return o, w, h


def make_product_renderer(pa, case, flags=0, device=0):
    """The product set up for the same case through its public options (portal_amd.SceneRenderer)."""
    scene, w, h, depth, aa, options, overrides, camera = FRAME_CASES[case]
    r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(scene)), device=device, flags=flags)
    r.set_option("render_depth", depth)
    r.set_option("aa_count", aa)
    if options.get("use_panini"):
        r.set_option("use_panini_projection", 1)
        r.set_option("panini_param", options["panini_param"])
    if "view_angle" in options:
        r.set_option("view_angle", options["view_angle"])
    for k, v in overrides.items():
        r.set_option(PRODUCT_OPTIONS[k], float(v))
    if camera and camera.get("stereo"):
        r.move_camera(camera["look_at"], camera["alpha"], camera["beta"], camera["r"])
    elif camera:
        r.set_camera(camera["look_at"], camera["alpha"], camera["beta"], camera["r"])
    return r, w, h


def render_case(cls, case):
    o, w, h = make_tracer(cls, case)
    return o.render(w, h)


def bits_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == np.float32 or b.dtype == np.float32:
        a, b = a.astype(F32), b.astype(F32)
        return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return a == b


# ---------------------------------------------------------------------------------------------
# seeded inputs by GLSL type
# ---------------------------------------------------------------------------------------------
SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, -3.4e38, 1e10, 1e-10, 0.01, 0.99, 1.01], F32)


def floats(rng, n, special=True):
    scale = np.exp(rng.uniform(-6.0, 6.0, n))
    x = (rng.standard_normal(n) * scale).astype(F32)
    k = rng.random(n)
    x = np.where(k < 0.35, rng.uniform(-2.0, 2.0, n).astype(F32), x)          # the range scene coordinates live in
    x = np.where((k >= 0.35) & (k < 0.45), np.round(rng.uniform(-4.0, 4.0, n) * 2).astype(F32) / 2, x)  # exact halves: ties, zeros
    if special:
        x = np.where(k > 0.97, SPECIAL[rng.integers(0, len(SPECIAL), n)], x)
    return x.astype(F32)


def make_value(ty, rng, n, structs, special=True):
    from oracle.glsl_values import Mat, Struct, Vec

    if ty == "float":
        return floats(rng, n, special)
    if ty == "int":
        return rng.integers(-3, 24, n).astype(I32)
    if ty == "bool":
        return rng.random(n) < 0.5
    if ty.startswith("vec"):
        return Vec([floats(rng, n, special) for _ in range(int(ty[3]))])
    if ty.startswith("mat"):
        k = int(ty[3])
        cols = [[floats(rng, n, special) for _ in range(k)] for _ in range(k)]
        if k == 4:  # mostly affine, as every matrix the host uploads is
            affine = rng.random(n) < 0.8
            for c in range(4):
                cols[c][3] = np.where(affine, F32(1.0 if c == 3 else 0.0), cols[c][3]).astype(F32)
        return Mat([Vec(c) for c in cols])
    if ty == "Ray":
        o, d = make_value("vec4", rng, n, structs, special), make_value("vec4", rng, n, structs, special)
        pt = rng.random(n) < 0.85
        o = Vec(list(o.c[:3]) + [np.where(pt, F32(1.0), o.c[3]).astype(F32)])
        d = Vec(list(d.c[:3]) + [np.where(pt, F32(0.0), d.c[3]).astype(F32)])
        return Struct("Ray", dict(o=o, d=d, tmul=floats(rng, n, special), in_subspace=rng.random(n) < 0.3))
    if ty in structs:
        return Struct(ty, {f: make_value(t, rng, n, structs, special) for t, f in structs[ty]})
    raise TypeError(ty)


def flatten(v):
    """value -> list of (path, ndarray) leaves in a fixed order"""
    from oracle.glsl_values import Mat, Struct, Vec

    if isinstance(v, Vec):
        return [(f"[{i}]", np.asarray(c)) for i, c in enumerate(v.c)]
    if isinstance(v, Mat):
        return [(f"[{j}]{p}", a) for j, col in enumerate(v.cols) for p, a in flatten(col)]
    if isinstance(v, Struct):
        return [(f".{k}{p}", a) for k in v.f for p, a in flatten(v.f[k])]
    return [("", np.asarray(v))]


def leaves_array(v, n):
    """all leaves as one float64-free (n, k) uint32 bit matrix (bools / ints widened), for storage and comparison"""
    cols = []
    for _, a in flatten(v):
        a = np.broadcast_to(a, (n,))
        if a.dtype == np.float32:
            bits = a.view(np.uint32).copy()
            bits[np.isnan(a)] = 0x7FC00000  # one NaN
            cols.append(bits)
        else:
            cols.append(a.astype(np.int64).astype(np.uint32))
    return np.stack(cols, axis=1)


def teleport_segments(scene, count=48):
    """Seeded segments a -> b through the scene's portal region, for `teleport_external_ray`."""
    rng = np.random.default_rng(zlib.crc32(scene.encode()) ^ 0x7E1E)
    out = []
    for k in range(count):
        a = rng.uniform(-3, 3, 3)
        b = rng.uniform(-3, 3, 3) if k % 2 else -a * rng.uniform(0.2, 1.0) + rng.uniform(-0.3, 0.3, 3)  # every other one passes near the origin
        out.append((a, b))
    return out


def function_table(shader):
    """[(name, param types, return type)] of every function of the assembled reference unit except main
    and the scene's own (generated / user) functions -- i.e. library.glsl + frag.glsl."""
    from oracle.glsl_interp import Parser

    out = []
    user = set()
    for _, code in shader.scene.library:
        try:
            user |= {item[2] for item in Parser(code, set(shader._program.structs)).parse_unit() if item[0] == "func"}
        except Exception:
            pass
    skip = {"main", "scene_intersect", "material_process", "scene_intersect_material_process", "ray_tracing", "teleport_external_ray", "get_color",
            "get_color2"}  # need a scene: covered by the frame tests
    for name, overloads in shader._program.funcs.items():
        if name in skip or name in user or name.startswith(("is_inside_", "intersect_", "intersect_material_")):
            continue
        for ptypes, item in overloads:
            if any(t == "sampler2D" for t in ptypes):
                continue
            out.append((name, ptypes, item[1]))
    return out


def seed_for(name, ptypes):
    return zlib.crc32((name + "(" + ",".join(ptypes) + ")").encode()) ^ 0x5EED2026


def make_args(name, ptypes, n, structs, special=True):
    rng = np.random.default_rng(seed_for(name, ptypes))
    args = [make_value(t, rng, n, structs, special) for t in ptypes]
    if name == "PaniniProjection":  # tc in [-1,1]^2, fov in (0, pi), d in [0,1] are its domain; half the lanes stay inside it
        inside = rng.random(n) < 0.5
        from oracle.glsl_values import Vec

        args[0] = Vec([np.where(inside, rng.uniform(-1.6, 1.6, n).astype(F32), c).astype(F32) for c in args[0].c])
        args[1] = np.where(inside, rng.uniform(0.2, 3.0, n).astype(F32), args[1]).astype(F32)
        args[2] = np.where(inside, rng.uniform(0.0, 1.0, n).astype(F32), args[2]).astype(F32)
    if name in ("quasi_random",):
        args[0] = rng.integers(0, 4096, n).astype(I32)
    if name in ("shift_right", "shift_left", "mask_last", "extract_bits"):
        args = [np.abs(np.round(a * 8)).astype(F32) if i == 0 else np.round(rng.uniform(0, 23, n)).astype(F32) for i, a in enumerate(args)]
    return args


def restated(oracle, name, ptypes, args, n):
    """The hand restatement's answer for the same call, or None when the restatement has no such function."""
    from oracle import glsl_math as M
    from oracle import glsl_values as V
    from oracle.glsl_values import Vec

    nat = oracle.nat
    if name == "PaniniProjection":
        return oracle.panini(args[0], args[1], args[2])
    if name == "sample_depth_gradient":
        return oracle.sample_depth_gradient(args[0])
    if name == "anaglyphCombineLinear":
        mode = args[2]
        left = np.stack([np.broadcast_to(c, (n,)) for c in args[0].c], axis=1)
        right = np.stack([np.broadcast_to(c, (n,)) for c in args[1].c], axis=1)
        out = np.where((mode == 0)[:, None], oracle.anaglyph_combine(left, right, 0), oracle.anaglyph_combine(left, right, 1))
        return Vec([out[:, k] for k in range(3)])
    if name == "quasi_random":
        a1 = M.lit("0.7548776662466927600500267982588025643670318456949186300834636687")
        a2 = M.lit("0.5698402909980532659121818632752155853637566123932930564053138358")
        af = args[0].astype(F32)
        return Vec([M.mod(M.add(F32(0.5), M.mul(a1, af)), F32(1.0)), M.mod(M.add(F32(0.5), M.mul(a2, af)), F32(1.0))])
    if name == "Pow2":
        return M.mul(args[0], args[0])
    fn = getattr(nat, name, None)
    if fn is None:
        return None
    return fn(*args)
