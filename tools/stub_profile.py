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
    # EXPERIMENTS (must draw the intact picture): uniforms that stay run-time values in a fully baked build, as literals.
    # `teleport_light_u` is kept dynamic because the camera-teleport entry forces it to 1; the `_`-options are the renderer's own switches
    "bake_teleport_light": [("#define teleport_light_u (PTL_U.teleport_light_u)", "#define teleport_light_u (1)")],
    "bake_black_border": [("#define _black_border_disable (PTL_U._black_border_disable)", "#define _black_border_disable (0)")],
    "bake_mode_switches": [("#define teleport_light_u (PTL_U.teleport_light_u)", "#define teleport_light_u (1)")] + [
        ("#define %s (PTL_U.%s)" % (n, n), "#define %s (%s)" % (n, v)) for n, v in (
            ("_black_border_disable", "0"), ("_grid_disable", "0"), ("_angle_color_disable", "0"), ("_darken_by_distance", "1"), ("_use_panini_projection", "0"),
            ("_use_360_camera", "0"), ("_use_180_camera", "0"), ("_draw_depth_map", "0"), ("_draw_anaglyph", "0"), ("_draw_side_by_side", "0"), ("_aa_count", "1"),
            ("_aa_start", "0"), ("_ray_tracing_depth", "40"), ("_teleport_external_ray", "0"))],
    "bake_a_colour_switches": [("#define %s (PTL_U.%s)" % (n, n), "#define %s (%s)" % (n, v)) for n, v in (
        ("_black_border_disable", "0"), ("_grid_disable", "0"), ("_angle_color_disable", "0"), ("_darken_by_distance", "1"))],
    "bake_b_camera_modes": [("#define %s (PTL_U.%s)" % (n, n), "#define %s (%s)" % (n, v)) for n, v in (
        ("_use_panini_projection", "0"), ("_use_360_camera", "0"), ("_use_180_camera", "0"), ("_draw_depth_map", "0"), ("_draw_anaglyph", "0"), ("_draw_side_by_side", "0"))],
    "bake_c_aa_count": [("#define _aa_count (PTL_U._aa_count)", "#define _aa_count (1)")],
    "bake_d_depth": [("#define _ray_tracing_depth (PTL_U._ray_tracing_depth)", "#define _ray_tracing_depth (40)")],
    "bake_abc": [("#define %s (PTL_U.%s)" % (n, n), "#define %s (%s)" % (n, v)) for n, v in (
        ("_black_border_disable", "0"), ("_grid_disable", "0"), ("_angle_color_disable", "0"), ("_darken_by_distance", "1"),
        ("_use_panini_projection", "0"), ("_use_360_camera", "0"), ("_use_180_camera", "0"), ("_draw_depth_map", "0"), ("_draw_anaglyph", "0"), ("_draw_side_by_side", "0"),
        ("_aa_count", "1"))],
    "bake_both": [("#define teleport_light_u (PTL_U.teleport_light_u)", "#define teleport_light_u (1)"),
                  ("#define _black_border_disable (PTL_U._black_border_disable)", "#define _black_border_disable (0)")],
    # every hit is final with a flat colour: ray generation + ONE scene_intersect per sample, no shading, no second trip
    "no_material": [("m = material_process(r, i);", "m = MaterialProcessing{true, vec3(0.5f), ray_none};"),
                    ("m = material_process(r, i2.scene);", "m = MaterialProcessing{true, vec3(0.5f), ray_none};")],
    # planes never hit: what do the Flat objects cost (transform, early-out, sqrt, divisions, is_inside)?
    "no_planes": [("    r = transform(plane_inv, r);\n    float len = length(r.d);", "    return intersection_none;\n    r = transform(plane_inv, r);\n    float len = length(r.d);")],
    # the snippet of the scene (intersection material) never hits
    "no_snippet": [("hit = intersect_material_0(r, ptl_far);", "hit = SceneIntersectionWithMaterial{scene_intersection_none, material_empty()};"),
                   ("hit = intersect_material_0_first(r, ptl_far);", "hit = SceneIntersectionWithMaterial{scene_intersection_none, material_empty()};")],
    # round 4: the chain of early returns that tells which ring of a portal a point is in (scenes/portal_in_portal.ron library `is_inside_portal`)
    "pip_no_is_inside_portal": [("  int material = material_second;\n  if (first) { material = material_first; }\n", "  return NOT_INSIDE;\n  int material = material_second;\n  if (first) { material = material_first; }\n")],
    # ... everything behind the bounding test of is_inside_portal_advanced (the walk through the nested copies + the rings)
    "pip_no_advanced_tail": [("if (definitely_not_in_portal(x, y)) return NOT_INSIDE;\n", "if (definitely_not_in_portal(x, y)) return NOT_INSIDE;\nif (x > -1e30f) return NOT_INSIDE;\n")],
    # ... the three boxes' own hit code (intersect_box) replaced by a miss: transform + normalisation stay
    "pip_no_box_body": [("vec3 rad = vec3(4.f, 4.f, 4.f);\n", "if (r.d.x > -2.f) return scene_intersection_none;\nvec3 rad = vec3(4.f, 4.f, 4.f);\n")],
    # WHAT IF (wrong picture where a portal's back is seen): the `back` argument of the snippet's inside tests cost nothing.  GLSL evaluates
    # is_collinear(hit.n, normal) -- two square roots and a division -- at every call, the callee looks at it on one path in ten
    "pip_b_back_free": [("is_collinear(hit_b.n, normal_b.sw<0,1,2>())", "false")],
    "pip_ab_back_free": [("is_collinear(hit_b.n, normal_b.sw<0,1,2>())", "false"), ("is_collinear(hit_a.n, normal_a)", "false")],
    "planes_back_free": [("is_collinear(hit.n, normal)", "false")],
    # colour of a hit wall: grid + normal shading of material_simple2
    "no_grid": [("#define _grid_disable (PTL_U._grid_disable)", "#define _grid_disable (1)")],
    # the scene snippet of portal_in_portal, piece by piece (scenes/portal_in_portal.ron:1127-1185): the nested copies of portal a ...
    "pip_no_a_part": [("\tif (nearer(result.scene.hit, hit_a)) {", "\tif (false) {")],
    # ... the nested copies of portal b: their inside test and material (the plane test and the ray chain stay)
    "pip_no_b_inside": [("\tif (nearer(result.scene.hit, hit_b)) {", "\tif (false) {")],
    # ... and the plane test of every copy of b as well: what remains is the chain r_b = a * (b0^-1 * r_b) and normal_b
    "pip_no_b_test": [("\tif (nearer(result.scene.hit, hit_b)) {", "\tif (false) {"),
                      ("\tSurfaceIntersection hit_b = plane_intersect_normalized(r3);", "\tSurfaceIntersection hit_b = intersection_none; if (len < 0.f) hit_b = plane_intersect_normalized(r3);")],
    # ... b's plane test alive, its inside test and material never entered
    "pip_b_plane_only": [("\tif (nearer(result.scene.hit, hit_b)) {", "\tif (nearer(result.scene.hit, hit_b) && hit_b.t < -1.0f) {")],
    # ... the loop of is_inside_portal_advanced that walks a hit point back through the nested copies (O(size) per copy, both portals)
    "pip_no_inner_loop": [("    for (int i = 0; i < size; i++) { // !FOR_VARIABLE!\n\t\tif (pos2.z > 0.f)", "    for (int i = 0; i < 0; i++) { // !FOR_VARIABLE!\n\t\tif (pos2.z > 0.f)")],
    # ... b's material (with the deferred chain of r_teleport_b it flushes)
    "pip_b_no_material": [("for (; ptl_pend_1 > 0; --ptl_pend_1) r_teleport_b=transform(a_mat,transform(b0_mat_inv,r_teleport_b)); result.material = material_teleport_transformed(offset_ray(r_teleport_b, hit_b.t), vec3(1.f));",
                           "result.material = material_empty();")],
    # EXPERIMENT (must draw the intact picture): the wave-level plane cull around the hand-inlined plane test of the copies of b
    "pip_b_cull": [("\tRay r3 = transform(b0_mat_inv, r_b);\n\tfloat len = length(r3.d);\n\tr3 = normalize_ray(r3);\n\n\tSurfaceIntersection hit_b = plane_intersect_normalized(r3);\n\tif (hit_b.hit) {\n\t    hit_b.t /= len;\n\t    hit_b.n = normal3;\n\t}\n",
                    "\tSurfaceIntersection hit_b = intersection_none;\n\tif (!ptl_plane_cull(r_b, b0_mat_inv, (result.scene.hit.hit ? result.scene.hit.t : __builtin_inff()))) {\n\tRay r3 = transform(b0_mat_inv, r_b);\n\tfloat len = length(r3.d);\n\tr3 = normalize_ray(r3);\n\thit_b = plane_intersect_normalized(r3);\n\tif (hit_b.hit) {\n\t    hit_b.t /= len;\n\t    hit_b.n = normal3;\n\t}\n\t}\n")],
    # ... the ray-independent chain normal_b = b0 * (a^-1 * normal_b) of the loop (and with it the normalisation inside normalize_normal):
    # does the specialised build fold it, or does every wave recompute it on every trip?
    "pip_no_normal_chain": [("\tnormal_b = b0_mat * (a_mat_inv * normal_b);\n", "\n")],
    # EXPERIMENT (intact picture): the loop over the nested copies fully unrolled -- the chain above becomes literals
    "pip_unrolled": [("int ptl_pend_0 = 0; int ptl_pend_1 = 0; for (int size = 0; size < show_teleported_u; size++) {", "int ptl_pend_0 = 0; int ptl_pend_1 = 0;\n_Pragma(\"unroll\") for (int size = 0; size < show_teleported_u; size++) {")],
    # WHAT IF (wrong picture): the ORIGIN half of the snippet's ray chain cost nothing -- on the first trip every ray of the frame starts
    # at the camera, so r_b.o, r3.o are functions of the iteration number alone and could come from the prologue kernel
    "pip_origin_chain_free": [("\tRay r3 = transform(b0_mat_inv, r_b);\n", "\tRay r3 = r_b; r3.d = b0_mat_inv * r_b.d; r3.o = r.o + vec4(float(size));\n"),
                              ("\tr_b = transform(a_mat, transform(b0_mat_inv, r_b));\n", "\tr_b.d = a_mat * (b0_mat_inv * r_b.d);\n")],
    # Complex objects (scene snippets intersect_<k>) skipped
    "no_complex": [("ihit = intersect_", "if (len < 0.0f) ihit = intersect_")],
    # no bounce loop at all: ray generation, AA loop, gamma, store
    "no_trace": [("    bool alive = true;\n    for (int j = 0; j < _ray_tracing_depth; j++) {",
                  "    bool alive = true;\n    if (r.d.x > 2.0f) return RayTraceResult{vec3(r.d.x, r.d.y, r.d.z), 0.0f, false};\n    for (int j = 0; j < 0; j++) {")],
}

# ---- round 5 experiments on the headline snippet (each must draw the intact picture) ----------------------------------------------------
_WALK_OLD = ("    for (int i = 0; i < size; i++) { // !FOR_VARIABLE!\n\t\tif (pos2.z > 0.f) return NOT_INSIDE;\n\t\tif (first) {\n\t\t\tpos2 = a_mat_inv * (b0_mat * pos2);\n"
             "\t\t} else {\n\t\t\tpos2 = b0_mat_inv * (a_mat * pos2);\n\t\t}\n\t}\n")
_WALK_FLAGS = ("    bool ptl_left = false;\n    for (int i = 0; i < size; i++) { // !FOR_VARIABLE!\n\t\tptl_left = ptl_left || (pos2.z > 0.f);\n\t\tif (first) {\n\t\t\tpos2 = a_mat_inv * (b0_mat * pos2);\n"
               "\t\t} else {\n\t\t\tpos2 = b0_mat_inv * (a_mat * pos2);\n\t\t}\n\t}\n\tif (ptl_left) return NOT_INSIDE;\n")
# the walk through the nested copies with its early return turned into a flag: straight-line code after unrolling (and the a-side walk, whose
# start point pos_a is the same for every copy, becomes a common subexpression of the ten unrolled copies)
PATCHES["r5_walk_flags"] = [(_WALK_OLD, _WALK_FLAGS)]
# rays are affine objects: o.w = 1, d.w = 0 (every matrix of this scene has the bottom row 0 0 0 1): transform() with the constants spelled
_W_OLD1 = "    return Ray{matrix * r.o, matrix * r.d, r.tmul, r.in_subspace};"
_W_NEW1 = "    return Ray{matrix * vec4(r.o.x, r.o.y, r.o.z, 1.0f), matrix * vec4(r.d.x, r.d.y, r.d.z, 0.0f), r.tmul, r.in_subspace};"
_W_OLD2 = "    else return Ray{ptl_mul_m<MASK>(matrix, r.o), ptl_mul_m<MASK>(matrix, r.d), r.tmul, r.in_subspace};"
_W_NEW2 = "    else return Ray{ptl_mul_m<MASK>(matrix, vec4(r.o.x, r.o.y, r.o.z, 1.0f)), ptl_mul_m<MASK>(matrix, vec4(r.d.x, r.d.y, r.d.z, 0.0f)), r.tmul, r.in_subspace};"
PATCHES["r5_w_known"] = [(_W_OLD1, _W_NEW1), (_W_OLD2, _W_NEW2)]
PATCHES["r5_walk_flags_w_known"] = PATCHES["r5_walk_flags"] + PATCHES["r5_w_known"]
# the wave-level cull (one sign test since round 4) around the hand-inlined plane test of every copy of portal b, general and first-trip copy
_BCULL = []
for _ray in ("transform(b0_mat_inv, r_b)", "(ptl_tab_ok_0 ? ptl_ray_o(transform(b0_mat_inv, r_b), PTL_U.ptl_hv1[size]) : (transform(b0_mat_inv, r_b)))"):
    _BCULL.append(("\tRay r3 = " + _ray + ";\n\tfloat len = length(r3.d);\n\tr3 = normalize_ray(r3);\n\n\tSurfaceIntersection hit_b = plane_intersect_normalized(r3);\n\tif (hit_b.hit) {\n\t    ptl_div_assign(hit_b.t , len);\n\t    hit_b.n = normal3;\n\t}\n",
                   "\tSurfaceIntersection hit_b = intersection_none;\n\tif (!ptl_plane_cull(r_b, b0_mat_inv, (result.scene.hit.hit ? result.scene.hit.t : __builtin_inff()))) {\n\tRay r3 = " + _ray +
                   ";\n\tfloat len = length(r3.d);\n\tr3 = normalize_ray(r3);\n\thit_b = plane_intersect_normalized(r3);\n\tif (hit_b.hit) {\n\t    ptl_div_assign(hit_b.t , len);\n\t    hit_b.n = normal3;\n\t}\n\t}\n"))
PATCHES["r5_b_cull"] = _BCULL
PATCHES["r5_all"] = PATCHES["r5_walk_flags_w_known"] + _BCULL


# round 5, STUB_FLAGS=1048576 (the patterns build): which of the Int / Bool uniforms that the Int-baked build has as literals make it 0.27 ms against 0.44?
_PIP_INTS = {"show_teleported_u": 10, "filter_teleported_u": 1, "teleport_light_u": 1, "shape_u": 0, "double_sided_u": 0, "show_arrow_u": 0, "show_object_u": 0,
             "pass_use_teleported_matrix_u": 0, "pass_not_disable_orange_portal_u": 0, "gray_room_u": 0, "show_cube_u": 0}
PATCHES["r5_bake_loop_bound"] = [("#define show_teleported_u (PTL_U.show_teleported_u)", "#define show_teleported_u (10)")]
PATCHES["r5_bake_flags_not_the_bound"] = [("#define %s (PTL_U.%s)" % (n, n), "#define %s (%d)" % (n, v)) for n, v in _PIP_INTS.items() if n != "show_teleported_u"]
PATCHES["r5_bake_all_ints"] = [("#define %s (PTL_U.%s)" % (n, n), "#define %s (%d)" % (n, v)) for n, v in _PIP_INTS.items()]
PATCHES["r5_bake_bound_and_unroll"] = PATCHES["r5_bake_loop_bound"] + [("for (int size = 0; size < show_teleported_u; size++) {", "_Pragma(\"unroll\") for (int size = 0; size < show_teleported_u; size++) {")]


# WHAT IF (this scene has no subspace: same picture): the `if (r.in_subspace == false) {` guard around every generated object test gone
PATCHES["r5_no_subspace_guards"] = [("if (r.in_subspace == false) {", "{")]


def _select_form_is_inside_portal(src):
    """EXPERIMENT (must draw the intact picture): the ring classification of scenes/portal_in_portal.ron's library as straight-line selects instead of
    the author's chain of early returns -- what a select-form rewrite of pure early-return functions in the translator would buy."""
    a = src.index("  int material = material_second;\n  if (first) { material = material_first; }\n")
    b = src.index("PTL_FN int ellipse_portal(")
    body = """  int material = first ? material_first : material_second;
  int inner = back ? material : (teleport_light_u == 1 ? TELEPORT : (first ? grid_material_first : grid_material_second));
  int black_material = (_black_border_disable == 1) ? material : solid_black_M;
  int result = NOT_INSIDE;
  result = (distance < size + black_border + border + black_border) ? black_material : result;
  result = (distance < size + black_border + border) ? material : result;
  result = (distance < size + black_border && !back) ? black_material : result;
  result = (distance < size) ? inner : result;
  return result;
}

"""
    return src[:a] + body + src[b:]


PATCHES["pip_select_is_inside_portal"] = _select_form_is_inside_portal

# ---- round 6: BASELINE's C5, scenes/mobius_monoportal.ron (where do its 12.6 ms go?).  The strip is found by a search: behind a bounding-sphere
# test, 8 seeds (+ 2 when one hit) x <= 10 Newton iterations x 2 evaluations of `mobius_step` (two sines, two cosines, a nearest-points solve) each.
PATCHES["mob_no_search"] = [("    if (intersect_mobius_sphere(r)) {", "    if (intersect_mobius_sphere(r) && r.d.x > 2.f) {")]          # the sphere test stays, nothing behind it
PATCHES["mob_one_seed"] = [("    best = update_best_approx(best, mobius_best_approx(PI, r, max, best));\n", "    return best;\n")]
PATCHES["mob_two_seeds"] = [("    int count1 = 2;\n", "    return best;\n    int count1 = 2;\n")]
PATCHES["mob_four_seeds"] = [("    int count2 = 4;\n", "    return best;\n    int count2 = 4;\n")]
PATCHES["mob_no_refinement_seeds"] = [("    if (best.t < 0.f) {\n        return best;\n    }\n", "    if (best.t > -2.f) {\n        return best;\n    }\n")]
PATCHES["mob_no_newton"] = [("    int count = 10; \n", "    int count = 0; \n")]
PATCHES["mob_newton_3"] = [("    int count = 10; \n", "    int count = 3; \n")]
PATCHES["mob_free_trig"] = [("    return vec3(cos(u), 0, sin(u));", "    return vec3(u, 0, 1.f - u);"),
                            ("    return ptl_div(vec3(cos(ptl_div(u,2.f))*cos(u), sin(ptl_div(u,2.f)), cos(ptl_div(u,2.f))*sin(u)),2.f); // mobius", "    return ptl_div(vec3(u * u, 0.5f * u, u - u * u),2.f); // mobius")]
PATCHES["mob_no_derivative_probe"] = [("        float du = ptl_div(-step.x,(mobius_step(u + eps_der, r).x - step.x))*eps_der;", "        float du = -step.x * 0.5f;")]

if __name__ == "__main__":
    only = [a[2:] for a in sys.argv[1:] if a.startswith("--")]
    sys.argv = [a for a in sys.argv if not a.startswith("--")]
    if only:
        PATCHES = {k: v for k, v in PATCHES.items() if k in only}
    name, w, h, depth = (sys.argv[1:] + ["portal_in_portal", "3840", "2160", "40"])[:4]
    w, h, depth = int(w), int(h), int(depth)
    scene = pa.Scene.from_file(pa.scene_path(name))
    device = -1 if os.environ.get("PTL_VARIANTS_PRECOMPILE") else 0  # -1: no GPU here, only fill the code-object cache
    flags = int(os.environ.get("STUB_FLAGS", str(pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)))  # 0: the un-specialised kernel
    r = pa.SceneRenderer(scene, device=device, flags=flags)
    r.set_option("render_depth", depth)
    source = r.kernel_source()  # the shipped text: the renderer's mode switches compiled in
    defines_of_build = list(scene.generated_defines()) + ([os.environ["STUB_DEFINE"]] if os.environ.get("STUB_DEFINE") else [])
    layout, size = scene.uniform_layout()
    import ctypes as C

    for variant, subs in PATCHES.items():
        src = source
        if (variant.startswith("pip_") and name != "portal_in_portal") or (variant.startswith("mob_") and name != "mobius_monoportal"):
            continue
        if callable(subs):
            src, subs = subs(src), []
        for old, new in subs:
            if old not in src:
                print(json.dumps({"scene": name, "variant": variant, "skipped": "pattern not in this build's source"}), flush=True)
                src = None
                break
            src = src.replace(old, new)
        if src is None:
            continue
        # the defines the generator asks the JIT for (codegen.cpp: first-trip copies, zero terms of baked matrices)
        k = pa.Kernel(src, layout, size, device=device, defines=defines_of_build)
        if device < 0:
            continue
        for uname, typ, _ in layout:
            if typ == pa.PTL_SAMPLER:
                continue
            v = r.uniform_value(uname, w, h)
            if v is not None:
                k.set_uniform(uname, typ, v)
        outs = [k.render(w, h) for _ in range(8)]
        times = [o["ms"] for o in outs]
        import hashlib
        print(json.dumps({"scene": name, "variant": variant, "ms": round(float(np.median(times[2:])), 4), "sha": hashlib.sha1(outs[-1]["rgba8"].tobytes()).hexdigest()[:10]}), flush=True)
