"""The affine-rays guard, hunted (VERDICT r5 #2d): `codegen.cpp snippets_keep_rays_affine` decides from the TEXT of a scene's snippets whether every
ray keeps o.w = 1 / d.w = 0, and a kernel generated with PTL_AFFINE_RAYS then never reads a ray's w.  A snippet the scan accepts although it moves a w
draws other pixels than the reference (`/root/reference/src/library.glsl:95-120`: `transform` multiplies all four components) without any error.

Two hunts over randomly assembled, legal GLSL ray writes (index, swizzle, compound, through a local, a user function, a struct member, a loop):
  * 10 000 programs against the SEMANTICS of what they do -- a tiny interpreter here tracks the two w's of every Ray value exactly (it knows
    nothing of the scan's rules): whatever the scan accepts must leave every w where it was;
  * a sample of accepted programs spliced into a scene whose continued rays meet portal transforms: the host build of the generated source draws the
    same bits with PTL_AFFINE_RAYS and without; and -- the hunt's teeth -- programs that DO move a w draw different frames when the scan is bypassed.
"""
import os
import random

import numpy as np
import pytest

from tests import synthetic

# ---- statement templates -------------------------------------------------------------------------------------------------------------------
# Each is (GLSL text, effect).  The effect acts on a model state {ray name: [o.w, d.w]} (floats; None = unknown), written independently of the scan: it
# is what GLSL does to the w components.  `R` is replaced by a ray in scope, `S` by a scalar expression, `V3` by a vec3 expression.
SCALARS = ["0.01", "t", "(a + b)", "hit.t * 0.5", "_offset_after_material", "f1(a, b)", "-t", "a * b / 3.", "sin(a)", "q.z"]
VEC3S = ["vec3(0.01, 0., 0.)", "hit.n * 0.1", "R.d.xyz * 0.5", "vec3(a, b, t)"]   # (R = the ray)


def _keep(state, ray):
    return None


def _set_o(w):
    def f(state, ray):
        state[ray][0] = w
    return f


def _set_d(w):
    def f(state, ray):
        state[ray][1] = w
    return f


def _scale_o(k):
    def f(state, ray):
        state[ray][0] = None if state[ray][0] is None else state[ray][0] * k
    return f


def _unknown_o(state, ray):
    state[ray][0] = None


def _unknown_d(state, ray):
    state[ray][1] = None


TEMPLATES = [
    # --- writes that cannot reach a w  (@ = a ray in scope, $S = a scalar expression, $V = a vec3 expression)
    ("@.o.x += $S;", _keep), ("@.o.y = $S;", _keep), ("@.d.z *= 0.5;", _keep), ("@.o[0] = $S;", _keep), ("@.o[1] -= $S;", _keep), ("@.d[2] += 0.01;", _keep),
    ("@.o.xyz += $V;", _keep), ("@.o.xy = @.o.yx;", _keep), ("@.d.zx = vec2(0.6, 0.8);", _keep), ("@.o.rgb -= $V;", _keep), ("@.d.stp = normalize(@.d.stp);", _keep),
    ("@.o.z++;", _keep), ("--@.d.x;", _keep), ("@.o += @.d * $S;", _keep), ("@.o = @.o + @.d * $S;", _keep), ("@.d = normalize(@.d);", _keep),
    ("@.o = vec4(@.o.xyz + $V, 1.);", _set_o(1.0)), ("@.d = vec4(normalize(@.d.xyz + $V), 0.);", _set_d(0.0)), ("@.o = vec4(@.o.x, $S, @.o.z, 1.0);", _set_o(1.0)),
    ("@ = transform(pa_mat, transform(pa_mat_inv, @));", _keep), ("@ = normalize_ray(@);", _keep), ("@ = offset_ray(@, $S);", _keep), ("@ = nudge(@, $S);", _keep),
    ("@ = Ray(vec4(@.o.xyz, 1.), vec4(@.d.xyz, 0.), @.tmul, @.in_subspace);", lambda st, r: st.__setitem__(r, [1.0, 0.0])),
    ("{ Holder h = Holder(@, 1.); h.ray.o.x += $S; h.ray.o += h.ray.d * $S; @ = h.ray; }", _keep),
    ("for (int k = 0; k < 2; k++) { @.o[2] += 0.01; @.o += @.d * 0.01; }", _keep), ("float w_%d = @.o.w + @.d[3] + @.o[3];", _keep), ("@.o = vec4(@.o.xyz, 1);", _set_o(1.0)),
    # --- writes that move (or may move) a w
    ("@.o.w = 2.;", _set_o(2.0)), ("@.o[3] = 2.0;", _set_o(2.0)), ("@.d[3] += 1.0;", _set_d(1.0)), ("@.d.w = 0.5;", _set_d(0.5)), ("@.o.a *= 2.;", _scale_o(2.0)), ("@.o.q = 3.;", _set_o(3.0)),
    ("@.o.xw = vec2(@.o.x, 2.);", _set_o(2.0)), ("@.o.wzyx.x = 2.;", _set_o(2.0)), ("@.o.xyzw = vec4(@.o.xyz, 4.);", _set_o(4.0)), ("@.d.xyzw.w = 1.;", _set_d(1.0)),
    ("@.o[k3] = 2.;", _set_o(2.0)), ("@.o[1 + 2] = 2.;", _set_o(2.0)), ("@.o.w++;", _scale_o(2.0)), ("++@.o[3];", _scale_o(2.0)), ("@.o *= 2.;", _scale_o(2.0)), ("@.o /= 4.;", _scale_o(0.25)),
    ("@.o = @.o * 2.;", _scale_o(2.0)), ("@.o -= vec4(0., 0., 0., 0.5);", _set_o(0.5)), ("@.d = @.d + vec4(0., 0., 0., 1.);", _set_d(1.0)), ("@.d = -@.d + vec4(0., 0., 0., 0.25);", _set_d(0.25)),
    ("@.o += @.d * $S + vec4(0., 0., 0., 5.);", _set_o(6.0)), ("@.o = @.o + @.d * $S + vec4(0., 0., 0., 5.);", _set_o(6.0)), ("@.o += @.d * $S - vec4(0., 0., 0., 0.5);", _set_o(0.5)),
    ("@.o += @.d * $S, @.o.w = 2.;", _set_o(2.0)), ("{ vec4 v = @.o; v.w = 3.; @.o = v; }", _set_o(3.0)), ("{ vec4 v = vec4(@.d.xyz, 1.); @.d = v; }", _set_d(1.0)),
    ("@.o = vec4(@.o.xyz, 2.);", _set_o(2.0)), ("@.d = vec4(@.d.xyz, 1.);", _set_d(1.0)),
    ("@ = Ray(@.o * 2., @.d, @.tmul, @.in_subspace);", _scale_o(2.0)), ("@ = Ray(vec4(@.o.xyz, 2.), vec4(@.d.xyz, 0.), @.tmul, @.in_subspace);", _set_o(2.0)),
    ("@ = transform(mat4(2.), @);", _scale_o(2.0)), ("{ mat4 m2 = mat4(2.); @ = transform(m2, @); }", _scale_o(2.0)), ("@ = transform(inverse(pa_mat) * 2., @);", _scale_o(2.0)),
    ("{ mat4 pa_mat_inv = mat4(2.); @ = transform(pa_mat_inv, @); }", _scale_o(2.0)),
    ("{ mat4 m2 = mat4(0.); m2[3] = vec4(0., 0., 0., 1.); @.o += @.d * $S + m2[3]; }", _set_o(2.0)), ("@ = widen(@);", _set_o(2.0)), ("@ = heavy(@).ray;", _set_d(1.0)),
    ("{ Holder h = Holder(@, 1.); h.ray.o[3] = 2.; @ = h.ray; }", _set_o(2.0)), ("{ Holder h = Holder(@, 1.); h.ray.d.w += 1.; @ = h.ray; }", _set_d(1.0)),
    ("for (int k = 0; k < 4; k++) { @.o[k] += 0.25; }", _set_o(1.25)), ("set_w(@.o.w);", _set_o(2.0)), ("bump(@);", _set_o(2.0)), ("{ float ip; float fr = modf(2.5, ip); @.o.w = ip; }", _set_o(2.0)),
    ("SET_W(@);", _set_o(2.0)), ("@.HALF.w = 2.;", _set_o(2.0)), ("@ = ray_none;", lambda st, r: st.__setitem__(r, [0.0, 0.0])),
]


def _is_clean(effect):
    st = {"r": [1.0, 0.0]}
    effect(st, "r")
    return st["r"] == [1.0, 0.0]


CLEAN = [k for k, (_, effect) in enumerate(TEMPLATES) if _is_clean(effect)]


def instantiate(text, ray, rng, n=0):
    text = text.replace("@", ray)
    while "$V" in text:
        text = text.replace("$V", rng.choice(VEC3S).replace("R", ray), 1)
    while "$S" in text:
        text = text.replace("$S", rng.choice(SCALARS), 1)
    return text % n if "%d" in text else text


# the library the statements call: two clean helpers, and the ways a function / struct / macro can move a w out of the statement's sight
LIBRARY_CLEAN = ("struct Holder { Ray ray; float k; };\nfloat f1(float x, float y) { return x * 0.5 + y; }\nRay nudge(Ray q2, float s) { q2.o += q2.d * s; q2.o.x += 0.01; return q2; }\n")
LIBRARY_DIRTY = {
    "widen(": "Ray widen(Ray q2) { q2.o[3] = 2.; return q2; }\n", "heavy(": "Holder heavy(Ray q2) { q2.d.w += 1.; return Holder(q2, 0.); }\n",
    "set_w(": "void set_w(out float w) { w = 2.; }\n", "bump(": "void bump(inout Ray q2) { q2.o.w = 2.; }\n",
    "SET_W(": "#define SET_W(ray) ray.o.w = 2.\n", ".HALF": "#define HALF o\n",
}
PRELUDE = "float a = 0.3; float b = 0.2; float t = 0.05; vec3 q = vec3(0.1, 0.2, 0.3); int k3 = 3;\n"


def random_program(rng):
    """(statements as GLSL, library text, True when every Ray value has kept o.w == 1 and d.w == 0 throughout -- by the model above, not by the scan)"""
    rays = ["r"]
    state = {"r": [1.0, 0.0]}
    lines, library = [], LIBRARY_CLEAN
    if rng.random() < 0.4:
        lines.append("Ray r2 = r;")
        rays.append("r2")
        state["r2"] = [1.0, 0.0]
    moved = False
    for n in range(rng.randint(1, 4)):
        text, effect = TEMPLATES[rng.randrange(len(TEMPLATES))] if rng.random() < 0.45 else TEMPLATES[rng.choice(CLEAN)]
        ray = rng.choice(rays)
        text = instantiate(text, ray, rng, n)
        for needle, definition in LIBRARY_DIRTY.items():
            if needle in text and definition not in library:
                library += definition
        effect(state, ray)
        if rng.random() < 0.2:  # spacing and comments in odd places
            text = text.replace(" = ", " /* = */ = ", 1).replace(".o", " . o", 1) if rng.random() < 0.5 else "// " + text + "\n" + text
        lines.append(text)
        moved = moved or any(w != [1.0, 0.0] for w in state.values())   # (a later statement may put a w back: the ray was not affine in between)
    return "\n".join(lines), library, not moved


def test_nothing_the_scan_accepts_moves_a_w(pa):
    rng = random.Random(20260930)
    accepted = refused_clean = dirty = 0
    for _ in range(10_000):
        body, library, clean = random_program(rng)
        ok, why = pa.snippets_keep_rays_affine(library + "MaterialProcessing bend(SurfaceIntersection hit, Ray r) {\n" + PRELUDE + body + "\nreturn material_next(vec3(0.9), r);\n}\n")
        if ok:
            accepted += 1
            assert clean, f"accepted although a w moves:\n{library}\n{body}"
        else:
            assert why
            refused_clean += 1 if clean else 0
            dirty += 0 if clean else 1
    print(f"10 000 random programs: {accepted} accepted (all clean), {dirty} refused that move a w, {refused_clean} refused although clean")
    assert accepted >= 1500 and dirty >= 4000   # the hunt saw plenty of both kinds
    # every clean template by itself is accepted: the whitelist is not vacuous
    assert len(CLEAN) >= 25
    for k in CLEAN:
        code = instantiate(TEMPLATES[k][0], "r", random.Random(k))
        assert pa.snippets_keep_rays_affine(LIBRARY_CLEAN + "void f(Ray r, SurfaceIntersection hit) {" + PRELUDE + code + "}") == (True, ""), code
    # ... and every template that moves a w is refused by itself
    for k, (text, _) in enumerate(TEMPLATES):
        if k not in CLEAN:
            code = instantiate(text, "r", random.Random(k))
            library = LIBRARY_CLEAN + "".join(d for needle, d in LIBRARY_DIRTY.items() if needle in code)
            ok, why = pa.snippets_keep_rays_affine(library + "void f(Ray r, SurfaceIntersection hit) {" + PRELUDE + code + "}")
            assert not ok and why, code


# ---- frames ---------------------------------------------------------------------------------------------------------------------------------
def bend_scene(body, library):
    """A pane at z = 0.9 whose material runs `body` on the ray and lets it go on; behind it a portal pair and the textured wall: every continued ray
    meets generated plane tests (`X_mat_inv * r.o`: where PTL_AFFINE_RAYS spells w = 1) and half of them a teleport."""
    esc = lambda s: s.replace("\\", "\\\\").replace('"', '\\"')
    matrices = '''
        (name: "pane", data: Simple(offset: (0.0, 0.0, 0.9), scale: 1.0, rotate: (0.0, 0.0, 0.0), mirror: (false, false, false))),
        (name: "pa", data: Simple(offset: (-0.45, 0.1, 0.5), scale: 0.4, rotate: (0.0, 0.35, 0.0), mirror: (false, false, false))),
        (name: "pb", data: Simple(offset: (0.5, -0.1, 0.4), scale: 0.4, rotate: (0.1, -0.4, 0.2), mirror: (false, false, false))),
    '''
    objects = '''
        (name: "pane", data: Flat(kind: Simple(Some(Named("pane"))), is_inside: (("if (abs(x) < 0.7 && abs(y) < 0.7) { return bend_M; } return NOT_INSIDE;")), in_subspace: Normal)),
        (name: "gate", data: Flat(kind: Portal(Some(Named("pa")), Some(Named("pb"))), is_inside: (("if (x * x + y * y < 1.) { return TELEPORT; } return NOT_INSIDE;")), in_subspace: Normal)),
    '''
    code = "r.o += r.d * _offset_after_material;\n" + PRELUDE + body + "\nreturn material_next(vec3(0.9, 0.95, 1.0), r);"
    materials = f'(name: "bend", data: Complex(code: (("{esc(code)}")))),'
    lib = f'(name: "helpers", data: (("{esc(library)}"))),'
    return synthetic.wall_scene(r=2.2, color=(0.9, 0.5, 0.2), normal_coef=0.5, grid=True, size=1.6, extra_objects=objects, extra_matrices=matrices, extra_materials=materials, library=lib)


def _frames(pa, text, flags, w=48, h=32):
    from oracle import host_build as hb

    out = []
    for extra in (0, pa.FLAG_NO_AFFINE_RAYS):
        sc = pa.Scene.from_text(text)
        r = pa.SceneRenderer(sc, device=-1, flags=flags | extra | pa.FLAG_QUICK_JIT)
        r.set_option("render_depth", 6)
        sc.generate_source(flags | extra)
        out.append((hb.host_kernel_for(r, sc, w, h, flags=flags | extra).render(w, h)["rgba32f"].copy(), "PTL_AFFINE_RAYS" in sc.generated_defines()))
    return out


def _accepted_programs(pa, seed, want):
    rng = random.Random(seed)
    found = []
    while len(found) < want:
        body, library, clean = random_program(rng)
        ok, _ = pa.snippets_keep_rays_affine(library + "void f(Ray r, SurfaceIntersection hit) {" + PRELUDE + body + "}")
        if ok and body.count("\n") >= 1:
            found.append((body, library))
    return found


@pytest.mark.parametrize("chunk", range(6))
def test_accepted_programs_draw_the_same_bits_with_and_without_affine_rays(pa, chunk):
    spec = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    for body, library in _accepted_programs(pa, 600 + chunk, 4):
        (with_affine, has), (general, has_not) = _frames(pa, bend_scene(body, library), spec)
        assert has and not has_not, body   # the scene as a whole passes the scan, and bit 23 switches the optimisation off
        assert np.array_equal(with_affine.view(np.uint32), general.view(np.uint32)), body
        assert len(np.unique(with_affine.reshape(-1, 4), axis=0)) > 40


@pytest.mark.parametrize("body", ["r.o[3] = 2.0;", "r.d[3] += 1.0;", "r.o += r.d * t + vec4(0., 0., 0., 5.);", "{ mat4 m2 = mat4(2.); r = transform(m2, r); }", "{ Holder h = Holder(r, 1.); h.ray.o[3] = 0.5; r = h.ray; }"])
def test_the_frame_comparison_has_teeth(pa, body, monkeypatch):
    """Round 5's holes, each in the pane's material: refused by the scan -- the kernel is generated WITHOUT affine rays and
    draws the general products' frame -- and, with the scan bypassed (the test hook PTL_AFFINE_RAYS_SKIP_SCAN), the affine-rays kernel draws another
    frame: what an unsound scan would have shipped silently, and what the hunt above would have caught."""
    spec = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    text = bend_scene(body, LIBRARY_CLEAN)
    ok, why = pa.snippets_keep_rays_affine(LIBRARY_CLEAN + "void f(Ray r) {" + PRELUDE + body + "}")
    assert not ok and why
    (refused, has), (general, _) = _frames(pa, text, spec)
    assert not has and np.array_equal(refused.view(np.uint32), general.view(np.uint32))
    monkeypatch.setenv("PTL_AFFINE_RAYS_SKIP_SCAN", "1")
    (forced, has), (general2, has_not) = _frames(pa, text, spec)
    assert has and not has_not and np.array_equal(general2.view(np.uint32), general.view(np.uint32))
    differing = int((forced.view(np.uint32) != general.view(np.uint32)).any(axis=2).sum())
    print(body, "->", differing, "of", forced.shape[0] * forced.shape[1], "pixels differ once the scan is bypassed")
    assert differing > 20


# ---- the dynamic belt: the checking build (PTL_CHECK_AFFINE) ---------------------------------------------------------------------------------
def _violations(pa, scene, flags, w=48, h=32, depth=6):
    from oracle import host_build as hb

    r = pa.SceneRenderer(scene, device=-1, flags=flags | pa.FLAG_CHECK_AFFINE | pa.FLAG_QUICK_JIT)
    r.set_option("render_depth", depth)
    scene.generate_source(flags | pa.FLAG_CHECK_AFFINE)
    assert "PTL_CHECK_AFFINE" in scene.generated_defines() and "PTL_AFFINE_RAYS" not in scene.generated_defines()
    out = hb.host_kernel_for(r, scene, w, h, flags=flags | pa.FLAG_CHECK_AFFINE, count_segments=True).render(w, h)
    return out["segments"], out["rgba32f"]


def test_the_checking_build_counts_nothing_on_clean_scenes_and_something_on_every_hole(pa):
    """VERDICT r5 #2c: a build that does not depend on the scan.  FLAG_CHECK_AFFINE = the general products, and every place where an affine-rays
    kernel assumes a w (matrix x origin, matrix x direction, the bounce loop) counts the halves that arrive with another one.  Zero on the five
    BASELINE scenes and on programs the scan accepts; above zero on each of round 5's holes -- with the frame of the general products either way."""
    spec = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    for name in ("basics", "monoportal", "triple_portal", "portal_in_portal", "mobius_monoportal"):
        for flags in (spec, pa.FLAG_SPECIALIZE_PATTERNS):
            count, _ = _violations(pa, pa.Scene.from_file(pa.scene_path(name)), flags, depth=8)
            assert count == 0, (name, flags, count)
    for body, library in _accepted_programs(pa, 77, 3):
        count, frame = _violations(pa, pa.Scene.from_text(bend_scene(body, library)), spec)
        assert count == 0, body
    for body in ["r.o[3] = 2.0;", "r.d[3] += 1.0;", "r.o += r.d * t + vec4(0., 0., 0., 5.);", "{ mat4 m2 = mat4(2.); r = transform(m2, r); }", "q.x = r.o.x; { Holder h = Holder(r, 1.); h.ray.o[3] = 0.5; r = h.ray; }"]:
        text = bend_scene(body, LIBRARY_CLEAN)
        count, frame = _violations(pa, pa.Scene.from_text(text), spec)
        (_, has), (general, _) = _frames(pa, text, spec)
        print(body, "->", count, "ray halves with another w")
        assert count > 20 and not has, body
        assert np.array_equal(frame.view(np.uint32), general.view(np.uint32)), body   # the checking build IS the general build, plus a counter


@pytest.fixture(scope="module")
def gpu(pa):
    if pa.device_count() < 1:
        pytest.fail("no HIP device visible: the render path has no CPU fallback")
    return pa


@pytest.mark.gpu
def test_a_renderer_that_checks_finds_the_hole_and_switches_affine_rays_off(gpu, monkeypatch):
    """The belt on the device: with the scan bypassed a renderer gets an affine-rays kernel for a snippet that writes a w; `check_affine` (the option,
    as `portal-amd check` and PTL_CHECK_AFFINE=1 use it) counts the violations on the GPU, switches the assumption off and rebuilds -- the frame
    drawn afterwards is the general products' frame.  A clean scene counts zero and keeps its kernel."""
    pa = gpu
    spec = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    w, h = 96, 64
    text = bend_scene("r.o[3] = 2.0;", LIBRARY_CLEAN)
    general = pa.SceneRenderer(pa.Scene.from_text(text), device=0, flags=spec | pa.FLAG_NO_AFFINE_RAYS | pa.FLAG_QUICK_JIT)
    general.set_option("render_depth", 6)
    want = general.draw(w, h, rgba32f=True)["rgba32f"]
    monkeypatch.setenv("PTL_AFFINE_RAYS_SKIP_SCAN", "1")
    broken = pa.SceneRenderer(pa.Scene.from_text(text), device=0, flags=spec | pa.FLAG_QUICK_JIT)
    broken.set_option("render_depth", 6)
    assert broken.affine_rays()
    wrong = broken.draw(w, h, rgba32f=True)["rgba32f"]
    assert int((wrong.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum()) > 50   # what would have shipped silently
    n = broken.check_affine()
    assert n > 0 and not broken.affine_rays() and broken.rejit_count() == 1 and "affine rays switched off" in pa.last_error()
    assert np.array_equal(broken.draw(w, h, rgba32f=True)["rgba32f"].view(np.uint32), want.view(np.uint32))
    # the option: checked before the first draw of a new source, without being asked
    auto = pa.SceneRenderer(pa.Scene.from_text(text), device=0, flags=spec | pa.FLAG_QUICK_JIT)
    auto.set_option("render_depth", 6)
    auto.set_option("check_affine", 1)
    assert np.array_equal(auto.draw(w, h, rgba32f=True)["rgba32f"].view(np.uint32), want.view(np.uint32)) and not auto.affine_rays()
    monkeypatch.delenv("PTL_AFFINE_RAYS_SKIP_SCAN")
    for name in ("portal_in_portal", "mobius_monoportal"):
        clean = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(name)), device=0, flags=spec | pa.FLAG_QUICK_JIT)
        clean.set_option("render_depth", 12)
        assert clean.affine_rays() and clean.check_affine(128, 72) == 0 and clean.affine_rays() and clean.rejit_count() == 0
