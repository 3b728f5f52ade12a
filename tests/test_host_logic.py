"""CPU tests of the host side (no GPU): C ABI surface, RON reader, formula evaluator, matrix
graph, template engine, GLSL->C++ translator, generator bookkeeping.

Where the reference has a unit test for the component, it is restated here with the same
inputs and expected values (src/code_generation.rs:100-184).  Everything else is pinned by
hand-derived values and by agreement with the independent oracle implementation.
"""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = ["basics", "monoportal", "triple_portal", "portal_in_portal", "mobius_monoportal"]


# ---------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(pa):
    header = open(os.path.join(ROOT, "include", "portal_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(ptl_[a-z0-9_]+)\s*\(", header))
    assert len(names) > 40
    lib = C.CDLL(pa.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, f"declared in include/portal_amd.h but not exported: {missing}"


def test_version_and_no_gpu_paths_fail_loudly(pa):
    assert "portal_amd" in pa.version()
    scene = pa.Scene.from_file(pa.scene_path("basics"))
    r = pa.SceneRenderer(scene, device=-1)  # compile-only
    assert len(r.code_object()) > 1000 and r.code_object()[:4] == b"\x7fELF"
    with pytest.raises(pa.PortalError):  # no CPU fallback for rendering
        r.draw(16, 16)
    frame = pa.Frame(64, 20, 1, 2, 1)  # in_place only changes where rows are stored, not which rows a launch renders
    assert frame.in_place == 1 and pa.shard_rows(frame) == pa.shard_rows(pa.Frame(64, 20, 1, 2)) == 8
    import torch

    if not torch.cuda.is_available():  # peer frame buffers are device memory: nothing to hand out here
        with pytest.raises(pa.PortalError):
            pa.ipc_export(pa.device_alloc(4096, 0))
        with pytest.raises(pa.PortalError):
            pa.ipc_open(bytes(64), 0)


def test_unknown_scene_and_bad_ron(pa):
    with pytest.raises(pa.PortalError):
        pa.Scene.from_file("/nonexistent/scene.ron")
    with pytest.raises(pa.PortalError, match="line 3"):
        pa.Scene.from_text("(\n cam: (look_at: (0,0,0),\n alpha: @")


# ---------------------------------------------------------------------------------------------
# template engine: the reference's own unit test, src/code_generation.rs:100-184
# ---------------------------------------------------------------------------------------------
def test_string_storage_and_apply_template_reference_vectors(pa):
    L = pa.lib()
    s1 = L.ptl_strstore_new()
    L.ptl_strstore_add_string(s1, b"1\n2\n3\n")
    L.ptl_strstore_add_identifier_string(s1, b"CustomId1", b"0", b"\n4\n5\n")
    assert L.ptl_strstore_text(s1) == b"1\n2\n3\n\n4\n5\n"
    assert L.ptl_strstore_current_line(s1) == 7
    a, b = C.c_int(), C.c_int()
    assert L.ptl_strstore_range(s1, b"CustomId1", b"0", C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (4, 8)

    s2 = L.ptl_strstore_new()
    L.ptl_strstore_add_string(s2, b"a\nb")
    L.ptl_strstore_add_identifier_string(s2, b"CustomId2", b"1", b"c\nd")
    assert L.ptl_strstore_text(s2) == b"a\nbc\nd"
    assert L.ptl_strstore_current_line(s2) == 3
    assert L.ptl_strstore_range(s2, b"CustomId2", b"1", C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (2, 4)

    names = (C.c_char_p * 2)(b"s1", b"s2")
    stores = (C.c_void_p * 2)(s1, s2)
    s = L.ptl_apply_template(b"abc\n//%s2//%\n\ne\nf\n//%s1//%\n9", names, stores, 2)
    assert L.ptl_strstore_text(s) == b"abc\na\nbc\nd\n\ne\nf\n1\n2\n3\n\n4\n5\n\n9"
    assert L.ptl_strstore_current_line(s) == 15
    assert L.ptl_strstore_range(s, b"CustomId1", b"0", C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (11, 15)
    assert L.ptl_strstore_range(s, b"CustomId2", b"1", C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (3, 5)
    # LineNumbersByKey::get_identifier: line 12 is local line 2 of CustomId1; line 1 belongs to nobody
    kind, name, local = C.create_string_buffer(32), C.create_string_buffer(32), C.c_int()
    assert L.ptl_strstore_get_identifier(s, 12, kind, 32, name, 32, C.byref(local)) == 0
    assert (kind.value, name.value, local.value) == (b"CustomId1", b"0", 2)
    assert L.ptl_strstore_get_identifier(s, 1, kind, 32, name, 32, C.byref(local)) == 1
    L.ptl_strstore_free(s)


# ---------------------------------------------------------------------------------------------
# formulas: hand values + agreement with the oracle's independent evaluator
# ---------------------------------------------------------------------------------------------
FORMULA_KAT = [  # (text from the config scenes, variables, expected by hand)
    ("0.035", {}, 0.035),
    ("1 - progress * 1.73", {"progress": 0.25}, 1 + (-(0.25 * 1.73))),
    ("deg2rad(-90) * min(progress, 0.5) / 0.5", {"progress": 0.3}, (-90 / 180.0 * math.pi) * 0.3 * (1 / 0.5)),
    ("-room_size_x", {"room_size_x": 4.0}, -4.0),
    ("(2^0.5)/2", {}, math.pow(2, 0.5) * (1 / 2.0)),
    ("((2^0.5)/2+border_size+black_border_size)*sin(pass_angle)*pass_scale * (1-2*pass_progress)",
     {"border_size": 0.035, "black_border_size": 0.01, "pass_angle": math.pi / 4, "pass_scale": 1.0, "pass_progress": 0.0},
     (math.pow(2, 0.5) * 0.5 + 0.035 + 0.01) * math.sin(math.pi / 4) * 1.0 * (1 + (-(0.0 * 2)))),
    ("if(progress == 0, 0.001, 0)+on(progress, 0., 0.8)", {"progress": 0.4}, 0.0 + 0.5),
    ("t1%180", {"t1": 200.0}, 20.0),
    ("2^3^2", {}, 512.0),
    ("3-2-1", {}, 0.0),
    ("10/2/5", {}, 10 * 0.5 * 0.2),
    ("1+2*3", {}, 7.0),
    ("lerp(1.5, 5, easing_in_out(time*2))", {}, (1 - 0.5) * 1.5 + 0.5 * 5),  # time = 0.25 -> easing_in_out(0.5) = 0.5
    ("pi()/2 - e()^0", {}, math.pi * 0.5 + (-1.0)),
    ("switch(2, 10, 20, 30)", {}, 20.0),
    ("sqrt(r1^2-o2^2)", {"r1": 5.0, "o2": 3.0}, 4.0),
    ("and(1, not(0)) + or(0, 0)", {}, 1.0),
]


@pytest.mark.parametrize("text,variables,expected", FORMULA_KAT)
def test_formula_known_answers(pa, text, variables, expected):
    from oracle import formula as OF

    got = pa.formula_eval(text, variables, time=0.25)
    assert got is not None and got == pytest.approx(expected, rel=1e-15, abs=1e-15)
    node = OF.compile_formula(text)

    def ns(name, args):
        known, v = OF.custom_function(name, args)
        if known:
            return v
        return 0.25 if name in ("time", "total_time") else variables.get(name)

    assert OF.evaluate(node, ns) == got  # two independent implementations, bit-equal binary64


def test_formula_errors(pa):
    assert pa.formula_eval("1 +") is None
    assert pa.formula_eval("unknown_name * 2") is None
    assert pa.formula_eval("(1 + 2") is None


def test_every_formula_of_the_config_scenes_agrees_with_the_oracle(pa):
    from oracle.scene_eval import OracleScene

    n = 0
    for name in SCENES:
        ps = pa.Scene.from_file(pa.scene_path(name))
        osc = OracleScene(pa.scene_path(name))
        for k, (uname, kind, payload) in enumerate(osc.uniforms):
            if uname is None:
                continue
            want = osc.eval_uniform(k)
            got = ps.eval_uniform(uname)
            if want is None:
                assert got is None
                continue
            assert got == want[1], (name, uname, payload)
            n += 1
    assert n > 80


# ---------------------------------------------------------------------------------------------
# matrix graph + uniform upload values
# ---------------------------------------------------------------------------------------------
def test_matrix_simple_is_translate_rotate_scale(pa):
    text = open(pa.scene_path("portal_in_portal")).read()
    s = pa.Scene.from_text(text)
    tr = s.eval_matrix("tr")  # Simple(offset (0.27,0.39,-1.01), scale 0.1, rotate (pi/2,0,0))
    assert tr[:3, 3] == pytest.approx([0.27, 0.39, -1.01])
    c, sn = math.cos(math.pi / 2), math.sin(math.pi / 2)
    want = 0.1 * np.array([[1, 0, 0], [0, c, -sn], [0, sn, c]])
    assert tr[:3, :3] == pytest.approx(want, abs=1e-15)
    # Teleport: b1 = b0 * a^-1 * b0   (matrix.rs:520-529)
    a, b0, b1 = s.eval_matrix("a"), s.eval_matrix("b0"), s.eval_matrix("b1")
    assert b1 == pytest.approx(b0 @ np.linalg.inv(a) @ b0, abs=1e-12)
    # Mul{to, what} = what * to   (matrix.rs:514-518)
    cube2, cube = s.eval_matrix("cube2"), s.eval_matrix("cube")
    assert cube2 == pytest.approx(cube @ np.diag([0.07, 0.07, 0.07, 1.0]), abs=1e-15)
    # a zero-scale matrix evaluates (its inverse is non-finite, like the reference's glam inverse)
    c0 = s.eval_matrix("c0")
    assert np.all(c0[:3, :3] == 0)
    assert not np.isfinite(s.uniform_values()["c0_mat_inv"]).all()


def test_uniform_overrides_and_time(pa):
    s = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    z0 = s.eval_matrix("b0")[2, 3]
    assert z0 == 1.0  # "1 - progress * 1.73" at progress 0
    assert s.set_uniform("progress", 0.5)
    assert s.eval_matrix("b0")[2, 3] == pytest.approx(1 - 0.5 * 1.73)
    assert not s.set_uniform("no_such_uniform", 1.0)
    assert s.eval_uniform("show_teleported") == 10 and s.eval_uniform("teleport_light") is True


@pytest.mark.parametrize("name", SCENES)
def test_scene_uniform_uploads_match_oracle_bit_for_bit(pa, name):
    """X_mat / X_mat_inv / A_to_B_mat_teleport / user uniforms as binary32: product == oracle."""
    from oracle.scene_eval import OracleScene

    got = pa.Scene.from_file(pa.scene_path(name)).uniform_values()
    want = OracleScene(pa.scene_path(name)).scene_uniform_values()
    # inline matrices have implementation-chosen names on both sides: compare them as multisets
    def split(d):
        named = {k: v for k, v in d.items() if not re.match(r"^(id\d+|oracle_inline\d+)_", k)}
        inline = sorted(np.asarray(v, np.float32).T.reshape(-1).view(np.uint32).tolist() if np.asarray(v).shape == (4, 4)
                        else np.asarray(v, np.float32).reshape(-1).view(np.uint32).tolist() for k, v in d.items() if k not in named)
        return named, inline

    gn, gi = split(got)
    wn, wi = split(want)
    assert set(gn) == set(wn)
    for k in gn:
        g, w = gn[k], wn[k]
        if np.asarray(g).shape == (4, 4):
            g = np.asarray(g, np.float32).T.reshape(16)  # m[row, col] -> column-major
        gb, wb = np.asarray(g).reshape(-1), np.asarray(w).reshape(-1)
        assert gb.dtype.kind == wb.dtype.kind, k
        same = (gb.view(np.uint32) == wb.view(np.uint32)) if gb.dtype.kind == "f" else (gb == wb)
        nan_ok = np.isnan(gb) & np.isnan(wb) if gb.dtype.kind == "f" else False
        assert np.all(same | nan_ok), k
    assert gi == wi


def test_builtin_uniforms_match_oracle(pa):
    from oracle.scene_eval import OracleScene, builtin_uniforms

    for name in SCENES:
        s = pa.Scene.from_file(pa.scene_path(name))
        r = pa.SceneRenderer(s, device=-1)
        r.set_option("render_depth", 40)
        want = builtin_uniforms(OracleScene(pa.scene_path(name)), 3840, 2160, render_depth=40)
        for k, w in want.items():
            g = r.uniform_value(k, 3840, 2160)
            assert g is not None, k
            if np.asarray(g).shape == (4, 4):
                g = np.asarray(g).T.reshape(16)
            assert np.array_equal(np.asarray(g, np.float32).reshape(-1), np.asarray(w, np.float32).reshape(-1)), (name, k)


def test_camera_matrix_known_answer(pa):
    """RotateAroundCam::get_matrix (src/main.rs:286-304): k looks from pos to look_at, i = k x up, j = k x i."""
    s = pa.Scene.from_file(pa.scene_path("basics"))
    r = pa.SceneRenderer(s, device=-1)
    r.set_camera((1.0, 2.0, 3.0), alpha=math.pi / 2, beta=math.pi / 2, r=5.0)
    m = r.uniform_value("_camera", 100, 100)
    assert m[:3, 3] == pytest.approx([1.0, 2.0, 8.0], abs=1e-6)       # pos = look_at + 5 * (0, 0, 1)
    assert m[:3, 2] == pytest.approx([0.0, 0.0, -1.0], abs=1e-6)      # forward
    assert m[:3, 0] == pytest.approx([1.0, 0.0, 0.0], abs=1e-6)       # k x (0,1,0)
    assert m[:3, 1] == pytest.approx([0.0, -1.0, 0.0], abs=1e-6)      # k x i: image y points world-down
    assert r.uniform_value("_camera_scale", 100, 100) == pytest.approx(1.0, abs=1e-7)
    assert r.uniform_value("_view_angle", 100, 100) == np.float32(math.pi / 2)
    assert r.uniform_value("_t_end", 100, 100) == np.float32(210.0)


# ---------------------------------------------------------------------------------------------
# GLSL -> C++ translator
# ---------------------------------------------------------------------------------------------
def test_translate_glsl_rewrites(pa):
    t = pa.translate_glsl
    assert t("float x = 1.;") == "float x = 1.f;"
    assert t("x = 2.5e-3 + 1e5 + 3 + 0x10;") == "x = 2.5e-3f + 1e5f + 3 + 0x10;"
    assert t("vec3 m = 1.0/r.d.xyz;") == "vec3 m = ptl_div(1.0f,r.d.sw<0,1,2>());"
    assert t("step(t1.yzx,t1.xyz)") == "step(t1.sw<1,2,0>(),t1.sw<0,1,2>())"
    assert t("c.rgb *= 2.0;") == "c.swr<0,1,2>() *= 2.0f;"
    assert t("p.xy = q.yx;") == "p.swr<0,1>() = q.sw<1,0>();"
    assert t("hit.t *= len; best.u == r.x") == "hit.t *= len; best.u == r.x"   # single components / struct fields untouched
    assert t("void f(in vec3 a, out float b, inout vec2 c)") == "void f( vec3 a,  float& b,  vec2& c)"
    assert t("float new = delete;") == "float new_ = delete_;"
    assert t("a // 1.0 .xyz\nb") == "a // 1.0 .xyz\nb"                        # comments are not rewritten
    src = "line1\n  x = 1.0; // !FOR_NUMBER!\nline3"
    assert t(src).count("\n") == src.count("\n")


def test_translate_glsl_turns_every_division_into_a_contract_call(pa):
    """Numerics contract 2 (device/ptl_glsl.h) DEFINES a / b (a * (1/b) with the contract's reciprocal) and C++ cannot overload the
    division of two scalars, so the translator rewrites `/` and `/=` into ptl_div / ptl_div_assign calls.  Operand extents follow
    GLSL's grammar: the right operand is one unary expression, the left one the multiplicative chain in front (left-associative)."""
    t = pa.translate_glsl
    cases = {
        "float t = -r.o.z/r.d.z;": "float t = ptl_div(-r.o.z,r.d.z);",
        "x = a * b / c;": "x = ptl_div(a * b , c);",
        "x = a / b / c;": "x = ptl_div(ptl_div(a , b) , c);",
        "x = a / (b + c) * d / e;": "x = ptl_div(ptl_div(a , (b + c)) * d , e);",
        "q = a + b / c - d;": "q = a + ptl_div(b , c) - d;",
        "z = -a / -b;": "z = ptl_div(-a , -b);",
        "w = (a + b) / float(n);": "w = ptl_div((a + b) , float(n));",
        "y = f(a / b, c) / g(d)[2].x;": "y = ptl_div(f(ptl_div(a , b), c) , g(d)[2].x);",
        "k = cond ? a / b : c / d;": "k = cond ? ptl_div(a , b) : ptl_div(c , d);",
        "return normal * dot(d, n) / dot(n, n) * 2.;": "return ptl_div(normal * dot(d, n) , dot(n, n)) * 2.f;",   # src/library.glsl:71
        "int k = i++ / 2;": "int k = ptl_div(i++ , 2);",                                                          # int / int stays an integer division (overload)
        "r.tmul /= len;": "ptl_div_assign(r.tmul , len);",
        "a /= b / c;": "ptl_div_assign(a , ptl_div(b , c));",
        "if (a / b > c) { e /= f; }": "if (ptl_div(a , b) > c) { ptl_div_assign(e , f); }",
        "v.xy /= s;": "v.swr<0,1>() /= s;",                      # a swizzle target keeps the proxy's own operator/= (same arithmetic)
        "x = a /* c */ / b; // a / b": "x = ptl_div(a /* c */ , b); // a / b",
        "p = x.y / z.w / 2.0;\nq /= p * 2.;": "p = ptl_div(ptl_div(x.y , z.w) , 2.0f);\nptl_div_assign(q , p * 2.f);",
    }
    for glsl, want in cases.items():
        assert t(glsl) == want, glsl
    with pytest.raises(pa.PortalError, match="operand of the division"):
        t("x = / 2.;")
    # every snippet of the reference's 82 scenes: after translation no division operator is left outside comments
    # (`/=` survives only behind a swizzle target)
    import glob, re
    from oracle.scene_eval import OracleScene
    seen = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "corpus", "scenes", "*.ron"))):
        if os.path.getsize(path) == 0:
            continue
        sc = OracleScene(path)
        snippets = [c for _, c in sc.library] + [o["code"] for o in sc.objects] + [m["code"] for m in sc.materials if m["kind"] == "Complex"] + [c for _, c in sc.intersection_materials]
        for code in snippets:
            out = re.sub(r"//[^\n]*|/\*.*?\*/", "", t(code), flags=re.S)
            assert not re.search(r"/(?!=)", out), (path, out[max(0, out.find("/") - 60):out.find("/") + 40])
            for m in re.finditer(r"/=", out):
                assert re.search(r"swr<[0-9,]+>\(\)\s*$", out[:m.start()]), (path, out[max(0, m.start() - 60):m.start() + 20])
            seen += code.count("/")
    assert seen > 800


def test_generated_source_line_bookkeeping(pa):
    """Every snippet line of the generated kernel maps back to its scene element (the job of
    LineNumbersByKey, src/code_generation.rs:10-41), and tagged lines are filtered in place."""
    s = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    src = s.generate_source().split("\n")
    owners = {}
    for ln in range(1, len(src) + 1):
        o = s.source_line_owner(ln)
        if o:
            owners.setdefault((o[0], o[1]), []).append((ln, o[2]))
    assert ("intersection_material", "portal_in_portal") in owners and ("library", "portal_advanced") in owners
    lines = owners[("library", "portal_advanced")]
    # consecutive snippets share a line when the earlier one has no trailing newline (same in the
    # reference, whose ranges are start..end+1): the shared line is reported for the first owner
    locs = [loc for _, loc in lines]
    assert locs == list(range(locs[0], locs[0] + len(lines))) and locs[0] in (1, 2)
    first = lines[0][0] - (locs[0] - 1)
    assert "is_inside_portal_advanced" in src[first - 1]
    text = "\n".join(src[ln - 1] for ln, _ in lines)
    assert "!FOR_NUMBER!" not in text and "!FOR_VARIABLE!" in text
    ron_code = re.search(r'name: "portal_advanced",\s*data: \(\("(.*?)"\)\)', open(pa.scene_path("portal_in_portal")).read(), re.S).group(1)
    assert locs[-1] in (ron_code.count("\n"), ron_code.count("\n") + 1)  # line-preserving (last line may be shared too)


def test_material_ids_and_defines(pa):
    """#define NAME_M (USER_MATERIAL_OFFSET + k) in declaration order, then two per portal object."""
    s = pa.Scene.from_file(pa.scene_path("triple_portal"))
    src = s.generate_source()
    defs = re.findall(r"#define (\w+_M) \(USER_MATERIAL_OFFSET \+ (\d+)\)", src)
    assert [int(k) for _, k in defs] == list(range(len(defs)))
    from oracle.scene_eval import OracleScene

    want = OracleScene(pa.scene_path("triple_portal")).material_ids()
    assert {n: 10 + int(k) for n, k in defs} == want


def test_uniform_layout_is_packed_and_static_asserted(pa):
    s = pa.Scene.from_file(pa.scene_path("monoportal"))
    layout, size = s.uniform_layout()
    sizes = {pa.PTL_MAT4: 64, pa.PTL_F32: 4, pa.PTL_I32: 4, pa.PTL_VEC2: 8, pa.PTL_VEC3: 12, pa.PTL_SAMPLER: 16}
    off = 0
    for name, typ, o in layout:
        assert o == off, name
        off += sizes[typ]
    assert layout[0] == ("monoportal_tex", pa.PTL_SAMPLER, 0) and size == (off + 7) // 8 * 8
    names = [n for n, _, _ in layout]
    assert "_camera" in names and "portal_rotate_angle_u" in names and "_external_ray_b" == names[-1]
    assert s.generate_source().count("static_assert(__builtin_offsetof(ptl_uniform_block") == len(layout)


def test_png_roundtrip_and_against_pil(pa, tmp_path):
    from PIL import Image

    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    path = str(tmp_path / "a.png")
    pa.png_write(path, img)
    assert np.array_equal(pa.png_read(path), img)
    assert np.array_equal(np.array(Image.open(path).convert("RGBA")), img)
    tex = os.path.join(ROOT, "scenes", "img", "monoportal.png")
    assert np.array_equal(pa.png_read(tex), np.array(Image.open(tex).convert("RGBA")))


def test_shard_rows_and_deinterleave(pa):
    w, h = 40, 100  # 13 row blocks, the last one has 4 rows
    full = np.arange(h * w * 4, dtype=np.uint32).astype(np.uint8).reshape(h, w, 4)
    out = np.zeros_like(full)
    total = 0
    for phase in range(4):
        f = pa.Frame(w, h, phase, 4)
        blocks = list(range(phase, 13, 4))
        rows = np.concatenate([np.arange(8 * b, min(h, 8 * b + 8)) for b in blocks])
        assert pa.shard_rows(f) == len(rows)
        total += len(rows)
        pa.deinterleave_rows(np.ascontiguousarray(full[rows]), f, out)
    assert total == h and np.array_equal(out, full)
    assert pa.shard_rows(pa.Frame(w, h, 4, 4)) == -1  # phase must be < stride


# ---------------------------------------------------------------------------------------------
# stages and named cameras (`render-frame --stage / --camera`)
# ---------------------------------------------------------------------------------------------
def test_stage_overrides_known_answers(pa):
    """monoportal, stage "doorway to portal" (scenes/monoportal.ron:811-855): `portal_offset` becomes the
    formula progress * const_max_offset + 0.01, `portal_rotate_angle` becomes Progress(0), progress 0."""
    s = pa.Scene.from_file(pa.scene_path("monoportal"))
    assert s.eval_uniform("portal_rotate_angle") == pytest.approx(math.pi)
    assert s.init_stage("doorway to portal") == ""
    assert s.eval_uniform("portal_rotate_angle") == 0.0
    assert s.eval_uniform("portal_offset") == pytest.approx(0.0 * 1.12 + 0.01)
    assert s.eval_uniform("portal_black_color_progress") == 1.0
    with pytest.raises(pa.PortalError):
        s.init_stage("no such stage")


@pytest.mark.parametrize("name", ["monoportal", "portal_in_portal", "triple_portal", "basics", "mobius_monoportal"])
def test_every_stage_and_camera_agrees_with_the_oracle(pa, name):
    from oracle.scene_eval import OracleScene, builtin_uniforms

    path = pa.scene_path(name)
    stages = pa.Scene.from_file(path).stages()
    assert stages
    for stage in stages:
        ps, osc = pa.Scene.from_file(path), OracleScene(path)
        cam_name = ps.init_stage(stage)
        cam_idx = osc.init_stage(stage)
        got, want = ps.uniform_values(), osc.scene_uniform_values()
        for k, w in want.items():
            if k.startswith("oracle_inline"):
                continue
            g = got[k]
            g = np.asarray(g, np.float32).T.reshape(16) if np.asarray(g).shape == (4, 4) else np.asarray(g).reshape(-1)
            wb = np.asarray(w).reshape(-1)
            same = (g.view(np.uint32) == wb.view(np.uint32)) if wb.dtype.kind == "f" else (g == wb)
            assert np.all(same | (np.isnan(g.astype(np.float64)) & np.isnan(wb.astype(np.float64)))), (stage, k)
        assert (cam_name == "") == (cam_idx < 0)
        if cam_idx >= 0:
            r = pa.SceneRenderer(ps, device=-1)
            r.use_camera(cam_name)
            b = builtin_uniforms(osc, 640, 360, camera=osc.camera_settings(cam_idx))
            for k in ("_camera", "_camera_mul_inv", "_camera_scale", "_camera_in_subspace"):
                g = r.uniform_value(k, 640, 360)
                g = np.asarray(g).T.reshape(16) if np.asarray(g).shape == (4, 4) else np.asarray(g).reshape(-1)
                assert np.array_equal(g.astype(np.float32), np.asarray(b[k], np.float32).reshape(-1)), (stage, k)


def test_named_camera_known_answer(pa):
    """portal_in_portal camera "a": look_at = centre of matrix `a` (0,0,-1) + 0.001 (src/gui/camera.rs:96-108)."""
    s = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    assert s.cameras()[:2] == ["a", "b"]
    r = pa.SceneRenderer(s, device=-1)
    r.use_camera("a")
    m = r.uniform_value("_camera", 100, 100)
    alpha, beta, rad = -5.794107164726811, 0.8981089851558188, 3.50000000000001
    pos = np.array([math.sin(beta) * math.cos(alpha), math.cos(beta), math.sin(beta) * math.sin(alpha)]) * rad + np.array([0.001, 0.001, -0.999])
    assert m[:3, 3] == pytest.approx(pos, abs=1e-6)
    with pytest.raises(pa.PortalError):
        r.use_camera("nope")
    r.use_camera("")  # back to the scene's own cam block
    assert np.array_equal(r.uniform_value("_camera", 100, 100), pa.SceneRenderer(s, device=-1).uniform_value("_camera", 100, 100))


def test_average_images_oracle_known_answers():
    """(9) src/main.rs:645-662: L_TO_S[S_TO_L[c]] == c for every byte; mean in linear light, not in bytes."""
    from oracle import postprocess as pp

    assert np.array_equal(pp.L_TO_S[pp.S_TO_L.astype(np.int64)], np.arange(256, dtype=np.uint8))
    a = np.zeros((2, 2, 4), np.uint8)
    b = np.full((2, 2, 4), 255, np.uint8)
    out = pp.average_images([a, b])
    assert out[0, 0].tolist() == [180, 180, 180, 255]          # sqrt(65025 // 2) = 180.3 -> 180 (a byte mean would give 127)
    assert np.array_equal(pp.average_images([b]), b)
    c = np.array([[[10, 20, 30, 7]]], np.uint8)
    assert pp.average_images([c, c, c])[0, 0].tolist() == [10, 20, 30, 255]


def _same_uniforms(got, want, where):
    for k, w in want.items():
        if "oracle_inline" in k:  # unnamed matrices: the two sides number them differently
            continue
        g = got[k]
        g = np.asarray(g, np.float32).T.reshape(16) if np.asarray(g).shape == (4, 4) else np.asarray(g).reshape(-1)
        wb = np.asarray(w).reshape(-1)
        same = (g.view(np.uint32) == wb.view(np.uint32)) if wb.dtype.kind == "f" else (g == wb)
        assert np.all(same | (np.isnan(g.astype(np.float64)) & np.isnan(wb.astype(np.float64)))), (where, k)


@pytest.mark.parametrize("name", ["portal_in_portal", "triple_portal", "basics"])
def test_every_real_animation_frame_state_agrees_with_the_oracle(pa, name):
    """The video pipeline's per-frame step (Scene::init_animation_by_name + SceneRenderer::update, src/gui/scene.rs:1208-1231,
    1353-1493, src/main.rs:1430-1538): for every clip of the scene, at the motion-blur sub-frame times render_animation uses
    (src/main.rs:1787-1800), the product and the oracle agree on time / total_time, on every scene uniform and on the camera."""
    from oracle.portal_oracle import CameraRig, Oracle
    from oracle.scene_eval import builtin_uniforms

    path = pa.scene_path(name)
    ps, o = pa.Scene.from_file(path), Oracle(path)
    clips = ps.animations()
    assert clips and [c[0] for c in clips] == [a["name"] for a in o.scene.animations]
    r = pa.SceneRenderer(ps, device=-1)
    r.set_option("allow_teleport", 0)  # crossing a portal needs the GPU ray query: covered by the -m gpu test
    rig = CameraRig(o)
    rig.allow_teleport = False
    for clip, duration in clips:
        ps.init_animation(clip)
        o.scene.init_animation(clip)
        count, blur, exposure = 3, 2, 0.5
        for i in range(count):
            for j in range(blur):
                t = (i / count + j / blur / count * exposure) * duration
                r.update(t)
                rig.update(t)
                assert (ps.eval_uniform("no such uniform") is None)
                assert r.camera_state()["in_subspace"] == rig.in_subspace
                _same_uniforms(ps.uniform_values(), o.scene.scene_uniform_values(), (clip, t))
                b = builtin_uniforms(o.scene, 320, 180, camera=rig.settings())
                for k in ("_camera", "_camera_mul_inv", "_camera_scale", "_camera_in_subspace"):
                    g = r.uniform_value(k, 320, 180)
                    g = np.asarray(g).T.reshape(16) if np.asarray(g).shape == (4, 4) else np.asarray(g).reshape(-1)
                    assert np.array_equal(g.astype(np.float32), np.asarray(b[k], np.float32).reshape(-1)), (clip, t, k)
    with pytest.raises(pa.PortalError):
        ps.init_animation("no such clip")


def test_real_animation_known_answers(pa):
    """portal_in_portal `intro.2`: 1 s, base stage "How 2", progress := easing_in_out(time), camera = end camera of intro.1
    (use_prev_cam + use_start_cam_as_end); `total_time` continues after intro.1's 3 s (src/gui/scene.rs:1412-1420)."""
    s = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    assert s.animations()[:2] == [("intro.1", 3.0), ("intro.2", 1.0)]
    s.init_animation("intro.2")
    u = s.update(0.25)
    assert (u["time"], u["total_time"]) == (0.25, 3.25)
    assert s.eval_uniform("progress") == pytest.approx((1 - math.cos(0.25 * math.pi)) / 2, abs=1e-15)
    assert u["camera"]["alpha"] == 0.4130013084411631 and u["camera"]["beta"] == 1.1425471496582034 and not u["camera"]["override_matrix"]
    assert s.update(1.0)["camera"]["override_matrix"]      # wrapped around: t_raw < prev_t_raw
    s.init_animation("intro.1")
    u = s.update(1.5)                                         # half way, InOut easing = 0.5
    assert u["time"] == 0.5 and u["camera"]["alpha"] == pytest.approx((-0.8084996795654297 + 0.4130013084411631) / 2, abs=1e-12)
    for kind, t, want in (("Linear", 0.3, 0.3), ("In", 1.0, 1.0), ("Out", 0.0, 0.0), ("InOut", 0.5, 0.5), ("InOutFast", 0.5, 0.5), ("ElasticOut", 1.0, 1.0)):
        from oracle.scene_eval import ease

        assert ease(kind, t) == pytest.approx(want, abs=1e-15)


def test_matrix_lerp_known_answers(pa):
    """Matrix::Lerp (src/gui/matrix.rs:614-627): TRS decomposition, per-part blend (quaternion shortest arc, normalised),
    recomposition.  Half way between the identity and (rotate 90 degrees about z, move 2 along x, scale 3) is
    (rotate 45 degrees, move 1, scale 2); the ends reproduce the operands; product == oracle bit for bit; Matrix::Camera
    follows ptl_scene_set_camera_matrix."""
    from oracle.scene_eval import OracleScene
    from tests import synthetic

    extra = '''
        (name: "far", data: Simple(offset: (2.0, 0.0, 0.0), scale: 3.0, rotate: (0.0, 0.0, 1.5707963267948966), mirror: (false, false, false))),
        (name: "mid", data: Lerp(t: Value(0.5), first: Some(Named("wall")), second: Some(Named("far")))),
        (name: "t0", data: Lerp(t: Value(0.0), first: Some(Named("wall")), second: Some(Named("far")))),
        (name: "t1", data: Lerp(t: Uniform(Some(Named("one"))), first: Some(Named("wall")), second: Some(Named("far")))),
        (name: "cam_follow", data: Mul(to: Some(Inline(Camera)), what: Some(Named("far")))),
    '''
    text = synthetic.wall_scene(extra_matrices=extra).replace('uniforms: ([', 'uniforms: ([ (name: "one", data: Float((min: None, max: None, value: 1.0))),')
    s, o = pa.Scene.from_text(text), OracleScene(text, is_text=True)
    mid = s.eval_matrix("mid")
    c = math.sqrt(0.5)
    want = np.array([[2 * c, -2 * c, 0, 1], [2 * c, 2 * c, 0, 0], [0, 0, 2, 0], [0, 0, 0, 1]])
    assert mid == pytest.approx(want, abs=1e-15)
    assert s.eval_matrix("t0") == pytest.approx(s.eval_matrix("wall"), abs=1e-15)
    assert s.eval_matrix("t1") == pytest.approx(s.eval_matrix("far"), abs=1e-15)
    shift = np.eye(4)
    shift[:3, 3] = (0.5, -1.0, 4.0)
    s.set_camera_matrix(shift)
    o.camera_object_matrix = [list(col) for col in shift.T]
    assert s.eval_matrix("cam_follow") == pytest.approx(s.eval_matrix("far") @ shift, abs=1e-15)   # Mul = what * to (matrix.rs:517-520)
    _same_uniforms(s.uniform_values(), o.scene_uniform_values(), "lerp")


def test_check_command_attributes_compile_errors_to_scene_elements(pa, tmp_path):
    """`portal-amd check` (reference: shader_error_parser + LineNumbersByKey::get_identifier, src/gui/scene.rs:1144-1171):
    hiprtc compiles for gfx950 without a GPU; a diagnostic inside a scene snippet is reported as element + local line."""
    import subprocess

    exe = os.path.join(os.path.dirname(pa.__file__), "portal-amd")
    ok = subprocess.run([exe, "check", pa.scene_path("basics")], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "ok (" in ok.stdout
    broken = tmp_path / "broken.ron"
    broken.write_text(open(pa.scene_path("basics")).read().replace("int is_inside_square(", "int is_inside_square(undeclared_type zz, ", 1))
    bad = subprocess.run([exe, "check", str(broken)], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1
    assert "library `room`, line 1: error: unknown type name 'undeclared_type'" in bad.stdout


def test_average_images_multiply_high_is_the_integer_division():
    """average_images.hip divides the channel sums by the sub-frame count with one multiply-high (magic = floor(2^32/n) + 1);
    every sum the kernel can see (0 .. 65025 n) for every n it accepts (2 .. 256) gives floor(sum / n) -- and 257 is the first
    count for which it does not (why 256 is the limit).  Also the argument that a 1-ulp sqrt cannot change L_TO_S: sqrt(l) stays
    >= 4.9e-4 away from every k + 0.5."""
    def exact(n):
        # x -> (x * magic) >> 32 is monotone, so it equals floor(x / n) on a whole bucket [k n, k n + n - 1] iff it does at both ends:
        # every bucket's two ends for k = 0 .. 65025 (the last bucket's upper end is beyond the largest sum, 65025 n: capped)
        magic = np.uint64(0xFFFFFFFF // n + 1)
        k = np.arange(0, 65026, dtype=np.uint64)
        lo, hi = k * np.uint64(n), np.minimum(k * np.uint64(n) + np.uint64(n - 1), np.uint64(65025 * n))
        return all(np.array_equal((x * magic) >> np.uint64(32), x // np.uint64(n)) for x in (lo, hi))

    for n in range(2, 257):
        assert exact(n), n
    assert not all(exact(n) for n in range(257, 300))
    s = np.sqrt(np.arange(65026, dtype=np.float64))
    assert np.abs((s + 0.5) - np.rint(s + 0.5)).min() > 4.8e-4


def test_clip_constant_specialisation_bakes_only_what_the_clip_keeps_fixed(pa):
    """FLAG_SPECIALIZE_STATIC: portal_in_portal clip `intro.4` pins progress = 1 (a constant: baked) and animates
    progress_2 = easing_in_out(1 - time) (reads `time`: stays a run-time uniform, and so does every matrix built from it);
    builtins (camera ...) and teleport_light_u are never baked."""
    import re

    s = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    s.init_animation("intro.4")
    s.update(0.3)
    src = s.generate_source(pa.FLAG_SPECIALIZE_STATIC)
    define = lambda name: re.search(r"#define %s \((.*)\)\n" % re.escape(name), src).group(1)
    assert define("progress_u") == "0x1p+0f"
    assert define("progress_2_u") == "PTL_U.progress_2_u"
    assert define("a_mat").startswith("mat4(") and define("b0_mat") == "PTL_U.b0_mat" and define("b0_mat_inv") == "PTL_U.b0_mat_inv"
    assert define("_camera") == "PTL_U._camera" and define("teleport_light_u") == "PTL_U.teleport_light_u"
    plain = s.generate_source(0)
    assert "#define progress_u (PTL_U.progress_u)" in plain
    # a still (no clip, nothing reads time): everything is constant, same text as FLAG_SPECIALIZE_ALL
    still = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    assert still.generate_source(pa.FLAG_SPECIALIZE_STATIC) == still.generate_source(pa.FLAG_SPECIALIZE_ALL)


def test_matrix_sqrt_known_answers(pa):
    """Matrix::Sqrt (src/gui/matrix.rs:606-612): M * M == A.  Half of (rotate 90 degrees about z, scale 4, move (3, 0, 0)) is
    (rotate 45 degrees, scale 2, and the translation t with (I + M) t = (3, 0, 0)); product == oracle bit for bit; a
    reflection has no real square root -> "can't be getted"."""
    from oracle.scene_eval import OracleScene
    from tests import synthetic

    extra = '''
        (name: "far", data: Simple(offset: (3.0, 0.0, 0.0), scale: 4.0, rotate: (0.0, 0.0, 1.5707963267948966), mirror: (false, false, false))),
        (name: "half", data: Sqrt(Some(Named("far")))),
        (name: "flip", data: Simple(offset: (0.0, 0.0, 0.0), scale: 1.0, rotate: (0.0, 0.0, 0.0), mirror: (true, false, false))),
        (name: "no_root", data: Sqrt(Some(Named("flip")))),
    '''
    text = synthetic.wall_scene(extra_matrices=extra)
    s, o = pa.Scene.from_text(text), OracleScene(text, is_text=True)
    far, half = s.eval_matrix("far"), s.eval_matrix("half")
    assert half @ half == pytest.approx(far, abs=1e-12)
    c = 2 * math.sqrt(0.5)
    assert half[:3, :3] == pytest.approx(np.array([[c, -c, 0], [c, c, 0], [0, 0, 2]]), abs=1e-12)
    assert s.eval_matrix("no_root") is None
    got, want = s.uniform_values(), o.scene_uniform_values()
    assert "no_root_mat" not in got and "no_root_mat" not in want
    _same_uniforms(got, want, "sqrt")


@pytest.mark.parametrize("name", ["basics", "monoportal", "triple_portal", "portal_in_portal", "mobius_monoportal"])
def test_scene_writer_round_trips_byte_for_byte(pa, name):
    """SURVEY 8f-4, the RON writer (serialize_scene_new_format + ron pretty printer, src/gui/scene_serialized.rs:22-24,654-1100):
    the reference wrote these files with its writer, so writing a freshly loaded scene must give the file back byte for byte --
    layout, float formatting (positional shortest round trip, ties like Rust), raw strings with the right number of hashes."""
    text = open(pa.scene_path(name), encoding="utf-8").read()
    assert pa.ron_format(text) == text
    assert pa.Scene.from_file(pa.scene_path(name)).to_ron() == text


def test_scene_writer_follows_edits(pa):
    """What was changed through the API is what the written file says, nothing else moves: a Float keeps its limits, a formula
    overridden by a value becomes a Float, the stage that was initialised is the file's current_stage; the written file loads
    back and writes out identically."""
    path = pa.scene_path("portal_in_portal")
    original = open(path, encoding="utf-8").read()
    s = pa.Scene.from_file(path)
    s.set_uniform("room_size_x", 5.5)
    written = s.to_ron()
    import difflib

    changed = [l for l in difflib.unified_diff(original.splitlines(), written.splitlines(), lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
    assert changed == ["-                value: 4.0,", "+                value: 5.5,"]
    again = pa.Scene.from_text(written)
    assert again.eval_uniform("room_size_x") == 5.5 and again.to_ron() == written        # a fixed point of load -> write
    s.init_stage("How 2")
    assert '\n    current_stage: Animation("How 2"),\n' in s.to_ron() and '\n    current_stage: Animation("How 2"),\n' not in original
    src = '(a: 0.99658966064453125, b: 1, c: [], d: {}, e: r###"x"#y"###)'
    assert pa.ron_format(src) == '(\n    a: 0.9965896606445313,\n    b: 1,\n    c: [],\n    d: {},\n    e: r##"x"#y"##,\n)'
    with pytest.raises(pa.PortalError):
        pa.ron_format("(a: ")


@pytest.mark.parametrize("name", ["basics", "monoportal", "triple_portal", "portal_in_portal", "mobius_monoportal"])
def test_scene_writer_materialises_stages_and_clips(pa, name):
    """Storage2::set_id / set copy a stage's values INTO the stored elements (src/gui/storage2.rs:195-204, animation.rs:171-183),
    so a scene saved after a stage or clip was initialised carries them.  Same here: for every stage and every clip, the
    written document, loaded again WITHOUT initialising anything, evaluates to the staged scene (named elements bit for bit;
    unnamed matrices are renumbered, so those are compared as a multiset)."""
    path = pa.scene_path(name)
    probe = pa.Scene.from_file(path)
    jobs = [("stage", n) for n in probe.stages()] + [("clip", n) for n, _ in probe.animations()]
    assert jobs
    for kind, n in jobs:
        s = pa.Scene.from_file(path)
        s.init_stage(n) if kind == "stage" else s.init_animation(n)
        s.update(0.3)
        again = pa.Scene.from_text(s.to_ron())
        again.update(0.3)
        if kind == "clip":   # formulas read `time`, and time means "fraction of the current clip" only while one is current
            again.init_animation(n)
            again.update(0.3)
        a, b = s.uniform_values(), again.uniform_values()
        named = lambda d: {k: v for k, v in d.items() if not (k.startswith("id") and k[2:3].isdigit())}
        unnamed = lambda d: sorted(np.asarray(v, np.float32).tobytes() for k, v in d.items() if k.startswith("id") and k[2:3].isdigit())
        assert named(a).keys() == named(b).keys(), (kind, n)
        for k, v in named(a).items():
            assert np.array_equal(np.asarray(v), np.asarray(named(b)[k]), equal_nan=True), (kind, n, k)
        assert unnamed(a) == unnamed(b), (kind, n)


def test_trefoil_text_codec_reference_vector(pa):
    """The reference's unit test `trefoil` (src/gui/uniform.rs:258-264): "1a 2a G,1b 3b B,2a 1a S" -> decode -> encode is the same text."""
    from tests import synthetic

    zeros = ", ".join(["(false, 0, 0)"] * 18)
    text = synthetic.wall_scene().replace('uniforms: ([', f'uniforms: ([ (name: "knot", data: TrefoilSpecial((({zeros})))),')
    s = pa.Scene.from_text(text)
    assert s.get_trefoil("knot") == ""
    s.set_trefoil("knot", "1a 2a G,1b 3b B,2a 1a S")
    assert s.get_trefoil("knot") == "1a 2a G,1b 3b B,2a 1a S"
    assert int(s.uniform_values()["ts_3_knot_u"]) == 15000 and pa.Scene.from_text(s.to_ron()).get_trefoil("knot") == "1a 2a G,1b 3b B,2a 1a S"


def test_cli_fails_loudly_without_a_gpu_and_has_no_cpu_fallback(pa, tmp_path):
    """On a machine without a HIP device the render commands stop with a message and a non-zero status (no silent CPU path); the
    commands that need no device (version, emit-source, check, write) work."""
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    exe = os.path.join(os.path.dirname(pa.__file__), "portal-amd")
    run = lambda *a: subprocess.run([exe, *a], capture_output=True, text=True, timeout=300)
    frame = run("render-frame", pa.scene_path("basics"), "--width", "32", "--height", "32", "--output", str(tmp_path / "f.png"))
    assert frame.returncode != 0 and "no ROCm-capable device" in (frame.stderr + frame.stdout) and not (tmp_path / "f.png").exists()
    clip = run("render", pa.scene_path("basics"), "--width", "32", "--height", "32", "--out-dir", str(tmp_path))
    assert clip.returncode != 0 and not (tmp_path / "anim").exists()
    assert run("version").returncode == 0 and "devices: 0" in run("version").stdout
    src = run("emit-source", pa.scene_path("basics"))
    assert src.returncode == 0 and "ptl_render_kernel" in src.stdout
    assert run("write", pa.scene_path("basics")).stdout == open(pa.scene_path("basics"), encoding="utf-8").read()
    assert run("render-frame", pa.scene_path("basics"), "--stage", "a", "--animation", "b").returncode == 2


_C_CLIENT = r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "portal_amd.h"
/* what a host written in another language does through its FFI, spelled in C99: load a scene, generate the kernel source,
 * compile it for gfx950 (device = -1: no GPU needed) and look at the code object. */
int main(int argc, char** argv) {
    ptl_scene* scene = NULL;
    ptl_renderer* r = NULL;
    char log[4096] = "";
    char* source = NULL;
    ptl_frame frame = {200, 100, 1, 3, 0};
    if (argc < 2 || ptl_scene_load_file(argv[1], &scene) != PTL_OK) { fprintf(stderr, "load: %s\n", ptl_last_error()); return 1; }
    if (ptl_scene_generate_source(scene, 0u, &source) != PTL_OK || !strstr(source, "ptl_render_kernel")) return 2;
    printf("source bytes %zu\n", strlen(source));
    ptl_free(source);
    if (ptl_frame_shard_rows(&frame) != 32) return 3; /* blocks 1, 4, 7, 10 of 13: 4 x 8 rows */
    if (ptl_renderer_create(scene, -1, NULL, 0u, &r, log, sizeof log) != PTL_OK) { fprintf(stderr, "create: %s\n%s\n", ptl_last_error(), log); return 4; }
    if (ptl_renderer_set_option(r, "render_depth", 20.0) != 0 || ptl_renderer_set_option(r, "no_such_option", 1.0) == 0) return 5;
    if (ptl_renderer_draw(r, &frame, NULL, NULL, NULL, NULL, NULL) != PTL_ERR_NO_DEVICE) return 6; /* no CPU fallback */
    printf("%s\n", ptl_version());
    ptl_renderer_destroy(r);
    ptl_scene_free(scene);
    return 0;
}
"""


def test_a_c99_program_links_against_the_abi_and_runs(pa, tmp_path):
    """include/portal_amd.h is a C header (no C++ in it) and libportal_amd.so a C ABI: a C99 client compiled with gcc -pedantic links
    against it and drives the no-GPU part of the path (load, generate, hiprtc-compile for gfx950); rendering without a device is
    refused with PTL_ERR_NO_DEVICE -- there is no CPU fallback behind the ABI either."""
    import subprocess

    import torch

    src = tmp_path / "client.c"
    src.write_text(_C_CLIENT)
    exe = tmp_path / "client"
    libdir = os.path.dirname(pa.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                         "-L", libdir, "-lportal_amd", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    if torch.cuda.is_available():
        pytest.skip("built; the run checks the no-device refusal and this machine has a GPU")
    run = subprocess.run([str(exe), pa.scene_path("monoportal")], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.returncode, run.stderr)
    assert "source bytes" in run.stdout and "portal_amd" in run.stdout


# ---------------------------------------------------------------------------------------------
# round 2: uniform prologue, tolerance mode, cache hygiene, translator diagnostics, layer 3, precompile
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SCENES)
def test_derived_uniforms_cover_every_flat_plane_and_vanish_when_baked(pa, name):
    """Dynamic-uniform source: every plane test of a Flat object goes through plane_intersect_derived with a `ptl_dv_<object>_<side>`
    entry that ptl_tracer::derive fills; with the matrices baked in (FLAG_SPECIALIZE_ALL) or with FLAG_NO_DERIVED_UNIFORMS the
    reference's per-call form is generated.  Derived members sit BEHIND the host-visible layout (uploads never touch them)."""
    scene = pa.Scene.from_file(pa.scene_path(name))
    src = scene.generate_source(0)
    body = src[src.index("PTL_FN SceneIntersection scene_intersect(const Ray& r"):src.index("// Prologue (ptl_derive_kernel")]
    tests_in_body = body.count("plane_intersect_derived(")
    assert tests_in_body > 0 and "plane_intersect(" not in body.replace("plane_intersect_derived(", "")
    derive = src[src.index("PTL_FN void derive(ptl_uniform_block* out)"):src.index("// Material id -> what happens to the path.")]
    assert derive.count("_nrm = unit;") == tests_in_body
    layout, size = scene.uniform_layout()
    block = src[src.index("struct ptl_uniform_block {"):src.index("};", src.index("struct ptl_uniform_block {"))]
    members = [l.split()[-1].rstrip(";") for l in block.splitlines()[1:] if l.strip()]
    first_derived = next(i for i, m in enumerate(members) if m.startswith("ptl_dv_"))
    assert all(m.startswith(("ptl_dv_", "ptl_dvo_", "ptl_hv")) for m in members[first_derived:]) and first_derived == len(layout)  # (ptl_hv*: glsl_hoist.h; ptl_dvo_*: first-trip plane tests)
    assert "hit = plane_intersect_derived(r" not in scene.generate_source(pa.FLAG_NO_DERIVED_UNIFORMS)
    assert "hit = plane_intersect_derived(r" not in scene.generate_source(pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)


def test_fast_math_is_a_flag_not_a_default(pa, tmp_path, monkeypatch):
    """FLAG_FAST_MATH only adds the PTL_FAST_MATH define (and with it other compile options): the generated source is the exact
    kernel's, the code object differs, and the default build is the exact one."""
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))
    scene = pa.Scene.from_file(pa.scene_path("monoportal"))
    assert scene.generate_source(pa.FLAG_FAST_MATH) == scene.generate_source(0)
    exact = pa.SceneRenderer(scene, device=-1).code_object()
    fast = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_FAST_MATH).code_object()
    assert exact[:4] == fast[:4] == b"\x7fELF" and exact != fast and len(fast) < len(exact)  # no division / sqrt expansions
    assert len(os.listdir(tmp_path)) == 2


def test_quick_jit_is_a_build_option_for_the_unbaked_kernel_only(pa, tmp_path, monkeypatch):
    """FLAG_QUICK_JIT (bit 18): same generated source, its own cache entry (-O1 instead of -O3), and no effect on a clip-constant
    build (bit 3), which is asked for because many frames will run on it."""
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("PTL_JIT_OPT", raising=False)  # conftest pins -O1 for this suite; here the two levels are the point
    scene = pa.Scene.from_file(pa.scene_path("monoportal"))
    assert scene.generate_source(pa.FLAG_QUICK_JIT) == scene.generate_source(0)
    baked = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL  # with baked loop bounds the quick build also leaves the loops rolled
    assert scene.generate_source(baked | pa.FLAG_QUICK_JIT) == scene.generate_source(baked | pa.FLAG_NO_UNROLL)
    full = pa.SceneRenderer(scene, device=-1).code_object()
    quick = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_QUICK_JIT).code_object()
    assert full[:4] == quick[:4] == b"\x7fELF" and full != quick
    assert len(os.listdir(tmp_path)) == 2
    static = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_SPECIALIZE_STATIC).code_object()
    assert pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_SPECIALIZE_STATIC | pa.FLAG_QUICK_JIT).code_object() == static
    assert len(os.listdir(tmp_path)) == 3
    r = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_QUICK_JIT)  # `render`: starts on the quick kernel, switches per clip
    assert r.code_object() == quick
    r.set_option("specialize_static", 1)
    assert r.code_object() == static


def test_specialised_builds_compile_the_renderers_mode_switches_in(pa, tmp_path, monkeypatch):
    """With any specialisation on, the camera models / output modes a frame does not use are not in the kernel at all
    (KernelOptions::baked_options; 7-10 % of the BASELINE scenes' kernel time, profiles/r03/stub_bake_switches.jsonl): the specialised
    build is a different code object from the one a switch flips to, the flip rebuilds (counted), flipping back finds the first one in
    the cache; the un-specialised kernel reads the switches at run time and never rebuilds."""
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))
    scene = pa.Scene.from_file(pa.scene_path("monoportal"))
    baked = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)
    pinhole = baked.code_object()
    assert len(os.listdir(tmp_path)) == 1 and baked.rejit_count() == 0
    baked.set_option("use_360_camera", 1)
    sphere = baked.code_object()
    assert sphere != pinhole and baked.rejit_count() == 1 and len(os.listdir(tmp_path)) == 2
    assert len(sphere) != len(pinhole)  # another camera model's code, not a flipped literal
    baked.set_option("render_depth", 7)   # not a mode switch: stays a run-time uniform
    assert baked.code_object() == sphere and baked.rejit_count() == 1
    baked.set_option("use_360_camera", 0)
    assert baked.code_object() == pinhole and baked.rejit_count() == 2 and len(os.listdir(tmp_path)) == 2
    dynamic = pa.SceneRenderer(scene, device=-1)
    before = dynamic.code_object()
    dynamic.set_option("use_360_camera", 1)
    dynamic.set_option("draw_depth_map", 1)
    assert dynamic.code_object() == before and dynamic.rejit_count() == 0
    static = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_SPECIALIZE_STATIC)
    flat = static.code_object()
    static.set_option("draw_depth_map", 1)
    assert static.code_object() != flat and static.rejit_count() == 1


def test_zero_patterns_of_runtime_matrices_shorten_their_products(pa):
    """KernelOptions::mask_zero_elements: with any specialisation on, a matrix that stays a run-time uniform has its zero pattern compiled
    in -- `transform(X_mat, ..)` of the snippets and the generated plane tests become the masked forms -- and the host build of that
    source draws the same bits as the full products.  Not for the un-specialised kernel (valid for every state), contract 1 or --fast."""
    import re

    from oracle import host_build as hb

    ints = pa.FLAG_SPECIALIZE_INTS
    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    src = scene.generate_source(ints)
    masks = dict(re.findall(r"#define PTL_MASK_(\w+) (0x[0-9a-f]{4})u", src))
    assert len(masks) > 20 and all(int(v, 16) != 0xFFFF for v in masks.values())
    assert "ptl_transform_m<PTL_MASK_b0_mat_inv>(b0_mat_inv," in src and "transform(b0_mat_inv," not in src.replace("ptl_transform_m<PTL_MASK_b0_mat_inv>(b0_mat_inv,", "")
    assert re.search(r"plane_intersect_derived(_o)?<PTL_MASK_\w+>\(r, ", src) and re.search(r"ptl_plane_cull(_o)?<PTL_MASK_\w+>\(r, ", src)
    for flags in (0, ints | pa.FLAG_NO_ZERO_MASKS, ints | pa.FLAG_EXACT_CR, ints | pa.FLAG_FAST_MATH):
        assert "#define PTL_MASK_" not in scene.generate_source(flags), flags
    assert "#define PTL_MASK_" not in scene.generate_source(ints | pa.FLAG_SPECIALIZE_ALL)  # every matrix is a literal there: ptl_mterm does it
    # the identity pattern of a pure translation: only the diagonal and the last column may be non-zero
    ident_like = [v for v in masks.values() if int(v, 16) & ~0xF421 == 0]
    assert ident_like, masks
    frames = {}
    for label, flags in (("masked", ints), ("full", ints | pa.FLAG_NO_ZERO_MASKS)):
        sc = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
        r = pa.SceneRenderer(sc, device=-1, flags=flags)
        r.set_option("render_depth", 8)
        frames[label] = hb.host_kernel_for(r, sc, 64, 36, flags=flags).render(64, 36)["rgba32f"].copy()
    assert np.array_equal(frames["masked"].view(np.uint32), frames["full"].view(np.uint32))
    assert len(np.unique(frames["masked"].reshape(-1, 4), axis=0)) > 100


_PRODUCT_SHAPES = """vec4 p = r.o;
vec4 a = portal_a_mat_inv * p;
vec4 b = portal_a_mat * (portal_b_mat_inv * vec4(p.xyz, 1.));
vec4 c = portal_b_mat * p.xyzw;
vec4 d = p.x * portal_a_mat * p;
vec4 e = portal_a_mat * -p;
mat4 m = portal_a_mat * portal_b_mat_inv;
vec4 f = (portal_a_mat_inv * transform(portal_b_mat, r).o) * 2.;
float s = dot(a + b + c + d + e + f, vec4(1.)) + (m * p).x;
SceneIntersectionWithMaterial result = SceneIntersectionWithMaterial(scene_intersection_none, material_empty());
if (s > 1e30) result.scene = SceneIntersection(CUSTOM_MATERIAL, SurfaceIntersection(true, 1., 0., 0., vec3(0., 0., 1.)), false);
return result;"""


def test_masked_product_rewrite_follows_the_shape_of_the_operand(pa):
    """`X_mat * <operand>` of a snippet becomes `ptl_mul_m<PTL_MASK_X_mat>(X_mat, <operand>)` for the operand shapes GLSL has -- a name, a
    parenthesised group, a constructor call, a swizzle (already `.sw<..>()` by then), a call with a member -- and stays as written where the
    matrix is not the whole left operand (`p.x * M * p`), where the right operand starts with a sign, or where it is another matrix (the
    overload set of ptl_mul_m takes that one); the source compiles for gfx950 and the host build draws the bits of the full products."""
    from oracle import host_build as hb

    ints = pa.FLAG_SPECIALIZE_INTS
    text = open(pa.scene_path("basics")).read()
    text = text.replace("intersection_materials: ([]),", 'intersection_materials: ([\n        (\n            name: "shapes",\n            data: ((("' + _PRODUCT_SHAPES + '"))),\n        ),\n    ]),')
    src = pa.Scene.from_text(text).generate_source(ints | pa.FLAG_NO_FIRST_TRIP)
    body = src[src.index("intersect_material_0(Ray r, float ptl_far) {"):]
    body = body[:body.index("return result;")]
    for want in ("vec4 a = ptl_mul_m<PTL_MASK_portal_a_mat_inv>(portal_a_mat_inv, p);",
                 "vec4 b = ptl_mul_m<PTL_MASK_portal_a_mat>(portal_a_mat, (ptl_mul_m<PTL_MASK_portal_b_mat_inv>(portal_b_mat_inv, vec4(p.sw<0,1,2>(), 1.f))));",
                 "vec4 c = ptl_mul_m<PTL_MASK_portal_b_mat>(portal_b_mat, p.sw<0,1,2,3>());",
                 "vec4 d = p.x * portal_a_mat * p;",
                 "vec4 e = portal_a_mat * -p;",
                 "mat4 m = PTL_U.ptl_hv0;",   # uniform-only: hoisted into the prologue, where the same rewrite meets it (below)
                 "(ptl_mul_m<PTL_MASK_portal_a_mat_inv>(portal_a_mat_inv, ptl_transform_m<PTL_MASK_portal_b_mat>(portal_b_mat, r).o))"):
        assert want in body, (want, body)
    assert "ptl_mul_m<PTL_MASK_portal_a_mat>(portal_a_mat, portal_b_mat_inv)" in src  # matrix x matrix: the overload for "anything else"
    frames = {}
    for label, flags in (("masked", ints), ("full", ints | pa.FLAG_NO_ZERO_MASKS)):
        sc = pa.Scene.from_text(text)
        r = pa.SceneRenderer(sc, device=-1, flags=flags)  # hiprtc for gfx950
        assert r.code_object()[:4] == b"\x7fELF"
        r.set_option("render_depth", 6)
        frames[label] = hb.host_kernel_for(r, sc, 48, 27, flags=flags).render(48, 27)["rgba32f"].copy()
    assert np.array_equal(frames["masked"].view(np.uint32), frames["full"].view(np.uint32))


def test_code_object_cache_rejects_foreign_files_and_names_the_toolchain(pa, tmp_path, monkeypatch):
    """The cache key covers source + options + the hiprtc library that compiled it; a truncated or non-ELF file under that name is
    ignored and replaced by a fresh build."""
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))
    scene = pa.Scene.from_file(pa.scene_path("basics"))
    good = pa.SceneRenderer(scene, device=-1).code_object()
    (name,) = os.listdir(tmp_path)
    assert re.fullmatch(r"ptl_[0-9a-f]{16}\.hsaco", name)
    path = tmp_path / name
    path.write_bytes(b"not a code object")
    again = pa.SceneRenderer(scene, device=-1).code_object()
    assert again[:4] == b"\x7fELF" and len(again) == len(good) and path.read_bytes()[:4] == b"\x7fELF"
    assert re.search(r"hiprtc_version=\d+\.\d+", pa.version())


def test_translator_refuses_struct_fields_that_spell_swizzles(pa):
    assert "a.uv" in pa.translate_glsl("struct A { vec2 uv; float t; }; float f(A a) { return a.uv.x + a.t; }")
    for field in ("st", "xy", "rgb", "pq"):
        with pytest.raises(pa.PortalError, match="spells a vector swizzle"):
            pa.translate_glsl("struct A { vec3 %s; }; float f(A a) { return a.%s.x; }" % (field, field))
    assert "sw<0,1>()" in pa.translate_glsl("vec2 f(vec4 v) { return v.xy; }")  # ordinary swizzles are untouched


def test_frame_group_and_precompile_without_a_gpu(pa, tmp_path):
    """Layer 3 needs devices: on a box without one it fails with a message, not a crash; `portal-amd precompile` needs none."""
    import subprocess

    import torch

    if not torch.cuda.is_available():
        with pytest.raises(pa.PortalError):
            pa.FrameGroup(pa.Scene.from_file(pa.scene_path("basics")), [0, 0])
    exe = os.path.join(ROOT, "portal_amd", "portal-amd")
    env = dict(os.environ, PTL_CACHE_DIR=str(tmp_path))
    done = subprocess.run([exe, "precompile", os.path.join(ROOT, "scenes", "basics.ron"), "--specialize", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert done.returncode == 0, done.stderr
    assert done.stdout.count("code object in") == 2 and len(os.listdir(tmp_path)) == 2


def test_translator_defers_loop_carried_ray_transforms_only_when_it_is_safe(pa):
    """glsl_translate.h `defer_loop_updates`: `X = transform(A_mat, transform(B_mat_inv, X));` at the top level of a `for` body, X read
    only in nested blocks -> counter + flush before every reading statement (+ behind the loop when X is read there); every
    violated condition leaves the text alone.  Line count is preserved either way."""
    base = """Ray ra = r;
Ray rb = r;
for (int k = 0; k < n_u; k++) {
    ra = transform(b0_mat, transform(a_mat_inv, ra));
    rb = transform(a_mat, rb);
    if (k > 2) {
        if (hit.t > 0.) { out1 = f(ra, hit.t); }
        out2 = g(rb);
    }
}
"""
    got = pa.translate_glsl(base)
    assert got.count("\n") == base.count("\n")
    assert "int ptl_pend_0 = 0; int ptl_pend_1 = 0; for (int k = 0;" in got
    assert "++ptl_pend_0;" in got and "++ptl_pend_1;" in got
    assert "{ for (; ptl_pend_0 > 0; --ptl_pend_0) ra=transform(b0_mat,transform(a_mat_inv,ra)); out1 = f(ra, hit.t); }" in got
    assert "for (; ptl_pend_1 > 0; --ptl_pend_1) rb=transform(a_mat,rb); out2 = g(rb);" in got
    assert got.count("ra=transform(") == 1 and "ra = transform(" not in got          # the eager update is gone, one flush site
    # read behind the loop: one more flush there
    after = pa.translate_glsl(base + "result = h(ra);\n")
    assert after.count("ra=transform(") == 2 and "}\nfor (; ptl_pend_0 > 0; --ptl_pend_0) ra=transform(b0_mat,transform(a_mat_inv,ra)); result = h(ra);" in after
    unchanged = [
        base.replace("out2 = g(rb);", "out2 = g(rb); rb = r;"),                                    # X assigned elsewhere in the loop
        base.replace("    if (k > 2) {", "    out0 = g(rb);\n    if (k > 2) {"),                     # X read at the top level of the body
        base.replace("rb = transform(a_mat, rb);", "rb = transform(m_local, rb);"),                 # not a uniform name
        base.replace("Ray rb = r;", "Ray rb = r; mat4 a_mat = mat4(1.);"),                          # a local that only looks like a uniform
        base.replace("rb = transform(a_mat, rb);", "rb = normalize_ray(transform(a_mat, rb));"),    # another function in the chain
        "for (int j = 0; j < 3; j++) {\n" + base + "}\n",                                          # loop nested in a loop
        base.replace("for (int k = 0; k < n_u; k++) {", "for (int k = 0; k < int(rb.tmul); k++) {"),  # X in the loop header
        base.replace("Ray rb = r;\n", "\n"),                                                        # X is not a local declared before the loop (a parameter, a global)
        "do {\n" + base + "} while (again);\n",                                                    # inside a do-while: left alone altogether
    ]
    for text in unchanged:
        out = pa.translate_glsl(text)
        update = next(line.strip() for line in text.splitlines() if line.strip().startswith("rb = ") and "transform(" in line)
        assert update.replace("1.)", "1.f)") in out and "ptl_pend_1" not in out and out.count("\n") == text.count("\n"), text
    # the shipped scenes: the headline's snippet has two such chains (scenes/portal_in_portal.ron:1146-1147), the others none
    counts = {name: pa.Scene.from_file(pa.scene_path(name)).generate_source(pa.FLAG_NO_FIRST_TRIP).count("int ptl_pend_") for name in SCENES}
    assert counts == {"basics": 0, "monoportal": 0, "triple_portal": 0, "portal_in_portal": 2, "mobius_monoportal": 0}
    # (with the first-trip forms -- opt-in since round 6, FLAG_KEEP_TRANSFORM_DODGES -- the headline's snippet is compiled twice: the copy for the first
    # trip defers the same two chains)
    assert pa.Scene.from_file(pa.scene_path("portal_in_portal")).generate_source(0).count("int ptl_pend_") == 2
    assert pa.Scene.from_file(pa.scene_path("portal_in_portal")).generate_source(pa.FLAG_KEEP_TRANSFORM_DODGES).count("int ptl_pend_") == 4
    assert "ptl_pend_" not in pa.Scene.from_file(pa.scene_path("portal_in_portal")).generate_source(pa.FLAG_NO_DEFERRED_UPDATES)


# ---------------------------------------------------------------------------------------------
# uniform-work hoisting of scene snippets (portal_amd/csrc/host/glsl_hoist.h)
# ---------------------------------------------------------------------------------------------
_HOIST_UNIFORMS = {"a_mat": "mat4", "a_mat_inv": "mat4", "b_mat": "mat4", "b_mat_inv": "mat4", "scale_u": "float", "n_u": "int", "dir_u": "vec3"}


def test_hoister_moves_maximal_uniform_expressions_and_keeps_the_lines(pa):
    code = "vec4 pos = r.o;\npos = b_mat * a_mat_inv * pos; // comment\nfloat k = length(dir_u) * scale_u + r.d.x;\nreturn pos * k;"
    out, prologue = pa.hoist_glsl(code, _HOIST_UNIFORMS, params=["r"])
    assert out.count("\n") == code.count("\n")
    assert "pos = PTL_U.ptl_hv0 * pos;" in out and "float k = PTL_U.ptl_hv1 + r.d.x;" in out
    assert "// member: mat4 ptl_hv0" in prologue and "// member: float ptl_hv1" in prologue
    assert "PTL_DV_OUT.ptl_hv0 = b_mat * a_mat_inv;" in prologue and "PTL_DV_OUT.ptl_hv1 = length(dir_u) * scale_u;" in prologue
    assert pa.hoist_glsl(code, _HOIST_UNIFORMS, params=["r"]) == (out, prologue)  # deterministic: the source text keys the code-object cache


@pytest.mark.parametrize("code,why", [
    ("float k = scale_u * 2.0;\nreturn r.o * k;", "one multiplication costs less than the load that would replace it"),
    ("float k = 1.0 + 2.0 * 3.0;\nreturn r.o * k;", "literals only: the compiler folds it"),
    ("mat4 a_mat = mat4(1.0);\nvec4 p = a_mat * (b_mat * vec4(1.0));\nreturn p + r.o;".replace("(b_mat * vec4(1.0))", "r.o"), "a local shadows the uniform's name"),
    ("vec3 n = normalize(dir_u);\nn = -n;\nreturn vec4(n * r.d.x, 0.0);", None),  # (n written twice: only its initialiser is uniform)
    ("vec3 n = normalize(dir_u) * r.d.x;\nreturn vec4(n, 0.0);", None),
    ("int q = n_u / 2;\nreturn r.o * float(q);", "an integer quotient may trap where the snippet guards it"),
    ("#define K 2.0\nfloat k = length(dir_u) * K;\nreturn r.o * k;", "macros: the text is not what the compiler sees"),
    ("float k = length(dir_u) * ;\nreturn r.o;", "unparsable"),
    ("do { x = 1; } while (false);\nfloat k = length(dir_u) * scale_u;\nreturn r.o * k;", "do-while"),
])
def test_hoister_leaves_alone_what_it_must(pa, code, why):
    out, prologue = pa.hoist_glsl(code, _HOIST_UNIFORMS, params=["r"])
    if why is None:  # the uniform PART is still moved; the varying rest stays
        assert "PTL_U.ptl_hv0" in out and "normalize(dir_u)" in prologue and "r.d.x" in out
    else:
        assert out == code and "member" not in prologue, why


def test_hoister_respects_out_parameters(pa):
    code = "vec3 n = normalize(dir_u);\nturn(n);\nreturn vec4(n * length(dir_u), 0.0);"
    out, _ = pa.hoist_glsl(code, _HOIST_UNIFORMS, out_functions=["turn"], params=["r"])
    assert "vec3 n = PTL_U.ptl_hv0;" in out and "n * PTL_U.ptl_hv1" in out  # normalize(dir_u), length(dir_u): still uniform expressions ...
    assert "PTL_U.ptl_hv2" not in out                                        # ... but n itself is not a uniform local any more
    plain, _ = pa.hoist_glsl(code, _HOIST_UNIFORMS, params=["r"])
    assert "PTL_U.ptl_hv1" in plain and "n * length(dir_u)" not in plain     # without the out parameter the product is one uniform value


_CHAIN = """vec4 nb = b_mat * vec4(0., 0., 1., 0.);
vec4 acc = r.o;
for (int i = 0; i < n_u; i++) {
	vec3 unit = normalize_normal(nb.xyz, r.d.xyz);
	acc += vec4(unit, 0.) * float(i);
%s	nb = b_mat * (a_mat_inv * nb);
	if (is_collinear(unit, nb.xyz)) { acc.w += 1.; }
}
return acc;"""


def test_hoister_tabulates_loop_carried_uniform_chains(pa):
    out, prologue = pa.hoist_glsl(_CHAIN % "", _HOIST_UNIFORMS, params=["r"])
    assert out.count("\n") == (_CHAIN % "").count("\n")
    assert "const bool ptl_tab_ok_" in out and "(n_u) <= 64" in out
    # the chain itself: a table read under the guard, the original update otherwise
    assert "nb = PTL_U.ptl_hv" in out and "[i + 1]; else nb = b_mat * (a_mat_inv * nb);" in out
    # normalize(nb.xyz) BEFORE the update reads entry i, length(nb.xyz) AFTER it entry i + 1; both keep their fallback
    assert "ptl_normalize_normal_unit((ptl_tab_ok_" in out and "[i] : (normalize(nb.xyz)))" in out
    assert "ptl_is_collinear_len(unit, nb.xyz, (ptl_tab_ok_" in out and "[i + 1] : (length(nb.xyz)))" in out
    assert "for (int ptl_k = 0; ptl_k < 66; ptl_k++) {" in prologue and prologue.rstrip().endswith("}")
    body = prologue[prologue.index("for (int ptl_k"):]
    assert body.index("[ptl_k] = nb;") < body.index("nb = b_mat * (a_mat_inv * nb);")  # tables first, the update last
    assert body.count("[ptl_k] = ") == 3


@pytest.mark.parametrize("edit,why", [
    (("for (int i = 0; i < n_u; i++)", "for (int i = 0; i < n_u; i += 1)"), "not the canonical loop header"),
    (("%s", "	if (acc.x > 1.) continue;\n"), "a continue can skip the update"),
    (("return acc;", "return acc + nb;"), "the chain's value is read behind the loop"),
    (("	nb = b_mat * (a_mat_inv * nb);", "	nb = b_mat * (a_mat_inv * nb) + r.d;"), "the update is not uniform"),
    (("	nb = b_mat * (a_mat_inv * nb);", "	if (acc.x > 0.) { nb = b_mat * (a_mat_inv * nb); }"), "the update is conditional"),
    (("	nb = b_mat * (a_mat_inv * nb);", "	nb = b_mat * (a_mat_inv * nb) * float(i);"), "the update depends on the trip number"),
])
def test_hoister_does_not_tabulate_what_is_not_a_function_of_the_trip_number(pa, edit, why):
    code = _CHAIN.replace(edit[0], edit[1]) if edit[0] != "%s" else _CHAIN % edit[1]
    code = code % "" if "%s" in code else code
    out, prologue = pa.hoist_glsl(code, _HOIST_UNIFORMS, params=["r"])
    assert "ptl_tab_ok_" not in out and "ptl_k" not in prologue, why


def test_hoisted_scene_source_is_selfconsistent(pa):
    """Every member the snippets read exists in the block (behind the uploaded part) and is written by derive(); with the scene
    uniforms baked in, or with FLAG_NO_UNIFORM_HOIST / FLAG_NO_DERIVED_UNIFORMS, nothing is hoisted."""
    import re
    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    src = scene.generate_source(0)
    block = src[src.index("struct ptl_uniform_block {"):src.index("};", src.index("struct ptl_uniform_block {"))]
    members = set(re.findall(r"\b(ptl_hv\d+)(?:\[\d+\])?;", block))
    assert len(members) >= 7 and "vec4 ptl_hv" in block and "[66];" in block
    derive = src[src.index("PTL_FN void derive(ptl_uniform_block* out)"):src.index("// Material id -> what happens to the path.")]
    assert set(re.findall(r"PTL_DV_OUT\.(ptl_hv\d+)", derive)) == members
    assert set(re.findall(r"PTL_U\.(ptl_hv\d+)", src)) == members
    baked = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    for flags in (pa.FLAG_NO_UNIFORM_HOIST, pa.FLAG_NO_DERIVED_UNIFORMS, baked | pa.FLAG_NO_FIRST_TRIP):
        assert "ptl_hv" not in scene.generate_source(flags)
    # round 5: a baked build of this scene has affine rays, its transforms are a few additions, and it gets no first-trip copy at all
    src = scene.generate_source(baked)
    assert "ptl_hv" not in src and "intersect_material_0_first(" not in src and "#define PTL_FIRST_TRIP_SNIPPETS" not in src and "ptl_pend_" not in src
    # with everything baked only the camera is left: the first-trip copy of the snippet (round 4's shape: FLAG_KEEP_TRANSFORM_DODGES; also
    # what a scene without affine rays gets) reads two tables of ray origins, nothing else
    src = scene.generate_source(baked | pa.FLAG_KEEP_TRANSFORM_DODGES)
    assert src.count("ptl_pend_") > 4
    block = src[src.index("struct ptl_uniform_block {"):src.index("};", src.index("struct ptl_uniform_block {"))]
    assert re.findall(r"(\w+) ptl_hv\d+(\[\d+\])?;", block) == [("vec4", "[66]"), ("vec4", "[66]")]
    assert src.count("ptl_ray_o(") == 3 and "intersect_material_0_first(Ray r, float ptl_far) {" in src
    # (the generator says in the source itself which first-trip forms the kernel has; with the matrices baked: the snippets', not the planes')
    assert "#define PTL_FIRST_TRIP_SNIPPETS 1" in src and "#define PTL_FIRST_TRIP_PLANES" not in src


@pytest.mark.parametrize("scene_file,moves", [
    ("scenes/portal_in_portal.ron", [("progress", 0.37), ("show_teleported", 70)]),   # 70 copies: past the tables, the guarded fallback runs
    ("tests/corpus/scenes/matryoshka.ron", []),
    ("tests/corpus/scenes/recursive_space.ron", []),
    ("tests/corpus/scenes/trefoil.ron", []),
])
def test_hoisted_and_plain_host_builds_draw_the_same_bits(pa, scene_file, moves):
    """The checker's host build of the generated source runs derive() too: frames with the uniform work hoisted and frames of
    the source as the reference wrote it must agree bit for bit -- also after uniforms have moved, also where a loop runs longer
    than its tables."""
    from oracle import host_build as hb
    path = os.path.join(ROOT, scene_file)
    root = os.path.join(ROOT, "tests", "corpus") if "corpus" in scene_file else None
    frames = {}
    for label, flags in (("hoisted", 0), ("plain", pa.FLAG_NO_UNIFORM_HOIST)):
        scene = pa.Scene.from_file(path)
        r = pa.SceneRenderer(scene, device=-1, flags=flags, **({"asset_root": root} if root else {}))
        r.set_option("render_depth", 12)
        got = [hb.host_kernel_for(r, scene, 48, 27, flags=flags, **({"asset_root": root} if root else {})).render(48, 27)["rgba32f"].copy()]
        for name, value in moves:
            assert scene.set_uniform(name, value)
            got.append(hb.host_kernel_for(r, scene, 48, 27, flags=flags, **({"asset_root": root} if root else {})).render(48, 27)["rgba32f"].copy())
        frames[label] = got
        if label == "hoisted":
            assert "ptl_hv" in scene.generate_source(flags)
    for a, b in zip(frames["hoisted"], frames["plain"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_hoister_does_not_take_a_scene_function_for_a_builtin(pa):
    """A scene may define `sqr`, `normalize_normal` ... itself (any signature): such calls are neither typed like the built-in nor
    re-targeted to a staged form."""
    code = "float k = sqr(scale_u + 1.0) * length(dir_u);\nvec3 n = normalize_normal(dir_u, r.d.xyz);\nreturn vec4(n * k, 0.0);"
    plain, _ = pa.hoist_glsl(code, _HOIST_UNIFORMS, params=["r"])
    assert "float k = PTL_U.ptl_hv0;" in plain and "ptl_normalize_normal_unit(PTL_U.ptl_hv1, r.d.xyz)" in plain
    own, prologue = pa.hoist_glsl(code, _HOIST_UNIFORMS, out_functions=["=sqr", "=normalize_normal"], params=["r"])
    assert "sqr(scale_u + 1.0) * PTL_U.ptl_hv0" in own and "length(dir_u)" in prologue     # only the part that IS a built-in moves
    assert "normalize_normal(dir_u, r.d.xyz)" in own and "ptl_normalize_normal_unit" not in own
    src = pa.Scene.from_file(pa.scene_path("portal_in_portal")).generate_source(0)
    assert "ptl_normalize_normal_unit(" in src and "ptl_is_collinear_len(" in src        # the reference's scenes define neither


_SPHERE_SNIPPET = """vec3 rel = r.o.xyz - vec3(0.3, 0.2, 0.1) * _t_start;
Ray q = r;
float c = dot(rel, rel) - 0.25 + q.o.x * 0.0;
float b = dot(rel, r.d.xyz);
float h = b * b - c;
SceneIntersectionWithMaterial result = SceneIntersectionWithMaterial(scene_intersection_none, material_empty());
if (h > 0.) {
  float t = -b - sqrt(h);
  if (t > 0.) {
    result.scene = SceneIntersection(CUSTOM_MATERIAL, SurfaceIntersection(true, t, 0., 0., normalize(rel + r.d.xyz * t)), false);
    result.material = material_final(color(0.9, 0.4, 0.1));
  }
}
return result;"""


def test_first_trip_copy_declares_the_rays_a_plain_uniform_expression_reads(pa):
    """in the first-trip copy of an intersection-material snippet `r.o` is the prologue's camera origin, so
    `r.o.xyz - vec3(..) * _t_start` (the head of a canonical sphere test) and `q.o` of a copy `Ray q = r;` are uniform-only and
    get hoisted -- the prologue must then declare `r` / `q` even though no ray EXPRESSION was rewritten.  The scene has to compile
    at the default flags and draw the same bits as the general copy."""
    from oracle import host_build as hb

    text = open(pa.scene_path("basics")).read()
    assert "intersection_materials: ([])," in text
    text = text.replace("intersection_materials: ([]),", 'intersection_materials: ([\n        (\n            name: "ball",\n            data: ((("' + _SPHERE_SNIPPET + '"))),\n        ),\n    ]),')
    frames = {}
    keep = pa.FLAG_KEEP_TRANSFORM_DODGES  # (round 6: the first-trip forms are opt-in; the default is the general copy alone)
    for label, flags in (("first", keep), ("general", 0), ("general_by_flag", keep | pa.FLAG_NO_FIRST_TRIP), ("first_baked", keep | pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)):
        scene = pa.Scene.from_text(text)
        src = scene.generate_source(flags)
        assert ("intersect_material_0_first(Ray r, float ptl_far) {" in src) == label.startswith("first")
        if label == "first":
            assert "intersect_material_0_first(Ray r, float ptl_far) {" in src
            derive = src[src.index("PTL_FN void derive(ptl_uniform_block* out)"):src.index("// Material id -> what happens to the path.")]
            assert "Ray r = Ray(PTL_DV_OUT.ptl_dv_origin" in derive  # the dummy parameter the hoisted `r.o...` expression names
        r = pa.SceneRenderer(scene, device=-1, flags=flags)  # hiprtc for gfx950: used to fail with "use of undeclared identifier 'r'"
        assert len(r.code_object()) > 1000
        r.set_option("render_depth", 6)
        frames[label] = hb.host_kernel_for(r, scene, 48, 27, flags=flags).render(48, 27)["rgba32f"].copy()
    assert np.array_equal(frames["first"].view(np.uint32), frames["general"].view(np.uint32))
    assert np.array_equal(frames["general_by_flag"].view(np.uint32), frames["general"].view(np.uint32))
    assert np.array_equal(frames["first_baked"].view(np.uint32), frames["general"].view(np.uint32))
    assert len(np.unique(frames["first"].reshape(-1, 4), axis=0)) > 30


def test_out_arguments_that_references_would_change_are_refused(pa):
    """`out` / `inout` are lowered to C++ references, GLSL copies in and out.  The two differ only when the callee
    reaches the argument under another name (a mutable global it also names) or one variable feeds two out parameters (undefined
    order in GLSL).  Such a scene is refused at code generation; everything else -- the whole reference corpus -- goes through."""
    import glob

    base = open(pa.scene_path("basics")).read()
    head = 'library: ([\n        (\n            name: "extra",\n            data: (("%s")),\n        ),'
    snippet = """vec3 rel = r.o.xyz;
float a = 1.0;
%s
return SceneIntersectionWithMaterial(scene_intersection_none, material_empty());"""
    im = 'intersection_materials: ([\n        (\n            name: "probe",\n            data: ((("%s"))),\n        ),\n    ]),'

    def scene(library, body):
        text = base.replace("library: ([", head % library, 1).replace("intersection_materials: ([]),", im % (snippet % body), 1)
        return pa.Scene.from_text(text)

    counter = "float counter = 0.0;\nvoid bump(inout float x) { x += 1.0; counter += x; }\nvoid both(out float p, out float q) { p = 1.0; q = 2.0; }\nvoid outer(inout float x) { bump(x); }\n"
    assert "bump(a)" in scene(counter, "bump(a);").generate_source(0)                  # a local: references == copy in / copy out
    assert "both(a, rel.x)" in scene(counter, "both(a, rel.x);").generate_source(0)
    with pytest.raises(RuntimeError, match="global `counter` is an out / inout argument"):
        scene(counter, "bump(counter);").generate_source(0)
    with pytest.raises(RuntimeError, match="global `counter` is an out / inout argument"):
        scene(counter, "outer(counter);").generate_source(0)                           # through a call
    with pytest.raises(RuntimeError, match="two out / inout parameters"):
        scene(counter, "both(a, a);").generate_source(0)
    assert "both(rel.x, rel.y)" in scene(counter, "both(rel.x, rel.y);").generate_source(0)  # two components: two objects
    with pytest.raises(RuntimeError, match="two out / inout parameters"):
        scene(counter, "both(rel.x, rel.x);").generate_source(0)
    const_global = "const float k = 2.0;\nfloat scratch = 0.0;\nvoid setk(out float x) { x = k; }\n"
    assert "setk(scratch)" in scene(const_global, "setk(scratch);").generate_source(0)  # the callee does not name `scratch`
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "corpus", "scenes", "**", "*.ron"), recursive=True))
    assert len(files) >= 80
    for path in files:
        pa.Scene.from_file(path).generate_source(0)


def test_hoister_counts_the_out_arguments_of_glsl_builtins_as_writes(pa):
    """`float ip = 0.0; f = modf(x, ip);` writes `ip` through an argument -- it is not a write-once uniform local, and
    `ip * k_u * k_u * k_u` must not move to the prologue with ip = 0.0 (latent: the prelude has no modf / frexp yet)."""
    uniforms = {"k_u": "float", "x_u": "float"}
    code = "float ip = 0.0;\nfloat f = modf(x_u, ip);\nreturn vec4(ip * k_u * k_u * k_u + r.d.x, f, 0.0, 0.0);"
    out, prologue = pa.hoist_glsl(code, uniforms, params=["r"])
    assert "ip * k_u * k_u * k_u" in out and "ip" not in prologue
    # the same shape without the call IS a uniform local and does move
    plain, prologue = pa.hoist_glsl(code.replace("float f = modf(x_u, ip);", "float f = x_u;"), uniforms, params=["r"])
    assert "PTL_U.ptl_hv" in plain and "ip * k_u * k_u * k_u" in prologue


def test_first_trip_plane_tests_are_selfconsistent_and_change_no_bit_on_the_host(pa):
    """KernelOptions::first_trip_planes: one `vec4 ptl_dvo_<object>_<side>` per generated plane test behind the derived uniforms, written
    by derive() as `<plane>_mat_inv * out->ptl_dv_origin`, read by scene_intersect_first only; the host build of both forms draws the
    same bits (first trips take the first form there too), also with the scene state baked in and after a camera move."""
    import re
    from oracle import host_build as hb

    keep = pa.FLAG_KEEP_TRANSFORM_DODGES  # (round 6: the first-trip forms are opt-in; the default is the one general scene_intersect)
    scene = pa.Scene.from_file(pa.scene_path("triple_portal"))
    src = scene.generate_source(keep)
    block = src[src.index("struct ptl_uniform_block {"):src.index("};", src.index("struct ptl_uniform_block {"))]
    members = set(re.findall(r"vec4 (ptl_dvo_\d+_[01]);", block))
    assert len(members) == 27
    derive = src[src.index("PTL_FN void derive(ptl_uniform_block* out)"):src.index("// Material id -> what happens to the path.")]
    assert set(re.findall(r"out->(ptl_dvo_\d+_[01]) = \w+_mat_inv \* out->ptl_dv_origin;", derive)) == members
    first = src[src.index("PTL_FN SceneIntersection scene_intersect_first("):src.index("#define PTL_DV_OUT")]
    general = src[src.index("PTL_FN SceneIntersection scene_intersect(const Ray& r"):src.index("PTL_FN SceneIntersection scene_intersect_first(")]
    assert set(re.findall(r"PTL_U\.(ptl_dvo_\d+_[01])", first)) == members and "PTL_U.ptl_dvo_" not in general
    assert first.count("ptl_plane_cull_o(") == 27 and general.count("ptl_plane_cull(") == 27
    for flags in (0, keep | pa.FLAG_NO_DERIVED_UNIFORMS, keep | pa.FLAG_NO_FIRST_TRIP_PLANES, pa.FLAG_SPECIALIZE_INTS | pa.FLAG_NO_FIRST_TRIP_PLANES, keep | pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL):
        off = scene.generate_source(flags)
        assert "PTL_U.ptl_dvo_" not in off and "#define PTL_FIRST_TRIP_PLANES" not in off
    # (the specialised builds that keep run-time matrices still have the plane form by default: measured a gain there, a loss in the un-specialised kernel)
    assert "#define PTL_FIRST_TRIP_PLANES" in scene.generate_source(pa.FLAG_SPECIALIZE_INTS) and "#define PTL_FIRST_TRIP_PLANES" in scene.generate_source(pa.FLAG_SPECIALIZE_PATTERNS)
    frames = {}
    for label, flags in (("first", keep), ("general", 0), ("first_baked", pa.FLAG_SPECIALIZE_INTS)):
        sc = pa.Scene.from_file(pa.scene_path("triple_portal"))
        r = pa.SceneRenderer(sc, device=-1, flags=flags)
        r.set_option("render_depth", 12)
        got = [hb.host_kernel_for(r, sc, 64, 36, flags=flags).render(64, 36)["rgba32f"].copy()]
        r.set_camera((0.2, -0.3, 0.4), 2.3, 1.1, 2.9)
        got.append(hb.host_kernel_for(r, sc, 64, 36, flags=flags).render(64, 36)["rgba32f"].copy())
        frames[label] = got
    for k in range(2):
        assert np.array_equal(frames["first"][k].view(np.uint32), frames["general"][k].view(np.uint32))
        assert np.array_equal(frames["first_baked"][k].view(np.uint32), frames["general"][k].view(np.uint32))
    assert not np.array_equal(frames["general"][0].view(np.uint32), frames["general"][1].view(np.uint32))


def _flat_wall_with_extra_matrix(extra):
    from tests.synthetic import wall_scene

    return wall_scene(extra_matrices=extra, extra_objects='(name: "flat2", data: Flat(kind: Simple(Some(Named("squash"))), is_inside: (("return wall_M;")), in_subspace: Normal)),')


def test_a_matrix_with_infinities_keeps_every_full_chain(pa):
    """Products that skip zero terms (PTL_DROP_ZERO_TERMS, PTL_MASK_*) equal the full chains for finite vectors and for all-NaN ones.  A scene
    matrix with infinite elements -- the inverse of a matrix flattened along ONE axis -- sends +-inf components down the rays, for which the
    two differ (inf * 1 against inf * 1 + 0 * inf = NaN); the generator sees the values and keeps every full chain for that kernel.  An
    all-NaN inverse (scaled to zero on all axes: how the reference's scenes switch an object off) does not need that."""
    exact = '(name: "squash", data: Exact(i: (x: Value(1.0), y: Value(0.0), z: Value(0.0)), j: (x: Value(0.0), y: Value(%s), z: Value(0.0)), k: (x: Value(0.0), y: Value(0.0), z: Value(1.0)), pos: (x: Value(0.3), y: Value(0.0), z: Value(-1.0)))),'
    spec, ints = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL, pa.FLAG_SPECIALIZE_INTS
    flat = pa.Scene.from_text(_flat_wall_with_extra_matrix(exact % "0.0"))
    inv = np.asarray(flat.uniform_values()["squash_mat_inv"])
    assert np.isinf(inv).any() or (np.isnan(inv).any() and not np.isnan(inv).all())
    src = flat.generate_source(spec)
    assert "PTL_DROP_ZERO_TERMS" not in flat.generated_defines()
    assert "#define PTL_MASK_" not in flat.generate_source(ints)
    assert "#define PTL_MASK_" not in flat.generate_source(pa.FLAG_SPECIALIZE_STATIC) and "PTL_DROP_ZERO_TERMS" not in flat.generated_defines()
    regular = pa.Scene.from_text(_flat_wall_with_extra_matrix(exact % "1.5"))
    regular.generate_source(spec)
    assert "PTL_DROP_ZERO_TERMS" in regular.generated_defines()
    assert "#define PTL_MASK_squash_mat_inv" in regular.generate_source(ints)
    # the reference's idiom -- scale 0 on all axes -- gives an all-NaN inverse and keeps the short chains (the headline scene has two)
    pip = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    assert np.isnan(np.asarray(pip.uniform_values()["c0_mat_inv"])).all()
    pip.generate_source(spec)
    assert "PTL_DROP_ZERO_TERMS" in pip.generated_defines()
    assert src != regular.generate_source(spec)


def test_generated_defines_go_with_the_generated_source(pa):
    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    spec = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    # (PTL_FIRST_TRIP: the kernel has first-trip copies of its snippets -- not with affine rays, where a transform is cheaper than its dodge, and since
    # round 6 not in the un-specialised kernel unless asked for (FLAG_KEEP_TRANSFORM_DODGES): measured a loss there)
    keep = pa.FLAG_KEEP_TRANSFORM_DODGES
    for flags, want in ((0, set()), (keep, {"PTL_FIRST_TRIP"}), (spec, {"PTL_DROP_ZERO_TERMS", "PTL_AFFINE_RAYS"}), (spec | pa.FLAG_NO_AFFINE_RAYS, {"PTL_FIRST_TRIP", "PTL_DROP_ZERO_TERMS"}),
                        (spec | keep, {"PTL_FIRST_TRIP", "PTL_DROP_ZERO_TERMS", "PTL_AFFINE_RAYS"}),
                        (pa.FLAG_SPECIALIZE_PATTERNS, {"PTL_AFFINE_RAYS"}), (spec | pa.FLAG_EXACT_CR, {"PTL_FIRST_TRIP", "PTL_CONTRACT_V1"}),
                        (spec | pa.FLAG_FAST_MATH, {"PTL_AFFINE_RAYS", "PTL_FAST_MATH"}), (pa.FLAG_COUNT_SEGMENTS | pa.FLAG_ANAGLYPH, {"PTL_COUNT_SEGMENTS", "PTL_ANAGLYPH"}),
                        (pa.FLAG_COUNT_SEGMENTS | pa.FLAG_ANAGLYPH | keep, {"PTL_FIRST_TRIP", "PTL_COUNT_SEGMENTS", "PTL_ANAGLYPH"})):
        scene.generate_source(flags)
        # (PTL_JIT_MODULE_INLINER: this scene's intersection-material snippet loops, and with the Ints baked the loop is force-unrolled -- the
        # JIT picks LLVM's module inliner for those builds, kernel.cpp)
        # (... and so do the patterns builds since round 5: the loop bound `show_teleported` is one of the scene's switches they compile in)
        assert set(scene.generated_defines()) == want | ({"PTL_JIT_MODULE_INLINER"} if flags & (pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_PATTERNS) else set()), flags
    mono = pa.Scene.from_file(pa.scene_path("monoportal"))
    mono.generate_source(spec)
    assert set(mono.generated_defines()) == {"PTL_DROP_ZERO_TERMS", "PTL_AFFINE_RAYS"}  # no looping snippet: the toolchain's default inliner


def test_renderer_options_given_at_creation_are_in_the_first_build(pa, tmp_path, monkeypatch):
    """ptl_renderer_create_with_options: a specialised renderer compiles its mode switches in, so a caller that draws side by side (`portal-amd
    render --stereoimage`, and its compile-only prefetch renderers) gets THAT kernel from the first build -- the same code object a renderer
    reaches by switching the option afterwards, without the build nothing runs on."""
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))
    flags = pa.FLAG_SPECIALIZE_STATIC | pa.FLAG_QUICK_JIT
    seeded = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("basics")), device=-1, flags=flags, options={"draw_side_by_side": 1, "render_depth": 7})
    assert seeded.rejit_count() == 0
    assert seeded.uniform_value("_ray_tracing_depth", 8, 8) == 7
    switched = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("basics")), device=-1, flags=flags)
    flat = switched.code_object()
    switched.set_option("draw_side_by_side", 1)
    assert switched.code_object() == seeded.code_object() != flat and switched.rejit_count() == 1
    with pytest.raises(pa.PortalError):
        pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("basics")), device=-1, flags=flags, options={"no_such_option": 1})
    with pytest.raises(pa.PortalError):  # a rebuild at creation time makes no sense: the flag bit says it
        pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("basics")), device=-1, flags=flags, options={"specialize_static": 1})


def test_divisions_inside_a_library_define_are_the_contracts(pa):
    """A `#define` of a scene library is GLSL like any other text (the reference's own library has `#define PI2 (acos(-1.) / 2.0)`): its
    divisions become ptl_div (numerics contract 2: a * (1/b)) and its float literals get their suffix -- the macro draws the same bits
    as the same expression written as a function."""
    from oracle import host_build as hb
    from tests.synthetic import wall_scene

    out = pa.translate_glsl("#define PI2 (acos(-1.) / 2.0)\n#define THIRD(x) ((x) / 3.)\n#define K 0.5\n#ifdef GL_ES\nfloat f(float a) { return a / PI2 + THIRD(a) * K; }\n#endif\n")
    assert out.splitlines()[0].replace(" ", "") == "#definePI2(ptl_div(acos(-1.f),2.0f))"
    assert out.splitlines()[1].replace(" ", "") == "#defineTHIRD(x)(ptl_div((x),3.f))"
    assert out.splitlines()[2] == "#define K 0.5f" and out.splitlines()[3] == "#ifdef GL_ES"
    frames = []
    for lib in ('#define THIRD(x) ((x) / 3.)\\n#define K 0.37', 'float THIRD(float x) { return x / 3.; }\\nconst float K = 0.37;'):
        text = wall_scene(size=0.9, grid=True, library='(name: "lib", data: (("%s")))' % lib).replace(
            "if (abs(x) < size_u && abs(y) < size_u)", "if (abs(x) < THIRD(size_u * 2.5 + y) + K && abs(y) < THIRD(2. + x) / K * 0.4)")
        scene = pa.Scene.from_text(text)
        assert "ptl_div(" in scene.generate_source(0).split("namespace glsl {")[-1]
        r = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_QUICK_JIT)
        frames.append(hb.host_kernel_for(r, scene, 48, 32).render(48, 32)["rgba32f"].copy())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    wall = (frames[0][..., :3] != np.float32(0.6)).any(axis=2).mean()  # the region the divisions bound: part of the frame, not all of it
    assert 0.05 < wall < 0.95, wall


def test_zero_pattern_probes_are_reused_while_the_state_they_depend_on_stands(pa):
    """A renderer with baked Bool / Int uniforms regenerates its source on every scene-version bump, i.e. on every camera move; the zero
    patterns of its run-time matrices (two scene copies, up to 33 Scene::update + 34 evaluations to find) are reused while every value
    they depend on is the same, probed again when one moves, and the generated source is the same text either way."""
    ints = pa.FLAG_SPECIALIZE_INTS
    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    first = scene.generate_source(ints)
    assert scene.zero_mask_probes() == (0, 1)
    scene.set_camera_matrix(np.eye(4) + 0.01)  # a camera move: nothing the patterns depend on
    assert scene.generate_source(ints) == first and scene.zero_mask_probes() == (1, 1)
    scene.set_uniform("portal_rotate_angle", 0.3) if "portal_rotate_angle" in scene.uniform_values() else scene.set_uniform("pass_offset", 0.4)
    moved = scene.generate_source(ints)
    assert scene.zero_mask_probes() == (1, 2)
    fresh = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    fresh.set_camera_matrix(np.eye(4) + 0.01)
    fresh.set_uniform("portal_rotate_angle", 0.3) if "portal_rotate_angle" in fresh.uniform_values() else fresh.set_uniform("pass_offset", 0.4)
    assert fresh.generate_source(ints) == moved
    # a scene whose matrices follow the camera (Matrix::Camera): the cached patterns must still cover what such a matrix holds now
    cam = pa.Scene.from_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus", "scenes", "portal_in_portal_plus_ultra.ron"))
    a = cam.generate_source(ints)
    cam.set_camera_matrix(np.array([[0.0, 0, 1, 0.3], [0, 1, 0, 0.2], [-1, 0, 0, 0.1], [0, 0, 0, 1]]))
    b = cam.generate_source(ints)
    fresh = pa.Scene.from_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus", "scenes", "portal_in_portal_plus_ultra.ron"))
    fresh.set_camera_matrix(np.array([[0.0, 0, 1, 0.3], [0, 1, 0, 0.2], [-1, 0, 0, 0.1], [0, 0, 0, 1]]))
    assert fresh.generate_source(ints) == b and a.count("#define PTL_MASK_") == b.count("#define PTL_MASK_")


_BOUND_OK = """vec3 normal_a = -get_normal(a_mat);
SurfaceIntersection hit_a = plane_intersect(r, a_mat_inv, normal_a);
SceneIntersectionWithMaterial result = SceneIntersectionWithMaterial(scene_intersection_none, material_empty());
for (int size = 0; size < 4; size++) {
  if (nearer(result.scene.hit, hit_a)) {
    int is_inside = inside_a(r.o + r.d * hit_a.t, hit_a.u, hit_a.v, size);
    if (is_inside != NOT_INSIDE) {
      result.scene = process_portal_intersection(result.scene, hit_a, is_inside, CUSTOM_MATERIAL);
      if (result.scene.material == CUSTOM_MATERIAL) {
        result.material = material_teleport_transformed(offset_ray(r, hit_a.t), vec3(1.));
      }
    }
  }
}
return result;"""


def test_distance_bound_is_added_only_to_snippets_whose_shape_allows_it(pa):
    """glsl_translate.h `bound_nearer_blocks`: the accumulator pattern of the reference's intersection-material snippets gets the caller's
    bound; anything through which a skipped candidate could still be seen -- the accumulator read outside its blocks, an else branch, a store
    of something other than the tested hit, an outer local assigned in a block, a function with out parameters, a material the block does not
    store itself, the previous candidate's id read in front of the store, another initial value -- leaves the text exactly as written."""
    text, n = pa.bound_glsl(_BOUND_OK)
    assert n == 1 and "if (nearer(result.scene.hit, hit_a) && !(hit_a.t > ptl_far)) {" in text
    assert text.replace(" && !(hit_a.t > ptl_far)", "") == _BOUND_OK
    refused = {
        "read outside": _BOUND_OK.replace("return result;", "if (result.scene.hit.hit) { result.material = material_empty(); }\nreturn result;"),
        "else branch": _BOUND_OK.replace("  }\n}\nreturn", "  } else { size += 1; }\n}\nreturn"),
        "stores another hit": _BOUND_OK.replace("process_portal_intersection(result.scene, hit_a,", "process_portal_intersection(result.scene, other_hit,"),
        "outer local assigned": _BOUND_OK.replace("int is_inside = inside_a(", "normal_a = vec3(0.); int is_inside = inside_a("),
        "outer local incremented": _BOUND_OK.replace("int is_inside = inside_a(", "size++; int is_inside = inside_a("),
        "material not stored": _BOUND_OK.replace("        result.material = material_teleport_transformed(offset_ray(r, hit_a.t), vec3(1.));\n", ""),
        "id read before the store": _BOUND_OK.replace("    int is_inside = inside_a(", "    if (result.scene.material == CUSTOM_MATERIAL) { int k = 1; }\n    int is_inside = inside_a("),
        "other initial value": _BOUND_OK.replace("SceneIntersectionWithMaterial(scene_intersection_none, material_empty());", "first_guess(r);"),
        "whole accumulator assigned": _BOUND_OK.replace("      result.scene = process_portal_intersection(", "      result = other(r); result.scene = process_portal_intersection("),
        "not the last statement": _BOUND_OK.replace("return result;", "return result;\nreturn other(r);"),
        "nearer of another shape": _BOUND_OK.replace("nearer(result.scene.hit, hit_a)", "nearer(result.scene.hit, hits[size])"),
        # control flow out of a block: skipping a far candidate that ends in `break` would let the loop reach candidates it never saw
        "break in a block": _BOUND_OK.replace("        result.material = material_teleport_transformed(offset_ray(r, hit_a.t), vec3(1.));\n      }\n    }\n",
                                              "        result.material = material_teleport_transformed(offset_ray(r, hit_a.t), vec3(1.));\n      }\n    }\n    break;\n"),
        "continue in a block": _BOUND_OK.replace("    if (is_inside != NOT_INSIDE) {", "    if (is_inside == NOT_INSIDE) { continue; }\n    if (is_inside != NOT_INSIDE) {"),
        "return in a block": _BOUND_OK.replace("    if (is_inside != NOT_INSIDE) {", "    if (size == 3) { return result; }\n    if (is_inside != NOT_INSIDE) {"),
    }
    for why, body in refused.items():
        assert body != _BOUND_OK, why
        assert pa.bound_glsl(body) == (body, 0), why
    assert pa.bound_glsl(_BOUND_OK, out_functions=["inside_a"]) == (_BOUND_OK, 0)
    assert pa.bound_glsl(_BOUND_OK, out_functions=["something_else"])[1] == 1
    # scene level: the headline scene's snippet qualifies (both copies: general and first-trip); a scene that names subspace portals does not
    # (opt-in, FLAG_BOUNDED_SNIPPETS: on the headline scene it measures no gain, profiles/r04/ab_bounded_snippets.jsonl)
    pip = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    src = pip.generate_source(pa.FLAG_BOUNDED_SNIPPETS | pa.FLAG_KEEP_TRANSFORM_DODGES)
    assert src.count("&& !(hit_a.t > ptl_far)") == 2 and src.count("&& !(hit_b.t > ptl_far)") == 2 and "PTL_BOUNDED_SNIPPETS" in pip.generated_defines()
    src = pip.generate_source(pa.FLAG_BOUNDED_SNIPPETS)  # (round 6: one copy of the snippet by default)
    assert src.count("&& !(hit_a.t > ptl_far)") == 1 and src.count("&& !(hit_b.t > ptl_far)") == 1 and "PTL_BOUNDED_SNIPPETS" in pip.generated_defines()
    src = pip.generate_source(0)
    assert "> ptl_far)" not in src and "PTL_BOUNDED_SNIPPETS" not in pip.generated_defines()
    ultra = pa.Scene.from_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus", "scenes", "portal_in_portal_plus_ultra.ron"))
    assert "> ptl_far)" not in ultra.generate_source(pa.FLAG_BOUNDED_SNIPPETS) and "PTL_BOUNDED_SNIPPETS" not in ultra.generated_defines()


@pytest.mark.parametrize("flags_name", ["dynamic", "baked"])
def test_bounded_snippets_draw_the_frames_of_the_snippets_as_written(pa, flags_name):
    """The bounce loop with the bound (FLAG_BOUNDED_SNIPPETS: scene_intersect first, its hit distance handed to the snippet) against the snippet as written,
    evaluated first: the same bits, from the scene's own camera and from four views that look into, along and out of the nested portals
    (where the snippet's candidates ARE the nearest hits) -- and the numpy oracle, which knows nothing of either order, agrees."""
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    flags = 0 if flags_name == "dynamic" else (pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)
    views = [None, ((0.0, 0.0, 0.0), 0.2, 1.5, 1.6), ((0.3, -0.1, 0.2), 2.8, 1.0, 2.4), ((0.1, 0.3, -0.2), 1.2, 1.4, 3.0), ((0.0, 0.1, 0.0), 4.0, 1.7, 0.8)]
    w, h = 96, 54
    for view in views:
        frames = []
        for extra in (pa.FLAG_BOUNDED_SNIPPETS, 0):
            sc = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
            r = pa.SceneRenderer(sc, device=-1, flags=0)
            r.set_option("render_depth", 30)
            if view:
                r.set_camera(*view)
            frames.append(hb.host_kernel_for(r, sc, w, h, flags=flags | extra).render(w, h)["rgba32f"].copy())
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), view
        if flags_name == "dynamic":
            o = Oracle(pa.scene_path("portal_in_portal"))
            o.options["render_depth"] = 30
            if view:
                o.camera = dict(look_at=view[0], alpha=view[1], beta=view[2], r=view[3])
            want = o.render(w, h)
            assert np.array_equal(frames[0].view(np.uint32), want["rgba32f"].view(np.uint32)), view


def test_slices_entry_is_a_substitution_that_keeps_the_scene_lines_and_the_pictures(pa):
    """PTL_FLAG_SLICES (codegen.cpp `apply_slices_entry`): the render entry `ptl_render_slices_kernel` reads the uniform block of slice blockIdx.z
    from a buffer of blocks.  Without the flag the source is byte for byte what it was; with it the scene's own lines keep their numbers (compiler
    diagnostics map back to scene elements by line), the prelude sits inside the tracer struct (its uniform-reading functions see the slice), the
    classic entry is gone, a per-slice prologue entry is there -- and the host build of that source still draws the oracle's bits."""
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    for name in ("basics", "portal_in_portal"):
        scene = pa.Scene.from_file(pa.scene_path(name))
        plain, sliced = scene.generate_source(0), scene.generate_source(pa.FLAG_SLICES)
        assert plain == pa.Scene.from_file(pa.scene_path(name)).generate_source(0) and "ptl_slices" not in plain and "ptl_home" not in plain
        assert "ptl_render_slices_kernel(const glsl::ptl_uniform_block* __restrict__ ptl_slices" in sliced and "\nptl_render_kernel(" not in sliced
        assert "ptl_derive_slices_kernel(glsl::ptl_uniform_block* blocks)" in sliced and "ptl_teleport_kernel" in sliced
        line_of = lambda text, key: text[:text.index(key)].count("\n")
        for key in ("PTL_FN int is_inside_0(", "PTL_FN SceneIntersection scene_intersect(const Ray& r"):
            assert line_of(plain, key) == line_of(sliced, key), key
        # the prelude: in front of the tracer struct in the plain source, inside it in the sliced one
        assert line_of(plain, "PTL_FN float color_normal(") < line_of(plain, "struct ptl_tracer {")
        assert line_of(sliced, "struct ptl_tracer {") < line_of(sliced, "PTL_FN float color_normal(") < line_of(sliced, "};  // struct ptl_tracer")
        w, h = 48, 27
        r = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_QUICK_JIT)
        r.set_option("render_depth", 10)
        o = Oracle(pa.scene_path(name))
        o.options["render_depth"] = 10
        want = o.render(w, h)
        for flags in (pa.FLAG_SLICES, pa.FLAG_SLICES | pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL):
            sc = pa.Scene.from_file(pa.scene_path(name))
            got = hb.host_kernel_for(r, sc, w, h, flags=flags).render(w, h)
            assert np.array_equal(got["rgba32f"].view(np.uint32), want["rgba32f"].view(np.uint32)), (name, flags)
    # the sliced source compiles for gfx950 without a GPU; staging is host work (a snapshot of the uniform block), launching needs the device
    r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("basics")), device=-1, flags=pa.FLAG_SLICES | pa.FLAG_QUICK_JIT)
    assert r.code_object()[:4] == b"\x7fELF"
    r.stage_slice(pa.Frame(64, 64, 0, 1), 0)
    with pytest.raises(pa.PortalError):
        r.draw_slices(pa.Frame(64, 64, 0, 1), 1)
    with pytest.raises(pa.PortalError, match="not all staged"):
        r.draw_slices(pa.Frame(64, 64, 0, 1), 2)


def test_library_functions_of_a_scene_are_force_inlined(pa):
    """Scene libraries (scene.rs:1037-1044) are pasted into the kernel as translated GLSL: every function DEFINITION among them gets PTL_FN on
    its own line -- prototypes, struct heads, globals and #define lines do not -- so that LLVM's module inliner (which does not inline plain
    `inline` members by itself) leaves no call behind: a call passes `this`, and the tracer object would live in scratch."""
    import re

    text = """
#define TWICE(x) float twice_macro(x)
struct Pair { float a; float b; };
const vec3 tint = vec3(1., 0.5, 0.25);
float helper(float x);
float helper(float x) {
  return x * 2.;
}
vec3 shade(vec3 c, in float k, out float used) { used = k; if (k > 0.) { return c * helper(k); } return c; }
int
count_up(int n)
{
  int s = 0;
  for (int i = 0; i < 4; i++) { s += i; }
  return s;
}
"""
    out = pa.translate_library_glsl(text)
    assert out.count("PTL_FN") == 3
    assert "PTL_FN float helper(float x) {" in out and "PTL_FN vec3 shade(" in out and re.search(r"PTL_FN int\s+count_up\(int n\)\s*\{", out)
    assert "float helper(float x);" in out and "PTL_FN float helper(float x);" not in out
    assert "struct Pair {" in out and "const vec3 tint" in out and "#define TWICE(x) float twice_macro(x)" in out
    assert out.count("\n") == text.count("\n")  # line numbers of the scene text survive
    # every library function of the reference's scenes
    for name in ("portal_in_portal", "monoportal", "triple_portal", "mobius_monoportal", "basics"):
        src = pa.Scene.from_file(pa.scene_path(name)).generate_source(0)
        lib = src[src.index("// --- scene library snippets"):src.index("// --- is_inside_N / intersect_N wrappers")]
        depth, bare = 0, []
        for line in lib.splitlines():
            m = re.match(r"\}?\s*(PTL_FN\s+)?(int|float|bool|void|vec[234]|mat[234]|Ray|SceneIntersection|SurfaceIntersection|MaterialProcessing)\s+\w+\s*\([^;]*$", line)
            if depth == 0 or line.startswith("}"):
                if m and not m.group(1) and "(" in line and not line.rstrip().endswith(";"):
                    bare.append(line)
            depth += line.count("{") - line.count("}")
        assert not bare, (name, bare[:3])


def test_renderer_builds_are_split_in_a_render_and_a_teleport_module(pa, tmp_path, monkeypatch):
    """The camera-teleport entry is a second copy of the tracer and a fifth of every hiprtc build: the renderer compiles its kernels without it
    (-DPTL_RENDER_MODULE) and the other half (-DPTL_TELEPORT_MODULE: teleport entry + prologue) when a query first needs it, or ahead of time
    through prebuild_teleport -- a cached code object of its own.  PTL_ONE_MODULE=1: everything in one module, as layer 1 compiles a plain source."""
    import glob

    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))
    scene = pa.Scene.from_file(pa.scene_path("basics"))
    r = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_QUICK_JIT)
    code = r.code_object()
    assert b"ptl_render_kernel" in code and b"ptl_derive_kernel" in code and b"ptl_teleport_kernel" not in code
    assert len(glob.glob(str(tmp_path / "*.hsaco"))) == 1
    r.prebuild_teleport()
    files = glob.glob(str(tmp_path / "*.hsaco"))
    assert len(files) == 2
    other = next(open(f, "rb").read() for f in files if open(f, "rb").read() != code)
    assert b"ptl_teleport_kernel" in other and b"ptl_derive_kernel" in other and b"ptl_render_kernel" not in other
    r.prebuild_teleport()  # once per kernel
    assert len(glob.glob(str(tmp_path / "*.hsaco"))) == 2
    monkeypatch.setenv("PTL_ONE_MODULE", "1")
    whole = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_QUICK_JIT).code_object()
    assert b"ptl_render_kernel" in whole and b"ptl_teleport_kernel" in whole and len(whole) > len(code)
    # layer 1: a generated source compiled without either define has every entry
    k = pa.compile_source(scene.generate_source(0), scene, defines=list(scene.generated_defines()) + ["PTL_QUICK_JIT"]) if hasattr(pa, "compile_source") else None
    if k is not None:
        assert b"ptl_teleport_kernel" in k


def test_the_jit_picks_the_module_inliner_where_a_snippet_loop_is_unrolled(pa, tmp_path, monkeypatch):
    """kernel.cpp compile_options: LLVM's module inliner for kernels whose intersection-material snippet has a force-unrolled loop (define
    PTL_JIT_MODULE_INLINER from codegen.cpp: portal_in_portal with its Ints baked); the toolchain's bottom-up inliner otherwise.  Observable through
    the code-object cache: forcing the same choice with PTL_MODULE_INLINER gives the same file, forcing the other one a second file."""
    import glob

    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))
    n = lambda: len(glob.glob(str(tmp_path / "*.hsaco")))
    mono = pa.Scene.from_file(pa.scene_path("monoportal"))
    pa.SceneRenderer(mono, device=-1, flags=pa.FLAG_QUICK_JIT)
    assert n() == 1
    monkeypatch.setenv("PTL_MODULE_INLINER", "0")
    pa.SceneRenderer(mono, device=-1, flags=pa.FLAG_QUICK_JIT)
    assert n() == 1  # the default for a scene without looping snippets IS the bottom-up inliner
    monkeypatch.setenv("PTL_MODULE_INLINER", "1")
    pa.SceneRenderer(mono, device=-1, flags=pa.FLAG_QUICK_JIT)
    assert n() == 2
    monkeypatch.delenv("PTL_MODULE_INLINER")
    pip = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    baked = pa.FLAG_SPECIALIZE_INTS  # (not a quick build: those leave the loop rolled)
    pa.SceneRenderer(pip, device=-1, flags=baked)
    assert n() == 3
    monkeypatch.setenv("PTL_MODULE_INLINER", "1")
    pa.SceneRenderer(pip, device=-1, flags=baked)
    assert n() == 3  # ... and for the unrolled nested-portal loop the module inliner
    monkeypatch.setenv("PTL_MODULE_INLINER", "0")
    pa.SceneRenderer(pip, device=-1, flags=baked)
    assert n() == 4


def _note_max(code: bytes, key: bytes) -> int:
    """The largest msgpack unsigned integer that follows `key` in a code object's kernel metadata (kernel.cpp `code_object_note_max`)."""
    best, at = -1, code.find(key)
    while at >= 0:
        p = code[at + len(key):at + len(key) + 5]
        v = p[0] if p[0] <= 0x7F else (p[1] if p[0] == 0xCC else ((p[1] << 8) | p[2] if p[0] == 0xCD else -1))
        best = max(best, v)
        at = code.find(key, at + 1)
    return best


def test_a_kernel_just_above_128_registers_is_rebuilt_under_the_four_wave_cap(pa, tmp_path, monkeypatch):
    """128 VGPRs is where the fourth wave per SIMD goes.  The JIT compiles a kernel that lands a few registers above it -- and was given no
    occupancy hint -- once more with __launch_bounds__(256, 4) and keeps that build when it has no spills (kernel.cpp).  The case that showed it:
    portal_in_portal with the Panini switch compiled in and the slices entry, 130 VGPRs under the module inliner (0.262 -> 0.314 ms per frame)."""
    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    flags = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL | pa.FLAG_SPECIALIZE_STATIC | pa.FLAG_SLICES
    monkeypatch.setenv("PTL_JIT_OPT", "-O3")  # the shipped level (an earlier test of this file may have left its -O1 behind)
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path / "a"))
    monkeypatch.setenv("PTL_NO_OCCUPANCY_RETRY", "1")
    first = pa.SceneRenderer(scene, device=-1, flags=flags, options={"use_panini_projection": 1}).code_object()
    monkeypatch.delenv("PTL_NO_OCCUPANCY_RETRY")
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path / "b"))
    kept = pa.SceneRenderer(scene, device=-1, flags=flags, options={"use_panini_projection": 1}).code_object()
    if _note_max(first, b".vgpr_count") <= 128:
        pytest.skip("this toolchain builds the case within 128 registers by itself")
    assert 128 < _note_max(first, b".vgpr_count") <= 168
    assert _note_max(kept, b".vgpr_count") <= 128 and _note_max(kept, b".vgpr_spill_count") == 0 and _note_max(kept, b".private_segment_fixed_size") == 0
    # a kernel far above the cap keeps its registers: the capped build would spill (the un-specialised kernel of the same scene)
    big = pa.SceneRenderer(scene, device=-1, flags=0).code_object()
    assert _note_max(big, b".vgpr_spill_count") == 0 and _note_max(big, b".private_segment_fixed_size") == 0


def test_unit_elements_of_runtime_matrices_are_part_of_the_compiled_pattern(pa):
    """Round 4: the pattern of a matrix that stays a run-time value also names its elements that are exactly +1 or -1 (PTL_UNIT_BITS behind the
    zero mask): the term of such an element is `x + acc` / `acc - x` -- the same operation with the value known, what a fully baked build gets
    from constant folding.  The portal matrices of the headline scene are pure translations: every diagonal element is a known 1."""
    import re

    scene = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    src = scene.generate_source(pa.FLAG_SPECIALIZE_INTS)
    vals = scene.uniform_values()
    found = re.findall(r"#define PTL_MASK_(\w+) (0x[0-9a-f]{4})u(?: \| PTL_UNIT_BITS\((0x[0-9a-f]{4}), (0x[0-9a-f]{4})\))?", src)
    assert len(found) > 40
    with_units = 0
    for name, zero, ones, negs in found:
        a = np.asarray(vals[name], np.float32).T.reshape(-1)  # column-major: element 4 * column + row
        zero, ones, negs = int(zero, 16), int(ones or "0", 16), int(negs or "0", 16)
        for e in range(16):
            assert (a[e] != 0) <= bool((zero >> e) & 1), (name, e)
            assert bool((ones >> e) & 1) == (a[e] == 1.0) and bool((negs >> e) & 1) == (a[e] == -1.0), (name, e, a[e])  # (nothing animates here: the pattern IS the current state)
        with_units += bool(ones | negs)
    assert with_units > 40
    masks = {n: (int(o or "0", 16), int(g or "0", 16)) for n, _, o, g in found}
    assert masks["a_mat"] == (0x8421, 0x4000) and masks["b0_mat"] == (0xC421, 0x0000)  # translations by -1 and +1 along z
    # a scene state in which the portal is turned: the rotation block is no longer made of ones and zeros
    moved = pa.Scene.from_file(pa.scene_path("portal_in_portal"))
    assert moved.set_uniform("progress", 0.3)
    later = dict((n, (z, o, g)) for n, z, o, g in re.findall(r"#define PTL_MASK_(\w+) (0x[0-9a-f]{4})u(?: \| PTL_UNIT_BITS\((0x[0-9a-f]{4}), (0x[0-9a-f]{4})\))?", moved.generate_source(pa.FLAG_SPECIALIZE_INTS)))
    assert any(later.get(n, ("0xffff", "", ""))[1:] != (o, g) for n, _, o, g in found)
    # contract 1 and the tolerance mode never get patterns
    for flags in (pa.FLAG_SPECIALIZE_INTS | pa.FLAG_EXACT_CR, pa.FLAG_SPECIALIZE_INTS | pa.FLAG_FAST_MATH):
        assert "PTL_UNIT_BITS(0x" not in scene.generate_source(flags)


def test_a_module_inliner_build_that_cannot_be_capped_falls_back_to_the_bottom_up_pipeline(pa, tmp_path, monkeypatch):
    """The module inliner's schedule needs ~10 VGPRs more than the bottom-up pipeline's.  Where that costs the fourth wave per SIMD and the cap
    alone only produces spills (portal_in_portal_plus_ultra with its Ints baked: 139 registers, 128 under the toolchain's own pipeline), the JIT
    builds the kernel the old way under the same cap and keeps that (kernel.cpp)."""
    import os

    path = os.path.join(os.path.dirname(__file__), "corpus", "scenes", "portal_in_portal_plus_ultra.ron")
    scene = pa.Scene.from_file(path)
    monkeypatch.setenv("PTL_JIT_OPT", "-O3")
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path / "a"))
    monkeypatch.setenv("PTL_NO_OCCUPANCY_RETRY", "1")
    flags = pa.FLAG_SPECIALIZE_INTS
    first = pa.SceneRenderer(scene, device=-1, flags=flags, asset_root=os.path.join(os.path.dirname(__file__), "corpus")).code_object()
    if not 128 < _note_max(first, b".vgpr_count") <= 168:
        pytest.skip("this toolchain builds the case outside the band the retry looks at")
    monkeypatch.delenv("PTL_NO_OCCUPANCY_RETRY")
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path / "b"))
    kept = pa.SceneRenderer(scene, device=-1, flags=flags, asset_root=os.path.join(os.path.dirname(__file__), "corpus")).code_object()
    assert _note_max(kept, b".vgpr_count") <= 128 and _note_max(kept, b".vgpr_spill_count") == 0 and _note_max(kept, b".private_segment_fixed_size") == 0


def test_bound_glsl_reports_malformed_input_instead_of_throwing_across_the_c_boundary(pa):
    """ptl_bound_glsl: whatever the tokenizer makes of broken text, the call returns -- text unchanged or NULL + ptl_last_error --
    and the process lives."""
    for body in ["/* never closed", "if (nearer(result.scene.hit, hit_a)) {", "}}}}", "\x01\x02 \"unterminated", ""]:
        try:
            text, n = pa.bound_glsl(body)
            assert n == 0 and text == body
        except pa.PortalError as e:
            assert "ptl_bound_glsl" in str(e)


def test_code_object_metadata_is_read_per_kernel(pa, monkeypatch):
    """ptl_code_object_note: the JIT's occupancy retry looks at the RENDER entries of a module only -- a one-module build also holds
    the one-wave teleport and prologue entries.  Checked against the ELF notes as llvm-readelf prints them."""
    import re
    import subprocess

    monkeypatch.setenv("PTL_ONE_MODULE", "1")
    r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("monoportal")), device=-1, flags=pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)
    code = r.code_object()
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    per_kernel = {}
    if os.path.exists(readelf):
        import tempfile

        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(code)
            f.flush()
            notes = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(ptl_\w+_kernel)\s*$", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s+(\.vgpr_count|\.sgpr_count|\.private_segment_fixed_size|\.vgpr_spill_count):\s+(\d+)", line)
            if m and name:
                per_kernel.setdefault(name, {})[m.group(1)] = int(m.group(2))
        assert {"ptl_render_kernel", "ptl_teleport_kernel", "ptl_derive_kernel"} <= set(per_kernel)
        for key in (".vgpr_count", ".sgpr_count", ".private_segment_fixed_size", ".vgpr_spill_count"):
            assert r.code_object_note(key, "ptl_render") == per_kernel["ptl_render_kernel"][key]
            assert r.code_object_note(key, "ptl_teleport") == per_kernel["ptl_teleport_kernel"][key]
            assert r.code_object_note(key, "") == max(v[key] for v in per_kernel.values())
    assert r.code_object_note(".vgpr_count") > r.code_object_note(".vgpr_count", "ptl_derive") > 0
    assert r.code_object_note(".vgpr_count", "no_such_kernel") == r.code_object_note(".vgpr_count", "")   # no kernel of that name: every kernel counts
    assert r.code_object_note(".no_such_key") == -1 and len(r.code_object_sha256()) == 64


# ---- affine rays (round 5) -----------------------------------------------------------------------------------------------------------
def test_affine_rays_scan_of_the_scene_snippets(pa):
    """codegen.cpp `snippets_keep_rays_affine`: what a snippet may do to a ray while every origin keeps w = 1 and every direction w = 0 --
    library transforms by scene matrices, offsets along the direction, halves spelled with their w -- and what switches the optimisation off."""
    keeps = ["Ray r3 = transform(b0_mat_inv, r_b); r3 = normalize_ray(r3);", "r.o += r.d * _offset_after_material;", "r.o = r.o + r.d * (_offset_after_material + hit.t);",
             "r.d = vec4(m * (inverse(n) * r.d.xyz), 0.);", "r.o = vec4(f(i.u, i.v), 1.0);", "return Ray(vec4(mobius_o(u), 1.), vec4(mobius_d(u), 0.), 1.0, false);",
             "r_mob.d = normalize(r_mob.d);", "r.o.xyz -= offset_box; r.o.x += a - b;", "if (r.o == r.d) { float w = r.o.w + r.d.w; }",
             "Ray q = transform(a_mat, transform(b0_mat_inv, r_b)); q = transform(a_to_b_mat_teleport, q);", "vec4 p = b0_mat * (a_mat_inv * pos);", "// r.d = -r.d;\nfloat x = 1.;", "return transform(a_mat_inv, r);", "Ray transform(mat4 m, Ray r) { return r; }", "return material_teleport(hit, r, a_to_b_mat_teleport);"]
    refused = {"Ray q = Ray(ray_o, ray_d, 1.0, false);": "Ray built from halves", "r.o -= center;": "not known to keep its w", "r.d = -r.d;": "not known to keep its w",
               "r.o.w = 2.;": "may reach the w", "r.d.xw += vec2(1.);": "may reach the w", "void f(inout Ray r) { }": "out parameter", "void g(out vec4 p) { p = vec4(0.); }": "out parameter",
               "r.d = get_mat(int(hit.v)) * r.d;": "not known to keep its w", "r.o = vec4(p, 0.);": "not known to keep its w", "r.d = normalize(q.d);": "not known to keep its w",
               "x.d /= 2.;": "not known to keep its w", "r3 = transform(mat_transform_inv, r);": "not a scene uniform", "Ray q = transform(inverse(a_mat), r);": "not a scene uniform",
               "return material_teleport(hit, r, inverse(a_mat));": "not a scene uniform", "r.o += r.d * a_mat;": "not known to keep its w",
               "r.o = r.o + r.d * mat4(2.);": "not known to keep its w", "return transform(m, r);": "not a scene uniform", "return offset_ray(transform(get_mat(k), r), 0.1);": "not a scene uniform",
               "Ray q = Ray(vec4(o, 1.), vec4(d, 1.), 1., false);": "Ray built from halves", "Ray q = Ray(vec4(o.x, o.y, 1.), vec4(d, 0.), 1., false);": "Ray built from halves"}
    # round 6: the scan is a whitelist over EVERY write to a ray half -- each of these passed round 5's
    keeps += ["r.o[0] = 2.; r.d[2] -= 1.; q.r.o[1] *= 3.;", "r.o.xy = r.o.yx; r.d.stp = vec3(0.); r.o.rgb += c;", "float w = r.o[3] + r.d.w + rs[k].o.w;",
              "r.o += r.d * -t;", "r.o += r.d * (a + b) * f(x, y) / q.z;", "rs[k].o.x = 1.; rs[k].o += rs[k].d * t;", "for (int k = 0; k < 3; k++) { r.o.x++; ++r.d.y; }",
              "for (int k = 0; k < 3; r.o += r.d * s) { k++; }", "mat3 m = mat3(1.); vec3 p = m * r.o.xyz; r.o.xyz = p;", "float d = hit.d; s.d = 2.;" if False else "float e = r.d.x;"]
    refused.update({
        "r.o[3] = 2.0;": "may reach the w", "r.d[3] += 1.0;": "may reach the w", "q.r.o[3] = 0.5;": "may reach the w", "r.o[k] = 1.;": "may reach the w", "r.o[1 + 2] = 1.;": "may reach the w",
        "r.o.xyz.x = 1.; r.o.wzyx.x = 2.;": "may reach the w", "r.d.a = 1.;": "may reach the w", "r.d.q++;": "may reach the w", "--r.o.w;": "may reach the w", "++r.o;": "not known to keep its w",
        "r.o.W = 2.;": "may reach the w", "r.o.xyzw.xyz = p;": "may reach the w",
        "r.o += r.d * t + vec4(0., 0., 0., 5.);": "not known to keep its w", "r.o = r.o + r.d * t + vec4(0., 0., 0., 5.);": "not known to keep its w",
        "r.o += r.d * t - v;": "not known to keep its w", "r.o += r.d * t, q.o = v;": "not known to keep its w", "r.o += r.d * c ? a : b;": "not known to keep its w",
        "mat4 m = mat4(2.); r.o += r.d * m;": "not known to keep its w", "mat4 a = mat4(1.), m2 = a; r.o += r.d * t * m2;": "not known to keep its w",
        "mat4 spin(float a) { return mat4(1.); }\nvoid f(Ray r) { r.o += r.d * spin(1.); }": "not known to keep its w", "void f(mat4 k, Ray r) { r.o = r.o + r.d * k; }": "not known to keep its w",
        "r.o += r.d * outerProduct(a, b);": "not known to keep its w", "r.o *= 2.;": "not known to keep its w", "r.o %= 2.;": "not known to keep its w",
        "#define SET_W(r) r.o.w = 2.\nvoid f(Ray r) { SET_W(r); }": "preprocessor", "#define HALF o\nvoid f(Ray r) { r.HALF.w = 2.; }": "preprocessor", "#define M \\\n mat4(2.)\nfloat x;": "preprocessor",
        "#if 1\nfloat x;\n#endif": "preprocessor",
        "float i; float f = modf(x, r.o.w);": "builtin with an out parameter", "int e; float m = frexp(x, e);": "builtin with an out parameter",
        "void g(out float w) { w = 2.; }\nvoid f(Ray r) { g(r.o.w); }": "out parameter", "void g(inout vec3 v) { }": "out parameter",
        "Ray q = ray_none; q = transform(a_mat, q);": "ray constant", "return MaterialProcessing(false, c, ray_none);": "ray constant",
        "void f(mat4 a_mat, Ray r) { r = transform(a_mat, r); }": "not a scene uniform", "mat4 b_mat_inv = mat4(2.); return material_teleport(hit, r, b_mat_inv);": "not a scene uniform",
    })
    for code in keeps:
        assert pa.snippets_keep_rays_affine(code) == (True, ""), code
    for code, why in refused.items():
        ok, said = pa.snippets_keep_rays_affine(code)
        assert not ok and why in said, (code, said)


def test_affine_rays_are_generated_only_where_they_hold(pa, tmp_path):
    """Which builds get PTL_AFFINE_RAYS: a build that may shorten products, of a scene whose matrices all have the bottom row 0 0 0 1 (or are NaN
    throughout) and whose snippets pass the scan -- never the un-specialised build, contract 1 or a kernel that keeps its
    full chains; a renderer whose camera leaves the affine maps rebuilds without it."""
    spec = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL

    def has(scene, flags):
        scene.generate_source(flags)
        return "PTL_AFFINE_RAYS" in scene.generated_defines()

    for name in ("basics", "monoportal", "triple_portal", "portal_in_portal", "mobius_monoportal"):
        scene = pa.Scene.from_file(pa.scene_path(name))
        assert [has(scene, f) for f in (0, pa.FLAG_SPECIALIZE_INTS, pa.FLAG_SPECIALIZE_PATTERNS, pa.FLAG_SPECIALIZE_STATIC, spec)] == [False, True, True, True, True], name
        assert not has(scene, spec | pa.FLAG_NO_AFFINE_RAYS) and not has(scene, spec | pa.FLAG_EXACT_CR) and has(scene, spec | pa.FLAG_FAST_MATH)  # (the tolerance mode has them too)
    corpus = os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus", "scenes")
    # a Ray from unknown halves / a matrix the snippet computes itself: refused by the scan
    for name in ("half_spheres", "portal_in_portal_plus_ultra", "trefoil", "cylinder"):
        assert not has(pa.Scene.from_file(os.path.join(corpus, name + ".ron")), spec), name
    # a projective matrix among the scene's uniforms (bottom row 0.25 0 0 1): no affine rays, in any build
    text = open(pa.scene_path("basics")).read()
    exact_full = ('data: ExactFull(c0: (x: Value(1.0), y: Value(0.0), z: Value(0.0), w: Value(0.25)), c1: (x: Value(0.0), y: Value(1.0), z: Value(0.0), w: Value(0.0)), '
                  'c2: (x: Value(0.0), y: Value(0.0), z: Value(1.0), w: Value(0.0)), c3: (x: Value(0.5), y: Value(0.0), z: Value(0.0), w: Value(1.0)))')
    import re

    m = re.search(r'name: "room_red",\s*data: \w+\((?:[^()]|\([^()]*\))*\)', text)
    assert m, "basics.ron: the matrix `room_red`"
    projective = pa.Scene.from_text(text[:m.start()] + 'name: "room_red", ' + exact_full + text[m.end():])
    assert not has(projective, spec) and not has(projective, pa.FLAG_SPECIALIZE_INTS)
    # the camera is a run-time value in every build: one that is not affine (a named camera whose teleport matrix has a bottom row of its own)
    cam_text = text.replace("matrix: (1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0),", "matrix: (1.0, 0.0, 0.0, 0.125, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0),", 1)
    assert cam_text != text
    for flags in (spec, pa.FLAG_SPECIALIZE_PATTERNS):
        r = pa.SceneRenderer(pa.Scene.from_text(cam_text), device=-1, flags=flags | pa.FLAG_QUICK_JIT)
        assert r.affine_rays() and r.rejit_count() == 0
        affine_binary = r.code_object()
        r.use_camera("doorway")
        assert not r.affine_rays() and r.rejit_count() == 1
        assert r.code_object() != affine_binary and "#define PTL_AFFINE_RAYS" not in r.kernel_source()   # really another kernel (the flag is part of the source text)
        r.use_camera("red")   # back among the affine maps: the assumption stays off for this stage (one rebuild, not one per camera move)
        assert not r.affine_rays() and r.rejit_count() == 1


@pytest.mark.parametrize("name, depth", [("basics", 6), ("monoportal", 10), ("triple_portal", 10), ("portal_in_portal", 8), ("mobius_monoportal", 12)])
def test_affine_rays_draw_the_bits_of_the_general_products(pa, name, depth):
    """PTL_AFFINE_RAYS spells o.w = 1 / d.w = 0 in the matrix-times-ray products: the same operations on the same values.  The host build of
    the generated source draws the same bits with and without it -- Int-baked (masked matrices), patterns-only, everything baked -- and (through
    tests/test_corpus.py, whose builds have it on) the bits of the numpy oracle, which knows nothing of it."""
    from oracle import host_build as hb

    w, h = 64, 36
    for label, flags in (("ints", pa.FLAG_SPECIALIZE_INTS), ("patterns", pa.FLAG_SPECIALIZE_PATTERNS), ("baked", pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL)):
        frames = []
        for extra in (0, pa.FLAG_NO_AFFINE_RAYS):
            sc = pa.Scene.from_file(pa.scene_path(name))
            r = pa.SceneRenderer(sc, device=-1, flags=flags | extra | pa.FLAG_QUICK_JIT)
            r.set_option("render_depth", depth)
            src = sc.generate_source(flags | extra)
            assert ("PTL_AFFINE_RAYS" in sc.generated_defines()) == (extra == 0), (name, label)
            frames.append(hb.host_kernel_for(r, sc, w, h, flags=flags | extra).render(w, h)["rgba32f"].copy())
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)), (name, label)
        assert len(np.unique(frames[0].reshape(-1, 4), axis=0)) > 50


# ---- round 6: ADVICE r5 ---------------------------------------------------------------------------------------------------------------------
def test_a_layer_1_kernel_with_affine_rays_refuses_a_matrix_that_is_not_affine(pa):
    """ADVICE r5 (include/portal_amd.h layer 1): a caller that binds only ptl_kernel_* owns `_camera` and the run-time `X_mat` uniforms; a kernel
    generated with PTL_AFFINE_RAYS is valid for affine matrices only, and `ptl_kernel_set_uniform` says so instead of drawing wrong frames."""
    spec = pa.FLAG_SPECIALIZE_INTS
    scene = pa.Scene.from_file(pa.scene_path("monoportal"))
    source = scene.generate_source(spec)
    defines = scene.generated_defines()
    assert "PTL_AFFINE_RAYS" in defines
    layout, size = scene.uniform_layout()
    k = pa.Kernel(source, layout, size, device=-1, defines=defines)
    general = pa.Kernel(scene.generate_source(spec | pa.FLAG_NO_AFFINE_RAYS), layout, size, device=-1, defines=scene.generated_defines())
    affine = np.eye(4, dtype=np.float32)
    affine[:3, 3] = (1.0, 2.0, 3.0)
    projective = affine.copy()
    projective[3, 0] = 0.25
    matrices = [n for n, t, _ in layout if t == pa.PTL_MAT4 and (n == "_camera" or n.endswith(("_mat", "_mat_inv", "_mat_teleport")))]
    assert "_camera" in matrices and len(matrices) >= 3
    for name in matrices:
        assert k.set_uniform(name, pa.PTL_MAT4, affine) == 0
        assert k.set_uniform(name, pa.PTL_MAT4, np.full((4, 4), np.nan, np.float32)) == 0   # a switched-off object
        with pytest.raises(pa.PortalError, match="not an affine matrix"):
            k.set_uniform(name, pa.PTL_MAT4, projective)
        assert general.set_uniform(name, pa.PTL_MAT4, projective) == 0   # the general products take any matrix


def test_one_return_to_affine_rays_per_stage(pa):
    """a camera that leaves the affine maps switches affine rays off; once it is back, a rebuild that happens anyway (here: a mode switch)
    returns to them -- once per stage, so a state that flickers cannot rebuild per frame."""
    text = open(pa.scene_path("basics")).read()
    cam_text = text.replace("matrix: (1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0),", "matrix: (1.0, 0.0, 0.0, 0.125, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0),", 1)
    assert cam_text != text
    r = pa.SceneRenderer(pa.Scene.from_text(cam_text), device=-1, flags=pa.FLAG_SPECIALIZE_PATTERNS | pa.FLAG_QUICK_JIT)
    assert r.affine_rays()
    r.use_camera("doorway")
    assert not r.affine_rays() and r.rejit_count() == 1
    r.use_camera("red")
    assert not r.affine_rays() and r.rejit_count() == 1      # no rebuild of its own
    r.set_option("use_360_camera", 1)                          # a rebuild for another reason ...
    assert r.affine_rays() and r.rejit_count() == 2           # ... returns to affine rays
    r.use_camera("doorway")
    assert not r.affine_rays() and r.rejit_count() == 3
    r.use_camera("red")
    r.set_option("use_360_camera", 0)
    assert not r.affine_rays() and r.rejit_count() == 4       # the second break of this stage keeps them off


@pytest.mark.parametrize("name, depth", [("basics", 6), ("portal_in_portal", 8), ("mobius_monoportal", 10)])
def test_material_tables_draw_the_bits_of_the_reference_chain(pa, name, depth):
    """Round 6 (A/B builds, off by default after measuring): the Simple materials' nine literals from a table -- in LDS, or behind scalar loads -- and one
    material_simple2 call for all of them, against the reference's chain of one inlined call per material (src/gui/scene.rs:736-760): same function, same
    argument values, so the host build of the generated source draws the same bits; the table holds every Simple material and the three DEBUG_* ones."""
    from oracle import host_build as hb

    w, h = 64, 36
    spec = pa.FLAG_SPECIALIZE_INTS | pa.FLAG_SPECIALIZE_ALL
    frames = []
    for extra in (0, pa.FLAG_MATERIAL_TABLE_LDS, pa.FLAG_MATERIAL_TABLE_SCALAR):
        sc = pa.Scene.from_file(pa.scene_path(name))
        r = pa.SceneRenderer(sc, device=-1, flags=spec | extra | pa.FLAG_QUICK_JIT)
        r.set_option("render_depth", depth)
        src = sc.generate_source(spec | extra)
        assert ("#define PTL_MATERIAL_TABLE " in src) == (extra != 0)
        copies = src.count("return material_simple2(hit, r, vec3(")
        if extra:   # no generated per-material copy is left (what remains is the scene author's own Complex materials)
            assert "ptl_material_in_table" in src and copies == chain_copies - sum(1 for m in sc.materials() if m["kind"] == "Simple") if hasattr(sc, "materials") else copies < chain_copies
        else:
            chain_copies = copies
        frames.append(hb.host_kernel_for(r, sc, w, h, flags=spec | extra).render(w, h)["rgba32f"].copy())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)) and np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32))
    assert len(np.unique(frames[0].reshape(-1, 4), axis=0)) > 50


def test_lane_options_are_renderer_options_that_do_not_touch_the_kernel(pa):
    """Round 6: `concurrent_draws` K and `lane_fence` 0 / 1 (two frames in flight, include/portal_amd.h) are scheduling options: accepted on a renderer
    without a device too, out of range refused, and neither changes the source a renderer would compile (no rebuild, same code-object key)."""
    r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("basics")), device=-1, flags=0)
    before = r.kernel_source()
    for name, value in (("concurrent_draws", 2), ("lane_fence", 0), ("lane_stagger_us", 90.0), ("lane_stagger_us", 0), ("lane_fence", 1), ("concurrent_draws", 1)):
        r.set_option(name, value)
        assert r.kernel_source() == before and r.rejit_count() == 0
    for bad in (0, 9):
        with pytest.raises(pa.PortalError):
            r.set_option("concurrent_draws", bad)
    with pytest.raises(pa.PortalError):
        r.set_option("lane_fences", 1)
    with pytest.raises(pa.PortalError):
        r.set_option("lane_stagger_us", -1.0)
    r.join()  # nothing in flight: a no-op
