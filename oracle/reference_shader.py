"""oracle/reference_shader.py -- the reference's OWN shader text, executed.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import
anything under oracle/; the product (portal_amd/) never does.

What this is.  The reference has no CPU tracer: its definition of the hot path is GLSL *text* --
`src/library.glsl` (the prelude), `src/frag.glsl` (the template with `//%slot//%` holes) and the
per-scene slot text `Scene::generate_shader_code` prints (`src/gui/scene.rs:693-1110`) -- handed to a
GL driver.  `oracle/portal_oracle.py` restates that text by hand (`Natives`, `ray_tracing`,
`get_color2` ...).  This module restates NOTHING of it: it reads the two files where they lie
(`/root/reference/src`), fills the slots exactly as `generate_shader_code` does (slot emitter below:
that part is Rust in the reference, so it is the one piece that has to be restated), runs the
reference's tagged-line filter and its template engine (`src/code_generation.rs:82-98`), and executes
the resulting translation unit -- `void main()` included -- with `oracle/glsl_interp.py` under the
builtin contract of `oracle/glsl_math.py`.  It is the `oracle/_ref` of this repository: the nearest
thing to "the reference itself, run here" that exists without a GL driver.

What it pins.  Everything the hand restatement could have got wrong: operation order, operand order,
branch structure, early returns, the bounce loop, darkening, Panini, side-by-side, encode_float, the
generated plane / portal / complex-object tests, material dispatch.  What it cannot pin: the precision
a particular GL driver gives `/`, `sqrt`, `sin` ... (GLSL ES 3.00 leaves that to the implementation);
both oracles and the GPU kernel share ONE builtin contract for those.  So the oracle's status after
this module is "pinned to the reference's source text; builtin precision by contract".

Where the text comes from.  `/root/reference/src/{library,frag}.glsl` when that tree is mounted (the
build container).  The GPU box has no such tree, so `build_artifact()` -- called by
`__graft_entry__.build()` -- packs the two files into `oracle/_ref/reference_shader.bin` (zlib, git-
ignored like any other build output, travels with the snapshot like a built .so).  No reference
source is committed; `tests/golden/reference_text_*.npz` hold OUTPUTS (frames, function values) this
module produced, with the script that made them.
"""
from __future__ import annotations

import hashlib
import json
import math
import os
import zlib
from decimal import Decimal

import numpy as np

from . import glsl_math as M
from . import glsl_values as V
from .glsl_interp import GlslError, Interp, tokenize
from .glsl_values import Mat, Sampler, Vec
from .portal_oracle import Oracle, to_rgba8

F32, I32 = np.float32, np.int32

REFERENCE_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ARTIFACT = os.path.join(HERE, "_ref", "reference_shader.bin")
FILES = ("library.glsl", "frag.glsl")


# =============================================================================================
# the text
# =============================================================================================
def build_artifact(src: str = REFERENCE_SRC, out: str = ARTIFACT) -> str:
    """The recipe: pack the reference's two shader files into oracle/_ref (a build output)."""
    blob = {name: open(os.path.join(src, name), encoding="utf-8").read() for name in FILES}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "wb") as f:
        f.write(zlib.compress(json.dumps(blob).encode("utf-8"), 9))
    return out


def available() -> bool:
    return all(os.path.exists(os.path.join(REFERENCE_SRC, n)) for n in FILES) or os.path.exists(ARTIFACT)


def reference_texts() -> dict:
    """{'library.glsl': text, 'frag.glsl': text, 'origin': where they came from}"""
    if all(os.path.exists(os.path.join(REFERENCE_SRC, n)) for n in FILES):
        out = {n: open(os.path.join(REFERENCE_SRC, n), encoding="utf-8").read() for n in FILES}
        out["origin"] = REFERENCE_SRC
        return out
    if os.path.exists(ARTIFACT):
        out = json.loads(zlib.decompress(open(ARTIFACT, "rb").read()).decode("utf-8"))
        out["origin"] = ARTIFACT
        return out
    raise FileNotFoundError("reference shader text: neither /root/reference/src nor oracle/_ref/reference_shader.bin "
                            "(run __graft_entry__.build() where /root/reference is mounted)")


def text_digest() -> str:
    t = reference_texts()
    return hashlib.sha256((t["library.glsl"] + "\0" + t["frag.glsl"]).encode("utf-8")).hexdigest()[:16]


# =============================================================================================
# Scene::generate_shader_code, the Rust half (src/gui/scene.rs:693-1075)
# =============================================================================================
def rust_lower_exp(x: float) -> str:
    """Rust's `{:e}` for f64: shortest round-trip digits, d[.ddd]e<exp> (core::fmt::float)."""
    if x != x:
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    if x == 0.0:
        return "-0e0" if math.copysign(1.0, x) < 0 else "0e0"
    sign, digits, exp = Decimal(repr(float(x))).as_tuple()
    digits = list(digits)
    e10 = exp + len(digits) - 1
    while len(digits) > 1 and digits[-1] == 0:
        digits.pop()
    mant = str(digits[0]) + ("." + "".join(map(str, digits[1:])) if len(digits) > 1 else "")
    return ("-" if sign else "") + mant + "e" + str(e10)


def _b(v) -> str:
    return "true" if v else "false"


def emit_slots(scene, uniform_types: dict) -> dict:
    """name -> text for every `//%name//%` hole of frag.glsl, as generate_shader_code prints them.
    `uniform_types`: name -> 'mat4' | 'float' | 'int' | ... for every uniform the host sets."""
    e = rust_lower_exp
    mname = lambda idx: scene.matrices[idx][0]
    slots = {}
    # uniforms (scene.rs:668-692 over scene.rs:424-543): every non-builtin uniform the host sets
    decl = []
    for name in sorted(n for n, t in uniform_types.items() if t == "mat4" and not n.startswith("_")):
        decl.append(f"uniform mat4 {name};\n")
    for name, t in uniform_types.items():
        if not name.startswith("_") and t != "mat4":
            decl.append(f"uniform {t} {name};\n")
    slots["uniforms"] = "".join(decl)
    # textures (scene.rs:702-718): BTreeSet of texture and video names
    names = sorted({n for n, _ in scene.textures} | set(scene.videos))
    slots["textures"] = "".join(f"uniform sampler2D {n}_tex;\n" for n in names)
    # materials (scene.rs:720-842)
    defines, processing, counter = [], [], 0
    for m in scene.materials:
        name_m = m["name"] + "_M"
        defines.append(f"#define {name_m} (USER_MATERIAL_OFFSET + {counter})\n")
        counter += 1
        processing.append(f"}} else if (i.material == {name_m}) {{\n")
        if m["kind"] == "Simple":
            c = m["color"]
            processing.append(f"return material_simple2(hit, r, vec3({e(c[0])}, {e(c[1])}, {e(c[2])}), {e(m['normal_coef'])}, {_b(m['grid'])}, "
                              f"{e(m['grid_scale'])}, {e(m['grid_coef'])}, {_b(m['grid2'])}, {_b(m['grid3'])});\n")
        elif m["kind"] == "Reflect":
            c = m["color"]
            processing.append(f"return material_reflect(hit, r, vec3({e(c[0])}, {e(c[1])}, {e(c[2])}));\n")
        elif m["kind"] == "Refract":
            c = m["color"]
            processing.append(f"return material_refract(hit, r, vec3({e(c[0])}, {e(c[1])}, {e(c[2])}), {e(m['refractive_index'])});\n")
        else:
            processing.append(m["code"])
            processing.append("\n")
    for pos, o in enumerate(scene.objects):
        if o["kind"] == "debug" or not o["portal"] or o["m0"] < 0 or o["m1"] < 0:
            continue
        a, b = mname(o["m0"]), mname(o["m1"])
        for which, (frm, to) in ((1, (a, b)), (2, (b, a))):
            defines.append(f"#define teleport_{pos}_{which}_M (USER_MATERIAL_OFFSET + {counter})\n")
            counter += 1
        processing.append(f"}} else if (i.material == teleport_{pos}_1_M) {{\n")
        processing.append(f"return material_teleport(hit, r, {a}_to_{b}_mat_teleport);")
        processing.append(f"}} else if (i.material == teleport_{pos}_2_M) {{\n")
        processing.append(f"return material_teleport(hit, r, {b}_to_{a}_mat_teleport);")
    slots["materials_defines"] = "".join(defines)
    slots["material_processing"] = "".join(processing)
    # intersection functions (scene.rs:847-883)
    fn = []
    for pos, o in enumerate(scene.objects):
        if o["kind"] == "flat":
            tail = ", bool first" if o["portal"] else ""
            fn.append(f"int is_inside_{pos}(vec4 pos, float x, float y, bool back{tail}) {{\n{o['code']}\n}}\n")
        elif o["kind"] == "complex":
            tail = ", bool first" if o["portal"] else ""
            fn.append(f"SceneIntersection intersect_{pos}(Ray r{tail}) {{\n{o['code']}\n}}\n")
    slots["intersection_functions"] = "".join(fn)
    # intersections (scene.rs:885-1009)
    body = []
    for pos, o in enumerate(scene.objects):
        if o["m0"] < 0 or (o["kind"] != "debug" and o["portal"] and o["m1"] < 0):
            raise GlslError(f"object {pos}: matrix not set (the reference generates no shader at all: `matrix?`)")
        if o["kind"] == "debug":
            x = mname(o["m0"])
            body.append(f"transformed_ray = transform({x}_mat_inv, r);\nlen = length(transformed_ray.d);\ntransformed_ray = normalize_ray(transformed_ray);")
            body.append("ihit = debug_intersect(transformed_ray);\nihit.hit.t /= len;\n")
            body.append(f"if (nearer(i, ihit)) {{ i = ihit; i.hit.n = normalize(adjugate({x}_mat_inv) * i.hit.n); }}\n\n")
            body.append("\n")
            continue
        guard = {"Normal": "if (r.in_subspace == false) {", "Subspace": "if (r.in_subspace == true) {", "Both": None}[o["sub"]]
        if guard:
            body.append(guard)
        sides = [(mname(o["m0"]), True, f"teleport_{pos}_1_M"), (mname(o["m1"]), False, f"teleport_{pos}_2_M")] if o["portal"] else None
        if o["kind"] == "flat":
            if sides is None:
                x = mname(o["m0"])
                body.append(f"normal = -get_normal({x}_mat);\n")
                body.append(f"hit = plane_intersect(r, {x}_mat_inv, get_normal({x}_mat));\n")
                body.append(f"if (nearer(i, hit)) {{ i = process_plane_intersection(i, hit, is_inside_{pos}(r.o + r.d * hit.t, hit.u, hit.v, "
                            f"is_collinear(hit.n, normal))); }}\n\n")
            else:
                for x, first, material in sides:
                    body.append(f"normal = {'-' if first else ''}get_normal({x}_mat);\n")
                    body.append(f"hit = plane_intersect(r, {x}_mat_inv, normal);\n")
                    body.append(f"if (nearer(i, hit)) {{ i = process_portal_intersection(i, hit, is_inside_{pos}(r.o + r.d * hit.t, hit.u, hit.v, "
                                f"is_collinear(hit.n, normal), {_b(first)}), {material}); }}\n\n")
        else:
            if sides is None:
                x = mname(o["m0"])
                body.append(f"transformed_ray = transform({x}_mat_inv, r);\nlen = length(transformed_ray.d);\ntransformed_ray = normalize_ray(transformed_ray);")
                body.append(f"ihit = intersect_{pos}(transformed_ray);\nihit.hit.t /= len;\n")
                body.append(f"if (nearer(i, ihit)) {{ i = ihit; i.hit.n = normalize(adjugate({x}_mat) * i.hit.n); }}\n\n")
            else:
                for x, first, material in sides:
                    body.append(f"transformed_ray = transform({x}_mat_inv, r);\nlen = length(transformed_ray.d);\ntransformed_ray = normalize_ray(transformed_ray);")
                    body.append(f"ihit = intersect_{pos}(transformed_ray, {_b(first)});\nihit.hit.t /= len;\n")
                    body.append(f"if (nearer(i, ihit) && ihit.material != NOT_INSIDE) {{ if (ihit.material == TELEPORT) {{ ihit.material = {material}; }} "
                                f"if (ihit.material == TELEPORT_SUBSPACE) {{ ihit.material = {material}; ihit.in_subspace = true; }} i = ihit; "
                                f"i.hit.n = normalize(adjugate({x}_mat) * i.hit.n); }}\n\n")
        if guard:
            body.append("}")
        body.append("\n")
    slots["intersections"] = "".join(body)
    # intersection materials (scene.rs:1011-1035)
    slots["intersection_material_functions"] = "".join(
        f"SceneIntersectionWithMaterial intersect_material_{k}(Ray r) {{\n{code}\n}}\n" for k, (_, code) in enumerate(scene.intersection_materials))
    slots["intersection_material_processing"] = "".join(
        f"hit = intersect_material_{k}(r);\nif (nearer(result.scene.hit, hit.scene.hit)) {{ result = hit; }}\n\n" for k in range(len(scene.intersection_materials)))
    # library (scene.rs:1037-1044): the snippets back to back, nothing between them
    slots["library"] = "".join(code for _, code in scene.library)
    # skybox (scene.rs:1052-1063)
    if scene.skybox is not None:
        slots["skybox_processing"] = ("vec4 rd2 = _camera_mul_inv * r.d;" "float u = atan(rd2.z, rd2.x);" "float v = atan(sqrt(rd2.x * rd2.x + rd2.z * rd2.z), rd2.y);"
                                      f"vec3 not_found_color = sqrvec(texture({scene.skybox}_tex, vec2((u/PI+1.)/2., v/PI)).rgb);")
    else:
        slots["skybox_processing"] = "vec3 not_found_color = color(0.6, 0.6, 0.6);"
    return slots


def apply_template(template: str, storages: dict) -> str:
    """src/code_generation.rs:82-98: split at `//%`, odd pieces are slot names."""
    storages = dict(storages)
    out = []
    for pos, piece in enumerate(template.split("//%")):
        out.append(storages.pop(piece) if pos % 2 == 1 else piece)
    return "".join(out)


NATIVE_DATA = dict(for_prefer_variable=True, disable_antialiasing=False, disable_anaglyph=True, disable_camera_teleportation=False,
                   use_300_version=True)  # src/main.rs:935-941 on a non-wasm target


def filter_tagged_lines(text: str, data: dict) -> str:
    """src/gui/scene.rs:1065-1107 (runs over the assembled text, user code included)."""
    out = []
    for line in text.split("\n"):
        skip = (("!FOR_NUMBER!" in line and data["for_prefer_variable"]) or ("!FOR_VARIABLE!" in line and not data["for_prefer_variable"])
                or ("!ANTIALIASING!" in line and data["disable_antialiasing"]) or ("!ANAGLYPH!" in line and data["disable_anaglyph"])
                or ("!CAMERA_TELEPORTATION!" in line and data["disable_camera_teleportation"]) or ("!GLSL100!" in line and data["use_300_version"])
                or ("!GLSL300!" in line and not data["use_300_version"]))
        out.append("" if skip else line)
    return "\n".join(out).strip()


def assemble(scene, uniform_types: dict, data: dict | None = None) -> str:
    """The fragment shader source the reference hands to the GL driver for this scene."""
    texts = reference_texts()
    slots = emit_slots(scene, uniform_types)
    slots["predefined_library"] = texts["library.glsl"]
    return filter_tagged_lines(apply_template(texts["frag.glsl"], slots), data or NATIVE_DATA)


# =============================================================================================
# a preprocessor for what the text uses: #version, #ifdef GL_ES / #endif, object-like #define
# =============================================================================================
def preprocess(text: str):
    """-> (text without directives, {macro: [tokens]}).  `precision` statements carry no numerics here."""
    macros, out = {}, []
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#"):
            parts = s[1:].split(None, 2)
            if parts and parts[0] == "define":
                if "(" in parts[1]:
                    raise GlslError(f"function-like macro not supported: {s}")
                body = parts[2] if len(parts) > 2 else ""
                macros[parts[1]] = [t for t in tokenize(body) if t[0] != "eof"]
            elif parts and parts[0] in ("version", "ifdef", "endif", "extension"):
                pass  # GL_ES is defined under `#version 300 es`: the guarded line is kept
            else:
                raise GlslError(f"unsupported directive: {s}")
            out.append("")
        elif s.startswith("precision ") and s.endswith(";"):
            out.append("")
        else:
            out.append(line)
    return "\n".join(out), macros


def expand_macros(tokens, macros, depth=0):
    if depth > 8:
        raise GlslError("macro expansion too deep")
    out = []
    for tok in tokens:
        if tok[0] == "id" and tok[1] in macros:
            out.extend(expand_macros([(k, t, tok[2]) for k, t, _ in macros[tok[1]]], macros, depth + 1))
        else:
            out.append(tok)
    return out


# =============================================================================================
# the tracer: Oracle's scene / uniform / camera plumbing, the reference's text for everything else
# =============================================================================================
class ReferenceShader(Oracle):
    """Same interface as `Oracle` (options, overrides, camera, render, shade_pixels,
    teleport_external_ray); every shader-side operation comes from the reference's text."""

    def __init__(self, scene_path: str, asset_root: str | None = None):
        super().__init__(scene_path, asset_root)
        self.data = dict(NATIVE_DATA)
        self.source = None

    def build(self, width, height):
        from PIL import Image

        from .glsl_interp import Parser

        self.data["disable_anaglyph"] = not self.anaglyph_compiled_in
        uniforms = self._uniform_values(width, height)
        self.uniforms = uniforms
        self.source = assemble(self.scene, {k: V.type_of(v) for k, v in uniforms.items()}, self.data)
        text, macros = preprocess(self.source)
        prog = Interp(1)
        prog.globals.update(uniforms)
        for name, path in self.scene.textures:
            full = os.path.join(self.asset_root, path)
            prog.globals[name + "_tex"] = Sampler(np.array(Image.open(full).convert("RGBA"))) if os.path.exists(full) else None
        for name in self.scene.videos:
            prog.globals.setdefault(name + "_tex", None)
        for k in ("_teleport_external_ray",):
            prog.globals.setdefault(k, I32(0))
        for k in ("_external_ray_a", "_external_ray_b"):
            prog.globals.setdefault(k, Vec([F32(0)] * 3))
        prog.globals["uv"] = Vec([F32(0), F32(0)])
        prog.globals["uv_screen"] = Vec([F32(0), F32(0)])
        p = Parser("", ())
        p.t = expand_macros(tokenize(text), macros)
        for item in p.parse_unit():
            prog.declare(item)
            if item[0] == "struct":
                p.types.add(item[1])
        self._program = prog
        self.material_ids = self.scene.material_ids()

    # ---- void main(), mode 0 (frag.glsl:518-527,550-551) ------------------------------------------------
    def shade_pixels(self, width, height, px, py):
        self.build(width, height)
        M.reset_stats()
        px, py = np.atleast_1d(px), np.atleast_1d(py)
        n = len(px)
        res = self.uniforms["_resolution"]
        # the vertex stage, evaluated at the pixel centre (src/gui/scene.rs:1688-1693)
        position = Vec([M.add(M.f32(px), F32(0.5)), M.add(M.f32(py), F32(0.5))])
        coef = M.fmin(res.c[0], res.c[1])
        uv_screen = V.binop("*", V.binop("/", V.binop("-", position, V.binop("/", res, F32(2.0))), coef), F32(2.0))
        it = Interp(n, self._program)
        it.globals["uv"] = V.expand(position, n)
        it.globals["uv_screen"] = V.expand(uv_screen, n)
        it.globals["_teleport_external_ray"] = I32(0)
        it.globals["FragColor"] = Vec([np.zeros(n, F32)] * 4)
        it.run_function("main", [])
        frag = V.expand(it.globals["FragColor"], n)
        rgba = np.stack([np.asarray(c, F32) for c in frag.c], axis=1)
        self.stats = dict(M.STATS)
        return dict(rgba32f=rgba, rgba8=to_rgba8(rgba), segments=np.zeros(n, np.int64))

    # ---- the reference's function `teleport_external_ray` (frag.glsl:209-257), host side main.rs:1361-1409 ----
    def teleport_external_ray(self, a, b):
        self.build(0, 0)
        g = self._program.globals
        if "teleport_light_u" in g:
            g["teleport_light_u"] = I32(1)  # src/main.rs:1367
        a32, b32 = np.asarray(a, np.float64).astype(F32), np.asarray(b, np.float64).astype(F32)
        it = Interp(1, self._program)
        g["_external_ray_a"], g["_external_ray_b"] = Vec(a32), Vec(b32)
        in_sub = np.asarray(int(self.uniforms["_camera_in_subspace"]) == 1)
        ray = it.construct("Ray", [V.make_vec(4, [Vec(a32), F32(1.0)]), V.make_vec(4, [V.binop("-", Vec(b32), Vec(a32)), F32(0.0)]), F32(1.0), in_sub], 0)
        t = V.expand(it.run_function("teleport_external_ray", [V.expand(ray, 1)]), 1)
        pos = np.array([np.asarray(c).reshape(-1)[0] for c in t.f["pos"].c], F32)
        hit = bool(np.asarray(t.f["encounter_object"]).reshape(-1)[0])
        sub = bool(np.asarray(t.f["change_subspace"]).reshape(-1)[0])
        if pos[0] == 0 and pos[1] == 0 and pos[2] == 0:  # the host reads "no teleport" off a zero position (main.rs:1399-1405)
            return None, hit, sub
        return pos, hit, sub

    def teleport_external_ray_through_framebuffer(self, a, b):
        """The same query the way the reference's host really asks it (src/main.rs:1361-1409): `main()` with
        `_teleport_external_ray = 1` over a 2x3 RGBA8 target, `encode_float` bytes decoded with `f32::from_le_bytes`."""
        self.build(0, 0)
        g = self._program.globals
        if "teleport_light_u" in g:
            g["teleport_light_u"] = I32(1)
        a32, b32 = np.asarray(a, np.float64).astype(F32), np.asarray(b, np.float64).astype(F32)
        g["_external_ray_a"], g["_external_ray_b"] = Vec(a32), Vec(b32)
        ys, xs = np.meshgrid(np.arange(3), np.arange(2), indexing="ij")
        n = 6
        it = Interp(n, self._program)
        it.globals["uv"] = Vec([M.add(M.f32(xs.ravel()), F32(0.5)), M.add(M.f32(ys.ravel()), F32(0.5))])
        it.globals["uv_screen"] = Vec([np.zeros(n, F32), np.zeros(n, F32)])
        it.globals["_teleport_external_ray"] = I32(1)
        it.globals["FragColor"] = Vec([np.zeros(n, F32)] * 4)
        try:
            it.run_function("main", [])
        finally:
            it.globals["_teleport_external_ray"] = I32(0)
        frag = V.expand(it.globals["FragColor"], n)
        arr = to_rgba8(np.stack([np.asarray(c, F32) for c in frag.c], axis=1)).reshape(-1)
        hit = bool(arr[5] == 255 or arr[13] == 255 or arr[21] == 255)
        sub = bool(arr[6] == 255 or arr[14] == 255 or arr[22] == 255)
        pos = np.array([np.frombuffer(bytes([arr[k], arr[k + 1], arr[k + 2], arr[k + 4]]), "<f4")[0] for k in (0, 8, 16)], F32)
        if pos[0] == 0 and pos[1] == 0 and pos[2] == 0:
            return None, hit, sub
        return pos, hit, sub

    # ---- any function of the assembled unit, for function-level parity tests ---------------------------------
    def call(self, name, args, n):
        it = Interp(n, self._program)
        return it.run_function(name, [V.expand(a, n) for a in args])


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 1 and sys.argv[1] == "build":
        print(build_artifact())
    else:
        o = ReferenceShader(sys.argv[1])
        o.options["render_depth"] = int(sys.argv[4]) if len(sys.argv) > 4 else 8
        w, h = int(sys.argv[2]), int(sys.argv[3])
        o.build(w, h)
        print(o.source)
