"""Differential fuzz of the GLSL legs: scene snippets go to the GPU as C++ through the product's translator
(glsl_translate.cpp) + device prelude (ptl_glsl.h), and to the oracle through its own GLSL interpreter
(oracle/glsl_interp.py) + numpy contract (oracle/glsl_math.py).  Random typed expressions over the builtins, swizzles,
constructors and matrix products the corpus uses are written into ONE material snippet (one band of the wall per
expression), compiled once for the host, rendered, and compared with the oracle bit for bit."""
import random

import numpy as np
import pytest

N_EXPR = 48


@pytest.fixture(autouse=True)
def _scratch_build_dir(tmp_path_factory, monkeypatch):
    """one-off host builds: keep them out of oracle/_build (which travels to the GPU box)"""
    from oracle import host_build as hb

    monkeypatch.setattr(hb, "BUILD_DIR", str(tmp_path_factory.getbasetemp() / "fuzz_build"))


class Gen:
    """Typed random GLSL expressions: gen(t, depth) -> text of type t in {"float", "vec2", "vec3", "vec4", "bool"}."""

    def __init__(self, seed, uniforms=False):
        self.r = random.Random(seed)
        # with `uniforms`: leaves that depend on run-time uniforms alone (scene uniforms, a portal-style matrix pair, locals initialised
        # from them, a loop-carried chain `cb`) next to the varying ones -- food for the uniform-work hoister (glsl_hoist.h)
        self.uniforms = uniforms

    def lit(self):
        return self.r.choice(["0.0", "1.0", "0.5", "2.0", "-1.5", "0.25", "3.0", "1e-3", "7.5", "0.1", "-0.3", "10.", ".75"])

    def gen(self, t, d):
        r = self.r
        if t == "bool":
            a, b = self.gen("float", d - 1), self.gen("float", d - 1)
            return r.choice([f"({a} < {b})", f"({a} >= {b})", f"(({a} > {b}) && ({b} < 0.7))", f"(!({a} <= {b}) || ({a} > 0.2))"])
        if t != "float":
            n = int(t[3])
            if d <= 0 or r.random() < 0.25:
                base = {2: ["vec2(x, y)", "vec2(y, 0.5)", "p.xy", "p.zx"], 3: ["p", "vec3(x, y, 1.0)", "vec3(y + 0.0)", "p.zyx", "q.xyz"], 4: ["q", "vec4(p, 1.0)", "vec4(x, y, y, x)", "q.wzyx"]}[n]
                if self.uniforms and r.random() < 0.5:
                    base = {2: ["vec2(ua_u, ub_u)", "un.xy", "cb.zw", "vec2(uk, 0.3)"], 3: ["un", "vec3(ua_u, 0.4, ub_u)", "get_normal(tilt_mat)", "cb.xyz", "(tilt_mat * vec4(un, 0.0)).xyz"],
                            4: ["cb", "vec4(un, ua_u)", "(tilt_mat_inv * vec4(ub_u, 0.2, ua_u, 1.0))", "(tilt_mat * (back_mat_inv * cb))"]}[n]
                return r.choice(base)
            k = r.randrange(11)
            a, b = self.gen(t, d - 1), self.gen(t, d - 1)
            f = self.gen("float", d - 1)
            if k == 0:
                return f"({a} + {b})"
            if k == 1:
                return f"({a} * {b})"
            if k == 2:
                return f"({a} - {b} * {f})"
            if k == 3:
                return f"({a} / ({f} + 2.5))"
            if k == 4:
                return r.choice([f"sin({a})", f"cos({a})", f"abs({a})", f"fract({a})", f"floor({a})", f"sqrt(abs({a}))", f"exp({a} * 0.3)", f"sign({a})"])
            if k == 5:
                return r.choice([f"min({a}, {b})", f"max({a}, {f})", f"mod({a}, {b} + 1.5)", f"mod({a}, 0.7)", f"pow(abs({a}), {b})", f"step({a}, {b})", f"step(0.3, {a})"])
            if k == 6:
                return r.choice([f"mix({a}, {b}, {f})", f"mix({a}, {b}, {self.gen(t, d - 1)})", f"clamp({a}, -0.5, 0.8)", f"smoothstep(-0.5, 0.9, {a})", f"clamp({a}, {b}, {b} + 1.0)", f"smoothstep({b}, {b} + 1.0, {a})"])
            if k == 7:
                return f"normalize({a} + {('vec%d' % n)}(0.3))"
            if k == 8 and n == 3:
                return r.choice([f"cross({a}, {b})", f"(mat3({a}, {b}, vec3(0.2, 0.4, 1.0)) * p)", f"reflect({a}, normalize({b} + vec3(0.1, 0.7, 0.2)))"])
            if k == 8 and n == 2:
                return f"(mat2(x, y, 0.5, 1.0) * {a})"
            if k == 8 and n == 4:
                return f"(mat4({a}, {b}, q, vec4(0.0, 0.0, 0.0, 1.0)) * vec4(p, 1.0))"
            if k == 9:
                return f"({self.gen('bool', d - 1)} ? {a} : {b})"
            return f"(-({a}))"
        # float
        if d <= 0 or r.random() < 0.2:
            if self.uniforms and r.random() < 0.5:
                return r.choice(["ua_u", "ub_u", "uk", "un.y", "cb.x", "size_u"])
            return r.choice(["x", "y", "p.z", "q.w", self.lit(), self.lit()])
        k = r.randrange(14)
        a, b = self.gen("float", d - 1), self.gen("float", d - 1)
        vt = r.choice(["vec2", "vec3", "vec4"])
        if k == 0:
            return f"({a} {r.choice('+-*')} {b})"
        if k == 1:
            return f"({a} / ({b} * {b} + 0.5))"
        if k == 2:
            return r.choice([f"sin({a})", f"cos({a})", f"tan({a} * 0.4)", f"asin(clamp({a}, -1.0, 1.0))", f"acos(clamp({a}, -1.0, 1.0))", f"atan({a})", f"atan({a}, {b})"])
        if k == 3:
            return r.choice([f"exp({a} * 0.5)", f"log(abs({a}) + 0.1)", f"exp2({a})", f"log2(abs({a}) + 0.01)", f"sqrt(abs({a}))", f"inversesqrt(abs({a}) + 0.2)", f"pow(abs({a}) + 0.1, {b})"])
        if k == 4:
            return r.choice([f"abs({a})", f"floor({a} * 3.0)", f"ceil({a})", f"fract({a} * 2.5)", f"sign({a})", f"radians({a} * 90.0)", f"degrees({a})"])
        if k == 5:
            return r.choice([f"min({a}, {b})", f"max({a}, {b})", f"mod({a}, {b} * {b} + 0.3)", f"step({a}, {b})", f"clamp({a}, -0.25, {b} + 1.0)", f"mix({a}, {b}, 0.3)", f"smoothstep({a}, {a} + 1.0, {b})"])
        if k == 6:
            return f"length({self.gen(vt, d - 1)})"
        if k == 7:
            return f"dot({self.gen(vt, d - 1)}, {self.gen(vt, d - 1)})"
        if k == 8:
            return f"distance({self.gen(vt, d - 1)}, {self.gen(vt, d - 1)})"
        if k == 9:
            return f"{self.gen(vt, d - 1)}.{r.choice('xy')}"
        if k == 10:
            return f"({self.gen('bool', d - 1)} ? {a} : {b})"
        if k == 11:
            return f"float(int({a} * 4.0))"
        if k == 12:
            return f"(-({a}))"
        return f"({self.gen('vec3', d - 1)}).{r.choice(['x', 'y', 'z', 'r', 'b'])}"


def fuzz_scene_with_uniforms(seed):
    """Like fuzz_scene, but the expressions also draw on run-time uniforms, on locals that are uniform values, and on a loop-carried
    uniform chain (tabulated by the hoister); the bands are evaluated inside the loop, before and behind the chain's update."""
    from tests import synthetic

    g = Gen(seed, uniforms=True)
    n_loop = 3
    exprs = [g.gen("vec3", 3) for _ in range(N_EXPR)]
    body = ["float x = hit.u;", "float y = hit.v;", "vec3 p = vec3(x * 1.3 - 0.2, y + 0.35, x * y + 0.6);", "vec4 q = vec4(y, -x, 0.4, x - y);",
            f"int band = int(floor((x * 0.5 + 0.5) * {N_EXPR}.0));", "vec3 c = vec3(0.0);",
            "float uk = ua_u * ub_u + sqrt(abs(ua_u)) / (ub_u + 2.0);", "vec3 un = normalize(get_normal(tilt_mat) * ua_u + vec3(0.1, ub_u, 0.3));",
            "vec4 cb = tilt_mat * vec4(0.3, ub_u, 1.0, 0.0);",
            f"for (int it = 0; it < reps_u; it++) {{"]
    for k, e in enumerate(exprs):
        if k == N_EXPR // 2:
            body.append("cb = tilt_mat * (back_mat_inv * cb);")
        body.append(f"if (band == {k}) {{ c += {e}; }}")
    body.append("}")
    body.append("return material_simple(hit, r, abs(c) * 0.125, 0.0, false, 1.0, 0.0);")
    code = "\n".join(body)
    mat = f'(name: "fuzz", data: Complex(code: (("{code}")))),'
    matrices = ('(name: "tilt", data: Simple(offset: (0.3, -0.2, 0.5), scale: 1.25, rotate: (0.3, 0.9, -0.4), mirror: (false, false, false))),'
                '(name: "back", data: Simple(offset: (-0.1, 0.4, 0.2), scale: 0.8, rotate: (-0.5, 0.2, 0.7), mirror: (false, true, false))),')
    text = synthetic.wall_scene(r=1.0, size=1.0, extra_materials=mat, extra_matrices=matrices).replace("return wall_M; }", "return fuzz_M; }")
    text = text.replace('uniforms: ([', 'uniforms: ([ (name: "ua", data: Float((min: None, max: None, value: 0.7))), (name: "ub", data: Float((min: None, max: None, value: -0.35))), '
                        f'(name: "reps", data: Int((min: None, max: None, value: {n_loop}))), ')
    return text, exprs


def fuzz_scene(seed):
    from tests import synthetic

    g = Gen(seed)
    exprs = [g.gen("vec3", 4) for _ in range(N_EXPR)]
    body = ["float x = hit.u;", "float y = hit.v;", "vec3 p = vec3(x * 1.3 - 0.2, y + 0.35, x * y + 0.6);", "vec4 q = vec4(y, -x, 0.4, x - y);",
            f"int band = int(floor((x * 0.5 + 0.5) * {N_EXPR}.0));", "vec3 c = vec3(0.0);"]
    for k, e in enumerate(exprs):
        body.append(("if" if k == 0 else "else if") + f" (band == {k}) {{ c = {e}; }}")
    body.append("return material_simple(hit, r, abs(c) * 0.25, 0.0, false, 1.0, 0.0);")
    code = "\n".join(body)
    mat = f'(name: "fuzz", data: Complex(code: (("{code}")))),'
    text = synthetic.wall_scene(r=1.0, size=1.0, extra_materials=mat).replace("return wall_M; }", "return fuzz_M; }")
    return text, exprs


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_glsl_expressions_product_equals_oracle(pa, tmp_path, seed):
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    text, exprs = fuzz_scene(seed)
    path = tmp_path / "fuzz.ron"
    path.write_text(text)
    w, h = 4 * N_EXPR, 12
    scene = pa.Scene.from_file(str(path))
    r = pa.SceneRenderer(scene, device=-1)
    r.set_option("render_depth", 2)
    r.set_option("view_angle", 1.5)
    hk = hb.HostKernel(scene.generate_source(0), *scene.uniform_layout(), opt="-O0")
    for name, typ, _ in scene.uniform_layout()[0]:
        if typ != pa.PTL_SAMPLER:
            hk.set_uniform(name, r.uniform_value(name, w, h))
    got = hk.render(w, h)["rgba32f"]
    o = Oracle(str(path))
    o.options.update(render_depth=2, view_angle=1.5)
    want = o.render(w, h)["rgba32f"]
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        ys, xs = np.nonzero(~same.all(axis=2))
        bands = sorted({int(x * N_EXPR / w) for x in xs})
        raise AssertionError(f"seed {seed}: {len(xs)} pixels differ, around bands {bands[:6]}: " + " | ".join(exprs[b] for b in bands[:3] if b < len(exprs)))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > N_EXPR       # the bands really show different values


@pytest.mark.parametrize("seed,flags", [(11, 0), (12, 0), (13, 0), (14, 0), (15, 0), (16, 0), (11, 1), (13, 1), (16, 1)],
                         ids=["11", "12", "13", "14", "15", "16", "11-masked", "13-masked", "16-masked"])
def test_random_glsl_expressions_over_uniforms_survive_the_hoister(pa, tmp_path, seed, flags):
    """The same differential test with uniform leaves, uniform locals and a tabulated loop-carried chain: the hoister moves a good part
    of every snippet into the prologue (the source must show it), and the host build -- which runs derive() -- still equals the oracle,
    which evaluates the snippet as written, bit for bit."""
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    text, exprs = fuzz_scene_with_uniforms(seed)
    path = tmp_path / "fuzz.ron"
    path.write_text(text)
    w, h = 4 * N_EXPR, 12
    scene = pa.Scene.from_file(str(path))
    # flags 1 (Bool / Int baked): the zero patterns of tilt_mat / back_mat are compiled in and the random products over them are rewritten
    # into their masked forms by shape (codegen.cpp::apply_zero_masks) -- in the snippet and in what the hoister moved to the prologue
    source = scene.generate_source(flags)
    assert source.count("PTL_U.ptl_hv") >= 10 and "ptl_tab_ok_" in source and "for (int ptl_k = 0;" in source
    if flags:
        assert source.count("ptl_mul_m<PTL_MASK_") >= 3
    r = pa.SceneRenderer(scene, device=-1, flags=flags)
    r.set_option("render_depth", 2)
    r.set_option("view_angle", 1.5)
    hk = hb.HostKernel(source, *scene.uniform_layout(), opt="-O0")
    for name, typ, _ in scene.uniform_layout()[0]:
        if typ != pa.PTL_SAMPLER:
            hk.set_uniform(name, r.uniform_value(name, w, h))
    got = hk.render(w, h)["rgba32f"]
    o = Oracle(str(path))
    o.options.update(render_depth=2, view_angle=1.5)
    want = o.render(w, h)["rgba32f"]
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        ys, xs = np.nonzero(~same.all(axis=2))
        bands = sorted({int(x * N_EXPR / w) for x in xs})
        raise AssertionError(f"seed {seed}: {len(xs)} pixels differ, around bands {bands[:6]}: " + " | ".join(exprs[b] for b in bands[:3] if b < len(exprs)))
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > N_EXPR // 2


@pytest.mark.parametrize("seed", [1, 2])
def test_hoister_survives_mutated_corpus_snippets(pa, seed):
    """The hoister runs on every snippet of every scene a user loads: whatever the text (here: the reference's ~1 600 snippets with
    random token deletions, duplications, swaps, truncations and stray punctuation), it must come back with a string that has the
    same number of lines -- never crash, never throw -- and the translator must accept or refuse that string in an orderly way."""
    import glob
    import os
    import re

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus", "scenes")
    texts = []
    for f in sorted(glob.glob(os.path.join(root, "*.ron"))):
        for m in re.finditer(r'\(\("(.*?)"\)\)', open(f).read(), re.S):
            if len(m.group(1)) > 40:
                texts.append(m.group(1).replace('\\"', '"'))
    assert len(texts) > 1000
    uniforms = {}
    for t in texts:
        for w in set(re.findall(r"\b\w+_mat(?:_inv|_teleport)?\b", t)):
            uniforms.setdefault(w, "mat4")
        for w in set(re.findall(r"\b\w+_u\b", t)):
            uniforms.setdefault(w, "float")
    rng = random.Random(seed)
    tok = re.compile(r"\s+|\w+|[^\w\s]")
    stray = ["(", ")", "{", "}", ";", ",", "for", "if", "else", "=", "++", "?", ":", "[", "]", "return", "continue", "break", ".", "*"]
    hoisted = 0
    for _ in range(400):
        t = rng.choice(texts)
        toks = tok.findall(t)
        for _ in range(rng.randrange(0, 6)):
            if not toks:
                break
            i, op = rng.randrange(len(toks)), rng.randrange(6)
            if op == 0:
                del toks[i]
            elif op == 1:
                toks.insert(i, toks[rng.randrange(len(toks))])
            elif op == 2:
                j = rng.randrange(len(toks))
                toks[i], toks[j] = toks[j], toks[i]
            elif op == 3:
                toks = toks[:i]
            elif op == 4:
                toks.insert(i, rng.choice(stray))
            else:
                toks[i] = rng.choice(list(uniforms))
        code = "".join(toks)
        out, _ = pa.hoist_glsl(code, uniforms, body_only=not re.search(r"^\s*\w+\s+\w+\s*\(", t), params=["r", "pos", "x", "y", "back", "first", "hit", "i"])
        assert out.count("\n") == code.count("\n")
        hoisted += "ptl_hv" in out
        try:
            pa.translate_glsl(out)
        except pa.PortalError:
            pass
    assert hoisted > 0
