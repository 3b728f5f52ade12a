"""oracle/portal_oracle.py -- CPU oracle for the per-pixel recursive portal ray tracer.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import anything under oracle/; the product (portal_amd/) never does.

PARITY: PINNED TO THE REFERENCE'S SOURCE TEXT; BUILTIN PRECISION BY CONTRACT.  The reference ships no CPU
tracer, no golden image and no known-answer vector for this path and
cannot be built or run here (no Rust toolchain, no GL context), so this file is a hand restatement.  Since round 3
it is checked against the reference's OWN definition of the arithmetic: oracle/reference_shader.py executes
/root/reference/src/library.glsl + src/frag.glsl (slots filled as src/gui/scene.rs:693-1075 fills them) with
oracle/glsl_interp.py, and tests/test_reference_text.py requires every function below == that text on 16 384
seeded + special lanes, whole frames of the five configs and six camera / mode variants == that text, the
camera-teleport query == that text (incl. its encode_float -> RGBA8 -> from_le_bytes route), and all 82 corpus
scenes == that text -- bit for bit.  What stays unpinned is what GLSL ES 3.00 itself leaves open: the precision a
GL driver gives `/`, sqrt, sin ... (oracle/glsl_math.py fixes one contract, shared by both oracles and the kernel),
the rasteriser's varying interpolation and RGBA8 rounding (third-party), plus glam / fasteval on the host side
(restated from their published algorithms).  Further pins: the hand-derived known answers of tests/test_oracle_kat.py
and, qualitatively, the captures of the real program (tests/test_reference_screenshot.py).

What it restates, in numpy over "lanes" (one lane = one pixel sample), independently of the
product's C++ host code, of its GLSL->C++ translator and of its device prelude:
  src/library.glsl (all)            -> the `Natives` below (names/arguments are the interface
                                       scene snippets call, so they are kept)
  src/frag.glsl                     -> ray_tracing(), get_color2(), shade_pixels()
  src/gui/scene.rs:885-1035         -> scene_intersect(), material_process(),
                                       scene_intersect_material_process(): what the generated
                                       code computes, interpreted directly from the scene model
  src/gui/scene.rs:1674-1697        -> pixel centre -> uv_screen
  scene snippets (GLSL in .ron)     -> executed by oracle/glsl_interp.py
Arithmetic: binary32 under the numerics contract (oracle/glsl_math.py).
"""
from __future__ import annotations

import os

import numpy as np

from . import glsl_math as M
from . import glsl_values as V
from .glsl_interp import Interp
from .glsl_values import Mat, Sampler, Struct, Vec
from .scene_eval import OracleScene, builtin_uniforms, camera_matrix

F32, I32 = np.float32, np.int32

CUSTOM_MATERIAL, NOT_INSIDE, TELEPORT, TELEPORT_SUBSPACE = -1, 0, 1, 2
DEBUG_RED, DEBUG_GREEN, DEBUG_BLUE, USER_MATERIAL_OFFSET = 3, 4, 5, 10

STRUCTS = {  # src/library.glsl:40-46,127-133,297-301,405-409,595-598
    "Ray": [("vec4", "o"), ("vec4", "d"), ("float", "tmul"), ("bool", "in_subspace")],
    "SurfaceIntersection": [("bool", "hit"), ("float", "t"), ("float", "u"), ("float", "v"), ("vec3", "n")],
    "MaterialProcessing": [("bool", "is_final"), ("vec3", "mul_to_color"), ("Ray", "new_ray")],
    "SceneIntersection": [("int", "material"), ("SurfaceIntersection", "hit"), ("bool", "in_subspace")],
    "SceneIntersectionWithMaterial": [("SceneIntersection", "scene"), ("MaterialProcessing", "material")],
}


def fl(x):
    return F32(x)


def vec(*c):
    return Vec(c)


def add(a, b):
    return V.binop("+", a, b)


def sub(a, b):
    return V.binop("-", a, b)


def mul(a, b):
    return V.binop("*", a, b)


def div(a, b):
    return V.binop("/", a, b)


def xyz(v: Vec) -> Vec:
    return Vec(v.c[:3])


def mk(tname, **f):
    return Struct(tname, f)


def Ray(o, d, tmul, in_subspace):
    return mk("Ray", o=o, d=d, tmul=M.f32(tmul), in_subspace=np.asarray(in_subspace, bool))


def Surf(hit, t, u, v, n):
    return mk("SurfaceIntersection", hit=np.asarray(hit, bool), t=M.f32(t), u=M.f32(u), v=M.f32(v), n=n)


def SceneI(material, hit, in_subspace):
    return mk("SceneIntersection", material=np.asarray(material, I32), hit=hit, in_subspace=np.asarray(in_subspace, bool))


def MatProc(is_final, mul_to_color, new_ray):
    return mk("MaterialProcessing", is_final=np.asarray(is_final, bool), mul_to_color=mul_to_color, new_ray=new_ray)


RAY_NONE = Ray(vec(0, 0, 0, 0), vec(0, 0, 0, 0), 0.0, False)               # library.glsl:53
INTERSECTION_NONE = Surf(False, fl(1e10), 0.0, 0.0, vec(0, 0, 0))            # library.glsl:136
SCENE_INTERSECTION_NONE = SceneI(0, INTERSECTION_NONE, False)                # library.glsl:411


# =============================================================================================
# src/library.glsl restated on the value model.  Each function is pure and evaluates all
# lanes; data-dependent branches become selects.
# =============================================================================================
class Natives:
    def __init__(self, uniforms):
        self.u = uniforms  # name -> value (0-d arrays / Vec / Mat)

    # -- library.glsl:19-34
    @staticmethod
    def between(a, x, b):
        return M.le(a, x) & M.le(x, b)

    @staticmethod
    def sqr(a):
        return M.mul(a, a)

    def sqrvec(self, v):
        return vec(self.sqr(v.c[0]), self.sqr(v.c[1]), self.sqr(v.c[2]))

    # -- library.glsl:48-51
    @staticmethod
    def offset_ray(r, t):
        return r.with_field("o", add(r.f["o"], mul(r.f["d"], t)))

    # -- library.glsl:56-62
    @staticmethod
    def normalize_normal(normal, direction):
        normal = V.normalize(normal)
        flip = M.gt(V.dot(normal, direction), fl(0))
        return V.select(flip, mul(normal, fl(-1.0)), normal)

    # -- library.glsl:65-67
    @staticmethod
    def is_collinear(a, b):
        q = M.div(V.dot(a, b), M.mul(V.length(a), V.length(b)))
        return M.lt(M.absf(M.sub(q, fl(1.0))), M.lit("0.01"))

    # -- library.glsl:70-72
    @staticmethod
    def my_reflect(direction, normal):
        t = div(mul(normal, V.dot(direction, normal)), V.dot(normal, normal))
        return sub(direction, mul(t, fl(2.0)))

    # -- library.glsl:75-92
    def my_refract(self, direction, normal, refractive_index):
        ri = M.f32(refractive_index)
        from_outside = M.gt(V.dot(normal, direction), fl(0))
        ri = V.select(from_outside, ri, M.div(fl(1.0), ri))
        normal = V.select(from_outside, V.neg(normal), normal)
        direction = V.normalize(direction)
        c = M.neg(V.dot(normal, direction))
        d = M.sub(fl(1.0), M.mul(M.mul(ri, ri), M.sub(fl(1.0), M.mul(c, c))))
        refr = add(mul(direction, ri), mul(normal, M.sub(M.mul(ri, c), M.sqrt(d))))
        return V.select(M.gt(d, fl(0)), refr, self.my_reflect(direction, normal))

    # -- library.glsl:95-120
    @staticmethod
    def transform(matrix, r):
        return Ray(mul(matrix, r.f["o"]), mul(matrix, r.f["d"]), r.f["tmul"], r.f["in_subspace"])

    @staticmethod
    def get_normal(matrix):
        return xyz(mul(matrix, vec(0.0, 0.0, 1.0, 0.0)))

    @staticmethod
    def normalize_ray(r):
        ln = V.length(r.f["d"])
        return r.with_field("d", div(r.f["d"], ln)).with_field("tmul", M.div(r.f["tmul"], ln))

    @staticmethod
    def adjugate(m):
        c0, c1, c2 = xyz(m.cols[0]), xyz(m.cols[1]), xyz(m.cols[2])
        return Mat([V.cross(c1, c2), V.cross(c2, c0), V.cross(c0, c1)])

    # -- library.glsl:138-162
    @staticmethod
    def plane_intersect_normalized(r):
        o, d = r.f["o"], r.f["d"]
        t = M.div(M.neg(o.c[2]), d.c[2])
        pos = add(o, mul(d, t))
        hit = Surf(True, t, pos.c[0], pos.c[1], vec(0.0, 0.0, 1.0))
        return V.select(M.lt(t, fl(0)), INTERSECTION_NONE, hit)

    def plane_intersect(self, r, plane_inv, normal):
        normal = self.normalize_normal(normal, xyz(r.f["d"]))
        r = self.transform(plane_inv, r)
        ln = V.length(r.f["d"])
        r = r.with_field("d", V.normalize(r.f["d"]))
        res = self.plane_intersect_normalized(r)
        hit = res.f["hit"]
        res = res.with_field("t", V.select(hit, M.div(res.f["t"], ln), res.f["t"]))
        return res.with_field("n", V.select(hit, normal, res.f["n"]))

    # -- library.glsl:169-288
    @staticmethod
    def color(r, g, b):
        return vec(M.mul(r, r), M.mul(g, g), M.mul(b, b))

    def color_normal(self, normal, direction):
        if int(self.u["_angle_color_disable"]) == 1:
            return fl(1.0)
        return M.absf(V.dot(V.normalize(xyz(direction)), V.normalize(normal)))

    def color_grid(self, start, uv):
        if int(self.u["_grid_disable"]) == 1:
            return start
        uv = V.map1(M.fract, mul(uv, fl(0.25)))
        sx, sy = M.step(uv.c[0], fl(0.5)), M.step(uv.c[1], fl(0.5))
        a = M.mix(M.lit("0.7"), M.lit("1.1"), sx)
        b = M.mix(M.lit("1.1"), M.lit("0.7"), sx)
        return mul(start, M.mix(a, b, sy))

    @staticmethod
    def circle_sdf(position):
        s = vec(fl(2.0), M.mul(M.sqrt(fl(3.0)), fl(2.0)))
        position = div(position, s)
        d1 = mul(sub(V.map1(M.fract, position), fl(0.5)), s)
        d2 = mul(sub(V.map1(M.fract, add(position, fl(0.5))), fl(0.5)), s)
        return M.sub(M.sqrt(M.fmin(V.dot(d1, d1), V.dot(d2, d2))), fl(1.0))

    def color_grid2(self, start, uv):
        d = self.circle_sdf(uv)
        val = V.select(M.lt(d, M.lit("-0.2")), M.lit("1.1"), M.lit("0.7"))
        return mul(start, val)

    def color_grid3(self, start, uv):
        if int(self.u["_grid_disable"]) == 1:
            return start
        uv = sub(V.map1(M.fract, mul(uv, fl(0.5))), vec(0.5, 0.5))
        dist = M.mul(M.fmax(M.absf(uv.c[0]), M.absf(uv.c[1])), fl(2.0))
        edge = V.select(M.gt(uv.c[0], uv.c[1]), mul(start, M.lit("0.7")), mul(start, M.lit("1.2")))
        inner = V.select(M.lt(dist, M.lit("0.94")), start, edge)
        return V.select(M.gt(dist, M.lit("0.985")), mul(start, M.lit("0.4")), inner)

    @staticmethod
    def color_add_weighted(a, b, coef):
        return add(mul(a, M.sub(fl(1.0), coef)), mul(b, coef))

    # -- library.glsl:303-384
    @staticmethod
    def material_empty():
        return MatProc(True, vec(0, 0, 0), RAY_NONE)

    @staticmethod
    def material_final(color):
        return MatProc(True, color, RAY_NONE)

    @staticmethod
    def material_next(mul_color, new_ray):
        return MatProc(False, mul_color, new_ray)

    def material_simple2(self, hit, r, color, normal_coef, grid, grid_scale, grid_coef, grid2, grid3):
        color = self.color_add_weighted(color, mul(color, self.color_normal(hit.f["n"], r.f["d"])), M.f32(normal_coef))
        grid, grid2, grid3 = (np.asarray(x, bool) for x in (grid, grid2, grid3))
        if grid.any():
            cell = mul(vec(hit.f["u"], hit.f["v"]), M.f32(grid_scale))
            pattern = V.select(grid3, self.color_grid3(color, cell), V.select(grid2, self.color_grid2(color, cell), self.color_grid(color, cell)))
            color = V.select(grid, self.color_add_weighted(color, pattern, M.f32(grid_coef)), color)
        return self.material_final(color)

    def material_simple(self, hit, r, color, normal_coef, grid, grid_scale, grid_coef):
        return self.material_simple2(hit, r, color, normal_coef, grid, grid_scale, grid_coef, False, False)

    def material_reflect(self, hit, r, add_to_color):
        d = Vec(list(self.my_reflect(xyz(r.f["d"]), hit.f["n"]).c) + [fl(0.0)])
        r = r.with_field("d", d)
        r = r.with_field("o", add(r.f["o"], mul(d, self.u["_offset_after_material"])))
        return self.material_next(add_to_color, r)

    def material_refract(self, hit, r, add_to_color, refractive_index):
        d = Vec(list(self.my_refract(xyz(r.f["d"]), hit.f["n"], refractive_index).c) + [fl(0.0)])
        r = r.with_field("d", d)
        r = r.with_field("o", add(r.f["o"], mul(d, self.u["_offset_after_material"])))
        return self.material_next(add_to_color, r)

    def material_teleport_transformed(self, r, n=None):
        r = r.with_field("o", add(r.f["o"], mul(r.f["d"], self.u["_offset_after_material"])))
        return self.material_next(vec(1.0, 1.0, 1.0), self.normalize_ray(r))

    def material_teleport(self, hit, r, teleport_matrix):
        return self.material_teleport_transformed(self.transform(teleport_matrix, r), hit.f["n"])

    def material_change_subspace(self, r):
        return self.material_next(vec(1.0, 1.0, 1.0), r.with_field("in_subspace", ~r.f["in_subspace"]))

    # -- library.glsl:413-423
    @staticmethod
    def nearer(result, current):
        if result.tname != "SurfaceIntersection":
            result = result.f["hit"]
        if current.tname != "SurfaceIntersection":
            current = current.f["hit"]
        return current.f["hit"] & M.gt(current.f["t"], fl(0)) & (~result.f["hit"] | (result.f["hit"] & M.lt(current.f["t"], result.f["t"])))

    # -- library.glsl:426-470 (capsule, after iq)
    @staticmethod
    def cap_normal(pos, a, b, radius):
        ba, pa_ = sub(b, a), sub(pos, a)
        h = M.clamp(M.div(V.dot(pa_, ba), V.dot(ba, ba)), fl(0.0), fl(1.0))
        return div(sub(pa_, mul(h, ba)), M.f32(radius))

    def cap(self, r, pa_, pb, radius):
        radius = M.f32(radius)
        ro, rd = xyz(r.f["o"]), xyz(r.f["d"])
        ba, oa = sub(pb, pa_), sub(ro, pa_)
        baba, bard, baoa, rdoa, oaoa = V.dot(ba, ba), V.dot(ba, rd), V.dot(ba, oa), V.dot(rd, oa), V.dot(oa, oa)
        a = M.sub(baba, M.mul(bard, bard))
        b = M.sub(M.mul(baba, rdoa), M.mul(baoa, bard))
        c = M.sub(M.sub(M.mul(baba, oaoa), M.mul(baoa, baoa)), M.mul(M.mul(radius, radius), baba))
        h = M.sub(M.mul(b, b), M.mul(a, c))
        t = M.div(M.sub(M.neg(b), M.sqrt(h)), a)
        y = M.add(baoa, M.mul(t, bard))
        body = M.ge(h, fl(0.0)) & M.gt(y, fl(0.0)) & M.lt(y, baba)
        body_hit = Surf(True, t, 0.0, 0.0, self.cap_normal(add(ro, mul(rd, t)), pa_, pb, radius))
        oc = V.select(M.le(y, fl(0.0)), oa, sub(ro, pb))
        b2 = V.dot(rd, oc)
        c2 = M.sub(V.dot(oc, oc), M.mul(radius, radius))
        h2 = M.sub(M.mul(b2, b2), c2)
        t2 = M.sub(M.neg(b2), M.sqrt(h2))
        caps = M.ge(h, fl(0.0)) & ~body & M.gt(h2, fl(0.0))
        caps_hit = Surf(True, t2, 0.0, 0.0, self.cap_normal(add(ro, mul(rd, t2)), pa_, pb, radius))
        return V.select(body, body_hit, V.select(caps, caps_hit, INTERSECTION_NONE))

    # -- library.glsl:473-504 (cylinder, after iq)
    def cylinder(self, r, pa_, pb, ra):
        ra = M.f32(ra)
        ro, rd = xyz(r.f["o"]), xyz(r.f["d"])
        ba, oc = sub(pb, pa_), sub(ro, pa_)
        baba, bard, baoc = V.dot(ba, ba), V.dot(ba, rd), V.dot(ba, oc)
        k2 = M.sub(baba, M.mul(bard, bard))
        k1 = M.sub(M.mul(baba, V.dot(oc, rd)), M.mul(baoc, bard))
        k0 = M.sub(M.sub(M.mul(baba, V.dot(oc, oc)), M.mul(baoc, baoc)), M.mul(M.mul(ra, ra), baba))
        h = M.sub(M.mul(k1, k1), M.mul(k2, k0))
        miss = M.lt(h, fl(0.0))
        hs = M.sqrt(h)

        def side(t):
            y = M.add(baoc, M.mul(t, bard))
            n = div(sub(add(oc, mul(t, rd)), div(mul(ba, y), baba)), ra)
            return M.gt(y, fl(0.0)) & M.lt(y, baba), Surf(True, t, 0.0, 0.0, n)

        ok_n, hit_n = side(M.div(M.sub(M.neg(k1), hs), k2))
        ok_f, hit_f = side(M.div(M.add(M.neg(k1), hs), k2))
        return V.select(miss, INTERSECTION_NONE, V.select(ok_n, hit_n, V.select(ok_f, hit_f, INTERSECTION_NONE)))

    # -- library.glsl:507-525
    def triangle(self, r, v0, v1, v2):
        ro, rd = xyz(r.f["o"]), xyz(r.f["d"])
        v1v0, v2v0, rov0 = sub(v1, v0), sub(v2, v0), sub(ro, v0)
        n = V.cross(v1v0, v2v0)
        q = V.cross(rov0, rd)
        d = M.div(fl(1.0), V.dot(rd, n))
        u = M.mul(d, V.dot(V.neg(q), v2v0))
        v = M.mul(d, V.dot(q, v1v0))
        t = M.mul(d, V.dot(V.neg(n), rov0))
        miss = M.lt(u, fl(0.0)) | M.lt(v, fl(0.0)) | M.gt(M.add(u, v), fl(1.0))
        hit = Surf(True, t, u, v, self.normalize_normal(V.cross(sub(v1, v0), sub(v2, v0)), rd))
        return V.select(miss, INTERSECTION_NONE, hit)

    # -- library.glsl:528-554
    def debug_intersect(self, r):
        i = SceneI(0, INTERSECTION_NONE, False)
        for axis, mat in ((vec(1.0, 0.0, 0.0), DEBUG_RED), (vec(0.0, 1.0, 0.0), DEBUG_GREEN), (vec(0.0, 0.0, 1.0), DEBUG_BLUE)):
            hit = self.cap(r, vec(0.0, 0.0, 0.0), axis, M.lit("0.03"))
            near = self.nearer(i, hit)
            i = V.select(near, i.with_field("material", I32(mat)).with_field("hit", hit), i)
        return i

    # -- library.glsl:560-589
    @staticmethod
    def process_plane_intersection(i, hit, inside):
        inside = np.asarray(inside, I32)
        take = (inside != NOT_INSIDE) & (inside != TELEPORT) & (inside != TELEPORT_SUBSPACE)
        return V.select(take, i.with_field("hit", hit).with_field("material", inside), i)

    @staticmethod
    def process_portal_intersection(i, hit, inside, teleport_material):
        inside = np.asarray(inside, I32)
        tm = np.asarray(teleport_material, I32)
        is_tp = (inside == TELEPORT) | (inside == TELEPORT_SUBSPACE)
        new = i.with_field("hit", hit).with_field("material", np.where(is_tp, tm, inside).astype(I32))
        new = new.with_field("in_subspace", np.where(inside == TELEPORT_SUBSPACE, True, i.f["in_subspace"]))
        return V.select(inside != NOT_INSIDE, new, i)


def _wrap(fn):
    return lambda interp, args, mask: fn(*args)


# =============================================================================================
# the tracer
# =============================================================================================
class Oracle:
    def __init__(self, scene_path: str, asset_root: str | None = None):
        self.scene = OracleScene(scene_path)
        self.asset_root = asset_root or os.path.dirname(os.path.dirname(os.path.abspath(scene_path)))
        self.options = dict(render_depth=100, aa_count=1, aa_start=0, view_angle=None, use_panini=False, panini_param=1.0)
        self.overrides = {}  # builtin uniform name -> value (e.g. _use_360_camera, _draw_depth_map, _draw_side_by_side, _camera_left_eye)
        self.camera = None
        self.anaglyph_compiled_in = False  # the reference's `disable_anaglyph` (default true: the mode does not exist in the shader)
        self._program = None
        self.stats = {}

    # ---- program = uniforms + natives + snippets, parsed once ---------------------------------
    def _uniform_values(self, width, height):
        # send_camera_object_matrix (src/main.rs:147,1530-1534): Matrix::Camera evaluates to the camera that draws
        cam = dict(self.scene.cam)
        cam.update(self.camera or {})
        self.scene.camera_object_matrix = camera_matrix(cam["look_at"], cam["alpha"], cam["beta"], cam["r"], cam.get("teleport_matrix"), cam.get("free_movement", False))
        vals = dict(self.scene.scene_uniform_values())
        vals.update(builtin_uniforms(self.scene, width, height, camera=self.camera, **self.options))
        vals.update(self.overrides)
        out = {}
        for name, v in vals.items():
            a = np.asarray(v)
            if a.shape == (16,):
                out[name] = Mat([Vec(a[4 * c:4 * c + 4]) for c in range(4)])
            elif a.shape == (2,):
                out[name] = Vec(a)
            else:
                out[name] = a[()]
        return out

    def build(self, width, height):
        from PIL import Image

        uniforms = self._uniform_values(width, height)
        prog = Interp(1)
        prog.structs.update(STRUCTS)
        prog.globals.update(uniforms)
        for name, path in self.scene.textures:
            full = os.path.join(self.asset_root, path)
            # a texture file that cannot be read leaves the sampler unbound (reference: texture_errors, src/main.rs:1082-1084)
            prog.globals[name + "_tex"] = Sampler(np.array(Image.open(full).convert("RGBA"))) if os.path.exists(full) else None
        for name in self.scene.videos:
            prog.globals.setdefault(name + "_tex", None)  # unbound sampler: texture() returns (0, 0, 0, 1)
        for name, mid in self.scene.material_ids().items():
            prog.globals[name] = I32(mid)
        consts = dict(CUSTOM_MATERIAL=CUSTOM_MATERIAL, NOT_INSIDE=NOT_INSIDE, TELEPORT=TELEPORT, TELEPORT_SUBSPACE=TELEPORT_SUBSPACE, DEBUG_RED=DEBUG_RED,
                      DEBUG_GREEN=DEBUG_GREEN, DEBUG_BLUE=DEBUG_BLUE, USER_MATERIAL_OFFSET=USER_MATERIAL_OFFSET)
        for k, v in consts.items():
            prog.globals[k] = I32(v)
        prog.globals["PI"] = M.acos(fl(-1.0))[()]                      # library.glsl:15
        prog.globals["PI2"] = M.div(M.acos(fl(-1.0)), fl(2.0))[()]     # library.glsl:16
        prog.globals["ray_none"] = RAY_NONE
        prog.globals["intersection_none"] = INTERSECTION_NONE
        prog.globals["scene_intersection_none"] = SCENE_INTERSECTION_NONE
        nat = Natives(uniforms)
        self.nat = nat
        for name in ("between", "sqr", "sqrvec", "offset_ray", "normalize_normal", "is_collinear", "my_reflect", "my_refract", "transform", "get_normal",
                     "normalize_ray", "adjugate", "plane_intersect_normalized", "plane_intersect", "color", "color_normal", "color_grid", "circle_sdf",
                     "color_grid2", "color_grid3", "color_add_weighted", "material_empty", "material_final", "material_next", "material_simple2",
                     "material_simple", "material_reflect", "material_refract", "material_teleport_transformed", "material_teleport",
                     "material_change_subspace", "nearer", "process_plane_intersection", "process_portal_intersection", "cap_normal", "cap", "cylinder",
                     "triangle", "debug_intersect"):
            prog.natives[name] = _wrap(getattr(nat, name))
        flags = dict(FOR_NUMBER=False, FOR_VARIABLE=True, ANTIALIASING=True, ANAGLYPH=False, CAMERA_TELEPORTATION=True, GLSL100=False, GLSL300=True)

        def filt(code):  # the tagged-line filter, src/gui/scene.rs:1065-1107 (native defaults main.rs:935-941)
            out = []
            for line in code.split("\n"):
                drop = any(("!" + tag + "!") in line and not keep for tag, keep in flags.items())
                out.append("" if drop else line)
            return "\n".join(out)

        for _, code in self.scene.library:
            prog.load_unit(filt(code))
        for pos, o in enumerate(self.scene.objects):
            if o["kind"] == "flat":
                params = [("vec4", "pos"), ("float", "x"), ("float", "y"), ("bool", "back")] + ([("bool", "first")] if o["portal"] else [])
                prog.define_function("int", f"is_inside_{pos}", params, filt(o["code"]))
            elif o["kind"] == "complex":
                params = [("Ray", "r")] + ([("bool", "first")] if o["portal"] else [])
                prog.define_function("SceneIntersection", f"intersect_{pos}", params, filt(o["code"]))
        for k, m in enumerate(self.scene.materials):
            if m["kind"] == "Complex":  # the snippet sees `hit`, `r`, `i` (scene.rs:736-778, frag.glsl:33-35)
                prog.define_function("MaterialProcessing", f"__material_{k}", [("SurfaceIntersection", "hit"), ("Ray", "r"), ("SceneIntersection", "i")], filt(m["code"]))
        for k, (_, code) in enumerate(self.scene.intersection_materials):
            prog.define_function("SceneIntersectionWithMaterial", f"intersect_material_{k}", [("Ray", "r")], filt(code))
        self._program = prog
        self.uniforms = uniforms
        self.material_ids = self.scene.material_ids()

    # ---- generated code, interpreted (scene.rs:885-1035) --------------------------------------
    def _mat(self, idx, suffix):
        return self.uniforms[self.scene.matrices[idx][0] + suffix]

    def scene_intersect(self, it: Interp, r, mask):
        nat, n = self.nat, it.n
        i = V.expand(SceneI(0, INTERSECTION_NONE, False), n)
        for pos, o in enumerate(self.scene.objects):
            guard = mask
            if o["kind"] == "debug":
                pass
            elif o["sub"] == "Normal":
                guard = mask & ~r.f["in_subspace"]
            elif o["sub"] == "Subspace":
                guard = mask & r.f["in_subspace"]
            if not guard.any():
                continue
            M.set_active(guard.sum())
            sides = [(o["m0"], True, f"teleport_{pos}_1_M"), (o["m1"], False, f"teleport_{pos}_2_M")] if o["portal"] else [(o["m0"], None, None)]
            for midx, first, tmat in sides:
                M.set_active(guard.sum())
                if o["kind"] == "flat":
                    gn = nat.get_normal(self._mat(midx, "_mat"))
                    if first is None:      # Flat/Simple, scene.rs:912-927
                        normal = V.neg(gn)
                        hit = nat.plane_intersect(r, self._mat(midx, "_mat_inv"), gn)
                    else:                  # Flat/Portal, scene.rs:928-948
                        normal = V.neg(gn) if first else gn
                        hit = nat.plane_intersect(r, self._mat(midx, "_mat_inv"), normal)
                    near = guard & nat.nearer(i, hit)
                    if not near.any():
                        continue
                    M.set_active(near.sum())
                    pos_w = add(r.f["o"], mul(r.f["d"], hit.f["t"]))
                    back = nat.is_collinear(hit.f["n"], normal)
                    args = [pos_w, hit.f["u"], hit.f["v"], back] + ([np.full(n, first)] if first is not None else [])
                    inside = V.expand(it.run_function(f"is_inside_{pos}", args, near), n)
                    if first is None:
                        new_i = nat.process_plane_intersection(i, hit, inside)
                    else:
                        new_i = nat.process_portal_intersection(i, hit, inside, I32(self.material_ids[tmat]))
                    i = V.select(near, V.expand(new_i, n), i)
                else:                      # Complex / DebugMatrix, scene.rs:893-904, 962-998
                    tr = nat.transform(self._mat(midx, "_mat_inv"), r)
                    ln = V.length(tr.f["d"])
                    tr = nat.normalize_ray(tr)
                    if o["kind"] == "debug":   # scene.rs:893-904; NOTE the normal uses adjugate of the *inverse* (reference quirk)
                        ihit = V.expand(nat.debug_intersect(tr), n)
                        ihit = ihit.with_field("hit", ihit.f["hit"].with_field("t", M.div(ihit.f["hit"].f["t"], ln)))
                        near = guard & nat.nearer(i, ihit)
                        if near.any():
                            nrm = V.normalize(mul(nat.adjugate(self._mat(midx, "_mat_inv")), ihit.f["hit"].f["n"]))
                            i = V.select(near, ihit.with_field("hit", ihit.f["hit"].with_field("n", nrm)), i)
                        continue
                    args = [V.expand(tr, n)] + ([np.full(n, first)] if first is not None else [])
                    ihit = V.expand(it.run_function(f"intersect_{pos}", args, guard), n)
                    M.set_active(guard.sum())
                    ihit = ihit.with_field("hit", ihit.f["hit"].with_field("t", M.div(ihit.f["hit"].f["t"], ln)))
                    near = guard & nat.nearer(i, ihit)
                    if first is not None:
                        near = near & (ihit.f["material"] != NOT_INSIDE)
                        tm = I32(self.material_ids[tmat])
                        mat = ihit.f["material"]
                        sub = ihit.f["in_subspace"] | (mat == TELEPORT_SUBSPACE)
                        mat = np.where((mat == TELEPORT) | (mat == TELEPORT_SUBSPACE), tm, mat).astype(I32)
                        ihit = ihit.with_field("material", mat).with_field("in_subspace", sub)
                    if not near.any():
                        continue
                    M.set_active(near.sum())
                    nrm = V.normalize(mul(nat.adjugate(self._mat(midx, "_mat")), ihit.f["hit"].f["n"]))
                    ihit = ihit.with_field("hit", ihit.f["hit"].with_field("n", nrm))
                    i = V.select(near, ihit, i)
        return i

    def material_process(self, it: Interp, r, i, mask):
        """frag.glsl:33-50 + the generated else-if chain (scene.rs:726-840)."""
        nat, n = self.nat, it.n
        hit = i.f["hit"]
        r = r.with_field("in_subspace", np.where(i.f["in_subspace"], ~r.f["in_subspace"], r.f["in_subspace"]))
        out = V.expand(nat.material_final(vec(0.0, 0.0, 0.0)), n)  # unknown material id
        mat = i.f["material"]

        def put(sel, value):
            nonlocal out
            out = V.select(sel, V.expand(value, n), out)

        for dbg, col in ((DEBUG_RED, (0.9, 0.2, 0.2)), (DEBUG_GREEN, (0.2, 0.9, 0.2)), (DEBUG_BLUE, (0.2, 0.2, 0.9))):
            sel = mask & (mat == dbg)
            if sel.any():
                c = nat.color(*(M.lit(str(x)) for x in col))
                put(sel, nat.material_simple2(hit, r, c, fl(0.5), False, fl(1.0), fl(0.0), False, False))
        for k, m in enumerate(self.scene.materials):
            sel = mask & (mat == self.material_ids[m["name"] + "_M"])
            if not sel.any():
                continue
            M.set_active(sel.sum())
            if m["kind"] == "Simple":
                # the generator prints `{:e}` decimals of the f64 values, the GLSL compiler rounds them to binary32
                c = vec(*(M.lit(repr(x)) for x in m["color"]))
                put(sel, nat.material_simple2(hit, r, c, M.lit(repr(m["normal_coef"])), m["grid"], M.lit(repr(m["grid_scale"])), M.lit(repr(m["grid_coef"])),
                                              m["grid2"], m["grid3"]))
            elif m["kind"] == "Reflect":
                put(sel, nat.material_reflect(hit, r, vec(*(M.lit(repr(x)) for x in m["color"]))))
            elif m["kind"] == "Refract":
                put(sel, nat.material_refract(hit, r, vec(*(M.lit(repr(x)) for x in m["color"])), M.lit(repr(m["refractive_index"]))))
            else:
                put(sel, it.run_function(f"__material_{k}", [V.expand(hit, n), V.expand(r, n), V.expand(i, n)], sel))
        for pos, o in enumerate(self.scene.objects):
            if o["kind"] == "debug" or not o["portal"] or o["m0"] < 0 or o["m1"] < 0:
                continue
            na, nb = self.scene.matrices[o["m0"]][0], self.scene.matrices[o["m1"]][0]
            for which, tname in ((1, f"{na}_to_{nb}_mat_teleport"), (2, f"{nb}_to_{na}_mat_teleport")):
                sel = mask & (mat == self.material_ids[f"teleport_{pos}_{which}_M"])
                if sel.any():
                    M.set_active(sel.sum())
                    put(sel, nat.material_teleport(hit, r, self.uniforms[tname]))
        return out

    def scene_intersect_material_process(self, it: Interp, r, mask):
        """frag.glsl:52-59 + scene.rs:1026-1035."""
        n = it.n
        result = V.expand(mk("SceneIntersectionWithMaterial", scene=SCENE_INTERSECTION_NONE, material=Natives.material_empty()), n)
        for k in range(len(self.scene.intersection_materials)):
            hit = V.expand(it.run_function(f"intersect_material_{k}", [V.expand(r, n)], mask), n)
            near = mask & Natives.nearer(result.f["scene"].f["hit"], hit.f["scene"].f["hit"])
            result = V.select(near, hit, result)
        return result

    # ---- frag.glsl:106-159 -----------------------------------------------------------------------
    def ray_tracing(self, r, camera_scale):
        """r: Ray over N lanes.  Returns (rgb float32 (N,3), segments per lane, depth (N,), has_depth (N,))."""
        nat, u = self.nat, self.uniforms
        N = len(np.asarray(r.f["tmul"]))
        depth = int(u["_ray_tracing_depth"])
        if self.scene.skybox is not None:   # scene.rs:1052-1059, evaluated once per ray_tracing call on the primary direction
            rd2 = mul(u["_camera_mul_inv"], r.f["d"])
            su = M.atan2(rd2.c[2], rd2.c[0])
            sv = M.atan2(M.sqrt(M.add(M.mul(rd2.c[0], rd2.c[0]), M.mul(rd2.c[2], rd2.c[2]))), rd2.c[1])
            pi = self._program.globals["PI"]
            tex = V.texture(self._program.globals[self.scene.skybox + "_tex"], vec(M.div(M.add(M.div(su, pi), fl(1.0)), fl(2.0)), M.div(sv, pi))) \
                if self._program.globals.get(self.scene.skybox + "_tex") is not None else vec(0.0, 0.0, 0.0, 1.0)
            not_found_all = nat.sqrvec(xyz(tex))
        else:
            not_found_all = None
        not_found = nat.color(M.lit("0.6"), M.lit("0.6"), M.lit("0.6"))
        result = np.zeros((N, 3), F32)  # depth exhausted -> color(0,0,0)
        depth_out = np.zeros(N, F32)
        has_depth = np.zeros(N, bool)
        segments = np.zeros(N, np.int64)
        idx = np.arange(N)              # lanes still tracing
        color = V.expand(vec(1.0, 1.0, 1.0), N)
        all_t = np.zeros(N, F32)
        t_start, t_end = u["_t_start"], u["_t_end"]
        for _ in range(depth):
            n = len(idx)
            if n == 0:
                break
            it = Interp(n, self._program)
            full = np.ones(n, bool)
            segments[idx] += 1
            M.set_active(n)
            i = self.scene_intersect(it, r, full)
            i2 = self.scene_intersect_material_process(it, r, full)
            M.set_active(n)
            use2 = nat.nearer(i.f["hit"], i2.f["scene"].f["hit"])
            use1 = ~use2 & i.f["hit"].f["hit"]
            t_hit = np.where(use2, i2.f["scene"].f["hit"].f["t"], i.f["hit"].f["t"]).astype(F32)
            moved = use1 | use2
            r_adv = r.with_field("o", V.select(moved, add(r.f["o"], mul(r.f["d"], t_hit)), r.f["o"]))
            all_t = np.where(moved, M.add(all_t, M.mul(t_hit, r.f["tmul"])), all_t).astype(F32)
            # reference leaves `m` unset when a snippet reports hit with t <= 0; defined as all-zero here too
            m = V.expand(MatProc(False, vec(0.0, 0.0, 0.0), RAY_NONE), n)
            custom = use2 & (i2.f["scene"].f["material"] == CUSTOM_MATERIAL)
            m = V.select(custom, i2.f["material"], m)
            via2 = use2 & ~custom
            if via2.any():
                m = V.select(via2, self.material_process(it, r_adv, i2.f["scene"], via2), m)
            if use1.any():
                m = V.select(use1, self.material_process(it, r_adv, i, use1), m)
            M.set_active(n)
            any_hit = i.f["hit"].f["hit"] | i2.f["scene"].f["hit"].f["hit"]
            # escaped the scene
            esc = ~any_hit
            nf = not_found if not_found_all is None else V.take(V.expand(not_found_all, N), idx)
            esc_col = V.select(r_adv.f["in_subspace"], vec(0.0, 0.0, 0.0), mul(color, nf))
            color_next = mul(color, m.f["mul_to_color"])
            final = any_hit & m.f["is_final"]
            # distance darkening (frag.glsl:135-146)
            lim_lo, lim_hi = M.mul(t_start, camera_scale), M.mul(t_end, camera_scale)
            dark = M.gt(all_t, lim_lo) & (int(u["_darken_by_distance"]) == 1)
            at = np.where(M.gt(all_t, lim_hi), lim_hi, all_t).astype(F32)
            gray = M.div(M.div(M.sub(at, lim_lo), M.sub(t_end, t_start)), camera_scale)
            g4 = nat.sqr(nat.sqr(gray))
            k4 = nat.sqr(nat.sqr(M.sub(fl(1.0), gray)))
            darkened = add(mul(nat.color(fl(0.0), fl(0.0), fl(0.0)), g4), mul(color_next, k4))
            fin_col = V.select(dark, darkened, color_next)
            done = esc | final
            out_col = V.select(esc, esc_col, fin_col)
            if final.any():  # depth = all_t / max(camera_scale, 1e-6), taken before the darkening clamp (frag.glsl:135)
                dval = M.div(all_t, M.fmax(camera_scale, M.lit("1e-6")))
                depth_out[idx[final]] = np.broadcast_to(dval, (n,))[final]
                has_depth[idx[final]] = True
            if done.any():
                lanes = idx[done]
                for c in range(3):
                    result[lanes, c] = np.broadcast_to(out_col.c[c], (n,))[done]
            keep = ~done
            idx = idx[keep]
            if len(idx) == 0:
                break
            sel = np.nonzero(keep)[0]
            r = V.take(V.expand(m.f["new_ray"], n), sel)
            color = V.take(V.expand(color_next, n), sel)
            all_t = all_t[sel]
        return result, segments, depth_out, has_depth

    # ---- frag.glsl:209-257 + host side src/main.rs:1361-1409 ----------------------------------------
    def teleport_external_ray(self, a, b):
        """Segment a -> b through at most ten portals.  Returns (pos float32[3] | None, encounter_object,
        change_subspace) like SceneRenderer::teleport_external_ray."""
        self.build(0, 0)
        u, nat = self.uniforms, self.nat
        # src/main.rs:1367: teleport_light_u is forced to 1 for this query
        if "teleport_light_u" in self._program.globals:
            self._program.globals["teleport_light_u"] = I32(1)
        a32, b32 = np.asarray(a, np.float64).astype(F32), np.asarray(b, np.float64).astype(F32)
        in_sub = int(u["_camera_in_subspace"]) == 1
        r = Ray(vec(a32[0], a32[1], a32[2], 1.0), Vec(list(sub(vec(*b32), vec(*a32)).c) + [fl(0.0)]), fl(1.0), np.array([in_sub]))
        r = V.expand(nat.normalize_ray(r), 1)
        have_result, stop_at_object = False, False
        all_t = np.zeros(1, F32)
        one = np.ones(1, bool)
        for _ in range(10):
            it = Interp(1, self._program)
            M.set_active(1)
            i = self.scene_intersect(it, r, one)
            i2 = self.scene_intersect_material_process(it, r, one)
            cont = False
            use2 = bool(nat.nearer(i.f["hit"], i2.f["scene"].f["hit"])[0])
            src = i2.f["scene"] if use2 else (i if bool(np.asarray(i.f["hit"].f["hit"]).reshape(-1)[0]) else None)
            if src is not None:
                t = src.f["hit"].f["t"]
                if bool((M.add(M.mul(t, r.f["tmul"]), all_t) < fl(1.0))[0]):
                    r = r.with_field("o", add(r.f["o"], mul(r.f["d"], t)))
                    all_t = M.add(all_t, M.mul(t, r.f["tmul"]))
                    if use2 and int(np.asarray(i2.f["scene"].f["material"]).reshape(-1)[0]) == CUSTOM_MATERIAL:
                        m = i2.f["material"]
                    else:
                        m = self.material_process(it, r, src, one)
                    final = bool(np.asarray(m.f["is_final"]).reshape(-1)[0])
                    cont = not final
                    stop_at_object = stop_at_object or final
            if not cont:
                break
            r = V.expand(m.f["new_ray"], 1)
            have_result = True
        change_subspace = int(bool(np.asarray(r.f["in_subspace"]).reshape(-1)[0])) != int(u["_camera_in_subspace"])
        if not have_result:
            return None, stop_at_object, change_subspace
        o = add(r.f["o"], div(mul(r.f["d"], M.sub(fl(1.0), all_t)), r.f["tmul"]))
        pos = np.array([np.asarray(c).reshape(-1)[0] for c in o.c[:3]], F32)
        if pos[0] == 0 and pos[1] == 0 and pos[2] == 0:
            return None, stop_at_object, change_subspace
        return pos, stop_at_object, change_subspace

    # ---- frag.glsl:305-342 -----------------------------------------------------------------------
    def panini(self, tc, fov, d):
        pow2 = lambda x: M.mul(x, x)
        d2 = M.mul(d, d)
        pi05 = M.mul(M.lit("3.14159265359"), fl(0.5))
        fo = M.sub(pi05, M.mul(fov, fl(0.5)))
        f = M.div(M.cos(fo), M.sin(fo))
        f2 = M.mul(f, f)
        b = M.div(M.sub(M.sqrt(M.fmax(fl(0.0), M.mul(pow2(M.add(d, d2)), M.add(f2, M.mul(f2, f2))))), M.add(M.mul(d, f), f)),
                  M.sub(M.add(d2, M.mul(d2, f2)), fl(1.0)))
        tc = mul(tc, b)
        h, v = tc.c
        h2 = M.mul(h, h)
        k = M.div(h2, pow2(M.add(d, fl(1.0))))
        k2 = M.mul(k, k)
        discr = M.fmax(fl(0.0), M.sub(M.mul(k2, d2), M.mul(M.add(k, fl(1.0)), M.sub(M.mul(k, d2), fl(1.0)))))
        cos_phi = M.div(M.add(M.mul(M.neg(k), d), M.sqrt(discr)), M.add(k, fl(1.0)))
        s_big = M.div(M.add(d, fl(1.0)), M.add(d, cos_phi))
        tan_theta = M.div(v, s_big)
        sin_phi = M.sqrt(M.fmax(fl(0.0), M.sub(fl(1.0), pow2(cos_phi))))
        sin_phi = np.where(M.lt(tc.c[0], fl(0.0)), M.mul(sin_phi, fl(-1.0)), sin_phi).astype(F32)
        s = M.inversesqrt(M.add(fl(1.0), pow2(tan_theta)))
        return mul(vec(sin_phi, tan_theta, cos_phi), s)

    # ---- frag.glsl:80-104 ----------------------------------------------------------------------------
    def sample_depth_gradient(self, depth):
        u, nat = self.uniforms, self.nat
        dmin, dmax = M.fmin(u["_depth_map_min"], u["_depth_map_max"]), M.fmax(u["_depth_map_min"], u["_depth_map_max"])
        norm = M.clamp(M.div(M.sub(depth, dmin), M.fmax(M.lit("1e-6"), M.sub(dmax, dmin))), fl(0.0), fl(1.0))
        t = M.sub(fl(1.0), norm)
        stops = [nat.sqrvec(vec(*(M.lit(x) for x in c))) for c in (("0.001462", "0.000466", "0.013866"), ("0.258234", "0.038571", "0.406485"),
                 ("0.578304", "0.148039", "0.404411"), ("0.865006", "0.316822", "0.226055"), ("0.987622", "0.645320", "0.039886"),
                 ("0.988362", "0.998364", "0.644924"))]
        p2 = M.lit("0.2")
        seg = [V.map3(M.mix, stops[k], stops[k + 1], M.div(M.sub(t, M.lit(lo)), p2) if lo != "0" else M.div(t, p2))
               for k, lo in enumerate(("0", "0.2", "0.4", "0.6", "0.8"))]
        out = seg[4]
        for k, hi in ((3, "0.8"), (2, "0.6"), (1, "0.4"), (0, "0.2")):
            out = V.select(M.lt(t, M.lit(hi)), seg[k], out)
        return out

    # ---- frag.glsl:408-464 ------------------------------------------------------------------------------
    def get_color2(self, image_position, camera, in_subspace, camera_scale, resolution=None):
        u = self.uniforms
        n = len(np.asarray(image_position.c[0]))
        resolution = resolution if resolution is not None else u["_resolution"]
        o = mul(camera, vec(0.0, 0.0, 0.0, 1.0))
        ix, iy = image_position.c
        black = np.zeros(n, bool)  # pixels outside the projection's valid area return vec3(0) without tracing
        pi = M.lit("3.14159265359")
        pi05 = M.mul(pi, fl(0.5))
        if int(u["_use_panini_projection"]) == 1:
            p = self.panini(vec(ix, iy), u["_view_angle"], u["_panini_param"])
            d = V.normalize(mul(camera, Vec(list(p.c) + [fl(0.0)])))
        elif int(u["_use_360_camera"]) == 1 or int(u["_use_180_camera"]) == 1:
            if int(u["_use_360_camera"]) == 1:   # frag.glsl:413-437
                coef = M.fmin(resolution.c[0], resolution.c[1])
                ax, ay = M.div(resolution.c[0], coef), M.div(resolution.c[1], coef)
                wide = M.ge(ax, M.mul(fl(2.0), ay))
                rx = np.where(wide, M.mul(fl(2.0), ay), ax).astype(F32)
                ry = np.where(wide, ay, M.div(ax, fl(2.0))).astype(F32)
                black = M.gt(M.absf(ix), rx) | M.gt(M.absf(iy), ry)
                yaw, pitch = M.mul(M.div(ix, rx), pi), M.mul(M.div(iy, ry), pi05)
            else:                                # frag.glsl:438-448
                black = M.gt(M.absf(ix), fl(1.0)) | M.gt(M.absf(iy), fl(1.0))
                yaw, pitch = M.mul(ix, pi05), M.mul(iy, pi05)
            black = np.broadcast_to(black, (n,))
            local = vec(M.mul(M.sin(yaw), M.cos(pitch)), M.sin(pitch), M.mul(M.cos(yaw), M.cos(pitch)))
            d = V.normalize(mul(camera, Vec(list(local.c) + [fl(0.0)])))
        else:
            h = M.tan(M.div(u["_view_angle"], fl(2.0)))
            d = V.normalize(mul(camera, vec(M.mul(ix, h), M.mul(iy, h), fl(1.0), fl(0.0))))
        rgb = np.zeros((n, 3), F32)
        seg = np.zeros(n, np.int64)
        live = np.nonzero(~black)[0]
        if len(live):
            ray = V.take(V.expand(Ray(o, d, fl(1.0), np.full(n, bool(in_subspace))), n), live)
            col, sg, depth, has_depth = self.ray_tracing(ray, camera_scale)
            if int(u["_draw_depth_map"]) == 1:   # frag.glsl:456-462
                grad = self.sample_depth_gradient(depth)
                col = np.where(has_depth[:, None], np.stack([np.broadcast_to(c, depth.shape) for c in grad.c], axis=1), F32(0)).astype(F32)
            rgb[live] = col
            seg[live] = sg
        return rgb, seg

    # ---- frag.glsl:343-406: red/cyan anaglyph of the two eye images (linear light) with ghosting compensation ----
    def anaglyph_combine(self, left, right, mode):
        u = self.uniforms
        clamp01 = lambda v: Vec([M.clamp(c, fl(0.0), fl(1.0)) for c in v.c])
        lv = clamp01(Vec([left[:, k] for k in range(3)]))
        rv = clamp01(Vec([right[:, k] for k in range(3)]))
        luma = vec(M.lit("0.299"), M.lit("0.587"), M.lit("0.114"))
        p, q = u["_anaglyph_p"], u["_anaglyph_q"]
        l, r = V.dot(lv, luma), V.dot(rv, luma)
        denom = M.fmax(M.lit("1e-6"), M.sub(fl(1.0), M.mul(p, q)))
        r_out = M.div(M.sub(l, M.mul(p, r)), denom)
        c_out = M.div(M.sub(r, M.mul(q, l)), denom)
        if mode == 0:
            out = clamp01(Vec([r_out, c_out, c_out]))
        else:
            sum_gb = M.add(rv.c[1], rv.c[2])
            with np.errstate(all="ignore"):
                k = np.where(sum_gb > M.lit("1e-6"), M.div(M.mul(fl(2.0), c_out), sum_gb), F32(0.0)).astype(F32)
            out = clamp01(Vec([r_out, M.mul(rv.c[1], k), M.mul(rv.c[2], k)]))
        return np.stack([np.broadcast_to(np.asarray(c, F32), (left.shape[0],)) for c in out.c], axis=1).astype(F32)

    # ---- frag.glsl:466-503 (mono / side-by-side / anaglyph) ----------------------------------------------
    def get_color(self, image_position):
        u = self.uniforms
        n = len(np.asarray(image_position.c[0]))
        if int(u["_draw_anaglyph"]) == 1 and self.anaglyph_compiled_in:
            cl, sl = self.get_color2(image_position, u["_camera_left_eye"], int(u["_left_eye_in_subspace"]) == 1, u["_left_eye_scale"], u["_resolution"])
            cr, sr = self.get_color2(image_position, u["_camera_right_eye"], int(u["_right_eye_in_subspace"]) == 1, u["_right_eye_scale"], u["_resolution"])
            return self.anaglyph_combine(cl, cr, int(u["_anaglyph_mode"])), sl + sr
        if int(u["_draw_side_by_side"]) != 1:
            return self.get_color2(image_position, u["_camera"], int(u["_camera_in_subspace"]) == 1, u["_camera_scale"], u["_resolution"])
        res = u["_resolution"]
        coef = M.fmin(res.c[0], res.c[1])
        position = add(mul(div(image_position, fl(2.0)), coef), div(res, fl(2.0)))
        half = vec(M.div(res.c[0], fl(2.0)), res.c[1])
        coef2 = M.fmin(half.c[0], half.c[1])
        left = np.broadcast_to(M.lt(position.c[0], half.c[0]), (n,))
        pos_l = mul(div(sub(position, div(half, fl(2.0))), coef2), fl(2.0))
        pos_r = mul(div(sub(sub(position, vec(half.c[0], fl(0.0))), div(half, fl(2.0))), coef2), fl(2.0))
        rgb = np.zeros((n, 3), F32)
        seg = np.zeros(n, np.int64)
        for sel, pos, cam, sub_, scale in ((left, pos_l, "_camera_left_eye", "_left_eye_in_subspace", "_left_eye_scale"),
                                           (~left, pos_r, "_camera_right_eye", "_right_eye_in_subspace", "_right_eye_scale")):
            lanes = np.nonzero(sel)[0]
            if len(lanes):
                c, s = self.get_color2(V.take(V.expand(pos, n), lanes), u[cam], int(u[sub_]) == 1, u[scale], half)
                rgb[lanes], seg[lanes] = c, s
        return rgb, seg

    # ---- frag.glsl:506-527,550-551 + vertex stage scene.rs:1674-1697 ---------------------------------
    def shade_pixels(self, width, height, px, py):
        """px, py: integer pixel coordinates (arrays).  Returns dict(rgba32f (N,4), rgba8 (N,4), segments (N,))."""
        self.build(width, height)
        M.reset_stats()
        u = self.uniforms
        res = u["_resolution"]
        position = vec(M.add(M.f32(px), fl(0.5)), M.add(M.f32(py), fl(0.5)))
        coef = M.fmin(res.c[0], res.c[1])
        uv_screen = mul(div(sub(position, div(res, fl(2.0))), coef), fl(2.0))
        pixel_size = M.div(fl(1.0), M.fmin(res.c[0], res.c[1]))
        n = len(np.atleast_1d(px))
        total = np.zeros((n, 3), F32)
        segments = np.zeros(n, np.int64)
        aa_start, aa_count = int(u["_aa_start"]), int(u["_aa_count"])
        a1, a2 = M.lit("0.7548776662466927600500267982588025643670318456949186300834636687"), M.lit("0.5698402909980532659121818632752155853637566123932930564053138358")
        for a in range(aa_start, aa_start + aa_count):
            af = F32(a)
            offset = vec(M.mod(M.add(fl(0.5), M.mul(a1, af)), fl(1.0)), M.mod(M.add(fl(0.5), M.mul(a2, af)), fl(1.0)))  # quasi_random
            pos = add(uv_screen, mul(mul(offset, pixel_size), fl(2.0)))
            rgb, seg = self.get_color(V.expand(pos, n))
            total = M.add(total, rgb)
            segments += seg
        M.set_active(n)
        rgb = M.sqrt(np.multiply(total, M.div(fl(1.0), F32(aa_count))))  # result / float(aa_count): vec / scalar contract
        rgba = np.concatenate([rgb, np.ones((n, 1), F32)], axis=1)
        self.stats = dict(M.STATS, segments=int(segments.sum()))
        return dict(rgba32f=rgba, rgba8=to_rgba8(rgba), segments=segments)

    def render(self, width, height, rows=None, cols=None):
        r0, r1 = rows if rows else (0, height)
        c0, c1 = cols if cols else (0, width)
        ys, xs = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing="ij")
        out = self.shade_pixels(width, height, xs.ravel(), ys.ravel())
        shape = (r1 - r0, c1 - c0)
        return dict(rgba32f=out["rgba32f"].reshape(*shape, 4), rgba8=out["rgba8"].reshape(*shape, 4), segments=out["segments"].reshape(shape))


class CameraRig:
    """RotateAroundCam + SceneRenderer::update / teleport_camera / teleport_matrix (src/main.rs:278-304, 1174-1264,
    1430-1538), restated in Python floats (binary64) on top of Oracle.teleport_external_ray."""

    FIELDS = ("look_at", "alpha", "beta", "r", "teleport_matrix", "in_subspace", "free_movement", "prev_cam_pos", "from_cam", "do_not_teleport")

    def __init__(self, oracle: Oracle):
        from .scene_eval import IDENT, camera_matrix, m_inverse, m_mul, m_mul_vec, _pos_vec

        self.o = oracle
        self._cm, self._inv, self._mul, self._mv, self._pv = camera_matrix, m_inverse, m_mul, m_mul_vec, _pos_vec
        c = oracle.scene.cam
        self.look_at, self.alpha, self.beta, self.r = list(c["look_at"]), c["alpha"], c["beta"], c["r"]
        self.teleport_matrix, self.in_subspace, self.free_movement = IDENT, False, False
        self.from_cam, self.do_not_teleport, self.allow_teleport = -1, False, True
        self.left_eye_matrix, self.right_eye_matrix, self.left_eye_in_subspace, self.right_eye_in_subspace = IDENT, IDENT, False, False
        self.stereo, self.eye_distance, self.swap_eyes = False, 0.07, False  # draw_side_by_side, src/main.rs:1028-1029
        self.prev_cam_pos = self.cam_pos()
        self.original = self._calculated()
        self.prev = self._snapshot()

    def _snapshot(self):
        return {k: (list(getattr(self, k)) if k in ("look_at", "prev_cam_pos") else getattr(self, k)) for k in self.FIELDS}

    def _restore(self, snap):
        for k, v in snap.items():
            setattr(self, k, list(v) if k in ("look_at", "prev_cam_pos") else v)

    def _calculated(self):
        return dict(look_at=list(self.look_at), alpha=self.alpha, beta=self.beta, r=self.r, teleport_matrix=self.teleport_matrix,
                    in_subspace=self.in_subspace, free_movement=self.free_movement)

    def matrix(self, snap=None):
        c = snap or self.__dict__
        return self._cm(c["look_at"], c["alpha"], c["beta"], c["r"], c["teleport_matrix"], c["free_movement"])

    def cam_pos(self):
        return self._mv(self.matrix(), [0.0, 0.0, 0.0, 1.0])[:3]

    def settings(self):
        return dict(look_at=self.look_at, alpha=self.alpha, beta=self.beta, r=self.r, teleport_matrix=self.teleport_matrix, in_subspace=self.in_subspace,
                    free_movement=self.free_movement, left_eye_matrix=self.left_eye_matrix, right_eye_matrix=self.right_eye_matrix,
                    left_eye_in_subspace=self.left_eye_in_subspace, right_eye_in_subspace=self.right_eye_in_subspace)

    def teleport_eye_matrices(self):
        """SceneRenderer::teleport_eye_matrices (src/main.rs:1121-1172)."""
        if not (self.stereo and self.allow_teleport):
            return
        dist = -self.eye_distance if self.swap_eyes else self.eye_distance

        def one_eye(x):
            start = self.cam_pos()
            m = self.matrix()
            direction = self._mv(m, [x, 0.0, 0.0, 1.0])[:3]
            shift = [direction[k] - start[k] for k in range(3)]
            result = self._mul([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], shift + [1.0]], m)
            sub_ = self.in_subspace
            new_pos, _hit, change_sub = self._query(start, direction)
            if new_pos is not None:
                for dx in (0.001, 0.0001, 0.00001, 0.000001):
                    tm = self._teleport_matrix(result, start, direction, new_pos, dx)
                    if tm is not None:
                        result = tm
                        if change_sub:
                            sub_ = not self.in_subspace
                        break
            return result, sub_

        self.left_eye_matrix, self.left_eye_in_subspace = one_eye(-dist)
        self.right_eye_matrix, self.right_eye_in_subspace = one_eye(dist)

    def _query(self, a, b):
        self.o.camera = self.settings()
        pos, hit, sub_ = self.o.teleport_external_ray(a, b)
        return (None if pos is None else [float(x) for x in pos]), hit, sub_

    def _teleport_matrix(self, matrix, start, direction, actual, dx):
        cols = []
        for axis in ([1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0]):
            v = [x * dx for x in self._mv(matrix, axis)][:3]
            pos, _, _ = self._query([start[k] + v[k] for k in range(3)], [direction[k] + v[k] for k in range(3)])
            if pos is None:
                return None
            cols.append([(pos[k] - actual[k]) / dx for k in range(3)] + [0.0])
        new_mat = cols + [[0.0, 0.0, 0.0, 1.0]]
        moved = self._mv(self._mul(new_mat, self._inv(matrix)), [direction[0], direction[1], direction[2], 1.0])
        return cols + [[actual[k] - moved[k] for k in range(3)] + [1.0]]

    def _teleport_camera(self, prev):
        """SceneRenderer::teleport_camera (allow_teleport on, stop_at_objects off).  -> (teleported, blocked)"""
        if self.do_not_teleport:
            self.do_not_teleport = False
            self.prev_cam_pos = self.cam_pos()
            return False, False
        if not self.allow_teleport:
            return False, False
        pos = self.cam_pos()
        new_pos, _hit, change_sub = self._query(self.prev_cam_pos, pos)
        if new_pos is None:
            self.prev_cam_pos = pos
            return False, False
        for dx in (0.001, 0.0001, 0.00001, 0.000001):
            m = self._teleport_matrix(self.teleport_matrix, self.prev_cam_pos, pos, new_pos, dx)
            if m is None:
                continue
            self.teleport_matrix = m
            if change_sub:
                self.in_subspace = not self.in_subspace
            self.prev_cam_pos = self.cam_pos()
            return True, False
        self._restore(prev)
        return False, True

    def move(self, look_at, alpha, beta, r):
        """One interactive step: place the orbit, then teleport_camera against the previous state.  -> (teleported, blocked)"""
        prev = self._snapshot()
        self.look_at, self.alpha, self.beta, self.r = list(look_at), alpha, beta, r
        out = self._teleport_camera(prev)
        self.teleport_eye_matrices()
        return out

    def update(self, seconds):
        """SceneRenderer::update (src/main.rs:1430-1538).  Leaves oracle.camera = the camera to draw with.  -> (teleported, blocked)"""
        sc = self.o.scene
        override = sc.update(seconds)
        sc.camera_object_matrix = self.matrix()
        cur = sc.current_cam
        if self.from_cam != cur:
            if cur >= 0:
                if self.from_cam < 0:
                    self.original = self._calculated()
                c = sc.calculated_cam(cur)
            else:
                c = self.original
            self.from_cam = cur
            self.look_at, self.alpha, self.beta, self.r = list(c["look_at"]), c["alpha"], c["beta"], c["r"]
            self.teleport_matrix, self.in_subspace, self.free_movement = c["teleport_matrix"], c["in_subspace"], c["free_movement"]
            if self.free_movement:
                pv = self._pv(self.alpha, self.beta, self.r)
                self.look_at = [pv[k] + self.look_at[k] for k in range(3)]
            self.do_not_teleport = True
        elif self.from_cam >= 0 and not self.free_movement:
            self.look_at = list(sc.calculated_cam(self.from_cam)["look_at"])
        if override is not None:
            self.look_at, self.alpha, self.beta, self.r = list(override["look_at"]), override["alpha"], override["beta"], override["r"]
            self.free_movement = override["free_movement"]
            if override["override_matrix"]:
                self.teleport_matrix, self.in_subspace = override["teleport_matrix"], override["in_subspace"]
                self.do_not_teleport = True
        result = (False, False)
        if self.matrix() != self.matrix(self.prev):
            result = self._teleport_camera(dict(self.prev))
        self.teleport_eye_matrices()
        self.prev = self._snapshot()
        sc.camera_object_matrix = self.matrix()
        self.o.camera = self.settings()
        return result


def to_rgba8(rgba32f):
    """GL fixed-point conversion: clamp to [0,1], *255, round to nearest (NaN -> 0)."""
    v = np.asarray(rgba32f, F32)
    with np.errstate(invalid="ignore"):
        q = np.floor(M.fma(v, F32(255.0), F32(0.5)))
        q = np.where(v >= 1.0, 255.0, np.where(v > 0.0, q, 0.0))
    return q.astype(np.uint8)


if __name__ == "__main__":
    import sys
    import time

    from PIL import Image

    path, w, h, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    o = Oracle(path)
    o.options["render_depth"] = depth
    t0 = time.time()
    out = o.render(w, h)
    print(f"{time.time()-t0:.1f}s segments={o.stats['segments']} flops/segment={o.stats['flops']/max(1,o.stats['segments']):.0f}")
    if len(sys.argv) > 5:
        Image.fromarray(out["rgba8"]).save(sys.argv[5])
