"""oracle/scene_eval.py -- scene file -> the constants a frame needs (oracle side).

TEST INFRASTRUCTURE ONLY (CPU oracle).  Nothing under portal_amd/ may import this.

Independent restatement (Python, binary64) of
  * the on-disk schema and loader   src/gui/scene_serialized.rs:610-646, 1102-1230
  * AnyUniform::get                  src/gui/uniform.rs:268-278, 1009-1140
  * Matrix::get                      src/gui/matrix.rs:510-631  (+ glam 0.13.1 formulas, Cargo.lock:718)
  * Scene::set_uniforms              src/gui/scene.rs:545-657   (X_mat, X_mat_inv, A_to_B_mat_teleport)
  * RotateAroundCam::get_matrix      src/main.rs:278-304
  * SceneRenderer::set_uniforms      src/main.rs:1266-1359      (builtin `_...` uniforms, defaults :1021-1040)
"""
from __future__ import annotations

import math

import numpy as np

from . import formula as F
from . import ron

# ---------------------------------------------------------------------------------------------
# binary64 4x4 matrices as lists of 4 columns of 4 floats (glam: x_axis..w_axis)
# ---------------------------------------------------------------------------------------------
# ---- IEEE binary64 semantics where Python raises (Rust / C return inf or NaN): the graph evaluator must not die on a
# singular matrix or an infinite angle, it must produce the same NaNs the product uploads
def _sin(x):
    return math.sin(x) if math.isfinite(x) else float("nan")


def _cos(x):
    return math.cos(x) if math.isfinite(x) else float("nan")


def _sqrt(x):
    return math.sqrt(x) if x >= 0 else (x if x == 0 else float("nan"))  # sqrt(-0.0) = -0.0, sqrt(negative / NaN) = NaN


def _fdiv(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0 or a != a:
            return float("nan")
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


IDENT = [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]


def m_mul_vec(m, v):
    r = [m[0][i] * v[0] for i in range(4)]
    for k in (1, 2, 3):
        r = [m[k][i] * v[k] + r[i] for i in range(4)]
    return r


def m_mul(a, b):
    return [m_mul_vec(a, col) for col in b]


def m_inverse(m):
    (m00, m01, m02, m03), (m10, m11, m12, m13), (m20, m21, m22, m23), (m30, m31, m32, m33) = m
    c00 = m22 * m33 - m32 * m23; c02 = m12 * m33 - m32 * m13; c03 = m12 * m23 - m22 * m13
    c04 = m21 * m33 - m31 * m23; c06 = m11 * m33 - m31 * m13; c07 = m11 * m23 - m21 * m13
    c08 = m21 * m32 - m31 * m22; c10 = m11 * m32 - m31 * m12; c11 = m11 * m22 - m21 * m12
    c12 = m20 * m33 - m30 * m23; c14 = m10 * m33 - m30 * m13; c15 = m10 * m23 - m20 * m13
    c16 = m20 * m32 - m30 * m22; c18 = m10 * m32 - m30 * m12; c19 = m10 * m22 - m20 * m12
    c20 = m20 * m31 - m30 * m21; c22 = m10 * m31 - m30 * m11; c23 = m10 * m21 - m20 * m11
    f0, f1, f2 = (c00, c00, c02, c03), (c04, c04, c06, c07), (c08, c08, c10, c11)
    f3, f4, f5 = (c12, c12, c14, c15), (c16, c16, c18, c19), (c20, c20, c22, c23)
    v0, v1, v2, v3 = (m10, m00, m00, m00), (m11, m01, m01, m01), (m12, m02, m02, m02), (m13, m03, m03, m03)
    i0 = [v1[k] * f0[k] - v2[k] * f1[k] + v3[k] * f2[k] for k in range(4)]
    i1 = [v0[k] * f0[k] - v2[k] * f3[k] + v3[k] * f4[k] for k in range(4)]
    i2 = [v0[k] * f1[k] - v1[k] * f3[k] + v3[k] * f5[k] for k in range(4)]
    i3 = [v0[k] * f2[k] - v1[k] * f4[k] + v2[k] * f5[k] for k in range(4)]
    sa, sb = (1.0, -1.0, 1.0, -1.0), (-1.0, 1.0, -1.0, 1.0)
    inv = [[i0[k] * sa[k] for k in range(4)], [i1[k] * sb[k] for k in range(4)], [i2[k] * sa[k] for k in range(4)], [i3[k] * sb[k] for k in range(4)]]
    d = [m[0][k] * (inv[0][0], inv[1][0], inv[2][0], inv[3][0])[k] for k in range(4)]
    det = d[0] + d[1] + d[2] + d[3]
    with np.errstate(all="ignore"):
        rcp = float(np.float64(1.0) / np.float64(det))
        return [[float(np.float64(x) * np.float64(rcp)) for x in col] for col in inv]


def q_mul(a, b):
    x0, y0, z0, w0 = a
    x1, y1, z1, w1 = b
    return (w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1, w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
            w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1, w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1)


def from_scale_rotation_translation(scale, rot_xyz, t):
    qx = (_sin(rot_xyz[0] * 0.5), 0.0, 0.0, _cos(rot_xyz[0] * 0.5))
    qy = (0.0, _sin(rot_xyz[1] * 0.5), 0.0, _cos(rot_xyz[1] * 0.5))
    qz = (0.0, 0.0, _sin(rot_xyz[2] * 0.5), _cos(rot_xyz[2] * 0.5))
    x, y, z, w = q_mul(q_mul(qx, qy), qz)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz = x * x2, x * y2, x * z2
    yy, yz, zz = y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    xa = [1.0 - (yy + zz), xy + wz, xz - wy, 0.0]
    ya = [xy - wz, 1.0 - (xx + zz), yz + wx, 0.0]
    za = [xz + wy, yz - wx, 1.0 - (xx + yy), 0.0]
    return [[c * scale[0] for c in xa], [c * scale[1] for c in ya], [c * scale[2] for c in za], [t[0], t[1], t[2], 1.0]]


def to_scale_rotation_translation(m):
    """glam 0.13 Mat4::to_scale_rotation_translation on columns m[c][r] -> (scale xyz, quaternion xyzw, translation xyz)."""
    (m00, m01, m02, m03), (m10, m11, m12, m13), (m20, m21, m22, m23), (m30, m31, m32, m33) = m
    a2323, a1323, a1223 = m22 * m33 - m23 * m32, m21 * m33 - m23 * m31, m21 * m32 - m22 * m31
    a0323, a0223, a0123 = m20 * m33 - m23 * m30, m20 * m32 - m22 * m30, m20 * m31 - m21 * m30
    det = (m00 * (m11 * a2323 - m12 * a1323 + m13 * a1223) - m01 * (m10 * a2323 - m12 * a0323 + m13 * a0223)
           + m02 * (m10 * a1323 - m11 * a0323 + m13 * a0123) - m03 * (m10 * a1223 - m11 * a0223 + m12 * a0123))
    length = lambda c: _sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3])
    signum = det if det != det else math.copysign(1.0, det)  # f64::signum: NaN stays NaN
    scale = [length(m[0]) * signum, length(m[1]), length(m[2])]
    ax = [[m[c][r] * _fdiv(1.0, scale[c]) for r in range(3)] for c in range(3)]
    (x0, x1, x2), (y0, y1, y2), (z0, z1, z2) = ax
    if z2 <= 0.0:      # Mike Day's branches, as in glam's from_rotation_axes
        dif10, omm22 = y1 - x0, 1.0 - z2
        if dif10 <= 0.0:
            f = omm22 - dif10
            i = _fdiv(0.5, _sqrt(f))
            q = [f * i, (x1 + y0) * i, (x2 + z0) * i, (y2 - z1) * i]
        else:
            f = omm22 + dif10
            i = _fdiv(0.5, _sqrt(f))
            q = [(x1 + y0) * i, f * i, (y2 + z1) * i, (z0 - x2) * i]
    else:
        sum10, opm22 = y1 + x0, 1.0 + z2
        if sum10 <= 0.0:
            f = opm22 - sum10
            i = _fdiv(0.5, _sqrt(f))
            q = [(x2 + z0) * i, (y2 + z1) * i, f * i, (x1 - y0) * i]
        else:
            f = opm22 + sum10
            i = _fdiv(0.5, _sqrt(f))
            q = [(y2 - z1) * i, (z0 - x2) * i, (x1 - y0) * i, f * i]
    return scale, q, [m30, m31, m32]


def compose_trs(scale, q, t):
    """glam 0.13 Mat4::from_scale_rotation_translation with a quaternion xyzw."""
    x, y, z, w = q
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz, wx, wy, wz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2, w * x2, w * y2, w * z2
    cx = [1.0 - (yy + zz), xy + wz, xz - wy, 0.0]
    cy = [xy - wz, 1.0 - (xx + zz), yz + wx, 0.0]
    cz = [xz + wy, yz - wx, 1.0 - (xx + yy), 0.0]
    return [[c * scale[0] for c in cx], [c * scale[1] for c in cy], [c * scale[2] for c in cz], [t[0], t[1], t[2], 1.0]]


def to_f32_colmajor(m) -> np.ndarray:
    """as_f32(): 16 float32 values, column-major."""
    with np.errstate(all="ignore"):
        return np.array([x for col in m for x in col], dtype=np.float64).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# scene model
# ---------------------------------------------------------------------------------------------
def _newtype(v):
    while isinstance(v, ron.Tuple) and v.name is None and len(v.items) == 1:
        v = v.items[0]
    return v


def _code(v):
    while isinstance(v, ron.Tuple) and len(v.items) == 1:
        v = v.items[0]
    assert isinstance(v, str), v
    return v


class OracleScene:
    def __init__(self, path_or_text: str, is_text: bool = False):
        doc = ron.loads(path_or_text) if is_text else ron.load(path_or_text)
        cam = doc["cam"]
        self.cam = dict(look_at=tuple(float(x) for x in cam["look_at"].items), alpha=float(cam["alpha"]), beta=float(cam["beta"]), r=float(cam["r"]),
                        offset_after_material=float(cam["offset_after_material"]))
        self.time = 0.0
        self.total_time = 0.0
        self.uniforms = []   # [name|None, kind, payload]
        self.matrices = []   # [name, named, node]
        self.textures = [(t["name"], _code(t["data"])) for t in _newtype(doc["textures"])]
        vids = doc.get("videos")
        self.videos = [v["name"] for v in (_newtype(vids) if vids is not None else [])]
        for u in _newtype(doc["uniforms"]):
            self.uniforms.append([u["name"], *self._uniform(u["data"])])
        mats = _newtype(doc["matrices"])
        self._mat_by_name = {}
        for m in mats:
            self._mat_by_name[m["name"]] = len(self.matrices)
            self.matrices.append([m["name"], True, None])
        for m in mats:
            self.matrices[self._mat_by_name[m["name"]]][2] = self._matrix(m["data"])
        self.objects = []
        for o in _newtype(doc["objects"]):
            d = o["data"]
            if d.name == "DebugMatrix":
                self.objects.append(dict(name=o["name"], kind="debug", portal=False, m0=self._mref(d.items[0]), m1=-1, code="", sub="Normal"))
                continue
            kind = d["kind"]
            portal = kind.name == "Portal"
            m0 = self._mref(kind.items[0])
            m1 = self._mref(kind.items[1]) if portal else -1
            sub = d.get("in_subspace")
            self.objects.append(dict(name=o["name"], kind="flat" if d.name == "Flat" else "complex", portal=portal, m0=m0, m1=m1,
                                     code=_code(d["is_inside"] if d.name == "Flat" else d["intersect"]), sub=sub.name if sub is not None else "Normal"))
        self.materials = []
        for m in _newtype(doc["materials"]):
            d = m["data"]
            entry = dict(name=m["name"], kind=d.name)
            if d.name == "Simple":
                entry.update(color=[float(x) for x in d["color"].items], normal_coef=float(d["normal_coef"]), grid=bool(d["grid"]), grid_scale=float(d["grid_scale"]),
                             grid_coef=float(d["grid_coef"]), grid2=bool(d.get("grid2", False)), grid3=bool(d.get("grid3", False)))
            elif d.name == "Reflect":
                entry.update(color=[float(x) for x in d["add_to_color"].items])
            elif d.name == "Refract":
                entry.update(color=[float(x) for x in d["add_to_color"].items], refractive_index=float(d["refractive_index"]))
            else:
                entry.update(code=_code(d["code"]))
            self.materials.append(entry)
        im = doc.get("intersection_materials")
        self.intersection_materials = [(m["name"], _code(m["data"])) for m in (_newtype(im) if im is not None else [])]
        self.library = [(m["name"], _code(m["data"])) for m in _newtype(doc["library"])]
        self.cameras = []   # dict(name, look_at=('matrix', idx)|('coord', xyz), alpha, beta, r, in_subspace, free_movement, matrix)
        cams = doc.get("cameras")
        for c in (_newtype(cams) if cams is not None else []):
            self.cameras.append(dict(self._camera(c["data"]), name=c["name"]))
        self.dev_uniforms, self.dev_matrices, self.stages = {}, {}, []
        dev = doc.get("dev_stage")
        if dev is not None:
            for k, v in (dev.get("uniforms") or {}).items():
                self.dev_uniforms[k] = self._uniform(v)
            for k, v in (dev.get("matrices") or {}).items():
                self.dev_matrices[k] = self._matrix(v)
        st = doc.get("animation_stages")
        for item in (_newtype(st) if st is not None else []):
            d = item["data"]

            def change(v, is_matrix):
                if v.name in ("Changed", "ChangedAndToUser"):
                    return ("changed", self._mref(v.items[0]) if is_matrix else self._uref(v.items[0]))
                return ("dev", -1)

            stage = dict(name=item["name"], uniforms=[(k, change(v, False)) for k, v in (d.get("uniforms") or {}).items()],
                         matrices=[(k, change(v, True)) for k, v in (d.get("matrices") or {}).items()], set_cam=None)
            sc = d.get("set_cam")
            outer = ron.unwrap_some(sc) if sc is not None else None
            if outer is not None:
                inner = ron.unwrap_some(outer)
                if inner is None:
                    stage["set_cam"] = -1
                elif inner.name == "Named":
                    stage["set_cam"] = next((k for k, c in enumerate(self.cameras) if c["name"] == inner.items[0]), -1)
                else:
                    self.cameras.append(dict(self._camera(inner.items[0]), name=None))
                    stage["set_cam"] = len(self.cameras) - 1
            self.stages.append(stage)
        # real animations (RealAnimationSer, scene_serialized.rs:556-583; loaded at :1370-1473)
        self.animations = []
        an = doc.get("animations")
        items = list(_newtype(an)) if an is not None else []
        anim_index = {a["name"]: k for k, a in enumerate(items)}

        def stage_ref(v):
            if v.name == "Animation":
                return ("stage", next((k for k, s_ in enumerate(self.stages) if s_["name"] == v.items[0]), None))
            if v.name == "RealAnimation":
                return ("anim", anim_index.get(v.items[0]))
            return ("dev", None)

        def cam_ref(v):
            inner = ron.unwrap_some(v) if v is not None else None
            if inner is None:
                return -1
            if inner.name == "Named":
                return next((k for k, c in enumerate(self.cameras) if c["name"] == inner.items[0]), -1)
            self.cameras.append(dict(self._camera(inner.items[0]), name=None))
            return len(self.cameras) - 1

        def opt(v):
            return None if v is None else ron.unwrap_some(v)

        for a in items:
            d = a["data"]
            entry = dict(name=a["name"], duration=float(d.get("duration", 0.0)), uniforms=[], matrices=[])
            kind, idx = stage_ref(d["animation_stage"])
            entry["base"] = ("dev", None) if idx is None else (kind, idx)
            for key, is_matrix in (("uniforms", False), ("matrices", True)):
                for name, v in (_newtype(d[key]) or {}).items():
                    if isinstance(v, ron.Tuple) and v.name == "Changed":
                        ref = self._mref(v.items[0]) if is_matrix else self._uref(v.items[0])
                        if ref >= 0:
                            entry[key].append((name, ref))
            entry["use_prev_cam"] = bool(d.get("use_prev_cam", False))
            entry["use_start_cam_as_end"] = bool(d.get("use_start_cam_as_end", False))
            entry["cam_start"], entry["cam_end"] = cam_ref(d.get("cam_start")), cam_ref(d.get("cam_end"))
            entry["any_start"], entry["any_end"] = opt(d.get("use_any_cam_as_start")), opt(d.get("use_any_cam_as_end"))
            entry["cam_any_start"] = anim_index.get(opt(d.get("cam_any_start")), -1)
            entry["cam_any_end"] = anim_index.get(opt(d.get("cam_any_end")), -1)
            e = d.get("cam_easing")
            entry["easing"] = e.name if e is not None else "Linear"
            eu = d.get("cam_easing_uniform")
            entry["easing_uniform"] = self._uref(eu) if eu is not None else -1
            self.animations.append(entry)
        self.current_stage, self.current_cam, self.prev_t_raw, self.camera_object_matrix = ("dev", None), -1, 0.0, IDENT
        cs = doc.get("current_stage")
        if cs is not None:
            kind, idx = stage_ref(cs)
            self.current_stage = ("dev", None) if idx is None else (kind, idx)
        self.uniform_alias, self.matrix_alias = {}, {}
        sky = doc.get("skybox")
        self.skybox = ron.unwrap_some(sky) if sky is not None else None
        self._busy_u, self._busy_m = set(), set()
        self._formulas = {}

    # --- parsing helpers
    def _uniform(self, d):
        tag = d.name
        if tag == "Bool":
            return "bool", bool(d.items[0])
        if tag == "Int":
            return "int", int(_newtype(d.items[0])["value"])
        if tag == "Float":
            return "float", float(_newtype(d.items[0])["value"])
        if tag in ("Angle", "Progress"):
            return "float", float(d.items[0])
        if tag in ("Formula", "FormulaInt"):
            return ("formula" if tag == "Formula" else "formula_int"), _code(d.items[0])
        if tag == "TrefoilSpecial":  # 18 x (enabled, value, color), src/gui/uniform.rs:19-20
            return "trefoil", [(bool(e.items[0]), int(e.items[1]), int(e.items[2])) for e in _newtype(d.items[0]).items]
        raise ValueError(f"unknown uniform kind {tag}")

    def _camera(self, d):
        la = d["look_at"]
        look = ("matrix", self._mref(la.items[0])) if la.name == "MatrixCenter" else ("coord", [float(x) for x in la.items[0].items])
        m = d.get("matrix")
        mat = [[float(x) for x in m.items[4 * c:4 * c + 4]] for c in range(4)] if m is not None else IDENT
        return dict(look_at=look, alpha=float(d["alpha"]), beta=float(d["beta"]), r=float(d["r"]), in_subspace=bool(d.get("in_subspace", False)),
                    free_movement=bool(d.get("free_movement", False)), matrix=mat)

    def init_stage(self, name):
        """Scene::init_stage_by_name (scene.rs:1237-1250, animation.rs:171-183).  Returns the camera index the stage selects, or -1."""
        k = next(k for k, s in enumerate(self.stages) if s["name"] == name)
        return self._init(("stage", k))

    def init_animation(self, name):
        """Scene::init_animation_by_name (scene.rs:1254-1267): base stage, then the clip's Changed(Some(..)) entries, then its start camera."""
        k = next(k for k, a in enumerate(self.animations) if a["name"] == name)
        self._init(("anim", k))

    def _init(self, stage, depth=0):
        kind, idx = stage
        if kind == "stage":
            self.current_cam = self._init_plain_stage(self.stages[idx])
        elif kind == "dev":  # DevStageChanging::init_stage: every stored dev value comes back
            for key, val in self.dev_uniforms.items():
                i = self.find_uniform(key)
                if i >= 0:
                    self.uniforms[i][1:] = list(val)
                    self.uniform_alias.pop(i, None)
            for key, val in self.dev_matrices.items():
                i = self._mat_by_name.get(key, -1)
                if i >= 0:
                    self.matrices[i][2] = val
                    self.matrix_alias.pop(i, None)
            self.current_cam = -1
        else:
            a = self.animations[idx]
            if a["base"] != stage and depth < 64:
                self._init(a["base"], depth + 1)
            for key, ref in a["uniforms"]:
                i = self.find_uniform(key)
                if i >= 0:
                    self.uniform_alias.pop(i, None)
                    if i != ref:
                        self.uniform_alias[i] = ref
            for key, ref in a["matrices"]:
                i = self._mat_by_name.get(key, -1)
                if i >= 0:
                    self.matrix_alias.pop(i, None)
                    if i != ref:
                        self.matrix_alias[i] = ref
            cam = self.start_cam(idx)
            if cam >= 0:
                self.current_cam = cam
        self.current_stage = stage
        return self.current_cam

    def start_cam(self, k, depth=0):
        """Scene::get_start_cam (scene.rs:1303-1328)."""
        if k is None or k < 0 or depth > 200:
            return -1
        a = self.animations[k]
        if a["use_prev_cam"]:
            return self.end_cam(k - 1, depth + 1) if k > 0 else -1
        if a["any_start"] is not None:
            j = a["cam_any_start"]
            return -1 if j < 0 else (self.end_cam(j, depth + 1) if a["any_start"] else self.start_cam(j, depth + 1))
        return a["cam_start"]

    def end_cam(self, k, depth=0):
        """Scene::get_end_cam (scene.rs:1330-1344)."""
        if k is None or k < 0 or depth > 200:
            return -1
        a = self.animations[k]
        if a["use_start_cam_as_end"]:
            return self.start_cam(k, depth + 1)
        if a["any_end"] is not None:
            j = a["cam_any_end"]
            return -1 if j < 0 else (self.end_cam(j, depth + 1) if a["any_end"] else self.start_cam(j, depth + 1))
        return a["cam_end"]

    def calculated_cam(self, idx):
        """Cam::get (camera.rs:96-140): look_at from a coordinate or a matrix centre (+0.001)."""
        c = self.cameras[idx]
        if c["look_at"][0] == "matrix":
            m = self.eval_matrix(c["look_at"][1])
            inv_w = _fdiv(1.0, m[3][3])
            look = [m[3][k] * inv_w + 0.001 for k in range(3)]
        else:
            look = list(c["look_at"][1])
        return dict(look_at=look, alpha=c["alpha"], beta=c["beta"], r=c["r"], teleport_matrix=c["matrix"], in_subspace=c["in_subspace"], free_movement=c["free_movement"])

    def update(self, seconds):
        """Scene::update (scene.rs:1353-1493) with run_animations off and no manual-time slider: sets time / total_time,
        returns the clip's interpolated camera (dict, incl. override_matrix) or None."""
        t, total = float(seconds), float(seconds)
        if self.current_stage[0] == "anim":
            k = self.current_stage[1]
            duration = self.animations[k]["duration"]
            if duration > 0.0:
                local = math.fmod(t, duration)
                t = local / duration
                total = sum(a["duration"] for a in self.animations[:k]) + local
            else:
                t, total = 0.0, 0.0
        self.time, self.total_time = t, total
        if self.current_stage[0] != "anim":
            return None
        a = self.animations[k]
        c1, c2 = self.start_cam(k), self.end_cam(k)
        if c1 < 0 or c2 < 0:
            return None
        cam1, cam2 = self.calculated_cam(c1), self.calculated_cam(c2)
        t_raw = math.fmod(self.time, 1.0)
        te = ease(a["easing"], t_raw)
        if a["easing_uniform"] >= 0:
            v = self.eval_uniform(a["easing_uniform"])
            if v is not None:
                v = float(v[1])  # `value.into()`: bool -> 0/1, int -> f64
                te = min(1.0, max(0.0, v if math.isfinite(v) else 0.0))
        out = dict(look_at=[cam1["look_at"][n] + (cam2["look_at"][n] - cam1["look_at"][n]) * te for n in range(3)],
                   alpha=(1.0 - te) * cam1["alpha"] + te * cam2["alpha"], beta=(1.0 - te) * cam1["beta"] + te * cam2["beta"],
                   r=(1.0 - te) * cam1["r"] + te * cam2["r"], in_subspace=cam1["in_subspace"], free_movement=cam1["free_movement"],
                   teleport_matrix=cam1["teleport_matrix"], override_matrix=(t_raw < self.prev_t_raw or t_raw == 0.0))
        self.prev_t_raw = t_raw
        return out

    def _init_plain_stage(self, stage):
        """StageChanging::init_stage (animation.rs:171-183).  set_id copies the replacement's value into the element;
        an alias is the same thing as long as replacements are immutable, which inline stage elements are."""
        for key, (kind, ref) in stage["uniforms"]:
            idx = self.find_uniform(key)
            if idx < 0:
                continue
            if kind == "changed" and ref >= 0:
                self.uniform_alias.pop(idx, None)
                if ref != idx:
                    self.uniform_alias[idx] = ref
            elif kind == "dev" and key in self.dev_uniforms:
                self.uniforms[idx][1:] = list(self.dev_uniforms[key])
                self.uniform_alias.pop(idx, None)
        for key, (kind, ref) in stage["matrices"]:
            idx = self._mat_by_name.get(key, -1)
            if idx < 0:
                continue
            if kind == "changed" and ref >= 0:
                self.matrix_alias.pop(idx, None)
                if ref != idx:
                    self.matrix_alias[idx] = ref
            elif kind == "dev" and key in self.dev_matrices:
                self.matrices[idx][2] = self.dev_matrices[key]
                self.matrix_alias.pop(idx, None)
        return -1 if stage["set_cam"] is None else stage["set_cam"]

    def camera_settings(self, idx):
        """SceneRenderer::update's camera switch (src/main.rs:1442-1478) for scene camera `idx`."""
        c = self.cameras[idx]
        if c["look_at"][0] == "matrix":
            m = self.eval_matrix(c["look_at"][1])
            inv_w = _fdiv(1.0, m[3][3])
            look = [m[3][k] * inv_w + 0.001 for k in range(3)]
        else:
            look = list(c["look_at"][1])
        out = dict(look_at=look, alpha=c["alpha"], beta=c["beta"], r=c["r"], teleport_matrix=c["matrix"], in_subspace=c["in_subspace"], free_movement=c["free_movement"])
        if c["free_movement"]:
            pv = _pos_vec(c["alpha"], c["beta"], c["r"])
            out["look_at"] = [pv[k] + look[k] for k in range(3)]
        return out

    def _uref(self, opt):
        r = ron.unwrap_some(opt)
        if r is None:
            return -1
        if r.name == "Named":
            return self.find_uniform(r.items[0])
        self.uniforms.append([None, *self._uniform(r.items[0])])
        return len(self.uniforms) - 1

    def _param(self, v):
        if v.name == "Value":
            return ("value", float(v.items[0]))
        return ("uniform", self._uref(v.items[0]))

    def _tvec(self, v, names="xyz"):
        return [self._param(v[k]) for k in names]

    def _mref(self, opt):
        r = ron.unwrap_some(opt)
        if r is None:
            return -1
        if r.name == "Named":
            return self._mat_by_name.get(r.items[0], -1)
        idx = len(self.matrices)
        self.matrices.append([f"oracle_inline{idx}", False, None])
        self.matrices[idx][2] = self._matrix(r.items[0])
        return idx

    def _matrix(self, v):
        t = v.name
        if t == "Mul":
            return ("Mul", self._mref(v["to"]), self._mref(v["what"]))
        if t == "Teleport":
            return ("Teleport", self._mref(v["first_portal"]), self._mref(v["second_portal"]), self._mref(v["what"]))
        if t == "Simple":
            return ("Simple", [float(x) for x in v["offset"].items], float(v["scale"]), [float(x) for x in v["rotate"].items], [bool(x) for x in v["mirror"].items])
        if t == "Parametrized":
            return ("Parametrized", self._tvec(v["offset"]), self._tvec(v["rotate"]), self._tvec(v["mirror"]), self._param(v["scale"]))
        if t == "Exact":
            return ("Exact", self._tvec(v["i"]), self._tvec(v["j"]), self._tvec(v["k"]), self._tvec(v["pos"]))
        if t == "ExactFull":
            return ("ExactFull", *[self._tvec(v[c], "xyzw") for c in ("c0", "c1", "c2", "c3")])
        if t == "If":
            return ("If", self._param(v["condition"]), self._mref(v["then"]), self._mref(v["otherwise"]))
        if t == "Inv":
            return ("Inv", self._mref(v.items[0]))
        if t == "Camera":
            return ("Camera",)
        if t == "Lerp":
            return ("Lerp", self._param(v["t"]), self._mref(v["first"]), self._mref(v["second"]))
        if t == "Sqrt":
            return ("Sqrt", self._mref(v.items[0]))
        return ("Unsupported", t)

    # --- evaluation
    def find_uniform(self, name):
        for k, u in enumerate(self.uniforms):
            if u[0] == name:
                return k
        return -1

    def eval_uniform(self, idx):
        """-> ('bool'|'int'|'float', value) or None"""
        hops = 0
        while idx in self.uniform_alias and hops < 64:
            idx, hops = self.uniform_alias[idx], hops + 1
        if idx < 0 or idx >= len(self.uniforms) or idx in self._busy_u:
            return None
        _, kind, payload = self.uniforms[idx]
        if kind in ("bool", "int", "float"):
            return kind, payload
        if kind == "trefoil":
            return None
        if payload not in self._formulas:
            try:
                self._formulas[payload] = F.compile_formula(payload)
            except F.FormulaError:
                self._formulas[payload] = None
        node = self._formulas[payload]
        if node is None:
            return None

        def ns(name, args):
            known, val = F.custom_function(name, args)
            if known:
                return val
            if name == "time":
                return self.time
            if name == "total_time":
                return self.total_time
            r = self.eval_uniform(self.find_uniform(name))
            if r is None:
                return None
            return float(r[1])

        self._busy_u.add(idx)
        try:
            v = F.evaluate(node, ns)
        finally:
            self._busy_u.discard(idx)
        if v is None:
            return None
        if kind == "formula":
            return "float", v
        if math.isnan(v):
            return "int", 0
        return "int", int(max(-2147483648.0, min(2147483647.0, v)))  # Rust `as i32` saturates

    def _p(self, p):
        if p[0] == "value":
            return p[1]
        r = self.eval_uniform(p[1])
        return None if r is None else float(r[1])

    def eval_matrix(self, idx):
        hops = 0
        while idx in self.matrix_alias and hops < 64:
            idx, hops = self.matrix_alias[idx], hops + 1
        if idx < 0 or idx >= len(self.matrices) or idx in self._busy_m:
            return None
        node = self.matrices[idx][2]
        self._busy_m.add(idx)
        try:
            return self._eval_node(node)
        finally:
            self._busy_m.discard(idx)

    def _eval_node(self, node):
        t = node[0]
        if t == "Mul":
            to, what = self.eval_matrix(node[1]), self.eval_matrix(node[2])
            return None if to is None or what is None else m_mul(what, to)
        if t == "Teleport":
            first, second, what = self.eval_matrix(node[1]), self.eval_matrix(node[2]), self.eval_matrix(node[3])
            if first is None or second is None or what is None:
                return None
            return m_mul(m_mul(second, m_inverse(first)), what)
        if t == "Simple":
            _, offset, scale, rot, mirror = node
            s = [scale * (-1.0 if mirror[k] else 1.0) for k in range(3)]
            return from_scale_rotation_translation(s, rot, offset)
        if t == "Parametrized":
            _, offset, rot, mirror, scale = node
            sc = self._p(scale)
            mir = [self._p(x) for x in mirror]
            ro = [self._p(x) for x in rot]
            of = [self._p(x) for x in offset]
            if sc is None or None in mir or None in ro or None in of:
                return None
            return from_scale_rotation_translation([sc * (1.0 - 2.0 * m) for m in mir], ro, of)
        if t == "Exact":
            cols = [[self._p(x) for x in node[k]] for k in (1, 2, 3, 4)]
            if any(None in c for c in cols):
                return None
            return [cols[0] + [0.0], cols[1] + [0.0], cols[2] + [0.0], cols[3] + [1.0]]
        if t == "ExactFull":
            cols = [[self._p(x) for x in node[k]] for k in (1, 2, 3, 4)]
            return None if any(None in c for c in cols) else cols
        if t == "If":
            c = self._p(node[1])
            return None if c is None else self.eval_matrix(node[2] if c > 0.5 else node[3])
        if t == "Inv":
            a = self.eval_matrix(node[1])
            return None if a is None else m_inverse(a)
        if t == "Sqrt":  # src/gui/matrix.rs:606-612: M with M*M = A.  The reference takes what a BFGS minimiser reaches (residual < 1e-4);
            a = self.eval_matrix(node[1])  # this is the exact principal root (Denman-Beavers), the point that minimisation heads for
            if a is None:
                return None
            if getattr(self, "override_sqrt", None) is not None:  # tools/mat_sqrt_bfgs.py: what a minimiser like the reference's reaches, put in the root's place
                return self.override_sqrt
            y, z = a, IDENT
            half = lambda p, q: [[(p[c][r] + q[c][r]) * 0.5 for r in range(4)] for c in range(4)]
            for _ in range(64):
                yn, zn = half(y, m_inverse(z)), half(z, m_inverse(y))
                done = yn == y
                y, z = yn, zn
                if done:
                    break
            sq = m_mul(y, y)
            err = sum((sq[c][r] - a[c][r]) ** 2 for c in range(4) for r in range(4))
            return y if err < 1e-9 else None
        if t == "Lerp":  # src/gui/matrix.rs:614-627 on glam 0.13's to_scale_rotation_translation / Quat::lerp / Vec3::lerp
            tt = self._p(node[1])
            a = None if tt is None else self.eval_matrix(node[2])
            b = None if a is None else self.eval_matrix(node[3])
            if b is None:
                return None
            (fs, fr, ft), (ss, sr, st) = to_scale_rotation_translation(a), to_scale_rotation_translation(b)
            mix3 = lambda p, q: [p[k] + (q[k] - p[k]) * tt for k in range(3)]
            bias = 1.0 if sum(fr[k] * sr[k] for k in range(4)) >= 0.0 else -1.0
            q = [fr[k] + (sr[k] * bias - fr[k]) * tt for k in range(4)]
            inv = _fdiv(1.0, _sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]))
            return compose_trs(mix3(fs, ss), [c * inv for c in q], mix3(ft, st))
        if t == "Camera":
            return self.camera_object_matrix  # formulas_cache.get_camera_matrix(): what SceneRenderer::update last sent
        return None

    # --- what the kernel sees
    def material_ids(self):
        """#define NAME_M (USER_MATERIAL_OFFSET + k), then two per portal object (scene.rs:720-842)."""
        ids, k = {}, 0
        for m in self.materials:
            ids[m["name"] + "_M"] = 10 + k
            k += 1
        for pos, o in enumerate(self.objects):
            if o["kind"] != "debug" and o["portal"] and o["m0"] >= 0 and o["m1"] >= 0:
                ids[f"teleport_{pos}_1_M"] = 10 + k
                ids[f"teleport_{pos}_2_M"] = 10 + k + 1
                k += 2
        return ids

    def scene_uniform_values(self):
        """name -> np.float32 (16,) column-major | np.float32 | np.int32  (Scene::set_uniforms)"""
        out = {}
        passed = []
        for o in self.objects:
            if o["m0"] >= 0:
                passed.append(o["m0"])
            if o["kind"] != "debug" and o["portal"] and o["m1"] >= 0:
                passed.append(o["m1"])
        passed += [k for k, m in enumerate(self.matrices) if m[1]]
        for idx in passed:
            m = self.eval_matrix(idx)
            if m is None:
                continue
            name = self.matrices[idx][0]
            out[name + "_mat"] = to_f32_colmajor(m)
            out[name + "_mat_inv"] = to_f32_colmajor(m_inverse(m))
        for o in self.objects:
            if o["kind"] == "debug" or not o["portal"] or o["m0"] < 0 or o["m1"] < 0:
                continue
            a, b = self.eval_matrix(o["m0"]), self.eval_matrix(o["m1"])
            if a is None or b is None:
                continue
            na, nb = self.matrices[o["m0"]][0], self.matrices[o["m1"]][0]
            out[f"{na}_to_{nb}_mat_teleport"] = to_f32_colmajor(m_mul(b, m_inverse(a)))
            if na != nb:
                out[f"{nb}_to_{na}_mat_teleport"] = to_f32_colmajor(m_mul(a, m_inverse(b)))
        for k, u in enumerate(self.uniforms):
            if u[0] is None:
                continue
            j, hops = k, 0
            while j in self.uniform_alias and hops < 64:
                j, hops = self.uniform_alias[j], hops + 1
            if self.uniforms[j][1] == "trefoil":  # packed as value + enabled*10000 + color*1000 (scene.rs:644-650)
                for i, (enabled, value, color) in enumerate(self.uniforms[j][2]):
                    out[f"ts_{i}_{u[0]}_u"] = np.int32(value + (10000 if enabled else 0) + color * 1000)
                continue
            r = self.eval_uniform(k)
            if r is None:
                continue
            kind, v = r
            out[u[0] + "_u"] = np.float32(v) if kind == "float" else np.int32(1 if v is True else 0 if v is False else v)
        return out


# ---------------------------------------------------------------------------------------------
# camera + builtin uniforms (src/main.rs)
# ---------------------------------------------------------------------------------------------
def ease(kind, t):
    """Easing::ease (src/gui/easing.rs:6-101)."""
    e_in = lambda x: 1.0 - _cos(x * math.pi * 0.5)
    e_io = lambda x: (1.0 - _cos(x * math.pi)) * 0.5
    if kind == "Linear":
        return t
    if kind == "In":
        return e_in(t)
    if kind == "Out":
        return 1.0 - e_in(1.0 - t)
    if kind == "InOut":
        return e_io(t)
    if kind == "InOutFast":
        return e_io(e_io(t))
    if kind == "ElasticOut":
        if t == 0.0 or t == 1.0:
            return t
        return 2.0 ** (-10.0 * t) * _sin((t * 10.0 - 0.75) * (2.0 * math.pi) / 3.0) + 1.0
    raise ValueError(kind)


def _norm3(v):
    inv = _fdiv(1.0, _sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
    return [v[0] * inv, v[1] * inv, v[2] * inv]


def _cross(a, b):
    return [a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]]


def _pos_vec(alpha, beta, r):
    return [_sin(beta) * _cos(alpha) * r, _cos(beta) * r, _sin(beta) * _sin(alpha) * r]


def camera_matrix(look_at, alpha, beta, r, teleport_matrix=None, free_movement=False):
    """RotateAroundCam::get_matrix (src/main.rs:286-304)."""
    pv = _pos_vec(alpha, beta, r)
    pos = [pv[k] + look_at[k] for k in range(3)]
    k = _norm3([look_at[n] - pos[n] for n in range(3)])
    i = _norm3(_cross(k, [0.0, 1.0, 0.0]))
    j = _norm3(_cross(k, i))
    p = list(look_at) if free_movement else pos
    return m_mul(teleport_matrix if teleport_matrix is not None else IDENT, [i + [0.0], j + [0.0], k + [0.0], p + [1.0]])


def builtin_uniforms(scene: OracleScene, width, height, render_depth=100, aa_count=1, aa_start=0, view_angle=None, use_panini=False, panini_param=1.0,
                     camera=None):
    cam = dict(scene.cam)
    if camera:
        cam.update(camera)
    tele = cam.get("teleport_matrix") or IDENT
    m = camera_matrix(cam["look_at"], cam["alpha"], cam["beta"], cam["r"], tele, cam.get("free_movement", False))
    calc_scale = lambda mm: sum(_sqrt(sum(x * x for x in mm[c])) for c in range(3)) / 3.0  # src/main.rs:1325-1333
    scale = calc_scale(m)
    f, i = np.float32, np.int32
    left, right = cam.get("left_eye_matrix") or IDENT, cam.get("right_eye_matrix") or IDENT
    return {
        "_resolution": np.array([width, height], np.float32),
        "_camera": to_f32_colmajor(m), "_camera_left_eye": to_f32_colmajor(left), "_camera_right_eye": to_f32_colmajor(right),
        "_camera_mul_inv": to_f32_colmajor(m_inverse(tele)),
        "_camera_in_subspace": i(1 if cam.get("in_subspace") else 0), "_left_eye_in_subspace": i(1 if cam.get("left_eye_in_subspace") else 0),
        "_right_eye_in_subspace": i(1 if cam.get("right_eye_in_subspace") else 0),
        "_view_angle": f(90.0 / 180.0 * math.pi if view_angle is None else view_angle),
        "_panini_param": f(panini_param), "_use_panini_projection": i(1 if use_panini else 0), "_use_360_camera": i(0), "_use_180_camera": i(0),
        "_ray_tracing_depth": i(render_depth), "_aa_count": i(aa_count), "_aa_start": i(aa_start),
        "_draw_side_by_side": i(0), "_draw_anaglyph": i(0), "_anaglyph_p": f(0.29), "_anaglyph_q": f(0.06), "_anaglyph_mode": i(0),
        "_draw_depth_map": i(0), "_depth_map_min": f(0.0), "_depth_map_max": f(10.0),
        "_offset_after_material": f(cam["offset_after_material"]),
        "_t_start": f(10.0), "_t_end": f(10.0 + 200.0), "_camera_scale": f(scale), "_left_eye_scale": f(calc_scale(left)), "_right_eye_scale": f(calc_scale(right)),
        "_angle_color_disable": i(0), "_grid_disable": i(0), "_black_border_disable": i(0), "_darken_by_distance": i(1), "_teleport_external_ray": i(0),
    }
