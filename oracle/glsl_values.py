"""oracle/glsl_values.py -- GLSL value model over numpy lanes (one array element per pixel).

TEST INFRASTRUCTURE ONLY (CPU oracle).  Nothing under portal_amd/ may import this.

Scalars are numpy arrays (float32 / int32 / bool), 0-d for uniform values or shape (n,) for
per-lane values.  Vec / Mat / Struct are immutable containers of such arrays.  All binary32
arithmetic goes through oracle.glsl_math, i.e. the numerics contract of
portal_amd/csrc/device/ptl_glsl.h restated: vector ops are component-wise; dot / length /
mat*vec / cross are fma chains (lowest component first); v / s multiplies by 1/s.
"""
from __future__ import annotations

import numpy as np

from . import glsl_math as M

F32 = np.float32
I32 = np.int32


class Vec:
    __slots__ = ("c",)

    def __init__(self, comps):
        self.c = tuple(M.f32(x) for x in comps)

    @property
    def n(self):
        return len(self.c)

    def __repr__(self):
        return f"vec{self.n}({', '.join(repr(x) for x in self.c)})"


class Mat:
    """Column-major: cols[i] is column i (a Vec), like GLSL m[i]."""

    __slots__ = ("cols",)

    def __init__(self, cols):
        self.cols = tuple(cols)

    @property
    def n(self):
        return len(self.cols)


class Struct:
    __slots__ = ("tname", "f")

    def __init__(self, tname, fields):
        self.tname = tname
        self.f = dict(fields)

    def with_field(self, name, value):
        f = dict(self.f)
        f[name] = value
        return Struct(self.tname, f)

    def __repr__(self):
        return f"{self.tname}({self.f})"


class Sampler:
    """RGBA8 texture: bilinear, clamp-to-edge, texel centres at +0.5, row 0 at v = 0."""

    def __init__(self, rgba8):
        self.texels = np.ascontiguousarray(rgba8, dtype=np.uint8)
        self.h, self.w = self.texels.shape[:2]


def type_of(v) -> str:
    if isinstance(v, Vec):
        return f"vec{v.n}"
    if isinstance(v, Mat):
        return f"mat{v.n}"
    if isinstance(v, Struct):
        return v.tname
    if isinstance(v, Sampler):
        return "sampler2D"
    a = np.asarray(v)
    if a.dtype == np.bool_:
        return "bool"
    if np.issubdtype(a.dtype, np.integer):
        return "int"
    return "float"


def is_float(v):
    return not isinstance(v, (Vec, Mat, Struct, Sampler)) and np.asarray(v).dtype == F32


def is_int(v):
    return not isinstance(v, (Vec, Mat, Struct, Sampler)) and np.issubdtype(np.asarray(v).dtype, np.integer)


def is_bool(v):
    return not isinstance(v, (Vec, Mat, Struct, Sampler)) and np.asarray(v).dtype == np.bool_


def to_float(v):
    if isinstance(v, (Vec, Mat)):
        return v
    a = np.asarray(v)
    return a if a.dtype == F32 else a.astype(F32)


def to_int(v):
    a = np.asarray(v)
    if a.dtype == F32:  # float -> int truncates toward zero; out-of-range is left to the caller
        with np.errstate(invalid="ignore"):
            return np.trunc(np.where(np.isfinite(a), a, 0)).astype(I32)
    return a.astype(I32)


def select(cond, a, b):
    """Per-lane cond ? a : b, recursively through containers."""
    if isinstance(a, Vec):
        return Vec([np.where(cond, x, y) for x, y in zip(a.c, b.c)])
    if isinstance(a, Mat):
        return Mat([select(cond, x, y) for x, y in zip(a.cols, b.cols)])
    if isinstance(a, Struct):
        return Struct(a.tname, {k: select(cond, a.f[k], b.f[k]) for k in a.f})
    if isinstance(a, Sampler):
        return a
    a_, b_ = np.asarray(a), np.asarray(b)
    out = np.where(cond, a_, b_)
    return out.astype(a_.dtype) if a_.dtype == b_.dtype else out


def take(v, idx):
    """Gather lanes `idx` (index array) from every per-lane array inside v."""
    if isinstance(v, Vec):
        return Vec([take(x, idx) for x in v.c])
    if isinstance(v, Mat):
        return Mat([take(x, idx) for x in v.cols])
    if isinstance(v, Struct):
        return Struct(v.tname, {k: take(x, idx) for k, x in v.f.items()})
    if isinstance(v, Sampler):
        return v
    a = np.asarray(v)
    return a if a.ndim == 0 else a[idx]


def expand(v, n):
    """Broadcast uniform (0-d) leaves to n lanes."""
    if isinstance(v, Vec):
        return Vec([expand(x, n) for x in v.c])
    if isinstance(v, Mat):
        return Mat([expand(x, n) for x in v.cols])
    if isinstance(v, Struct):
        return Struct(v.tname, {k: expand(x, n) for k, x in v.f.items()})
    if isinstance(v, Sampler):
        return v
    a = np.asarray(v)
    return np.broadcast_to(a, (n,)).copy() if a.ndim == 0 else a


# ---------------------------------------------------------------------------------------
# constructors
# ---------------------------------------------------------------------------------------
def _flatten(args):
    out = []
    for a in args:
        if isinstance(a, Vec):
            out.extend(a.c)
        elif isinstance(a, Mat):
            for col in a.cols:
                out.extend(col.c)
        else:
            out.append(to_float(a))
    return out


def make_vec(n, args):
    if len(args) == 1 and not isinstance(args[0], (Vec, Mat)):
        x = to_float(args[0])
        return Vec([x] * n)
    flat = _flatten(args)
    if len(flat) < n:
        raise TypeError(f"vec{n} constructor: not enough components")
    return Vec(flat[:n])


def make_mat(n, args):
    if len(args) == 1 and isinstance(args[0], Mat):
        m = args[0]
        cols = []
        for j in range(n):
            comps = []
            for i in range(n):
                if j < m.n and i < m.n:
                    comps.append(m.cols[j].c[i])
                else:
                    comps.append(F32(1.0) if i == j else F32(0.0))
            cols.append(Vec(comps))
        return Mat(cols)
    if len(args) == 1 and not isinstance(args[0], Vec):
        d = to_float(args[0])
        return Mat([Vec([d if i == j else F32(0.0) for i in range(n)]) for j in range(n)])
    flat = _flatten(args)
    if len(flat) != n * n:
        raise TypeError(f"mat{n} constructor: need {n*n} components, got {len(flat)}")
    return Mat([Vec(flat[j * n:(j + 1) * n]) for j in range(n)])


SWIZZLE_SETS = ("xyzw", "rgba", "stpq")


def swizzle_indices(name):
    for s in SWIZZLE_SETS:
        if all(ch in s for ch in name):
            return [s.index(ch) for ch in name]
    return None


# ---------------------------------------------------------------------------------------
# arithmetic (GLSL operator semantics + the numerics contract)
# ---------------------------------------------------------------------------------------
def _int_div(a, b):
    a, b = np.asarray(a, I32), np.asarray(b, I32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(b != 0, np.trunc(a.astype(np.float64) / np.where(b != 0, b, 1)).astype(I32), 0)
    return q.astype(I32)


def _scalar_binop(op, a, b):
    if is_int(a) and is_int(b):
        a, b = np.asarray(a, I32), np.asarray(b, I32)
        if op == "+":
            return (a + b).astype(I32)
        if op == "-":
            return (a - b).astype(I32)
        if op == "*":
            return (a * b).astype(I32)
        if op == "/":
            return _int_div(a, b)
        if op == "%":
            return (a - _int_div(a, b) * b).astype(I32)
    a, b = to_float(a), to_float(b)
    if op == "+":
        return M.add(a, b)
    if op == "-":
        return M.sub(a, b)
    if op == "*":
        return M.mul(a, b)
    if op == "/":
        return M.div(a, b)
    raise TypeError(f"bad scalar operator {op}")


def mat_vec(m: Mat, v: Vec) -> Vec:
    out = []
    for i in range(m.n):
        acc = M.term0(m.cols[0].c[i], v.c[0])  # contract 2: the chain starts from +0 (ptl_glsl.h `ptl_term`); contract 1: the bare product
        M.note_matrix_term(m.cols[0].c[i], v.c[0], True)
        for j in range(1, m.n):
            before = acc
            acc = M.fma(m.cols[j].c[i], v.c[j], acc)
            M.note_matrix_term(m.cols[j].c[i], v.c[j], False, before)
        out.append(acc)
    return Vec(out)


def dot(a: Vec, b: Vec):
    acc = M.term0(a.c[0], b.c[0])
    for i in range(1, a.n):
        acc = M.fma(a.c[i], b.c[i], acc)
    return acc


def binop(op, a, b):
    if isinstance(a, Mat) or isinstance(b, Mat):
        if op == "*":
            if isinstance(a, Mat) and isinstance(b, Vec):
                return mat_vec(a, b)
            if isinstance(a, Vec) and isinstance(b, Mat):
                return Vec([dot(a, col) for col in b.cols])
            if isinstance(a, Mat) and isinstance(b, Mat):
                return Mat([mat_vec(a, col) for col in b.cols])
            if isinstance(a, Mat):
                return Mat([binop("*", col, b) for col in a.cols])
            return Mat([binop("*", a, col) for col in b.cols])
        if op in "+-" and isinstance(a, Mat) and isinstance(b, Mat):
            return Mat([binop(op, x, y) for x, y in zip(a.cols, b.cols)])
        raise TypeError(f"unsupported matrix operator {op}")
    if isinstance(a, Vec) and isinstance(b, Vec):
        return Vec([_scalar_binop(op, x, y) for x, y in zip(a.c, b.c)])
    if isinstance(a, Vec):
        s = to_float(b)
        if op == "/":  # contract: vec / scalar multiplies by the correctly-rounded reciprocal
            inv = M.div(F32(1.0), s)
            return Vec([M.mul(x, inv) for x in a.c])
        return Vec([_scalar_binop(op, x, s) for x in a.c])
    if isinstance(b, Vec):
        s = to_float(a)
        return Vec([_scalar_binop(op, s, y) for y in b.c])
    return _scalar_binop(op, a, b)


def neg(a):
    if isinstance(a, Vec):
        return Vec([M.neg(x) for x in a.c])
    if isinstance(a, Mat):
        return Mat([neg(c) for c in a.cols])
    if is_int(a):
        return (-np.asarray(a, I32)).astype(I32)
    return M.neg(a)


def compare(op, a, b):
    if isinstance(a, Vec) and op in ("==", "!="):
        eq = np.logical_and.reduce([M.eq(x, y) for x, y in zip(a.c, b.c)])
        return eq if op == "==" else ~eq
    if is_bool(a) and is_bool(b):
        return (np.asarray(a) == np.asarray(b)) if op == "==" else (np.asarray(a) != np.asarray(b))
    if is_int(a) and is_int(b):
        a, b = np.asarray(a), np.asarray(b)
    else:
        a, b = to_float(a), to_float(b)
        M._count(1, "cmp", (a, b,))
    return {"<": np.less, ">": np.greater, "<=": np.less_equal, ">=": np.greater_equal, "==": np.equal, "!=": np.not_equal}[op](a, b)


# geometric builtins (contract section "geometric")
def length(a):
    if isinstance(a, Vec):
        return M.sqrt(dot(a, a))
    return M.absf(a)


def normalize(a: Vec) -> Vec:
    return binop("/", a, length(a))


def cross(a: Vec, b: Vec) -> Vec:
    if M.CONTRACT == 1:
        x = M.fma(a.c[1], b.c[2], M.neg(M.mul(a.c[2], b.c[1])))
        y = M.fma(a.c[2], b.c[0], M.neg(M.mul(a.c[0], b.c[2])))
        z = M.fma(a.c[0], b.c[1], M.neg(M.mul(a.c[1], b.c[0])))
        return Vec([x, y, z])
    x = M.fma(a.c[1], b.c[2], M.term0(M.neg(a.c[2]), b.c[1]))
    y = M.fma(a.c[2], b.c[0], M.term0(M.neg(a.c[0]), b.c[2]))
    z = M.fma(a.c[0], b.c[1], M.term0(M.neg(a.c[1]), b.c[0]))
    return Vec([x, y, z])


def map1(fn, a):
    if isinstance(a, Vec):
        return Vec([fn(x) for x in a.c])
    return fn(to_float(a))


def map2(fn, a, b):
    if isinstance(a, Vec) and isinstance(b, Vec):
        return Vec([fn(x, y) for x, y in zip(a.c, b.c)])
    if isinstance(a, Vec):
        return Vec([fn(x, to_float(b)) for x in a.c])
    if isinstance(b, Vec):
        return Vec([fn(to_float(a), y) for y in b.c])
    return fn(to_float(a), to_float(b))


def map3(fn, a, b, c):
    if isinstance(a, Vec) or isinstance(b, Vec) or isinstance(c, Vec):
        n = next(v.n for v in (a, b, c) if isinstance(v, Vec))
        get = lambda v, i: v.c[i] if isinstance(v, Vec) else to_float(v)
        return Vec([fn(get(a, i), get(b, i), get(c, i)) for i in range(n)])
    return fn(to_float(a), to_float(b), to_float(c))


def texture(s: Sampler, uv: Vec) -> Vec:
    """Contract: x = u*W - 0.5 (NaN -> 0), clamped to [-1, W]; floor/fract; 4 clamped fetches;
    channel = byte / 255; mix(mix(c00, c10, fx), mix(c01, c11, fx), fy)."""
    if s is None:  # unbound sampler (contract: texels == nullptr)
        n = np.broadcast(np.asarray(uv.c[0]), np.asarray(uv.c[1])).shape
        return Vec([np.zeros(n, F32), np.zeros(n, F32), np.zeros(n, F32), np.ones(n, F32)])
    w, h = F32(s.w), F32(s.h)
    x = M.sub(M.mul(uv.c[0], w), F32(0.5))
    y = M.sub(M.mul(uv.c[1], h), F32(0.5))
    x = np.where(np.isnan(x), F32(0), x).astype(F32)
    y = np.where(np.isnan(y), F32(0), y).astype(F32)
    x = M.clamp(x, F32(-1.0), w)
    y = M.clamp(y, F32(-1.0), h)
    x0, y0 = M.floor(x), M.floor(y)
    fx, fy = M.sub(x, x0), M.sub(y, y0)
    ix, iy = x0.astype(I32), y0.astype(I32)

    def fetch(jx, jy):
        jx = np.clip(jx, 0, s.w - 1)
        jy = np.clip(jy, 0, s.h - 1)
        t = s.texels[jy, jx].astype(F32)  # (..., 4)
        return Vec([M.div(t[..., k], F32(255.0)) for k in range(4)])

    c00, c10, c01, c11 = fetch(ix, iy), fetch(ix + 1, iy), fetch(ix, iy + 1), fetch(ix + 1, iy + 1)
    a = map3(M.mix, c00, c10, fx)
    b = map3(M.mix, c01, c11, fx)
    return map3(M.mix, a, b, fy)
