"""oracle/glsl_math.py -- the binary32 numerics contract, restated in numpy.

TEST INFRASTRUCTURE ONLY (CPU oracle).  Nothing under portal_amd/ may import this.

The reference leaves the precision of GLSL builtins to the GL driver (src/library.glsl:5-7
`precision highp float`; GLSL ES 3.00 section 4.5.1 / chapter 8).  The MI355X build pins them
down as a *contract* (portal_amd/csrc/device/ptl_glsl.h): every builtin is a fixed sequence
of correctly-rounded binary32 operations (+ - * / sqrt fma floor rint).  This file states the
same contract a second time, independently, over numpy float32 arrays (one array element per
pixel / lane), so that the oracle and the kernel can be compared bit for bit.

numpy has no fused multiply-add; `fma` below emulates the single rounding exactly: the
product of two binary32 values is exact in binary64, the sum is rounded to binary64 with
round-to-odd (via the TwoSum error term), and the final rounding to binary32 is then the
correct single rounding (53 >= 2*24 + 2 bits).

Every primitive adds to the global operation counter `STATS` (weighted by the number of
active lanes set through `set_active`) -- that is how per-scene "algorithmic flops per bounce"
 are measured.  Counting rule: + - * / sqrt floor rint min max compare = 1,
fma = 2.
"""
from __future__ import annotations

from decimal import Decimal
from fractions import Fraction

import numpy as np

F32 = np.float32
F64 = np.float64

# ---------------------------------------------------------------------------------------
# operation counter
# ---------------------------------------------------------------------------------------
STATS = {"flops": 0.0, "div": 0.0, "sqrt": 0.0, "fma": 0.0,
         # the same, restricted to operations with at least one lane-varying (per-ray) operand: what a kernel with every
         # scene uniform baked in (FLAG_SPECIALIZE_ALL) still has to execute -- uniform-only subexpressions fold at JIT time
         "flops_varying": 0.0, "div_varying": 0.0, "sqrt_varying": 0.0, "fma_varying": 0.0,
         # of `flops_varying`: terms of matrix products whose matrix element is a uniform zero -- a build with the matrices baked in
         # skips them (ptl_glsl.h `ptl_mterm`), so they are not executed work there (tools/count_flops.py subtracts them)
         "zero_term_flops_varying": 0.0, "unit_term_flops_varying": 0.0, "known_w_term_flops_varying": 0.0,
         # of `flops`: comparisons, min / max, step -- VALU instructions without arithmetic (reported beside the totals, tools/count_flops.py)
         "cmp": 0.0, "cmp_varying": 0.0}
_active = 1.0


COUNT_VARYING = False  # tools/count_flops.py switches it on; the per-operand test below costs about as much as the arithmetic


def _lane_varying(x) -> bool:
    """True when the operand differs between lanes.  Decided by VALUE, not by shape: the interpreter broadcasts uniforms to
    lane arrays when it stores them in variables, so a lane-shaped operand whose elements are all the same bit pattern is
    (with random pixels) a ray-independent quantity -- the kind a kernel with baked uniforms folds at compile time."""
    a = np.asarray(x)
    if a.ndim == 0 or a.size <= 1:
        return False
    flat = np.ascontiguousarray(a).reshape(-1)
    if flat.dtype == F32:
        flat = flat.view(np.uint32)
    return bool((flat != flat[0]).any())


def set_active(n: float) -> None:
    global _active
    _active = float(n)


def reset_stats() -> None:
    for k in STATS:
        STATS[k] = 0.0


def _count(n: float, kind: str | None = None, operands=()) -> None:
    STATS["flops"] += n * _active
    if kind:
        STATS[kind] += _active
    if not COUNT_VARYING:
        return
    for x in operands:
        if _lane_varying(x):
            STATS["flops_varying"] += n * _active
            if kind:
                STATS[kind + "_varying"] += _active
            break


# ---------------------------------------------------------------------------------------
# constants: decimal literal -> nearest binary32 with ONE rounding (what `1.23e-4f` means in C++)
# ---------------------------------------------------------------------------------------
def lit(text: str) -> np.float32:
    text = text.strip().rstrip("fF")
    if text.lower().startswith(("0x", "-0x", "+0x")):
        return F32(float.fromhex(text))  # hex floats in the contract are exactly representable
    exact = Fraction(Decimal(text))
    guess = F32(float(exact))
    if not np.isfinite(guess):
        return guess
    best, best_err = guess, abs(Fraction(float(guess)) - exact)
    for cand in (np.nextafter(guess, F32(-np.inf)), np.nextafter(guess, F32(np.inf))):
        err = abs(Fraction(float(cand)) - exact)
        if err < best_err or (err == best_err and (int(np.asarray(cand).view(np.uint32)) & 1) == 0):
            best, best_err = cand, err
    return F32(best)


def f32(x) -> np.ndarray:
    return np.asarray(x, dtype=F32)


# ---------------------------------------------------------------------------------------
# primitives (all correctly rounded in binary32)
# ---------------------------------------------------------------------------------------
_err = dict(over="ignore", invalid="ignore", divide="ignore", under="ignore")


def add(a, b):
    _count(1, None, (a, b,))
    with np.errstate(**_err):
        return np.add(a, b, dtype=F32)


def sub(a, b):
    _count(1, None, (a, b,))
    with np.errstate(**_err):
        return np.subtract(a, b, dtype=F32)


def mul(a, b):
    _count(1, None, (a, b,))
    with np.errstate(**_err):
        return np.multiply(a, b, dtype=F32)


# The numerics contract of portal_amd/csrc/device/ptl_glsl.h comes in two versions (see its "CONTRACT 2" comment):
#   2 (default since round 3): 1/x correctly rounded with the extremes flushed (|x| < 2^-126 -> +-inf, |x| > 2^126 -> +-0),
#     a / b = a * (1/b), sqrt correctly rounded with |x| < 2^-100 -> +0;
#   1 (rounds 1-2, the product's FLAG_EXACT_CR / `--exact-cr`): IEEE division and square root on every input.
CONTRACT = 2


def set_contract(version: int) -> int:
    """Select the contract version; returns the previous one."""
    global CONTRACT
    if version not in (1, 2):
        raise ValueError("contract version must be 1 or 2")
    previous, CONTRACT = CONTRACT, version
    return previous


def rcp(x):
    """1/x of the active contract (uncounted: callers count the division it belongs to)."""
    x = f32(x)
    with np.errstate(**_err):
        y = np.divide(F32(1), x, dtype=F32)
        if CONTRACT == 1:
            return y
        m = np.abs(x)
        y = np.where(m < F32(2.0 ** -126), np.copysign(F32(np.inf), x), y)
        y = np.where(m > F32(2.0 ** 126), np.copysign(F32(0), x), y)
    return y.astype(F32)


def div(a, b):
    _count(1, "div", (a, b,))
    with np.errstate(**_err):
        if CONTRACT == 1:
            return np.divide(a, b, dtype=F32)
        return np.multiply(f32(a), rcp(b), dtype=F32)


def neg(a):
    return np.negative(f32(a))


def sqrt(a):
    _count(1, "sqrt", (a,))
    with np.errstate(**_err):
        a = f32(a)
        r = np.sqrt(a)
        if CONTRACT == 1:
            return r
        return np.where(np.abs(a) < F32(2.0 ** -100), F32(0), r).astype(F32)


def absf(a):
    # not counted as an operation: |x| is a sign-bit edit -- a free source modifier of the instruction that consumes it on gfx950 -- and
    # counting it pushed `roofline.frac` above the hardware's own instruction ceiling (bench.py `frac_ceiling_valu_plus_fma`)
    return np.abs(f32(a))


def floor(a):
    _count(1, None, (a,))
    return np.floor(f32(a))


def ceil(a):
    _count(1, None, (a,))
    return np.ceil(f32(a))


def trunc(a):
    _count(1, None, (a,))
    return np.trunc(f32(a))


def rint(a):
    _count(1, None, (a,))
    return np.rint(f32(a))


def fma(a, b, c):
    """round_binary32(a*b + c) with a single rounding."""
    _count(2, "fma", (a, b, c,))
    a64, b64, c64 = np.asarray(a, F32).astype(F64), np.asarray(b, F32).astype(F64), np.asarray(c, F32).astype(F64)
    with np.errstate(**_err):
        p = a64 * b64                     # exact
        s = p + c64                       # rounded to binary64 (nearest even)
        bb = s - p
        e = (p - (s - bb)) + (c64 - bb)   # TwoSum: s + e == p + c64 exactly
        s = np.atleast_1d(s).copy()
        e = np.broadcast_to(np.atleast_1d(e), s.shape)
        bits = s.view(np.int64)
        fix = np.isfinite(s) & (e != 0) & ((bits & 1) == 0) & (s != 0)
        # move one ulp toward the true sum: magnitude grows iff e has the sign of s
        grow = (e > 0) == (s > 0)
        bits[fix & grow] += 1
        bits[fix & ~grow] -= 1
        out = s.astype(F32)
    shape = np.broadcast(np.asarray(a), np.asarray(b), np.asarray(c)).shape
    return out.reshape(shape) if shape != out.shape else out


def term0(a, b):
    """First term of a dot / matrix product chain.  Contract 2: fma(a, b, +0) -- the product, with an exact-zero result made +0
    (portal_amd/csrc/device/ptl_glsl.h `ptl_term0`); contract 1: the bare product.  Counted as one multiplication."""
    _count(1, None, (a, b,))
    with np.errstate(**_err):
        if CONTRACT == 1:
            return np.multiply(a, b, dtype=F32)
        # fma(a, b, +0): the exact product (binary64 holds it) plus +0, rounded once.  An exact zero product gives +0 whatever its
        # sign; a non-zero product keeps its sign even when it underflows to zero.
        p = np.asarray(a, F32).astype(F64) * np.asarray(b, F32).astype(F64)
        return np.where(p == 0, F64(0), p).astype(F32)


def note_matrix_term(m, v, first: bool, acc=None) -> None:
    """Bookkeeping only (COUNT_VARYING): a matrix-product term with matrix element `m`, vector component `v` and the accumulator it is
    added to.  What a kernel that knows the matrix (baked, or its pattern) and that rays are affine does NOT execute of the 2 operations
    (1 for the first term of a chain) this module counts for the term:
      zero_term      m is a ray-independent zero: the term is skipped                                   (round 3, ptl_glsl.h `ptl_mterm`)
      unit_term      m is a ray-independent +-1 and v varies: fma(+-1, v, acc) is ONE add / sub          (round 4 / 5: the multiplication is not executed)
      known_w_term   v is a ray-independent 0 (a direction's w) under a varying accumulator: skipped;    (round 5, PTL_AFFINE_RAYS)
                     v is a ray-independent 1 (an origin's w), m ray-independent: `acc + m`, one add"""
    if not COUNT_VARYING:
        return

    def uniform_value(x):
        a = np.asarray(x)
        if a.ndim == 0:
            return float(a)
        if a.size > 0 and not _lane_varying(a):
            return float(a.reshape(-1)[0])
        return None

    mv, vv = uniform_value(m), uniform_value(v)
    full = 1.0 if first else 2.0
    if mv == 0.0 and _lane_varying(v):
        STATS["zero_term_flops_varying"] += full * _active
    elif mv is not None and abs(mv) == 1.0 and _lane_varying(v) and not first:
        STATS["unit_term_flops_varying"] += 1.0 * _active
    elif vv is not None and mv is not None and mv != 0.0 and acc is not None and _lane_varying(acc) and not first:
        if vv == 0.0:
            STATS["known_w_term_flops_varying"] += 2.0 * _active
        elif vv == 1.0:
            STATS["known_w_term_flops_varying"] += 1.0 * _active


def lt(a, b):
    _count(1, "cmp", (a, b,))
    return np.less(a, b)


def gt(a, b):
    _count(1, "cmp", (a, b,))
    return np.greater(a, b)


def le(a, b):
    _count(1, "cmp", (a, b,))
    return np.less_equal(a, b)


def ge(a, b):
    _count(1, "cmp", (a, b,))
    return np.greater_equal(a, b)


def eq(a, b):
    _count(1, "cmp", (a, b,))
    return np.equal(a, b)


def ne(a, b):
    _count(1, "cmp", (a, b,))
    return np.not_equal(a, b)


def select(c, a, b):
    return np.where(c, a, b)


# --- GLSL scalar builtins (contract: ptl_glsl.h "scalar primitives") -----------------------
def fmin(a, b):  # min(a,b) = b < a ? b : a
    _count(1, "cmp", (a, b,))
    return np.where(np.less(b, a), f32(b), f32(a)).astype(F32)


def fmax(a, b):  # max(a,b) = a < b ? b : a
    _count(1, "cmp", (a, b,))
    return np.where(np.less(a, b), f32(b), f32(a)).astype(F32)


def clamp(x, lo, hi):
    return fmin(fmax(x, lo), hi)


def fract(x):
    return sub(x, floor(x))


def mod(x, y):
    return sub(x, mul(y, floor(div(x, y))))


def sign(x):
    x = f32(x)
    _count(2, None, (x,))
    return np.where(x > 0, F32(1), np.where(x < 0, F32(-1), F32(0))).astype(F32)


def step(edge, x):
    _count(1, "cmp", (edge, x,))
    return np.where(np.less(x, edge), F32(0), F32(1)).astype(F32)


def mix(a, b, t):
    return add(mul(a, sub(F32(1), t)), mul(b, t))


def smoothstep(e0, e1, x):
    t = clamp(div(sub(x, e0), sub(e1, e0)), F32(0), F32(1))
    return mul(mul(t, t), sub(F32(3), mul(F32(2), t)))


def inversesqrt(x):
    return div(F32(1), sqrt(x))


RAD = lit("0x1.1df46ap-6")
DEG = lit("0x1.ca5dc2p+5")


def radians(d):
    return mul(d, RAD)


def degrees(r):
    return mul(r, DEG)


# --- sin / cos / tan -----------------------------------------------------------------------
TWO_OVER_PI = lit("0x1.45f306p-1")
PIO2_HI, PIO2_MID, PIO2_LO = lit("0x1.921fb6p+0"), lit("-0x1.777a5cp-25"), lit("-0x1.ee59dap-50")
S1, S2, S3 = lit("-1.6666654611e-1"), lit("8.3321608736e-3"), lit("-1.9515295891e-4")
C1, C2, C3 = lit("4.166664568298827e-2"), lit("-1.388731625493765e-3"), lit("2.443315711809948e-5")
PI_F, PIO2_F, PIO4_F = lit("0x1.921fb6p+1"), lit("0x1.921fb6p+0"), lit("0x1.921fb6p-1")


def _sincos_kernel(x):
    x = f32(x)
    k = rint(mul(x, TWO_OVER_PI))
    nk = neg(k)
    r = fma(nk, PIO2_HI, x)
    r = fma(nk, PIO2_MID, r)
    r = fma(nk, PIO2_LO, r)
    q = sub(k, mul(F32(4), floor(mul(k, F32(0.25)))))
    z = mul(r, r)
    ps = fma(z, S3, S2)
    ps = fma(z, ps, S1)
    pc = fma(z, C3, C2)
    pc = fma(z, pc, C1)
    s = fma(mul(r, z), ps, r)
    c = fma(mul(z, z), pc, fma(F32(-0.5), z, F32(1)))
    return s, c, q


def sin(x):
    s, c, q = _sincos_kernel(x)
    v = np.where((q == 1) | (q == 3), c, s)
    return np.where((q == 2) | (q == 3), -v, v).astype(F32)


def cos(x):
    s, c, q = _sincos_kernel(x)
    v = np.where((q == 1) | (q == 3), s, c)
    return np.where((q == 1) | (q == 2), -v, v).astype(F32)


def tan(x):
    return div(sin(x), cos(x))


# --- atan / atan2 --------------------------------------------------------------------------
T3P8, TP8 = lit("2.414213562373095"), lit("0.4142135623730950")
A1, A2, A3, A4 = lit("8.05374449538e-2"), lit("-1.38776856032e-1"), lit("1.99777106478e-1"), lit("-3.33329491539e-1")


def atan(x0):
    x0 = f32(x0)
    x = absf(x0)
    big = gt(x, T3P8)
    mid = ~big & gt(x, TP8)
    y = np.where(big, PIO2_F, np.where(mid, PIO4_F, F32(0))).astype(F32)
    xr = np.where(big, neg(div(F32(1), x)), np.where(mid, div(sub(x, F32(1)), add(x, F32(1))), x)).astype(F32)
    z = mul(xr, xr)
    p = fma(z, A1, A2)
    p = fma(z, p, A3)
    p = fma(z, p, A4)
    y = add(y, fma(mul(p, z), xr, xr))
    return np.where(x0 < 0, -y, y).astype(F32)


def atan2(y, x):
    y, x = f32(y), f32(x)
    w = np.where(x < 0, np.where(y < 0, -PI_F, PI_F), F32(0)).astype(F32)
    general = add(w, atan(div(y, x)))
    axis = np.where(y > 0, PIO2_F, np.where(y < 0, -PIO2_F, F32(0))).astype(F32)
    return np.where(x == 0, axis, general).astype(F32)


# --- asin / acos ---------------------------------------------------------------------------
P1, P2, P3, P4, P5 = lit("4.2163199048e-2"), lit("2.4181311049e-2"), lit("4.5470025998e-2"), lit("7.4953002686e-2"), lit("1.6666752422e-1")


def asin(x0):
    x0 = f32(x0)
    a = absf(x0)
    big = gt(a, F32(0.5))
    z = np.where(big, mul(F32(0.5), sub(F32(1), a)), mul(a, a)).astype(F32)
    x = np.where(big, sqrt(z), a).astype(F32)
    p = fma(z, P1, P2)
    p = fma(z, p, P3)
    p = fma(z, p, P4)
    p = fma(z, p, P5)
    r = fma(mul(p, z), x, x)
    r = np.where(big, sub(PIO2_F, add(r, r)), r).astype(F32)
    return np.where(x0 < 0, -r, r).astype(F32)


def acos(x):
    x = f32(x)
    lo = sub(PI_F, mul(F32(2), asin(sqrt(mul(F32(0.5), add(F32(1), x))))))
    hi = mul(F32(2), asin(sqrt(mul(F32(0.5), sub(F32(1), x)))))
    midv = sub(PIO2_F, asin(x))
    return np.where(x < F32(-0.5), lo, np.where(x > F32(0.5), hi, midv)).astype(F32)


# --- exp2 / log2 / exp / log / pow ------------------------------------------------------------
def _pow2i(e):
    return ((np.asarray(e, np.int32) + 127) << 23).astype(np.int32).view(F32)


def _scale2(z, n):
    n = np.asarray(n, np.int32)
    h = n >> 1
    return mul(mul(z, _pow2i(h)), _pow2i(n - h))


E2 = [lit(c) for c in ("1.535336188319500e-4", "1.339887440266574e-3", "9.618437357674640e-3", "5.550332471162809e-2", "2.402264791363012e-1", "6.931472028550421e-1")]


def exp2(x):
    x = f32(x)
    xs = np.where(np.isfinite(x) & (x <= 128) & (x >= -150), x, F32(0)).astype(F32)
    n = rint(xs)
    f = sub(xs, n)
    p = fma(f, E2[0], E2[1])
    for c in E2[2:]:
        p = fma(f, p, c)
    r = _scale2(fma(f, p, F32(1)), n.astype(np.int32))
    r = np.where(x > 128, F32(np.inf), np.where(x < -150, F32(0), r))
    return np.where(np.isnan(x), x, r).astype(F32)


LOGC = [lit(c) for c in ("7.0376836292e-2", "-1.1514610310e-1", "1.1676998740e-1", "-1.2420140846e-1", "1.4249322787e-1", "-1.6668057665e-1", "2.0000714765e-1", "-2.4999993993e-1", "3.3333331174e-1")]
SQRTHF = lit("0.707106781186547524")
LN2_HI, LN2_LO = lit("0.693359375"), lit("-2.12194440e-4")
LOG2EA = lit("0.44269504088896340735992")


def _log_split(x):
    x = f32(x)
    small = x < lit("0x1p-126")
    xs = np.where(small, mul(x, lit("0x1p+24")), x).astype(F32)
    e_adj = np.where(small, F32(-24), F32(0)).astype(F32)
    b = np.atleast_1d(xs).view(np.int32)
    e = ((b >> 23) & 0xFF) - 126
    m = ((b & 0x007FFFFF) | 0x3F000000).astype(np.int32).view(F32)
    lowm = m < SQRTHF
    e = np.where(lowm, e - 1, e)
    m = np.where(lowm, add(m, m), m).astype(F32)
    return sub(m, F32(1)), add(e.astype(F32), e_adj)


def _log_poly(m):
    p = fma(m, LOGC[0], LOGC[1])
    for c in LOGC[2:]:
        p = fma(m, p, c)
    z = mul(m, m)
    return fma(F32(-0.5), z, mul(mul(p, m), z))


def _log_special(x, r):
    x = np.atleast_1d(f32(x))
    r = np.where(x == np.inf, x, r)
    r = np.where(x == 0, F32(-np.inf), r)
    r = np.where(np.isnan(x) | (x < 0), F32(np.nan), r)
    return r.astype(F32)


def log(x):
    xs = np.where(np.isfinite(f32(x)) & (f32(x) > 0), f32(x), F32(1)).astype(F32)
    m, e = _log_split(xs)
    y = _log_poly(m)
    y = fma(e, LN2_LO, y)
    return _log_special(x, fma(e, LN2_HI, add(m, y)))


def log2(x):
    xs = np.where(np.isfinite(f32(x)) & (f32(x) > 0), f32(x), F32(1)).astype(F32)
    m, e = _log_split(xs)
    y = _log_poly(m)
    z = mul(y, LOG2EA)
    z = fma(m, LOG2EA, z)
    z = add(z, y)
    z = add(z, m)
    return _log_special(x, add(z, e))


EXPC = [lit(c) for c in ("1.9875691500e-4", "1.3981999507e-3", "8.3334519073e-3", "4.1665795894e-2", "1.6666665459e-1", "5.0000001201e-1")]
LOG2E = lit("0x1.715476p+0")
EXP_HI, EXP_LO = lit("88.72283905206835"), lit("-103.972076416015625")


def exp(x):
    x = f32(x)
    xs = np.where(np.isfinite(x) & (x <= EXP_HI) & (x >= EXP_LO), x, F32(0)).astype(F32)
    n = floor(fma(xs, LOG2E, F32(0.5)))
    r = fma(neg(n), LN2_HI, xs)
    r = fma(neg(n), LN2_LO, r)
    p = fma(r, EXPC[0], EXPC[1])
    for c in EXPC[2:]:
        p = fma(r, p, c)
    y = add(fma(p, mul(r, r), r), F32(1))
    out = _scale2(y, n.astype(np.int32))
    out = np.where(x > EXP_HI, F32(np.inf), np.where(x < EXP_LO, F32(0), out))
    return np.where(np.isnan(x), x, out).astype(F32)


def pow(x, y):
    x, y = f32(x), f32(y)
    xs = np.where(x == 0, F32(1), x).astype(F32)
    r = exp2(mul(y, log2(xs)))
    r = np.where(x == 0, np.where(y > 0, F32(0), F32(np.inf)), r)
    return np.where(y == 0, F32(1), r).astype(F32)
