"""tests/probe.py -- a hand-written kernel against the product's conventions that evaluates the
numerics-contract builtins on given inputs ("pixel" x = sample index, y = function id).

The same translation unit is compiled by g++ (oracle/host_build) and by hiprtc (GPU tests)
through layer 1 of the C ABI, and its results are compared bit for bit with the numpy
restatement (oracle/glsl_math.py).
"""
import numpy as np

# (name, C++ expression over float a, b, c, numpy callable over float32 arrays)
def functions():
    from oracle import glsl_math as M
    from oracle import glsl_values as V

    v3 = lambda a, b, c: V.Vec([a, b, c])
    return [
        ("sin", "sin(a)", lambda a, b, c: M.sin(a)),
        ("cos", "cos(a)", lambda a, b, c: M.cos(a)),
        ("tan", "tan(a)", lambda a, b, c: M.tan(a)),
        ("atan", "atan(a)", lambda a, b, c: M.atan(a)),
        ("atan2", "atan(a, b)", lambda a, b, c: M.atan2(a, b)),
        ("asin", "asin(a)", lambda a, b, c: M.asin(a)),
        ("acos", "acos(a)", lambda a, b, c: M.acos(a)),
        ("exp", "exp(a)", lambda a, b, c: M.exp(a)),
        ("log", "log(a)", lambda a, b, c: M.log(a)),
        ("exp2", "exp2(a)", lambda a, b, c: M.exp2(a)),
        ("log2", "log2(a)", lambda a, b, c: M.log2(a)),
        ("pow", "pow(abs(a), b)", lambda a, b, c: M.pow(M.absf(a), b)),
        ("sqrt", "sqrt(a)", lambda a, b, c: M.sqrt(a)),
        ("inversesqrt", "inversesqrt(a)", lambda a, b, c: M.inversesqrt(a)),
        ("div", "ptl_div(a, b)", lambda a, b, c: M.div(a, b)),  # what the translator turns a scalar `a / b` into
        ("fma", "fma(a, b, c)", lambda a, b, c: M.fma(a, b, c)),
        ("mod", "mod(a, b)", lambda a, b, c: M.mod(a, b)),
        ("fract", "fract(a)", lambda a, b, c: M.fract(a)),
        ("sign", "sign(a)", lambda a, b, c: M.sign(a)),
        ("step", "step(a, b)", lambda a, b, c: M.step(a, b)),
        ("mix", "mix(a, b, c)", lambda a, b, c: M.mix(a, b, c)),
        ("smoothstep", "smoothstep(a, b, c)", lambda a, b, c: M.smoothstep(a, b, c)),
        ("clamp", "clamp(a, b, c)", lambda a, b, c: M.clamp(a, b, c)),
        ("min", "min(a, b)", lambda a, b, c: M.fmin(a, b)),
        ("max", "max(a, b)", lambda a, b, c: M.fmax(a, b)),
        ("dot3", "dot(vec3(a, b, c), vec3(b, c, a))", lambda a, b, c: V.dot(v3(a, b, c), v3(b, c, a))),
        ("length3", "length(vec3(a, b, c))", lambda a, b, c: V.length(v3(a, b, c))),
        ("normalize3y", "normalize(vec3(a, b, c)).y", lambda a, b, c: V.normalize(v3(a, b, c)).c[1]),
        ("cross_x", "cross(vec3(a, b, c), vec3(c, a, b)).x", lambda a, b, c: V.cross(v3(a, b, c), v3(c, a, b)).c[0]),
        ("vecdiv_z", "(vec3(a, b, c) / b).z", lambda a, b, c: V.binop("/", v3(a, b, c), b).c[2]),
        ("mat3vec_y", "(mat3(vec3(a, b, c), vec3(c, a, b), vec3(b, c, a)) * vec3(b, a, c)).y",
         lambda a, b, c: V.mat_vec(V.Mat([v3(a, b, c), v3(c, a, b), v3(b, c, a)]), v3(b, a, c)).c[1]),
        ("radians", "radians(a)", lambda a, b, c: M.radians(a)),
    ]


def source(pa):
    cases = "\n".join(f"        case {k}: return {expr};  // {name}" for k, (name, expr, _) in enumerate(functions()))
    return (
        pa.device_source("glsl")
        + """
#define PTL_COUNT_SEGMENT() ((void)0)
#define PTL_NO_TELEPORT_ENTRY 1
namespace glsl {
struct ptl_uniform_block { sampler2D in_tex; int n_u; int pad_u; };
#if PTL_DEVICE_BUILD
__constant__ ptl_uniform_block ptl_u;
#else
ptl_uniform_block ptl_u;
#endif
PTL_FN float probe(int fn, float a, float b, float c) {
    switch (fn) {
"""
        + cases
        + """
        default: return 0.0f;
    }
}
PTL_FN vec4 shade_pixel(vec2 position) {
    int i = (int)position.x, fn = (int)position.y;
    const float* in = reinterpret_cast<const float*>(ptl_u.in_tex.texels);
    float a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
    return vec4(probe(fn, a, b, c), a, b, c);
}
PTL_FN unsigned int pack_rgba8(vec4 c) { return 0u; }
}  // namespace glsl
"""
        + pa.device_source("entry")
    )


LAYOUT = [("in_tex", 5, 0), ("n_u", 2, 16), ("pad_u", 2, 20)]
BLOCK_SIZE = 24


def inputs(n=4096, seed=1234):
    """n samples x (a, b, c): a mix of ranges, exact specials and raw random bit patterns."""
    rng = np.random.default_rng(seed)
    parts = [
        rng.uniform(-1, 1, (n // 4, 3)),
        rng.uniform(-30, 30, (n // 4, 3)),
        rng.standard_normal((n // 8, 3)) * 1e3,
        10.0 ** rng.uniform(-30, 30, (n // 8, 3)) * rng.choice([-1, 1], (n // 8, 3)),
    ]
    x = np.concatenate(parts).astype(np.float32)
    specials = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, np.pi, -np.pi, np.pi / 2, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, -3.4028235e38, np.inf,
                         -np.inf, np.nan, 0.70710678, 88.7, -103.9, 128.0, -150.0, 1e-4, 1e4, 2.4142137, 0.41421357], np.float32)
    sp = np.stack([np.tile(specials, len(specials)), np.repeat(specials, len(specials)), np.roll(np.tile(specials, len(specials)), 5)], axis=1)
    raw = rng.integers(0, 2**32, (n - len(x) - len(sp), 3), dtype=np.uint64).astype(np.uint32).view(np.float32)
    out = np.concatenate([x, sp, raw]).astype(np.float32)
    return np.ascontiguousarray(out[:n])


def as_texture(samples):
    """float32 (n,3) -> RGBA8 texture of shape (1, 3n, 4) whose bytes are the floats."""
    return samples.reshape(-1).view(np.uint8).reshape(1, -1, 4)


def numpy_results(samples):
    a, b, c = (np.ascontiguousarray(samples[:, k]) for k in range(3))
    with np.errstate(all="ignore"):
        return np.stack([np.broadcast_to(np.asarray(fn(a, b, c), np.float32), a.shape) for _, _, fn in functions()])


def same_bits(x, y):
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    return (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))
