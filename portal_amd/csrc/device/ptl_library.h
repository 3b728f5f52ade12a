// ptl_library.h -- the fixed prelude every generated portal kernel starts with.
//
// HIP C++ restatement of the reference's GLSL prelude src/library.glsl (598 lines): the
// types and helper functions that scene snippets inside the .ron files call by name, so
// names, argument order and arithmetic order are the interface and are kept; everything
// else (struct layout, control flow shape, value-returning style) is written for
// hipcc/gfx950.  Each function cites the reference lines it follows.
//
// Expects: ptl_glsl.h included, and the uniform accessors _grid_disable,
// _angle_color_disable, _offset_after_material, _black_border_disable defined by the
// generated uniform block (codegen.cpp).

namespace glsl {

#define PI acos(-1.0f)            /* src/library.glsl:15 */
#define PI2 ptl_div(acos(-1.0f), 2.0f)  /* src/library.glsl:16 */

// src/library.glsl:19-34
PTL_FN bool between(float a, float x, float b) { return a <= x && x <= b; }
PTL_FN float sqr(float a) { return a * a; }
PTL_FN vec3 sqrvec(vec3 v) { return vec3(sqr(v.x), sqr(v.y), sqr(v.z)); }

// --- rays -------------------------------------------------------------- library.glsl:40-53
struct Ray {
    vec4 o;            // origin
    vec4 d;            // direction
    float tmul;        // distance multiplier accumulated through scaling portals
    bool in_subspace;  // "plus ultra" scenes
};
PTL_FN Ray ptl_ray_none() { return Ray{vec4(0.0f), vec4(0.0f), 0.0f, false}; }
#define ray_none (ptl_ray_none())

PTL_FN Ray offset_ray(Ray r, float t) {
    r.o += r.d * t;
    return r;
}

// library.glsl:56-62 -- unit normal facing against `dir`
PTL_FN vec3 normalize_normal(vec3 normal, vec3 dir) {
    normal = normalize(normal);
    if (dot(normal, dir) > 0.0f) normal *= -1.0f;
    return normal;
}

// library.glsl:65-67
PTL_FN bool is_collinear(vec3 a, vec3 b) {
    return abs(ptl_div(dot(a, b), length(a) * length(b)) - 1.0f) < 0.01f;
}

// library.glsl:70-72
PTL_FN vec3 my_reflect(vec3 dir, vec3 normal) {
    return dir - normal * dot(dir, normal) / dot(normal, normal) * 2.0f;  // vec3 / float: multiplies by 1 / dot(normal, normal)
}

// library.glsl:75-92
PTL_FN vec3 my_refract(vec3 dir, vec3 normal, float refractive_index) {
    float ri = refractive_index;
    bool from_outside = dot(normal, dir) > 0.0f;
    if (!from_outside) {
        ri = ptl_rcp(ri);
    } else {
        normal = -normal;
    }
    dir = normalize(dir);
    float c = -dot(normal, dir);
    float d = 1.0f - ri * ri * (1.0f - c * c);
    if (d > 0.0f) return dir * ri + normal * (ri * c - sqrt(d));
    return my_reflect(dir, normal);
}

// library.glsl:95-120
#if PTL_DEVICE_BUILD && defined(PTL_PACKED_TRANSFORM)
// Origin and direction go through the same matrix: the two products are computed as PAIRS (o.k, d.k) with the packed binary32
// instructions of gfx950 (v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per lane and issue slot, each correctly rounded like its
// scalar form), the matrix element broadcast to both halves.  Same operations in the same order as matrix * r.o and matrix * r.d.
typedef float ptl_f2 __attribute__((ext_vector_type(2)));
PTL_FN Ray transform(const mat4& m, const Ray& r) {
    const ptl_f2 px = {r.o.x, r.d.x}, py = {r.o.y, r.d.y}, pz = {r.o.z, r.d.z}, pw = {r.o.w, r.d.w};
#define PTL_ROW(k) __builtin_elementwise_fma((ptl_f2)(m.c[3].k), pw, __builtin_elementwise_fma((ptl_f2)(m.c[2].k), pz, \
                   __builtin_elementwise_fma((ptl_f2)(m.c[1].k), py, (ptl_f2)(m.c[0].k) * px)))
    const ptl_f2 x = PTL_ROW(x), y = PTL_ROW(y), z = PTL_ROW(z), w = PTL_ROW(w);
#undef PTL_ROW
    return Ray{vec4(x[0], y[0], z[0], w[0]), vec4(x[1], y[1], z[1], w[1]), r.tmul, r.in_subspace};
}
#else
PTL_FN Ray transform(const mat4& matrix, const Ray& r) {
#if defined(PTL_AFFINE_RAYS) || defined(PTL_CHECK_AFFINE)
    return Ray{ptl_mul_origin(matrix, r.o), ptl_mul_direction(matrix, r.d), r.tmul, r.in_subspace};  // o.w = 1, d.w = 0 (ptl_glsl.h `ptl_row_m`; the checking build: the plain products, w looked at)
#else
    return Ray{matrix * r.o, matrix * r.d, r.tmul, r.in_subspace};
#endif
}
#endif
// PTL_AFFINE_RAYS: the ray with its two w components spelled (they hold these very values): whatever reads them afterwards folds
PTL_FN Ray ptl_affine(const Ray& r) {
#ifdef PTL_AFFINE_RAYS
    return Ray{vec4(r.o.x, r.o.y, r.o.z, 1.0f), vec4(r.d.x, r.d.y, r.d.z, 0.0f), r.tmul, r.in_subspace};
#else
    ptl_check_w(r.o, 1.0f);  // (PTL_CHECK_AFFINE: the ray the bounce loop is about to trace; otherwise nothing)
    ptl_check_w(r.d, 0.0f);
    return r;
#endif
}
// transform() for a matrix with a known zero pattern (ptl_mul_m, device/ptl_glsl.h): what the generator writes for `transform(X_mat, ..)`
// when X_mat is a run-time uniform whose pattern it knows (PTL_MASK_X_mat)
template <ptl_mask_t MASK> PTL_FN Ray ptl_transform_m(const mat4& matrix, const Ray& r) {
    if constexpr (MASK == 0xffffu) return transform(matrix, r);
    else return Ray{ptl_mul_origin<MASK>(matrix, r.o), ptl_mul_direction<MASK>(matrix, r.d), r.tmul, r.in_subspace};
}
PTL_FN vec3 get_normal(const mat4& matrix) { return (matrix * vec4(0.0f, 0.0f, 1.0f, 0.0f)).sw<0, 1, 2>(); }
PTL_FN Ray normalize_ray(Ray r) {
    float len = length(r.d);
    r.d /= len;
    r.tmul = ptl_div(r.tmul, len);
    return r;
}
PTL_FN mat3 adjugate(const mat4& m) {
    return mat3(cross(vec3(m[1]), vec3(m[2])), cross(vec3(m[2]), vec3(m[0])), cross(vec3(m[0]), vec3(m[1])));
}

// --- surface hits ---------------------------------------------------- library.glsl:127-162
struct SurfaceIntersection {
    bool hit;
    float t;  // distance along the ray
    float u;  // surface coordinates
    float v;
    vec3 n;   // normal at the hit point
};
PTL_FN SurfaceIntersection ptl_intersection_none() { return SurfaceIntersection{false, 1e10f, 0.0f, 0.0f, vec3(0.0f)}; }
#define intersection_none (ptl_intersection_none())

PTL_FN SurfaceIntersection plane_intersect_normalized(const Ray& r) {
    float t = ptl_div(-r.o.z, r.d.z);
    if (t < 0.0f) return intersection_none;
    vec4 pos = r.o + r.d * t;
    return SurfaceIntersection{true, t, pos.x, pos.y, vec3(0.0f, 0.0f, 1.0f)};
}

// Ray against the z = 0 plane of `plane` given `plane_inv` = inverse(plane).
#if PTL_DEVICE_BUILD && defined(PTL_FAST_MATH)
// Tolerance mode: t = -o'.z / d'.z directly in the plane's frame.  The exact form normalises d' first and divides t by |d'| again
// -- a square root, a reciprocal and two divisions that cancel algebraically.
PTL_FN SurfaceIntersection ptl_plane_hit_fast(const Ray& r, const mat4& plane_inv, vec3 unit_normal) {
    const vec4 o = ptl_mul_origin(plane_inv, r.o), d = ptl_mul_direction(plane_inv, r.d);  // (the plain products unless the kernel has affine rays)
    const float t = -o.z * __builtin_amdgcn_rcpf(d.z);
    if (t < 0.0f) return intersection_none;
    return SurfaceIntersection{true, t, fma(d.x, t, o.x), fma(d.y, t, o.y), unit_normal};
}
PTL_FN SurfaceIntersection plane_intersect(Ray r, const mat4& plane_inv, vec3 normal) {
    return ptl_plane_hit_fast(r, plane_inv, normalize_normal(normal, r.d.sw<0, 1, 2>()));
}
#else
PTL_FN SurfaceIntersection plane_intersect(Ray r, const mat4& plane_inv, vec3 normal) {
    normal = normalize_normal(normal, r.d.sw<0, 1, 2>());
    r = transform(plane_inv, r);
    float len = length(r.d);
    r.d = normalize(r.d);
    SurfaceIntersection result = plane_intersect_normalized(r);
    if (result.hit) {
        result.t = ptl_div(result.t, len);
        result.n = normal;
    }
    return result;
}
#endif

// Wave-level exact cull of a plane test.  A plane only matters to scene_intersect if its hit is NEARER than the best one so far
// (nearer(): hit, t > 0, t < best).  The z row of plane_inv * ray alone -- 8 FMAs that the full test evaluates anyway -- settles that
// without the square root and the three reciprocals of plane_intersect: in the plane's frame the ray is at height o'.z and moves with
// d'.z per unit of t, so it crosses the plane at t = -o'.z / d'.z; it cannot be nearer than `best_t` exactly when it has not crossed yet
// at t = best_t, i.e. when o'.z + best_t * d'.z still has the sign of o'.z.  ONE test for both cases:
//   behind   o'.z and d'.z have the same sign (the ray moves away: it never crosses for t > 0; with best_t = +inf the sum is
//            +-inf with the sign of d'.z);
//   farther  the crossing lies beyond best_t.
// The height is taken a little farther out, at best_t * (1 + 2^-16): the exact chain's t (six roundings, < 2^-20 relative) and the
// rounding of this one FMA cannot bridge that margin, so a culled test's hit distance really is above best_t.  Zero, NaN (0 * inf,
// a NaN ray) and underflowing products compare false: not culled, tested in full -- conservative.  (Round 3 made the two decisions
// separately -- four sign compares, a reciprocal estimate, two more compares: 39 issue cycles per plane against 11.)
// A cull decides nothing about the picture -- the culled test could never have been selected -- so frames stay bit-identical;
// what it saves is the work.  It is taken per WAVE (ballot): the 64 rays of an 8x8 tile nearly always agree about which walls are
// behind them or beyond the surface they have already found, and a uniform branch costs a scalar compare.
PTL_FN bool ptl_cannot_be_nearer(float oz, float dz, float best_t) {
    return oz * __builtin_fmaf(best_t * (1.0f + 0x1p-16f), dz, oz) > 0.0f;
}
template <ptl_mask_t MASK = 0xffffu> PTL_FN bool ptl_plane_cull(const Ray& r, const mat4& plane_inv, float best_t) {
#if defined(PTL_NO_PLANE_CULL)
    (void)r; (void)plane_inv; (void)best_t;
    return false;
#else
    const float oz = ptl_row_m<MASK, 2, PTL_W_OF_ORIGIN>(plane_inv, r.o);
    const float dz = ptl_row_m<MASK, 2, PTL_W_OF_DIRECTION>(plane_inv, r.d);
#if PTL_DEVICE_BUILD
    return __builtin_amdgcn_ballot_w64(!ptl_cannot_be_nearer(oz, dz, best_t)) == 0ull;
#else
    return ptl_cannot_be_nearer(oz, dz, best_t);
#endif
#endif
}
// The cull for a ray whose origin in the plane's frame is already known (first-trip plane tests: `o_in_plane` = plane_inv * r.o from
// the prologue kernel): the same decision from the same two numbers, half the products.
template <ptl_mask_t MASK = 0xffffu> PTL_FN bool ptl_plane_cull_o(const Ray& r, const mat4& plane_inv, const vec4& o_in_plane, float best_t) {
#if defined(PTL_NO_PLANE_CULL)
    (void)r; (void)plane_inv; (void)o_in_plane; (void)best_t;
    return false;
#else
    const float oz = o_in_plane.z;
    const float dz = ptl_row_m<MASK, 2, PTL_W_OF_DIRECTION>(plane_inv, r.d);
#if PTL_DEVICE_BUILD
    return __builtin_amdgcn_ballot_w64(!ptl_cannot_be_nearer(oz, dz, best_t)) == 0ull;
#else
    return ptl_cannot_be_nearer(oz, dz, best_t);
#endif
#endif
}
#define PTL_BEST_T(i) (((i).hit.hit && (i).hit.t < ptl_far) ? (i).hit.t : ptl_far)  /* inside scene_intersect: best hit so far, capped by the caller's bound */

// plane_intersect with the ray-independent half done beforehand: `unit_normal` = normalize(normal) comes from the
// prologue kernel (ptl_tracer::derive); `flipped` tells the caller which of the two precomputed is_collinear verdicts applies.
template <ptl_mask_t MASK = 0xffffu> PTL_FN SurfaceIntersection plane_intersect_derived(Ray r, const mat4& plane_inv, vec3 unit_normal, bool& flipped) {
    flipped = dot(unit_normal, r.d.sw<0, 1, 2>()) > 0.0f;
    if (flipped) unit_normal *= -1.0f;
#if PTL_DEVICE_BUILD && defined(PTL_FAST_MATH)
    return ptl_plane_hit_fast(r, plane_inv, unit_normal);
#endif
    r = ptl_transform_m<MASK>(plane_inv, r);
    float len = length(r.d);
    r.d = normalize(r.d);
    SurfaceIntersection result = plane_intersect_normalized(r);
    if (result.hit) {
        result.t = ptl_div(result.t, len);
        result.n = unit_normal;
    }
    return result;
}

// plane_intersect / plane_intersect_derived for a ray whose origin in the plane's frame is already known (first-trip plane tests,
// KernelOptions::first_trip_planes): `transform(plane_inv, r)` becomes (o_in_plane, plane_inv * r.d); every other operation as above.
PTL_FN SurfaceIntersection plane_intersect_o(Ray r, const mat4& plane_inv, vec3 normal, const vec4& o_in_plane) {
#if PTL_DEVICE_BUILD && defined(PTL_FAST_MATH)
    (void)o_in_plane;
    return plane_intersect(r, plane_inv, normal);  // the tolerance mode has its own plane test (same source as the exact build: FLAG_FAST_MATH only adds a define)
#endif
    normal = normalize_normal(normal, r.d.sw<0, 1, 2>());
    r = Ray{o_in_plane, ptl_mul_direction(plane_inv, r.d), r.tmul, r.in_subspace};
    float len = length(r.d);
    r.d = normalize(r.d);
    SurfaceIntersection result = plane_intersect_normalized(r);
    if (result.hit) {
        result.t = ptl_div(result.t, len);
        result.n = normal;
    }
    return result;
}
template <ptl_mask_t MASK = 0xffffu> PTL_FN SurfaceIntersection plane_intersect_derived_o(Ray r, const mat4& plane_inv, vec3 unit_normal, bool& flipped, const vec4& o_in_plane) {
#if PTL_DEVICE_BUILD && defined(PTL_FAST_MATH)
    (void)o_in_plane;
    return plane_intersect_derived(r, plane_inv, unit_normal, flipped);
#endif
    flipped = dot(unit_normal, r.d.sw<0, 1, 2>()) > 0.0f;
    if (flipped) unit_normal *= -1.0f;
    r = Ray{o_in_plane, ptl_mul_direction<MASK>(plane_inv, r.d), r.tmul, r.in_subspace};
    float len = length(r.d);
    r.d = normalize(r.d);
    SurfaceIntersection result = plane_intersect_normalized(r);
    if (result.hit) {
        result.t = ptl_div(result.t, len);
        result.n = unit_normal;
    }
    return result;
}

// Staged forms of three library functions for calls whose normal argument is a uniform-only expression of a scene snippet
// (host/glsl_hoist.h re-targets such calls): the part that depends on that argument alone -- normalize(normal), length(b) -- is
// evaluated by the prologue kernel and passed in; what is left is the original's remaining operations in the original's order.
PTL_FN vec3 ptl_normalize_normal_unit(vec3 unit_normal, vec3 dir) {  // normalize_normal(n, dir) given normalize(n)
    if (dot(unit_normal, dir) > 0.0f) unit_normal *= -1.0f;
    return unit_normal;
}
PTL_FN SurfaceIntersection ptl_plane_intersect_unit(Ray r, const mat4& plane_inv, vec3 unit_normal) {  // plane_intersect(r, inv, n) given normalize(n)
    bool flipped;
    return plane_intersect_derived(r, plane_inv, unit_normal, flipped);
}
PTL_FN Ray ptl_ray_o(Ray r, vec4 origin) {  // r with its origin taken from the prologue's table: r's own origin arithmetic becomes dead code
    r.o = origin;
    return r;
}
PTL_FN bool ptl_is_collinear_len(vec3 a, vec3 b, float length_b) {  // is_collinear(a, b) given length(b)
    return abs(ptl_div(dot(a, b), length(a) * length_b) - 1.0f) < 0.01f;
}
PTL_FN bool ptl_is_collinear_len0(vec3 a, vec3 b, float length_a) {  // is_collinear(a, b) given length(a)
    return abs(ptl_div(dot(a, b), length_a * length(b)) - 1.0f) < 0.01f;
}

// --- colours ------------------------------------------------------- library.glsl:169-288
PTL_FN vec3 color(float r, float g, float b) { return vec3(r * r, g * g, b * b); }

PTL_FN float color_normal(vec3 normal, vec4 direction) {
    if (_angle_color_disable == 1) return 1.0f;
    return abs(dot(normalize(direction.sw<0, 1, 2>()), normalize(normal)));
}

PTL_FN vec3 color_grid(vec3 start, vec2 uv) {
    if (_grid_disable == 1) return start;
    uv = fract(uv * 0.25f);
    return start * mix(mix(0.7f, 1.1f, step(uv.x, 0.5f)), mix(1.1f, 0.7f, step(uv.x, 0.5f)), step(uv.y, 0.5f));
}

PTL_FN float circle_sdf(vec2 position) {
    vec2 s = vec2(2.0f, sqrt(3.0f) * 2.0f);
    position /= s;
    vec2 d1 = (fract(position) - 0.5f) * s;
    vec2 d2 = (fract(position + 0.5f) - 0.5f) * s;
    return sqrt(min(dot(d1, d1), dot(d2, d2))) - 1.0f;
}
PTL_FN vec3 color_grid2(vec3 start, vec2 uv) {
    float d = circle_sdf(uv);
    float val = 0.7f;
    if (d < -0.2f) val = 1.1f;
    return start * val;
}

PTL_FN vec3 color_grid3(vec3 start, vec2 uv) {
    if (_grid_disable == 1) return start;
    uv = fract(uv * 0.5f) - vec2(0.5f, 0.5f);
    float dist = max(abs(uv.x), abs(uv.y)) * 2.0f;
    if (dist > 0.985f) return start * 0.4f;
    if (dist < 0.94f) return start;
    if (uv.x > uv.y) return start * 0.7f;
    return start * 1.2f;
}

PTL_FN vec3 color_add_weighted(vec3 a, vec3 b, float coef) { return a * (1.0f - coef) + b * coef; }

// --- materials ------------------------------------------------------ library.glsl:297-384
struct MaterialProcessing {
    bool is_final;      // false: keep tracing along new_ray
    vec3 mul_to_color;  // throughput factor (final: the surface colour)
    Ray new_ray;
};
PTL_FN MaterialProcessing material_empty() { return MaterialProcessing{true, vec3(0.0f), ray_none}; }
PTL_FN MaterialProcessing material_final(vec3 color) { return MaterialProcessing{true, color, ray_none}; }
PTL_FN MaterialProcessing material_next(vec3 mul_color, Ray new_ray) { return MaterialProcessing{false, mul_color, new_ray}; }

PTL_FN MaterialProcessing material_simple2(SurfaceIntersection hit, Ray r, vec3 color, float normal_coef, bool grid,
                                           float grid_scale, float grid_coef, bool grid2, bool grid3) {
    color = color_add_weighted(color, color * color_normal(hit.n, r.d), normal_coef);
    if (grid) {
        vec2 cell = vec2(hit.u, hit.v) * grid_scale;
        if (grid3) {
            color = color_add_weighted(color, color_grid3(color, cell), grid_coef);
        } else if (grid2) {
            color = color_add_weighted(color, color_grid2(color, cell), grid_coef);
        } else {
            color = color_add_weighted(color, color_grid(color, cell), grid_coef);
        }
    }
    return material_final(color);
}
PTL_FN MaterialProcessing material_simple(SurfaceIntersection hit, Ray r, vec3 color, float normal_coef, bool grid,
                                          float grid_scale, float grid_coef) {
    return material_simple2(hit, r, color, normal_coef, grid, grid_scale, grid_coef, false, false);
}
PTL_FN MaterialProcessing material_reflect(SurfaceIntersection hit, Ray r, vec3 add_to_color) {
    r.d = vec4(my_reflect(r.d.sw<0, 1, 2>(), hit.n), 0.0f);
    r.o += r.d * _offset_after_material;
    return material_next(add_to_color, r);
}
PTL_FN MaterialProcessing material_refract(SurfaceIntersection hit, Ray r, vec3 add_to_color, float refractive_index) {
    r.d = vec4(my_refract(r.d.sw<0, 1, 2>(), hit.n, refractive_index), 0.0f);
    r.o += r.d * _offset_after_material;
    return material_next(add_to_color, r);
}
PTL_FN MaterialProcessing material_teleport_transformed(Ray r, vec3 n) {
    r.o += r.d * _offset_after_material;
    r = normalize_ray(r);
    return material_next(vec3(1.0f), r);
}
PTL_FN MaterialProcessing material_teleport(SurfaceIntersection hit, Ray r, const mat4& teleport_matrix) {
    return material_teleport_transformed(transform(teleport_matrix, r), hit.n);
}
PTL_FN MaterialProcessing material_change_subspace(Ray r) {
    r.in_subspace = !r.in_subspace;
    return material_next(vec3(1.0f), r);
}

// material ids -------------------------------------------------------- library.glsl:387-398
#define CUSTOM_MATERIAL (-1)
#define NOT_INSIDE 0
#define TELEPORT 1
#define TELEPORT_SUBSPACE 2
#define DEBUG_RED 3
#define DEBUG_GREEN 4
#define DEBUG_BLUE 5
#define USER_MATERIAL_OFFSET 10

// --- scene hits ------------------------------------------------------ library.glsl:405-423
struct SceneIntersection {
    int material;
    SurfaceIntersection hit;
    bool in_subspace;
};
PTL_FN SceneIntersection ptl_scene_intersection_none() { return SceneIntersection{0, intersection_none, false}; }
#define scene_intersection_none (ptl_scene_intersection_none())

PTL_FN bool nearer(const SurfaceIntersection& result, const SurfaceIntersection& current) {
    return current.hit && (current.t > 0.0f) && (!result.hit || (result.hit && current.t < result.t));
}
PTL_FN bool nearer(const SceneIntersection& result, const SurfaceIntersection& current) { return nearer(result.hit, current); }
PTL_FN bool nearer(const SceneIntersection& result, const SceneIntersection& current) { return nearer(result.hit, current.hit); }

// --- primitives (after iq, shadertoy Xt3SzX / 4lcSRn) --------------- library.glsl:426-525
PTL_FN vec3 cap_normal(vec3 pos, vec3 a, vec3 b, float radius) {
    vec3 ba = b - a;
    vec3 pa = pos - a;
    float h = clamp(ptl_div(dot(pa, ba), dot(ba, ba)), 0.0f, 1.0f);
    return (pa - h * ba) / radius;
}
PTL_FN SurfaceIntersection cap(Ray r, vec3 pa, vec3 pb, float radius) {
    vec3 ro = r.o.sw<0, 1, 2>();
    vec3 rd = r.d.sw<0, 1, 2>();
    vec3 ba = pb - pa;
    vec3 oa = ro - pa;
    float baba = dot(ba, ba);
    float bard = dot(ba, rd);
    float baoa = dot(ba, oa);
    float rdoa = dot(rd, oa);
    float oaoa = dot(oa, oa);
    float a = baba - bard * bard;
    float b = baba * rdoa - baoa * bard;
    float c = baba * oaoa - baoa * baoa - radius * radius * baba;
    float h = b * b - a * c;
    if (h >= 0.0f) {
        float t = ptl_div(-b - sqrt(h), a);
        float y = baoa + t * bard;
        if (y > 0.0f && y < baba) {  // body
            vec3 pos = ro + rd * t;
            return SurfaceIntersection{true, t, 0.0f, 0.0f, cap_normal(pos, pa, pb, radius)};
        }
        vec3 oc = (y <= 0.0f) ? oa : ro - pb;  // caps
        b = dot(rd, oc);
        c = dot(oc, oc) - radius * radius;
        h = b * b - c;
        if (h > 0.0f) {
            t = -b - sqrt(h);
            vec3 pos = ro + rd * t;
            return SurfaceIntersection{true, t, 0.0f, 0.0f, cap_normal(pos, pa, pb, radius)};
        }
    }
    return intersection_none;
}
PTL_FN SurfaceIntersection cylinder(Ray r, vec3 pa, vec3 pb, float ra) {
    vec3 ro = r.o.sw<0, 1, 2>();
    vec3 rd = r.d.sw<0, 1, 2>();
    vec3 ba = pb - pa;
    vec3 oc = ro - pa;
    float baba = dot(ba, ba);
    float bard = dot(ba, rd);
    float baoc = dot(ba, oc);
    float k2 = baba - bard * bard;
    float k1 = baba * dot(oc, rd) - baoc * bard;
    float k0 = baba * dot(oc, oc) - baoc * baoc - ra * ra * baba;
    float h = k1 * k1 - k2 * k0;
    if (h < 0.0f) return intersection_none;
    h = sqrt(h);
    float t = ptl_div(-k1 - h, k2);  // near side
    float y = baoc + t * bard;
    if (y > 0.0f && y < baba) return SurfaceIntersection{true, t, 0.0f, 0.0f, (oc + t * rd - ba * y / baba) / ra};
    t = ptl_div(-k1 + h, k2);  // far side
    y = baoc + t * bard;
    if (y > 0.0f && y < baba) return SurfaceIntersection{true, t, 0.0f, 0.0f, (oc + t * rd - ba * y / baba) / ra};
    return intersection_none;
}
PTL_FN SurfaceIntersection triangle(Ray r, vec3 v0, vec3 v1, vec3 v2) {
    vec3 ro = r.o.sw<0, 1, 2>();
    vec3 rd = r.d.sw<0, 1, 2>();
    vec3 v1v0 = v1 - v0;
    vec3 v2v0 = v2 - v0;
    vec3 rov0 = ro - v0;
    vec3 n = cross(v1v0, v2v0);
    vec3 q = cross(rov0, rd);
    float d = ptl_rcp(dot(rd, n));
    float u = d * dot(-q, v2v0);
    float v = d * dot(q, v1v0);
    float t = d * dot(-n, rov0);
    if (u < 0.0f || v < 0.0f || (u + v) > 1.0f) return intersection_none;
    return SurfaceIntersection{true, t, u, v, normalize_normal(cross(v1 - v0, v2 - v0), rd)};
}

// library.glsl:528-554 -- the three axis capsules drawn for Object::DebugMatrix
PTL_FN SceneIntersection debug_intersect(Ray r) {
    const vec3 pa = vec3(0.0f);
    const float radius = 0.03f;
    SceneIntersection i = SceneIntersection{0, intersection_none, false};
    SurfaceIntersection hit = cap(r, pa, vec3(1.0f, 0.0f, 0.0f), radius);
    if (nearer(i, hit)) { i.material = DEBUG_RED; i.hit = hit; }
    hit = cap(r, pa, vec3(0.0f, 1.0f, 0.0f), radius);
    if (nearer(i, hit)) { i.material = DEBUG_GREEN; i.hit = hit; }
    hit = cap(r, pa, vec3(0.0f, 0.0f, 1.0f), radius);
    if (nearer(i, hit)) { i.material = DEBUG_BLUE; i.hit = hit; }
    return i;
}

// library.glsl:560-589 -- what an is_inside_N() verdict does to the running nearest hit
PTL_FN SceneIntersection process_plane_intersection(SceneIntersection i, SurfaceIntersection hit, int inside) {
    if (inside != NOT_INSIDE && inside != TELEPORT && inside != TELEPORT_SUBSPACE) {
        i.hit = hit;
        i.material = inside;
    }
    return i;
}
PTL_FN SceneIntersection process_portal_intersection(SceneIntersection i, SurfaceIntersection hit, int inside, int teleport_material) {
    if (inside == NOT_INSIDE) return i;
    i.hit = hit;
    if (inside == TELEPORT) {
        i.material = teleport_material;
    } else if (inside == TELEPORT_SUBSPACE) {
        i.material = teleport_material;
        i.in_subspace = true;
    } else {
        i.material = inside;
    }
    return i;
}

// library.glsl:595-598
struct SceneIntersectionWithMaterial {
    SceneIntersection scene;      // scene.material == CUSTOM_MATERIAL: use `material` below
    MaterialProcessing material;
};

}  // namespace glsl
