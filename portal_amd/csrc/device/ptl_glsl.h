// ptl_glsl.h -- GLSL ES 3.00 value types and builtins for the generated portal-trace kernel.
//
// This header is compiled twice from the same text: by hiprtc for gfx950 (the product)
// and by g++ for the host build used as CPU baseline / cross-check (oracle/host_build).
// It therefore fixes a *numerics contract*: every builtin is spelled out in IEEE-754
// binary32 operations (+ - * / sqrt fma, all correctly rounded) in a fixed order, so
// that any conforming implementation - this header on either compiler, or the numpy
// oracle in oracle/glsl_math.py - produces bit-identical results.  Compile with
// -ffp-contract=off: fused multiply-adds appear only where this header says fma().
//
// Replaces: the GLSL builtins the reference gets from the GL driver's compiler
// (src/library.glsl:1-7 `precision highp float`; GLSL ES 3.00 spec ch. 8), which the
// reference leaves implementation-defined.  Contract choices where GLSL leaves freedom:
//   dot / length / mat*vec / cross     : fma chains, lowest component first
//   normalize(v), v / s                : multiply by the correctly-rounded reciprocal
//   min(a,b)=b<a?b:a  max(a,b)=a<b?b:a : GLSL ES 3.00 8.3 (NaN handling follows from this)
//   sin cos tan asin acos atan exp log exp2 log2 pow : polynomial kernels below

#if defined(__HIPCC_RTC__) || defined(__HIP_DEVICE_COMPILE__)
#define PTL_FN __device__ __forceinline__
#define PTL_DEVICE_BUILD 1
#else
#define PTL_FN inline
#define PTL_DEVICE_BUILD 0
#endif

// Counters, compiled in only on request (the kernel's `segments` argument receives the sum): bounce-loop trips (PTL_COUNT_SEGMENTS: segment
// Mray/s) or -- round 6, PTL_CHECK_AFFINE, the dynamic belt behind codegen.cpp `snippets_keep_rays_affine` -- the number of times a
// ray half reached a place where a kernel with affine rays ASSUMES its w (the matrix-times-ray products, the bounce loop's `ptl_affine`) with
// another w than 1 (origin) / 0 (direction).  Such a build has the general products, so its frame is right either way; a non-zero count says
// that an affine-rays kernel of this scene state would not be (ptl_renderer_check_affine).
#if defined(PTL_CHECK_AFFINE) && !defined(PTL_COUNT_SEGMENTS)
#define PTL_COUNT_SEGMENTS 1
#endif
#ifdef PTL_COUNT_SEGMENTS
#if PTL_DEVICE_BUILD
__shared__ unsigned int ptl_segments_lds[256];
#define PTL_COUNTER_BUMP() (ptl_segments_lds[threadIdx.x] += 1u)
#else
thread_local unsigned long long ptl_segments_tls = 0;
#define PTL_COUNTER_BUMP() (ptl_segments_tls += 1ull)
#endif
#endif
#if defined(PTL_CHECK_AFFINE)
#define PTL_COUNT_SEGMENT() ((void)0)
#define PTL_NOTE_NOT_AFFINE() PTL_COUNTER_BUMP()
#elif defined(PTL_COUNT_SEGMENTS)
#define PTL_COUNT_SEGMENT() PTL_COUNTER_BUMP()
#else
#define PTL_COUNT_SEGMENT() ((void)0)
#endif

namespace glsl {

// ---------------------------------------------------------------------------------------
// scalar primitives
// ---------------------------------------------------------------------------------------
PTL_FN float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// One term of a dot / matrix product chain: acc + a * b, rounded once.
// CONTRACT 2: these chains START FROM +0 -- dot(a, b) = fma(a.z, b.z, fma(a.y, b.y, fma(a.x, b.x, +0))) -- where contract 1 took the bare
// product a.x * b.x as the first term.  Same instruction count (a full-rate FMA for a full-rate MUL), same value; the only difference
// is that an exact-zero result is +0, never -0.  What it buys: with an accumulator that is not -0, a term with a ZERO factor is an
// exact no-op for every finite other factor (acc + (+-0) = acc, also when acc is +0).  So a MATRIX product may skip the terms whose
// matrix element is zero -- `ptl_mterm` below, compiled in when the scene's matrices are baked into the source (PTL_DROP_ZERO_TERMS:
// the comparison then folds at JIT time and the term is gone) -- which IEEE arithmetic alone never allows the compiler (0 * x is -0
// for a negative x, NaN for an infinite one).  With the scene state baked in, 41 % of the floating-point instructions of the headline
// snippet's loop were such products: portal matrices are mostly translations and quarter turns.
// The deviation, stated: if the vector component of a skipped term is infinite or NaN, the full chain gives NaN and the shortened one
// a number; and if a partial sum UNDERFLOWS to zero from a negative value (|a * b| < 2^-150: it is then -0 after all) the two can
// differ in the sign of a zero.  The definition is the full chain (oracle/glsl_values.py; every build without PTL_DROP_ZERO_TERMS).
PTL_FN float ptl_term(float a, float b, float acc) { return __builtin_fmaf(a, b, acc); }
PTL_FN float ptl_term0(float a, float b) {  // the first term of a chain
#if defined(PTL_CONTRACT_V1) || defined(PTL_CHAIN_FROM_PRODUCT)  /* (the second: an A/B switch for tools/variants.py, not a contract) */
    return a * b;
#else
    return __builtin_fmaf(a, b, 0.0f);
#endif
}
PTL_FN float ptl_mterm(float m, float v, float acc) {  // m: a matrix element
#if defined(PTL_DROP_ZERO_TERMS) && !defined(PTL_CONTRACT_V1) && !defined(PTL_KEEP_ZERO_TERMS)  /* (the last: A/B switch, tools/variants.py) */
    return m == 0.0f ? acc : __builtin_fmaf(m, v, acc);
#else
    return __builtin_fmaf(m, v, acc);
#endif
}
PTL_FN float ptl_mterm0(float m, float v) {
#if defined(PTL_DROP_ZERO_TERMS) && !defined(PTL_CONTRACT_V1) && !defined(PTL_KEEP_ZERO_TERMS)
    return m == 0.0f ? 0.0f : __builtin_fmaf(m, v, 0.0f);
#else
    return ptl_term0(m, v);
#endif
}

// ---- division, reciprocal, square root: CONTRACT 2 (round 3, the default) ----------------------------------------------------
// Every scalar division of the generated kernel is ptl_div(a, b); the scene snippets' `/` and `/=` are rewritten to it by the
// translator (host/glsl_translate.cpp), the prelude and the template spell it.  The definitions, for every input:
//   1/x   = the IEEE correctly rounded reciprocal, with the extremes FLUSHED: |x| < 2^-126 (zero or subnormal) gives +-inf,
//           |x| > 2^126 (the quotient would be subnormal; also +-inf) gives +-0, NaN gives NaN.
//   a / b = a * (1/b): ONE more rounding (<= 1.5 ulp; GLSL ES 3.00 4.5.1 allows 2.5 for a / b).  int / int is integer division.
//   sqrt(x) = the IEEE correctly rounded root for x >= 2^-100, +inf and NaN; |x| < 2^-100 (zeros, subnormals, the smallest
//           normals of either sign) gives +0; other negatives give NaN.
//   inversesqrt(x) = 1 / sqrt(x), normalize(v) = v * (1 / length(v)), v / s = v * (1/s): as in contract 1.
// Why: contract 1 (below, `PTL_CONTRACT_V1` / FLAG_EXACT_CR / `--exact-cr`) asked for IEEE results on EVERY input, and on gfx950
// that costs 26 (1/x), 37 (a/b) and 45 (sqrt) issue cycles per wave -- the range scaling (v_div_scale, v_div_fmas), v_div_fixup
// and compare / select pairs are 4-cycle instructions -- ~40 % of the headline kernel's VALU cycles.  Portability pins the
// MIDDLE of the range to correct rounding (a result computed from a ~1 ulp hardware estimate is seed-independent only if it is
// the correctly rounded one), and there two exact FMA steps suffice; what the long forms bought was the extremes.  Contract 2
// says what the SHORT forms do at the extremes -- the transcendental unit flushes subnormals, so they come out as the flushes
// above -- in words numpy can restate (oracle/glsl_math.py: `rcp`, `sqrt`) and the host build spells with the language's own
// operators (below).  Measured (tools/contract_probe.py, profiles/r03/contract_probe.jsonl): 18 / 18 / 31 cycles.
// Not an argument but a measurement: tests/test_gpu_parity.py::test_sqrt_and_reciprocal_match_the_contract_for_every_input runs
// ALL 2^32 bit patterns through both definitions on the GPU (the device sequences against the operator forms).
PTL_FN float ptl_rcp_model(float x) {   // the definition in the language's own operators (host build; constant folding on the GPU)
    const float m = __builtin_fabsf(x);
    if (m < 0x1p-126f) return __builtin_copysignf(__builtin_inff(), x);
    if (m > 0x1p+126f) return __builtin_copysignf(0.0f, x);
    return 1.0f / x;
}
PTL_FN float ptl_sqrt_model(float x) {
    if (__builtin_fabsf(x) < 0x1p-100f) return 0.0f;
    return __builtin_sqrtf(x);
}
#if defined(PTL_PLAIN_SQRT_RCP)
#define PTL_PLAIN_SQRT 1
#define PTL_PLAIN_RCP 1
#endif
// PTL_FAST_MATH (FLAG_FAST_MATH, `--fast`): the TOLERANCE mode.  What a GL driver does with the reference's shader: the
// hardware estimates themselves (v_sqrt_f32 / v_rcp_f32 / v_rsq_f32, 1 ulp), a / b = a * rcp(b) and FMA contraction (the JIT
// passes -ffp-contract=fast).  Not part of the bit-exact contract: frames differ from the exact kernel in the last bits, and a
// pixel on an edge can take another path (tests measure how many, vs 1e-5).
#if PTL_DEVICE_BUILD && defined(PTL_FAST_MATH)
PTL_FN float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PTL_FN float ptl_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#elif defined(PTL_CONTRACT_V1)
// ---- contract 1 (rounds 1-2): IEEE correctly rounded for every input -------------------------------------------------------
//   1/x    : the compiler refines the estimate three times; ONE exact FMA residual step is already correctly rounded on this
//            hardware.  Range scaling (v_div_scale / v_div_fmas) and the zero / infinity / NaN cases (v_div_fixup) stay: 7.
//   sqrt(x): the correctly rounded root is the estimate s, s - 1 ulp or s + 1 ulp; the signs of x - s*(s -/+ ulp), each one
//            FMA, tell which.  With the comparisons ordered as below the special values (+-0, +inf, NaN, negatives) fall
//            through unchanged, so the compiler's separate class test and select are not needed: 14.
#if PTL_DEVICE_BUILD && !defined(PTL_PLAIN_SQRT)
PTL_FN float sqrt(float x) {
    const bool tiny = x < 0x1p-96f;                // below, the residuals would underflow: work on x * 2^32, give back s * 2^-16
    const float xs = tiny ? x * 0x1p+32f : x;
    float s = __builtin_amdgcn_sqrtf(xs);
    const int bits = __builtin_bit_cast(int, s);
    const float below = __builtin_bit_cast(float, bits - 1), above = __builtin_bit_cast(float, bits + 1);
    const float r_below = __builtin_fmaf(-below, s, xs), r_above = __builtin_fmaf(-above, s, xs);
    s = r_below <= 0.0f ? below : s;               // (false for a NaN residual: +-0, +inf and NaN keep s)
    s = r_above > 0.0f ? above : s;
    return tiny ? s * 0x1p-16f : s;
}
#else
PTL_FN float sqrt(float x) { return __builtin_sqrtf(x); }
#endif
#if PTL_DEVICE_BUILD && !defined(PTL_PLAIN_RCP)
PTL_FN float ptl_rcp(float x) {
    bool unused, rescale;
    const float d = __builtin_amdgcn_div_scalef(1.0f, x, false, &unused);  // x, or x * 2^+-64 when 1/x needs the room
    const float n = __builtin_amdgcn_div_scalef(1.0f, x, true, &rescale);  // 1, scaled to match
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    const float residual = __builtin_fmaf(-d, q, n);                        // exact
    return __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(residual, r, q, rescale), x, 1.0f);
}
#else
PTL_FN float ptl_rcp(float x) { return 1.0f / x; }
#endif
#elif PTL_DEVICE_BUILD && !defined(PTL_PLAIN_SQRT)
// ---- contract 2 on gfx950 -----------------------------------------------------------------------------------------------------
//   1/x    : y0 = v_rcp_f32(x) (1 ulp; +-inf for zeros AND subnormals, +-0 for |x| > 2^126 AND infinities); one exact residual
//            e = 1 - x*y0 and y = y0 + y0*e is the correctly rounded reciprocal wherever y0 is a normal number.  At the flushed
//            ends the residual is 1 (y stays +-0) or not a number (y is NaN: the estimate itself is the answer).  5 instructions.
//   sqrt(x): g = v_sqrt_f32 (1 ulp), h = v_rsq_f32 / 2; d = x - g*g is exact above 2^-103 (a multiple of ulp(g)^2) and
//            s = g + d*h rounds correctly (Markstein).  Inputs below 2^-100 are flushed to +0 first (a select on |x|, so negative
//            numbers of ordinary size still give NaN); +0 and +inf make the correction NaN and keep the estimate (0, inf).
//   The "not a number -> keep the estimate" step is ONE instruction: v_med3_f32(y, y0, y).  With a NaN among its operands v_med3 returns
//   v_min3 of them, and v_min returns the operand that is a number -- y0 when y is NaN (NaN when both are); without a NaN the median
//   of {y, y0, y} is y.  A compare + select pair costs 7.5 issue cycles on gfx950 against 2.3 (profiles/r01/valu_rates.jsonl), and
//   these two guards were 37 % of the headline kernel's compares and 55 % of its selects (tools/isa_hist.py).  -DPTL_CMP_GUARD: the pair (A/B).
PTL_FN float ptl_number_or(float y, float fallback) {  // y, or `fallback` where y is NaN
#if defined(PTL_CMP_GUARD)
    return y == y ? y : fallback;
#else
    return __builtin_amdgcn_fmed3f(y, fallback, y);
#endif
}
PTL_FN float ptl_rcp(float x) {
    if (__builtin_constant_p(x)) return ptl_rcp_model(x);  // after JIT specialisation: the compiler folds the operator form
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y = __builtin_fmaf(y0, e, y0);
    return ptl_number_or(y, y0);
}
PTL_FN float sqrt(float x) {
    if (__builtin_constant_p(x)) return ptl_sqrt_model(x);
    const float xe = __builtin_fabsf(x) < 0x1p-100f ? 0.0f : x;
    const float g = __builtin_amdgcn_sqrtf(xe);
    const float h = 0.5f * __builtin_amdgcn_rsqf(xe);
    const float d = __builtin_fmaf(-g, g, xe);
    const float s = __builtin_fmaf(d, h, g);
    return ptl_number_or(s, g);
}
#else
PTL_FN float ptl_rcp(float x) { return ptl_rcp_model(x); }
PTL_FN float sqrt(float x) { return ptl_sqrt_model(x); }
#endif
// a / b.  Overloads, not a template: `int / int` has to stay an integer division, and a call with an int and a float operand
// (which desktop GLSL accepts) must not fall back to the language's own division.
#if defined(PTL_CONTRACT_V1) && !(PTL_DEVICE_BUILD && defined(PTL_FAST_MATH))
PTL_FN float ptl_div(float a, float b) { return a / b; }
#else
PTL_FN float ptl_div(float a, float b) { return a * ptl_rcp(b); }
#endif
PTL_FN float ptl_div(float a, int b) { return ptl_div(a, (float)b); }
PTL_FN float ptl_div(int a, float b) { return ptl_div((float)a, b); }
PTL_FN int ptl_div(int a, int b) { return a / b; }
PTL_FN unsigned ptl_div(unsigned a, unsigned b) { return a / b; }
PTL_FN float abs(float x) { return __builtin_fabsf(x); }
PTL_FN int abs(int x) { return x < 0 ? -x : x; }
PTL_FN float floor(float x) { return __builtin_floorf(x); }
PTL_FN float ceil(float x) { return __builtin_ceilf(x); }
PTL_FN float trunc(float x) { return __builtin_truncf(x); }
PTL_FN float roundEven(float x) { return __builtin_rintf(x); }
PTL_FN float round(float x) { return __builtin_rintf(x); }
#if PTL_DEVICE_BUILD && defined(PTL_FAST_MATH)
PTL_FN float inversesqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
PTL_FN float inversesqrt(float x) { return ptl_rcp(sqrt(x)); }
#endif
PTL_FN float fract(float x) { return x - floor(x); }
PTL_FN float mod(float x, float y) { return x - y * floor(ptl_div(x, y)); }
PTL_FN float min(float a, float b) { return b < a ? b : a; }
PTL_FN float max(float a, float b) { return a < b ? b : a; }
PTL_FN int min(int a, int b) { return b < a ? b : a; }
PTL_FN int max(int a, int b) { return a < b ? b : a; }
PTL_FN float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
PTL_FN int clamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
PTL_FN float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
PTL_FN float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
PTL_FN float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
PTL_FN float smoothstep(float e0, float e1, float x) {
    float t = clamp(ptl_div(x - e0, e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
PTL_FN float radians(float d) { return d * 0x1.1df46ap-6f; }
PTL_FN float degrees(float r) { return r * 0x1.ca5dc2p+5f; }

PTL_FN int floatBitsToInt(float x) { return __builtin_bit_cast(int, x); }
PTL_FN float intBitsToFloat(int b) { return __builtin_bit_cast(float, b); }
PTL_FN bool isnan(float x) { return x != x; }
PTL_FN bool isinf(float x) { return abs(x) == __builtin_inff(); }

// 2^e for e in [-126, 127]
PTL_FN float ptl_pow2i(int e) { return intBitsToFloat((e + 127) << 23); }
// z * 2^n for n in [-252, 254], exact except for the final (sub)normal rounding
PTL_FN float ptl_scale2(float z, int n) {
    int h = n >> 1;
    return z * ptl_pow2i(h) * ptl_pow2i(n - h);
}

// ---------------------------------------------------------------------------------------
// sin / cos / tan: k = rint(x*2/pi); 3-term Cody-Waite reduction with fma; degree-7 / 8
// minimax kernels on [-pi/4, pi/4] (Cephes sinf/cosf coefficients).
// ---------------------------------------------------------------------------------------
struct ptl_SinCos { float s, c; };
PTL_FN ptl_SinCos ptl_sincos(float x, float& quadrant) {
    float k = __builtin_rintf(x * 0x1.45f306p-1f);
    float r = fma(-k, 0x1.921fb6p+0f, x);
    r = fma(-k, -0x1.777a5cp-25f, r);
    r = fma(-k, -0x1.ee59dap-50f, r);
    quadrant = k - 4.0f * floor(k * 0.25f);
    float z = r * r;
    float ps = fma(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fma(z, ps, -1.6666654611e-1f);
    float pc = fma(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fma(z, pc, 4.166664568298827e-2f);
    ptl_SinCos o;
    o.s = fma(r * z, ps, r);
    o.c = fma(z * z, pc, fma(-0.5f, z, 1.0f));
    return o;
}
PTL_FN float sin(float x) {
    float q;
    ptl_SinCos k = ptl_sincos(x, q);
    float v = (q == 1.0f || q == 3.0f) ? k.c : k.s;
    return (q == 2.0f || q == 3.0f) ? -v : v;
}
PTL_FN float cos(float x) {
    float q;
    ptl_SinCos k = ptl_sincos(x, q);
    float v = (q == 1.0f || q == 3.0f) ? k.s : k.c;
    return (q == 1.0f || q == 2.0f) ? -v : v;
}
PTL_FN float tan(float x) { return ptl_div(sin(x), cos(x)); }

// ---------------------------------------------------------------------------------------
// atan / atan2 (Cephes atanf): reduce to [0, tan(pi/8)], degree-9 odd kernel.
// ---------------------------------------------------------------------------------------
PTL_FN float atan(float x0) {
    float x = abs(x0);
    float y = 0.0f;
    if (x > 2.414213562373095f) {
        y = 0x1.921fb6p+0f;
        x = -ptl_rcp(x);
    } else if (x > 0.4142135623730950f) {
        y = 0x1.921fb6p-1f;
        x = ptl_div(x - 1.0f, x + 1.0f);
    }
    float z = x * x;
    float p = fma(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = fma(z, p, 1.99777106478e-1f);
    p = fma(z, p, -3.33329491539e-1f);
    y = y + fma(p * z, x, x);
    return x0 < 0.0f ? -y : y;
}
PTL_FN float atan(float y, float x) {
    if (x == 0.0f) {
        return y > 0.0f ? 0x1.921fb6p+0f : (y < 0.0f ? -0x1.921fb6p+0f : 0.0f);
    }
    float w = 0.0f;
    if (x < 0.0f) w = y < 0.0f ? -0x1.921fb6p+1f : 0x1.921fb6p+1f;
    return w + atan(ptl_div(y, x));
}

// ---------------------------------------------------------------------------------------
// asin / acos (Cephes asinf / acosf).  acos(-1.) == float(pi), which the reference's
// `#define PI acos(-1.)` (src/library.glsl:15) relies on.
// ---------------------------------------------------------------------------------------
PTL_FN float asin(float x0) {
    float a = abs(x0);
    bool big = a > 0.5f;
    float z = big ? 0.5f * (1.0f - a) : a * a;
    float x = big ? sqrt(z) : a;
    float p = fma(z, 4.2163199048e-2f, 2.4181311049e-2f);
    p = fma(z, p, 4.5470025998e-2f);
    p = fma(z, p, 7.4953002686e-2f);
    p = fma(z, p, 1.6666752422e-1f);
    float r = fma(p * z, x, x);
    if (big) r = 0x1.921fb6p+0f - (r + r);
    return x0 < 0.0f ? -r : r;
}
PTL_FN float acos(float x) {
    if (x < -0.5f) return 0x1.921fb6p+1f - 2.0f * asin(sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin(sqrt(0.5f * (1.0f - x)));
    return 0x1.921fb6p+0f - asin(x);
}

// ---------------------------------------------------------------------------------------
// exp2 / log2 / exp / log / pow
// ---------------------------------------------------------------------------------------
PTL_FN float exp2(float x) {
    if (x != x) return x;
    if (x > 128.0f) return __builtin_inff();
    if (x < -150.0f) return 0.0f;
    float n = __builtin_rintf(x);
    float f = x - n;
    float p = fma(f, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = fma(f, p, 9.618437357674640e-3f);
    p = fma(f, p, 5.550332471162809e-2f);
    p = fma(f, p, 2.402264791363012e-1f);
    p = fma(f, p, 6.931472028550421e-1f);
    return ptl_scale2(fma(f, p, 1.0f), (int)n);
}
// frexp-style split: x = m * 2^e with m in [sqrt(1/2), sqrt(2)); returns m - 1
PTL_FN float ptl_log_split(float x, float& e_out) {
    float e_adj = 0.0f;
    if (x < 0x1p-126f) {
        x = x * 0x1p+24f;
        e_adj = -24.0f;
    }
    int b = floatBitsToInt(x);
    int e = ((b >> 23) & 0xff) - 126;
    float m = intBitsToFloat((b & 0x007fffff) | 0x3f000000);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m;
    }
    e_out = (float)e + e_adj;
    return m - 1.0f;
}
PTL_FN float ptl_log_poly(float m) {
    float p = fma(m, 7.0376836292e-2f, -1.1514610310e-1f);
    p = fma(m, p, 1.1676998740e-1f);
    p = fma(m, p, -1.2420140846e-1f);
    p = fma(m, p, 1.4249322787e-1f);
    p = fma(m, p, -1.6668057665e-1f);
    p = fma(m, p, 2.0000714765e-1f);
    p = fma(m, p, -2.4999993993e-1f);
    p = fma(m, p, 3.3333331174e-1f);
    float z = m * m;
    return fma(-0.5f, z, p * m * z);
}
PTL_FN float log(float x) {
    if (x != x || x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    float e;
    float m = ptl_log_split(x, e);
    float y = ptl_log_poly(m);
    y = fma(e, -2.12194440e-4f, y);
    return fma(e, 0.693359375f, m + y);
}
PTL_FN float log2(float x) {
    if (x != x || x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    float e;
    float m = ptl_log_split(x, e);
    float y = ptl_log_poly(m);
    // log2(1+m) = (m + y) * log2(e), with log2(e) = 1 + 0.44269504088896340735992
    float z = y * 0.44269504088896340735992f;
    z = fma(m, 0.44269504088896340735992f, z);
    z = z + y;
    z = z + m;
    return z + e;
}
PTL_FN float exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -103.972076416015625f) return 0.0f;
    float n = floor(fma(x, 0x1.715476p+0f, 0.5f));
    float r = fma(-n, 0.693359375f, x);
    r = fma(-n, -2.12194440e-4f, r);
    float p = fma(r, 1.9875691500e-4f, 1.3981999507e-3f);
    p = fma(r, p, 8.3334519073e-3f);
    p = fma(r, p, 4.1665795894e-2f);
    p = fma(r, p, 1.6666665459e-1f);
    p = fma(r, p, 5.0000001201e-1f);
    float y = fma(p, r * r, r) + 1.0f;
    return ptl_scale2(y, (int)n);
}
// GLSL: pow(x, y) undefined for x < 0 and for x == 0 && y <= 0.
PTL_FN float pow(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
    return exp2(y * log2(x));
}

// ---------------------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------------------
struct vec2;
struct vec3;
struct vec4;

template <class V, int N, int A, int B, int C, int D> struct ptl_swizzle_ref;
template <int N> struct ptl_vec_of;
template <> struct ptl_vec_of<2> { using type = vec2; };
template <> struct ptl_vec_of<3> { using type = vec3; };
template <> struct ptl_vec_of<4> { using type = vec4; };

struct vec2 {
    union { float x; float r; float s; };
    union { float y; float g; };
    vec2() = default;
    PTL_FN explicit vec2(float a) : x(a), y(a) {}
    PTL_FN vec2(float a, float b) : x(a), y(b) {}
    PTL_FN explicit vec2(const vec3& v);
    PTL_FN explicit vec2(const vec4& v);
    template <int I> PTL_FN float c() const { return I == 0 ? x : y; }
    template <int I> PTL_FN float& cr() { if constexpr (I == 0) return x; else return y; }
    PTL_FN float& operator[](int i) { return i == 0 ? x : y; }
    PTL_FN float operator[](int i) const { return i == 0 ? x : y; }
    template <int A, int B> PTL_FN vec2 sw() const;
    template <int A, int B, int C> PTL_FN vec3 sw() const;
    template <int A, int B, int C, int D> PTL_FN vec4 sw() const;
    template <int A, int B> PTL_FN ptl_swizzle_ref<vec2, 2, A, B, 0, 0> swr();
    template <int A, int B, int C> PTL_FN ptl_swizzle_ref<vec2, 3, A, B, C, 0> swr();
    template <int A, int B, int C, int D> PTL_FN ptl_swizzle_ref<vec2, 4, A, B, C, D> swr();
};

struct vec3 {
    union { float x; float r; float s; };
    union { float y; float g; };
    union { float z; float b; };
    vec3() = default;
    PTL_FN explicit vec3(float a) : x(a), y(a), z(a) {}
    PTL_FN vec3(float a, float b_, float c_) : x(a), y(b_), z(c_) {}
    PTL_FN vec3(const vec2& v, float c_) : x(v.x), y(v.y), z(c_) {}
    PTL_FN vec3(float a, const vec2& v) : x(a), y(v.x), z(v.y) {}
    PTL_FN explicit vec3(const vec4& v);
    template <int I> PTL_FN float c() const { return I == 0 ? x : (I == 1 ? y : z); }
    template <int I> PTL_FN float& cr() {
        if constexpr (I == 0) return x; else if constexpr (I == 1) return y; else return z;
    }
    PTL_FN float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    PTL_FN float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    template <int A, int B> PTL_FN vec2 sw() const;
    template <int A, int B, int C> PTL_FN vec3 sw() const;
    template <int A, int B, int C, int D> PTL_FN vec4 sw() const;
    template <int A, int B> PTL_FN ptl_swizzle_ref<vec3, 2, A, B, 0, 0> swr();
    template <int A, int B, int C> PTL_FN ptl_swizzle_ref<vec3, 3, A, B, C, 0> swr();
    template <int A, int B, int C, int D> PTL_FN ptl_swizzle_ref<vec3, 4, A, B, C, D> swr();
};

struct vec4 {
    union { float x; float r; float s; };
    union { float y; float g; };
    union { float z; float b; };
    union { float w; float a; };
    vec4() = default;
    PTL_FN explicit vec4(float v) : x(v), y(v), z(v), w(v) {}
    PTL_FN vec4(float a_, float b_, float c_, float d_) : x(a_), y(b_), z(c_), w(d_) {}
    PTL_FN vec4(const vec3& v, float d_) : x(v.x), y(v.y), z(v.z), w(d_) {}
    PTL_FN vec4(float a_, const vec3& v) : x(a_), y(v.x), z(v.y), w(v.z) {}
    PTL_FN vec4(const vec2& p, const vec2& q) : x(p.x), y(p.y), z(q.x), w(q.y) {}
    PTL_FN vec4(const vec2& p, float c_, float d_) : x(p.x), y(p.y), z(c_), w(d_) {}
    PTL_FN vec4(float a_, const vec2& p, float d_) : x(a_), y(p.x), z(p.y), w(d_) {}
    PTL_FN vec4(float a_, float b_, const vec2& p) : x(a_), y(b_), z(p.x), w(p.y) {}
    template <int I> PTL_FN float c() const { return I == 0 ? x : (I == 1 ? y : (I == 2 ? z : w)); }
    template <int I> PTL_FN float& cr() {
        if constexpr (I == 0) return x; else if constexpr (I == 1) return y;
        else if constexpr (I == 2) return z; else return w;
    }
    PTL_FN float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    PTL_FN float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    template <int A, int B> PTL_FN vec2 sw() const;
    template <int A, int B, int C> PTL_FN vec3 sw() const;
    template <int A, int B, int C, int D> PTL_FN vec4 sw() const;
    template <int A, int B> PTL_FN ptl_swizzle_ref<vec4, 2, A, B, 0, 0> swr();
    template <int A, int B, int C> PTL_FN ptl_swizzle_ref<vec4, 3, A, B, C, 0> swr();
    template <int A, int B, int C, int D> PTL_FN ptl_swizzle_ref<vec4, 4, A, B, C, D> swr();
};

PTL_FN vec2::vec2(const vec3& v) : x(v.x), y(v.y) {}
PTL_FN vec2::vec2(const vec4& v) : x(v.x), y(v.y) {}
PTL_FN vec3::vec3(const vec4& v) : x(v.x), y(v.y), z(v.z) {}

#define PTL_SWIZZLES(V)                                                                          \
    template <int A, int B> PTL_FN vec2 V::sw() const { return vec2(c<A>(), c<B>()); }            \
    template <int A, int B, int C> PTL_FN vec3 V::sw() const { return vec3(c<A>(), c<B>(), c<C>()); } \
    template <int A, int B, int C, int D> PTL_FN vec4 V::sw() const {                            \
        return vec4(c<A>(), c<B>(), c<C>(), c<D>());                                             \
    }
PTL_SWIZZLES(vec2)
PTL_SWIZZLES(vec3)
PTL_SWIZZLES(vec4)
#undef PTL_SWIZZLES

// writable swizzle view: `v.xy = e`, `v.zyx += e` (translator emits v.swr<..>())
template <class V, int N, int A, int B, int C, int D> struct ptl_swizzle_ref {
    using T = typename ptl_vec_of<N>::type;
    V& v;
    PTL_FN T get() const {
        if constexpr (N == 2) return T(v.template c<A>(), v.template c<B>());
        else if constexpr (N == 3) return T(v.template c<A>(), v.template c<B>(), v.template c<C>());
        else return T(v.template c<A>(), v.template c<B>(), v.template c<C>(), v.template c<D>());
    }
    PTL_FN operator T() const { return get(); }
    PTL_FN ptl_swizzle_ref& operator=(const T& t) {
        v.template cr<A>() = t.x;
        v.template cr<B>() = t.y;
        if constexpr (N >= 3) v.template cr<C>() = t.z;
        if constexpr (N >= 4) v.template cr<D>() = t.w;
        return *this;
    }
    PTL_FN ptl_swizzle_ref& operator+=(const T& t) { return *this = get() + t; }
    PTL_FN ptl_swizzle_ref& operator-=(const T& t) { return *this = get() - t; }
    PTL_FN ptl_swizzle_ref& operator*=(const T& t) { return *this = get() * t; }
    PTL_FN ptl_swizzle_ref& operator/=(const T& t) { return *this = get() / t; }
    PTL_FN ptl_swizzle_ref& operator*=(float t) { return *this = get() * t; }
    PTL_FN ptl_swizzle_ref& operator/=(float t) { return *this = get() / t; }
};
#define PTL_SWIZZLE_REFS(V)                                                                      \
    template <int A, int B> PTL_FN ptl_swizzle_ref<V, 2, A, B, 0, 0> V::swr() { return {*this}; }  \
    template <int A, int B, int C> PTL_FN ptl_swizzle_ref<V, 3, A, B, C, 0> V::swr() { return {*this}; } \
    template <int A, int B, int C, int D> PTL_FN ptl_swizzle_ref<V, 4, A, B, C, D> V::swr() { return {*this}; }
PTL_SWIZZLE_REFS(vec2)
PTL_SWIZZLE_REFS(vec3)
PTL_SWIZZLE_REFS(vec4)
#undef PTL_SWIZZLE_REFS

// component-wise operators --------------------------------------------------------------
#define PTL_VEC_BINOP(OP)                                                                        \
    PTL_FN vec2 operator OP(const vec2& a, const vec2& b) { return vec2(a.x OP b.x, a.y OP b.y); } \
    PTL_FN vec3 operator OP(const vec3& a, const vec3& b) { return vec3(a.x OP b.x, a.y OP b.y, a.z OP b.z); } \
    PTL_FN vec4 operator OP(const vec4& a, const vec4& b) { return vec4(a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w); } \
    PTL_FN vec2 operator OP(const vec2& a, float b) { return vec2(a.x OP b, a.y OP b); }          \
    PTL_FN vec3 operator OP(const vec3& a, float b) { return vec3(a.x OP b, a.y OP b, a.z OP b); } \
    PTL_FN vec4 operator OP(const vec4& a, float b) { return vec4(a.x OP b, a.y OP b, a.z OP b, a.w OP b); } \
    PTL_FN vec2 operator OP(float a, const vec2& b) { return vec2(a OP b.x, a OP b.y); }          \
    PTL_FN vec3 operator OP(float a, const vec3& b) { return vec3(a OP b.x, a OP b.y, a OP b.z); } \
    PTL_FN vec4 operator OP(float a, const vec4& b) { return vec4(a OP b.x, a OP b.y, a OP b.z, a OP b.w); }
PTL_VEC_BINOP(+)
PTL_VEC_BINOP(-)
PTL_VEC_BINOP(*)
#undef PTL_VEC_BINOP
// division: vec/vec and float/vec are true component divisions; vec/float multiplies by
// the correctly-rounded reciprocal (contract, see header comment).
PTL_FN vec2 operator/(const vec2& a, const vec2& b) { return vec2(ptl_div(a.x, b.x), ptl_div(a.y, b.y)); }
PTL_FN vec3 operator/(const vec3& a, const vec3& b) { return vec3(ptl_div(a.x, b.x), ptl_div(a.y, b.y), ptl_div(a.z, b.z)); }
PTL_FN vec4 operator/(const vec4& a, const vec4& b) { return vec4(ptl_div(a.x, b.x), ptl_div(a.y, b.y), ptl_div(a.z, b.z), ptl_div(a.w, b.w)); }
PTL_FN vec2 operator/(float a, const vec2& b) { return vec2(ptl_div(a, b.x), ptl_div(a, b.y)); }
PTL_FN vec3 operator/(float a, const vec3& b) { return vec3(ptl_div(a, b.x), ptl_div(a, b.y), ptl_div(a, b.z)); }
PTL_FN vec4 operator/(float a, const vec4& b) { return vec4(ptl_div(a, b.x), ptl_div(a, b.y), ptl_div(a, b.z), ptl_div(a, b.w)); }
PTL_FN vec2 operator/(const vec2& a, float b) { float i = ptl_rcp(b); return vec2(a.x * i, a.y * i); }
PTL_FN vec3 operator/(const vec3& a, float b) { float i = ptl_rcp(b); return vec3(a.x * i, a.y * i, a.z * i); }
PTL_FN vec4 operator/(const vec4& a, float b) { float i = ptl_rcp(b); return vec4(a.x * i, a.y * i, a.z * i, a.w * i); }

// what the translator turns `a / b` and `a /= b` of a scene snippet into when an operand is not a scalar
template <class A, class B> PTL_FN auto ptl_div(const A& a, const B& b) -> decltype(a / b) { return a / b; }
template <class A, class B> PTL_FN void ptl_div_assign(A& a, const B& b) { a = ptl_div(a, b); }

PTL_FN vec2 operator-(const vec2& a) { return vec2(-a.x, -a.y); }
PTL_FN vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
PTL_FN vec4 operator-(const vec4& a) { return vec4(-a.x, -a.y, -a.z, -a.w); }
PTL_FN vec2 operator+(const vec2& a) { return a; }
PTL_FN vec3 operator+(const vec3& a) { return a; }
PTL_FN vec4 operator+(const vec4& a) { return a; }

#define PTL_VEC_ASSIGN(V)                                                                        \
    PTL_FN V& operator+=(V& a, const V& b) { a = a + b; return a; }                               \
    PTL_FN V& operator-=(V& a, const V& b) { a = a - b; return a; }                               \
    PTL_FN V& operator*=(V& a, const V& b) { a = a * b; return a; }                               \
    PTL_FN V& operator/=(V& a, const V& b) { a = a / b; return a; }                               \
    PTL_FN V& operator+=(V& a, float b) { a = a + b; return a; }                                  \
    PTL_FN V& operator-=(V& a, float b) { a = a - b; return a; }                                  \
    PTL_FN V& operator*=(V& a, float b) { a = a * b; return a; }                                  \
    PTL_FN V& operator/=(V& a, float b) { a = a / b; return a; }
PTL_VEC_ASSIGN(vec2)
PTL_VEC_ASSIGN(vec3)
PTL_VEC_ASSIGN(vec4)
#undef PTL_VEC_ASSIGN

PTL_FN bool operator==(const vec2& a, const vec2& b) { return a.x == b.x && a.y == b.y; }
PTL_FN bool operator==(const vec3& a, const vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
PTL_FN bool operator==(const vec4& a, const vec4& b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
PTL_FN bool operator!=(const vec2& a, const vec2& b) { return !(a == b); }
PTL_FN bool operator!=(const vec3& a, const vec3& b) { return !(a == b); }
PTL_FN bool operator!=(const vec4& a, const vec4& b) { return !(a == b); }

// component-wise builtins ---------------------------------------------------------------
#define PTL_MAP1(F)                                                                              \
    PTL_FN vec2 F(const vec2& a) { return vec2(F(a.x), F(a.y)); }                                 \
    PTL_FN vec3 F(const vec3& a) { return vec3(F(a.x), F(a.y), F(a.z)); }                         \
    PTL_FN vec4 F(const vec4& a) { return vec4(F(a.x), F(a.y), F(a.z), F(a.w)); }
PTL_MAP1(sin) PTL_MAP1(cos) PTL_MAP1(tan) PTL_MAP1(asin) PTL_MAP1(acos) PTL_MAP1(atan)
PTL_MAP1(exp) PTL_MAP1(log) PTL_MAP1(exp2) PTL_MAP1(log2) PTL_MAP1(sqrt) PTL_MAP1(inversesqrt)
PTL_MAP1(abs) PTL_MAP1(sign) PTL_MAP1(floor) PTL_MAP1(ceil) PTL_MAP1(fract) PTL_MAP1(trunc)
PTL_MAP1(round) PTL_MAP1(roundEven) PTL_MAP1(radians) PTL_MAP1(degrees)
#undef PTL_MAP1
#define PTL_MAP2(F)                                                                              \
    PTL_FN vec2 F(const vec2& a, const vec2& b) { return vec2(F(a.x, b.x), F(a.y, b.y)); }         \
    PTL_FN vec3 F(const vec3& a, const vec3& b) { return vec3(F(a.x, b.x), F(a.y, b.y), F(a.z, b.z)); } \
    PTL_FN vec4 F(const vec4& a, const vec4& b) { return vec4(F(a.x, b.x), F(a.y, b.y), F(a.z, b.z), F(a.w, b.w)); }
#define PTL_MAP2S(F) /* second operand scalar */                                                 \
    PTL_FN vec2 F(const vec2& a, float b) { return vec2(F(a.x, b), F(a.y, b)); }                   \
    PTL_FN vec3 F(const vec3& a, float b) { return vec3(F(a.x, b), F(a.y, b), F(a.z, b)); }        \
    PTL_FN vec4 F(const vec4& a, float b) { return vec4(F(a.x, b), F(a.y, b), F(a.z, b), F(a.w, b)); }
PTL_MAP2(min) PTL_MAP2(max) PTL_MAP2(mod) PTL_MAP2(pow) PTL_MAP2(atan) PTL_MAP2(step)
PTL_MAP2S(min) PTL_MAP2S(max) PTL_MAP2S(mod)
#undef PTL_MAP2
#undef PTL_MAP2S
PTL_FN vec2 step(float e, const vec2& a) { return vec2(step(e, a.x), step(e, a.y)); }
PTL_FN vec3 step(float e, const vec3& a) { return vec3(step(e, a.x), step(e, a.y), step(e, a.z)); }
PTL_FN vec4 step(float e, const vec4& a) { return vec4(step(e, a.x), step(e, a.y), step(e, a.z), step(e, a.w)); }
PTL_FN vec2 clamp(const vec2& a, float lo, float hi) { return vec2(clamp(a.x, lo, hi), clamp(a.y, lo, hi)); }
PTL_FN vec3 clamp(const vec3& a, float lo, float hi) { return vec3(clamp(a.x, lo, hi), clamp(a.y, lo, hi), clamp(a.z, lo, hi)); }
PTL_FN vec4 clamp(const vec4& a, float lo, float hi) { return vec4(clamp(a.x, lo, hi), clamp(a.y, lo, hi), clamp(a.z, lo, hi), clamp(a.w, lo, hi)); }
PTL_FN vec2 clamp(const vec2& a, const vec2& lo, const vec2& hi) { return min(max(a, lo), hi); }
PTL_FN vec3 clamp(const vec3& a, const vec3& lo, const vec3& hi) { return min(max(a, lo), hi); }
PTL_FN vec4 clamp(const vec4& a, const vec4& lo, const vec4& hi) { return min(max(a, lo), hi); }
PTL_FN vec2 mix(const vec2& a, const vec2& b, float t) { return vec2(mix(a.x, b.x, t), mix(a.y, b.y, t)); }
PTL_FN vec3 mix(const vec3& a, const vec3& b, float t) { return vec3(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)); }
PTL_FN vec4 mix(const vec4& a, const vec4& b, float t) { return vec4(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t), mix(a.w, b.w, t)); }
PTL_FN vec2 mix(const vec2& a, const vec2& b, const vec2& t) { return vec2(mix(a.x, b.x, t.x), mix(a.y, b.y, t.y)); }
PTL_FN vec3 mix(const vec3& a, const vec3& b, const vec3& t) { return vec3(mix(a.x, b.x, t.x), mix(a.y, b.y, t.y), mix(a.z, b.z, t.z)); }
PTL_FN vec4 mix(const vec4& a, const vec4& b, const vec4& t) { return vec4(mix(a.x, b.x, t.x), mix(a.y, b.y, t.y), mix(a.z, b.z, t.z), mix(a.w, b.w, t.w)); }
PTL_FN vec2 smoothstep(float e0, float e1, const vec2& a) { return vec2(smoothstep(e0, e1, a.x), smoothstep(e0, e1, a.y)); }
PTL_FN vec3 smoothstep(float e0, float e1, const vec3& a) { return vec3(smoothstep(e0, e1, a.x), smoothstep(e0, e1, a.y), smoothstep(e0, e1, a.z)); }
PTL_FN vec4 smoothstep(float e0, float e1, const vec4& a) {
    return vec4(smoothstep(e0, e1, a.x), smoothstep(e0, e1, a.y), smoothstep(e0, e1, a.z), smoothstep(e0, e1, a.w));
}
PTL_FN vec2 smoothstep(const vec2& e0, const vec2& e1, const vec2& a) { return vec2(smoothstep(e0.x, e1.x, a.x), smoothstep(e0.y, e1.y, a.y)); }
PTL_FN vec3 smoothstep(const vec3& e0, const vec3& e1, const vec3& a) {
    return vec3(smoothstep(e0.x, e1.x, a.x), smoothstep(e0.y, e1.y, a.y), smoothstep(e0.z, e1.z, a.z));
}
PTL_FN vec4 smoothstep(const vec4& e0, const vec4& e1, const vec4& a) {
    return vec4(smoothstep(e0.x, e1.x, a.x), smoothstep(e0.y, e1.y, a.y), smoothstep(e0.z, e1.z, a.z), smoothstep(e0.w, e1.w, a.w));
}

// geometric -----------------------------------------------------------------------------
PTL_FN float dot(const vec2& a, const vec2& b) { return ptl_term(a.y, b.y, ptl_term0(a.x, b.x)); }
PTL_FN float dot(const vec3& a, const vec3& b) { return ptl_term(a.z, b.z, ptl_term(a.y, b.y, ptl_term0(a.x, b.x))); }
PTL_FN float dot(const vec4& a, const vec4& b) { return ptl_term(a.w, b.w, ptl_term(a.z, b.z, ptl_term(a.y, b.y, ptl_term0(a.x, b.x)))); }
PTL_FN float length(float a) { return abs(a); }
PTL_FN float length(const vec2& a) { return sqrt(dot(a, a)); }
PTL_FN float length(const vec3& a) { return sqrt(dot(a, a)); }
PTL_FN float length(const vec4& a) { return sqrt(dot(a, a)); }
PTL_FN float distance(const vec2& a, const vec2& b) { return length(a - b); }
PTL_FN float distance(const vec3& a, const vec3& b) { return length(a - b); }
PTL_FN float distance(const vec4& a, const vec4& b) { return length(a - b); }
PTL_FN vec2 normalize(const vec2& a) { return a / length(a); }
PTL_FN vec3 normalize(const vec3& a) { return a / length(a); }
PTL_FN vec4 normalize(const vec4& a) { return a / length(a); }
PTL_FN vec3 cross(const vec3& a, const vec3& b) {
#if defined(PTL_CONTRACT_V1)
    return vec3(fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x)));
#else
    return vec3(ptl_term(a.y, b.z, ptl_term0(-a.z, b.y)), ptl_term(a.z, b.x, ptl_term0(-a.x, b.z)), ptl_term(a.x, b.y, ptl_term0(-a.y, b.x)));
#endif
}
PTL_FN vec2 reflect(const vec2& i, const vec2& n) { return i - n * (2.0f * dot(n, i)); }
PTL_FN vec3 reflect(const vec3& i, const vec3& n) { return i - n * (2.0f * dot(n, i)); }
PTL_FN vec3 refract(const vec3& i, const vec3& n, float eta) {
    float d = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return vec3(0.0f);
    return i * eta - n * (eta * d + sqrt(k));
}
PTL_FN vec3 faceforward(const vec3& n, const vec3& i, const vec3& nref) { return dot(nref, i) < 0.0f ? n : -n; }

// ---------------------------------------------------------------------------------------
// matrices (column-major, m[i] is column i, as in GLSL)
// ---------------------------------------------------------------------------------------
struct mat4;
struct mat2 {
    vec2 c[2];
    mat2() = default;
    PTL_FN explicit mat2(float d) { c[0] = vec2(d, 0.0f); c[1] = vec2(0.0f, d); }
    PTL_FN mat2(const vec2& a, const vec2& b) { c[0] = a; c[1] = b; }
    PTL_FN mat2(float a, float b, float cc, float d) { c[0] = vec2(a, b); c[1] = vec2(cc, d); }
    PTL_FN vec2& operator[](int i) { return c[i]; }
    PTL_FN const vec2& operator[](int i) const { return c[i]; }
};
struct mat3 {
    vec3 c[3];
    mat3() = default;
    PTL_FN explicit mat3(float d) { c[0] = vec3(d, 0.0f, 0.0f); c[1] = vec3(0.0f, d, 0.0f); c[2] = vec3(0.0f, 0.0f, d); }
    PTL_FN mat3(const vec3& a, const vec3& b, const vec3& cc) { c[0] = a; c[1] = b; c[2] = cc; }
    PTL_FN mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
        c[0] = vec3(a0, a1, a2); c[1] = vec3(b0, b1, b2); c[2] = vec3(c0, c1, c2);
    }
    PTL_FN explicit mat3(const mat4& m);
    PTL_FN vec3& operator[](int i) { return c[i]; }
    PTL_FN const vec3& operator[](int i) const { return c[i]; }
};
struct mat4 {
    vec4 c[4];
    mat4() = default;
    PTL_FN explicit mat4(float d) {
        c[0] = vec4(d, 0.0f, 0.0f, 0.0f); c[1] = vec4(0.0f, d, 0.0f, 0.0f);
        c[2] = vec4(0.0f, 0.0f, d, 0.0f); c[3] = vec4(0.0f, 0.0f, 0.0f, d);
    }
    PTL_FN mat4(const vec4& a, const vec4& b, const vec4& cc, const vec4& d) { c[0] = a; c[1] = b; c[2] = cc; c[3] = d; }
    PTL_FN mat4(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3,
                float c0, float c1, float c2, float c3, float d0, float d1, float d2, float d3) {
        c[0] = vec4(a0, a1, a2, a3); c[1] = vec4(b0, b1, b2, b3);
        c[2] = vec4(c0, c1, c2, c3); c[3] = vec4(d0, d1, d2, d3);
    }
    PTL_FN vec4& operator[](int i) { return c[i]; }
    PTL_FN const vec4& operator[](int i) const { return c[i]; }
};
PTL_FN mat3::mat3(const mat4& m) { c[0] = vec3(m.c[0]); c[1] = vec3(m.c[1]); c[2] = vec3(m.c[2]); }

PTL_FN vec2 operator*(const mat2& m, const vec2& v) {
    return vec2(ptl_mterm(m.c[1].x, v.y, ptl_mterm0(m.c[0].x, v.x)), ptl_mterm(m.c[1].y, v.y, ptl_mterm0(m.c[0].y, v.x)));
}
PTL_FN vec3 operator*(const mat3& m, const vec3& v) {
    return vec3(ptl_mterm(m.c[2].x, v.z, ptl_mterm(m.c[1].x, v.y, ptl_mterm0(m.c[0].x, v.x))),
                ptl_mterm(m.c[2].y, v.z, ptl_mterm(m.c[1].y, v.y, ptl_mterm0(m.c[0].y, v.x))),
                ptl_mterm(m.c[2].z, v.z, ptl_mterm(m.c[1].z, v.y, ptl_mterm0(m.c[0].z, v.x))));
}
#if PTL_DEVICE_BUILD && defined(PTL_PACKED_MATVEC)
// Two result components per instruction: v_pk_mul_f32 / v_pk_fma_f32 (gfx950 packed binary32: two IEEE operations per lane and
// issue slot, each rounded exactly like its scalar form) on the column halves (c[k].x, c[k].y) and (c[k].z, c[k].w), the vector
// component broadcast to both halves.  Same operations in the same order as the scalar form below.
typedef float ptl_f2 __attribute__((ext_vector_type(2)));
PTL_FN vec4 operator*(const mat4& m, const vec4& v) {
    const ptl_f2 lo = __builtin_elementwise_fma(ptl_f2{m.c[3].x, m.c[3].y}, (ptl_f2)(v.w), __builtin_elementwise_fma(ptl_f2{m.c[2].x, m.c[2].y}, (ptl_f2)(v.z),
                      __builtin_elementwise_fma(ptl_f2{m.c[1].x, m.c[1].y}, (ptl_f2)(v.y), ptl_f2{m.c[0].x, m.c[0].y} * (ptl_f2)(v.x))));
    const ptl_f2 hi = __builtin_elementwise_fma(ptl_f2{m.c[3].z, m.c[3].w}, (ptl_f2)(v.w), __builtin_elementwise_fma(ptl_f2{m.c[2].z, m.c[2].w}, (ptl_f2)(v.z),
                      __builtin_elementwise_fma(ptl_f2{m.c[1].z, m.c[1].w}, (ptl_f2)(v.y), ptl_f2{m.c[0].z, m.c[0].w} * (ptl_f2)(v.x))));
    return vec4(lo[0], lo[1], hi[0], hi[1]);
}
#else
PTL_FN vec4 operator*(const mat4& m, const vec4& v) {
    return vec4(ptl_mterm(m.c[3].x, v.w, ptl_mterm(m.c[2].x, v.z, ptl_mterm(m.c[1].x, v.y, ptl_mterm0(m.c[0].x, v.x)))),
                ptl_mterm(m.c[3].y, v.w, ptl_mterm(m.c[2].y, v.z, ptl_mterm(m.c[1].y, v.y, ptl_mterm0(m.c[0].y, v.x)))),
                ptl_mterm(m.c[3].z, v.w, ptl_mterm(m.c[2].z, v.z, ptl_mterm(m.c[1].z, v.y, ptl_mterm0(m.c[0].z, v.x)))),
                ptl_mterm(m.c[3].w, v.w, ptl_mterm(m.c[2].w, v.z, ptl_mterm(m.c[1].w, v.y, ptl_mterm0(m.c[0].w, v.x)))));
}
#endif
// The product for a matrix whose VALUES are run-time uniforms but whose ZERO PATTERN was known when the kernel was generated
// (KernelOptions::mask_zero_elements: bit 4 * column + row of MASK set = that element may be non-zero; the host rebuilds the kernel
// if a masked element ever stops being zero).  It skips exactly the terms ptl_mterm skips for a baked matrix -- same chain from +0,
// same deviation for non-finite vector components -- but decides at COMPILE time: no comparison is executed, and the skipped elements
// are never loaded.  Contract 1 and the tolerance mode never get a mask other than 0xffff (codegen.cpp).
template <int R> PTL_FN float ptl_comp(const vec4& v) {
    if constexpr (R == 0) return v.x;
    else if constexpr (R == 1) return v.y;
    else if constexpr (R == 2) return v.z;
    else return v.w;
}
// Round 4: the pattern also says which elements are exactly +1 or -1 (bits 16-31 and 32-47 of MASK, same element order; PTL_UNIT_BITS):
// the portal and wall matrices of the reference's scenes are translations and quarter turns, and a fully baked build gets `x + acc` /
// `acc - x` for such terms from constant folding (a chain of identity-diagonal products then collapses: (x + 0) + 0 is x + 0).  With the
// element's value known the term is the SAME operation on the same numbers -- fma(1, x, acc) -- so this part of the pattern changes no
// bit at all; the host rebuilds the kernel when such an element moves, as for the zeros (capi.cpp `zero_patterns_broken`).
typedef unsigned long long ptl_mask_t;
#define PTL_UNIT_BITS(ones, negs) ((ptl_mask_t)(ones) << 16 | (ptl_mask_t)(negs) << 32)
template <ptl_mask_t MASK, int K> PTL_FN float ptl_element(float loaded) {  // element K = 4 * column + row of a matrix with pattern MASK
    if constexpr ((MASK >> (16 + K)) & 1u) return 1.0f;
    else if constexpr ((MASK >> (32 + K)) & 1u) return -1.0f;
    else return loaded;
}
// Round 5, `W`: what is known about v.w.  Rays are affine objects -- every origin has w = 1, every direction w = 0 (the camera builds them
// so, library.glsl's materials keep them so, and an affine matrix maps them so) -- and in a kernel generated with PTL_AFFINE_RAYS
// (codegen.cpp `affine_rays`: every matrix of the scene has the bottom row 0 0 0 1, no scene snippet writes a ray's w) the products of a
// matrix with a ray's two halves say it: PTL_W_ONE = the fourth term is `acc + element` (the same fma with the operand's value spelled: no
// bit can move), PTL_W_ZERO = the fourth term is skipped like a term with a zero MATRIX element (exact for a finite element and an
// accumulator that is not -0: the stated deviation of the shortened products, under the same guard).  The row (0 0 0 1) then folds to the
// constants 1 / 0 at JIT time, so the w of a transformed ray costs nothing and a direction never pays for the translation column:
// headline kernel 0.280 -> 0.235 ms, identical frames (profiles/r05/stub_profile_a.jsonl `r5_w_known`).
enum { PTL_W_ZERO = 0, PTL_W_ONE = 1, PTL_W_ANY = 2 };
template <ptl_mask_t MASK, int R, int W = PTL_W_ANY> PTL_FN float ptl_row_m(const mat4& m, const vec4& v) {
    if constexpr (MASK == 0xffffu) {  // nothing known about the matrix: the ordinary chain of this build
        const float head = ptl_mterm(ptl_comp<R>(m.c[2]), v.z, ptl_mterm(ptl_comp<R>(m.c[1]), v.y, ptl_mterm0(ptl_comp<R>(m.c[0]), v.x)));
        if constexpr (W == PTL_W_ZERO) return head;
        else if constexpr (W == PTL_W_ONE) return ptl_mterm(ptl_comp<R>(m.c[3]), 1.0f, head);
        else return ptl_mterm(ptl_comp<R>(m.c[3]), v.w, head);
    } else {
        float acc = 0.0f;
        if constexpr ((MASK >> (0 + R)) & 1u) acc = __builtin_fmaf(ptl_element<MASK, 0 + R>(ptl_comp<R>(m.c[0])), v.x, acc);
        if constexpr ((MASK >> (4 + R)) & 1u) acc = __builtin_fmaf(ptl_element<MASK, 4 + R>(ptl_comp<R>(m.c[1])), v.y, acc);
        if constexpr ((MASK >> (8 + R)) & 1u) acc = __builtin_fmaf(ptl_element<MASK, 8 + R>(ptl_comp<R>(m.c[2])), v.z, acc);
        if constexpr (((MASK >> (12 + R)) & 1u) && W != PTL_W_ZERO) acc = __builtin_fmaf(ptl_element<MASK, 12 + R>(ptl_comp<R>(m.c[3])), W == PTL_W_ONE ? 1.0f : v.w, acc);
        return acc;
    }
}
template <ptl_mask_t MASK, int W = PTL_W_ANY> PTL_FN vec4 ptl_mul_m(const mat4& m, const vec4& v) {
    return vec4(ptl_row_m<MASK, 0, W>(m, v), ptl_row_m<MASK, 1, W>(m, v), ptl_row_m<MASK, 2, W>(m, v), ptl_row_m<MASK, 3, W>(m, v));
}
// matrix * (a ray's origin) and matrix * (a ray's direction): the plain products unless the kernel was generated with PTL_AFFINE_RAYS
#ifdef PTL_AFFINE_RAYS
#define PTL_W_OF_ORIGIN PTL_W_ONE
#define PTL_W_OF_DIRECTION PTL_W_ZERO
#else
#define PTL_W_OF_ORIGIN PTL_W_ANY
#define PTL_W_OF_DIRECTION PTL_W_ANY
#endif
// PTL_CHECK_AFFINE: a half that arrives with another w (a vector that is NaN throughout -- behind a switched-off object's all-NaN matrix -- gives NaN
// products whatever its w is taken for, and is not counted)
#ifdef PTL_CHECK_AFFINE
PTL_FN void ptl_check_w(const vec4& v, float w) {
    if (!(v.w == w) && (v.x == v.x || v.y == v.y || v.z == v.z || v.w == v.w)) PTL_NOTE_NOT_AFFINE();
}
#else
PTL_FN void ptl_check_w(const vec4&, float) {}
#endif
template <ptl_mask_t MASK = 0xffffu> PTL_FN vec4 ptl_mul_origin(const mat4& m, const vec4& o) {
    ptl_check_w(o, 1.0f);
    return ptl_mul_m<MASK, PTL_W_OF_ORIGIN>(m, o);
}
template <ptl_mask_t MASK = 0xffffu> PTL_FN vec4 ptl_mul_direction(const mat4& m, const vec4& d) {
    ptl_check_w(d, 0.0f);
    return ptl_mul_m<MASK, PTL_W_OF_DIRECTION>(m, d);
}
// (`X_mat * <anything else>` that the generator rewrote by its shape alone -- a matrix, a scalar: the ordinary product)
template <ptl_mask_t MASK, class T> PTL_FN auto ptl_mul_m(const mat4& m, const T& x) -> decltype(m * x) { return m * x; }
// the same product for a matrix that is a run-time value in every build (the camera): no zero tests (they would be executed)
PTL_FN vec4 ptl_mul_runtime(const mat4& m, const vec4& v) {
    return vec4(ptl_term(m.c[3].x, v.w, ptl_term(m.c[2].x, v.z, ptl_term(m.c[1].x, v.y, ptl_term0(m.c[0].x, v.x)))),
                ptl_term(m.c[3].y, v.w, ptl_term(m.c[2].y, v.z, ptl_term(m.c[1].y, v.y, ptl_term0(m.c[0].y, v.x)))),
                ptl_term(m.c[3].z, v.w, ptl_term(m.c[2].z, v.z, ptl_term(m.c[1].z, v.y, ptl_term0(m.c[0].z, v.x)))),
                ptl_term(m.c[3].w, v.w, ptl_term(m.c[2].w, v.z, ptl_term(m.c[1].w, v.y, ptl_term0(m.c[0].w, v.x)))));
}
// ... times a DIRECTION, in a kernel with affine rays: the three terms of the upper-left 3 x 3 block, w = 0 (the matrix is affine: capi.cpp `camera_is_affine`)
PTL_FN vec4 ptl_mul_runtime_direction(const mat4& m, const vec4& v) {
    return vec4(ptl_term(m.c[2].x, v.z, ptl_term(m.c[1].x, v.y, ptl_term0(m.c[0].x, v.x))),
                ptl_term(m.c[2].y, v.z, ptl_term(m.c[1].y, v.y, ptl_term0(m.c[0].y, v.x))),
                ptl_term(m.c[2].z, v.z, ptl_term(m.c[1].z, v.y, ptl_term0(m.c[0].z, v.x))), 0.0f);
}
// row vector times matrix: component i is dot(v, column i)
PTL_FN vec2 operator*(const vec2& v, const mat2& m) { return vec2(dot(v, m.c[0]), dot(v, m.c[1])); }
PTL_FN vec3 operator*(const vec3& v, const mat3& m) { return vec3(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])); }
PTL_FN vec4 operator*(const vec4& v, const mat4& m) { return vec4(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2]), dot(v, m.c[3])); }
PTL_FN mat2 operator*(const mat2& a, const mat2& b) { return mat2(a * b.c[0], a * b.c[1]); }
PTL_FN mat3 operator*(const mat3& a, const mat3& b) { return mat3(a * b.c[0], a * b.c[1], a * b.c[2]); }
PTL_FN mat4 operator*(const mat4& a, const mat4& b) { return mat4(a * b.c[0], a * b.c[1], a * b.c[2], a * b.c[3]); }
PTL_FN mat2 operator*(const mat2& a, float s) { return mat2(a.c[0] * s, a.c[1] * s); }
PTL_FN mat3 operator*(const mat3& a, float s) { return mat3(a.c[0] * s, a.c[1] * s, a.c[2] * s); }
PTL_FN mat4 operator*(const mat4& a, float s) { return mat4(a.c[0] * s, a.c[1] * s, a.c[2] * s, a.c[3] * s); }
PTL_FN mat2 operator*(float s, const mat2& a) { return a * s; }
PTL_FN mat3 operator*(float s, const mat3& a) { return a * s; }
PTL_FN mat4 operator*(float s, const mat4& a) { return a * s; }
PTL_FN mat2 operator+(const mat2& a, const mat2& b) { return mat2(a.c[0] + b.c[0], a.c[1] + b.c[1]); }
PTL_FN mat3 operator+(const mat3& a, const mat3& b) { return mat3(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]); }
PTL_FN mat4 operator+(const mat4& a, const mat4& b) { return mat4(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2], a.c[3] + b.c[3]); }
PTL_FN mat2 operator-(const mat2& a, const mat2& b) { return mat2(a.c[0] - b.c[0], a.c[1] - b.c[1]); }
PTL_FN mat3 operator-(const mat3& a, const mat3& b) { return mat3(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]); }
PTL_FN mat4 operator-(const mat4& a, const mat4& b) { return mat4(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2], a.c[3] - b.c[3]); }
PTL_FN mat2& operator*=(mat2& a, const mat2& b) { a = a * b; return a; }
PTL_FN mat3& operator*=(mat3& a, const mat3& b) { a = a * b; return a; }
PTL_FN mat4& operator*=(mat4& a, const mat4& b) { a = a * b; return a; }
PTL_FN vec2& operator*=(vec2& v, const mat2& m) { v = v * m; return v; }
PTL_FN vec3& operator*=(vec3& v, const mat3& m) { v = v * m; return v; }
PTL_FN vec4& operator*=(vec4& v, const mat4& m) { v = v * m; return v; }
// transpose / determinant / inverse are templates with a dummy parameter: some scenes ship their own
// `mat3 inverse(mat3)` (GLSL lets user code shadow these), and a non-template function wins overload
// resolution against a template, so the scene's definition is the one that gets called.
template <class PtlBuiltin = void> PTL_FN mat2 transpose(const mat2& m) { return mat2(m.c[0].x, m.c[1].x, m.c[0].y, m.c[1].y); }
template <class PtlBuiltin = void> PTL_FN mat3 transpose(const mat3& m) {
    return mat3(m.c[0].x, m.c[1].x, m.c[2].x, m.c[0].y, m.c[1].y, m.c[2].y, m.c[0].z, m.c[1].z, m.c[2].z);
}
template <class PtlBuiltin = void> PTL_FN mat4 transpose(const mat4& m) {
    return mat4(m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x, m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y,
                m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z, m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w);
}
template <class PtlBuiltin = void> PTL_FN float determinant(const mat2& m) { return m.c[0].x * m.c[1].y - m.c[1].x * m.c[0].y; }
template <class PtlBuiltin = void> PTL_FN float determinant(const mat3& m) { return dot(m.c[0], cross(m.c[1], m.c[2])); }
template <class PtlBuiltin = void> PTL_FN mat2 inverse(const mat2& m) {
    float i = ptl_rcp(determinant(m));
    return mat2(m.c[1].y * i, -m.c[0].y * i, -m.c[1].x * i, m.c[0].x * i);
}
template <class PtlBuiltin = void> PTL_FN mat3 inverse(const mat3& m) {
    vec3 r0 = cross(m.c[1], m.c[2]), r1 = cross(m.c[2], m.c[0]), r2 = cross(m.c[0], m.c[1]);
    float i = ptl_rcp(dot(m.c[0], r0));
    return mat3(r0.x * i, r1.x * i, r2.x * i, r0.y * i, r1.y * i, r2.y * i, r0.z * i, r1.z * i, r2.z * i);
}

// ---------------------------------------------------------------------------------------
// textures: RGBA8, bilinear, clamp-to-edge, texel centres at +0.5, row 0 at v = 0
// (macroquad Texture2D::from_file_with_format defaults; reference src/main.rs:1066-1083).
// ---------------------------------------------------------------------------------------
struct sampler2D {
    const unsigned char* texels;
    int width;
    int height;
};
PTL_FN vec4 ptl_texel(const sampler2D& s, int ix, int iy) {
    ix = clamp(ix, 0, s.width - 1);
    iy = clamp(iy, 0, s.height - 1);
    const unsigned char* p = s.texels + 4 * ((long)iy * s.width + ix);
    return vec4(ptl_div((float)p[0], 255.0f), ptl_div((float)p[1], 255.0f), ptl_div((float)p[2], 255.0f), ptl_div((float)p[3], 255.0f));
}
PTL_FN vec4 texture(const sampler2D& s, const vec2& uv) {
    if (s.texels == nullptr || s.width <= 0 || s.height <= 0) return vec4(0.0f, 0.0f, 0.0f, 1.0f);
    float x = uv.x * (float)s.width - 0.5f;
    float y = uv.y * (float)s.height - 0.5f;
    if (x != x) x = 0.0f;
    if (y != y) y = 0.0f;
    x = clamp(x, -1.0f, (float)s.width);
    y = clamp(y, -1.0f, (float)s.height);
    float x0 = floor(x), y0 = floor(y);
    float fx = x - x0, fy = y - y0;
    int ix = (int)x0, iy = (int)y0;
    vec4 a = mix(ptl_texel(s, ix, iy), ptl_texel(s, ix + 1, iy), fx);
    vec4 b = mix(ptl_texel(s, ix, iy + 1), ptl_texel(s, ix + 1, iy + 1), fx);
    return mix(a, b, fy);
}

}  // namespace glsl
