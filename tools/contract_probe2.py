#!/usr/bin/env python3
"""tools/contract_probe2.py -- round 3, second pass over the contract-2 sequences of device/ptl_glsl.h: the same results from fewer
issue cycles.  tools/contract_probe.py found WHICH short sequences are correctly rounded; here the question is only how to spell the
two things around them that are 4-cycle instructions on gfx950 (profiles/r01/valu_rates.jsonl): the "keep the estimate where the
correction is NaN" compare + select, and the second transcendental of `1 / sqrt(x)`.

  * `v_med3_f32(y, y0, y)` is y when nothing is NaN (the median of {y, y, y0}) and MIN3 = the operand that is not NaN otherwise
    (ISA: "if any source is NaN, D = MIN3(S0, S1, S2)", and V_MIN_F32 returns the other operand for a quiet NaN): one VOP3
    instruction for `y == y ? y : y0`.
  * `1 / sqrt(x)`: the v_rsq_f32 estimate the square root's correction already computes is a ~2 ulp estimate of 1 / s as well
    (s = the correctly rounded root), so the reciprocal's v_rcp_f32 can be dropped IF one residual step from that seed still
    rounds correctly for every x -- which only an exhaustive run can say.

Every candidate is compared on ALL 2^32 bit patterns with the shipped function it would replace (bit for bit, NaN == NaN), then timed
like contract_probe.py (8 dependent chains per lane, net of the chain's add).

    python tools/contract_probe2.py            # on the GPU box; JSON lines
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contract_probe as cp  # noqa: E402

CANDIDATES = r"""
PTL_FN float rcp_shipped(float x) { return ptl_rcp(x); }
PTL_FN float sqrt_shipped(float x) { return sqrt(x); }
PTL_FN float rsqrt_shipped(float x) { return ptl_rcp(sqrt(x)); }
PTL_FN float rcp_cmp(float x) {                                  // round 3's first form: compare + select
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y = __builtin_fmaf(y0, e, y0);
    return y == y ? y : y0;
}
PTL_FN float rcp_med3(float x) {
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y = __builtin_fmaf(y0, e, y0);
    return __builtin_amdgcn_fmed3f(y, y0, y);
}
PTL_FN float sqrt_cmp(float x) {
    const float xe = __builtin_fabsf(x) < 0x1p-100f ? 0.0f : x;
    const float g = __builtin_amdgcn_sqrtf(xe);
    const float h = 0.5f * __builtin_amdgcn_rsqf(xe);
    const float d = __builtin_fmaf(-g, g, xe);
    const float s = __builtin_fmaf(d, h, g);
    return s == s ? s : g;
}
PTL_FN float sqrt_med3(float x) {
    const float xe = __builtin_fabsf(x) < 0x1p-100f ? 0.0f : x;
    const float g = __builtin_amdgcn_sqrtf(xe);
    const float h = 0.5f * __builtin_amdgcn_rsqf(xe);
    const float d = __builtin_fmaf(-g, g, xe);
    const float s = __builtin_fmaf(d, h, g);
    return __builtin_amdgcn_fmed3f(s, g, s);
}
// 1 / sqrt(x) with the reciprocal seeded by the rsq estimate (no v_rcp_f32)
PTL_FN float rsqrt_fused(float x) {
    const float xe = __builtin_fabsf(x) < 0x1p-100f ? 0.0f : x;
    const float g = __builtin_amdgcn_sqrtf(xe);
    const float r = __builtin_amdgcn_rsqf(xe);
    const float h = 0.5f * r;
    const float d = __builtin_fmaf(-g, g, xe);
    const float s0 = __builtin_fmaf(d, h, g);
    const float s = __builtin_amdgcn_fmed3f(s0, g, s0);
    const float e = __builtin_fmaf(-s, r, 1.0f);
    const float y = __builtin_fmaf(r, e, r);
    return __builtin_amdgcn_fmed3f(y, r, y);
}
// the same with two residual steps (in case one is not enough from a 2 ulp seed)
PTL_FN float rsqrt_fused2(float x) {
    const float xe = __builtin_fabsf(x) < 0x1p-100f ? 0.0f : x;
    const float g = __builtin_amdgcn_sqrtf(xe);
    const float r = __builtin_amdgcn_rsqf(xe);
    const float h = 0.5f * r;
    const float d = __builtin_fmaf(-g, g, xe);
    const float s0 = __builtin_fmaf(d, h, g);
    const float s = __builtin_amdgcn_fmed3f(s0, g, s0);
    const float e1 = __builtin_fmaf(-s, r, 1.0f);
    const float r1 = __builtin_fmaf(r, e1, r);
    const float e = __builtin_fmaf(-s, r1, 1.0f);
    const float y = __builtin_fmaf(r1, e, r1);
    return __builtin_amdgcn_fmed3f(y, r, y);
}
PTL_FN float med3_alone(float x) { return __builtin_amdgcn_fmed3f(x, 1.5f, x); }
PTL_FN float cmp_select(float x) { return x == x ? x : 1.5f; }
"""

CHECKS = [("rcp_cmp", "rcp_shipped(x)"), ("rcp_med3", "rcp_shipped(x)"), ("sqrt_cmp", "sqrt_shipped(x)"), ("sqrt_med3", "sqrt_shipped(x)"),
          ("rsqrt_fused", "rsqrt_shipped(x)"), ("rsqrt_fused2", "rsqrt_shipped(x)")]

TIMING = cp.TIMING.split("%(candidates)s")[0] + "%(candidates)s\n" + r"""
PTL_FN float op_fma(float x, float k) { return __builtin_fmaf(x, k, k); }
#define UNARY(name) PTL_FN float op_##name(float x, float k) { return name(x) + k; }
UNARY(rcp_cmp) UNARY(rcp_med3) UNARY(sqrt_cmp) UNARY(sqrt_med3) UNARY(rsqrt_shipped) UNARY(rsqrt_fused) UNARY(rsqrt_fused2) UNARY(med3_alone) UNARY(cmp_select)
""" + "PTL_FN vec4 shade_pixel" + cp.TIMING.split("PTL_FN vec4 shade_pixel")[1]
TIMED = ["op_fma", "op_med3_alone", "op_cmp_select", "op_rcp_cmp", "op_rcp_med3", "op_sqrt_cmp", "op_sqrt_med3", "op_rsqrt_shipped", "op_rsqrt_fused", "op_rsqrt_fused2"]


def main():
    import portal_amd as pa

    device = int(os.environ.get("PTL_DEVICE", "0"))
    classes = ["middle 2^-126<=|x|<=2^126", "subnormal input", "|x|>2^126 finite", "zero/inf/nan"]
    for fn, want in CHECKS:
        k = cp.kernel(pa, cp.EXHAUSTIVE % dict(candidates=CANDIDATES, fn=fn, want=want), device)
        if device < 0:
            continue
        out = k.render(4096, 256, rgba8=False, rgba32f=True)
        bad = out["rgba32f"].reshape(-1, 4).astype(np.float64).sum(axis=0)
        print(json.dumps({"exhaustive": fn, "against": want, "mismatches": {c: int(b) for c, b in zip(classes, bad)}, "ms": round(out["ms"], 2)}), flush=True)
    base = None
    for fn in TIMED:
        k = cp.kernel(pa, TIMING % dict(candidates=CANDIDATES, fn=fn, iter=cp.ITER), device)
        if device < 0:
            continue
        runs = [k.render(4096, 256, rgba8=False, rgba32f=True) for _ in range(4)]
        best = min(runs, key=lambda r: r["ms"])
        px = best["rgba32f"].reshape(-1, 4)
        mhz = float(np.median(px[:, 1] / np.maximum(px[:, 2], 1.0)) * 100.0)
        calls = (4096 * 256 // 64) * cp.ITER * 8
        cycles = best["ms"] * 1e-3 * mhz * 1e6 * 1024 / calls
        base = cycles if fn == "op_fma" else base
        print(json.dumps({"timing": fn, "ms": round(best["ms"], 3), "shader_clock_mhz": round(mhz), "cycles_per_call_per_simd": round(cycles, 1),
                          "net_of_the_chain_add": round(cycles - (base or 0.0), 1)}), flush=True)


if __name__ == "__main__":
    main()
