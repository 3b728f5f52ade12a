#!/usr/bin/env python3
"""tools/valu_rates.py -- issue cost of single gfx950 VALU instructions, measured (development aid).

Which instruction sequences are cheap is decided by what each instruction costs the SIMD, and the ISA guide does not say for the
division helpers (v_div_scale / v_div_fmas / v_div_fixup) or the transcendental unit.  One hand-written kernel per instruction
(layer 1 of the C ABI): every lane runs `ITER` iterations of 8 independent chains of the instruction (inline asm, so nothing is
folded), 2^20 lanes.  Printed: time, the shader clock while the loop ran (s_memtime ticks per s_memrealtime tick) and cycles per wave-instruction per SIMD --
v_fma_f32 is the yardstick (a wave64 FP32 instruction issues every 2 cycles on CDNA4).
"""
import json, os, sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

OPS = {  # name -> asm for one instruction on register %0 (chains are independent registers); $b = a second live register
    "v_fma_f32": "v_fma_f32 %0, %0, %1, %0",
    "v_mul_f32": "v_mul_f32 %0, %0, %1",
    "v_add_f32": "v_add_f32 %0, %0, %1",
    "v_pk_fma_f32": None,  # filled below (64-bit operands)
    "v_rcp_f32": "v_rcp_f32 %0, %0",
    "v_sqrt_f32": "v_sqrt_f32 %0, %0",
    "v_rsq_f32": "v_rsq_f32 %0, %0",
    "v_div_scale_f32": "v_div_scale_f32 %0, vcc, %0, %1, %0",
    "v_div_fmas_f32": "v_div_fmas_f32 %0, %0, %1, %0",
    "v_div_fixup_f32": "v_div_fixup_f32 %0, %0, %1, %0",
    "v_cmp_lt_f32": "v_cmp_lt_f32 vcc, %0, %1",
    "v_cndmask_b32 (vcc never written: see the pairs below)": "v_cndmask_b32 %0, %0, %1, vcc",
    "v_cmp + s_nop 1 + v_cndmask (vcc)": "v_cmp_lt_f32 vcc, %0, %1\\n s_nop 1\\n v_cndmask_b32 %0, %0, %1, vcc",
    "v_cmp_e64 + s_nop 1 + v_cndmask_e64 (sgpr pair)": "v_cmp_lt_f32 s[40:41], %0, %1\\n s_nop 1\\n v_cndmask_b32 %0, %0, %1, s[40:41]",
    "v_add_u32": "v_add_u32 %0, %0, %1",
    "v_floor_f32": "v_floor_f32 %0, %0",
    "v_mov_b32": "v_mov_b32 %0, %1",
    "v_ldexp_f32": "v_ldexp_f32 %0, %0, %1",
    "v_frexp_mant_f32": "v_frexp_mant_f32 %0, %0",
    "v_cmp_class_f32": "v_cmp_class_f32 vcc, %0, %1",
    "s_nop_0": "s_nop 0",
}
ITER, CHAINS, UNROLL = 2048, 8, 4
W, H = 4096, 256  # 2^20 lanes = 16384 waves = 16 per SIMD


def source(pa, asm):
    body = "\n".join(f'            asm volatile("{asm}" : "+v"(r[{c}]) : "v"(k) : "vcc", "s40", "s41");' for _ in range(UNROLL) for c in range(CHAINS))
    return pa.device_source("glsl") + r"""
#define PTL_COUNT_SEGMENT() ((void)0)
#define PTL_NO_TELEPORT_ENTRY 1
namespace glsl {
struct ptl_uniform_block { int seed_u; int pad_u; };
__constant__ ptl_uniform_block ptl_u;
PTL_FN vec4 shade_pixel(vec2 position) {
    float r[%d];
    for (int c = 0; c < %d; ++c) r[c] = position.x * 0.001f + position.y + (float)c + 1.5f;
    float k = 1.0000001f + (float)ptl_u.seed_u;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();  // shader clock / 100 MHz
    for (int i = 0; i < %d; ++i) {
%s
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.0f;
    for (int c = 0; c < %d; ++c) acc += r[c];
    return vec4(acc, (float)(c1 - c0), (float)(t1 - t0), 1.0f);
}
PTL_FN unsigned int pack_rgba8(vec4 c) { return 0u; }
}  // namespace glsl
""" % (CHAINS, CHAINS, ITER // UNROLL, body, CHAINS) + pa.device_source("entry")


def main():
    import portal_amd as pa

    device = int(os.environ.get("PTL_DEVICE", "0"))
    results = {}
    for name, asm in OPS.items():
        if asm is None:
            continue
        try:
            k = pa.Kernel(source(pa, asm), [("seed_u", 2, 0), ("pad_u", 2, 4)], 8, device=device)
        except pa.PortalError as e:
            print(json.dumps({"op": name, "error": str(e)[-300:]}), flush=True)
            continue
        if device < 0:
            continue
        runs = [k.render(W, H, rgba8=False, rgba32f=True) for _ in range(4)]
        best = min(runs, key=lambda r: r["ms"])
        ms = best["ms"]
        px = best["rgba32f"].reshape(-1, 4)
        mhz = float(np.median(px[:, 1] / np.maximum(px[:, 2], 1.0)) * 100.0)  # shader-clock ticks per 100 MHz tick, while the loop ran
        wave_instr = (W * H // 64) * ITER * CHAINS
        cycles = ms * 1e-3 * mhz * 1e6 * 1024 / wave_instr  # 256 CUs x 4 SIMDs, at the clock the loop actually ran at
        results[name] = cycles
        print(json.dumps({"op": name, "ms": round(ms, 4), "shader_clock_mhz": round(mhz), "cycles_per_wave_instr_per_simd": round(cycles, 2)}), flush=True)
    if "v_fma_f32" in results:
        base = results["v_fma_f32"]
        print(json.dumps({"relative_to_v_fma_f32": {k: round(v / base, 2) for k, v in results.items()}}), flush=True)


if __name__ == "__main__":
    main()
