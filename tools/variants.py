#!/usr/bin/env python3
"""tools/variants.py -- time kernel build variants on the GPU (development aid, not a test).

usage: python tools/variants.py [scene:W:H:depth:aa ...]   (defaults: the BASELINE configs)
Variants are hiprtc flag sets passed through PTL_HIPRTC_FLAGS; each is compiled (cached),
run 6 times, and the median kernel time, register counts and an output checksum are printed
(the checksum must not change between variants: they are all the same arithmetic).
"""
import hashlib, os, re, subprocess, sys, json
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANTS = {
    "base": "",
    "base_w3": "-DPTL_WAVES_PER_EU=3",
    "base_w4": "-DPTL_WAVES_PER_EU=4",
    "hoist": "-DPTL_UNIFORM_HOIST",
    "reload": "-DPTL_UNIFORM_RELOAD",
    "spec": "SPECIALIZE",
    "all": "SPECIALIZE_ALL",
    "all_w4": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4",
    "base_O1": "-O1",
    "all_O1": "SPECIALIZE_ALL -O1",
    "all_O1_w4": "SPECIALIZE_ALL -O1 -DPTL_WAVES_PER_EU=4",
    "all_O1_w5": "SPECIALIZE_ALL -O1 -DPTL_WAVES_PER_EU=5",
    "all_O1_w6": "SPECIALIZE_ALL -O1 -DPTL_WAVES_PER_EU=6",
    "all_O1_slp": "SPECIALIZE_ALL -O1 -fslp-vectorize",
    "all_O2_novec": "SPECIALIZE_ALL -O2 -fno-slp-vectorize -fno-vectorize -fno-unroll-loops",
    "all_O1_nosched": "SPECIALIZE_ALL -O1 -mllvm -amdgpu-disable-unclustered-high-rp-reschedule",
    "base_O1_w4": "-O1 -DPTL_WAVES_PER_EU=4",
    "all_ilp": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=max-ilp",
    "all_memclause": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=max-memory-clause",
    "all_iter_minreg": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg",
    "all_iter_ilp": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-ilp",
    "all_w5": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=5",
    "all_nomisched": "SPECIALIZE_ALL -mllvm -enable-misched=false",
    "all_O0ish": "SPECIALIZE_ALL -O1 -mllvm -disable-licm-promotion -mllvm -enable-gvn-hoist=false",
    "all_ifcvt": "SPECIALIZE_ALL -mllvm -amdgpu-early-ifcvt=1",
    "all_skip4": "SPECIALIZE_ALL -mllvm -amdgpu-skip-threshold=4",
    "all_skip64": "SPECIALIZE_ALL -mllvm -amdgpu-skip-threshold=64",
    "all_bias0": "SPECIALIZE_ALL -mllvm -amdgpu-schedule-metric-bias=0",
    "all_bias100": "SPECIALIZE_ALL -mllvm -amdgpu-schedule-metric-bias=100",
    "all_nopostsched": "SPECIALIZE_ALL -mllvm -enable-post-misched=0",
    "all_minreg_w4": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg -DPTL_WAVES_PER_EU=4",
    "all_minreg_ifcvt": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg -mllvm -amdgpu-early-ifcvt=1",
    "all_relaxocc": "SPECIALIZE_ALL -mllvm -amdgpu-schedule-relaxed-occupancy=true",
    "all_xcd": "SPECIALIZE_ALL -DPTL_XCD_SWIZZLE",
    "all_bw1": "SPECIALIZE_ALL BLOCK_WAVES=1",
    "all_bw2": "SPECIALIZE_ALL BLOCK_WAVES=2",
    "base_xcd": "-DPTL_XCD_SWIZZLE",
    "all_Os": "SPECIALIZE_ALL -Os",
    "all_Oz": "SPECIALIZE_ALL -Oz",
    "base_Os": "-Os",
    "base_O1_w3": "-O1 -DPTL_WAVES_PER_EU=3",
    "all_ra_default": "SPECIALIZE_ALL RA_DEFAULT",   # the toolchain's own (greedy) allocator: faster by 0-3 %, but it miscompiles
    "base_ra_default": "RA_DEFAULT",
    "all_rabasic": "SPECIALIZE_ALL -mllvm -vgpr-regalloc=basic",
    "all_minreg_rabasic": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg -mllvm -vgpr-regalloc=basic",
    "all_w4_rabasic": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -vgpr-regalloc=basic",
    "base_rabasic": "-mllvm -vgpr-regalloc=basic",
    "all_plain": "SPECIALIZE_ALL -DPTL_PLAIN_SQRT_RCP",
    "base_plain": "-DPTL_PLAIN_SQRT_RCP",
    "all_minreg_plain": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg -DPTL_PLAIN_SQRT_RCP",
    # round 2: uniform prologue (derived uniforms), explicit wave loop, tolerance mode
    "r2_dyn": "",
    "r2_dyn_w4": "-DPTL_WAVES_PER_EU=4",
    "r2_dyn_noderived": "NO_DERIVED",
    "r2_dyn_noderived_w4": "NO_DERIVED -DPTL_WAVES_PER_EU=4",
    "r2_dyn_nowaveloop": "-DPTL_NO_WAVE_LOOP",
    "r2_dyn_nowaveloop_noderived": "NO_DERIVED -DPTL_NO_WAVE_LOOP",
    "r2_all": "SPECIALIZE_ALL",
    "r2_all_nowaveloop": "SPECIALIZE_ALL -DPTL_NO_WAVE_LOOP",
    "r2_all_noderived": "SPECIALIZE_ALL NO_DERIVED",
    "r2_all_pt": "SPECIALIZE_ALL -DPTL_PACKED_TRANSFORM",
    "r2_all_pt_w3": "SPECIALIZE_ALL -DPTL_PACKED_TRANSFORM -DPTL_WAVES_PER_EU=3",
    "r2_all_pt_w4": "SPECIALIZE_ALL -DPTL_PACKED_TRANSFORM -DPTL_WAVES_PER_EU=4",
    "r2_all_pm": "SPECIALIZE_ALL -DPTL_PACKED_MATVEC",
    "r2_all_pm_w3": "SPECIALIZE_ALL -DPTL_PACKED_MATVEC -DPTL_WAVES_PER_EU=3",
    "r2_all_pm_w4": "SPECIALIZE_ALL -DPTL_PACKED_MATVEC -DPTL_WAVES_PER_EU=4",
    "r2_all_w3": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=3",
    "r2_all_w4": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4",
    "r2_dyn_pt": "-DPTL_PACKED_TRANSFORM",
    "r2_dyn_pm": "-DPTL_PACKED_MATVEC",
    "r2_dyn_w3": "-DPTL_WAVES_PER_EU=3",
    "r2_dyn_pt_w3": "-DPTL_PACKED_TRANSFORM -DPTL_WAVES_PER_EU=3",
    "r2_dyn_pm_w3": "-DPTL_PACKED_MATVEC -DPTL_WAVES_PER_EU=3",
    # scalar uniform loads (offset laundering, ptl_trace.tpl PTL_RELAUNDER) are the default since variants7; the round-1 form for A/B:
    "r2_dyn_lp": "-DPTL_LAUNDER_POINTER",
    "r2_dyn_lp_w3": "-DPTL_LAUNDER_POINTER -DPTL_WAVES_PER_EU=3",
    "r2_ints_lp": "SPECIALIZE -DPTL_LAUNDER_POINTER",
    "r2_all_lp": "SPECIALIZE_ALL -DPTL_LAUNDER_POINTER",
    "r2_dyn_w4": "-DPTL_WAVES_PER_EU=4",
    "r2_dyn_w5": "-DPTL_WAVES_PER_EU=5",
    "r2_ints_w4": "SPECIALIZE -DPTL_WAVES_PER_EU=4",
    "r2_dyn_npr": "-DPTL_NO_PIXEL_RELAUNDER",
    "r2_ints_npr": "SPECIALIZE -DPTL_NO_PIXEL_RELAUNDER",
    "r2_all_npr": "SPECIALIZE_ALL -DPTL_NO_PIXEL_RELAUNDER",
    "r2_dyn_nohoist": "NO_HOIST",
    "r2_ints_nohoist": "SPECIALIZE NO_HOIST",
    "r2_dyn_nohoist_w5": "NO_HOIST -DPTL_WAVES_PER_EU=5",
    "r2_dyn_O2": "-O2", "r2_dyn_O3": "-O3", "r2_dyn_Os": "-Os", "r2_dyn_minreg": "-mllvm -amdgpu-sched-strategy=iterative-minreg", "r2_dyn_ilp": "-mllvm -amdgpu-sched-strategy=max-ilp",
    "r2_ints_O2": "SPECIALIZE -O2", "r2_ints_O3": "SPECIALIZE -O3", "r2_ints_Os": "SPECIALIZE -Os", "r2_ints_minreg": "SPECIALIZE -mllvm -amdgpu-sched-strategy=iterative-minreg",
    "r2_all_O2": "SPECIALIZE_ALL -O2", "r2_all_O3": "SPECIALIZE_ALL -O3",
    # first-trip snippet variants are the default since variants13 (there: "_ft" = with them); the general form alone for A/B:
    "r2_all_noft": "SPECIALIZE_ALL NO_FIRST_TRIP", "r2_all_noft_w4": "SPECIALIZE_ALL NO_FIRST_TRIP -DPTL_WAVES_PER_EU=4",
    "r2_ints_noft": "SPECIALIZE NO_FIRST_TRIP", "r2_dyn_noft": "NO_FIRST_TRIP",
    "r2_ints": "SPECIALIZE",
    "r2_ints_noderived": "SPECIALIZE NO_DERIVED",
    "r2_dyn_nocull": "-DPTL_NO_PLANE_CULL",
    "r2_all_nocull": "SPECIALIZE_ALL -DPTL_NO_PLANE_CULL",
    "r2_ints_nocull": "SPECIALIZE -DPTL_NO_PLANE_CULL",
    "r2_fast_all_nocull": "FAST SPECIALIZE_ALL -DPTL_NO_PLANE_CULL",
    "r2_all_minreg": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg",
    "r2_all_minreg_nocull": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg -DPTL_NO_PLANE_CULL",
    "r2_dyn_nodefer": "NO_DEFER",
    "r2_all_nodefer": "SPECIALIZE_ALL NO_DEFER",
    "r2_fast_all_nodefer": "FAST SPECIALIZE_ALL NO_DEFER",
    "r2_fast_dyn": "FAST",
    "r2_fast_dyn_w4": "FAST -DPTL_WAVES_PER_EU=4",
    "r2_fast_all": "FAST SPECIALIZE_ALL",
    "r2_fast_all_w4": "FAST SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4",
    # round 3: numerics contract 2 is the default; EXACT_CR = contract 1 (FLAG_EXACT_CR), the round-2 arithmetic
    "r3_all": "SPECIALIZE_ALL", "r3_all_w3": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=3", "r3_all_w4": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4", "r3_all_w5": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=5",
    "r3_v1_all": "EXACT_CR SPECIALIZE_ALL", "r3_v1_all_w4": "EXACT_CR SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4",
    "r3_dyn": "", "r3_v1_dyn": "EXACT_CR", "r3_ints": "SPECIALIZE", "r3_v1_ints": "EXACT_CR SPECIALIZE",
    "r3_fast_all_w4": "FAST SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4",
    # upper bound of what folding products with literal zeros could give the EXACT kernel (not a shippable build: nnan also deletes the NaN selects of 1/x and sqrt)
    "r3_all_w6": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=6", "r3_all_minreg": "SPECIALIZE_ALL -mllvm -amdgpu-sched-strategy=iterative-minreg",
    "r3_dyn_mulfirst": "-DPTL_CHAIN_FROM_PRODUCT", "r3_all_mulfirst": "SPECIALIZE_ALL -DPTL_CHAIN_FROM_PRODUCT",
    "r3_all_nounroll": "SPECIALIZE_ALL NO_UNROLL", "r3_all_w4_nounroll": "SPECIALIZE_ALL NO_UNROLL -DPTL_WAVES_PER_EU=4", "r3_ints_nounroll": "SPECIALIZE NO_UNROLL",
    "r3_all_w4_ra_default": "SPECIALIZE_ALL RA_DEFAULT -DPTL_WAVES_PER_EU=4", "r3_all_ra_default": "SPECIALIZE_ALL RA_DEFAULT", "r3_all_w4_ra_fast": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -vgpr-regalloc=fast",
    "r3_dyn_ra_default": "RA_DEFAULT", "r3_ints_ra_default": "SPECIALIZE RA_DEFAULT",
    "r3_all_noftp": "SPECIALIZE_ALL NO_FTP", "r3_all_w4_noftp": "SPECIALIZE_ALL NO_FTP -DPTL_WAVES_PER_EU=4", "r3_ints_noftp": "SPECIALIZE NO_FTP", "r3_dyn_noftp": "NO_FTP",
    "r3_all_w4_ifcvt": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -amdgpu-early-ifcvt=1", "r3_all_w4_O2": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -O2 -fno-slp-vectorize",
    "r3_all_w4_O3": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -O3 -fno-slp-vectorize", "r3_all_w4_Os": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -Os",
    "r3_all_w4_skip4": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -amdgpu-skip-threshold=4", "r3_all_w4_skip32": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -amdgpu-skip-threshold=32",
    "r3_all_w4_ilp": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -amdgpu-sched-strategy=max-ilp", "r3_all_w4_nomisched": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -enable-misched=false",
    "r3_dyn_O2": "-O2 -fno-slp-vectorize", "r3_dyn_O3": "-O3 -fno-slp-vectorize", "r3_ints_O2": "SPECIALIZE -O2 -fno-slp-vectorize", "r3_ints_O3": "SPECIALIZE -O3 -fno-slp-vectorize",
    "r3_all_O2": "SPECIALIZE_ALL -O2 -fno-slp-vectorize", "r3_all_O3": "SPECIALIZE_ALL -O3 -fno-slp-vectorize", "r3_all_w4_O3slp": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -O3",
    "r3_ints_reload": "SPECIALIZE -DPTL_UNIFORM_RELOAD", "r3_dyn_reload": "-DPTL_UNIFORM_RELOAD", "r3_dyn_O1": "-O1", "r3_dyn_O1_reload": "-O1 -DPTL_UNIFORM_RELOAD",
    "r3_all_modinl": "SPECIALIZE_ALL -mllvm -enable-module-inliner", "r3_ints_modinl": "SPECIALIZE -mllvm -enable-module-inliner", "r3_dyn_modinl": "-mllvm -enable-module-inliner",
    "r3_ints_nomasks": "SPECIALIZE NO_MASKS",
    "r3_all_peel": "SPECIALIZE_ALL -DPTL_PEEL_FIRST_TRIP", "r3_ints_peel": "SPECIALIZE -DPTL_PEEL_FIRST_TRIP", "r3_dyn_peel": "-DPTL_PEEL_FIRST_TRIP",
    "r3_all_w4_zerofold": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -fno-signed-zeros -fno-honor-nans",
    # round 4: the NaN guard of 1/x and sqrt as one v_med3_f32 (default) against the compare + select pair of round 3
    "r4_all": "SPECIALIZE_ALL", "r4_all_w4": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4", "r4_all_cmpguard": "SPECIALIZE_ALL -DPTL_CMP_GUARD", "r4_all_w4_cmpguard": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -DPTL_CMP_GUARD",
    # round 4, after the one-sign-test cull: IR-level if-conversion / sinking knobs of the same source
    "r5_base": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4",
    "r5_phi8": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -phi-node-folding-threshold=8",
    "r5_phi32": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -phi-node-folding-threshold=32",
    "r5_two16": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -two-entry-phi-node-folding-threshold=16",
    "r5_two64": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -two-entry-phi-node-folding-threshold=64",
    "r5_phi8_two16": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -phi-node-folding-threshold=8 -mllvm -two-entry-phi-node-folding-threshold=16",
    "r5_gvnsink": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -enable-gvn-sink=true",
    "r5_gvnhoist": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -enable-gvn-hoist=true",
    "r5_nojt": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -jump-threading-threshold=0",
    # LLVM's module inliner (kernel.cpp compile_options): same arithmetic, half the hiprtc time -- what does the kernel cost?
    "r5_mi": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -mllvm -enable-module-inliner", "r5_base_w0": "SPECIALIZE_ALL", "r5_mi_w0": "SPECIALIZE_ALL -mllvm -enable-module-inliner",
    "r5_ints": "SPECIALIZE", "r5_ints_mi": "SPECIALIZE -mllvm -enable-module-inliner", "r5_dyn": "", "r5_dyn_mi": "-mllvm -enable-module-inliner",
    "r5_bw4": "SPECIALIZE_ALL", "r5_bw2": "SPECIALIZE_ALL BLOCK_WAVES=2", "r5_bw1": "SPECIALIZE_ALL BLOCK_WAVES=1",
    "r5_nocull": "SPECIALIZE_ALL -DPTL_WAVES_PER_EU=4 -DPTL_NO_PLANE_CULL",
    "r4_ints": "SPECIALIZE", "r4_ints_cmpguard": "SPECIALIZE -DPTL_CMP_GUARD", "r4_dyn": "", "r4_dyn_cmpguard": "-DPTL_CMP_GUARD",
}
CASES = ["monoportal:1920:1080:20:1", "triple_portal:3840:2160:40:1", "portal_in_portal:3840:2160:40:1", "mobius_monoportal:3840:2160:64:1", "mobius_monoportal:3840:2160:64:4"]


def notes(code):
    path = "/tmp/_variant.hsaco"
    open(path, "wb").write(code)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    g = lambda k: (re.findall(re.escape(k) + r":\s*(\d+)", out) or ["?"])[0]
    return f"v{g('.vgpr_count')} a{g('.agpr_count')} s{g('.sgpr_count')} ss{g('.sgpr_spill_count')} vs{g('.vgpr_spill_count')} scr{g('.private_segment_fixed_size')}"


def run_one(case, vname, flags):
    import portal_amd as pa
    scene_name, w, h, d, aa = case.split(":")
    w, h, d, aa = int(w), int(h), int(d), int(aa)
    toks = flags.split()
    rflags = (pa.FLAG_SPECIALIZE_INTS if ("SPECIALIZE" in toks or "SPECIALIZE_ALL" in toks) else 0) | (pa.FLAG_SPECIALIZE_ALL if "SPECIALIZE_ALL" in toks else 0)
    rflags |= (pa.FLAG_NO_DERIVED_UNIFORMS if "NO_DERIVED" in toks else 0) | (pa.FLAG_FAST_MATH if "FAST" in toks else 0) | (pa.FLAG_NO_DEFERRED_UPDATES if "NO_DEFER" in toks else 0) | (pa.FLAG_NO_UNIFORM_HOIST if "NO_HOIST" in toks else 0) | (pa.FLAG_NO_FIRST_TRIP if "NO_FIRST_TRIP" in toks else 0) | (pa.FLAG_EXACT_CR if "EXACT_CR" in toks else 0) | (pa.FLAG_NO_UNROLL if "NO_UNROLL" in toks else 0) | (pa.FLAG_NO_FIRST_TRIP_PLANES if "NO_FTP" in toks else 0) | (pa.FLAG_NO_ZERO_MASKS if "NO_MASKS" in toks else 0)
    # the VGPR allocator is an option the JIT always passes (kernel.cpp): select it through its own switch, not a second -mllvm
    ra = [t.split("=", 1)[1] for t in toks if t.startswith("-vgpr-regalloc=")]
    if "RA_DEFAULT" in toks:
        ra = ["default"]
    if ra:
        os.environ["PTL_VGPR_REGALLOC"] = ra[-1]
        keep, skip = [], False
        for i, t in enumerate(toks):
            if t == "-mllvm" and i + 1 < len(toks) and toks[i + 1].startswith("-vgpr-regalloc="):
                continue
            if t.startswith("-vgpr-regalloc="):
                continue
            keep.append(t)
        toks = keep
    os.environ["PTL_HIPRTC_FLAGS"] = " ".join(t for t in toks if t.startswith("-"))
    for t in toks:
        if t.startswith("BLOCK_WAVES="):
            os.environ["PTL_BLOCK_WAVES"] = t.split("=")[1]
    # a case may name a scene FILE (tests/corpus/scenes/matryoshka.ron:W:H:depth:aa): its assets are looked up beside its `scenes` directory
    extra = {}
    if scene_name.endswith(".ron"):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), scene_name)
        extra["asset_root"] = os.path.dirname(os.path.dirname(path))
    else:
        path = pa.scene_path(scene_name)
    scene = pa.Scene.from_file(path)
    if os.environ.get("PTL_VARIANTS_PRECOMPILE"):  # no GPU here: fill the code-object cache that travels to the GPU box
        r = pa.SceneRenderer(scene, device=-1, flags=rflags, **extra)
        return {"case": case, "variant": vname, "regs": notes(r.code_object())}
    r = pa.SceneRenderer(scene, device=0, flags=rflags, **extra)
    r.set_option("render_depth", d)
    r.set_option("aa_count", aa)
    times, digest = [], None
    for k in range(6):
        out = r.draw(w, h, rgba8=True)
        times.append(out["ms"])
        digest = hashlib.sha1(out["rgba8"].tobytes()).hexdigest()[:10]
    return {"case": case, "variant": vname, "ms": float(np.median(times[1:])), "min_ms": float(min(times)), "regs": notes(r.code_object()), "sha": digest}


if __name__ == "__main__":
    if sys.argv[1:2] == ["--one"]:  # child: one (case, variant); an unknown -mllvm option makes LLVM exit() the whole process
        print(json.dumps(run_one(sys.argv[2], sys.argv[3], VARIANTS[sys.argv[3]])), flush=True)
        sys.exit(0)
    cases = [a for a in sys.argv[1:] if ":" in a] or CASES
    names = [a for a in sys.argv[1:] if ":" not in a] or list(VARIANTS)
    for case in cases:
        for v in names:
            done = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", case, v], capture_output=True, text=True)
            line = [l for l in done.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else json.dumps({"case": case, "variant": v, "error": (done.stderr or done.stdout)[-300:]}), flush=True)
