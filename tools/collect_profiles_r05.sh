#!/bin/bash
# tools/collect_profiles_r05.sh -- round 5's measurement pass (run ON the GPU box through gpurun).  Everything is bounded by `timeout`:
# a rocprofv3 pass with FOUR counters in a group sat for 46 minutes on this pool (profiles/r05/README.md) -- two per group is what works.
#   1. the bench line without stored PMC passes (so that one exists whatever happens next),
#   2. PMC class counters of the binaries bench.py times -- the headline's build and the no-hint build of C2, C3, the Panini variant and C5 --
#      ten rocprofv3 passes each, every file with the sha256 of the code object it counted,
#   3. the bench line again (it now finds them: PTL_PMC_DIR), the rocprofv3 kernel trace of the same command, three repeats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
timeout 900 python bench.py > $OUT/bench_pip4k_1gpu_before_pmc.json 2> $OUT/bench_pip4k_1gpu_before_pmc.err
BUILD=$(python -c "import json;print(json.loads(open('$OUT/bench_pip4k_1gpu_before_pmc.json').read().strip().splitlines()[-1])['config']['build'])" 2>/dev/null || echo w4)
echo "headline build: $BUILD"
pmc() {  # build, file name, bench arguments, workload text
    local R=$PWD NAME=$2 i=0
    rm -rf /tmp/pmc_$NAME
    for group in "SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU" "SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
                 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_BRANCH" "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i + 1))
        ( cd /tmp && timeout 120 rocprofv3 --pmc $group --output-format csv -d /tmp/pmc_$NAME/p$i -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-segments --no-second-workload --build $1 $3 > /tmp/pmc_$NAME.log 2>&1 ) || echo "pass $i ($group) failed or timed out"
    done
    PMC_BENCH_LOG=/tmp/pmc_$NAME.log python tools/pmc_summary.py $OUT/$NAME.json "$4, all scene uniforms baked, build $1, 1 GPU; the 5 timed launches of each pass; FETCH_SIZE / WRITE_SIZE in KB" 5 /tmp/pmc_$NAME/p* | cut -c1-400
}
export -f pmc
if [ "${1:-all}" != "bench-only" ]; then
    START=$(date +%s)
    pmc $BUILD pmc_portal_in_portal_3840x2160_d40_spec_$BUILD "" "portal_in_portal 3840x2160 depth 40"
    [ $(( $(date +%s) - START )) -lt 500 ] && pmc w0 pmc_monoportal_1920x1080_d20_spec_w0 "--workload c2" "monoportal 1920x1080 depth 20"
    [ $(( $(date +%s) - START )) -lt 500 ] && pmc w0 pmc_triple_portal_3840x2160_d40_spec_w0 "--workload c3" "triple_portal 3840x2160 depth 40"
    [ $(( $(date +%s) - START )) -lt 500 ] && pmc w0 pmc_portal_in_portal_3840x2160_d40_panini_spec_w0 "--panini 1.0 --fov 140" "portal_in_portal 3840x2160 depth 40 Panini d=1 fov 140"
    [ $(( $(date +%s) - START )) -lt 500 ] && pmc w0 pmc_mobius_monoportal_7680x4320_d64_aa4_spec_w0 "--workload c5" "mobius_monoportal 7680x4320 aa 4 depth 64"
    echo "PMC passes took $(( $(date +%s) - START )) s"
fi
export PTL_PMC_DIR=$PWD/$OUT
timeout 900 python bench.py > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_pip4k_1gpu.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms")}, d["config"]["build"], {k: v["ms"] for k, v in d["config"]["tuning_ms"].items()})
print("roofline", r["frac"], r.get("frac_counted_by_the_oracle"), r.get("hw_arith_frac"), r.get("count_over_hardware"), r.get("pmc_source"), r.get("pmc_unavailable"))
print("other builds:", d.get("kernel_ms_without_jit_specialisation"), d.get("kernel_ms_with_only_int_uniforms_baked"), d.get("kernel_ms_with_only_zero_patterns_and_mode_switches"), d.get("jit_seconds"))
for w in d.get("workloads", []):
    rr = w.get("roofline", {})
    print(w.get("name"), w.get("ms_per_step"), w.get("kernel_ms", w.get("kernel_ms_per_rank")), w.get("trips_per_primary_ray"), (w.get("oracle_check") or {}).get("bit_exact"),
          (w.get("cpu_baseline") or {}).get("value"), "frac", rr.get("frac"), rr.get("frac_counted_by_the_oracle"), rr.get("hw_arith_frac"), rr.get("pmc_unavailable"), w.get("error"))
print(d.get("oracle_check_of_the_timed_build", {}).get("bit_exact"), d.get("reference_text_check_of_the_timed_build", {}).get("bit_exact"))
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_trace -o trace -- python $OLDPWD/bench.py --no-cpu-baseline --no-second-workload --no-segments > /tmp/r05_trace.log 2>&1 )
cp $(find /tmp/r05_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pip4k_bench.csv 2>/dev/null
head -4 $OUT/kernel_stats_pip4k_bench.csv
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-second-workload 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'], d['config']['build'])"; done > $OUT/bench_pip4k_repeat3.txt
cat $OUT/bench_pip4k_repeat3.txt
