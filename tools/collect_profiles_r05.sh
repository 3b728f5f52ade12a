#!/bin/bash
# tools/collect_profiles_r05.sh -- round 5's measurement pass (run ON the GPU box through gpurun):
#   PMC class counters of the binaries bench.py times -- the headline's w4 build and the no-hint build of every other workload of the bench line
#   (C2, C3, C5, the Panini variant, the deep view, recursive_room) -- five rocprofv3 passes each, every file with the sha256 of the code object it
#   counted; then the bench line (which now finds them), the rocprofv3 kernel trace of the same command, three repeats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
export PMC_GROUPS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32;SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32;SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES;GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_BRANCH;FETCH_SIZE WRITE_SIZE"
python -c "import torch" >/dev/null 2>&1
pmc() {  # build, file name, bench arguments, workload text
    BENCH_ARGS="$3" WORKLOAD="$4, all scene uniforms baked, build $1, 1 GPU; the 5 timed launches of each pass; FETCH_SIZE / WRITE_SIZE in KB" bash tools/collect_pmc.sh $1 $2 > $OUT/$2.log 2>&1
    mv gpurun_out/$2.json $OUT/ 2>/dev/null
    tail -1 $OUT/$2.log | cut -c1-300
}
if [ "${1:-all}" != "bench-only" ]; then
pmc w4 pmc_portal_in_portal_3840x2160_d40_spec_w4 "" "portal_in_portal 3840x2160 depth 40"
pmc w0 pmc_monoportal_1920x1080_d20_spec_w0 "--workload c2" "monoportal 1920x1080 depth 20"
pmc w0 pmc_triple_portal_3840x2160_d40_spec_w0 "--workload c3" "triple_portal 3840x2160 depth 40"
pmc w0 pmc_portal_in_portal_3840x2160_d40_panini_spec_w0 "--panini 1.0 --fov 140" "portal_in_portal 3840x2160 depth 40 Panini d=1 fov 140"
pmc w0 pmc_portal_in_portal_3840x2160_d40_cam0_0_0_0.2_1.5_1.6_spec_w0 "--workload c4-deep" "portal_in_portal 3840x2160 depth 40, camera into the nested portals"
pmc w0 pmc_recursive_room_3840x2160_d40_spec_w0 "--workload recursive-room" "tests/corpus/scenes/recursive_room.ron 3840x2160 depth 40 (26 trips per primary ray)"
pmc w0 pmc_mobius_monoportal_7680x4320_d64_aa4_spec_w0 "--workload c5" "mobius_monoportal 7680x4320 aa 4 depth 64"
fi
export PTL_PMC_DIR=$PWD/$OUT
timeout 1500 python bench.py > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_pip4k_1gpu.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms")}, d["config"]["build"], {k: v["ms"] for k, v in d["config"]["tuning_ms"].items()})
print("roofline", r["frac"], r.get("frac_counted_by_the_oracle"), r.get("hw_arith_frac"), r.get("count_over_hardware"), r.get("pmc_source"), r.get("pmc_unavailable"))
print("other builds:", d.get("kernel_ms_without_jit_specialisation"), d.get("kernel_ms_with_only_int_uniforms_baked"), d.get("kernel_ms_with_only_zero_patterns_and_mode_switches"))
for w in d.get("workloads", []):
    rr = w.get("roofline", {})
    print(w.get("name"), w.get("ms_per_step"), w.get("kernel_ms", w.get("kernel_ms_per_rank")), w.get("trips_per_primary_ray"), (w.get("oracle_check") or {}).get("bit_exact"),
          (w.get("cpu_baseline") or {}).get("value"), "frac", rr.get("frac"), rr.get("frac_counted_by_the_oracle"), rr.get("hw_arith_frac"), rr.get("pmc_unavailable"), w.get("error"))
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_trace -o trace -- python $OLDPWD/bench.py --no-cpu-baseline --no-second-workload --no-segments > /tmp/r05_trace.log 2>&1
cd $OLDPWD
cp $(find /tmp/r05_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pip4k_bench.csv 2>/dev/null
head -5 $OUT/kernel_stats_pip4k_bench.csv
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-second-workload 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'], d['config']['build'])"; done > $OUT/bench_pip4k_repeat3.txt
cat $OUT/bench_pip4k_repeat3.txt
