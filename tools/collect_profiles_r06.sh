#!/bin/bash
# tools/collect_profiles_r06.sh -- round 6's measurement pass (run ON the GPU box through gpurun), everything bounded by `timeout`:
#   1. the driver's own bench command (--gpus 1 --steps 20 --warmup 5) and the default run, before any PMC file of this round's kernels exists;
#   2. PMC class counters of the kernels bench.py times -- the headline's build and the no-hint build of C2, C3, the Panini variant, C5, the deep view and
#      recursive_room -- ten rocprofv3 passes each (two counters per pass at most), every file with the sha256 of the code object AND of the kernel source;
#   3. both bench lines again (they now find the counters: PTL_PMC_DIR), the rocprofv3 kernel trace of both commands, three repeats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
PTL_BENCH_DETAIL=$PWD/$OUT/bench_detail_before_pmc.json timeout 900 python bench.py > $OUT/bench_pip4k_1gpu_before_pmc.json 2> $OUT/bench_pip4k_1gpu_before_pmc.err
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
    PMC_BENCH_LOG=/tmp/pmc_$NAME.log python tools/pmc_summary.py $OUT/$NAME.json "$4, all scene uniforms baked, build $1, 1 GPU; the 5 timed launches of each pass; FETCH_SIZE / WRITE_SIZE in KB" 5 /tmp/pmc_$NAME/p* | cut -c1-300
}
START=$(date +%s)
pmc $BUILD pmc_portal_in_portal_3840x2160_d40_spec_$BUILD "" "portal_in_portal 3840x2160 depth 40"
pmc w0 pmc_monoportal_1920x1080_d20_spec_w0 "--workload c2" "monoportal 1920x1080 depth 20"
pmc w0 pmc_triple_portal_3840x2160_d40_spec_w0 "--workload c3" "triple_portal 3840x2160 depth 40"
pmc w0 pmc_portal_in_portal_3840x2160_d40_panini_spec_w0 "--panini 1.0 --fov 140" "portal_in_portal 3840x2160 depth 40 Panini d=1 fov 140"
pmc w0 pmc_mobius_monoportal_7680x4320_d64_aa4_spec_w0 "--workload c5" "mobius_monoportal 7680x4320 aa 4 depth 64"
pmc w0 pmc_portal_in_portal_3840x2160_d40_cam0_0_0_0.2_1.5_1.6_spec_w0 "--workload c4-deep" "portal_in_portal 3840x2160 depth 40, camera into the nested portals"
pmc w0 pmc_recursive_room_3840x2160_d40_spec_w0 "--workload recursive-room" "tests/corpus/scenes/recursive_room.ron 3840x2160 depth 40 (26 trips per primary ray)"
echo "PMC passes took $(( $(date +%s) - START )) s"
export PTL_PMC_DIR=$PWD/$OUT
( time PTL_BENCH_DETAIL=$PWD/$OUT/bench_detail_driver_command.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
PTL_BENCH_DETAIL=$PWD/$OUT/bench_detail_pip4k_1gpu.json timeout 900 python bench.py > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
python - <<'PY'
import json
for name in ("driver_command", "pip4k_1gpu"):
    line = json.loads(open(f"gpurun_out/r06/bench_{name}.json").read().strip().splitlines()[-1])
    d = json.load(open(f"gpurun_out/r06/bench_detail_{name}.json"))
    r = d["roofline"]
    print(name, "line bytes", len(json.dumps(line, separators=(",", ":"))), {k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms", "steps")}, d["config"]["build"], {k: v["ms"] for k, v in d["config"]["tuning_ms"].items()})
    print("  roofline", r["frac"], r.get("frac_counted_by_the_oracle"), r.get("hw_arith_frac"), r.get("pmc_match"), r.get("pmc_source"), r.get("pmc_unavailable"))
    print("  other builds:", d.get("kernel_ms_without_jit_specialisation"), d.get("kernel_ms_with_only_int_uniforms_baked"), d.get("kernel_ms_with_only_zero_patterns_and_mode_switches"), d.get("jit_seconds"))
    for w in d.get("workloads", []):
        rr = w.get("roofline", {})
        print("  ", w.get("name"), w.get("ms_per_step"), w.get("kernel_ms", w.get("kernel_ms_per_rank")), w.get("trips_per_primary_ray"), (w.get("oracle_check") or {}).get("bit_exact"),
              (w.get("cpu_baseline") or {}).get("value"), "frac", rr.get("frac"), rr.get("hw_arith_frac"), rr.get("pmc_match"), rr.get("pmc_unavailable"), w.get("error"))
    print("  checks:", d.get("oracle_check_of_the_timed_build", {}).get("bit_exact"), d.get("reference_text_check_of_the_timed_build", {}).get("bit_exact"))
PY
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_trace -o trace -- python $R/bench.py --no-cpu-baseline --no-second-workload --no-segments > /tmp/r06_trace.log 2>&1 )
cp $(find /tmp/r06_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pip4k_bench.csv 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_trace20 -o trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload --no-segments > /tmp/r06_trace20.log 2>&1 )
cp $(find /tmp/r06_trace20 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pip4k_bench_driver_command.csv 2>/dev/null
head -4 $OUT/kernel_stats_pip4k_bench.csv $OUT/kernel_stats_pip4k_bench_driver_command.csv
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['steps'], d['ms_per_step'], d['kernel_ms'], d['config']['build'])"; done > $OUT/bench_pip4k_repeat3.txt
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-second-workload 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['steps'], d['ms_per_step'], d['kernel_ms'], d['config']['build'])"; done >> $OUT/bench_pip4k_repeat3.txt
cat $OUT/bench_pip4k_repeat3.txt
