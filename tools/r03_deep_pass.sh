#!/bin/bash
# round 3: the deep-recursion regime -- bench lines, lane utilisation from PMC -- and the zero-folding upper bound
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
for w in c4-deep c4-deep2 plus-ultra; do python bench.py --workload $w --no-second-workload 2>> $O/deep.err; done > $O/bench_deep_views_1gpu.jsonl
for w in c4-deep c4-deep2 plus-ultra; do
  PMC_GROUPS="SQ_WAVES SQ_INSTS_VALU;SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES;SQ_THREAD_CYCLES_VALU;SQ_INSTS_SALU SQ_INSTS_BRANCH;SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" BENCH_ARGS="--workload $w" \
    WORKLOAD="bench.py --workload $w, all scene uniforms baked, build w4, 1 GPU; the 5 timed launches of each pass" bash tools/collect_pmc.sh w4 pmc_lanes_$w > /dev/null 2>&1
done
mv gpurun_out/pmc_lanes_*.json $O/ 2> /dev/null
python tools/variants.py portal_in_portal:3840:2160:40:1 triple_portal:3840:2160:40:1 r3_all_w4 r3_all_w4_zerofold r3_fast_all_w4 > $O/variants2_zerofold_upper_bound.jsonl 2>> $O/deep.err
python - <<'PY'
import json, glob
for l in open("gpurun_out/r03/bench_deep_views_1gpu.jsonl"):
    d = json.loads(l); print(d["config"]["workload"], "ms", d["ms_per_step"], "kernel", d["kernel_ms"], "trips/ray", d["config"]["trips_per_primary_ray"], "Mray/s", d["value"], "oracle", d.get("oracle_check_of_the_timed_build", {}).get("bit_exact"))
for f in sorted(glob.glob("gpurun_out/r03/pmc_lanes_*.json")):
    c = json.load(open(f))["counters"]; print(f, "lane utilisation", c["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / 64 / c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"], "VALU/wave", c["SQ_INSTS_VALU"]["mean_per_launch"] / c["SQ_WAVES"]["mean_per_launch"])
PY
cat $O/variants2_zerofold_upper_bound.jsonl
