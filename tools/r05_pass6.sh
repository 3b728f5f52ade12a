#!/bin/bash
# round 5, closing pass: the full GPU suite on the final tree, PMC of the two deep workloads, the bench line with every stored PMC file, the clip pipeline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 900 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
pmc() {
    local R=$PWD NAME=$2 i=0
    rm -rf /tmp/pmc_$NAME
    for group in "SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU" "SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
                 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_BRANCH" "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i + 1))
        ( cd /tmp && timeout 120 rocprofv3 --pmc $group --output-format csv -d /tmp/pmc_$NAME/p$i -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-segments --no-second-workload --build $1 $3 > /tmp/pmc_$NAME.log 2>&1 ) || echo "pass $i ($group) failed or timed out"
    done
    PMC_BENCH_LOG=/tmp/pmc_$NAME.log python tools/pmc_summary.py $OUT/$NAME.json "$4, all scene uniforms baked, build $1, 1 GPU; the 5 timed launches of each pass; FETCH_SIZE / WRITE_SIZE in KB" 5 /tmp/pmc_$NAME/p* | cut -c1-200
}
pmc w0 pmc_portal_in_portal_3840x2160_d40_cam0_0_0_0.2_1.5_1.6_spec_w0 "--workload c4-deep" "portal_in_portal 3840x2160 depth 40, camera into the nested portals"
pmc w0 pmc_recursive_room_3840x2160_d40_spec_w0 "--workload recursive-room" "tests/corpus/scenes/recursive_room.ron 3840x2160 depth 40 (26 trips per primary ray)"
cp $OUT/pmc_portal_in_portal_3840x2160_d40_cam0_0_0_0.2_1.5_1.6_spec_w0.json $OUT/pmc_recursive_room_3840x2160_d40_spec_w0.json profiles/r05/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_pip4k_1gpu.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms")}, d["config"]["build"], {k: v["ms"] for k, v in d["config"]["tuning_ms"].items()})
print("roofline", r["frac"], r.get("frac_counted_by_the_oracle"), r.get("hw_arith_frac"), r.get("pmc_source"), r.get("pmc_unavailable"))
for w in d.get("workloads", []):
    rr = w.get("roofline", {})
    print(w.get("name"), w.get("ms_per_step"), w.get("trips_per_primary_ray"), (w.get("oracle_check") or {}).get("bit_exact"), "frac", rr.get("frac"), rr.get("hw_arith_frac"), rr.get("pmc_unavailable"), w.get("error"))
PY
for spec in 1 0; do
  rm -rf /tmp/vid_$spec
  echo "== portal-amd render portal_in_portal intro.1 --fps 60 --motion-blur-frames 4 --timing --specialize $spec"
  timeout 200 portal_amd/portal-amd render scenes/portal_in_portal.ron intro.1 --fps 60 --motion-blur-frames 4 --timing --specialize $spec --out-dir /tmp/vid_$spec 2>&1 | grep -v '^$' | tail -6
done > $OUT/video_pip_intro1_4k_aa4_blur4.log
( cd /tmp/vid_1 && find . -name '*.png' | sort | xargs md5sum | md5sum ) >> $OUT/video_pip_intro1_4k_aa4_blur4.log
( cd /tmp/vid_0 && find . -name '*.png' | sort | xargs md5sum | md5sum ) >> $OUT/video_pip_intro1_4k_aa4_blur4.log
cat $OUT/video_pip_intro1_4k_aa4_blur4.log | grep -v done
