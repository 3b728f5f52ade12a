#!/bin/bash
# round 5: PMC class counters of the headline's Bool / Int-baked and un-specialised builds (what round 4 recorded as 202.8 M and 393.0 M VALU instructions)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
pmc() {
    local R=$PWD NAME=$2 i=0
    rm -rf /tmp/pmc_$NAME
    for group in "SQ_WAVES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" "SQ_WAVE_CYCLES SQ_INSTS_BRANCH"; do
        i=$((i + 1))
        ( cd /tmp && timeout 100 rocprofv3 --pmc $group --output-format csv -d /tmp/pmc_$NAME/p$i -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-segments --no-second-workload --build $1 $3 > /tmp/pmc_$NAME.log 2>&1 ) || echo "pass $i ($group) failed or timed out"
    done
    PMC_BENCH_LOG=/tmp/pmc_$NAME.log python tools/pmc_summary.py $OUT/$NAME.json "$4, build $1, 1 GPU; the 5 timed launches of each pass" 5 /tmp/pmc_$NAME/p* | cut -c1-400
}
pmc w4 pmc_portal_in_portal_3840x2160_d40_ints_w4 "--specialize 1" "portal_in_portal 3840x2160 depth 40, Bool / Int uniforms baked (zero / unit patterns, affine rays)"
pmc w4 pmc_portal_in_portal_3840x2160_d40_dyn_w4 "--specialize 0" "portal_in_portal 3840x2160 depth 40, un-specialised"
