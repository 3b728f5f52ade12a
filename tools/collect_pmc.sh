#!/bin/bash
# tools/collect_pmc.sh BUILD [OUT_NAME] -- run ON the GPU box (through gpurun): rocprofv3 PMC passes of the headline bench for one
# candidate build, ONE counter group per run (never combined with trace domains), folded into gpurun_out/OUT_NAME.json by
# tools/pmc_summary.py.  Extra hiprtc flags come from the environment (PTL_HIPRTC_FLAGS), extra bench arguments from BENCH_ARGS.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
BUILD=${1:-minreg}
NAME=${2:-pmc_portal_in_portal_3840x2160_d40_spec_$BUILD}   # bench.py looks the headline up under profiles/r03/<this name>.json
mkdir -p $O
cd /tmp
rm -rf /tmp/pmc_$NAME
i=0
# PMC_GROUPS="A B;C;D E" replaces the default groups (one rocprofv3 run per ';'-separated group), e.g. the stall anatomy of
# profiles/r02/pmc_stalls_*.json: "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY;SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_BRANCH SQ_IFETCH;SQC_ICACHE_REQ SQC_ICACHE_MISSES;..."
IFS=';' read -r -a GROUPS_ <<< "${PMC_GROUPS:-SQ_WAVES SQ_INSTS_VALU;SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES;SQ_THREAD_CYCLES_VALU;SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32;SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32;SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32;GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY;SQ_WAVE_CYCLES SQ_INSTS_BRANCH;FETCH_SIZE;WRITE_SIZE}"
for group in "${GROUPS_[@]}"; do
    i=$((i + 1))
    rocprofv3 --pmc $group --output-format csv -d /tmp/pmc_$NAME/p$i -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-segments --no-second-workload --build $BUILD ${BENCH_ARGS:-} > /tmp/pmc_$NAME.log 2>&1 || tail -3 /tmp/pmc_$NAME.log
done
PMC_BENCH_LOG=/tmp/pmc_$NAME.log python $R/tools/pmc_summary.py $O/$NAME.json "${WORKLOAD:-portal_in_portal 3840x2160 depth 40, all scene uniforms baked, build $BUILD, flags '${PTL_HIPRTC_FLAGS:-}', 1 GPU; the 5 timed launches of each pass; FETCH_SIZE / WRITE_SIZE in KB}" 5 /tmp/pmc_$NAME/p*
