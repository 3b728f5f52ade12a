#!/bin/bash
# tools/collect_profiles_r04.sh [quick] -- run ON the GPU box (through gpurun): the round-4 measurements DESIGN.md / profiles/r04/README.md quote,
# into gpurun_out/r04/.  rocprofv3 kernel traces and PMC passes are separate runs (PMC is never combined with other trace domains).
# Round 4 stores PMC class counters for EVERY workload whose roofline bench.py prints: the headline, C2, C3, C5 and the
# Panini variant -- `frac` is capped at the hardware's instruction ceiling wherever such a file exists.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp
pmc() {  # pmc BUILD NAME "WORKLOAD TEXT" [bench args...]
  local build=$1 name=$2 text=$3; shift 3
  BENCH_ARGS="$*" WORKLOAD="$text, all scene uniforms baked, build $build, 1 GPU; the 5 timed launches of each pass; FETCH_SIZE / WRITE_SIZE in KB" \
    bash $R/tools/collect_pmc.sh $build $name > /dev/null 2>&1
  mv $R/gpurun_out/$name.json $O/ 2> /dev/null
}
# PMC first: bench.py reads the stored files (profiles/r04 after they are copied there; on this box: PTL_PMC_DIR)
BUILD=${BUILD:-w4}
pmc $BUILD pmc_portal_in_portal_3840x2160_d40_spec_$BUILD "portal_in_portal 3840x2160 depth 40"
if [ "${1:-}" != "quick" ]; then
  pmc w0 pmc_monoportal_1920x1080_d20_spec_w0 "monoportal 1920x1080 depth 20" --workload c2
  pmc w0 pmc_triple_portal_3840x2160_d40_spec_w0 "triple_portal 3840x2160 depth 40" --workload c3
  pmc w0 pmc_mobius_monoportal_7680x4320_d64_aa4_spec_w0 "mobius_monoportal 7680x4320 aa 4 depth 64" --workload c5
  pmc w0 pmc_portal_in_portal_3840x2160_d40_panini_spec_w0 "portal_in_portal 3840x2160 depth 40, Panini d = 1 fov 140" --panini 1.0 --fov 140
fi
export PTL_PMC_DIR=$O
python $R/bench.py > $O/bench_pip4k_1gpu.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-second-workload > /dev/null 2>&1
cp /tmp/prof_bench/*kernel_stats.csv $O/kernel_stats_pip4k_bench.csv
if [ "${1:-}" != "quick" ]; then
  ( python $R/bench.py --workload c2 --no-cpu-baseline
    python $R/bench.py --panini 1.0 --fov 140 --no-cpu-baseline --no-second-workload
    python $R/bench.py --workload c3 --no-cpu-baseline
    python $R/bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline ) > $O/bench_other_configs_1gpu.jsonl 2> /dev/null
  for i in 1 2 3; do python $R/bench.py --no-cpu-baseline --no-second-workload 2> /dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['value'], d['config']['build'], d['roofline']['frac'])"; done > $O/bench_pip4k_repeat3.txt
  cat $O/bench_pip4k_repeat3.txt
fi
head -3 $O/kernel_stats_pip4k_bench.csv; ls $O
