#!/bin/bash
# tools/collect_profiles_r03.sh [quick] -- run ON the GPU box (through gpurun): the round-3 measurements DESIGN.md quotes, into gpurun_out/r03/.
# rocprofv3 kernel traces and PMC passes are separate runs (PMC is never combined with other trace domains).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp
python $R/bench.py > $O/bench_pip4k_1gpu.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-second-workload > /dev/null 2>&1
cp /tmp/prof_bench/*kernel_stats.csv $O/kernel_stats_pip4k_bench.csv
BUILD=$(python -c "import json; print(json.load(open('$O/bench_pip4k_1gpu.json'))['config']['build'])")
bash $R/tools/collect_pmc.sh $BUILD > /dev/null 2>&1
mv $R/gpurun_out/pmc_portal_in_portal_3840x2160_d40_spec_*.json $O/ 2> /dev/null
if [ "${1:-}" != "quick" ]; then
  ( python $R/bench.py --workload c2 --no-cpu-baseline
    python $R/bench.py --panini 1.0 --fov 140 --no-cpu-baseline
    python $R/bench.py --workload c3 --no-cpu-baseline
    python $R/bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline ) > $O/bench_other_configs_1gpu.jsonl 2> /dev/null
  for i in 1 2 3; do python $R/bench.py --no-cpu-baseline --no-second-workload 2> /dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['value'], d['config']['build'], d['roofline']['frac'])"; done > $O/bench_pip4k_repeat3.txt
  for b in w0 w3 w4; do [ "$b" != "$BUILD" ] && bash $R/tools/collect_pmc.sh $b > /dev/null 2>&1; done
  BENCH_ARGS="--specialize 0" bash $R/tools/collect_pmc.sh w0 pmc_portal_in_portal_3840x2160_d40_dynamic_w0 > /dev/null 2>&1
  BENCH_ARGS="--workload c5" WORKLOAD="mobius_monoportal 7680x4320 aa 4 depth 64, all scene uniforms baked, build w0, 1 GPU; the 5 timed launches of each pass; FETCH_SIZE / WRITE_SIZE in KB" bash $R/tools/collect_pmc.sh w0 pmc_mobius_monoportal_7680x4320_d64_aa4_spec_w0 > /dev/null 2>&1
  mv $R/gpurun_out/pmc_*.json $O/ 2> /dev/null
  cat $O/bench_pip4k_repeat3.txt
fi
head -3 $O/kernel_stats_pip4k_bench.csv; ls $O
