#!/bin/bash
# round 5, pass 3: affine rays on the GPU -- the new A/B tests, the bench line (w5 candidate, workloads list), PMC of the timed build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "affine or slices or several_draws" ) > $OUT/pytest_gpu_affine.log 2>&1
tail -5 $OUT/pytest_gpu_affine.log
timeout 1200 python bench.py > $OUT/bench_pip4k_affine.json 2> $OUT/bench_pip4k_affine.err
tail -5 $OUT/bench_pip4k_affine.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_pip4k_affine.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms")}, d["config"]["build"], {k: v["ms"] for k, v in d["config"]["tuning_ms"].items()})
print("other builds:", d.get("kernel_ms_without_jit_specialisation"), d.get("kernel_ms_with_only_int_uniforms_baked"), d.get("kernel_ms_with_only_zero_patterns_and_mode_switches"))
for w in d.get("workloads", []):
    print(w.get("name"), w.get("ms_per_step"), w.get("kernel_ms", w.get("kernel_ms_per_rank")), w.get("trips_per_primary_ray"), (w.get("oracle_check") or {}).get("bit_exact"), (w.get("cpu_baseline") or {}).get("value"), w.get("error"))
print(d.get("oracle_check_of_the_timed_build", {}).get("bit_exact"), d.get("reference_text_check_of_the_timed_build", {}).get("bit_exact"))
PY
BUILD=$(python -c "import json;print(json.loads(open('gpurun_out/r05/bench_pip4k_affine.json').read().strip().splitlines()[-1])['config']['build'])")
bash tools/collect_pmc.sh $BUILD pmc_portal_in_portal_3840x2160_d40_spec_$BUILD > $OUT/pmc_headline.log 2>&1
mv gpurun_out/pmc_portal_in_portal_3840x2160_d40_spec_$BUILD.json $OUT/
tail -2 $OUT/pmc_headline.log
