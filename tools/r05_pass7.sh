#!/bin/bash
# round 5, last pass on the final tree: a random-scene hunt on new seeds, the driver's GPU-suite command, the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 400 python tests/gpu_fuzz_hunt.py 5000 100 60 ) > $OUT/fuzz_hunt.log 2>&1
tail -4 $OUT/fuzz_hunt.log
( time timeout 900 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_pip4k_1gpu.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms")}, d["config"]["build"], {k: v["ms"] for k, v in d["config"]["tuning_ms"].items()})
print("roofline", r["frac"], r.get("frac_counted_by_the_oracle"), r.get("hw_arith_frac"), r.get("pmc_source"), r.get("pmc_unavailable"))
print("fast", d.get("fast_math_mode"), "others", d.get("kernel_ms_without_jit_specialisation"), d.get("kernel_ms_with_only_int_uniforms_baked"), d.get("kernel_ms_with_only_zero_patterns_and_mode_switches"), d.get("jit_seconds"))
for w in d.get("workloads", []):
    rr = w.get("roofline", {})
    print(w.get("name"), w.get("ms_per_step"), w.get("trips_per_primary_ray"), (w.get("oracle_check") or {}).get("bit_exact"), "frac", rr.get("frac"), rr.get("hw_arith_frac"), rr.get("pmc_unavailable"), w.get("error"))
PY
