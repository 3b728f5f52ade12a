#!/bin/bash
# round 6, final records of the two bench commands on the final tree (C5 back to one frame at a time)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
export PTL_PMC_DIR=$PWD/profiles/r06
( time PTL_BENCH_DETAIL=$PWD/$OUT/bench_detail_driver_command.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
( time PTL_BENCH_DETAIL=$PWD/$OUT/bench_detail_pip4k_1gpu.json timeout 900 python bench.py ) > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
python - <<'PY'
import json
for name in ("driver_command", "pip4k_1gpu"):
    line = open(f"gpurun_out/r06/bench_{name}.json").read().strip().splitlines()[-1]
    d = json.load(open(f"gpurun_out/r06/bench_detail_{name}.json"))
    print(name, "line bytes", len(line), {k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms", "steps")}, d["config"]["build"], d["config"].get("ms_per_step_one_frame_in_flight"))
    print("  other builds:", d.get("kernel_ms_without_jit_specialisation"), d.get("kernel_ms_with_only_int_uniforms_baked"), d.get("kernel_ms_with_only_zero_patterns_and_mode_switches"), d.get("jit_seconds"), (d.get("several_frames_per_launch") or {}).get("kernel_ms_per_frame"), (d.get("fast_math_mode") or {}).get("kernel_ms"))
    for w in d.get("workloads", []):
        print("  ", w.get("name"), w.get("ms_per_step"), w.get("kernel_ms", w.get("kernel_ms_per_rank")), w.get("ms_per_step_one_frame_in_flight"), w.get("frames_in_flight"), (w.get("oracle_check") or {}).get("bit_exact"), (w.get("cpu_baseline") or {}).get("value"))
PY
grep real $OUT/bench_driver_command.err $OUT/bench_pip4k_1gpu.err
