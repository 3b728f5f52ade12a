#!/bin/bash
# round 6, pass 10: lane streams shared by the renderers of a process (the other workloads of the bench line showed no gain from two frames in flight: their
# renderers' own stream pairs landed on hardware queues already in use); lanes tests; the default bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "concurrent_draws or first_trip" ) > $OUT/pytest_gpu_lanes.log 2>&1
tail -4 $OUT/pytest_gpu_lanes.log
( time python bench.py ) > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
cp gpurun_out/bench_detail.json $OUT/bench_detail_pip4k_1gpu.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_detail_pip4k_1gpu.json"))
print(d["steps"], d["ms_per_step"], d["kernel_ms"], d["config"].get("ms_per_step_one_frame_in_flight"), d["config"]["build"])
for w in d.get("workloads", []):
    print(" ", w.get("name"), w.get("ms_per_step"), w.get("kernel_ms", w.get("kernel_ms_per_rank")), w.get("ms_per_step_one_frame_in_flight"), w.get("frames_identical_to_one_in_flight"))
PY
