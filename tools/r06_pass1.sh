#!/bin/bash
# round 6, first GPU pass: the new belt test, the material table A/B on the headline + C3 + C2, the driver's bench command (compact line), the GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1   # page the image in before anything is timed
( time timeout 600 python -m pytest tests/test_affine_guard_fuzz.py -x -q -m gpu ) > $OUT/pytest_gpu_affine_belt.log 2>&1
tail -4 $OUT/pytest_gpu_affine_belt.log
for wl in "" "--workload c3" "--workload c2"; do
  for extra in 0 67108864; do
    echo "== bench $wl extra_flags=$extra" >> $OUT/ab_material_table.jsonl
    timeout 600 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-second-workload --extra-flags $extra 2>/dev/null | tail -1 >> $OUT/ab_material_table.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r06/ab_material_table.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l)
    print("   ", d["ms_per_step"], d["kernel_ms"], d["config"]["build"], d["roofline"].get("valu_insts_per_launch"))
PY
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
tail -c 3000 $OUT/bench_driver_command.json
cp bench_detail.json $OUT/bench_detail_driver_command.json 2>/dev/null
tail -3 $OUT/bench_driver_command.err | cut -c1-300
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu_driver_command.log 2>&1
tail -6 $OUT/pytest_gpu_driver_command.log
