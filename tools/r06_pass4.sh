#!/bin/bash
# round 6, pass 4: occupancy beyond five waves per SIMD (80 / 72 / 64 VGPRs with 14 / 23 / 54 spilled registers) on the headline, C3 and C2; the new GPU tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
rm -f $OUT/occupancy_beyond_five_waves.jsonl
for wl in "" "--workload c3" "--workload c2" "--workload c5"; do
  for waves in 5 6 7 8; do
    timeout 600 python bench.py $wl --steps 100 --warmup 20 --no-cpu-baseline --no-second-workload --no-segments --waves $waves 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(json.dumps({'workload': d['config']['workload'], 'waves_hint': $waves, 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['kernel_ms']}))" >> $OUT/occupancy_beyond_five_waves.jsonl
  done
done
cat $OUT/occupancy_beyond_five_waves.jsonl
( time timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_affine_guard_fuzz.py -x -q -m gpu -k "material_tables or renderer_that_checks" ) > $OUT/pytest_gpu_new_tests.log 2>&1
tail -4 $OUT/pytest_gpu_new_tests.log
