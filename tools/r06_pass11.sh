#!/bin/bash
# round 6, pass 11: lanes that start together stay together -- the second lane's first launch of a region half a launch later (PTL_BENCH_STAGGER)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
rm -f $OUT/stagger.jsonl
for wl in "" "--workload c2" "--workload c3"; do
  for steps in 20 100 400; do
    for st in 0 0.25 0.5 0.75; do
      PTL_BENCH_STAGGER=$st timeout 600 python bench.py $wl --steps $steps --warmup 5 --no-cpu-baseline --no-second-workload --no-segments --build w5 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print(json.dumps({'workload': c['workload'], 'steps': d['steps'], 'stagger': $st, 'ms_per_step': d['ms_per_step'], 'one_in_flight': c.get('ms_per_step_one_frame_in_flight'), 'kernel_ms': d['kernel_ms']}))" >> $OUT/stagger.jsonl
    done
  done
done
cat $OUT/stagger.jsonl
