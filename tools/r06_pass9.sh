#!/bin/bash
# round 6, pass 9: first-trip plane tests in the specialised builds that keep run-time matrices (patterns, Int-baked) -- rounds 3-5 had them by default,
# since round 6 they are opt-in: A/B through the hook; then the driver's GPU-suite command and both bench commands on the new default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
rm -f $OUT/ab_first_trip_planes.jsonl
for s in portal_in_portal triple_portal monoportal; do
  PTL_AB_FIRST_TRIP_PLANES=1 timeout 600 python tools/ab_views.py --scene $s pat_ftp=SPECIALIZE_PATTERNS ints_ftp=SPECIALIZE_INTS 2>/dev/null >> $OUT/ab_first_trip_planes.jsonl
  timeout 600 python tools/ab_views.py --scene $s pat=SPECIALIZE_PATTERNS ints=SPECIALIZE_INTS 2>/dev/null >> $OUT/ab_first_trip_planes.jsonl
done
python - <<'PY'
import json, collections
t = collections.defaultdict(dict)
for l in open("gpurun_out/r06/ab_first_trip_planes.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); t[(r["scene"], r["variant"])][r["view"]] = r["ms"]
for k, v in t.items(): print(k, v)
PY
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
tail -1 $OUT/bench_driver_command.json | cut -c1-600
cp gpurun_out/bench_detail.json $OUT/bench_detail_driver_command.json 2>/dev/null
( time python bench.py ) > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
tail -1 $OUT/bench_pip4k_1gpu.json | cut -c1-600
cp gpurun_out/bench_detail.json $OUT/bench_detail_pip4k_1gpu.json 2>/dev/null
