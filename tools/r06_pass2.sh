#!/bin/bash
# round 6, pass 2: baked display switches + teleport_light (the new default) against the two material tables; the 8-rank rehearsals; the driver's bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
rm -f $OUT/ab_material_table2.jsonl
for wl in "" "--workload c3" "--workload c2"; do
  for extra in 0 67108864 134217728; do
    echo "== bench $wl extra_flags=$extra" >> $OUT/ab_material_table2.jsonl
    timeout 600 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-second-workload --extra-flags $extra 2>/dev/null | tail -1 >> $OUT/ab_material_table2.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r06/ab_material_table2.jsonl"):
    if l.startswith("=="): print(l.strip()); continue
    d = json.loads(l)
    t = json.load(open("bench_detail.json"))["config"]["tuning_ms"] if False else None
    print("   ", d["ms_per_step"], d["kernel_ms"], d["config"]["build"])
PY
( time timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k "eight_rank" -s ) > $OUT/pytest_gpu_eight_ranks.log 2>&1
grep -E "passed|failed|bench.py --gpus 8" $OUT/pytest_gpu_eight_ranks.log | cut -c1-400
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
tail -c 1500 $OUT/bench_driver_command.json
cp bench_detail.json $OUT/bench_detail_driver_command.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_detail_driver_command.json"))
print({k: v["ms"] for k, v in d["config"]["tuning_ms"].items()})
PY
