#!/bin/bash
# round 6, pass 6: two frames in flight (renderer option lane_fence 0 + bench.py --lanes): the changed GPU tests, lanes 1 / 2 / 3 on the headline, C2, C3,
# the un-specialised kernel under occupancy hints 5 / 6, then the driver's command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k "concurrent_draws or (rehearsal_assembles and gather)" ) > $OUT/pytest_gpu_lanes.log 2>&1
tail -4 $OUT/pytest_gpu_lanes.log
rm -f $OUT/frames_in_flight.jsonl
for wl in "" "--workload c2" "--workload c3"; do
  for lanes in 1 2 3; do
    timeout 600 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-second-workload --no-segments --lanes $lanes 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
c = d['config']
print(json.dumps({'workload': c['workload'], 'lanes': $lanes, 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['kernel_ms'], 'one_in_flight': c.get('ms_per_step_one_frame_in_flight'), 'identical': c.get('frames_identical_to_one_in_flight'), 'build': c['build']}))" >> $OUT/frames_in_flight.jsonl
  done
done
cat $OUT/frames_in_flight.jsonl
python - > $OUT/unspec_occupancy.jsonl 2>/dev/null <<'PY'
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import portal_amd as pa
for opt in ("", "-O1"):
    for waves in (0, 4, 5, 6):
        if opt:
            os.environ["PTL_JIT_OPT"] = opt
        else:
            os.environ.pop("PTL_JIT_OPT", None)
        r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path("portal_in_portal")), device=0, flags=pa.flag_waves(waves))
        r.set_option("render_depth", 40)
        outs = [r.draw(3840, 2160, rgba8=True) for _ in range(8)]
        res = r.resources()
        print(json.dumps({"build": "un-specialised" + opt, "waves_hint": waves, "ms": round(float(np.median([o["ms"] for o in outs[2:]])), 4), **res}), flush=True)
PY
cat $OUT/unspec_occupancy.jsonl
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command_lanes.json 2> $OUT/bench_driver_command_lanes.err
tail -3 $OUT/bench_driver_command_lanes.err | cut -c1-300
tail -1 $OUT/bench_driver_command_lanes.json | cut -c1-1500
cp gpurun_out/bench_detail.json $OUT/bench_detail_driver_command_lanes.json 2>/dev/null
