#!/bin/bash
# round 6, pass 7: the store pattern with two launches in flight (N-star: the framebuffer write at 4K), the driver's command twice and the default bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
make kernels >/dev/null 2>&1
timeout 600 python tools/fb_store_bench.py 3840 2160 > $OUT/fb_store_4k_in_flight.jsonl 2>/dev/null
timeout 600 python tools/fb_store_bench.py 7680 4320 > $OUT/fb_store_8k_in_flight.jsonl 2>/dev/null
cut -c1-700 $OUT/fb_store_4k_in_flight.jsonl
for i in 1 2; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload --no-segments 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
print(json.dumps({'steps': d['steps'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['kernel_ms'], 'one_in_flight': c.get('ms_per_step_one_frame_in_flight'), 'build': c['build']}))" >> $OUT/bench_short_region.jsonl
done
cat $OUT/bench_short_region.jsonl
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command_lanes.json 2> $OUT/bench_driver_command_lanes.err
tail -1 $OUT/bench_driver_command_lanes.json | cut -c1-900
cp gpurun_out/bench_detail.json $OUT/bench_detail_driver_command_lanes.json 2>/dev/null
