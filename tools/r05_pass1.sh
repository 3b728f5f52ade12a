#!/bin/bash
# round 5, first GPU pass: the driver's own GPU-suite command (timed), the bench line, the store-phase micro-benchmark
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1   # page the image in before anything is timed
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu_driver_command.log 2>&1
tail -5 $OUT/pytest_gpu_driver_command.log
timeout 900 python bench.py > $OUT/bench_pip4k_start_of_round.json 2> $OUT/bench_pip4k_start_of_round.err
tail -c 600 $OUT/bench_pip4k_start_of_round.json
( timeout 300 python tools/fb_store_bench.py 3840 2160; timeout 300 python tools/fb_store_bench.py 7680 4320 ) > $OUT/fb_store.jsonl 2>&1
cat $OUT/fb_store.jsonl
