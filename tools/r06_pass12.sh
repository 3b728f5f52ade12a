#!/bin/bash
# round 6, pass 12: the stagger as the renderer's option (lane_stagger_us): lanes tests, then the records of both bench commands
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k "concurrent_draws or (rehearsal_assembles and gather)" ) > $OUT/pytest_gpu_lanes.log 2>&1
tail -4 $OUT/pytest_gpu_lanes.log
bash tools/r06_final2.sh
