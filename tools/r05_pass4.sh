#!/bin/bash
# round 5, final pass: the driver's GPU-suite command on the current tree, then the round's profiles (everything under `timeout`)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 900 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
bash tools/collect_profiles_r05.sh
