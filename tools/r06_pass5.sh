#!/bin/bash
# round 6, pass 5: the driver's GPU-suite command on the current tree, then the round's profiles
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
bash tools/collect_profiles_r06.sh
