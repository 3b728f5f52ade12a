#!/bin/bash
# round 5, pass 2: source-level experiments on the baked headline kernel (each must keep the frame hash)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/stub_profile.py "$@" > $OUT/stub_profile_${TAG:-experiments}.jsonl 2>&1
cat $OUT/stub_profile_${TAG:-experiments}.jsonl
