#!/bin/bash
# round 6, pass 3: what separates the un-specialised kernel from the patterns build; where C5's time goes (stub profile of the Moebius search)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
timeout 900 python tools/unspec_anatomy.py > $OUT/unspec_anatomy.jsonl 2>&1
cat $OUT/unspec_anatomy.jsonl | cut -c1-260
timeout 900 python tools/stub_profile.py mobius_monoportal 7680 4320 64 --intact --mob_no_search --mob_one_seed --mob_two_seeds --mob_four_seeds --mob_no_refinement_seeds --mob_no_newton --mob_newton_3 --mob_free_trig --mob_no_derivative_probe --no_planes --no_material > $OUT/stub_profile_c5.jsonl 2>&1
cat $OUT/stub_profile_c5.jsonl
