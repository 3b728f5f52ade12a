#!/bin/bash
# round 6, pass 8: the un-specialised headline kernel is 17 090 instructions (~130 KB; the instruction cache is 64 KB per two CUs): what do the builds
# that shrink it cost or buy -- no first-trip copies (snippets / plane tests), the Simple materials through a table (LDS / scalar waterfall)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
V="plain=w4 plain0=0 ft=NO_FIRST_TRIP|w4 ftp=NO_FIRST_TRIP_PLANES|w4 both=NO_FIRST_TRIP|NO_FIRST_TRIP_PLANES|w4 both0=NO_FIRST_TRIP|NO_FIRST_TRIP_PLANES"
V="$V mtl=MATERIAL_TABLE_LDS|w4 mts=MATERIAL_TABLE_SCALAR|w4 both_mts=NO_FIRST_TRIP|NO_FIRST_TRIP_PLANES|MATERIAL_TABLE_SCALAR|w4 both_mtl=NO_FIRST_TRIP|NO_FIRST_TRIP_PLANES|MATERIAL_TABLE_LDS|w4"
V="$V both_mts0=NO_FIRST_TRIP|NO_FIRST_TRIP_PLANES|MATERIAL_TABLE_SCALAR both_mts5=NO_FIRST_TRIP|NO_FIRST_TRIP_PLANES|MATERIAL_TABLE_SCALAR|w5 ftp_mts=NO_FIRST_TRIP_PLANES|MATERIAL_TABLE_SCALAR|w4 ft_mts=NO_FIRST_TRIP|MATERIAL_TABLE_SCALAR|w4"
timeout 1500 python tools/ab_views.py --scene portal_in_portal $V > $OUT/ab_unspec_code_size.jsonl 2>/dev/null
V="pat=SPECIALIZE_PATTERNS pat_mts=SPECIALIZE_PATTERNS|MATERIAL_TABLE_SCALAR pat_mtl=SPECIALIZE_PATTERNS|MATERIAL_TABLE_LDS ints=SPECIALIZE_INTS ints_mts=SPECIALIZE_INTS|MATERIAL_TABLE_SCALAR ints_mtl=SPECIALIZE_INTS|MATERIAL_TABLE_LDS"
timeout 900 python tools/ab_views.py --scene portal_in_portal $V >> $OUT/ab_unspec_code_size.jsonl 2>/dev/null
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r06/ab_unspec_code_size.jsonl") if l.startswith("{")]
for r in rows:
    if r["view"] in ("default", "deep"):
        print(r["variant"], r["view"], r["ms"], r["sha"])
PY
for s in triple_portal monoportal mobius_monoportal; do
  timeout 600 python tools/ab_views.py --scene $s plain=0 both=NO_FIRST_TRIP\|NO_FIRST_TRIP_PLANES both_mts=NO_FIRST_TRIP\|NO_FIRST_TRIP_PLANES\|MATERIAL_TABLE_SCALAR mts=MATERIAL_TABLE_SCALAR 2>/dev/null | grep '"default"' >> $OUT/ab_unspec_code_size_other_scenes.jsonl
done
cat $OUT/ab_unspec_code_size_other_scenes.jsonl
