#!/bin/bash
# tools/r02_gpu_pass.sh [steps...] -- run ON the GPU box through gpurun; writes everything under gpurun_out/r02/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
STEPS=${*:-"tests variants fast e2e bench"}
for step in $STEPS; do
case $step in
tests)    timeout 1500 python -m pytest tests -q -m gpu -x --durations=10 2>&1 | tail -40 > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log ;;
variants) timeout 900 python tools/variants.py portal_in_portal:3840:2160:40:1 triple_portal:3840:2160:40:1 monoportal:1920:1080:20:1 mobius_monoportal:7680:4320:64:4 \
              r2_dyn r2_dyn_nocull r2_ints r2_ints_nocull r2_all r2_all_nocull r2_all_minreg r2_all_minreg_nocull r2_fast_all r2_fast_all_nocull > $O/variants2_plane_cull.jsonl 2>&1; cat $O/variants2_plane_cull.jsonl ;;
fast)     timeout 600 python -X faulthandler tools/fast_mode_report.py > $O/fast_mode.jsonl 2>&1; cat $O/fast_mode.jsonl ;;
e2e)      timeout 600 bash tools/e2e_render_frame.sh > $O/render_frame_e2e.log 2>&1; cat $O/render_frame_e2e.log ;;
bench)    timeout 900 python bench.py > $O/bench_pip4k_1gpu.json 2> $O/bench_pip4k_1gpu.err; cat $O/bench_pip4k_1gpu.json; tail -3 $O/bench_pip4k_1gpu.err ;;
esac
done
