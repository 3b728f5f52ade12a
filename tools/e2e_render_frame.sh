#!/bin/bash
# tools/e2e_render_frame.sh -- run ON the GPU box: wall time of `portal-amd render-frame` for the headline frame, split by
# --timing into scene load / kernel build (hiprtc or the code-object cache) / update + draw + download / PNG, for
#   cold   empty code-object cache, comgr's own cache off: what a first frame of a new scene state costs
#   warm   second run of the same command: the code object comes from portal_amd's cache
# with --specialize 0 (every scene uniform read at run time), by default (scene state baked in) and with --fast.
R=${GRAFT_REPO_ROOT:-/root/repo}
export AMD_COMGR_CACHE=0
export PTL_CACHE_DIR=$(mktemp -d)
cd $R
for variant in "--specialize 0" "" "--fast"; do
    for run in cold warm; do
        echo "== render-frame portal_in_portal 3840x2160 depth 40 [$variant] $run"
        portal_amd/portal-amd render-frame scenes/portal_in_portal.ron --width 3840 --height 2160 --render-depth 40 --timing $variant --output /tmp/e2e.png 2>&1 | grep -v '^$'
    done
done
echo "== precompile (no GPU needed), then render-frame with the cache it filled"
export PTL_CACHE_DIR=$(mktemp -d)
portal_amd/portal-amd precompile scenes/portal_in_portal.ron
portal_amd/portal-amd render-frame scenes/portal_in_portal.ron --width 3840 --height 2160 --render-depth 40 --timing --output /tmp/e2e.png
echo "== two ranks on this one GPU (control-flow rehearsal of --gpus N)"
portal_amd/portal-amd render-frame scenes/portal_in_portal.ron --width 3840 --height 2160 --render-depth 40 --devices 0,0 --output /tmp/e2e2.png
portal_amd/portal-amd render-frame scenes/portal_in_portal.ron --width 3840 --height 2160 --render-depth 40 --devices 0,0 --transport copy --output /tmp/e2e3.png
portal_amd/portal-amd render-frame scenes/portal_in_portal.ron --width 3840 --height 2160 --render-depth 40 --devices 0,0 --multi-process --output /tmp/e2e4.png
portal_amd/portal-amd render-frame scenes/portal_in_portal.ron --width 3840 --height 2160 --render-depth 40 --devices 0 --transport rccl --output /tmp/e2e5.png
cmp /tmp/e2e2.png /tmp/e2e3.png && cmp /tmp/e2e2.png /tmp/e2e4.png && cmp /tmp/e2e2.png /tmp/e2e5.png && echo "all four multi-rank PNGs identical"
