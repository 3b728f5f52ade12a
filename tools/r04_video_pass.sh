#!/bin/bash
# round 4: the video pipeline and the single-frame CLI with the round-4 kernels (same commands as profiles/r02/video_pip_intro1_4k_aa4_blur4.log, render_frame_e2e.log)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
for spec in 1 0; do
  rm -rf /tmp/vid_$spec
  echo "== portal-amd render portal_in_portal intro.1 --fps 60 --motion-blur-frames 4 --timing --specialize $spec"
  portal_amd/portal-amd render scenes/portal_in_portal.ron intro.1 --fps 60 --motion-blur-frames 4 --timing --specialize $spec --out-dir /tmp/vid_$spec 2>&1 | grep -v '^$' | tail -6
done > $O/video_pip_intro1_4k_aa4_blur4.log
( cd /tmp/vid_1 && find . -name '*.png' | sort | xargs md5sum | md5sum ) >> $O/video_pip_intro1_4k_aa4_blur4.log
( cd /tmp/vid_0 && find . -name '*.png' | sort | xargs md5sum | md5sum ) >> $O/video_pip_intro1_4k_aa4_blur4.log
# the same clip with one launch per sub-frame (round 3's form), and at 1080p both ways: what the one-launch form buys where frames are small
for batch in 0; do
  rm -rf /tmp/vid_nb
  echo "== ... --specialize 1 --batch-subframes 0"
  portal_amd/portal-amd render scenes/portal_in_portal.ron intro.1 --fps 60 --motion-blur-frames 4 --timing --specialize 1 --batch-subframes 0 --out-dir /tmp/vid_nb 2>&1 | grep -v '^$' | tail -6
done >> $O/video_pip_intro1_4k_aa4_blur4.log
( cd /tmp/vid_nb && find . -name '*.png' | sort | xargs md5sum | md5sum ) >> $O/video_pip_intro1_4k_aa4_blur4.log
for batch in 1 0; do
  rm -rf /tmp/vid_hd_$batch
  echo "== portal-amd render portal_in_portal intro.1 --width 1920 --height 1080 --aa-count 1 --fps 60 --motion-blur-frames 8 --timing --specialize 1 --batch-subframes $batch"
  portal_amd/portal-amd render scenes/portal_in_portal.ron intro.1 --width 1920 --height 1080 --aa-count 1 --fps 60 --motion-blur-frames 8 --timing --specialize 1 --batch-subframes $batch --out-dir /tmp/vid_hd_$batch 2>&1 | grep -v '^$' | tail -6
done > $O/video_pip_intro1_1080p_aa1_blur8.log
( cd /tmp/vid_hd_1 && find . -name '*.png' | sort | xargs md5sum | md5sum ) >> $O/video_pip_intro1_1080p_aa1_blur8.log
( cd /tmp/vid_hd_0 && find . -name '*.png' | sort | xargs md5sum | md5sum ) >> $O/video_pip_intro1_1080p_aa1_blur8.log
bash tools/e2e_render_frame.sh > $O/render_frame_e2e.log 2>&1
cat $O/video_pip_intro1_4k_aa4_blur4.log $O/video_pip_intro1_1080p_aa1_blur8.log | grep -v done; grep -E "==|kernel|total|hiprtc|wall" $O/render_frame_e2e.log | head -40
