#!/bin/bash
# tools/collect_profiles.sh -- run ON the GPU box (through gpurun): the measurements DESIGN.md quotes, into gpurun_out/.
# rocprofv3 kernel traces only (PMC passes are separate runs: never combined with other trace domains).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
python $R/bench.py > $O/bench_pip4k_1gpu.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/prof_bench/*kernel_stats.csv $O/kernel_stats_pip4k_bench.csv
( python $R/bench.py --scene monoportal --width 1920 --height 1080 --depth 20 --no-cpu-baseline
  python $R/bench.py --panini 1.0 --fov 140 --no-cpu-baseline
  python $R/bench.py --scene triple_portal --width 3840 --height 2160 --depth 40 --no-cpu-baseline
  python $R/bench.py --scene mobius_monoportal --width 7680 --height 4320 --depth 64 --aa 4 --steps 5 --warmup 1 --no-cpu-baseline ) > $O/bench_other_configs_1gpu.jsonl 2> /dev/null
python $R/tools/average_bench.py > $O/average_images.jsonl 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_avg -o a -- python $R/tools/average_bench.py > /dev/null 2>&1
cp /tmp/prof_avg/*kernel_stats.csv $O/kernel_stats_average_images.csv
for s in portal_in_portal triple_portal; do python $R/tools/stub_profile.py $s 3840 2160 40 2> /dev/null; done > $O/stub_profile.jsonl
python $R/tools/stub_profile.py mobius_monoportal 3840 2160 64 2> /dev/null >> $O/stub_profile.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vid -o v -- $R/portal_amd/portal-amd render $R/scenes/portal_in_portal.ron intro.1 --fps 60 --motion-blur-frames 4 --out-dir /tmp/vid_prof > $O/video_prof.log 2>&1
cp /tmp/prof_vid/*kernel_stats.csv $O/kernel_stats_video_pip_intro1_4k_aa4_blur4_clip_specialised.csv
head -3 $O/kernel_stats_pip4k_bench.csv; cat $O/bench_pip4k_1gpu.json | cut -c1-300
python $R/tools/valu_rates.py > $O/valu_rates.jsonl 2> /dev/null
for b in w0 minreg; do bash $R/tools/collect_pmc.sh $b > /dev/null 2>&1; done
