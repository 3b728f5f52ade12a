#!/bin/bash
# round 6, final pass: the driver's GPU-suite command, both bench commands (the driver's --steps 20 and the default), three repeats of each, and the
# rocprofv3 kernel traces: of the one-stream command (whose per-dispatch average the line's kernel_ms has to agree with) and of the default
# command with two frames in flight (where a dispatch's own duration is not what a frame costs).  The PMC files of profiles/r06 stay: the timed binaries
# are the ones they counted (sha256 in the line and in the files).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch" >/dev/null 2>&1
export PTL_PMC_DIR=$PWD/profiles/r06
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
( time PTL_BENCH_DETAIL=$PWD/$OUT/bench_detail_driver_command.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
( time PTL_BENCH_DETAIL=$PWD/$OUT/bench_detail_pip4k_1gpu.json timeout 900 python bench.py ) > $OUT/bench_pip4k_1gpu.json 2> $OUT/bench_pip4k_1gpu.err
python - <<'PY'
import json
for name in ("driver_command", "pip4k_1gpu"):
    line = open(f"gpurun_out/r06/bench_{name}.json").read().strip().splitlines()[-1]
    d = json.load(open(f"gpurun_out/r06/bench_detail_{name}.json"))
    r = d["roofline"]
    print(name, "line bytes", len(line), {k: d.get(k) for k in ("value", "ms_per_step", "kernel_ms", "steps")}, d["config"]["build"], d["config"].get("ms_per_step_one_frame_in_flight"))
    print("  roofline", r["frac"], r.get("hw_arith_frac"), r.get("pmc_match"), r.get("pmc_unavailable"))
    print("  other builds:", d.get("kernel_ms_without_jit_specialisation"), d.get("kernel_ms_with_only_int_uniforms_baked"), d.get("kernel_ms_with_only_zero_patterns_and_mode_switches"), d.get("jit_seconds"))
    for w in d.get("workloads", []):
        rr = w.get("roofline", {})
        print("  ", w.get("name"), w.get("ms_per_step"), w.get("kernel_ms", w.get("kernel_ms_per_rank")), w.get("ms_per_step_one_frame_in_flight"), w.get("frames_identical_to_one_in_flight"), (w.get("oracle_check") or {}).get("bit_exact"), "frac", rr.get("frac"), rr.get("pmc_match"), rr.get("pmc_unavailable"), w.get("error"))
    print("  checks:", d.get("oracle_check_of_the_timed_build", {}).get("bit_exact"), d.get("reference_text_check_of_the_timed_build", {}).get("bit_exact"))
PY
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_trace1 -o trace -- python $R/bench.py --lanes 1 --no-cpu-baseline --no-second-workload --no-segments > /tmp/r06_trace1.log 2>&1 )
cp $(find /tmp/r06_trace1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pip4k_bench_one_stream.csv 2>/dev/null
tail -1 /tmp/r06_trace1.log > $OUT/bench_pip4k_one_stream_under_rocprof.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_trace2 -o trace -- python $R/bench.py --no-cpu-baseline --no-second-workload --no-segments > /tmp/r06_trace2.log 2>&1 )
cp $(find /tmp/r06_trace2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pip4k_bench.csv 2>/dev/null
tail -1 /tmp/r06_trace2.log > $OUT/bench_pip4k_under_rocprof.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_trace20 -o trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload --no-segments > /tmp/r06_trace20.log 2>&1 )
cp $(find /tmp/r06_trace20 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pip4k_bench_driver_command.csv 2>/dev/null
head -3 $OUT/kernel_stats_pip4k_bench_one_stream.csv $OUT/kernel_stats_pip4k_bench.csv $OUT/kernel_stats_pip4k_bench_driver_command.csv | cut -c1-160
rm -f $OUT/bench_pip4k_repeat3.txt
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-workload 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['steps'], d['ms_per_step'], d['kernel_ms'], d['config'].get('ms_per_step_one_frame_in_flight'), d['config']['build'])" >> $OUT/bench_pip4k_repeat3.txt; done
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-second-workload 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['steps'], d['ms_per_step'], d['kernel_ms'], d['config'].get('ms_per_step_one_frame_in_flight'), d['config']['build'])" >> $OUT/bench_pip4k_repeat3.txt; done
cat $OUT/bench_pip4k_repeat3.txt
timeout 300 python tools/two_streams.py > $OUT/two_streams.jsonl 2>/dev/null; cat $OUT/two_streams.jsonl | cut -c1-300
