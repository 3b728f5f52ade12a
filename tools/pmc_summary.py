#!/usr/bin/env python3
"""tools/pmc_summary.py -- fold rocprofv3 --pmc passes (one directory per pass, CSV output) into the JSON bench.py reads.

usage: tools/pmc_summary.py OUT.json "workload text" TIMED_LAUNCHES DIR [DIR ...]
Only ptl_render_kernel dispatches count; of each pass the LAST `TIMED_LAUNCHES` dispatches (the timed steps; earlier ones are the
variant autotune and warm-up).  Values are summed over the XCD/SE instances rocprofv3 reports per dispatch."""
import collections
import csv
import glob
import json
import os
import sys

if __name__ == "__main__":
    out_path, workload, timed = sys.argv[1], sys.argv[2], int(sys.argv[3])
    counters = {}
    for d in sys.argv[4:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(float))  # counter -> dispatch -> value
            for row in csv.DictReader(open(path)):
                if "ptl_render_kernel" in row["Kernel_Name"]:
                    per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
            for name, by_dispatch in per.items():
                vals = [by_dispatch[k] for k in sorted(by_dispatch)][-timed:]
                counters[name] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
    doc = {"workload": workload, "counters": counters}
    # round 5: which binary was counted.  PMC_BENCH_LOG = the stdout/stderr of the LAST counted bench run: its JSON line names the sha256 of
    # the code object the timed launches ran (config.code_object_sha256); bench.py uses a PMC file only for a run that timed the same binary
    log = os.environ.get("PMC_BENCH_LOG")
    if log and os.path.exists(log):
        for line in open(log, errors="replace"):
            if line.startswith("{") and '"code_object_sha256"' in line:
                try:
                    bench = json.loads(line)
                    doc["code_object_sha256"] = bench["config"]["code_object_sha256"]
                    doc["kernel_source_sha256"] = bench["config"].get("kernel_source_sha256")
                    doc["bench_kernel_ms_under_the_profiler"] = bench.get("kernel_ms")
                    doc["toolchain"] = bench["config"].get("toolchain") or ("hiprtc_version=" + str(bench["config"].get("hiprtc_version")))
                except Exception:
                    pass
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps({k: round(v["mean_per_launch"], 1) for k, v in counters.items()}))
