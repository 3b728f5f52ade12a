#!/usr/bin/env python3
"""tools/pc_sample.py -- where the wave-cycles of a generated kernel go, by PC sampling (rocprofv3 --pc-sampling-beta-enabled).

    python tools/pc_sample.py run  [--scene portal_in_portal] [--flags 5] [--waves 4] [--frames 1500] [--size 3840x2160] [--depth 40]
                                   [--method host_trap|stochastic] [--interval 1000] [--out gpurun_out/pc]      (on the GPU box)
    python tools/pc_sample.py draw ...                    (the child rocprofv3 runs: draws the frames, nothing else)
    python tools/pc_sample.py fold DIR                    (aggregate rocprofv3's sample files in DIR -> DIR/pc_samples.json)

The kernel is built with -gline-tables-only (same code, PTL_HIPRTC_FLAGS), so rocprofv3 can print the source line beside every
sampled instruction; `fold` reduces the (large) sample table to counts per instruction / mnemonic class / source line / function.
Development aid; the summaries worth keeping are copied to profiles/.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)


def args_of(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["run", "draw", "fold"])
    ap.add_argument("dir", nargs="?", default="")
    ap.add_argument("--scene", default="portal_in_portal")
    ap.add_argument("--flags", type=int, default=5)
    ap.add_argument("--waves", type=int, default=4)
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--depth", type=int, default=40)
    ap.add_argument("--aa", type=int, default=1)
    ap.add_argument("--method", default="host_trap")
    ap.add_argument("--unit", default="time")
    ap.add_argument("--interval", type=int, default=1000)
    ap.add_argument("--out", default="gpurun_out/pc")
    return ap.parse_args(argv)


def draw(a):
    import portal_amd as pa

    w, h = (int(x) for x in a.size.split("x"))
    r = pa.SceneRenderer(pa.Scene.from_file(pa.scene_path(a.scene)), device=0, flags=a.flags | pa.flag_waves(a.waves))
    r.set_option("render_depth", a.depth)
    r.set_option("aa_count", a.aa)
    src = r.kernel_source()
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "kernel_source.hip"), "w") as f:
        f.write(src)
    buf = pa.device_alloc(w * h * 4, 0)
    frame = pa.Frame(w, h, 0, 1, 0)
    for k in range(a.frames):
        ms = r.draw_device(frame, out_rgba8=buf, timed=(k % 64 == 63 or k == a.frames - 1))  # a timed draw waits: at most 64 launches queued
    print("drew", a.frames, "frames; last", ms, "ms")


def fold(d):
    files = [f for f in glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True)]
    if not files:
        raise SystemExit(f"no pc sampling csv under {d}: {os.listdir(d)}")
    from tools.isa_hist import classify

    per_inst, per_line, per_class = collections.Counter(), collections.Counter(), collections.Counter()
    lanes = collections.Counter()
    total = 0
    head = []
    for path in files:
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            cols = rd.fieldnames or []
            ic = next((c for c in cols if c.lower() == "instruction"), None)
            cc = next((c for c in cols if "comment" in c.lower()), None)
            ec = next((c for c in cols if "exec" in c.lower()), None)
            for row in rd:
                if len(head) < 40:
                    head.append(row)
                inst = (row.get(ic) or "").strip()
                if not inst:
                    continue
                total += 1
                comment = (row.get(cc) or "").strip() if cc else ""
                per_inst[(inst, comment)] += 1
                m = inst.split()[0]
                per_class[classify(m)] += 1
                line = re.search(r":(\d+)\s*$", comment)
                per_line[int(line.group(1)) if line else -1] += 1
                if ec and row.get(ec):
                    try:
                        lanes[m] += bin(int(row[ec], 0) if row[ec].startswith("0x") else int(row[ec])).count("1")
                    except ValueError:
                        pass
    out = {"files": files, "columns": cols, "samples": total, "classes": dict(per_class.most_common()),
           "lines": {str(k): v for k, v in per_line.most_common(400)},
           "instructions": [{"inst": k[0], "where": k[1], "n": v} for k, v in per_inst.most_common(1500)], "head": head}
    with open(os.path.join(d, "pc_samples.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(json.dumps({"samples": total, "classes": out["classes"]}, indent=1))


def run(a):
    os.makedirs(a.out, exist_ok=True)
    env = dict(os.environ, PTL_HIPRTC_FLAGS="-gline-tables-only", ROCPROFILER_PC_SAMPLING_BETA_ENABLED="1", TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pc-sampling-beta-enabled", "--pc-sampling-method", a.method, "--pc-sampling-unit", a.unit, "--pc-sampling-interval", str(a.interval),
           "--output-format", "csv", "-d", os.path.abspath(os.path.join(a.out, "raw")), "--", sys.executable, os.path.abspath(__file__), "draw",
           "--scene", a.scene, "--flags", str(a.flags), "--waves", str(a.waves), "--frames", str(a.frames), "--size", a.size, "--depth", str(a.depth), "--aa", str(a.aa),
           "--out", os.path.abspath(a.out)]
    print(" ".join(cmd), flush=True)
    done = subprocess.run(cmd, env=env, cwd="/tmp", timeout=600)
    print("rocprofv3 rc", done.returncode, flush=True)
    fold(os.path.join(a.out, "raw"))
    subprocess.run(["bash", "-c", f"mv {a.out}/raw/pc_samples.json {a.out}/; du -sh {a.out}/raw; rm -rf {a.out}/raw"])


if __name__ == "__main__":
    a = args_of(sys.argv[1:])
    {"run": run, "draw": draw, "fold": lambda a: fold(a.dir)}[a.cmd](a)
