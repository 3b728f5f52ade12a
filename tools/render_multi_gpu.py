#!/usr/bin/env python3
"""tools/render_multi_gpu.py -- the video pipeline on every GPU of the node.

Frames of a clip are independent, so there is no collective: process k of N runs `portal-amd render ... --device k --shard k/N`
and traces frames k, k+N, k+2N ... into the shared anim/ directory; when all are done one more invocation (no shard) finds every
frame already there, so it only hands the sequence to ffmpeg (or parks it under video/<scene>/<clip>.frames).

usage: tools/render_multi_gpu.py [--gpus N] -- <arguments of `portal-amd render`>
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "portal_amd", "portal-amd")

if __name__ == "__main__":
    argv = sys.argv[1:]
    gpus = None
    if argv[:1] == ["--gpus"]:
        gpus, argv = int(argv[1]), argv[2:]
    if argv[:1] == ["--"]:
        argv = argv[1:]
    if gpus is None:
        out = subprocess.run([EXE, "version"], capture_output=True, text=True).stdout
        gpus = max(1, int(out.split("devices:")[1].split()[0])) if "devices:" in out else 1
    procs = [subprocess.Popen([EXE, "render", *argv, "--device", str(k), "--shard", f"{k}/{gpus}"]) for k in range(gpus)]
    codes = [p.wait() for p in procs]
    if any(codes):
        sys.exit(f"shard exit codes: {codes}")
    sys.exit(subprocess.run([EXE, "render", *argv]).returncode)
