#!/usr/bin/env python3
"""tools/jit_inliner_survey.py -- hiprtc time and code-object resources of every scene's kernel under LLVM's bottom-up inliner
pipeline (the toolchain's default) and under its module inliner (`-mllvm -enable-module-inliner`).  No GPU needed.

    python tools/jit_inliner_survey.py [--flags 13] [--jobs 7] [--out profiles/r04/jit_inliner_survey.jsonl] [scene.ron ...]

Per scene and mode: cold compile seconds (code-object cache and comgr's own cache off), VGPRs, scratch bytes, instruction count.
Scratch must stay 0 under the module inliner: a function that is not force-inlined keeps the tracer object in memory (`this`)."""
import argparse, glob, json, os, re, subprocess, sys, tempfile, time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def one(path, flags, mode):
    os.environ["PTL_CACHE_DIR"] = tempfile.mkdtemp()
    os.environ["AMD_COMGR_CACHE"] = "0"
    os.environ["PTL_MODULE_INLINER"] = "1" if mode == "module" else "0"
    sys.path.insert(0, HERE)
    import portal_amd as pa
    extra = {}
    if "/corpus/" in path:
        extra["asset_root"] = os.path.dirname(os.path.dirname(path))
    scene = pa.Scene.from_file(path)
    t = time.time()
    r = pa.SceneRenderer(scene, device=-1, flags=flags, **extra)
    dt = time.time() - t
    code = r.code_object()
    source = r.kernel_source()
    body = source[source.find("// --- scene library snippets"):source.find("struct ExternalRayTeleportation")] if "// --- scene library snippets" in source else source
    f = tempfile.mktemp(suffix=".hsaco")
    open(f, "wb").write(code)
    notes = subprocess.run([READELF, "--notes", f], capture_output=True, text=True).stdout
    k = notes[notes.find("ptl_render"):] if "ptl_render" in notes else notes
    # the render kernel's record: fields around its .name
    recs = re.split(r"\n\s*- ", notes)
    rec = next((x for x in recs if re.search(r"\.name:\s+ptl_render", x)), notes)
    g = lambda key: int((re.findall(re.escape(key) + r":\s*(\d+)", rec) or ["-1"])[0])
    dis = subprocess.run([OBJDUMP, "-d", f], capture_output=True, text=True).stdout
    os.unlink(f)
    return {"scene": os.path.relpath(path, HERE), "flags": flags, "inliner": mode, "jit_s": round(dt, 2), "vgprs": g(".vgpr_count"), "scratch": g(".private_segment_fixed_size"),
            "vgpr_spills": g(".vgpr_spill_count"), "instructions": len(re.findall(r"^\s+[vs]_|^\s+(?:scratch|global|flat|buffer|ds)_", dis, re.M)), "calls": len(re.findall(r"s_swappc_b64", dis)), "scene_source_bytes": len(body)}


if __name__ == "__main__":
    if sys.argv[1:2] == ["--one"]:
        print(json.dumps(one(sys.argv[2], int(sys.argv[3]), sys.argv[4])), flush=True)
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, nargs="*", default=[13, 0])
    ap.add_argument("--jobs", type=int, default=7)
    ap.add_argument("--out", default=None)
    ap.add_argument("scenes", nargs="*")
    a = ap.parse_args()
    scenes = a.scenes or sorted(glob.glob(os.path.join(HERE, "scenes/*.ron"))) + sorted(glob.glob(os.path.join(HERE, "tests/corpus/scenes/*.ron")))
    work = [(s, f, m) for s in scenes for f in a.flags for m in ("bottom-up", "module")]
    running, lines = [], []
    out = open(a.out, "w") if a.out else None
    def reap(block):
        for p, w in list(running):
            if block or p.poll() is not None:
                so, se = p.communicate()
                line = [l for l in so.splitlines() if l.startswith("{")]
                text = line[-1] if line else json.dumps({"scene": w[0], "flags": w[1], "inliner": w[2], "error": (se or so)[-300:]})
                print(text, flush=True)
                if out: out.write(text + "\n"); out.flush()
                lines.append(json.loads(text))
                running.remove((p, w))
    for w in work:
        while len(running) >= a.jobs:
            reap(False); time.sleep(0.05)
        running.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", w[0], str(w[1]), w[2]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), w))
    while running:
        reap(True)
    ok = [l for l in lines if "error" not in l]
    for f in a.flags:
        for m in ("bottom-up", "module"):
            sel = [l for l in ok if l["flags"] == f and l["inliner"] == m]
            if sel:
                print(f"# flags {f} {m}: {len(sel)} kernels, jit {sum(l['jit_s'] for l in sel):.1f} s total, with scratch: {sum(1 for l in sel if l['scratch'] > 0)}, with calls: {sum(1 for l in sel if l['calls'] > 0)}", flush=True)
