#!/usr/bin/env python3
"""tools/isa_lines.py -- which SOURCE lines the instructions of a generated kernel come from (where do the moves, compares,
selects and exec-mask bookkeeping of the headline kernel sit?).

    python tools/isa_lines.py [--scene portal_in_portal] [--flags 5] [--waves 5] [--out profiles/r06/isa_lines_<tag>.json] [--top 25]

Builds the kernel a SceneRenderer with those flags would draw with, once more with `-gline-tables-only` (line tables and inlined-subroutine
records, no other debug info: the instruction stream must be -- and is checked to be -- the one of the shipped build), disassembles it, asks
llvm-symbolizer for the inline chain of every instruction address and books each instruction, by class (tools/isa_hist.py), to
  * the innermost function it was inlined from (a `ptl_glsl.h` builtin, a library function, a scene snippet ...),
  * the first frame below the bounce loop's stages (`scene_intersect`, `intersect_material_N`, `material_process` ...): the STAGE it belongs to,
  * its source line (file:line of the generated translation unit, with the text of that line).
STATIC counts (what the compiler emitted), like isa_hist.py; the PMC files under profiles/ say what the hardware executed.  No GPU needed."""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "tools"))
from isa_hist import OBJDUMP, classify  # noqa: E402

SYMBOLIZER = OBJDUMP.replace("llvm-objdump", "llvm-symbolizer")
# frames that only carry others: the stage of an instruction is the first frame BELOW these on its inline chain
CARRIERS = {"ptl_render_kernel", "shade_pixel", "get_color", "get_color2", "ray_tracing", "trace_segment", "scene_intersect_material_process"}
BOOKKEEPING = ("valu_move", "valu_compare", "valu_select", "salu", "branch")


def build(scene_path, flags, extra_env):
    import portal_amd as pa

    saved = {k: os.environ.get(k) for k in extra_env}
    os.environ.update(extra_env)
    try:
        r = pa.SceneRenderer(pa.Scene.from_file(scene_path), device=-1, flags=flags)
        return r.code_object(), r.kernel_source()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def instructions(code, kernel):
    """[(address, mnemonic)] of one kernel symbol."""
    with tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False) as f:
        f.write(code)
        path = f.name
    text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f"--disassemble-symbols={kernel}", path], capture_output=True, text=True, check=True).stdout
    out = []
    for line in text.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\b.*//\s*([0-9A-F]{8,}):", line)
        if m:
            out.append((int(m.group(2), 16), m.group(1)))
    return out, path


def chains(path, addresses):
    """address -> [(function, line)] innermost first."""
    feed = "\n".join(hex(a) for a in addresses) + "\n"
    text = subprocess.run([SYMBOLIZER, f"--obj={path}", "--inlines", "-f", "--output-style=LLVM"], input=feed, capture_output=True, text=True, check=True).stdout
    out, cur = [], []
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        if not lines[i].strip():
            out.append(cur)
            cur = []
            i += 1
            continue
        fn = lines[i].strip()
        where = lines[i + 1].strip() if i + 1 < len(lines) else ""
        m = re.search(r":(\d+):\d+$", where)
        cur.append((fn, int(m.group(1)) if m else 0, os.path.basename(where.rsplit(":", 2)[0]) if m else where))
        i += 2
    if cur:
        out.append(cur)
    return dict(zip(addresses, out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="portal_in_portal")
    ap.add_argument("--flags", type=int, default=5)
    ap.add_argument("--waves", type=int, default=5)
    ap.add_argument("--kernel", default="ptl_render_kernel")
    ap.add_argument("--out", default="")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    import portal_amd as pa

    flags = a.flags | pa.flag_waves(a.waves)
    scene_path = pa.scene_path(a.scene)
    shipped, source = build(scene_path, flags, {})
    with tempfile.TemporaryDirectory() as cache:  # a cache of its own: the key does not know about PTL_HIPRTC_FLAGS of another process
        lined, _ = build(scene_path, flags, {"PTL_HIPRTC_FLAGS": "-gline-tables-only", "PTL_CACHE_DIR": cache})
    ins_shipped, p0 = instructions(shipped, a.kernel)
    ins, path = instructions(lined, a.kernel)
    os.unlink(p0)
    same_stream = [m for _, m in ins] == [m for _, m in ins_shipped]
    where = chains(path, [adr for adr, _ in ins])
    os.unlink(path)
    src_lines = source.splitlines()

    def text_of(line_no):
        return src_lines[line_no - 1].strip()[:110] if 0 < line_no <= len(src_lines) else ""

    by_fn = collections.defaultdict(collections.Counter)
    by_stage = collections.defaultdict(collections.Counter)
    by_line = collections.defaultdict(collections.Counter)
    for adr, mnem in ins:
        cls = classify(mnem)
        chain = where.get(adr) or [("?", 0, "?")]
        inner = chain[0]
        by_fn[inner[0]][cls] += 1
        stage = next((f for f, _, _ in reversed(chain) if f not in CARRIERS), chain[-1][0])
        by_stage[stage][cls] += 1
        # the line of the GENERATED unit on the chain nearest to the instruction (a builtin's own line in the prelude says little)
        unit = next(((f, ln) for f, ln, file in chain if file.endswith("portal_scene.hip")), (inner[0], inner[1]))
        by_line[unit[1]][cls] += 1

    def rows(table, key_name, top):
        out = []
        for k, c in sorted(table.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
            valu = sum(v for kk, v in c.items() if kk.startswith("valu_"))
            out.append({key_name: k, "instructions": sum(c.values()), "valu": valu, "fp32_arith": c["valu_fp32_arith"], "transcendental": c["valu_transcendental"],
                        "moves": c["valu_move"], "compares": c["valu_compare"], "selects": c["valu_select"], "salu": c["salu"], "branches": c["branch"]})
        return out

    total = collections.Counter()
    for c in by_fn.values():
        total.update(c)
    result = {"build": f"{os.path.relpath(scene_path, HERE)} flags={a.flags} waves={a.waves}", "kernel": a.kernel, "instructions": len(ins),
              "same_instruction_stream_as_the_shipped_build": same_stream, "classes": dict(total.most_common()),
              "by_stage": rows(by_stage, "stage", 40), "by_innermost_function": rows(by_fn, "function", 60)}
    lines = []
    for ln, c in sorted(by_line.items(), key=lambda kv: -sum(kv[1][k] for k in BOOKKEEPING))[: max(a.top, 40)]:
        lines.append({"line": ln, "text": text_of(ln), "bookkeeping": sum(c[k] for k in BOOKKEEPING), "instructions": sum(c.values()), "moves": c["valu_move"],
                      "compares": c["valu_compare"], "selects": c["valu_select"], "salu": c["salu"], "branches": c["branch"], "fp32_arith": c["valu_fp32_arith"]})
    result["lines_with_most_bookkeeping"] = lines
    if a.out:
        with open(a.out, "w") as f:
            json.dump(result, f, indent=1)
    print(f"{result['build']}: {len(ins)} instructions; same stream as the shipped build: {same_stream}")
    print("by stage:")
    for r in result["by_stage"][: a.top]:
        print(f"  {r['stage'][:44]:44s} {r['instructions']:5d}  valu {r['valu']:5d}  arith {r['fp32_arith']:5d}  mov {r['moves']:4d}  cmp {r['compares']:4d}  sel {r['selects']:4d}  salu {r['salu']:4d}  br {r['branches']:3d}")
    print("by innermost function:")
    for r in result["by_innermost_function"][: a.top]:
        print(f"  {r['function'][:44]:44s} {r['instructions']:5d}  valu {r['valu']:5d}  arith {r['fp32_arith']:5d}  mov {r['moves']:4d}  cmp {r['compares']:4d}  sel {r['selects']:4d}  salu {r['salu']:4d}  br {r['branches']:3d}")
    print("lines of the generated unit with most moves + compares + selects + SALU + branches:")
    for r in lines[: a.top]:
        print(f"  {r['line']:5d} {r['bookkeeping']:4d}/{r['instructions']:4d}  mov {r['moves']:3d} cmp {r['compares']:3d} sel {r['selects']:3d} salu {r['salu']:3d} br {r['branches']:3d} | {r['text']}")


if __name__ == "__main__":
    main()
