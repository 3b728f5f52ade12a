#!/usr/bin/env python3
"""tools/isa_hist.py -- static instruction histogram of a generated kernel's gfx950 code object.

    python tools/isa_hist.py [--scene portal_in_portal | --scene-file tests/corpus/scenes/x.ron] [--flags 5] [--waves 4]
                             [--kernel ptl_render_kernel] [--out profiles/r04/isa_hist_<tag>.json] [--top 40]

Builds (or finds in the code-object cache) the kernel exactly as a SceneRenderer with those flags would, disassembles it with
llvm-objdump and counts mnemonics of one kernel symbol: per mnemonic, and per class -- FP32 arithmetic (fma / mul / add / mac),
transcendental (rcp / sqrt / rsq / sin / cos / exp / log), integer, compares, selects (v_cndmask), moves (v_mov, v_accvgpr, readlane),
other VALU; SALU, scalar branches, scalar memory, vector memory, LDS, waitcnt / nop.  STATIC counts (what the compiler emitted), the
companion of the PMC counts under profiles/ (what the hardware executed).  No GPU needed.
"""
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
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

FP_ARITH = re.compile(r"^v_(pk_)?(fma|fmac|mul|add|sub|mac|mad|subrev|fmaak|fmamk)_(f32|legacy_f32)")
TRANS = re.compile(r"^v_(rcp|sqrt|rsq|sin|cos|exp|log)_")
CMP = re.compile(r"^v_cmpx?_")
SELECT = re.compile(r"^v_cndmask")
MOVE = re.compile(r"^v_(mov|accvgpr|readlane|readfirstlane|writelane|swap|permlane|bfrev|mov_b64|pk_mov)")
INT = re.compile(r"^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|min|max|med3|add3|lshl_add|lshl_or|and_or|or3|xad|cvt_.*(i32|u32)|alignbit|perm|mbcnt|bcnt)_?(u32|i32|co_u32|b32|u16|i16|u24|i24|lo_u32|hi_u32|u64|nc_u32|.*)$")
FP_OTHER = re.compile(r"^v_(min|max|med3|floor|fract|trunc|ceil|rndne|cvt|frexp|ldexp|div_scale|div_fmas|div_fixup|cubeid)_?")


def classify(m: str) -> str:
    if m.startswith("v_"):
        if FP_ARITH.match(m):
            return "valu_fp32_arith"
        if TRANS.match(m):
            return "valu_transcendental"
        if CMP.match(m):
            return "valu_compare"
        if SELECT.match(m):
            return "valu_select"
        if MOVE.match(m):
            return "valu_move"
        if FP_OTHER.match(m) and ("f32" in m or "f64" in m or "f16" in m):
            return "valu_fp_other"
        if INT.match(m):
            return "valu_int"
        return "valu_other"
    if m.startswith("s_cbranch") or m in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_call_b64", "s_endpgm"):
        return "branch"
    if m.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "smem"
    if m.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio", "s_sethalt", "s_setreg", "s_getreg")):
        return "wait_nop"
    if m.startswith("s_"):
        return "salu"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_", "image_")):
        return "vmem"
    if m.startswith("ds_"):
        return "lds"
    return "other"


def disassemble(code: bytes, kernel: str) -> list[str]:
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
        f.write(code)
        f.flush()
        text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--no-show-raw-insn", f"--disassemble-symbols={kernel}", f.name], capture_output=True, text=True, check=True).stdout
    out = []
    for line in text.splitlines():
        line = line.strip()
        if not line or line.endswith(":") or line.startswith(("/", "Disassembly", ";")) or "file format" in line:
            continue
        m = line.split()[0]
        if re.match(r"^[a-z_0-9]+$", m):
            out.append(m)
    return out


def resources(code: bytes, kernel: str) -> dict:
    """VGPRs, SGPRs, scratch and LDS bytes of `kernel` from the code object's metadata note (no device needed)."""
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
        f.write(code)
        f.flush()
        text = subprocess.run([OBJDUMP.replace("objdump", "readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
    out, inside = {}, False
    for block in text.split("- .agpr_count:")[1:]:
        if re.search(r"\.name:\s+" + re.escape(kernel) + r"\s", block):
            for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                m = re.search(r"\." + key + r":\s+(\d+)", block)
                if m:
                    out[key] = int(m.group(1))
            m = re.match(r"\s*(\d+)", block)
            if m:
                out["agpr_count"] = int(m.group(1))
    return out


def histogram(code: bytes, kernel: str = "ptl_render_kernel") -> dict:
    mnems = disassemble(code, kernel)
    per = collections.Counter(mnems)
    classes = collections.Counter()
    for m, c in per.items():
        classes[classify(m)] += c
    valu = sum(c for k, c in classes.items() if k.startswith("valu_"))
    return {"kernel": kernel, "instructions": len(mnems), "valu": valu, "classes": dict(sorted(classes.items(), key=lambda kv: -kv[1])),
            "valu_share_of_bookkeeping": round((classes["valu_move"] + classes["valu_select"] + classes["valu_compare"]) / max(1, valu), 4),
            "mnemonics": dict(per.most_common())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="portal_in_portal")
    ap.add_argument("--scene-file", default="")
    ap.add_argument("--flags", type=int, default=5)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--kernel", default="ptl_render_kernel")
    ap.add_argument("--hsaco", default="", help="disassemble this code object instead of building one")
    ap.add_argument("--out", default="")
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    if a.hsaco:
        code, what = open(a.hsaco, "rb").read(), a.hsaco
    else:
        import portal_amd as pa

        path = os.path.join(HERE, a.scene_file) if a.scene_file else pa.scene_path(a.scene)
        extra = {"asset_root": os.path.dirname(os.path.dirname(path))} if a.scene_file else {}
        r = pa.SceneRenderer(pa.Scene.from_file(path), device=-1, flags=a.flags | pa.flag_waves(a.waves), **extra)
        code, what = r.code_object(), f"{os.path.relpath(path, HERE)} flags={a.flags} waves={a.waves}"
    h = histogram(code, a.kernel)
    h["build"] = what
    h["resources"] = resources(code, a.kernel)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(h, f, indent=1)
    print(f"{what}: {h['instructions']} instructions, {h['valu']} VALU, bookkeeping share of VALU {h['valu_share_of_bookkeeping']}, {h['resources']}")
    for k, v in h["classes"].items():
        print(f"  {k:22s} {v:6d}")
    for m, c in list(h["mnemonics"].items())[: a.top]:
        print(f"    {m:28s} {c:6d}")


if __name__ == "__main__":
    main()
