"""oracle/host_build.py -- the generated kernel source compiled for the HOST with g++.

TEST INFRASTRUCTURE ONLY.  Nothing under portal_amd/ imports this; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do.

What it is: the exact translation unit hiprtc gets for a scene (portal_amd.Scene
.generate_source) is also valid host C++ (device/ptl_entry.h ends in an OpenMP row loop
when PTL_DEVICE_BUILD is 0).  Compiled with `g++ -O2 -ffp-contract=off -mfma -fopenmp` it is
  * the CPU baseline BASELINE.md section 3 plans ("kind": "port"): the same arithmetic on
    the host cores, and
  * a cross-check of the *compiler and hardware* leg of parity: device/ptl_glsl.h fixes
    every operation in IEEE binary32, so gfx950 and x86-64 must agree bit for bit.
What it is NOT: an independent check of the codegen or of the prelude's logic -- that is
oracle/portal_oracle.py (numpy restatement, separate code path), which since round 3 is pinned to the reference's
own shader text (oracle/reference_shader.py, tests/test_reference_text.py).  This host build is product source
compiled by another compiler: agreement with it says "hipcc == g++ on this arithmetic", nothing more.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(_HERE, "_build")

CXXFLAGS = ["-std=c++20", "-O2", "-ffp-contract=off", "-mfma", "-fopenmp", "-fPIC", "-shared", "-x", "c++", "-w"]


def compile_host(source: str, count_segments: bool = False, opt: str = "-O2", defines=()) -> str:
    """g++-compile `source` into a shared object (cached by content hash); returns its path.
    The optimisation level cannot change results (IEEE arithmetic, -ffp-contract=off, no fast-math)."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    flags = [opt if f == "-O2" else f for f in CXXFLAGS] + (["-DPTL_COUNT_SEGMENTS"] if count_segments else []) + [f"-D{d}" for d in defines]
    key = hashlib.sha256((source + "\0" + " ".join(flags)).encode()).hexdigest()[:20]
    so = os.path.join(BUILD_DIR, f"host_{key}.so")
    if not os.path.exists(so):
        src = os.path.join(BUILD_DIR, f"host_{key}.cpp")
        with open(src, "w") as f:
            f.write(source)
        tmp = so + f".tmp{os.getpid()}"
        subprocess.run(["g++", *flags, src, "-o", tmp], check=True)
        os.replace(tmp, so)
    return so


class HostKernel:
    """The host-compiled kernel of one scene: set uniforms by name, render pixel windows."""

    def __init__(self, source: str, layout, block_size: int, count_segments: bool = False, opt: str = "-O2", defines=()):
        self.so_path = compile_host(source, count_segments, opt, defines)
        self.lib = C.CDLL(self.so_path)
        self.lib.ptl_host_uniform_block.restype = C.c_void_p
        self.lib.ptl_host_uniform_block.argtypes = [C.POINTER(C.c_ulong)]
        self.lib.ptl_host_render.restype = C.c_ulonglong
        self.lib.ptl_host_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        self.lib.ptl_host_teleport.restype = None
        self.lib.ptl_host_teleport.argtypes = [C.c_void_p]
        size = C.c_ulong()
        self.block = self.lib.ptl_host_uniform_block(C.byref(size))
        assert size.value >= block_size, (size.value, block_size)
        self.layout = {name: (typ, off) for name, typ, off in layout}
        self._textures = []

    def set_uniform(self, name: str, value) -> bool:
        if name not in self.layout:
            return False
        typ, off = self.layout[name]
        if typ == 2:  # int
            data = np.array([int(value)], dtype=np.int32)
        elif typ == 0:  # mat4 given as m[row, col] -> column-major
            data = np.ascontiguousarray(np.asarray(value, dtype=np.float32).T).reshape(16)
        else:
            data = np.atleast_1d(np.asarray(value, dtype=np.float32))
        C.memmove(self.block + off, data.ctypes.data, data.nbytes)
        return True

    def set_texture(self, sampler: str, rgba8: np.ndarray) -> bool:
        if sampler not in self.layout:
            return False
        tex = np.ascontiguousarray(rgba8, dtype=np.uint8)
        self._textures.append(tex)  # keep alive
        _, off = self.layout[sampler]
        rec = np.zeros(1, dtype=[("p", np.uint64), ("w", np.int32), ("h", np.int32)])
        rec["p"], rec["w"], rec["h"] = tex.ctypes.data, tex.shape[1], tex.shape[0]
        C.memmove(self.block + off, rec.ctypes.data, 16)
        return True

    def teleport_external_ray(self, a, b):
        """-> (pos float32[3] | None, encounter_object, change_subspace); needs _external_ray_a/_b in the layout."""
        self.set_uniform("_external_ray_a", np.asarray(a, np.float32))
        self.set_uniform("_external_ray_b", np.asarray(b, np.float32))
        out = np.zeros(6, np.float32)
        self.lib.ptl_host_teleport(out.ctypes.data)
        return (out[:3].copy() if out[3] else None), bool(out[4]), bool(out[5])

    def render(self, width: int, height: int, rows=None, cols=None, threads: int = 0, rgba8: bool = True, rgba32f: bool = True):
        """Render rows x cols of a width x height frame.  `rows` is a (r0, r1) range or an
        explicit list/array of row indices (output row i = rows[i]); `cols` a (c0, c1) range."""
        if rows is None:
            rows = (0, height)
        row_idx = np.arange(rows[0], rows[1], dtype=np.int32) if isinstance(rows, tuple) else np.ascontiguousarray(rows, dtype=np.int32)
        c0, c1 = cols if cols else (0, width)
        a8 = np.zeros((len(row_idx), c1 - c0, 4), np.uint8) if rgba8 else None
        a32 = np.zeros((len(row_idx), c1 - c0, 4), np.float32) if rgba32f else None
        threads = threads or (os.cpu_count() or 1)
        seg = self.lib.ptl_host_render(a8.ctypes.data if rgba8 else None, a32.ctypes.data if rgba32f else None, width, height,
                                       row_idx.ctypes.data, len(row_idx), c0, c1, threads)
        return {"rgba8": a8, "rgba32f": a32, "segments": int(seg)}


def host_kernel_for(renderer, scene, width: int, height: int, flags: int = 0, count_segments: bool = False, asset_root: str | None = None) -> HostKernel:
    """Build the host kernel of `scene` and load it with exactly the uniform values the product
    renderer (`portal_amd.SceneRenderer`, device may be -1) would upload for a width x height frame."""
    import portal_amd as pa

    source = scene.generate_source(flags)
    layout, size = scene.uniform_layout()
    # (what the generator asks the JIT for with this source; PTL_COUNT_SEGMENTS comes through `count_segments`, the occupancy hint is the renderer's)
    defines = tuple(d for d in scene.generated_defines() if d != "PTL_COUNT_SEGMENTS" and not d.startswith("PTL_WAVES_PER_EU") and d != "PTL_QUICK_JIT")
    hk = HostKernel(source, layout, size, count_segments, defines=defines)
    for name, typ, _ in layout:
        if typ == pa.PTL_SAMPLER:
            continue
        v = renderer.uniform_value(name, width, height)
        if v is not None:
            hk.set_uniform(name, v)
    # textures: decoded with PIL here (independent of the product's PNG reader)
    from PIL import Image

    root = asset_root or pa.REPO_ROOT
    paths = scene.textures()
    for name, typ, _ in layout:
        if typ == pa.PTL_SAMPLER:
            rel = paths.get(name[: -len("_tex")])  # videos and unreadable files stay unbound
            if rel is not None and os.path.exists(os.path.join(root, rel)):
                hk.set_texture(name, np.array(Image.open(os.path.join(root, rel)).convert("RGBA")))
    return hk
