"""portal_amd -- MI355X-native offline renderer for optozorax/portal scenes.

Python is only the thin host-side mirror of the reference's operator interface
(``Scene`` / ``SceneRenderer``, reference src/gui/scene.rs and src/main.rs:732-1544) over the
C ABI of ``libportal_amd.so`` (include/portal_amd.h).  All work -- .ron loading, uniform /
matrix evaluation, scene -> HIP source generation, hiprtc compilation, kernel launch --
happens in the native library; there is no Python or CPU fallback for the render path: if
the library or a GPU is missing the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libportal_amd.so")

# default code-object cache (gfx950 binaries keyed by source hash); travels with the repo
os.environ.setdefault("PTL_CACHE_DIR", os.path.join(_HERE, "_cache"))



def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process: PyTorch ships its own libamdhip64 and loads it on `import torch`.  If this library
    bound /opt/rocm's copy first and torch came later, the process would hold two runtimes whose streams and device
    pointers do not mix.  So when torch is installed but not imported yet, point the C side at torch's copy by path
    (importing torch itself would cost seconds for callers that never need it)."""
    if "PTL_HIP_LIB" in os.environ:
        return
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for root in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        cand = os.path.join(root, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            os.environ["PTL_HIP_LIB"] = cand  # hiprtc stays /opt/rocm's: a compiler, not shared state
            return


_share_torch_hip_runtime()

PTL_MAT4, PTL_F32, PTL_I32, PTL_VEC2, PTL_VEC3, PTL_SAMPLER = range(6)
FLAG_SPECIALIZE_INTS = 1
FLAG_COUNT_SEGMENTS = 2
FLAG_SPECIALIZE_ALL = 4
FLAG_SPECIALIZE_STATIC = 8  # bake what stays constant while a clip plays (checked before every draw, rebuilt if it moved)
FLAG_ANAGLYPH = 16  # compile the anaglyph stereo mode in (the reference's `disable_anaglyph = false`)
FLAG_NO_DERIVED_UNIFORMS = 32  # keep the per-call plane tests (default: ray-independent halves evaluated by the prologue kernel)
FLAG_NO_FIRST_TRIP = 8192  # no first-trip copies of the intersection-material snippets (round 6: the un-specialised kernel has none unless FLAG_KEEP_TRANSFORM_DODGES asks)
FLAG_NO_UNIFORM_HOIST = 4096  # scene snippets evaluate their uniform-only expressions per ray (default: once per upload, in the prologue kernel)
FLAG_NO_DEFERRED_UPDATES = 128  # translate scene snippets exactly as written (default: loop-carried ray transforms are applied lazily)
FLAG_QUICK_JIT = 262144  # compile at -O1 instead of the shipped -O3: half the JIT time, a 5-20 % slower kernel (one-off frames)
FLAG_SPECIALIZE_PATTERNS = 1048576  # compile in only what survives moving values: zero patterns of the matrices + the renderer's mode switches
FLAG_BOUNDED_SNIPPETS = 2097152  # opt-in: scene_intersect first, its hit distance bounds the intersection-material snippets (exact; measured: no gain on the headline)
FLAG_KEEP_TRANSFORM_DODGES = 16777216  # A/B, round 4's shape: the first-trip forms (snippet copies + plane tests) also in the un-specialised kernel (opt-in there since round 6), snippet copies + deferred loop updates also with affine rays; identical frames
FLAG_MATERIAL_TABLE_LDS = 1 << 26  # A/B (measured slower, off by default): the Simple materials' literals in a per-workgroup LDS table, ONE material_simple2 call
FLAG_MATERIAL_TABLE_SCALAR = 1 << 27  # A/B (measured: no gain): the same table in constant memory, a scalar load per distinct material of the wave (waterfall)
FLAG_CHECK_AFFINE = 1 << 25  # diagnostics: general products, and `segments` counts the ray halves that meet a product / the bounce loop with a w that is not 1 / 0
FLAG_NO_AFFINE_RAYS = 8388608  # A/B: matrix-times-ray products never assume o.w = 1 / d.w = 0 (default in specialised builds of affine scenes: they do; identical frames)
FLAG_SLICES = 4194304  # the render entry reads its uniform block from a buffer of blocks (one per blockIdx.z): stage_slice / draw_slices, one launch for several draws
FLAG_NO_ZERO_MASKS = 524288  # A/B: run-time matrices keep their full products although their zero pattern is known (KernelOptions::mask_zero_elements)
FLAG_ASYNC_REJIT = 131072  # a specialised renderer never stalls on a rebuild: it draws with the un-specialised kernel until a worker thread has the new one
FLAG_NO_FIRST_TRIP_PLANES = 65536  # no first-trip copy of the generated plane tests (round 6: the un-specialised kernel has none unless FLAG_KEEP_TRANSFORM_DODGES asks)
FLAG_NO_UNROLL = 32768  # keep snippet loops whose bound is a baked Int uniform as loops (default: unrolled up to 16 iterations; identical frames)
FLAG_EXACT_CR = 16384  # numerics contract 1 of rounds 1-2: IEEE correctly rounded / and sqrt on EVERY input (default: contract 2, device/ptl_glsl.h)
FLAG_FAST_MATH = 64  # tolerance mode: hardware rcp / sqrt estimates, FMA contraction (not bit-exact; exact stays the default)


def flag_waves(n: int) -> int:
    """Occupancy hint for SceneRenderer flags: build with __launch_bounds__(256, n)."""
    return (int(n) & 0xF) << 8


class PortalError(RuntimeError):
    pass


class UniformDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int), ("offset", C.c_size_t)]


class CalculatedCam(C.Structure):
    """ptl_calculated_cam (CalculatedCam, src/gui/camera.rs:22-32)."""

    _fields_ = [("look_at", C.c_double * 3), ("alpha", C.c_double), ("beta", C.c_double), ("r", C.c_double), ("free_movement", C.c_int),
                ("in_subspace", C.c_int), ("override_matrix", C.c_int), ("matrix", C.c_double * 16)]


class Frame(C.Structure):
    """Row-block sharding of one frame (ptl_frame)."""

    _fields_ = [("width", C.c_int), ("height", C.c_int), ("rb_phase", C.c_int), ("rb_stride", C.c_int), ("in_place", C.c_int)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise PortalError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make`")
    lib = C.CDLL(LIB_PATH)
    vp, cp, ci, cd, cs = C.c_void_p, C.c_char_p, C.c_int, C.c_double, C.c_size_t
    P = C.POINTER
    sig = {
        "ptl_last_error": (cp, []),
        "ptl_version": (cp, []),
        "ptl_device_count": (ci, []),
        "ptl_kernel_compile": (ci, [ci, cp, P(UniformDesc), ci, cs, P(cp), ci, P(vp), cp, cs]),
        "ptl_kernel_code_object": (ci, [vp, P(vp), P(cs)]),
        "ptl_kernel_resources": (ci, [vp, P(ci), P(ci), P(ci)]),
        "ptl_kernel_set_uniform": (ci, [vp, cp, ci, vp]),
        "ptl_kernel_set_texture": (ci, [vp, cp, vp, ci, ci]),
        "ptl_frame_shard_rows": (ci, [P(Frame)]),
        "ptl_kernel_render": (ci, [vp, P(Frame), vp, vp, vp, vp, P(C.c_float)]),
        "ptl_kernel_render_to_host": (ci, [vp, P(Frame), vp, vp, P(C.c_uint64), P(C.c_float)]),
        "ptl_kernel_teleport_ray": (ci, [vp, P(C.c_float), P(C.c_float), P(C.c_float), P(ci), P(ci), P(ci)]),
        "ptl_kernel_destroy": (None, [vp]),
        "ptl_scene_load_file": (ci, [cp, P(vp)]),
        "ptl_scene_load_text": (ci, [cp, P(vp)]),
        "ptl_scene_free": (None, [vp]),
        "ptl_scene_set_uniform": (ci, [vp, cp, cd]),
        "ptl_scene_set_time": (ci, [vp, cd, cd]),
        "ptl_scene_set_trefoil": (ci, [vp, cp, cp]),
        "ptl_scene_get_trefoil": (ci, [vp, cp, cp, cs]),
        "ptl_scene_init_stage": (ci, [vp, cp, cp, cs]),
        "ptl_scene_stage_name": (ci, [vp, ci, cp, cs]),
        "ptl_scene_camera_name": (ci, [vp, ci, cp, cs]),
        "ptl_scene_set_camera_matrix": (ci, [vp, P(cd)]),
        "ptl_scene_animation": (ci, [vp, ci, cp, cs, P(cd)]),
        "ptl_scene_init_animation": (ci, [vp, cp]),
        "ptl_scene_update": (ci, [vp, cd, P(cd), P(cd), P(ci), P(CalculatedCam)]),
        "ptl_renderer_update": (ci, [vp, cd, P(ci), P(ci)]),
        "ptl_renderer_rejit_count": (ci, [vp]),
        "ptl_renderer_rejit_pending": (ci, [vp]),
        "ptl_scene_eval_uniform": (ci, [vp, cp, P(ci), P(cd)]),
        "ptl_scene_eval_matrix": (ci, [vp, cp, P(cd)]),
        "ptl_scene_cam": (ci, [vp, P(cd)]),
        "ptl_scene_texture": (ci, [vp, ci, cp, cs, cp, cs]),
        "ptl_scene_generate_source": (ci, [vp, C.c_uint, P(vp)]),
        "ptl_scene_generated_defines": (ci, [vp, cp, cs]),
        "ptl_scene_zero_mask_probes": (ci, [vp, P(ci), P(ci)]),
        "ptl_scene_uniform_layout": (ci, [vp, P(P(UniformDesc)), P(ci), P(cs)]),
        "ptl_scene_set_uniforms": (ci, [vp, vp]),
        "ptl_scene_visit_uniforms": (ci, [vp, vp, vp]),
        "ptl_scene_source_line_owner": (ci, [vp, ci, cp, cs, cp, cs, P(ci)]),
        "ptl_free": (None, [vp]),
        "ptl_renderer_create": (ci, [vp, ci, cp, C.c_uint, P(vp), cp, cs]),
        "ptl_renderer_kernel_source": (ci, [vp, P(vp)]),
        "ptl_renderer_join": (ci, [vp, vp]),
        "ptl_renderer_stage_slice": (ci, [vp, P(Frame), ci]),
        "ptl_renderer_draw_slices": (ci, [vp, P(Frame), ci, vp, vp, C.c_ulonglong, vp, P(C.c_float)]),
        "ptl_kernel_max_slices": (ci, [vp]),
        "ptl_kernel_stage_slice": (ci, [vp, ci]),
        "ptl_kernel_hold_textures": (ci, [vp, ci]),
        "ptl_renderer_affine_rays": (ci, [vp]),
        "ptl_renderer_check_affine": (ci, [vp, ci, ci, P(C.c_ulonglong)]),
        "ptl_dmath": (ci, [cp, P(cd), P(cd), P(cd), P(cd)]),
        "ptl_snippets_keep_rays_affine": (ci, [cp, cp, cs]),
        "ptl_code_object_note": (ci, [vp, cs, cp, cp]),
        "ptl_kernel_render_slices": (ci, [vp, P(Frame), ci, vp, vp, C.c_ulonglong, vp, P(C.c_float)]),
        "ptl_kernel_clone": (ci, [vp, P(vp)]),
        "ptl_kernel_copy_uniforms": (ci, [vp, vp]),
        "ptl_renderer_create_with_options": (ci, [vp, ci, cp, C.c_uint, P(cp), P(cd), ci, P(vp), cp, cs]),
        "ptl_renderer_set_option": (ci, [vp, cp, cd]),
        "ptl_renderer_set_camera": (ci, [vp, P(cd), cd, cd, cd]),
        "ptl_renderer_use_camera": (ci, [vp, cp]),
        "ptl_renderer_uniform_value": (ci, [vp, ci, ci, cp, P(C.c_float), P(ci)]),
        "ptl_renderer_draw": (ci, [vp, P(Frame), vp, vp, vp, vp, P(C.c_float)]),
        "ptl_renderer_draw_to_host": (ci, [vp, P(Frame), vp, vp, P(C.c_uint64), P(C.c_float)]),
        "ptl_renderer_teleport_ray": (ci, [vp, P(cd), P(cd), P(cd), P(ci), P(ci), P(ci)]),
        "ptl_renderer_prebuild_teleport": (ci, [vp]),
        "ptl_kernel_prebuild_teleport": (ci, [vp]),
        "ptl_renderer_move_camera": (ci, [vp, P(cd), cd, cd, cd, P(ci), P(ci)]),
        "ptl_renderer_camera_state": (ci, [vp, P(cd), P(ci), P(cd)]),
        "ptl_renderer_kernel": (vp, [vp]),
        "ptl_renderer_destroy": (None, [vp]),
        "ptl_deinterleave_rows": (ci, [vp, P(Frame), vp]),
        "ptl_average_images": (ci, [ci, P(vp), ci, vp, ci, ci, vp, P(C.c_float)]),
        "ptl_device_alloc": (ci, [ci, cs, P(vp)]),
        "ptl_device_free": (ci, [vp]),
        "ptl_device_download": (ci, [vp, vp, cs, vp]),
        "ptl_device_copy2d_async": (ci, [vp, cs, vp, cs, cs, cs, vp]),
        "ptl_ipc_export": (ci, [vp, cp]),
        "ptl_ipc_open": (ci, [ci, cp, P(vp)]),
        "ptl_ipc_close": (ci, [vp]),
        "ptl_frame_group_create": (ci, [vp, P(ci), ci, cp, C.c_uint, ci, P(vp), cp, cs]),
        "ptl_frame_group_size": (ci, [vp]),
        "ptl_frame_group_renderer": (vp, [vp, ci]),
        "ptl_frame_group_set_option": (ci, [vp, cp, cd]),
        "ptl_frame_group_use_camera": (ci, [vp, cp]),
        "ptl_frame_group_set_camera": (ci, [vp, P(cd), cd, cd, cd]),
        "ptl_frame_group_update": (ci, [vp, cd]),
        "ptl_frame_group_draw": (ci, [vp, ci, ci, P(vp), P(C.c_float)]),
        "ptl_frame_group_submit": (ci, [vp, ci, ci, P(ci)]),
        "ptl_frame_group_wait": (ci, [vp, ci, P(vp), P(C.c_float)]),
        "ptl_frame_group_download": (ci, [vp, vp]),
        "ptl_frame_group_destroy": (None, [vp]),
        "ptl_host_alloc": (ci, [cs, P(vp)]),
        "ptl_host_free": (ci, [vp]),
        "ptl_ron_format": (vp, [cp]),
        "ptl_scene_to_ron": (ci, [vp, P(vp)]),
        "ptl_png_read": (ci, [cp, P(vp), P(ci), P(ci)]),
        "ptl_png_write": (ci, [cp, vp, ci, ci]),
        "ptl_strstore_new": (vp, []),
        "ptl_strstore_free": (None, [vp]),
        "ptl_strstore_add_string": (None, [vp, cp]),
        "ptl_strstore_add_identifier_string": (None, [vp, cp, cp, cp]),
        "ptl_apply_template": (vp, [cp, P(cp), P(vp), ci]),
        "ptl_strstore_text": (cp, [vp]),
        "ptl_strstore_current_line": (ci, [vp]),
        "ptl_strstore_range": (ci, [vp, cp, cp, P(ci), P(ci)]),
        "ptl_strstore_get_identifier": (ci, [vp, ci, cp, cs, cp, cs, P(ci)]),
        "ptl_device_source": (cp, [cp]),
        "ptl_translate_glsl": (vp, [cp]),
        "ptl_translate_library_glsl": (vp, [cp]),
        "ptl_bound_glsl": (vp, [cp, cp, P(ci)]),
        "ptl_formula_eval": (ci, [cp, P(cp), P(cd), ci, cd, P(cd)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return lib


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _err() -> str:
    return lib().ptl_last_error().decode("utf-8", "replace")


def last_error() -> str:
    """ptl_last_error(): the message of the most recent failure -- or note, e.g. why a renderer switched affine rays off -- on this thread."""
    return _err()


def _check(rc: int, what: str) -> int:
    if rc < 0:
        raise PortalError(f"{what} failed ({rc}): {_err()}")
    return rc


def version() -> str:
    return lib().ptl_version().decode()


def device_count() -> int:
    return lib().ptl_device_count()


def shard_rows(frame: Frame) -> int:
    return lib().ptl_frame_shard_rows(C.byref(frame))


# --------------------------------------------------------------------------------------------
class Scene:
    """A loaded .ron scene (reference: Scene::from_serialized, src/gui/scene.rs:142-146)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_file(cls, path: str) -> "Scene":
        h = C.c_void_p()
        rc = lib().ptl_scene_load_file(path.encode(), C.byref(h))
        if rc != 0:
            raise PortalError(f"Failed to parse scene `{path}`: {_err()}")
        return cls(h)

    @classmethod
    def from_text(cls, text: str) -> "Scene":
        h = C.c_void_p()
        rc = lib().ptl_scene_load_text(text.encode("utf-8"), C.byref(h))
        if rc != 0:
            raise PortalError(f"Failed to parse scene: {_err()}")
        return cls(h)

    def __del__(self):
        try:
            if self._h:
                lib().ptl_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_uniform(self, name: str, value: float) -> bool:
        return lib().ptl_scene_set_uniform(self._h, name.encode(), float(value)) == 0

    def set_time(self, time: float, total_time: Optional[float] = None) -> None:
        lib().ptl_scene_set_time(self._h, float(time), float(time if total_time is None else total_time))

    def init_stage(self, name: str) -> str:
        """Scene::init_stage_by_name (`render-frame --stage`).  Returns the camera the stage selects ("" = original)."""
        cam = C.create_string_buffer(256)
        rc = _check(lib().ptl_scene_init_stage(self._h, name.encode("utf-8"), cam, 256), "init_stage")
        if rc != 0:
            raise PortalError(f"Scene has no stage named `{name}`")
        return cam.value.decode("utf-8")

    def set_camera_matrix(self, m) -> None:
        """What Matrix::Camera evaluates to: 4x4, m[row][col] (a renderer sends its camera's matrix itself)."""
        a = (C.c_double * 16)(*np.asarray(m, np.float64).reshape(4, 4).T.reshape(-1))
        _check(lib().ptl_scene_set_camera_matrix(self._h, a), "set_camera_matrix")

    def set_trefoil(self, name: str, text: str) -> None:
        """A Trefoil uniform from the reference's text form, e.g. "1a 2a G,1b 3b B,2a 1a S" (TrefoilSpecial::decode)."""
        if _check(lib().ptl_scene_set_trefoil(self._h, name.encode(), text.encode()), "set_trefoil") != 0:
            raise PortalError(f"`{name}` is not a Trefoil uniform, or `{text}` does not decode")

    def get_trefoil(self, name: str) -> str:
        buf = C.create_string_buffer(512)
        if _check(lib().ptl_scene_get_trefoil(self._h, name.encode(), buf, 512), "get_trefoil") != 0:
            raise PortalError(f"`{name}` is not a Trefoil uniform")
        return buf.value.decode()

    def to_ron(self) -> str:
        """The scene as a .ron document in the reference's layout (its writer: serialize_scene_new_format + pretty RON)."""
        p = C.c_void_p()
        _check(lib().ptl_scene_to_ron(self._h, C.byref(p)), "to_ron")
        try:
            return C.string_at(p).decode("utf-8")
        finally:
            lib().ptl_free(p)

    def animations(self):
        """[(name, duration seconds)] of the scene's real animations (the clips `render` turns into videos)."""
        out, i, buf, dur = [], 0, C.create_string_buffer(256), C.c_double()
        while lib().ptl_scene_animation(self._h, i, buf, 256, C.byref(dur)) == 0:
            out.append((buf.value.decode("utf-8"), dur.value))
            i += 1
        return out

    def init_animation(self, name: str) -> None:
        """Scene::init_animation_by_name (`render-frame --animation`, `render`)."""
        if _check(lib().ptl_scene_init_animation(self._h, name.encode("utf-8")), "init_animation") != 0:
            raise PortalError(f"Scene has no animation named `{name}`")

    def update(self, seconds: float) -> dict:
        """Scene::update -> {"time", "total_time", "camera": None | dict(look_at, alpha, beta, r, ...)}."""
        t, tt, has, cam = C.c_double(), C.c_double(), C.c_int(), CalculatedCam()
        _check(lib().ptl_scene_update(self._h, float(seconds), C.byref(t), C.byref(tt), C.byref(has), C.byref(cam)), "update")
        camera = None
        if has.value:
            camera = dict(look_at=list(cam.look_at), alpha=cam.alpha, beta=cam.beta, r=cam.r, free_movement=bool(cam.free_movement),
                          in_subspace=bool(cam.in_subspace), override_matrix=bool(cam.override_matrix),
                          matrix=np.array(cam.matrix, np.float64).reshape(4, 4).T.copy())
        return {"time": t.value, "total_time": tt.value, "camera": camera}

    def _names(self, fn):
        out, i, buf = [], 0, C.create_string_buffer(256)
        while fn(self._h, i, buf, 256) == 0:
            out.append(buf.value.decode("utf-8"))
            i += 1
        return out

    def stages(self):
        return self._names(lib().ptl_scene_stage_name)

    def cameras(self):
        return self._names(lib().ptl_scene_camera_name)

    def eval_uniform(self, name: str):
        """AnyUniform::get -> bool | int | float, or None if it cannot be evaluated."""
        kind, val = C.c_int(), C.c_double()
        rc = lib().ptl_scene_eval_uniform(self._h, name.encode(), C.byref(kind), C.byref(val))
        _check(rc, "eval_uniform")
        if rc != 0:
            return None
        return bool(val.value) if kind.value == 0 else int(val.value) if kind.value == 1 else val.value

    def eval_matrix(self, name: str) -> Optional[np.ndarray]:
        """Matrix::get as a 4x4 float64 array, m[row, col] (stored column-major in the ABI)."""
        out = (C.c_double * 16)()
        rc = lib().ptl_scene_eval_matrix(self._h, name.encode(), out)
        _check(rc, "eval_matrix")
        if rc != 0:
            return None
        return np.array(out, dtype=np.float64).reshape(4, 4).T.copy()

    def cam(self) -> dict:
        out = (C.c_double * 7)()
        _check(lib().ptl_scene_cam(self._h, out), "scene_cam")
        return {"look_at": tuple(out[0:3]), "alpha": out[3], "beta": out[4], "r": out[5], "offset_after_material": out[6]}

    def textures(self) -> dict:
        """name -> path of the scene's textures (src/gui/texture.rs)."""
        out, i = {}, 0
        name, path = C.create_string_buffer(256), C.create_string_buffer(1024)
        while lib().ptl_scene_texture(self._h, i, name, 256, path, 1024) == 0:
            out[name.value.decode()] = path.value.decode()
            i += 1
        return out

    def generate_source(self, flags: int = 0) -> str:
        """Scene::generate_shader_code: the complete HIP C++ translation unit."""
        p = C.c_void_p()
        _check(lib().ptl_scene_generate_source(self._h, flags, C.byref(p)), "generate_source")
        try:
            return C.string_at(p).decode("utf-8")
        finally:
            lib().ptl_free(p)

    def zero_mask_probes(self):
        """(reused, probed): generations that reused the last zero patterns / that probed the scene for them."""
        a, b = C.c_int(), C.c_int()
        _check(lib().ptl_scene_zero_mask_probes(self._h, C.byref(a), C.byref(b)), "zero_mask_probes")
        return a.value, b.value

    def generated_defines(self):
        """The preprocessor defines that go with the source generated last (``generate_source`` / a renderer's build)."""
        buf = C.create_string_buffer(4096)
        _check(lib().ptl_scene_generated_defines(self._h, buf, len(buf)), "generated_defines")
        return tuple(buf.value.decode().split())

    def uniform_layout(self):
        """[(name, type, offset)], block_size -- Scene::uniforms plus the block layout."""
        descs, n, size = C.POINTER(UniformDesc)(), C.c_int(), C.c_size_t()
        _check(lib().ptl_scene_uniform_layout(self._h, C.byref(descs), C.byref(n), C.byref(size)), "uniform_layout")
        return [(descs[i].name.decode(), descs[i].type, descs[i].offset) for i in range(n.value)], size.value

    def uniform_values(self) -> dict:
        """What Scene::set_uniforms uploads: name -> float32 array (mat4: 4x4 m[row, col]) or int."""
        out = {}
        CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p)

        def cb(_user, name, typ, value):
            if typ == PTL_MAT4:
                out[name.decode()] = np.ctypeslib.as_array(C.cast(value, C.POINTER(C.c_float)), (16,)).copy().reshape(4, 4).T.copy()
            elif typ == PTL_I32:
                out[name.decode()] = int(C.cast(value, C.POINTER(C.c_int))[0])
            else:
                out[name.decode()] = np.float32(C.cast(value, C.POINTER(C.c_float))[0])

        fn = CB(cb)
        _check(lib().ptl_scene_visit_uniforms(self._h, C.cast(fn, C.c_void_p), None), "visit_uniforms")
        return out

    def source_line_owner(self, line: int):
        kind, name, local = C.create_string_buffer(64), C.create_string_buffer(256), C.c_int()
        rc = lib().ptl_scene_source_line_owner(self._h, line, kind, 64, name, 256, C.byref(local))
        if rc != 0:
            return None
        return kind.value.decode(), name.value.decode(), local.value


class SceneRenderer:
    """Reference: SceneRenderer (src/main.rs:732-1544), the offline image path.

    ``device=-1`` builds a GPU-less renderer (source generation, hiprtc compile, uniform
    queries); drawing then raises.
    """

    def __init__(self, scene: Scene, device: int = 0, asset_root: Optional[str] = None, flags: int = 0, options: Optional[dict] = None):
        """``options``: renderer options (``set_option`` names) applied before the first build -- a specialised renderer compiles its
        mode switches in, so e.g. ``options={"draw_side_by_side": 1}`` gets the side-by-side kernel from the start."""
        self.scene = scene
        h = C.c_void_p()
        log = C.create_string_buffer(1 << 16)
        root = (asset_root if asset_root is not None else REPO_ROOT).encode()
        if options:
            names = (C.c_char_p * len(options))(*[k.encode() for k in options])
            values = (C.c_double * len(options))(*[float(v) for v in options.values()])
            rc = lib().ptl_renderer_create_with_options(scene._h, device, root, flags, names, values, len(options), C.byref(h), log, len(log))
        else:
            rc = lib().ptl_renderer_create(scene._h, device, root, flags, C.byref(h), log, len(log))
        self.compile_log = log.value.decode("utf-8", "replace")
        if rc != 0:
            raise PortalError(f"SceneRenderer::new failed ({rc}): {_err()}\n{self.compile_log}")
        self._h = h
        self.device = device

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ptl_renderer_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def stage_slice(self, frame: Frame, index: int) -> None:
        """Everything a draw of `frame` does short of launching; the uniform block it arrives at becomes slice `index` (FLAG_SLICES renderers)."""
        _check(lib().ptl_renderer_stage_slice(self._h, C.byref(frame), index), "ptl_renderer_stage_slice")

    def draw_slices(self, frame: Frame, n: int, out_rgba8: int = 0, out_rgba32f: int = 0, slice_pixels: int = 0, stream: int = 0, timed: bool = False):
        """ONE launch for the staged slices 0 .. n-1: slice z at out_rgba8 + 4 * z * slice_pixels bytes (device addresses)."""
        ms = C.c_float()
        _check(lib().ptl_renderer_draw_slices(self._h, C.byref(frame), n, C.c_void_p(out_rgba8 or None), C.c_void_p(out_rgba32f or None), slice_pixels,
                                              C.c_void_p(stream or None), C.byref(ms) if timed else None), "ptl_renderer_draw_slices")
        return ms.value if timed else None

    def join(self, stream: int = 0) -> None:
        """With ``set_option("concurrent_draws", K)``: put `stream` (0 = the default stream) behind every draw issued so far."""
        _check(lib().ptl_renderer_join(self._h, C.c_void_p(stream or None)), "ptl_renderer_join")

    def kernel_source(self) -> str:
        """The translation unit the current kernel was compiled from (mode switches compiled in where the build is specialised)."""
        p = C.c_void_p()
        _check(lib().ptl_renderer_kernel_source(self._h, C.byref(p)), "kernel_source")
        try:
            return C.string_at(p).decode("utf-8")
        finally:
            lib().ptl_free(p)

    def set_option(self, name: str, value: float) -> None:
        rc = lib().ptl_renderer_set_option(self._h, name.encode(), float(value))
        if rc != 0:
            raise PortalError(f"unknown renderer option `{name}`")

    def set_camera(self, look_at, alpha: float, beta: float, r: float) -> None:
        la = (C.c_double * 3)(*look_at)
        _check(lib().ptl_renderer_set_camera(self._h, la, alpha, beta, r), "set_camera")

    def use_camera(self, name: str) -> None:
        """`render-frame --camera NAME`; "" restores the scene's own `cam` block."""
        if _check(lib().ptl_renderer_use_camera(self._h, name.encode("utf-8")), "use_camera") != 0:
            raise PortalError(f"Scene has no camera named `{name}`")

    def uniform_value(self, name: str, width: int, height: int):
        out, n = (C.c_float * 16)(), C.c_int()
        rc = lib().ptl_renderer_uniform_value(self._h, width, height, name.encode(), out, C.byref(n))
        _check(rc, "uniform_value")
        if rc != 0:
            return None
        a = np.array(out[: n.value], dtype=np.float32)
        return a.reshape(4, 4).T.copy() if n.value == 16 else (a[0] if n.value == 1 else a)

    def teleport_external_ray(self, a, b):
        """SceneRenderer::teleport_external_ray (src/main.rs:1361-1409) -> (pos | None, encounter_object, change_subspace)."""
        pa_, pb_, out = (C.c_double * 3)(*a), (C.c_double * 3)(*b), (C.c_double * 3)()
        hit, sub, tel = C.c_int(), C.c_int(), C.c_int()
        _check(lib().ptl_renderer_teleport_ray(self._h, pa_, pb_, out, C.byref(hit), C.byref(sub), C.byref(tel)), "teleport_external_ray")
        return (tuple(out) if tel.value else None), bool(hit.value), bool(sub.value)

    def prebuild_teleport(self) -> None:
        """Compile (or load from the cache) the camera-teleport half of the current kernel now instead of at the first query."""
        _check(lib().ptl_renderer_prebuild_teleport(self._h), "prebuild_teleport")

    def resources(self) -> dict:
        """Per-lane registers / scratch bytes / LDS bytes of the loaded kernel."""
        regs, scratch, lds = C.c_int(), C.c_int(), C.c_int()
        _check(lib().ptl_kernel_resources(lib().ptl_renderer_kernel(self._h), C.byref(regs), C.byref(scratch), C.byref(lds)), "kernel_resources")
        return {"registers": regs.value, "scratch_bytes": scratch.value, "lds_bytes": lds.value}

    def move_camera(self, look_at, alpha: float, beta: float, r: float):
        """One interactive camera step incl. SceneRenderer::teleport_camera.  -> (teleported, blocked)"""
        la, tel, blk = (C.c_double * 3)(*look_at), C.c_int(), C.c_int()
        _check(lib().ptl_renderer_move_camera(self._h, la, alpha, beta, r, C.byref(tel), C.byref(blk)), "move_camera")
        return bool(tel.value), bool(blk.value)

    def update(self, seconds: float):
        """SceneRenderer::update: the per-frame step of the video pipeline.  -> (teleported, blocked)"""
        tel, blk = C.c_int(), C.c_int()
        _check(lib().ptl_renderer_update(self._h, float(seconds), C.byref(tel), C.byref(blk)), "update")
        return bool(tel.value), bool(blk.value)

    def rejit_count(self) -> int:
        return lib().ptl_renderer_rejit_count(self._h)

    def affine_rays(self) -> bool:
        """The current kernel spells o.w = 1 / d.w = 0 in its matrix-times-ray products (every scene matrix and the camera affine)."""
        lib().ptl_renderer_kernel(self._h)  # (the kernel the next draw would use)
        return lib().ptl_renderer_affine_rays(self._h) == 1

    def check_affine(self, width: int = 64, height: int = 36) -> int:
        """The dynamic belt behind the snippet scan: draws the current state with the checking build (FLAG_CHECK_AFFINE) and returns how many ray
        halves met a place where an affine-rays kernel assumes a w with another one; above zero the renderer switches affine rays off and rebuilds."""
        n = C.c_ulonglong(0)
        _check(lib().ptl_renderer_check_affine(self._h, width, height, C.byref(n)), "ptl_renderer_check_affine")
        return int(n.value)

    def rejit_pending(self) -> bool:
        """FLAG_ASYNC_REJIT: a specialised build is being compiled in the background / the un-specialised kernel is in use."""
        return lib().ptl_renderer_rejit_pending(self._h) == 1

    def camera_state(self) -> dict:
        m, sub, pos = (C.c_double * 16)(), C.c_int(), (C.c_double * 3)()
        _check(lib().ptl_renderer_camera_state(self._h, m, C.byref(sub), pos), "camera_state")
        return {"teleport_matrix": np.array(m, np.float64).reshape(4, 4).T.copy(), "in_subspace": bool(sub.value), "position": np.array(pos)}

    def code_object(self) -> bytes:
        k = lib().ptl_renderer_kernel(self._h)
        data, size = C.c_void_p(), C.c_size_t()
        _check(lib().ptl_kernel_code_object(k, C.byref(data), C.byref(size)), "code_object")
        return C.string_at(data, size.value)

    def code_object_sha256(self) -> str:
        """sha256 of the loaded gfx950 code object: what ties a bench line to the PMC passes of the same binary."""
        import hashlib

        return hashlib.sha256(self.code_object()).hexdigest()

    def code_object_note(self, key: str, kernel_prefix: str = "ptl_render") -> int:
        """Kernel metadata of the loaded code object (".vgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size", ".sgpr_count"):
        the largest value over the kernels whose name starts with ``kernel_prefix`` ("" = all)."""
        code = self.code_object()
        return lib().ptl_code_object_note(code, len(code), key.encode(), kernel_prefix.encode())

    # -- drawing ---------------------------------------------------------------------------
    def draw_device(self, frame: Frame, out_rgba8: int = 0, out_rgba32f: int = 0, segments: int = 0, stream: int = 0, timed: bool = False):
        """draw_texture into DEVICE buffers given as integer addresses (e.g. torch data_ptr()).
        Returns the kernel time in ms when ``timed`` (waits for completion), else None."""
        ms = C.c_float()
        rc = lib().ptl_renderer_draw(self._h, C.byref(frame), C.c_void_p(out_rgba8 or None), C.c_void_p(out_rgba32f or None),
                                     C.c_void_p(segments or None), C.c_void_p(stream or None), C.byref(ms) if timed else None)
        _check(rc, "draw_texture")
        return ms.value if timed else None

    def draw(self, width: int, height: int, rgba8: bool = True, rgba32f: bool = False, segments: bool = False, rb_phase: int = 0, rb_stride: int = 1):
        """draw_texture + read back to numpy.  Returns dict(rgba8=HxWx4 u8, rgba32f=HxWx4 f32,
        segments=int, ms=float); H = rows of the shard."""
        frame = Frame(width, height, rb_phase, rb_stride)
        rows = shard_rows(frame)
        if rows < 0:
            raise PortalError("bad frame")
        a8 = np.empty((rows, width, 4), np.uint8) if rgba8 else None
        a32 = np.empty((rows, width, 4), np.float32) if rgba32f else None
        seg, ms = C.c_uint64(0), C.c_float()
        rc = lib().ptl_renderer_draw_to_host(self._h, C.byref(frame), a8.ctypes.data if rgba8 else None, a32.ctypes.data if rgba32f else None,
                                             C.byref(seg) if segments else None, C.byref(ms))
        _check(rc, "draw_texture")
        return {"rgba8": a8, "rgba32f": a32, "segments": seg.value if segments else None, "ms": ms.value}


GROUP_PEER_STORES, GROUP_COPY_GATHER, GROUP_RCCL_GATHER = 0, 1, 2


class FrameGroup:
    """Layer 3: one frame across several GPUs from one process (include/portal_amd.h, csrc/host/multigpu.cpp).
    `devices` may name a device more than once (rehearsal of the N-rank control flow on one GPU)."""

    def __init__(self, scene: Scene, devices, asset_root: Optional[str] = None, flags: int = 0, transport: int = GROUP_PEER_STORES):
        self.scene = scene
        self.devices = list(devices)
        h = C.c_void_p()
        log = C.create_string_buffer(1 << 16)
        arr = (C.c_int * len(self.devices))(*self.devices)
        rc = lib().ptl_frame_group_create(scene._h, arr, len(self.devices), (asset_root or REPO_ROOT).encode(), flags, transport, C.byref(h), log, len(log))
        if rc != 0:
            raise PortalError(f"ptl_frame_group_create failed ({rc}): {_err()}\n{log.value.decode(errors='replace')}")
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ptl_frame_group_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_option(self, name: str, value: float) -> None:
        _check(lib().ptl_frame_group_set_option(self._h, name.encode(), float(value)), "ptl_frame_group_set_option")

    def set_camera(self, look_at, alpha: float, beta: float, r: float) -> None:
        la = (C.c_double * 3)(*look_at)
        _check(lib().ptl_frame_group_set_camera(self._h, la, alpha, beta, r), "ptl_frame_group_set_camera")

    def update(self, seconds: float) -> None:
        _check(lib().ptl_frame_group_update(self._h, float(seconds)), "ptl_frame_group_update")

    def submit(self, width: int, height: int) -> int:
        """Enqueue one frame (trace on every rank + its transfer to devices[0]) without waiting; returns the ticket for `wait`.
        Two frames may be in flight: the transfer of frame n overlaps the trace of frame n + 1."""
        ticket = C.c_int(-1)
        _check(lib().ptl_frame_group_submit(self._h, width, height, C.byref(ticket)), "ptl_frame_group_submit")
        self._size = (width, height)  # (one size for everything in flight: another one is refused while a frame is on its way)
        return ticket.value

    def wait(self, ticket: int) -> dict:
        """The frame of `ticket`, assembled: dict(rgba8, kernel_ms per rank, device_ptr)."""
        ptr = C.c_void_p()
        ms = (C.c_float * len(self.devices))()
        _check(lib().ptl_frame_group_wait(self._h, ticket, C.byref(ptr), ms), "ptl_frame_group_wait")
        width, height = self._size
        img = np.empty((height, width, 4), np.uint8)
        _check(lib().ptl_frame_group_download(self._h, img.ctypes.data_as(C.c_void_p)), "ptl_frame_group_download")
        return {"rgba8": img, "kernel_ms": [float(x) for x in ms], "device_ptr": ptr.value}

    def draw(self, width: int, height: int) -> dict:
        """Returns dict(rgba8 (H, W, 4) uint8, kernel_ms per rank, device_ptr of the frame on devices[0])."""
        ptr = C.c_void_p()
        ms = (C.c_float * len(self.devices))()
        _check(lib().ptl_frame_group_draw(self._h, width, height, C.byref(ptr), ms), "ptl_frame_group_draw")
        img = np.empty((height, width, 4), np.uint8)
        _check(lib().ptl_frame_group_download(self._h, img.ctypes.data_as(C.c_void_p)), "ptl_frame_group_download")
        return {"rgba8": img, "kernel_ms": [float(x) for x in ms], "device_ptr": ptr.value}


class Kernel:
    """Layer 1 of the C ABI (ptl_kernel_*): a hand-written or generated HIP source compiled with
    hiprtc -- the drop-in replacement for macroquad's load_material / set_uniform / set_texture /
    draw (reference src/gui/scene.rs:1132-1143, src/main.rs:1077-1078,1269-1358,1424-1425)."""

    def __init__(self, source: str, uniforms, block_size: int, device: int = 0, defines=()):
        self._keep = [n.encode() for n, _, _ in uniforms]
        descs = (UniformDesc * len(uniforms))(*[UniformDesc(self._keep[i], t, o) for i, (_, t, o) in enumerate(uniforms)])
        defs = (C.c_char_p * len(defines))(*[d.encode() for d in defines])
        h, log = C.c_void_p(), C.create_string_buffer(1 << 16)
        rc = lib().ptl_kernel_compile(device, source.encode("utf-8"), descs, len(uniforms), block_size, defs, len(defines), C.byref(h), log, len(log))
        self.compile_log = log.value.decode("utf-8", "replace")
        if rc != 0:
            raise PortalError(f"ptl_kernel_compile failed ({rc}): {_err()}\n{self.compile_log}")
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ptl_kernel_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_uniform(self, name: str, typ: int, value) -> int:
        if typ == PTL_I32:
            buf = np.array([value], np.int32)
        elif typ == PTL_MAT4:
            buf = np.ascontiguousarray(np.asarray(value, np.float32).T).reshape(16)  # m[row, col] -> column-major
        else:
            buf = np.atleast_1d(np.asarray(value, np.float32))
        return _check(lib().ptl_kernel_set_uniform(self._h, name.encode(), typ, buf.ctypes.data), "set_uniform")

    def set_texture(self, sampler: str, rgba8: np.ndarray) -> int:
        a = np.ascontiguousarray(rgba8, np.uint8)
        return _check(lib().ptl_kernel_set_texture(self._h, sampler.encode(), a.ctypes.data, a.shape[1], a.shape[0]), "set_texture")

    def render(self, width: int, height: int, rgba8: bool = True, rgba32f: bool = False, rb_phase: int = 0, rb_stride: int = 1):
        frame = Frame(width, height, rb_phase, rb_stride)
        rows = shard_rows(frame)
        a8 = np.empty((rows, width, 4), np.uint8) if rgba8 else None
        a32 = np.empty((rows, width, 4), np.float32) if rgba32f else None
        ms = C.c_float()
        rc = lib().ptl_kernel_render_to_host(self._h, C.byref(frame), a8.ctypes.data if rgba8 else None, a32.ctypes.data if rgba32f else None, None, C.byref(ms))
        _check(rc, "ptl_kernel_render_to_host")
        return {"rgba8": a8, "rgba32f": a32, "ms": ms.value}


def device_source(which: str) -> str:
    p = lib().ptl_device_source(which.encode())
    if p is None:
        raise PortalError(f"no device source `{which}`")
    return p.decode("utf-8")


def deinterleave_rows(shard: np.ndarray, frame: Frame, full: np.ndarray) -> None:
    _check(lib().ptl_deinterleave_rows(shard.ctypes.data, C.byref(frame), full.ctypes.data), "deinterleave_rows")


def average_images_device(frame_ptrs, out_ptr: int, width: int, height: int, device: int = 0, stream: int = 0, timed: bool = False):
    """average_images (src/main.rs:645-722) on DEVICE buffers given as integer addresses."""
    arr = (C.c_void_p * len(frame_ptrs))(*frame_ptrs)
    ms = C.c_float()
    _check(lib().ptl_average_images(device, arr, len(frame_ptrs), C.c_void_p(out_ptr), width, height, C.c_void_p(stream or None), C.byref(ms) if timed else None),
           "average_images")
    return ms.value if timed else None


def device_alloc(nbytes: int, device: int = 0) -> int:
    """A whole device allocation (hipMalloc), as an integer address.  What `ipc_export` needs."""
    p = C.c_void_p()
    _check(lib().ptl_device_alloc(device, nbytes, C.byref(p)), "device_alloc")
    return int(p.value)


def device_free(ptr: int) -> None:
    _check(lib().ptl_device_free(C.c_void_p(ptr)), "device_free")


def device_download(ptr: int, nbytes: int, stream: int = 0) -> np.ndarray:
    out = np.empty(nbytes, dtype=np.uint8)
    _check(lib().ptl_device_download(out.ctypes.data, C.c_void_p(ptr), nbytes, C.c_void_p(stream or None)), "device_download")
    return out


def device_copy2d_async(dst: int, dst_pitch: int, src: int, src_pitch: int, width_bytes: int, rows: int, stream: int = 0) -> None:
    """Strided device-to-device copy on `stream` (hipMemcpy2DAsync): one packed shard into an interleaved frame, possibly on another GPU."""
    _check(lib().ptl_device_copy2d_async(C.c_void_p(dst), dst_pitch, C.c_void_p(src), src_pitch, width_bytes, rows, C.c_void_p(stream or None)), "ptl_device_copy2d_async")


IPC_HANDLE_BYTES = 64


def ipc_export(ptr: int) -> bytes:
    """Handle of a `device_alloc` buffer for the other processes of the node (ptl_ipc_export)."""
    buf = C.create_string_buffer(IPC_HANDLE_BYTES)
    _check(lib().ptl_ipc_export(C.c_void_p(ptr), buf), "ipc_export")
    return buf.raw


def ipc_open(handle: bytes, device: int = 0) -> int:
    """Map another process's exported buffer into this process (on `device`); returns the local address."""
    if len(handle) != IPC_HANDLE_BYTES:
        raise PortalError("an IPC handle is 64 bytes")
    p = C.c_void_p()
    _check(lib().ptl_ipc_open(device, handle, C.byref(p)), "ipc_open")
    return int(p.value)


def ipc_close(ptr: int) -> None:
    _check(lib().ptl_ipc_close(C.c_void_p(ptr)), "ipc_close")


def png_write(path: str, rgba8: np.ndarray) -> None:
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    _check(lib().ptl_png_write(path.encode(), a.ctypes.data, a.shape[1], a.shape[0]), "png_write")


def png_read(path: str) -> np.ndarray:
    p, w, h = C.c_void_p(), C.c_int(), C.c_int()
    _check(lib().ptl_png_read(path.encode(), C.byref(p), C.byref(w), C.byref(h)), "png_read")
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
    finally:
        lib().ptl_free(p)


def ron_format(text: str) -> str:
    """Parse RON and write it back in the layout of the reference's scene files (ron 0.10 pretty printer)."""
    p = lib().ptl_ron_format(text.encode("utf-8"))
    if not p:
        raise PortalError(_err())
    try:
        return C.string_at(p).decode("utf-8")
    finally:
        lib().ptl_free(p)


def bound_glsl(body: str, out_functions=()):
    """(text, n): an intersection-material snippet with the caller's distance bound in the `nearer` blocks it recognises (n of them)."""
    n = C.c_int()
    p = lib().ptl_bound_glsl(body.encode("utf-8"), ",".join(out_functions).encode(), C.byref(n))
    if not p:
        raise PortalError(lib().ptl_last_error().decode())
    try:
        return C.string_at(p).decode("utf-8"), n.value
    finally:
        lib().ptl_free(p)


def dmath(op: str, a, b=None, c=None) -> np.ndarray:
    """One binary64 primitive of the host's matrix arithmetic (ptl_dmath): 16 doubles, column-major, as a flat array."""
    arr = lambda v: (C.c_double * len(v))(*[float(x) for x in v]) if v is not None else None
    out = (C.c_double * 16)()
    _check(lib().ptl_dmath(op.encode(), arr(a), arr(b), arr(c), out), "ptl_dmath")
    return np.array(out, np.float64)


def snippets_keep_rays_affine(code: str):
    """(ok, why): does this GLSL text keep every ray's w at 1 (origin) / 0 (direction)?  (the scan behind the affine-rays builds)"""
    why = C.create_string_buffer(512)
    rc = lib().ptl_snippets_keep_rays_affine(code.encode("utf-8"), why, len(why))
    if rc < 0:
        raise PortalError(_err())
    return rc == 1, why.value.decode("utf-8", "replace")


def translate_glsl(code: str) -> str:
    p = lib().ptl_translate_glsl(code.encode("utf-8"))
    if not p:
        raise PortalError(_err())
    try:
        return C.string_at(p).decode("utf-8")
    finally:
        lib().ptl_free(p)


def translate_library_glsl(code: str) -> str:
    p = lib().ptl_translate_library_glsl(code.encode("utf-8"))
    if not p:
        raise PortalError(_err())
    try:
        return C.string_at(p).decode("utf-8")
    finally:
        lib().ptl_free(p)


def hoist_glsl(code: str, uniforms: dict, out_functions=(), body_only: bool = True, params=()):
    """glsl_hoist.h on one snippet (for tests): `uniforms` maps name -> GLSL type.  Returns (rewritten GLSL, prologue text with one
    "// member: type name" line per created member in front)."""
    L = lib()
    L.ptl_hoist_glsl.restype = C.c_void_p
    L.ptl_hoist_glsl.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    prologue = C.c_void_p()
    p = L.ptl_hoist_glsl(code.encode("utf-8"), ";".join(f"{t} {n}" for n, t in uniforms.items()).encode(), ";".join(out_functions).encode(),
                         1 if body_only else 0, ";".join(params).encode(), C.byref(prologue))
    try:
        return C.string_at(p).decode("utf-8"), C.string_at(prologue.value).decode("utf-8")
    finally:
        L.ptl_free(p)
        L.ptl_free(prologue.value)


def formula_eval(text: str, variables: Optional[dict] = None, time: float = 0.0) -> Optional[float]:
    variables = variables or {}
    names = (C.c_char_p * len(variables))(*[k.encode() for k in variables])
    vals = (C.c_double * len(variables))(*[float(v) for v in variables.values()])
    out = C.c_double()
    rc = lib().ptl_formula_eval(text.encode(), names, vals, len(variables), time, C.byref(out))
    return out.value if rc == 0 else None


def scene_path(name: str) -> str:
    """`monoportal` -> <repo>/scenes/monoportal.ron (the reference resolves names through its
    built-in registry, src/gui/scenes.rs:589-603)."""
    if os.path.exists(name):
        return name
    cand = os.path.join(REPO_ROOT, "scenes", name if name.endswith(".ron") else name + ".ron")
    if os.path.exists(cand):
        return cand
    raise PortalError(f"Unknown scene `{name}`")
