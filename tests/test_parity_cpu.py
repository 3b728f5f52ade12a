"""CPU parity: the product's generated kernel (compiled for the host, oracle/host_build) against
the independent numpy oracle and the committed golden frames -- bit-exact.

This checks everything of the product except the gfx950 compiler + hardware leg (that leg is
tests/test_gpu_parity.py): .ron loading, uniform/matrix evaluation, scene -> source generation,
the GLSL -> C++ translation of scene snippets, the device prelude and the trace template.
"""
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def parse_case(path):
    base = os.path.basename(path)[: -len(".npz")]
    scene, dims, depth, aa = base.rsplit("_", 3)
    w, h = dims.split("x")
    return scene, int(w), int(h), int(depth[1:]), int(aa[2:])


def bits_equal(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def host_render(pa, scene_name, w, h, depth, aa, flags=0, options=()):
    from oracle import host_build as hb

    scene = pa.Scene.from_file(pa.scene_path(scene_name))
    r = pa.SceneRenderer(scene, device=-1, flags=flags)
    r.set_option("render_depth", depth)
    r.set_option("aa_count", aa)
    for k, v in options:
        r.set_option(k, v)
    hk = hb.host_kernel_for(r, scene, w, h, flags=flags | pa.FLAG_COUNT_SEGMENTS, count_segments=True)
    return hk.render(w, h)


def test_golden_files_present():
    assert len(GOLDEN) == 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_product_host_build_matches_golden(pa, path):
    scene, w, h, depth, aa = parse_case(path)
    g = np.load(path)
    out = host_render(pa, scene, w, h, depth, aa)
    want = g["rgba32f_bits"].view(np.float32)
    ok = bits_equal(out["rgba32f"], want)
    assert ok.all(), f"{int((~ok).any(axis=2).sum())} of {w*h} pixels differ from the golden frame"
    assert np.array_equal(out["rgba8"], g["rgba8"])
    assert out["segments"] == int(g["segments"].sum())


@pytest.mark.parametrize("path", GOLDEN[:4], ids=[os.path.basename(p) for p in GOLDEN[:4]])
def test_oracle_reproduces_golden(path):
    """The golden frames are this oracle's own output (make_golden.py): guards against drift."""
    from oracle.portal_oracle import Oracle

    scene, w, h, depth, aa = parse_case(path)
    g = np.load(path)
    o = Oracle(os.path.join(ROOT, "scenes", scene + ".ron"))
    o.options.update(render_depth=depth, aa_count=aa)
    rows = (h // 2 - 6, h // 2 + 6)  # a 12-row band keeps the CPU suite short; the full frame is checked via the product above
    out = o.render(w, h, rows=rows)
    assert bits_equal(out["rgba32f"], g["rgba32f_bits"].view(np.float32)[rows[0]:rows[1]]).all()
    assert np.array_equal(out["segments"], g["segments"][rows[0]:rows[1]])


@pytest.mark.parametrize("flags_name", ["FLAG_SPECIALIZE_INTS", "FLAG_SPECIALIZE_ALL"])
@pytest.mark.parametrize("scene", ["portal_in_portal", "triple_portal"])
def test_jit_specialisation_does_not_change_a_single_bit(pa, scene, flags_name):
    base = host_render(pa, scene, 64, 36, 40, 1)
    spec = host_render(pa, scene, 64, 36, 40, 1, flags=getattr(pa, flags_name))
    assert bits_equal(base["rgba32f"], spec["rgba32f"]).all() and base["segments"] == spec["segments"]


def test_panini_projection_matches_oracle(pa):
    """north_star names the Panini projection (src/frag.glsl:305-342): product == oracle, bit-exact."""
    from oracle.portal_oracle import Oracle

    w, h = 64, 36
    opts = [("use_panini_projection", 1), ("panini_param", 1.0), ("view_angle", np.radians(140.0))]
    got = host_render(pa, "portal_in_portal", w, h, 40, 1, options=opts)
    o = Oracle(os.path.join(ROOT, "scenes", "portal_in_portal.ron"))
    o.options.update(render_depth=40, use_panini=True, panini_param=1.0, view_angle=np.radians(140.0))
    want = o.render(w, h)
    assert bits_equal(got["rgba32f"], want["rgba32f"]).all()
    plain = host_render(pa, "portal_in_portal", w, h, 40, 1)
    assert not np.array_equal(plain["rgba8"], got["rgba8"])  # the projection really is in effect


def test_antialiasing_is_the_mean_of_r2_offset_samples(pa):
    """frag.glsl:515-526: result = sqrt(mean over a of get_color(uv + R2(a) * pixel * 2))."""
    one = host_render(pa, "monoportal", 48, 27, 20, 1)
    four = host_render(pa, "monoportal", 48, 27, 20, 4)
    assert four["segments"] > 3 * one["segments"]
    from oracle.portal_oracle import Oracle

    o = Oracle(os.path.join(ROOT, "scenes", "monoportal.ron"))
    o.options.update(render_depth=20, aa_count=4)
    want = o.render(48, 27, rows=(10, 16))
    assert bits_equal(four["rgba32f"][10:16], want["rgba32f"]).all()


def test_window_rendering_equals_full_frame(pa):
    from oracle import host_build as hb

    scene = pa.Scene.from_file(pa.scene_path("basics"))
    r = pa.SceneRenderer(scene, device=-1)
    r.set_option("render_depth", 4)
    hk = hb.host_kernel_for(r, scene, 80, 60)
    full = hk.render(80, 60)["rgba32f"]
    win = hk.render(80, 60, rows=(13, 29), cols=(7, 61))["rgba32f"]
    assert np.array_equal(full[13:29, 7:61].view(np.uint32), win.view(np.uint32))
    picked = hk.render(80, 60, rows=[59, 0, 17])["rgba32f"]
    assert np.array_equal(picked.view(np.uint32), full[[59, 0, 17]].view(np.uint32))


# ---- camera teleportation ---------------------------------------------------
def _portal_segments(pa, scene_name, n_per_portal=5, seed=3):
    """Segments that cross (or just miss) each portal of the scene: from local z=+0.4 to z=-0.4."""
    vals = pa.Scene.from_file(pa.scene_path(scene_name)).uniform_values()
    rng = np.random.default_rng(seed)
    out = []
    for tname in sorted(k for k in vals if k.endswith("_mat_teleport")):
        a_name = tname[: -len("_mat_teleport")].split("_to_")[0]
        A = np.asarray(vals[a_name + "_mat"], np.float64)
        T = np.asarray(vals[tname], np.float64)
        for _ in range(n_per_portal):
            uv = rng.uniform(-0.7, 0.7, 2)
            a = (A @ np.array([uv[0], uv[1], 0.4, 1.0]))[:3]
            b = (A @ np.array([uv[0] + 0.1, uv[1] - 0.05, -0.4, 1.0]))[:3]
            out.append((a, b, T))
    return out


@pytest.mark.parametrize("scene_name", ["basics", "monoportal", "triple_portal"])
def test_teleport_external_ray_matches_oracle_and_the_portal_matrix(pa, scene_name):
    """teleport_external_ray (src/frag.glsl:209-257, src/main.rs:1361-1409): product == oracle bit for
    bit; and when a segment does cross a portal its end point b lands at T*b, T = B*A^-1 (known answer)."""
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    scene = pa.Scene.from_file(pa.scene_path(scene_name))
    r = pa.SceneRenderer(scene, device=-1)
    hk = hb.host_kernel_for(r, scene, 0, 0)
    hk.set_uniform("teleport_light_u", 1)  # src/main.rs:1367
    o = Oracle(pa.scene_path(scene_name))
    crossed = 0
    for a, b, T in _portal_segments(pa, scene_name):
        got, want = hk.teleport_external_ray(a, b), o.teleport_external_ray(a, b)
        assert got[1:] == want[1:] and (got[0] is None) == (want[0] is None)
        if got[0] is not None:
            assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
            expect = (T @ np.array([*b, 1.0]))[:3]
            if np.allclose(got[0], expect, atol=2e-4):
                crossed += 1
    assert crossed >= 3


@pytest.mark.parametrize("scene_name,stage", [("monoportal", "rotate portal"), ("portal_in_portal", "picture 2"), ("triple_portal", None)])
def test_staged_scene_with_stage_camera_matches_oracle(pa, scene_name, stage):
    """`render-frame --stage NAME`: overrides + the camera the stage selects."""
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    path = pa.scene_path(scene_name)
    scene = pa.Scene.from_file(path)
    stage = stage or scene.stages()[-1]
    cam_name = scene.init_stage(stage)
    r = pa.SceneRenderer(scene, device=-1)
    r.set_option("render_depth", 30)
    if cam_name:
        r.use_camera(cam_name)
    got = hb.host_kernel_for(r, scene, 56, 32).render(56, 32)
    o = Oracle(path)
    cam_idx = o.scene.init_stage(stage)
    if cam_idx >= 0:
        o.camera = o.scene.camera_settings(cam_idx)
    o.options["render_depth"] = 30
    want = o.render(56, 32)
    assert bits_equal(got["rgba32f"], want["rgba32f"]).all()
    plain = host_render(pa, scene_name, 56, 32, 30, 1)
    assert not np.array_equal(plain["rgba8"], got["rgba8"])  # the stage really changed the picture


@pytest.mark.parametrize("mode,options,overrides", [
    ("360", [("use_360_camera", 1)], {"_use_360_camera": np.int32(1)}),
    ("180", [("use_180_camera", 1)], {"_use_180_camera": np.int32(1)}),
    ("depth", [("draw_depth_map", 1), ("depth_map_min", 1.0), ("depth_map_max", 7.5)],
     {"_draw_depth_map": np.int32(1), "_depth_map_min": np.float32(1.0), "_depth_map_max": np.float32(7.5)}),
    ("side_by_side", [("draw_side_by_side", 1)], {"_draw_side_by_side": np.int32(1)}),
])
def test_other_kernel_modes_match_oracle(pa, mode, options, overrides):
    """360 / 180 equirect cameras (frag.glsl:413-448), depth-map colouring (:80-104,456-462), side-by-side
    (:479-499; both eyes are the identity matrix offline, like the reference before teleport_eye_matrices)."""
    from oracle.portal_oracle import Oracle

    w, h = 64, 36
    got = host_render(pa, "monoportal", w, h, 20, 1, options=options)
    o = Oracle(os.path.join(ROOT, "scenes", "monoportal.ron"))
    o.options["render_depth"] = 20
    o.overrides = overrides
    want = o.render(w, h)
    assert bits_equal(got["rgba32f"], want["rgba32f"]).all()
    assert got["segments"] == int(want["segments"].sum())
    assert not np.array_equal(got["rgba8"], host_render(pa, "monoportal", w, h, 20, 1)["rgba8"])


@pytest.mark.parametrize("colorful", [0, 1])
def test_anaglyph_mode_matches_oracle(pa, colorful):
    """Anaglyph stereo (frag.glsl:343-406,467-473), compiled in with FLAG_ANAGLYPH (the reference's `disable_anaglyph = false`):
    two traces per sample, combined in linear light with ghosting compensation; mode 1 keeps the right eye's hue.  Offline
    both eye matrices are the identity (no GPU ray query on this box; the -m gpu stereo test uses teleported eyes).  Without
    the flag the uniform exists but the mode does not: the frame is the plain one (scene.rs:1072-1080)."""
    from oracle.portal_oracle import Oracle

    w, h = 48, 27
    options = [("draw_anaglyph", 1), ("anaglyph_mode", colorful), ("anaglyph_p", 0.31), ("anaglyph_q", 0.05)]
    got = host_render(pa, "monoportal", w, h, 12, 1, flags=pa.FLAG_ANAGLYPH, options=options)
    o = Oracle(os.path.join(ROOT, "scenes", "monoportal.ron"))
    o.options["render_depth"] = 12
    o.anaglyph_compiled_in = True
    o.overrides = {"_draw_anaglyph": np.int32(1), "_anaglyph_mode": np.int32(colorful), "_anaglyph_p": np.float32(0.31), "_anaglyph_q": np.float32(0.05)}
    want = o.render(w, h)
    assert bits_equal(got["rgba32f"], want["rgba32f"]).all()
    assert got["segments"] == int(want["segments"].sum())
    px = got["rgba32f"][h // 2, w // 2]
    assert (px[1] == px[2]) == (colorful == 0)                   # grayscale mode: G == B == the cyan channel
    stripped = host_render(pa, "monoportal", w, h, 12, 1, options=options)
    assert np.array_equal(stripped["rgba8"], host_render(pa, "monoportal", w, h, 12, 1)["rgba8"])


@pytest.mark.parametrize("in_subspace", [0, 1])
def test_kitchen_sink_scene_matches_oracle(pa, tmp_path, in_subspace):
    """A synthetic scene with what the five BASELINE scenes lack together (tests/synthetic.py::kitchen_sink_scene): GLSL Complex
    object, Refract + Reflect materials, DebugMatrix gizmo, a subspace-only object, a skybox texture, a formula-driven matrix.
    Product (host build of the generated source) == oracle, bit for bit, from the normal space and from inside the subspace."""
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle
    from tests import synthetic

    synthetic.write_sky_texture(pa, str(tmp_path))
    text = synthetic.kitchen_sink_scene()
    path = tmp_path / "sink.ron"
    path.write_text(text)
    w, h = 64, 36
    scene = pa.Scene.from_file(str(path))
    r = pa.SceneRenderer(scene, device=-1, asset_root=str(tmp_path))
    r.set_option("render_depth", 12)
    r.set_option("in_subspace", in_subspace)
    got = hb.host_kernel_for(r, scene, w, h, asset_root=str(tmp_path)).render(w, h)
    o = Oracle(str(path), asset_root=str(tmp_path))
    o.options["render_depth"] = 12
    o.camera = {"in_subspace": bool(in_subspace)}
    want = o.render(w, h)
    assert bits_equal(got["rgba32f"], want["rgba32f"]).all()
    assert len(np.unique(got["rgba8"].reshape(-1, 4), axis=0)) > 50       # a real picture, not a flat fill


CONTRACT1 = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "contract1", "*.npz")))


@pytest.mark.parametrize("path", CONTRACT1[:4], ids=[os.path.basename(p) for p in CONTRACT1[:4]])
def test_exact_cr_build_and_oracle_reproduce_the_contract_1_goldens(pa, path):
    """FLAG_EXACT_CR / `--exact-cr` keeps the numerics contract of rounds 1-2 (IEEE `/` and sqrt on every input): the host build of
    that kernel source and the numpy oracle switched to contract 1 both give the golden frames committed in round 1, bit for bit --
    and the default (contract 2) differs from them in the last bits only."""
    from oracle import glsl_math as M
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    base = os.path.basename(path)[: -len(".npz")]
    scene_name, dims, depth, aa = base.rsplit("_", 3)
    w, h = (int(x) for x in dims.split("x"))
    depth, aa = int(depth[1:]), int(aa[2:])
    g = np.load(path)
    scene = pa.Scene.from_file(pa.scene_path(scene_name))
    r = pa.SceneRenderer(scene, device=-1, flags=pa.FLAG_EXACT_CR)
    r.set_option("render_depth", depth)
    r.set_option("aa_count", aa)
    got = hb.host_kernel_for(r, scene, w, h, flags=pa.FLAG_EXACT_CR).render(w, h)
    assert np.array_equal(got["rgba32f"].view(np.uint32), g["rgba32f_bits"])
    previous = M.set_contract(1)
    try:
        o = Oracle(pa.scene_path(scene_name))
        o.options.update(render_depth=depth, aa_count=aa)
        want = o.render(w, h)
    finally:
        M.set_contract(previous)
    assert np.array_equal(want["rgba32f"].view(np.uint32), g["rgba32f_bits"]) and np.array_equal(want["rgba8"], g["rgba8"])
    now = np.load(os.path.join(os.path.dirname(os.path.dirname(path)), os.path.basename(path)))["rgba32f_bits"].view(np.float32)
    old = g["rgba32f_bits"].view(np.float32)
    moved = np.abs(now - old) > 1e-5
    assert moved.any(axis=2).sum() <= 0.005 * w * h  # pixels on an edge whose path a last bit decides (the Moebius strip's Newton search: 5 of 2 304)
