"""The reference's whole scene corpus (84 .ron files) through the product and the oracle.

Only runs where the reference checkout is mounted (/root/reference: the build container); the
GPU box has no copy, there the five BASELINE scenes under scenes/ are what is tested.
For every scene: load -> generate the kernel source -> compile it for the host -> render a small
frame -> compare bit for bit with the independent numpy oracle.  A sample is also compiled with
hiprtc for gfx950 (no GPU needed).
"""
import glob
import os

import numpy as np
import pytest

CORPUS = "/root/reference/scenes"
pytestmark = pytest.mark.skipif(not os.path.isdir(CORPUS), reason="reference scene corpus not mounted")

OUT_OF_SCOPE = set()  # nothing: all 82 non-empty scene files go through


@pytest.fixture(autouse=True)
def _scratch_build_dir(tmp_path_factory, monkeypatch):
    """82 one-off host builds: keep them out of oracle/_build (which travels to the GPU box)."""
    from oracle import host_build as hb

    monkeypatch.setattr(hb, "BUILD_DIR", str(tmp_path_factory.getbasetemp() / "corpus_build"))


def scene_files():
    return [f for f in sorted(glob.glob(os.path.join(CORPUS, "*.ron"))) if os.path.getsize(f) > 0 and os.path.basename(f)[:-4] not in OUT_OF_SCOPE]


def masked_build_sample():
    """Scenes that also go through the Bool / Int baked build (zero patterns of the run-time matrices compiled in, masked products): every
    fifth file plus the ones whose snippets multiply by matrices in every way the corpus knows (written-out products, chains, Matrix::Camera)."""
    named = {"portal_in_portal", "portal_in_portal_plus_ultra", "triple_portal", "mobius_monoportal", "monoportal", "matryoshka", "trefoil_knot_portal", "cone_portal"}
    return [f for k, f in enumerate(scene_files()) if k % 5 == 0 or os.path.basename(f)[:-4] in named]


def _parity_cases():
    return [(f, "plain") for f in scene_files()] + [(f, "ints") for f in masked_build_sample()]


@pytest.mark.parametrize("path,build", _parity_cases(), ids=[os.path.basename(f)[:-4] + ("" if b == "plain" else "-ints") for f, b in _parity_cases()])
def test_corpus_scene_product_equals_oracle(pa, path, build):
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    w, h, depth = 24, 14, 12
    scene = pa.Scene.from_file(path)
    source = scene.generate_source(0 if build == "plain" else pa.FLAG_SPECIALIZE_INTS)
    layout, size = scene.uniform_layout()
    hk = hb.HostKernel(source, layout, size, opt="-O0")  # tiny frames: compile time dominates
    o = Oracle(path, asset_root="/root/reference")
    o.options["render_depth"] = depth
    # uniform values: the product's (scene + builtins) go into the host kernel; the oracle computes its own
    from oracle.scene_eval import builtin_uniforms

    from oracle.scene_eval import camera_matrix

    c = o.scene.cam  # Matrix::Camera evaluates to the drawing camera's matrix (the renderer sends it; here the test does)
    scene.set_camera_matrix(np.array(camera_matrix(c["look_at"], c["alpha"], c["beta"], c["r"])).T)
    vals = scene.uniform_values()
    for name, typ, _ in layout:
        if name in vals:
            hk.set_uniform(name, vals[name])
    for name, v in builtin_uniforms(o.scene, w, h, render_depth=depth).items():
        a = np.asarray(v)
        hk.set_uniform(name, a.reshape(4, 4).T if a.shape == (16,) else a)
    from PIL import Image

    for tex_name, rel in scene.textures().items():
        full = os.path.join("/root/reference", rel)
        if os.path.exists(full):
            hk.set_texture(tex_name + "_tex", np.array(Image.open(full).convert("RGBA")))
    got = hk.render(w, h, threads=2)
    want = o.render(w, h)
    a, b = got["rgba32f"], want["rgba32f"]
    bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    assert not bad.any(), f"{int(bad.any(axis=2).sum())} of {w*h} pixels differ"


def test_builtin_uniform_values_used_above_are_the_products(pa):
    """The corpus test feeds oracle-computed builtins into the product kernel; make sure they equal
    what the product's SceneRenderer would upload (so that shortcut hides nothing)."""
    from oracle.scene_eval import OracleScene, builtin_uniforms

    for name in ("cone", "room", "zeno_portal"):
        path = os.path.join(CORPUS, name + ".ron")
        r = pa.SceneRenderer(pa.Scene.from_file(path), device=-1, asset_root="/root/reference")
        r.set_option("render_depth", 12)
        for k, v in builtin_uniforms(OracleScene(path), 24, 14, render_depth=12).items():
            g = r.uniform_value(k, 24, 14)
            g = np.asarray(g).T.reshape(16) if np.asarray(g).shape == (4, 4) else np.asarray(g).reshape(-1)
            assert np.array_equal(g.astype(np.float32), np.asarray(v, np.float32).reshape(-1)), (name, k)


@pytest.mark.parametrize("path", scene_files(), ids=[os.path.basename(f)[:-4] for f in scene_files()])
def test_corpus_scene_compiles_for_gfx950(pa, path, tmp_path, monkeypatch):
    """hiprtc, no GPU, every scene of the corpus, with clip-constant specialisation (the video pipeline's build): skyboxes,
    subspaces, DebugMatrix, scene-defined inverse(), video samplers, `const in`, Trefoil ... all become gfx950 code objects."""
    monkeypatch.setenv("PTL_CACHE_DIR", str(tmp_path))  # keep 82 code objects out of the repository's cache
    scene = pa.Scene.from_file(path)
    r = pa.SceneRenderer(scene, device=-1, asset_root="/root/reference", flags=pa.FLAG_SPECIALIZE_STATIC)
    assert r.code_object()[:4] == b"\x7fELF"
    # ... and with only Bool / Int uniforms baked: every matrix stays a run-time value and the generator rewrites the products with the
    # masked ones by their shape (`X_mat * <operand>`, `transform(X_mat, ..)`) -- whatever the scene's authors wrote has to survive that
    if path in masked_build_sample():
        r = pa.SceneRenderer(pa.Scene.from_file(path), device=-1, asset_root="/root/reference", flags=pa.FLAG_SPECIALIZE_INTS)
        assert r.code_object()[:4] == b"\x7fELF"


def animated_scene_files():
    out = []
    for f in scene_files():
        with open(f, encoding="utf-8") as fh:
            if "animation_stage:" in fh.read():
                out.append(f)
    return out


@pytest.mark.parametrize("path", animated_scene_files(), ids=[os.path.basename(f)[:-4] for f in animated_scene_files()])
def test_corpus_real_animations_product_equals_oracle(pa, path):
    """Every clip of every scene that has real animations (28 scenes, ~480 clips): the per-frame host step of the video
    pipeline gives the oracle's formula time, uniform values and camera at two times per clip.  Host logic only (no kernel
    build: a renderer is not needed to evaluate a scene), camera compared through Scene::update's interpolated camera."""
    from oracle.scene_eval import OracleScene

    ps, osc = pa.Scene.from_file(path), OracleScene(path)
    clips = ps.animations()
    assert [c[0] for c in clips] == [a["name"] for a in osc.animations]
    import tests.test_host_logic as hl

    for clip, duration in clips:
        ps.init_animation(clip)
        osc.init_animation(clip)
        for frac in (0.0, 0.625):
            t = frac * duration
            got, want = ps.update(t), osc.update(t)
            assert (got["time"], got["total_time"]) == (osc.time, osc.total_time), (clip, t)
            assert (got["camera"] is None) == (want is None), (clip, t)
            if want is not None:
                for k in ("look_at", "alpha", "beta", "r", "free_movement", "in_subspace", "override_matrix"):
                    assert got["camera"][k] == want[k], (clip, t, k)
                assert np.array_equal(got["camera"]["matrix"], np.array(want["teleport_matrix"]).T), (clip, t)
            hl._same_uniforms(ps.uniform_values(), osc.scene_uniform_values(), (clip, t))


def test_zero_patterns_compiled_into_clip_kernels_hold_through_every_corpus_clip(pa):
    """KernelOptions::mask_zero_elements in the clip-constant build: the zero pattern of an animated matrix is taken over 33 moments of the
    clip (Scene::update on a copy of the scene).  A pattern that breaks in the middle of a clip is not a wrong pixel -- the renderer checks
    every upload and rebuilds -- but it is a rebuild in the middle of a clip; so: every clip of every animated scene of the corpus (471 clips,
    ~3 900 masked matrices), at 13 other moments, never leaves the patterns its kernel was generated with."""
    import re

    clips = masked = units = 0
    for path in animated_scene_files():
        ps = pa.Scene.from_file(path)
        for clip, duration in ps.animations():
            ps.init_animation(clip)
            ps.update(0.0)
            src = ps.generate_source(pa.FLAG_SPECIALIZE_STATIC)
            masks = {n: int(v, 16) for n, v in re.findall(r"#define PTL_MASK_(\w+) (0x[0-9a-f]{4})u", src)}
            # ... and the elements compiled in as +1 / -1 (round 4: PTL_UNIT_BITS(ones, negs) behind the zero pattern)
            unit = {n: (int(a, 16), int(b, 16)) for n, a, b in re.findall(r"#define PTL_MASK_(\w+) 0x[0-9a-f]{4}u \| PTL_UNIT_BITS\((0x[0-9a-f]{4}), (0x[0-9a-f]{4})\)", src)}
            clips += 1
            masked += len(masks)
            units += sum(bin(a).count("1") + bin(b).count("1") for a, b in unit.values())
            for k in range(13):
                ps.update(duration * (k + 0.37) / 13.0)
                vals = ps.uniform_values()
                for name, mask in masks.items():
                    a = np.asarray(vals[name], np.float32).T.reshape(-1)  # column-major like the uniform block: bit 4 * column + row
                    assert not any(a[e] != 0 and not (mask >> e) & 1 for e in range(16)), (os.path.basename(path), clip, name, k)
                    ones, negs = unit.get(name, (0, 0))
                    assert not any(((ones >> e) & 1 and a[e] != 1) or ((negs >> e) & 1 and a[e] != -1) for e in range(16)), (os.path.basename(path), clip, name, k)
    assert clips > 400 and masked > 3000 and units > 3000


@pytest.mark.parametrize("path", scene_files(), ids=[os.path.basename(f)[:-4] for f in scene_files()])
def test_corpus_scene_writer_reproduces_the_file(pa, path):
    """All 82 scene files were written by the reference's RON writer: parse -> write must give each back byte for byte."""
    text = open(path, encoding="utf-8").read()
    assert pa.ron_format(text) == text
    assert pa.Scene.from_file(path).to_ron() == text


def test_trefoil_text_form_reference_vector(pa):
    """The reference's own unit test (src/gui/uniform.rs:258-264): "1a 2a G,1b 3b B,2a 1a S" decodes and encodes back to itself;
    on trefoil.ron it lands in the 18 packed `ts_<i>_trefoil_u` ints (value + enabled * 10000 + colour * 1000) and in the written file."""
    s = pa.Scene.from_file(os.path.join(CORPUS, "trefoil.ron"))
    text = "1a 2a G,1b 3b B,2a 1a S"
    s.set_trefoil("trefoil", text)
    assert s.get_trefoil("trefoil") == text
    v = s.uniform_values()
    assert int(v["ts_0_trefoil_u"]) == 3 + 10000 + 1000 and int(v["ts_1_trefoil_u"]) == 7 + 10000 + 2000 and int(v["ts_3_trefoil_u"]) == 0 + 10000 + 5000
    assert int(v["ts_2_trefoil_u"]) == 0 and int(v["ts_17_trefoil_u"]) == 0
    assert "TrefoilSpecial((((true, 3, 1), (true, 7, 2), (false, 0, 0), (true, 0, 5), (false, 0, 0)," in s.to_ron()
    for bad in ("1a 2a", "1a 2a Q", "7a 1a G", "1a  2a G"):
        with pytest.raises(pa.PortalError):
            s.set_trefoil("trefoil", bad)
    with pytest.raises(pa.PortalError):
        s.set_trefoil("use_origin_room", text)
