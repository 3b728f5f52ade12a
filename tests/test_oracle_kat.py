"""Known-answer tests: values derived BY HAND from the reference's formulas.

These are what pins the oracle (the reference has no test that touches the trace arithmetic).
Each end-to-end case is run through BOTH the numpy oracle and the product's kernel source
compiled for the host, so the product is pinned by the same hand-derived numbers.
"""
import math
import os

import numpy as np
import pytest

from tests import synthetic

F = np.float32


def natives(**u):
    from oracle.portal_oracle import Natives

    base = dict(_angle_color_disable=np.int32(0), _grid_disable=np.int32(0), _offset_after_material=F(2.5e-5), _black_border_disable=np.int32(0))
    base.update(u)
    return Natives(base)


def ident_mat():
    from oracle.glsl_values import Mat, Vec

    return Mat([Vec([F(1 if i == j else 0) for i in range(4)]) for j in range(4)])


def test_plane_intersect_identity_plane():
    """(1) src/library.glsl:138-162: ray (0,0,-2) -> +z against the identity plane: t=2, u=v=0, n=(0,0,-1)."""
    from oracle.portal_oracle import Ray, vec

    n = natives()
    r = Ray(vec(0.0, 0.0, -2.0, 1.0), vec(0.0, 0.0, 1.0, 0.0), 1.0, False)
    hit = n.plane_intersect(r, ident_mat(), n.get_normal(ident_mat()))
    assert bool(hit.f["hit"]) and hit.f["t"] == 2 and hit.f["u"] == 0 and hit.f["v"] == 0
    assert [float(x) for x in hit.f["n"].c] == [0.0, 0.0, -1.0]
    # a ray pointing away misses (t < 0 -> intersection_none with t = 1e10)
    away = n.plane_intersect(Ray(vec(0.0, 0.0, -2.0, 1.0), vec(0.0, 0.0, -1.0, 0.0), 1.0, False), ident_mat(), n.get_normal(ident_mat()))
    assert not bool(away.f["hit"]) and away.f["t"] == F(1e10)
    # oblique ray: hit point (u, v) = o.xy + d.xy * t
    d = np.array([0.6, 0.0, 0.8], np.float32)
    ob = n.plane_intersect(Ray(vec(1.0, 2.0, -4.0, 1.0), vec(*d, 0.0), 1.0, False), ident_mat(), n.get_normal(ident_mat()))
    assert ob.f["t"] == pytest.approx(5.0, rel=1e-6) and ob.f["u"] == pytest.approx(4.0, rel=1e-6) and ob.f["v"] == pytest.approx(2.0, rel=1e-6)


def test_portal_teleport_preserves_surface_coordinates(pa):
    """(2) a_to_b = B * A^-1 (src/gui/scene.rs:624-627): a ray hitting portal A at local (u, v)
    continues from portal B at the same local (u, v), with direction mapped by the same matrix."""
    from oracle.glsl_values import Mat, Vec, binop
    from oracle.portal_oracle import Ray, vec

    vals = pa.Scene.from_file(pa.scene_path("basics")).uniform_values()
    tele = [k for k in vals if k.endswith("_mat_teleport")]
    assert tele
    name = tele[0]
    a_name, b_name = name[: -len("_mat_teleport")].split("_to_")
    to_mat = lambda m: Mat([Vec(np.asarray(m, np.float32)[:, c]) for c in range(4)])
    A, A_inv, B_inv, T = to_mat(vals[a_name + "_mat"]), to_mat(vals[a_name + "_mat_inv"]), to_mat(vals[b_name + "_mat_inv"]), to_mat(vals[name])
    n = natives()
    # start one unit in front of portal A at local (0.3, -0.2), fly along A's -normal
    start = binop("*", A, vec(0.3, -0.2, 1.0, 1.0))
    direction = binop("*", A, vec(0.0, 0.0, -1.0, 0.0))
    r = Ray(start, direction, 1.0, False)
    hit = n.plane_intersect(r, A_inv, n.get_normal(A))
    assert bool(hit.f["hit"]) and hit.f["u"] == pytest.approx(0.3, abs=1e-5) and hit.f["v"] == pytest.approx(-0.2, abs=1e-5)
    at_a = n.offset_ray(r, hit.f["t"])
    out = n.transform(T, at_a)
    local = binop("*", B_inv, out.f["o"])
    assert [float(x) for x in local.c[:3]] == pytest.approx([0.3, -0.2, 0.0], abs=2e-5)
    local_d = binop("*", B_inv, out.f["d"])
    assert [float(x) for x in local_d.c[:3]] == pytest.approx([0.0, 0.0, -1.0], abs=2e-5)


def test_panini_centre_and_pinhole_corner():
    """(4) PaniniProjection((0,0), fov, d) = (0,0,1) (src/frag.glsl:305-342); pinhole corner ray at
    fov 90 deg = normalize(aspect, 1, 1) (src/frag.glsl:450-451)."""
    from oracle.portal_oracle import Oracle, vec

    o = Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "basics.ron"))
    for d in (0.0, 0.5, 1.0):
        p = o.panini(vec(0.0, 0.0), F(math.radians(140.0)), F(d))
        assert [float(x) for x in p.c] == pytest.approx([0.0, 0.0, 1.0], abs=1e-6)
    # symmetric in x, unit length
    a = o.panini(vec(0.5, 0.25), F(math.radians(120.0)), F(1.0))
    b = o.panini(vec(-0.5, 0.25), F(math.radians(120.0)), F(1.0))
    assert float(a.c[0]) == -float(b.c[0]) and float(a.c[1]) == float(b.c[1]) and float(a.c[2]) == float(b.c[2])
    assert sum(float(x) ** 2 for x in a.c) == pytest.approx(1.0, abs=1e-6)


def test_quasi_random_and_color_grid_period():
    """(5) quasi_random(0) = (0.5, 0.5) (src/frag.glsl:506-513); (7) color_grid has period 4 and a
    2x2 checker of 1.1 / 0.7 inside one period (src/library.glsl:184-188)."""
    from oracle import glsl_math as M
    from oracle.portal_oracle import vec

    assert M.mod(M.add(F(0.5), M.mul(F(0.7548776662466927), F(0))), F(1.0)) == 0.5
    n = natives()
    start = vec(1.0, 1.0, 1.0)
    cell = lambda x, y: float(n.color_grid(start, vec(x, y)).c[0])
    assert cell(0.5, 0.5) == cell(4.5, 0.5) == cell(0.5, 8.5) == cell(-3.5, 0.5)
    assert {round(cell(0.5, 0.5), 6), round(cell(2.5, 0.5), 6)} == {round(float(F(0.7)), 6), round(float(F(1.1)), 6)}
    assert cell(0.5, 0.5) == cell(2.5, 2.5) and cell(2.5, 0.5) == cell(0.5, 2.5)
    g = natives(_grid_disable=np.int32(1))
    assert float(g.color_grid(start, vec(0.5, 0.5)).c[0]) == 1.0


def render_both(pa, ron_text, w, h, depth=8, tmp_path=None):
    """(oracle frame, product-host-build frame) of a synthetic scene."""
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    path = str(tmp_path / "scene.ron")
    open(path, "w").write(ron_text)
    o = Oracle(path)
    o.options["render_depth"] = depth
    want = o.render(w, h)
    s = pa.Scene.from_file(path)
    r = pa.SceneRenderer(s, device=-1)
    r.set_option("render_depth", depth)
    got = hb.host_kernel_for(r, s, w, h).render(w, h)
    return want, got


def test_flat_wall_colour_and_orientation(pa, tmp_path):
    """A wall facing the camera at distance 2, material Simple(color, normal_coef 0, no grid):
    the centre sample sees exactly `color` (linear), encoded sqrt(color) * 255 rounded; the image
    is uniform up to the angle term (normal_coef = 0 -> exactly uniform)."""
    color = (0.25, 0.5, 1.0)
    want, got = render_both(pa, synthetic.wall_scene(r=2.0, color=color), 16, 16, tmp_path=tmp_path)
    for frame in (want, got):
        px = frame["rgba32f"][7, 7]  # sample at uv_screen = 0 (pixel centre + R2(0) offset)
        assert px[:3] == pytest.approx([math.sqrt(c) for c in color], rel=2e-7) and px[3] == 1.0
        assert frame["rgba8"][7, 7].tolist() == [128, 180, 255, 255]
        assert np.all(frame["rgba8"] == frame["rgba8"][0, 0])
    assert np.array_equal(want["rgba32f"].view(np.uint32), got["rgba32f"].view(np.uint32))


def test_finite_wall_miss_colour(pa, tmp_path):
    """Rays that leave the scene return 0.6^2 grey (src/frag.glsl:150-155, library.glsl:169-171)."""
    want, got = render_both(pa, synthetic.wall_scene(r=2.0, size=0.5), 32, 32, tmp_path=tmp_path)
    for frame in (want, got):
        assert frame["rgba32f"][0, 0, :3] == pytest.approx([0.6] * 3, rel=1e-6)    # sqrt(0.36)
        assert frame["rgba8"][0, 0].tolist() == [153, 153, 153, 255]
        assert frame["rgba8"][15, 15].tolist() == [128, 180, 255, 255]
        inside = (frame["rgba8"][:, :, 0] == 128)
        # |world x| = 2 |uv_screen.x| < 0.5 (strict), samples at uv = (px - 15) / 16  ->  |px - 15| < 4: 7 x 7 pixels
        assert inside.sum() == 7 * 7 and inside[12:19, 12:19].all()
    assert np.array_equal(want["rgba8"], got["rgba8"])


@pytest.mark.parametrize("distance,factor", [(5.0, 1.0), (10.0, 1.0), (110.0, 0.5 ** 4), (210.0, 0.0), (400.0, 0.0)])
def test_distance_darkening(pa, tmp_path, distance, factor):
    """(8) src/frag.glsl:136-146 with t_start 10, t_end 210: colour * (1 - g)^4, g = (t - 10) / 200,
    clamped: unchanged up to 10, black from 210."""
    color = (0.64, 0.64, 0.64)
    want, got = render_both(pa, synthetic.wall_scene(r=distance, color=color), 8, 8, tmp_path=tmp_path)
    for frame in (want, got):
        lin = frame["rgba32f"][3, 3, 0] ** 2
        assert lin == pytest.approx(0.64 * factor, rel=3e-5, abs=1e-7)
    assert np.array_equal(want["rgba32f"].view(np.uint32), got["rgba32f"].view(np.uint32))


def test_angle_shading_term(pa, tmp_path):
    """material_simple2 with normal_coef = 1: colour * |cos(angle between ray and normal)|
    (src/library.glsl:177-181,318-335).  At uv_screen = (x, 0), cos = 1 / sqrt(1 + x^2)."""
    want, got = render_both(pa, synthetic.wall_scene(r=2.0, color=(1.0, 1.0, 1.0), normal_coef=1.0), 16, 16, tmp_path=tmp_path)
    for frame in (want, got):
        for col in (7, 11, 15):
            x = (col + 1 - 8) * 2 / 16.0          # uv_screen.x of the sample in that pixel, y = 0 in row 7
            assert frame["rgba32f"][7, col, 0] ** 2 == pytest.approx(1 / math.sqrt(1 + x * x), rel=1e-5)
    assert np.array_equal(want["rgba32f"].view(np.uint32), got["rgba32f"].view(np.uint32))


def test_mirror_material_and_depth_limit(pa, tmp_path):
    """Two facing mirrors (Reflect) around the camera: the path never terminates, so the pixel is
    black (`depth exhausted`, src/frag.glsl:158) for any depth; with one mirror replaced by the
    coloured wall the ray comes back after one bounce with the mirror's colour factor."""
    extra_m = '(name: "back", data: Simple(offset: (0.0, 0.0, 4.0), scale: 1.0, rotate: (0.0, 0.0, 0.0), mirror: (false, false, false))),'
    extra_o = '(name: "back", data: Flat(kind: Simple(Some(Named("back"))), is_inside: (("return mirror_M;")), in_subspace: Normal)),'
    extra_mat = '(name: "mirror", data: Reflect(add_to_color: (0.5, 0.5, 0.5))),'
    text = synthetic.wall_scene(r=2.0, color=(1.0, 1.0, 1.0), extra_matrices=extra_m, extra_objects=extra_o, extra_materials=extra_mat)
    # camera at z = 2 looks at the wall (z = 0); the mirror at z = 4 is behind it: not visible -> plain wall
    want, got = render_both(pa, text, 8, 8, depth=4, tmp_path=tmp_path)
    assert want["rgba8"][3, 3].tolist() == [255, 255, 255, 255] and np.array_equal(want["rgba8"], got["rgba8"])
    # now look the other way (alpha = -pi/2 puts the camera at z = -2... so use a wall that is itself a mirror)
    text2 = text.replace("return wall_M; } return NOT_INSIDE;", "return mirror_M; } return NOT_INSIDE;")
    want2, got2 = render_both(pa, text2, 8, 8, depth=6, tmp_path=tmp_path)
    assert want2["rgba8"][3, 3].tolist() == [0, 0, 0, 255]            # mirror <-> mirror: depth exhausted -> black
    assert want2["segments"][3, 3] == 6                                # exactly `depth` bounce-loop trips
    assert np.array_equal(want2["rgba8"], got2["rgba8"])
