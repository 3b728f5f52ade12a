"""GPU parity: the hiprtc-compiled kernel on MI355X against the checker, through the C ABI.

Bar (BASELINE.json north_star): pixel-for-pixel within 1e-5 per channel.  What is enforced
here is stricter: the linear-float frame is BIT-IDENTICAL to the host build of the same
arithmetic contract (device/ptl_glsl.h), and the RGBA8 frame is byte-identical.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # scene, width, height, depth, aa
    ("basics", 256, 256, 4, 1),
    ("monoportal", 320, 180, 20, 1),
    ("triple_portal", 256, 144, 40, 1),
    ("portal_in_portal", 256, 144, 40, 1),
    ("mobius_monoportal", 160, 90, 64, 2),
]


@pytest.fixture(scope="module")
def gpu(pa):
    if pa.device_count() < 1:
        pytest.fail("no HIP device visible: the render path has no CPU fallback")
    return pa


@pytest.mark.parametrize("scene_name,w,h,depth,aa", CASES)
def test_frame_bit_exact_vs_host_build(gpu, scene_name, w, h, depth, aa):
    from oracle import host_build

    pa = gpu
    scene = pa.Scene.from_file(pa.scene_path(scene_name))
    r = pa.SceneRenderer(scene, device=0)
    r.set_option("render_depth", depth)
    r.set_option("aa_count", aa)
    out = r.draw(w, h, rgba8=True, rgba32f=True)
    ref = host_build.host_kernel_for(r, scene, w, h).render(w, h)
    a, b = out["rgba32f"], ref["rgba32f"]
    same = a.view(np.uint32) == b.view(np.uint32)
    both_nan = np.isnan(a) & np.isnan(b)
    bad = ~(same | both_nan)
    assert not bad.any(), f"{int(bad.any(axis=2).sum())} of {w*h} pixels differ; max abs err {np.nanmax(np.abs(a - b))}"
    assert np.array_equal(out["rgba8"], ref["rgba8"])
    assert np.nanmax(np.abs(a - b)) <= 1e-5  # the north_star tolerance, implied by the above


def test_sharded_frame_equals_whole_frame(gpu):
    """Row-block interleave (phase, stride) + de-interleave reproduces the single-launch frame."""
    pa = gpu
    scene = pa.Scene.from_file(pa.scene_path("monoportal"))
    r = pa.SceneRenderer(scene, device=0)
    r.set_option("render_depth", 20)
    w, h = 200, 100  # ragged: 100 rows = 12 full blocks + 4 rows, 200 px = 6 full 32-px blocks + 8
    whole = r.draw(w, h)["rgba8"]
    full = np.zeros_like(whole)
    for phase in range(3):
        shard = r.draw(w, h, rb_phase=phase, rb_stride=3)["rgba8"]
        pa.deinterleave_rows(shard, pa.Frame(w, h, phase, 3), full)
    assert np.array_equal(full, whole)


def test_segment_counter_matches_host(gpu):
    from oracle import host_build

    pa = gpu
    scene = pa.Scene.from_file(pa.scene_path("monoportal"))
    r = pa.SceneRenderer(scene, device=0, flags=pa.FLAG_COUNT_SEGMENTS)
    r.set_option("render_depth", 20)
    out = r.draw(128, 72, segments=True)
    hk = host_build.host_kernel_for(r, scene, 128, 72, flags=pa.FLAG_COUNT_SEGMENTS, count_segments=True)
    ref = hk.render(128, 72)
    assert out["segments"] == ref["segments"] > 128 * 72
