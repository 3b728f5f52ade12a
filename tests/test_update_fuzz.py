"""Differential fuzz of the per-frame host step (SceneRenderer::update + Scene::update + init_stage, src/main.rs:1430-1538,
src/gui/scene.rs:1180-1236,1353-1493): random sequences of "initialise this stage", "initialise this clip", "use this camera"
and "advance to t seconds" (forwards, backwards, beyond the clip's end) on the product's renderer and on the oracle's CameraRig;
after every step both agree on formula time, every scene uniform and the camera uniforms, bit for bit.  (Portal crossing needs
the GPU ray query and is switched off here; tests/test_gpu_parity.py covers it.)"""
import random

import numpy as np
import pytest


@pytest.mark.parametrize("scene_name,seed", [(n, s) for n in ("portal_in_portal", "triple_portal", "basics") for s in range(4)])
def test_random_update_sequences_agree(pa, scene_name, seed):
    from oracle.portal_oracle import CameraRig, Oracle
    from oracle.scene_eval import builtin_uniforms
    from tests.test_host_logic import _same_uniforms

    path = pa.scene_path(scene_name)
    ps, o = pa.Scene.from_file(path), Oracle(path)
    r = pa.SceneRenderer(ps, device=-1)
    r.set_option("allow_teleport", 0)
    rig = CameraRig(o)
    rig.allow_teleport = False
    rnd = random.Random(seed * 7919 + len(scene_name))
    stages, clips, cameras = ps.stages(), ps.animations(), [c for c in ps.cameras() if not c.startswith("#")]
    for step in range(40):
        k = rnd.random()
        if k < 0.15 and stages:
            name = rnd.choice(stages)
            ps.init_stage(name)
            o.scene.init_stage(name)
        elif k < 0.35 and clips:
            name = rnd.choice(clips)[0]
            ps.init_animation(name)
            o.scene.init_animation(name)
        t = rnd.choice([0.0, rnd.uniform(0, 1), rnd.uniform(0, 6), rnd.uniform(0, 40)])
        r.update(t)
        rig.update(t)
        got, want = ps.update(t), o.scene.update(t)  # the scene-level step once more on both sides (same t): exposes time / total_time
        assert (got["time"], got["total_time"]) == (o.scene.time, o.scene.total_time) and (got["camera"] is None) == (want is None), (step, t)
        assert r.camera_state()["in_subspace"] == rig.in_subspace, (step, t)
        _same_uniforms(ps.uniform_values(), o.scene.scene_uniform_values(), (scene_name, seed, step))
        b = builtin_uniforms(o.scene, 160, 90, camera=rig.settings())
        for name in ("_camera", "_camera_mul_inv", "_camera_scale", "_camera_in_subspace"):
            g = r.uniform_value(name, 160, 90)
            g = np.asarray(g).T.reshape(16) if np.asarray(g).shape == (4, 4) else np.asarray(g).reshape(-1)
            assert np.array_equal(g.astype(np.float32), np.asarray(b[name], np.float32).reshape(-1)), (scene_name, seed, step, name)
