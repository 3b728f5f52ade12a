"""Differential fuzz of the distance bound in intersection-material snippets (host/glsl_translate.h `bound_nearer_blocks`).

Random scenes: a few walls at random places (scene_intersect's candidates) and one intersection-material snippet that walks K random
discs -- each in an `if (nearer(result.scene.hit, hit_k)) { ... }` block, accepted as a teleport (CUSTOM_MATERIAL + its own material), as a
solid colour, or not at all -- so that snippet candidates in front of, between and behind the walls all occur.  The kernel with the bound (FLAG_BOUNDED_SNIPPETS:
scene_intersect first, its distance handed to the snippet), the kernel with the snippet as written (evaluated first) and the numpy
oracle (which knows neither order) must draw the same bits."""
import random

import numpy as np
import pytest

from tests.synthetic import HEAD, TAIL


def bound_fuzz_scene(seed):
    rnd = random.Random(seed)
    f = lambda lo, hi: repr(round(rnd.uniform(lo, hi), 3))
    k_discs = rnd.randint(2, 5)
    matrices, objects, materials = [], [], []
    for w in range(rnd.randint(1, 3)):  # walls: squares of random size somewhere in front of the camera
        matrices.append(f'(name: "wall{w}", data: Simple(offset: ({f(-1, 1)}, {f(-1, 1)}, {f(-3.0, 0.5)}), scale: {f(0.6, 2.5)}, rotate: ({f(-0.6, 0.6)}, {f(-0.6, 0.6)}, {f(-3, 3)}), mirror: (false, false, false))),')
        objects.append(f'(name: "wall{w}", data: Flat(kind: Simple(Some(Named("wall{w}"))), is_inside: (("if (abs(x) < 1. && abs(y) < 1.) {{ return wall{w}_M; }} return NOT_INSIDE;")), in_subspace: Normal)),')
        materials.append(f'(name: "wall{w}", data: Simple(color: ({f(0.1, 1)}, {f(0.1, 1)}, {f(0.1, 1)}), normal_coef: {f(0, 0.6)}, grid: {rnd.choice(["true", "false"])}, grid_scale: 1.0, grid_coef: 0.3, grid2: false, grid3: false)),')
    for k in range(k_discs):
        matrices.append(f'(name: "disc{k}", data: Simple(offset: ({f(-1.2, 1.2)}, {f(-1.2, 1.2)}, {f(-3.5, 1.0)}), scale: {f(0.3, 1.4)}, rotate: ({f(-1, 1)}, {f(-1, 1)}, {f(-3, 3)}), mirror: (false, false, false))),')
        materials.append(f'(name: "disc{k}", data: Simple(color: ({f(0.1, 1)}, {f(0.1, 1)}, {f(0.1, 1)}), normal_coef: {f(0, 0.6)}, grid: false, grid_scale: 1.0, grid_coef: 0.3, grid2: false, grid3: false)),')
    matrices.append(f'(name: "jump", data: Simple(offset: ({f(-0.5, 0.5)}, {f(-0.5, 0.5)}, {f(-0.5, 0.5)}), scale: {f(0.8, 1.2)}, rotate: ({f(-0.5, 0.5)}, {f(-0.5, 0.5)}, {f(-0.5, 0.5)}), mirror: (false, false, false))),')
    code = ["SceneIntersectionWithMaterial result = SceneIntersectionWithMaterial(scene_intersection_none, material_empty());"]
    for k in range(k_discs):
        # inside the inner radius: a teleport through `jump` (CUSTOM_MATERIAL, the block stores the material); in the ring: a solid colour; else nothing
        inner, outer = round(rnd.uniform(0.2, 0.7), 3), round(rnd.uniform(0.7, 1.0), 3)
        code += [f"SurfaceIntersection hit{k} = plane_intersect(r, disc{k}_mat_inv, get_normal(disc{k}_mat));",
                 f"if (nearer(result.scene.hit, hit{k})) {{",
                 f"  float d{k} = sqrt(hit{k}.u * hit{k}.u + hit{k}.v * hit{k}.v);",
                 f"  int inside{k} = NOT_INSIDE;",
                 f"  if (d{k} < {inner}) {{ inside{k} = TELEPORT; }} else if (d{k} < {outer}) {{ inside{k} = disc{k}_M; }}",
                 f"  if (inside{k} != NOT_INSIDE) {{",
                 f"    result.scene = process_portal_intersection(result.scene, hit{k}, inside{k}, CUSTOM_MATERIAL);",
                 f"    if (result.scene.material == CUSTOM_MATERIAL) {{",
                 f"      result.material = material_teleport_transformed(offset_ray(transform(jump_mat, r), hit{k}.t), vec3({f(0.5, 1)}, {f(0.5, 1)}, {f(0.5, 1)}));",
                 "    }", "  }", "}"]
    code.append("return result;")
    nl = "\\n"
    text = (HEAD % dict(r=2.5)) + f"""
    uniforms: ([]),
    matrices: ([ {' '.join(matrices)} ]),
    objects: ([ {' '.join(objects)} ]),
    cameras: ([]),
    textures: ([]),
    materials: ([ {' '.join(materials)} ]),
    intersection_materials: ([ (name: "discs", data: ((("{nl.join(code)}")))) ]),
    library: ([]),
""" + TAIL
    cam = dict(look_at=(round(rnd.uniform(-0.3, 0.3), 3), round(rnd.uniform(-0.3, 0.3), 3), round(rnd.uniform(-1.5, 0.0), 3)), alpha=round(rnd.uniform(1.2, 1.9), 3),
               beta=round(rnd.uniform(1.2, 1.9), 3), r=round(rnd.uniform(1.5, 3.0), 3))
    return text, cam


@pytest.mark.parametrize("seed", range(7000, 7010))
def test_bounded_snippet_equals_the_snippet_as_written_and_the_oracle(pa, seed, tmp_path):
    from oracle import host_build as hb
    from oracle.portal_oracle import Oracle

    text, cam = bound_fuzz_scene(seed)
    path = str(tmp_path / "b.ron")
    with open(path, "w") as f:
        f.write(text)
    w, h, depth = 48, 32, 5
    frames = []
    for extra in (pa.FLAG_BOUNDED_SNIPPETS, 0):
        sc = pa.Scene.from_file(path)
        src = sc.generate_source(extra)
        assert ("> ptl_far)" in src) == (extra != 0), "the fuzz snippet must qualify for the bound"
        r = pa.SceneRenderer(sc, device=-1, flags=pa.FLAG_QUICK_JIT)
        r.set_option("render_depth", depth)
        r.set_camera(cam["look_at"], cam["alpha"], cam["beta"], cam["r"])
        frames.append(hb.host_kernel_for(r, sc, w, h, flags=extra).render(w, h)["rgba32f"].copy())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    o = Oracle(path)
    o.options["render_depth"] = depth
    o.camera = dict(cam)
    want = o.render(w, h)
    same = (frames[0].view(np.uint32) == want["rgba32f"].view(np.uint32)) | (np.isnan(frames[0]) & np.isnan(want["rgba32f"]))
    assert same.all()
    assert len(np.unique(frames[0].reshape(-1, 4), axis=0)) > 3  # walls and discs are in the picture
