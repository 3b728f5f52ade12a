"""Tiny hand-written scenes in the reference's .ron format, for known-answer tests."""

HEAD = """(
    desc: (eng: "", rus: ""),
    cam: (look_at: (0.0, 0.0, 0.0), alpha: 1.5707963267948966, beta: 1.5707963267948966, r: %(r)r, offset_after_material: 0.000025),
"""
TAIL = """
    animation_stages: ([]),
)
"""


def wall_scene(r=2.0, color=(0.25, 0.5, 1.0), normal_coef=0.0, grid=False, size=1000.0, extra_objects="", extra_matrices="", extra_materials="", library=""):
    """Camera at (0,0,r) looking down -z at the plane z = 0 (identity matrix `wall`)."""
    return (HEAD % dict(r=r)) + """
    uniforms: ([ (name: "size", data: Float((min: None, max: None, value: %(size)r))) ]),
    matrices: ([
        (name: "wall", data: Simple(offset: (0.0, 0.0, 0.0), scale: 1.0, rotate: (0.0, 0.0, 0.0), mirror: (false, false, false))),
        %(extra_matrices)s
    ]),
    objects: ([
        (name: "wall", data: Flat(kind: Simple(Some(Named("wall"))), is_inside: (("if (abs(x) < size_u && abs(y) < size_u) { return wall_M; } return NOT_INSIDE;")), in_subspace: Normal)),
        %(extra_objects)s
    ]),
    cameras: ([]),
    textures: ([]),
    materials: ([
        (name: "wall", data: Simple(color: (%(c0)r, %(c1)r, %(c2)r), normal_coef: %(normal_coef)r, grid: %(grid)s, grid_scale: 1.0, grid_coef: 0.3, grid2: false, grid3: false)),
        %(extra_materials)s
    ]),
    intersection_materials: ([]),
    library: ([ %(library)s ]),
""" % dict(size=size, c0=color[0], c1=color[1], c2=color[2], normal_coef=normal_coef, grid="true" if grid else "false", extra_objects=extra_objects,
           extra_matrices=extra_matrices, extra_materials=extra_materials, library=library) + TAIL


def kitchen_sink_scene():
    """One small scene with the features the five BASELINE scenes do not have together: a Complex object written in GLSL (a unit
    sphere), Refract and Reflect materials, a DebugMatrix gizmo, an object that only exists in the subspace, a skybox texture
    (`sky_tex`, the caller provides img/sky.png under the asset root) and a formula-driven matrix."""
    matrices = '''
        (name: "ball", data: Parametrized(offset: (x: Value(0.35), y: Uniform(Some(Named("lift"))), z: Value(0.6)), rotate: (x: Value(0.0), y: Value(0.3), z: Value(0.0)),
                                          mirror: (x: Value(0.0), y: Value(0.0), z: Value(0.0)), scale: Value(0.45))),
        (name: "side", data: Simple(offset: (-0.9, 0.0, 0.5), scale: 1.0, rotate: (0.0, 1.5707963267948966, 0.0), mirror: (false, false, false))),
        (name: "gizmo", data: Simple(offset: (-0.3, -0.4, 0.8), scale: 0.4, rotate: (0.2, 0.4, 0.1), mirror: (false, false, false))),
        (name: "hidden", data: Simple(offset: (0.0, 0.0, 0.3), scale: 1.0, rotate: (0.0, 0.0, 0.0), mirror: (false, false, false))),
    '''
    sphere = ("vec3 op = -r.o.xyz;\\nfloat b = dot(op, r.d.xyz);\\nfloat det = b*b - dot(op, op) + 1.0;\\nif (det < 0.) return scene_intersection_none;\\n"
              "det = sqrt(det);\\nfloat t = b - det;\\nif (t < 0.) t = b + det;\\nif (t < 0.) return scene_intersection_none;\\nvec4 pos = r.o + r.d * t;\\n"
              "return SceneIntersection(glass_M, SurfaceIntersection(true, t, pos.x, pos.y, normalize(pos.xyz)), false);")
    objects = f'''
        (name: "ball", data: Complex(kind: Simple(Some(Named("ball"))), intersect: (("{sphere}")), in_subspace: Normal)),
        (name: "side", data: Flat(kind: Simple(Some(Named("side"))), is_inside: (("if (abs(x) < 0.8 && abs(y) < 0.8) {{ return mirror_M; }} return NOT_INSIDE;")), in_subspace: Normal)),
        (name: "gizmo", data: DebugMatrix(Some(Named("gizmo")))),
        (name: "hidden", data: Flat(kind: Simple(Some(Named("hidden"))), is_inside: (("return wall_M;")), in_subspace: Subspace)),
    '''
    materials = '''
        (name: "glass", data: Refract(add_to_color: (0.9, 0.95, 1.0), refractive_index: 1.4)),
        (name: "mirror", data: Reflect(add_to_color: (0.8, 0.8, 0.8))),
    '''
    text = wall_scene(r=2.5, color=(0.9, 0.6, 0.3), normal_coef=0.4, grid=True, size=0.9, extra_objects=objects, extra_matrices=matrices, extra_materials=materials)
    text = text.replace('uniforms: ([', 'uniforms: ([ (name: "lift", data: Formula(("0.1 + 0.05 * sin(3)"))),')
    text = text.replace("    textures: ([]),", '    textures: ([ (name: "sky", data: "img/sky.png") ]),')
    return text.replace("    animation_stages: ([]),", '    animation_stages: ([]),\n    skybox: Some("sky"),')


def write_sky_texture(pa, root):
    """A 16 x 8 gradient under <root>/img/sky.png."""
    import os

    import numpy as np

    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    img = np.zeros((8, 16, 4), np.uint8)
    img[..., 0] = np.linspace(30, 220, 16, dtype=np.uint8)[None, :]
    img[..., 1] = np.linspace(200, 40, 8, dtype=np.uint8)[:, None]
    img[..., 2] = 128
    img[..., 3] = 255
    pa.png_write(os.path.join(root, "img", "sky.png"), img)
