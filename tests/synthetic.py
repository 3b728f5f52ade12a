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
