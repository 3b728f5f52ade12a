"""Differential fuzz of the matrix / uniform graph evaluators: random chains of every Matrix kind (Simple, Parametrized, Exact,
ExactFull, Mul, Teleport, If, Inv, Lerp, Sqrt, Camera) over random uniforms (values and formulas) through the product's C++
evaluator (scene.cpp, dmath.h) and the oracle's Python one (oracle/scene_eval.py).  What both upload to the kernel -- X_mat,
X_mat_inv as binary32 -- must agree bit for bit, including the NaN / inf of singular cases, or be missing on both sides."""
import random

import numpy as np
import pytest

from tests import synthetic


def random_scene(seed):
    r = random.Random(seed)
    num = lambda lo=-2.0, hi=2.0: repr(round(r.uniform(lo, hi), r.choice([0, 1, 3, 6])))
    uniforms = ['(name: "u0", data: Float((min: None, max: None, value: %s)))' % num(), '(name: "u1", data: Angle(%s))' % num(-3.2, 3.2),
                '(name: "u2", data: Progress(%s))' % num(0.0, 1.0), '(name: "u3", data: Bool(%s))' % r.choice(["true", "false"]),
                '(name: "u4", data: Formula(("u0 * 0.5 + sin(u1) + time")))', '(name: "u5", data: Int((min: None, max: None, value: %d)))' % r.randint(-2, 3),
                '(name: "u6", data: Formula(("if(u3, u2, 1 - u2) / (u5 - 1)")))']
    names = []

    def param():
        k = r.randrange(4)
        if k == 0:
            return "Value(%s)" % num()
        if k == 1:
            return 'Uniform(Some(Named("u%d")))' % r.randrange(7)
        if k == 2:
            return 'Uniform(Some(Inline(Formula(("%s")))))' % r.choice(["u0 + 1", "cos(u1) * 2", "u2 ^ 2", "1 / u5", "time * 3 - 1", "sqrt(u0)"])
        return "Value(%s)" % r.choice(["0.0", "1.0", "-1.0", "0.5"])

    def ref():
        if not names or r.random() < 0.15:
            return "Some(Inline(%s))" % leaf()
        return 'Some(Named("%s"))' % r.choice(names)

    def tvec(fields="xyz"):
        return "(" + ", ".join(f"{c}: {param()}" for c in fields) + ")"

    def leaf():
        k = r.randrange(4)
        if k == 0:
            return "Simple(offset: (%s, %s, %s), scale: %s, rotate: (%s, %s, %s), mirror: (%s, %s, %s))" % (
                num(), num(), num(), r.choice([num(0.1, 3.0), "0.0", "1.0"]), num(-3.2, 3.2), num(-3.2, 3.2), num(-3.2, 3.2),
                *[r.choice(["true", "false"]) for _ in range(3)])
        if k == 1:
            return "Parametrized(offset: %s, rotate: %s, mirror: %s, scale: %s)" % (tvec(), tvec(), tvec(), param())
        if k == 2:
            return "Exact(i: %s, j: %s, k: %s, pos: %s)" % (tvec(), tvec(), tvec(), tvec())
        return "ExactFull(c0: %s, c1: %s, c2: %s, c3: %s)" % (tvec("xyzw"), tvec("xyzw"), tvec("xyzw"), tvec("xyzw"))

    def node():
        k = r.randrange(9)
        if k <= 2:
            return leaf()
        if k == 3:
            return "Mul(to: %s, what: %s)" % (ref(), ref())
        if k == 4:
            return "Teleport(first_portal: %s, second_portal: %s, what: %s)" % (ref(), ref(), ref())
        if k == 5:
            return "If(condition: %s, then: %s, otherwise: %s)" % (param(), ref(), ref())
        if k == 6:
            return "Inv(%s)" % ref()
        if k == 7:
            return "Lerp(t: %s, first: %s, second: %s)" % (param(), ref(), ref())
        return r.choice(["Sqrt(%s)" % ref(), "Mul(to: Some(Inline(Camera)), what: %s)" % ref()])

    entries = []
    for k in range(10):
        entries.append('(name: "m%d", data: %s),' % (k, node()))
        names.append("m%d" % k)
    text = synthetic.wall_scene(extra_matrices="\n".join(entries))
    return text.replace('uniforms: ([', 'uniforms: ([ ' + ", ".join(uniforms) + ","), names


@pytest.mark.parametrize("seed", range(200))
def test_random_matrix_graphs_product_equals_oracle(pa, seed):
    from oracle.scene_eval import OracleScene

    text, names = random_scene(seed)
    s, o = pa.Scene.from_text(text), OracleScene(text, is_text=True)
    cam = np.eye(4)
    cam[:3, 3] = (0.3, -1.2, 2.0)
    cam[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    s.set_camera_matrix(cam)
    o.camera_object_matrix = [list(col) for col in cam.T]
    for seconds in (0.0, 0.4):
        s.update(seconds)
        o.update(seconds)
        got, want = s.uniform_values(), o.scene_uniform_values()
        for name in names + ["u4", "u6"]:
            for key in ([name + "_mat", name + "_mat_inv"] if name.startswith("m") else [name + "_u"]):
                assert (key in got) == (key in want), (seed, key)
                if key in got:
                    g = np.asarray(got[key], np.float32)
                    g = g.T.reshape(-1) if g.shape == (4, 4) else g.reshape(-1)
                    w = np.asarray(want[key], np.float32).reshape(-1)
                    assert np.all((g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w))), (seed, seconds, key, g, w)
