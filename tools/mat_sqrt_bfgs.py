#!/usr/bin/env python3
"""tools/mat_sqrt_bfgs.py -- Matrix::Sqrt as the reference computes it against the exact principal root this build returns (VERDICT r4 #6 / next #7b).

The reference (/root/reference/src/gui/matrix.rs:909-988, argmin 0.9): minimise ||M*M - A||^2 over the 12 free elements of an affine M, start M = A,
BFGS with identity inverse Hessian, More-Thuente line search (c1 = 1e-4, c2 = 0.9), forward-difference gradient (finitediff: step sqrt(eps)), at most
60 iterations, accepted when the cost is < 1e-4.  argmin cannot be run here (no Rust); this restates the METHOD with scipy's BFGS (same objective,
same start, same gradient, a Wolfe line search with the same constants) to see where such a minimiser ends up relative to the exact root, and what
that distance is worth in pixels of the one corpus scene that uses Matrix::Sqrt (portal_in_portal_plus_ultra.ron with `show_sqrt` on).
MEASUREMENT TOOL: imports oracle/ (the numpy restatement) -- nothing under portal_amd/ uses this."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)


def cost(p, a):
    m = np.vstack([p.reshape(3, 4), [0.0, 0.0, 0.0, 1.0]])  # rows; the reference's vec_to_mat2 + transpose
    d = m @ m - a
    return float((d * d).sum())


def forward_gradient(p, a):
    h = np.sqrt(np.finfo(np.float64).eps)
    f0 = cost(p, a)
    g = np.empty_like(p)
    for i in range(len(p)):
        q = p.copy()
        q[i] += h
        g[i] = (cost(q, a) - f0) / h
    return g


def reference_method(a_rows, max_iters=60):
    from scipy.optimize import minimize

    res = minimize(cost, a_rows[:3].reshape(-1).copy(), args=(a_rows,), method="BFGS", jac=forward_gradient, options={"maxiter": max_iters, "c1": 1e-4, "c2": 0.9, "gtol": 0.0})
    return np.vstack([res.x.reshape(3, 4), [0.0, 0.0, 0.0, 1.0]]), float(res.fun), int(res.nit)


if __name__ == "__main__":
    import portal_amd as pa
    from oracle import scene_eval as SE
    from oracle.portal_oracle import Oracle

    path = os.path.join(HERE, "tests", "corpus", "scenes", "portal_in_portal_plus_ultra.ron")
    root = os.path.join(HERE, "tests", "corpus")
    out = {"scene": "tests/corpus/scenes/portal_in_portal_plus_ultra.ron", "states": []}
    # (the scene's saved view does not look at the square-root portal: these cameras do -- 16 % / 3 % / 1.5 % of the frame are that portal)
    views = [dict(look_at=(0.0, 0.0, -0.5), alpha=4.0, beta=1.57, r=1.5), dict(look_at=(0.0, 0.0, -0.5), alpha=4.0, beta=1.0, r=3.0), dict(look_at=(0.0, 0.0, -0.5), alpha=2.8, beta=1.57, r=1.5)]
    for view in views:
        state = {"show_sqrt": 1.0}
        sc = pa.Scene.from_file(path)
        for k, v in state.items():
            sc.set_uniform(k, v)
        a = np.array(sc.eval_matrix("b0"), np.float64)                                # m[row, col]
        a_cols = a.T.copy()                                                          # [column][row]
        # the exact principal root as this build (scene.cpp) and the oracle (scene_eval.py) return it: Denman-Beavers in binary64
        y, z = [list(c) for c in a_cols], SE.IDENT
        for _ in range(64):
            yn = [[(y[c][r] + SE.m_inverse(z)[c][r]) * 0.5 for r in range(4)] for c in range(4)]
            zn = [[(z[c][r] + SE.m_inverse(y)[c][r]) * 0.5 for r in range(4)] for c in range(4)]
            done = yn == y
            y, z = yn, zn
            if done:
                break
        exact = np.array(y).T
        got, c_end, iters = reference_method(a)
        rec = {"state": state, "camera": view, "iterations": iters, "cost_reached": c_end, "cost_of_the_exact_root": cost(exact[:3].reshape(-1), a),
               "accepted_by_the_reference": c_end < 1e-4, "max_abs_difference_from_the_exact_root": float(np.abs(got - exact).max()),
               "max_abs_difference_after_f32_rounding": float(np.abs(got.astype(np.float32).astype(np.float64) - exact.astype(np.float32)).max()),
               "f32_elements_that_differ": int((got.astype(np.float32) != exact.astype(np.float32)).sum())}
        # what it is worth in pixels: the oracle's frame with the exact root against the frame with the minimiser's matrix put in its place
        w, h = 320, 180
        frames = []
        for m in (None, got):
            o = Oracle(path, asset_root=root)
            for k, v in state.items():
                u = o.scene.uniforms[o.scene.find_uniform(k)]
                u[2] = bool(v) if u[1] == "bool" else (int(v) if u[1] == "int" else float(v))
            o.options["render_depth"] = 24
            o.camera = dict(view)
            if m is not None:
                o.scene.override_sqrt = [[float(m[r][c]) for r in range(4)] for c in range(4)]  # [column][row]
            frames.append(o.render(w, h))
        with_portal = Oracle(path, asset_root=root)   # (how much of this frame IS the square-root portal: the same view with it switched off)
        with_portal.options["render_depth"] = 24
        with_portal.camera = dict(view)
        rec["pixels_the_portal_decides"] = int((with_portal.render(w, h)["rgba8"] != frames[0]["rgba8"]).any(axis=2).sum())
        differ32 = int((frames[0]["rgba32f"].view(np.uint32) != frames[1]["rgba32f"].view(np.uint32)).any(axis=2).sum())
        differ8 = int((frames[0]["rgba8"] != frames[1]["rgba8"]).any(axis=2).sum())
        err = np.abs(frames[0]["rgba32f"][..., :3].astype(np.float64) - frames[1]["rgba32f"][..., :3].astype(np.float64))
        rec.update({"frame": f"{w}x{h} depth 24", "pixels_with_other_float_bits": differ32, "pixels_with_other_rgba8": differ8, "pixels_beyond_1e-5": int((err.max(axis=2) > 1e-5).sum()),
                    "max_abs_colour_difference": float(np.nanmax(err))})
        out["states"].append(rec)
        print(json.dumps(rec), flush=True)
    os.makedirs(os.path.join(HERE, "profiles", "r05"), exist_ok=True)
    json.dump(out, open(os.path.join(HERE, "profiles", "r05", "mat_sqrt_bfgs.json"), "w"), indent=1)
