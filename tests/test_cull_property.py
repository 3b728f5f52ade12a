"""The wave-level plane cull (portal_amd/csrc/device/ptl_library.h `ptl_cannot_be_nearer`) never removes a plane test whose result
scene_intersect could have selected.

The cull is the one place where the generated kernel decides NOT to run a piece of the reference's algorithm (library.glsl:138-162
`plane_intersect`), so the claim "frames stay bit-identical" rests on this implication, for every float input:

    culled(o'.z, d'.z, best)   =>   not ( the contract's plane chain gives a hit with  t > 0  and  (t < best  or  best is +inf) )

Checked here on adversarial scalars -- every exponent range, zeros of both signs, denormals, infinities, NaN, rays a hair before /
beyond the bound -- with the oracle's own contract arithmetic (oracle/glsl_math.py) for the chain and a numpy restatement of the three
instructions of the cull.  The scene-level form of the same claim is tests/test_parity_cpu.py / test_corpus.py (the host build runs
the cull, the numpy oracle has none).
"""
import numpy as np

from oracle import glsl_math as M

F32 = np.float32


def cannot_be_nearer(oz, dz, best):
    """ptl_library.h `ptl_cannot_be_nearer`: oz * fma(best * (1 + 2^-16), dz, oz) > 0, each step rounded to binary32."""
    with np.errstate(all="ignore"):
        far = (best * F32(1.0 + 2.0 ** -16)).astype(F32)
        # one rounding of the exact far * dz + oz: binary64 holds the product of two binary32 numbers exactly, and the sum of it with a
        # third is rounded once more to binary64 first -- a double rounding that can differ from the FMA by one ulp of binary32 only
        # within 2^-29 relative of a tie, never in sign, and the sign is all that is read.
        z_far = (far.astype(np.float64) * dz.astype(np.float64) + oz.astype(np.float64)).astype(F32)
        return (oz * z_far).astype(F32) > 0


def chain_t(oz, dx, dy, dz):
    """t of plane_intersect for a ray already in the plane's frame (oracle/portal_oracle.py `plane_intersect`): normalise d', intersect
    z = 0, scale back.  Returns (hit, t)."""
    with np.errstate(all="ignore"):
        dot = M.fma(dz, dz, M.fma(dy, dy, M.term0(dx, dx)))
        ln = M.sqrt(dot)
        dnz = M.div(dz, ln)
        t_n = M.div(M.neg(oz), dnz)
        hit = ~(t_n < 0)
        return hit, M.div(t_n, ln)


def selectable(hit, t, best):
    with np.errstate(all="ignore"):
        return hit & (t > 0) & ((t < best) | np.isposinf(best))


def specials():
    e = [0.0, -0.0, 1e-45, -1e-45, 1e-39, -1e-39, 2.0 ** -126, -(2.0 ** -126), 2.0 ** -100, 2.0 ** -64, -(2.0 ** -64), 1e-20, -1e-20, 1e-6, -1e-6,
         0.5, -0.5, 1.0, -1.0, 3.0, -3.0, 1e6, -1e6, 1e19, -1e19, 2.0 ** 63, 2.0 ** 64, -(2.0 ** 64), 2.0 ** 126, 2.0 ** 127, -(2.0 ** 127),
         3.4e38, -3.4e38, np.inf, -np.inf, np.nan]
    return np.array(e, dtype=F32)


def test_cull_never_removes_a_selectable_hit_on_special_values():
    prev = M.set_contract(2)
    try:
        s = specials()
        oz, dx, dz, best = np.meshgrid(s, s[[0, 7, 13, 17, 21, 25, 31, 33, 35]], s, s, indexing="ij")
        oz, dx, dz, best = (a.ravel().astype(F32) for a in (oz, dx, dz, best))
        dy = np.zeros_like(dx)
        culled = cannot_be_nearer(oz, dz, best)
        hit, t = chain_t(oz, dx, dy, dz)
        bad = culled & selectable(hit, t, best)
        assert culled.sum() > 1000
        assert not bad.any(), [(oz[i], dx[i], dz[i], best[i], t[i]) for i in np.flatnonzero(bad)[:5]]
    finally:
        M.set_contract(prev)


def test_cull_never_removes_a_selectable_hit_near_the_bound():
    """Rays whose crossing is within a few 2^-16 of the best distance: both sides of the margin."""
    prev = M.set_contract(2)
    try:
        rng = np.random.default_rng(4)
        n = 400_000
        mag = lambda lo, hi: (np.exp2(rng.uniform(lo, hi, n)) * rng.choice([-1.0, 1.0], n)).astype(F32)
        dz, dx, dy = mag(-30, 30), mag(-30, 30), mag(-30, 30)
        t_true = np.exp2(rng.uniform(-20, 20, n))
        oz = (-(t_true * dz.astype(np.float64))).astype(F32)
        with np.errstate(all="ignore"):
            ln = np.sqrt(dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2 + dz.astype(np.float64) ** 2)
        # the bound a relative 2^-18 .. 2^-13 below / above the crossing, and exactly on it
        rel = rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-24, -12, n)) * rng.choice([0.0, 1.0], n, p=[0.1, 0.9])
        best = (t_true * (1.0 + rel)).astype(F32)
        culled = cannot_be_nearer(oz, dz, best)
        hit, t = chain_t(oz, dx, dy, dz)
        assert np.isfinite(ln).all()
        bad = culled & selectable(hit, t, best)
        assert not bad.any(), [(oz[i], dx[i], dy[i], dz[i], best[i], t[i]) for i in np.flatnonzero(bad)[:5]]
        # and the cull is worth having: nearly every ray whose crossing lies beyond the margin is culled
        beyond = rel < -(2.0 ** -15)
        assert culled[beyond].mean() > 0.99
        assert not culled[rel > 2.0 ** -15].any()
    finally:
        M.set_contract(prev)


def test_cull_never_removes_a_selectable_hit_on_random_floats():
    """Random bit patterns (every exponent, denormals, NaNs included) for all five inputs."""
    prev = M.set_contract(2)
    try:
        rng = np.random.default_rng(5)
        n = 1_000_000
        bits = lambda: rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(F32)
        oz, dx, dy, dz, best = bits(), bits(), bits(), bits(), bits()
        best = np.where(rng.random(n) < 0.2, F32(np.inf), np.abs(best)).astype(F32)
        culled = cannot_be_nearer(oz, dz, best)
        hit, t = chain_t(oz, dx, dy, dz)
        bad = culled & selectable(hit, t, best)
        assert culled.sum() > 100_000
        assert not bad.any(), [(oz[i], dx[i], dy[i], dz[i], best[i], t[i]) for i in np.flatnonzero(bad)[:5]]
    finally:
        M.set_contract(prev)
