"""The numerics contract (portal_amd/csrc/device/ptl_glsl.h) against its numpy restatement
(oracle/glsl_math.py): bit-identical on the host build, and accurate against binary64."""
import numpy as np
import pytest

from tests import probe


@pytest.fixture(scope="module")
def host_results(pa):
    from oracle import host_build as hb

    samples = probe.inputs()
    hk = hb.HostKernel(probe.source(pa), probe.LAYOUT, probe.BLOCK_SIZE)
    hk.set_texture("in_tex", probe.as_texture(samples))
    hk.set_uniform("n_u", len(samples))
    out = hk.render(len(samples), len(probe.functions()), rgba8=False)["rgba32f"]
    assert np.array_equal(out[0, :, 1].view(np.uint32), samples[:, 0].view(np.uint32))  # inputs arrived intact
    return samples, out[:, :, 0]


def test_contract_cpp_equals_numpy_bit_for_bit(host_results):
    samples, got = host_results
    want = probe.numpy_results(samples)
    for k, (name, _, _) in enumerate(probe.functions()):
        ok = probe.same_bits(got[k], want[k])
        bad = np.nonzero(~ok)[0]
        assert ok.all(), f"{name}: {len(bad)} of {len(ok)} differ, e.g. inputs {samples[bad[0]]} -> c++ {got[k][bad[0]]!r} numpy {want[k][bad[0]]!r}"


def ulp_error(got, exact):
    got64 = got.astype(np.float64)
    spacing = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
    return np.abs(got64 - exact) / np.maximum(spacing, 1e-45)


@pytest.mark.parametrize("name,ref,domain,max_ulp", [
    ("sin", np.sin, (-100.0, 100.0), 2.0), ("cos", np.cos, (-100.0, 100.0), 2.0), ("atan", np.arctan, (-1e6, 1e6), 2.0),
    ("asin", np.arcsin, (-1.0, 1.0), 3.0), ("acos", np.arccos, (-1.0, 1.0), 3.0), ("exp", np.exp, (-80.0, 80.0), 2.0),
    ("log", np.log, (1e-30, 1e30), 2.0), ("exp2", np.exp2, (-120.0, 120.0), 2.0), ("log2", np.log2, (1e-30, 1e30), 3.0),
    ("sqrt", np.sqrt, (0.0, 1e30), 0.5), ("tan", np.tan, (-1.5, 1.5), 4.0),
])
def test_contract_accuracy_against_binary64(name, ref, domain, max_ulp):
    from oracle import glsl_math as M

    rng = np.random.default_rng(5)
    lo, hi = domain
    x = (np.exp(rng.uniform(np.log(lo), np.log(hi), 20000)) if lo > 0 else rng.uniform(lo, hi, 20000)).astype(np.float32)
    got = getattr(M, name)(x)
    err = ulp_error(got, ref(x.astype(np.float64)))
    assert err.max() <= max_ulp, f"{name}: {err.max():.2f} ulp at x={x[err.argmax()]!r}"


def test_contract_exact_identities():
    from oracle import glsl_math as M

    f = np.float32
    assert M.acos(f(-1.0)) == f(np.pi)                 # `#define PI acos(-1.)` (src/library.glsl:15)
    assert M.sin(f(0.0)) == 0 and M.cos(f(0.0)) == 1 and M.atan(f(0.0)) == 0 and M.exp(f(0.0)) == 1 and M.log(f(1.0)) == 0
    assert M.atan2(f(1.0), f(0.0)) == f(np.pi / 2) and M.atan2(f(0.0), f(-1.0)) == f(np.pi)
    assert M.mod(f(5.5), f(2.0)) == f(1.5) and M.mod(f(-0.5), f(2.0)) == f(1.5)   # x - y*floor(x/y)
    assert M.step(f(0.5), f(0.5)) == 1 and M.step(f(0.5), f(0.25)) == 0
    assert M.sign(f(-3.0)) == -1 and M.sign(f(0.0)) == 0
    assert M.exp2(f(10.0)) == 1024 and M.log2(f(1024.0)) == 10 and M.pow(f(2.0), f(0.5)) == pytest.approx(2 ** 0.5, rel=3e-7)
    assert np.isnan(M.fmin(f(np.nan), f(1.0))) and M.fmin(f(1.0), f(np.nan)) == 1   # min(a,b) = b < a ? b : a


def test_fma_emulation_is_a_single_rounding():
    """Cross-check the numpy fma emulation against exact rational arithmetic."""
    from fractions import Fraction

    from oracle import glsl_math as M

    rng = np.random.default_rng(11)
    a = rng.standard_normal(300).astype(np.float32)
    b = rng.standard_normal(300).astype(np.float32)
    c = (-(a.astype(np.float64) * b.astype(np.float64)) * (1 + rng.uniform(-1e-7, 1e-7, 300))).astype(np.float32)  # heavy cancellation
    got = M.fma(a, b, c)
    for k in range(300):
        exact = Fraction(float(a[k])) * Fraction(float(b[k])) + Fraction(float(c[k]))
        near = np.float32(float(exact))  # float(Fraction) rounds correctly to binary64; then to binary32:
        lo, hi = np.nextafter(near, np.float32(-np.inf)), np.nextafter(near, np.float32(np.inf))
        best = min((lo, near, hi), key=lambda v: (abs(Fraction(float(v)) - exact), int(np.float32(v).view(np.uint32)) & 1))
        assert got[k] == best, (a[k], b[k], c[k])
