"""Differential fuzz of the two independent restatements of fasteval 0.2.4 + the reference's custom function table:
the product's C++ evaluator (portal_amd/csrc/host/formula.cpp, through the C ABI) against the oracle's Python one
(oracle/formula.py).  Random expressions over the grammar the scene corpus uses; results must agree to the last bit of
binary64 (NaN == NaN), or both must refuse the expression."""
import math
import os
import struct

import pytest

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, example, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

NUMBERS = ["0", "1", "2", "3", "0.5", "1.5", "10", "0.25", "180", "1e-3", "2.5e2", "7", "0.1", "1e10"]
NAMES = ["a", "b", "c", "time"]
UNARY = ["sin", "cos", "tan", "abs", "floor", "ceil", "round", "int", "sign", "sqrt", "deg2rad", "rad2deg", "inv", "not", "asin", "acos", "atan", "log",
         "easing_in", "easing_out", "easing_in_out", "easing_in_out_fast", "easing_plus_minus", "easing_elastic_out", "easing_linear"]
BINARY_FN = ["min", "max", "atan2", "and", "or", "early_finish", "later_start", "log", "round"]
TERNARY_FN = ["if", "on", "lerp", "bump", "switch"]
OPS = ["+", "-", "*", "/", "^", "%", "<", "<=", ">", ">=", "==", "!=", "&&", "||"]


def expressions():
    atom = st.one_of(st.sampled_from(NUMBERS), st.sampled_from(NAMES))

    def extend(children):
        return st.one_of(
            st.tuples(children, st.sampled_from(OPS), children).map(lambda t: f"{t[0]} {t[1]} {t[2]}"),
            st.tuples(children, st.sampled_from(OPS), children).map(lambda t: f"({t[0]}){t[1]}({t[2]})"),
            children.map(lambda e: f"-{e}"),
            children.map(lambda e: f"-({e})"),
            children.map(lambda e: f"!({e})"),
            children.map(lambda e: f"({e})"),
            st.tuples(st.sampled_from(UNARY), children).map(lambda t: f"{t[0]}({t[1]})"),
            st.tuples(st.sampled_from(BINARY_FN), children, children).map(lambda t: f"{t[0]}({t[1]}, {t[2]})"),
            st.tuples(st.sampled_from(TERNARY_FN), children, children, children).map(lambda t: f"{t[0]}({t[1]}, {t[2]}, {t[3]})"),
        )

    return st.recursive(atom, extend, max_leaves=12)


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


# In-suite runs are DETERMINISTIC (derandomize: the examples derive from the test's source, no example database): the suite's colour must
# not depend on a random draw.  The random hunt is a tool: `PTL_FUZZ_RANDOM=1 pytest tests/test_formula_fuzz.py`.  Cases a hunt found stay
# below as fixed examples (round 4: the ORACLE crashed on fmod(inf, 1) while folding `%`; Rust's f64 `%` gives NaN).
@settings(max_examples=3000, deadline=None, suppress_health_check=[HealthCheck.too_slow], derandomize=not os.environ.get("PTL_FUZZ_RANDOM"),
          database=None)
@example(text="-2 ^ 1e10 % 1", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="0 % 0", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="2 ^ 1e5", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="log(-1)", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="log(0)", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="a % (1e10 ^ 1e10)", a=2.0, b=0.0, c=0.0, time=0.0)
@example(text="(1e10 ^ 1e10) % a", a=2.0, b=0.0, c=0.0, time=0.0)
@example(text="(0 - 2) ^ 0.5", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="0 ^ (0 - 1)", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="sqrt(0 - 1) + tan(1e10 ^ 1e10)", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="round(0, 7) + int(1e10 ^ 1e10) + floor(0 / 0)", a=0.0, b=0.0, c=0.0, time=0.0)
@example(text="asin(2) + acos(a) + log(a, b)", a=-2.5, b=0.0, c=0.0, time=0.0)
@example(text="switch(1e10 ^ 1e10, a, b) + switch(0 / 0, a, b)", a=1.0, b=3.0, c=0.0, time=0.0)
@given(text=expressions(), a=st.sampled_from([0.0, 1.0, -1.0, 0.3, 2.0, -2.5, 1e-9, 1e9]), b=st.sampled_from([0.0, 1.0, 0.5, -0.75, 3.0, 90.0]),
       c=st.sampled_from([0.0, 1.0, 2.0, -1.0, 0.125]), time=st.sampled_from([0.0, 0.25, 0.5, 1.0]))
def test_product_and_oracle_formula_evaluators_agree(pa, text, a, b, c, time):
    from oracle import formula as OF

    variables = {"a": a, "b": b, "c": c}
    got = pa.formula_eval(text, variables, time=time)
    try:
        node = OF.compile_formula(text)
    except OF.FormulaError:
        node = None

    def ns(name, args):
        known, v = OF.custom_function(name, args)
        if known:
            return v
        return time if name in ("time", "total_time") else variables.get(name)

    want = OF.evaluate(node, ns) if node is not None else None
    if got is None or want is None:
        assert got is None and want is None, (text, got, want)
    else:
        assert bits(got) == bits(want) or (math.isnan(got) and math.isnan(want)), (text, got, want)
