"""Differential fuzz of the RON legs: random documents (structs, tuples, newtypes, units, lists, maps, options, numbers, plain
and raw strings, comments, arbitrary whitespace) are read by the product's reader and written by its writer
(portal_amd/csrc/host/ron.cpp through ptl_ron_format); the result must mean the same to the oracle's independent reader
(oracle/ron.py), and writing must be idempotent."""
import math
import os

import pytest

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

IDENT = st.sampled_from(["a", "b1", "name", "data", "Some_thing", "x", "look_at", "Kind", "Float", "Simple", "Portal"])
WS = st.sampled_from(["", " ", "\n", "  \n\t", " // note\n", " /* c */ "])


def documents():
    floats = st.floats(allow_nan=False, allow_infinity=False, width=64).map(lambda f: repr(f) if ("e" not in repr(f) and "." in repr(f)) else f"{f:.17g}" + ("" if "." in f"{f:.17g}" or "e" in f"{f:.17g}" else ".0"))
    plain = st.text(alphabet=st.sampled_from(list("abc XYZ_0123.,:;(){}[]#'\n\t«»é")), max_size=12).map(lambda s: '"' + s + '"')
    raw = st.text(alphabet=st.sampled_from(list('ab"\\/# \n')), max_size=10).filter(lambda s: '"##' not in s).map(lambda s: 'r##"' + s + '"##')
    atom = st.one_of(st.integers(-10**9, 10**9).map(str), floats, st.sampled_from(["true", "false", "None", "Dev", "Normal"]), plain, raw)

    def extend(children):
        sep = lambda parts, ws: ("," + ws).join(parts)
        return st.one_of(
            st.tuples(st.lists(children, max_size=4), WS).map(lambda t: "[" + t[1] + sep(t[0], t[1]) + ("," if t[0] else "") + t[1] + "]"),
            st.tuples(st.lists(st.tuples(plain, children), max_size=3), WS).map(lambda t: "{" + t[1] + sep([k + ":" + t[1] + v for k, v in t[0]], t[1]) + t[1] + "}"),
            st.tuples(st.sampled_from(["", "Name", "Float", "Some"]), st.lists(children, min_size=1, max_size=4), WS).map(lambda t: t[0] + "(" + t[2] + sep(t[1], t[2]) + t[2] + ")"),
            st.tuples(st.sampled_from(["", "Simple", "Data"]), st.lists(st.tuples(IDENT, children), min_size=1, max_size=4, unique_by=lambda kv: kv[0]), WS).map(
                lambda t: t[0] + "(" + t[2] + sep([k + ":" + t[2] + v for k, v in t[1]], t[2]) + "," + t[2] + ")"),
        )

    return st.recursive(atom, extend, max_leaves=14)


def same(a, b):
    from oracle import ron

    if isinstance(a, float) and isinstance(b, float):
        return a == b or (math.isnan(a) and math.isnan(b))
    if type(a) is not type(b):
        return False
    if isinstance(a, ron.Struct):
        return a.name == b.name and list(a.fields) == list(b.fields) and all(same(a.fields[k], b.fields[k]) for k in a.fields)
    if isinstance(a, ron.Tuple):
        return a.name == b.name and len(a.items) == len(b.items) and all(same(x, y) for x, y in zip(a.items, b.items))
    if isinstance(a, ron.Unit):
        return a.name == b.name
    if isinstance(a, list):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return list(a) == list(b) and all(same(a[k], b[k]) for k in a)
    return a == b


@settings(max_examples=800, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.filter_too_much],
          derandomize=not os.environ.get("PTL_FUZZ_RANDOM"), database=None)  # deterministic in the suite; PTL_FUZZ_RANDOM=1 hunts
@given(text=documents())
def test_product_ron_writer_preserves_what_the_oracle_reads(pa, text):
    from oracle import ron

    try:
        want = ron.loads(text)
    except Exception:
        want = None
    try:
        written = pa.ron_format(text)
    except pa.PortalError:
        written = None
    if want is None or written is None:
        assert want is None and written is None, (text, want, written)
        return
    assert same(ron.loads(written), want), (text, written)
    assert pa.ron_format(written) == written, (text, written)
