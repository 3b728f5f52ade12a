"""fasteval 0.2.4 (Cargo.lock:636 of the reference; call sites /root/reference/src/gui/uniform.rs:602-634,1009-1140), narrowed by a table.

The crate is not vendored and there is no Rust toolchain, so its parser / evaluator is RESTATED twice -- portal_amd/csrc/host/formula.cpp (product) and
oracle/formula.py (checker) -- and cross-fuzzed (tests/test_formula_fuzz.py).  Both were written by the same hand, so a shared misreading survives
the fuzz.  This table is a third statement that shares no code with either: every expected value below is spelled as the plain binary64 arithmetic the
crate's PUBLISHED rules prescribe for that string, derived by hand:

  R1  precedence, tightest first, one level each:  ^   %   /   *   -   +   (== != < <= >= >)   && (and)   || (or)        [README "Operators"]
  R2  ^ groups right to left, every other level left to right; unary - + ! bind tighter than any binary operator (`-2^2` is `(-2)^2`)
  R3  comparisons give 1.0 / 0.0; == and != hold within 8 machine epsilons of the difference; `!x` is 1.0 when |x| <= 8 eps, else 0.0
  R4  `a && b`: a when a is zero, else b;  `a || b`: a when a is non-zero, else b  (the VALUES, short-circuit)
  R5  the compiled form (the reference compiles: uniform.rs:628-634): a - b = a + (-b), a / b = a * (1 / b); % is Rust's f64 `%` (fmod: the sign of
      the dividend); ^ is powf; the constants of a + / * chain fold into ONE constant that is applied last
  R6  literals: decimal with exponent, SI suffixes K k M G T (1e3 1e3 1e6 1e9 1e12) and m u n p (1e-3 1e-6 1e-9 1e-12); [ ] group like ( )
  R7  builtins: e() pi() int ceil floor abs sign log(base?, x) round(modulus?, x) min(..) max(..) sin cos tan asin acos atan sinh cosh tanh
      (log with one argument is base 10; round halves away from zero; min / max take one or more arguments)
The README's own showcase line (R1-R7 at once) is the first entry.  Where a rule above is itself a recollection of the published text rather than a
quotation (no network here), the header of oracle/formula.py says the same and DESIGN.md lists fasteval as "restated, narrowed, not pinned".
"""
import math

import pytest

EPS8 = 8.0 * 2.220446049250313e-16
X, Y, Z, T = 0.3, -1.75, 4.0, 2.5
VARS = {"x": X, "y": Y, "z": Z, "t": T}


def inv(v):
    return 1.0 / v


def fmod(a, b):
    return math.fmod(a, b)


def b2f(c):
    return 1.0 if c else 0.0


TABLE = [
    # --- the README's showcase
    ("1+2*3/4^5%6 + log(100K) + log(e(),100) + [3*(3-3)/3] + (2<3) && 1.23", 1.23),
    # --- R1: one level per operator
    ("2 * 3 / 4", 2.0 * (3.0 * inv(4.0))), ("8 / 4 * 2", (8.0 * inv(4.0)) * 2.0), ("2 * 10 % 4", 2.0 * fmod(10.0, 4.0)), ("10 % 4 * 3", fmod(10.0, 4.0) * 3.0),
    ("10 / 4 % 3", 10.0 * inv(fmod(4.0, 3.0))), ("10 % 4 / 3", fmod(10.0, 4.0) * inv(3.0)), ("2 + 3 * 4", 2.0 + 12.0), ("2 * 3 + 4", 6.0 + 4.0), ("2 - 3 * 4", 2.0 + -(12.0)),
    ("2 ^ 3 * 4", 8.0 * 4.0), ("2 * 3 ^ 2", 2.0 * 9.0), ("2 ^ 3 % 5", fmod(8.0, 5.0)), ("7 % 2 ^ 2", fmod(7.0, 4.0)), ("12 / 2 ^ 2", 12.0 * inv(4.0)), ("1 + 2 - 3 + 4", (1.0 + (2.0 + -3.0)) + 4.0),
    ("10 - 4 - 3", (10.0 + -4.0) + -3.0), ("10 - 4 + 3", (10.0 + -4.0) + 3.0), ("10 + 4 - 3", 10.0 + (4.0 + -3.0)), ("1 + 2 < 4", b2f(3.0 < 4.0)), ("4 > 1 + 2", b2f(4.0 > 3.0)),
    ("1 < 2 && 3", 3.0), ("0 && 1 || 1", 1.0), ("1 || 0 && 0", 1.0), ("0 || 0 && 5", 0.0), ("2 && 0 || 7", 7.0), ("1 + 1 && 0", 0.0), ("0 * 5 || 2 + 2", 4.0),
    ("5 / 3", 5.0 * inv(3.0)), ("7 / 3 / 2", (7.0 * inv(3.0)) * inv(2.0)), ("1 / 3 * 3", (1.0 * inv(3.0)) * 3.0), ("100 / 7 % 4", 100.0 * inv(fmod(7.0, 4.0))), ("2 * 3 % 4 / 5", 2.0 * (fmod(3.0, 4.0) * inv(5.0))),
    # --- R2: associativity and the unary operators
    ("2^3^2", 2.0 ** (3.0 ** 2.0)), ("2 ^ 0.5 ^ 2", 2.0 ** (0.5 ** 2.0)), ("4^3^2^0", 4.0 ** (3.0 ** (2.0 ** 0.0))), ("-2^2", (-2.0) ** 2.0), ("-2^-2", (-2.0) ** -2.0), ("2^-2^2", 2.0 ** ((-2.0) ** 2.0)),
    ("-(2)^2", (-2.0) ** 2.0), ("-(2^2)", -(2.0 ** 2.0)), ("0-2^2", 0.0 + -(4.0)), ("+2", 2.0), ("--2", 2.0), ("-+-2", 2.0), ("-!0", -1.0), ("!-1", 0.0), ("!!5", 1.0), ("!0 + 1", 2.0), ("3 - -2", 3.0 + 2.0),
    ("3 * -2", -6.0), ("3 ^ -1", 3.0 ** -1.0), ("-3 % 2", fmod(-3.0, 2.0)), ("8 % 5 % 2", fmod(fmod(8.0, 5.0), 2.0)), ("100 / 10 / 5", (100.0 * inv(10.0)) * inv(5.0)), ("2 - 3 - 4 - 5", ((2.0 + -3.0) + -4.0) + -5.0),
    # --- R3: comparisons and not
    ("1 < 2 < 3", b2f(b2f(1.0 < 2.0) < 3.0)), ("3 > 2 > 1", b2f(b2f(3.0 > 2.0) > 1.0)), ("1 == 1 == 1", 1.0), ("2 == 2 == 2", b2f(abs(1.0 - 2.0) <= EPS8)), ("1 != 2", 1.0), ("1 != 1", 0.0),
    ("0.1 + 0.2 == 0.3", 1.0), ("0.1 + 0.2 != 0.3", 0.0), ("1 == 1.000000000000001", b2f(abs(1.0 - 1.000000000000001) <= EPS8)), ("1 == 1.00000000000001", 0.0), ("2 <= 2", 1.0), ("2 >= 3", 0.0),
    ("2 < 2", 0.0), ("-1 < 0", 1.0), ("!0", 1.0), ("!3", 0.0), ("!0.0000000000000001", 1.0), ("!0.00000000000001", 0.0), ("!(1 > 2)", 1.0), ("(1 < 2) + (2 < 3) + (3 < 2)", 2.0), ("1 < 2 == 1", 1.0),
    ("3 > 2 != 0", 1.0), ("1 >= 1 > 0", 1.0), ("5 < 4 <= 0", 1.0),
    # --- R4: the values of && and ||
    ("3 and 0", 0.0), ("0 and 3", 0.0), ("3 and 4", 4.0), ("0 or 7", 7.0), ("5 or 7", 5.0), ("0 or 0", 0.0), ("2 && 3 && 4", 4.0), ("2 && 0 && 4", 0.0), ("0 || 0 || 9", 9.0), ("0 || 6 || 9", 6.0),
    ("-1 && 2", 2.0), ("0.5 || 2", 0.5), ("1 and 2 or 3", 2.0), ("0 and 2 or 3", 3.0), ("1 or 2 and 3", 1.0), ("(1 or 2) and 3", 3.0), ("0 or 2 and 3", 3.0), ("x && y", Y), ("x || y", X), ("(x - x) || z", Z), ("(x - x) && z", 0.0),
    # --- R5: the compiled forms
    ("7 % 3", 1.0), ("-7 % 3", -1.0), ("7 % -3", 1.0), ("-7 % -3", -1.0), ("5.5 % 2", 1.5), ("-5.5 % 2", -1.5), ("1 % 0.3", fmod(1.0, 0.3)), ("z % x", fmod(Z, X)), ("y % z", fmod(Y, Z)), ("10 % 4 % 3", fmod(fmod(10.0, 4.0), 3.0)),
    ("x / y", X * inv(Y)), ("z / 3", Z * inv(3.0)), ("1 / x", 1.0 * inv(X)), ("x / 3 / y", (X * inv(3.0)) * inv(Y)), ("x - y", X + -Y), ("x - y - z", (X + -Y) + -Z), ("x * y / z", X * (Y * inv(Z))), ("x / y * z", (X * inv(Y)) * Z),
    ("2 ^ 0.5", math.sqrt(2.0) if False else 2.0 ** 0.5), ("x ^ 2", X ** 2.0), ("z ^ 0.5", 2.0), ("z ^ -0.5", 0.5), ("2 ^ 10", 1024.0), ("(-8) ^ 3", -512.0), ("x ^ y", X ** Y), ("t ^ t", T ** T), ("z ^ x ^ 2", Z ** (X ** 2.0)),
    ("x + 1 + 2", X + 3.0), ("1 + x + 2", X + 3.0), ("2 * x * 3", X * 6.0), ("x * 2 * 3", X * 6.0), ("0.1 + x + 0.2", X + (0.1 + 0.2)), ("3 * x * (1/3)", X * (3.0 * (1.0 * inv(3.0)))), ("x + y + 1", (X + Y) + 1.0),
    ("1 + x + y", (X + Y) + 1.0), ("x - 1", X + -1.0), ("1 - x", -X + 1.0), ("x / 2", X * 0.5), ("2 / x", inv(X) * 2.0), ("x / 3", X * inv(3.0)), ("(x + 1) * (y - 2)", (X + 1.0) * (Y + -2.0)),
    # --- R6: literals and brackets
    ("1.5k", 1500.0), ("1.5K", 1500.0), ("2M", 2e6), ("3G", 3e9), ("4T", 4e12), ("5m", 5e-3), ("6u", 6e-6), ("7n", 7e-9), ("8p", 8e-12), ("1e3", 1000.0), ("1.5e-3", 0.0015), ("2E2", 200.0), (".5", 0.5), ("5.", 5.0),
    ("100K / 1M", 1e5 * inv(1e6)), ("[1 + 2] * 3", 9.0), ("[(1 + 2)] * [3]", 9.0), ("2 * [3 + [4 * 5]]", 46.0), ("1e-2 + 1m", 0.01 + 0.001), ("1k ^ 2", 1e6), ("0.25", 0.25), ("007", 7.0), ("1e0", 1.0),
    # --- R7: builtins
    ("e()", math.e), ("pi()", math.pi), ("2 * pi()", 2.0 * math.pi), ("e() ^ 2", math.e ** 2.0), ("int(2.7)", 2.0), ("int(-2.7)", -2.0), ("ceil(2.1)", 3.0), ("ceil(-2.1)", -2.0), ("floor(2.9)", 2.0), ("floor(-2.1)", -3.0),
    ("abs(-3)", 3.0), ("abs(y)", -Y), ("sign(-3)", -1.0), ("sign(4)", 1.0), ("log(100)", 2.0), ("log(1000)", math.log10(1000.0)), ("log(2, 8)", math.log(8.0) / math.log(2.0) if False else math.log(8.0, 2.0)),
    ("log(e(), 100)", math.log(100.0)), ("round(2.5)", 3.0), ("round(-2.5)", -3.0), ("round(2.4)", 2.0), ("round(0.5, 2.3)", 2.5), ("round(10, 37)", 40.0), ("min(3, 1, 2)", 1.0), ("max(3, 1, 2)", 3.0), ("min(5)", 5.0),
    ("max(x, y, z)", Z), ("min(x, y, z)", Y), ("sin(0)", 0.0), ("cos(0)", 1.0), ("sin(pi() / 2)", math.sin(math.pi * 0.5)), ("cos(pi())", math.cos(math.pi)), ("tan(0.5)", math.tan(0.5)), ("asin(0.5)", math.asin(0.5)),
    ("acos(0.5)", math.acos(0.5)), ("atan(2)", math.atan(2.0)), ("sinh(1)", math.sinh(1.0)), ("cosh(1)", math.cosh(1.0)), ("tanh(1)", math.tanh(1.0)), ("sin(x) ^ 2 + cos(x) ^ 2", math.sin(X) ** 2.0 + math.cos(X) ** 2.0),
    ("sin(x * 2)", math.sin(X * 2.0)), ("cos(-y)", math.cos(-Y)), ("abs(x - z)", abs(X + -Z)), ("floor(z / 3)", math.floor(Z * inv(3.0))), ("max(1, 2) * min(3, 4)", 6.0), ("-abs(-2)", -2.0), ("abs(-2) ^ 2", 4.0),
    # --- formulas of the reference's own scenes, by the same rules (scenes/portal_in_portal.ron:332 and friends; x, y, z, t stand for their uniforms)
    ("(-(2^0.5)/2*(1-x)-z/2)", ((-(2.0 ** 0.5)) * (inv(2.0) * (1.0 + -X))) + -(Z * inv(2.0))),
    ("1 - x * 2", -(X * 2.0) + 1.0), ("(1 - t) * x + t * y", ((-T + 1.0) * X) + (T * Y)), ("x * (1 - t) + y * t", (X * (-T + 1.0)) + (Y * T)), ("pi() / 2 * x", (math.pi * inv(2.0)) * X),
    ("2 * pi() * t / 4", 2.0 * (math.pi * (T * inv(4.0)))), ("sin(t * pi() / 2)", math.sin(T * (math.pi * inv(2.0)))), ("z / 2 - x", (Z * inv(2.0)) + -X), ("-z / 2", (-Z) * inv(2.0)), ("x < 0.5 && t > 2", b2f(T > 2.0)),
    ("(x > 1) * 3 + (x <= 1) * 5", (0.0 * 3.0) + (1.0 * 5.0)), ("max(0, min(1, (t - 2) / 2))", max(0.0, min(1.0, (T + -2.0) * inv(2.0)))), ("abs(y) % 1", fmod(-Y, 1.0)), ("z ^ 2 - 4 * x * y", (Z ** 2.0) + -(4.0 * X * Y) if False else (Z ** 2.0) + -((X * Y) * 4.0)),
    ("(-z + (z^2 - 4*x*y)^0.5) / (2*x)", (-Z + ((Z ** 2.0) + -((X * Y) * 4.0)) ** 0.5) * inv(X * 2.0)),
]


def _same(got, want):
    if got is None or want is None:
        return got is want
    return got == want or (math.isnan(got) and math.isnan(want)) or abs(got - want) <= 1e-15 * max(1.0, abs(want))


def test_the_table_is_large_and_every_entry_is_distinct():
    assert len(TABLE) >= 200 and len({t for t, _ in TABLE}) == len(TABLE)


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_the_published_rules_hold_in_both_restatements(pa, which):
    """Values to 1e-15 relative (most are exact; a libm call may differ from Python's in the last bit)."""
    from oracle import formula as OF

    wrong = []
    for text, want in TABLE:
        if which == "product":
            got = pa.formula_eval(text, VARS)
        else:
            got = OF.evaluate(OF.compile_formula(text), lambda name, args: VARS.get(name) if not args else None)
        if not _same(got, want):
            wrong.append((text, got, want))
    assert not wrong, wrong[:10]


@pytest.mark.parametrize("text", ["2 (3)", "1 +", "(1 + 2", "1 + * 2", "sin()", "[1 + 2)", "2 ** 3", "", "1 2", "x y", "1 = 1"])
def test_malformed_formulas_are_errors_in_both(pa, text):
    from oracle import formula as OF

    assert pa.formula_eval(text, VARS) is None
    try:   # (the oracle refuses at parse time or yields no value at evaluation)
        assert OF.evaluate(OF.compile_formula(text), lambda name, args: VARS.get(name) if not args else None) is None
    except AssertionError:
        raise
    except Exception:
        pass
