"""oracle/formula.py -- uniform formulas, restated for the oracle.

TEST INFRASTRUCTURE ONLY (CPU oracle).  Nothing under portal_amd/ may import this.

Follows: the third-party crate `fasteval 0.2.4` (Cargo.lock:636; not vendored in the
reference) as used at src/gui/uniform.rs:602-634 (parse, compile) and :1133-1136 (eval), plus
the reference's own callback table src/gui/uniform.rs:1014-1124 and src/gui/easing.rs:6-45.
fasteval's published grammar: operators, from tightest to loosest binding,
    ^  (right assoc)   %   /   *   -   +   == != < <= >= >  (one level)   &&   ||
Its compiler rewrites  a - b  as  a + (-b)  and  a / b  as  a * (1/b), folds the constants of
an addition / multiplication chain into one constant applied last, and compares == / != with
a tolerance of 8 machine epsilons.  All arithmetic is binary64.
"""
from __future__ import annotations

import math
import re

_TOK = re.compile(r"\s*(?:(?P<num>(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)|(?P<id>[A-Za-z_]\w*)|(?P<op>\|\||&&|==|!=|<=|>=|[-+*/%^<>!(),;\[\]]))")

PREC = {"||": 1, "&&": 2, "!=": 3, "==": 3, ">=": 3, "<=": 3, ">": 3, "<": 3, "+": 4, "-": 5, "*": 6, "/": 7, "%": 8, "^": 9}
SI = {"k": 1e3, "K": 1e3, "M": 1e6, "G": 1e9, "T": 1e12, "m": 1e-3, "u": 1e-6, "n": 1e-9, "p": 1e-12}
EPS8 = 8.0 * 2.220446049250313e-16


class FormulaError(ValueError):
    pass


def _tokens(text):
    pos, out = 0, []
    text = text.rstrip()
    while pos < len(text):
        m = _TOK.match(text, pos)
        if not m or m.end() == pos:
            raise FormulaError(f"bad character at {pos} in {text!r}")
        if m.lastgroup == "num":
            val = float(m.group("num"))
            end = m.end()
            if end < len(text) and text[end] in SI and not (end + 1 < len(text) and (text[end + 1].isalnum() or text[end + 1] == "_")):
                val *= SI[text[end]]
                end += 1
            out.append(("num", val))
            pos = end
            continue
        if m.lastgroup == "id":
            out.append(("id", m.group("id")))
        else:
            out.append(("op", m.group("op")))
        pos = m.end()
    return out


class _Parser:
    """flat expression = value (op value)*, like fasteval's Expression{first, pairs}"""

    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else ("eof", None)

    def expr(self):
        first = self.value()
        pairs = []
        while True:
            kind, op = self.peek()
            if kind == "id" and op in ("and", "or"):  # keyword spellings, only in operator position
                op = "&&" if op == "and" else "||"
            elif not (kind == "op" and op in PREC):
                break
            self.i += 1
            pairs.append((op, self.value()))
        return ("expr", first, pairs)

    def value(self):
        kind, v = self.peek()
        self.i += 1
        if kind == "num":
            return ("const", v)
        if kind == "op" and v in "-+!":
            return ({"-": "neg", "+": "pos", "!": "not"}[v], self.value())
        if kind == "op" and v in "([":
            e = self.expr()
            self.close(")" if v == "(" else "]")
            return e
        if kind == "id":
            if self.peek() in (("op", "("), ("op", "[")):
                close = ")" if self.peek()[1] == "(" else "]"
                self.i += 1
                args = []
                while self.peek() != ("op", close):
                    args.append(self.expr())
                    if self.peek() in (("op", ","), ("op", ";")):
                        self.i += 1
                    elif self.peek() != ("op", close):
                        raise FormulaError("bad argument list")
                self.i += 1
                return ("call", v, args)
            return ("var", v)
        raise FormulaError(f"unexpected token {v!r}")

    def close(self, c):
        if self.peek() != ("op", c):
            raise FormulaError(f"missing {c}")
        self.i += 1


# ---- compile to an instruction tree with fasteval's evaluation order -------------------------
def _is_const(n):
    return n[0] == "const"


def _neg(n):
    return ("const", -n[1]) if _is_const(n) else ("neg", n)


def _inv(n):
    return ("const", _fdiv(1.0, n[1])) if _is_const(n) else ("inv", n)


def _chain(kind, nodes, identity):
    out, folded = None, identity
    for n in nodes:
        if _is_const(n):
            folded = folded + n[1] if kind == "add" else folded * n[1]
        else:
            out = n if out is None else (kind, out, n)
    if out is None:
        return ("const", folded)
    if folded != identity:
        out = (kind, out, ("const", folded))
    return out


def _cmp(op, l, r):
    if op == "==":
        return 1.0 if abs(l - r) <= EPS8 else 0.0
    if op == "!=":
        return 0.0 if abs(l - r) <= EPS8 else 1.0
    return 1.0 if {"<": l < r, ">": l > r, "<=": l <= r, ">=": l >= r}[op] else 0.0


BUILTIN_ARITY = {"int": (1, 1), "ceil": (1, 1), "floor": (1, 1), "abs": (1, 1), "sign": (1, 1), "log": (1, 2), "round": (1, 2), "min": (1, 64), "max": (1, 64),
                 "sin": (1, 1), "cos": (1, 1), "tan": (1, 1), "asin": (1, 1), "acos": (1, 1), "atan": (1, 1), "sinh": (1, 1), "cosh": (1, 1), "tanh": (1, 1),
                 "asinh": (1, 1), "acosh": (1, 1), "atanh": (1, 1)}


# ---- IEEE binary64 semantics where Python raises instead (Rust, like C, returns inf / NaN) ----------------------
def _fdiv(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0 or a != a:
            return float("nan")
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def _fmod(a, b):
    """C fmod() / Rust's f64 `%`: NaN for an infinite dividend or a zero divisor, where Python raises."""
    try:
        return math.fmod(a, b)
    except ValueError:
        return float("nan")


def _ln(x, fn=math.log):
    if x != x or x < 0:
        return float("nan")
    if x == 0:
        return float("-inf")
    return fn(x)


def _is_odd_integer(p):
    return math.isfinite(p) and p == math.floor(p) and math.fmod(p, 2.0) != 0.0


def _libm(name, x):
    """math.<name>(x) with C's results where Python raises: domain errors are NaN, except the poles (atanh(+-1) = +-inf);
    overflow is +-inf with the sign the function has there."""
    try:
        return getattr(math, name)(x)
    except ValueError:
        if name == "atanh" and abs(x) == 1.0:
            return math.copysign(math.inf, x)
        return float("nan")
    except OverflowError:
        return math.copysign(math.inf, x) if name in ("sinh", "tan") else math.inf


def _builtin(name, a):
    if name in ("int", "ceil", "floor"):  # Python's versions return ints: restore the IEEE sign of a zero result (ceil(-0.5) = -0.0)
        if not math.isfinite(a[0]):
            return a[0]
        r = float({"int": math.trunc, "ceil": math.ceil, "floor": math.floor}[name](a[0]))
        return math.copysign(0.0, a[0]) if r == 0.0 else r
    if name == "abs":
        return abs(a[0])
    if name == "sign":
        return a[0] if math.isnan(a[0]) else math.copysign(1.0, a[0])
    if name == "log":
        base, n = (a[0], a[1]) if len(a) == 2 else (10.0, a[0])
        if base == 2.0:
            return _ln(n, math.log2)
        if base == 10.0:
            return _ln(n, math.log10)
        return _fdiv(_ln(n), _ln(base))  # f64::log(self, base) = self.ln() / base.ln()
    if name == "round":
        modulus, n = (a[0], a[1]) if len(a) == 2 else (1.0, a[0])
        q = _fdiv(n, modulus)
        r = math.floor(abs(q) + 0.5) * math.copysign(1.0, q) if math.isfinite(q) else q  # half away from zero
        return r * modulus
    if name == "min":  # f64::min folded left to right: NaN operands are ignored, the earlier argument keeps a +-0 tie
        m = a[0]
        for x in a[1:]:
            m = _fmin(m, x)
        return m
    if name == "max":
        m = a[0]
        for x in a[1:]:
            m = _fmax(m, x)
        return m
    return _libm(name, a[0])


def _compile_value(v):
    k = v[0]
    if k == "const":
        return v
    if k == "expr":
        return _compile_expr(v)
    if k == "pos":
        return _compile_value(v[1])
    if k == "neg":
        return _neg(_compile_value(v[1]))
    if k == "not":
        n = _compile_value(v[1])
        return ("const", 1.0 if abs(n[1]) <= EPS8 else 0.0) if _is_const(n) else ("not", n)
    if k == "var":
        return v
    if k == "call":
        name, args = v[1], [_compile_expr(a) for a in v[2]]
        if name == "pi" and not args:
            return ("const", math.pi)
        if name == "e" and not args:
            return ("const", math.e)
        if name in BUILTIN_ARITY and BUILTIN_ARITY[name][0] <= len(args) <= BUILTIN_ARITY[name][1]:
            if all(_is_const(a) for a in args):
                return ("const", _builtin(name, [a[1] for a in args]))
            return ("builtin", name, args)
        return ("call", name, args)
    raise FormulaError(k)


def _compile_expr(e):
    return _compile_slice(e[1], e[2])


def _compile_slice(first, pairs):
    if not pairs:
        return _compile_value(first)
    lowest = min(PREC[op] for op, _ in pairs)

    def split(pred):
        parts, cur_first, cur, ops = [], first, [], []
        for op, val in pairs:
            if pred(op):
                parts.append((cur_first, cur))
                ops.append(op)
                cur_first, cur = val, []
            else:
                cur.append((op, val))
        parts.append((cur_first, cur))
        return parts, ops

    if lowest == 3:  # comparisons: one level, left to right
        parts, ops = split(lambda o: PREC[o] == 3)
        acc = _compile_slice(*parts[0])
        for op, part in zip(ops, parts[1:]):
            rhs = _compile_slice(*part)
            acc = ("const", _cmp(op, acc[1], rhs[1])) if _is_const(acc) and _is_const(rhs) else ("cmp", op, acc, rhs)
        return acc
    op_txt = next(o for o, p in PREC.items() if p == lowest and o not in ("!=", "==", ">=", "<=", ">", "<"))
    parts, _ = split(lambda o: PREC[o] == lowest)
    nodes = [_compile_slice(*p) for p in parts]
    if op_txt in ("||", "&&"):
        acc = nodes[0]
        for n in nodes[1:]:
            acc = ("or" if op_txt == "||" else "and", acc, n)
        return acc
    if op_txt == "+":
        return _chain("add", nodes, 0.0)
    if op_txt == "-":
        return _chain("add", [nodes[0]] + [_neg(n) for n in nodes[1:]], 0.0)
    if op_txt == "*":
        return _chain("mul", nodes, 1.0)
    if op_txt == "/":
        return _chain("mul", [nodes[0]] + [_inv(n) for n in nodes[1:]], 1.0)
    if op_txt == "%":
        acc = nodes[0]
        for n in nodes[1:]:
            acc = ("const", _fmod(acc[1], n[1])) if _is_const(acc) and _is_const(n) and n[1] != 0 else ("mod", acc, n)
        return acc
    acc = nodes[-1]  # ^ : right to left
    for n in reversed(nodes[:-1]):
        acc = ("const", _pow(n[1], acc[1])) if _is_const(acc) and _is_const(n) else ("exp", n, acc)
    return acc


def _pow(b, p):
    """C pow(): Python raises for the poles and on overflow."""
    try:
        return math.pow(b, p)
    except OverflowError:
        return -math.inf if b < 0 and _is_odd_integer(p) else math.inf
    except ValueError:
        if b == 0 and p < 0:  # pow(+-0, negative) = inf, carrying the zero's sign for odd integer exponents
            return math.copysign(math.inf, b) if _is_odd_integer(p) else math.inf
        return float("nan")  # negative base, non-integer exponent


def compile_formula(text: str):
    p = _Parser(_tokens(text))
    e = p.expr()
    if p.i != len(p.t):
        raise FormulaError(f"trailing input in {text!r}")
    return _compile_expr(e)


_div = _fdiv


def evaluate(node, ns):
    """ns(name, args) -> float | None"""
    k = node[0]
    if k == "const":
        return node[1]
    if k == "neg":
        v = evaluate(node[1], ns)
        return None if v is None else -v
    if k == "inv":
        v = evaluate(node[1], ns)
        return None if v is None else _div(1.0, v)
    if k == "not":
        v = evaluate(node[1], ns)
        return None if v is None else (1.0 if abs(v) <= EPS8 else 0.0)
    if k in ("add", "mul", "mod", "exp"):
        a, b = evaluate(node[1], ns), evaluate(node[2], ns)
        if a is None or b is None:
            return None
        if k == "add":
            return a + b
        if k == "mul":
            return a * b
        if k == "mod":
            return _fmod(a, b)
        return _pow(a, b)
    if k == "cmp":
        a, b = evaluate(node[2], ns), evaluate(node[3], ns)
        return None if a is None or b is None else _cmp(node[1], a, b)
    if k in ("or", "and"):
        a = evaluate(node[1], ns)
        if a is None:
            return None
        zero = abs(a) <= EPS8
        if (k == "or" and not zero) or (k == "and" and zero):
            return a
        return evaluate(node[2], ns)
    if k == "var":
        return ns(node[1], [])
    if k in ("call", "builtin"):
        args = []
        for a in node[2]:
            v = evaluate(a, ns)
            if v is None:
                return None
            args.append(v)
        return _builtin(node[1], args) if k == "builtin" else ns(node[1], args)
    raise FormulaError(k)


def _fmax(a, b):  # f64::max: the other operand if one is NaN; a tie keeps the first
    return b if a != a else (b if b > a else a)


def _fmin(a, b):
    return b if a != a else (b if b < a else a)


# ---- the reference's callback table (src/gui/uniform.rs:1014-1124) ----------------------------
def _is1(v):
    return abs(v - 1.0) < 1e-6


def _easing_in(t):
    return 1.0 - _libm('cos', t * math.pi * 0.5)


def _easing_in_out(t):
    return (1.0 - _libm('cos', t * math.pi)) * 0.5


def custom_function(name, a):
    """Returns (known, value|None)."""
    try:
        if name == "if":
            return True, (a[1] if _is1(a[0]) else a[2])
        if name == "and":
            return True, (1.0 if _is1(a[0]) and _is1(a[1]) else 0.0)
        if name == "or":
            return True, (1.0 if _is1(a[0]) or _is1(a[1]) else 0.0)
        if name == "not":
            return True, (0.0 if _is1(a[0]) else 1.0)
        if name == "deg2rad":
            return True, a[0] / 180.0 * math.pi
        if name == "rad2deg":
            return True, a[0] * 180.0 / math.pi
        if name == "switch":
            k = (len(a) if a[0] >= 1e18 else int(a[0])) if a[0] > 0 else 0  # Rust `as usize`: NaN and negatives are 0, huge saturates
            return True, (a[k] if k < len(a) else None)
        if name == "on":
            v, lo, hi = a[0], a[1], a[2]
            return True, (0.0 if v < lo else 1.0 if v > hi else _fdiv(v - lo, hi - lo))
        if name == "inv":
            return True, 1.0 - a[0]
        if name == "sqrt":
            return True, (math.sqrt(a[0]) if a[0] >= 0 else float("nan"))
        if name == "atan2":
            return True, math.atan2(a[0], a[1])
        if name == "easing_linear":
            return True, a[0]
        if name == "easing_in":
            return True, _easing_in(a[0])
        if name == "easing_out":
            return True, 1.0 - _easing_in(1.0 - a[0])
        if name == "easing_in_out":
            return True, _easing_in_out(a[0])
        if name == "easing_in_out_fast":
            return True, _easing_in_out(_easing_in_out(a[0]))
        if name == "easing_plus_minus":
            t = a[0] * (2.0 * math.pi)
            t2 = 2.0 * t
            return True, _libm('sin', t) * (3.0 - _libm('cos', t) - _libm('cos', t2) - _libm('cos', t) * _libm('cos', t2)) / 4.0
        if name == "easing_elastic_out":
            x = a[0]
            c4 = (2.0 * math.pi) / 3.0
            return True, (0.0 if x == 0.0 else 1.0 if x == 1.0 else _pow(2.0, -10.0 * x) * _libm('sin', (x * 10.0 - 0.75) * c4) + 1.0)
        if name == "bump":
            x = _fdiv(a[0] - a[1], a[2])
            return True, (0.5 * (1.0 + _libm('cos', math.pi * x)) if abs(x) < 1.0 else 0.0)
        if name == "later_start":
            t, time = a[0], 1.0 - a[1]
            return True, _fmax(0.0, _fdiv(t, time) - _fdiv(1.0 - time, time))
        if name == "early_finish":
            return True, _fmin(1.0, _fdiv(a[0], a[1]))
        if name == "lerp":
            return True, (1.0 - a[2]) * a[0] + a[2] * a[1]
    except IndexError:
        return True, None
    return False, None
