"""oracle/glsl_interp.py -- parser + SIMT interpreter for the GLSL ES 3.00 subset that scene
files embed (is_inside / intersect / material / library snippets).

TEST INFRASTRUCTURE ONLY (CPU oracle).  Nothing under portal_amd/ may import this.

The reference hands these snippets to the GL driver's GLSL compiler
(src/gui/scene.rs:776,865,877,1020,1041).  The MI355X build rewrites them lexically into C++
(portal_amd/csrc/host/glsl_translate.cpp).  The oracle does neither: it parses the GLSL as
GLSL and executes it on numpy lanes under an execution mask (if/else, loops, early return,
break/continue all per lane), so that a translation bug in the product cannot hide.
"""
from __future__ import annotations

import re

import numpy as np

from . import glsl_math as M
from . import glsl_values as V
from .glsl_values import Mat, Sampler, Struct, Vec

F32, I32 = np.float32, np.int32


class GlslError(Exception):
    pass


# =============================================================================================
# tokenizer
# =============================================================================================
_TOKEN = re.compile(
    r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>(?:0[xX][0-9a-fA-F]+[uU]?)|(?:(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fFuU]?))
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op>\+\+|--|<<|>>|<=|>=|==|!=|&&|\|\||\^\^|\+=|-=|\*=|/=|%=|[-+*/%<>=!&|^~?:;,.(){}\[\]])
    """,
    re.X | re.S,
)


def tokenize(src: str):
    out, pos, line = [], 0, 1
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise GlslError(f"line {line}: cannot tokenize {src[pos:pos+20]!r}")
        text = m.group(0)
        if m.lastgroup != "ws":
            out.append((m.lastgroup, text, line))
        line += text.count("\n")
        pos = m.end()
    out.append(("eof", "", line))
    return out


TYPE_NAMES = {"void", "float", "int", "bool", "vec2", "vec3", "vec4", "mat2", "mat3", "mat4", "sampler2D", "uint"}
QUALIFIERS = {"const", "in", "out", "inout", "highp", "mediump", "lowp", "uniform"}


# =============================================================================================
# parser (AST = nested tuples)
# =============================================================================================
class Parser:
    def __init__(self, src: str, known_types=()):
        self.t = tokenize(src)
        self.i = 0
        self.types = set(TYPE_NAMES) | set(known_types)

    # -- helpers
    def peek(self, k=0):
        return self.t[min(self.i + k, len(self.t) - 1)]

    def at(self, text):
        return self.peek()[1] == text and self.peek()[0] != "eof"

    def eat(self, text=None):
        tok = self.peek()
        if text is not None and tok[1] != text:
            raise GlslError(f"line {tok[2]}: expected {text!r}, found {tok[1]!r}")
        self.i += 1
        return tok

    def is_type(self, tok):
        return tok[0] == "id" and tok[1] in self.types

    # -- top level
    def parse_unit(self):
        items = []
        while self.peek()[0] != "eof":
            if self.at(";"):
                self.eat()
                continue
            items.append(self.parse_external())
        return items

    def parse_external(self):
        if self.at("struct"):
            return self.parse_struct()
        quals = []
        while self.peek()[1] in QUALIFIERS or self.at("layout"):
            if self.at("layout"):  # layout(location=0) out vec4 FragColor;  (src/frag.glsl:286)
                self.eat()
                self.eat("(")
                while not self.at(")"):
                    self.eat()
                self.eat(")")
                continue
            quals.append(self.eat()[1])
        ty = self.parse_type()
        name = self.eat()
        if name[0] != "id":
            raise GlslError(f"line {name[2]}: expected identifier")
        if self.at("("):
            return self.parse_function(ty, name[1])
        init = None
        if self.at("="):
            self.eat()
            init = self.parse_assignment()
        self.eat(";")
        return ("global", ty, name[1], init, tuple(quals))

    def parse_struct(self):
        self.eat("struct")
        name = self.eat()[1]
        self.eat("{")
        fields = []
        while not self.at("}"):
            ty = self.parse_type()
            while True:
                fields.append((ty, self.eat()[1]))
                if self.at(","):
                    self.eat()
                    continue
                break
            self.eat(";")
        self.eat("}")
        self.eat(";")
        self.types.add(name)
        return ("struct", name, fields)

    def parse_type(self):
        tok = self.eat()
        if not self.is_type(tok):
            raise GlslError(f"line {tok[2]}: expected a type, found {tok[1]!r}")
        return tok[1]

    def parse_function(self, ret, name):
        self.eat("(")
        params = []
        if self.at("void") and self.peek(1)[1] == ")":
            self.eat()
        while not self.at(")"):
            qual = "in"
            while self.peek()[1] in QUALIFIERS:
                q = self.eat()[1]
                if q in ("in", "out", "inout"):
                    qual = q
            ty = self.parse_type()
            pname = self.eat()[1]
            params.append((qual, ty, pname))
            if self.at(","):
                self.eat()
        self.eat(")")
        body = self.parse_block()
        return ("func", ret, name, params, body)

    # -- statements
    def parse_block(self):
        self.eat("{")
        stmts = []
        while not self.at("}"):
            stmts.append(self.parse_statement())
        self.eat("}")
        return ("block", stmts)

    def parse_body_until_eof(self):
        """A function body given without braces (scene snippets are stored that way)."""
        stmts = []
        while self.peek()[0] != "eof":
            stmts.append(self.parse_statement())
        return ("block", stmts)

    def parse_statement(self):
        tok = self.peek()
        if tok[1] == "{" and tok[0] == "op":
            return self.parse_block()
        if tok[1] == ";":
            self.eat()
            return ("block", [])
        if tok[1] == "if":
            self.eat()
            self.eat("(")
            cond = self.parse_expression()
            self.eat(")")
            then = self.parse_statement()
            els = None
            if self.at("else"):
                self.eat()
                els = self.parse_statement()
            return ("if", cond, then, els)
        if tok[1] == "for":
            self.eat()
            self.eat("(")
            init = None if self.at(";") else self.parse_simple_statement()
            if init is None:
                self.eat(";")
            cond = None if self.at(";") else self.parse_expression()
            self.eat(";")
            step = None if self.at(")") else self.parse_expression()
            self.eat(")")
            return ("for", init, cond, step, self.parse_statement())
        if tok[1] == "while":
            self.eat()
            self.eat("(")
            cond = self.parse_expression()
            self.eat(")")
            return ("for", None, cond, None, self.parse_statement())
        if tok[1] == "return":
            self.eat()
            e = None if self.at(";") else self.parse_expression()
            self.eat(";")
            return ("return", e)
        if tok[1] == "break":
            self.eat()
            self.eat(";")
            return ("break",)
        if tok[1] == "continue":
            self.eat()
            self.eat(";")
            return ("continue",)
        return self.parse_simple_statement()

    def parse_simple_statement(self):
        """declaration or expression statement, consumes the trailing ';'"""
        save = self.i
        while self.peek()[1] in QUALIFIERS:
            self.eat()
        tok = self.peek()
        if self.is_type(tok) and self.peek(1)[0] == "id":
            ty = self.parse_type()
            decls = []
            while True:
                name = self.eat()[1]
                init = None
                if self.at("="):
                    self.eat()
                    init = self.parse_assignment()
                decls.append((name, init))
                if self.at(","):
                    self.eat()
                    continue
                break
            self.eat(";")
            return ("decl", ty, decls)
        self.i = save
        e = self.parse_expression()
        self.eat(";")
        return ("expr", e)

    # -- expressions (precedence climbing)
    def parse_expression(self):
        e = self.parse_assignment()
        while self.at(","):
            self.eat()
            e = ("comma", e, self.parse_assignment())
        return e

    def parse_assignment(self):
        lhs = self.parse_ternary()
        if self.peek()[1] in ("=", "+=", "-=", "*=", "/=", "%=") and self.peek()[0] == "op":
            op = self.eat()[1]
            rhs = self.parse_assignment()
            return ("assign", op, lhs, rhs)
        return lhs

    def parse_ternary(self):
        c = self.parse_binary(0)
        if self.at("?"):
            self.eat()
            a = self.parse_assignment()
            self.eat(":")
            b = self.parse_assignment()
            return ("ternary", c, a, b)
        return c

    LEVELS = [("||",), ("^^",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def parse_binary(self, level):
        if level == len(self.LEVELS):
            return self.parse_unary()
        lhs = self.parse_binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[level]:
            op = self.eat()[1]
            rhs = self.parse_binary(level + 1)
            lhs = ("binop", op, lhs, rhs)
        return lhs

    def parse_unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("-", "+", "!", "~"):
            self.eat()
            return ("unop", tok[1], self.parse_unary())
        if tok[0] == "op" and tok[1] in ("++", "--"):
            self.eat()
            target = self.parse_unary()
            return ("assign", "+=" if tok[1] == "++" else "-=", target, ("num", 1, False))
        return self.parse_postfix()

    def parse_postfix(self):
        e = self.parse_primary()
        while True:
            if self.at("."):
                self.eat()
                e = ("field", e, self.eat()[1])
            elif self.at("["):
                self.eat()
                idx = self.parse_expression()
                self.eat("]")
                e = ("index", e, idx)
            elif self.peek()[0] == "op" and self.peek()[1] in ("++", "--"):
                op = self.eat()[1]
                e = ("postinc", "+=" if op == "++" else "-=", e)
            else:
                return e

    def parse_primary(self):
        tok = self.eat()
        if tok[0] == "num":
            text = tok[1]
            if text.lower().startswith("0x"):
                return ("num", int(text.rstrip("uU"), 16), False)
            is_float = any(ch in text for ch in ".eE") or text[-1] in "fF"
            if is_float:
                return ("num", M.lit(text), True)
            return ("num", int(text.rstrip("uU")), False)
        if tok[0] == "id":
            if tok[1] in ("true", "false"):
                return ("bool", tok[1] == "true")
            if self.at("("):
                self.eat()
                args = []
                while not self.at(")"):
                    args.append(self.parse_assignment())
                    if self.at(","):
                        self.eat()
                self.eat(")")
                return ("call", tok[1], args, tok[2])
            return ("var", tok[1], tok[2])
        if tok[1] == "(":
            e = self.parse_expression()
            self.eat(")")
            return e
        raise GlslError(f"line {tok[2]}: unexpected token {tok[1]!r}")


# =============================================================================================
# interpreter
# =============================================================================================
class Frame:
    __slots__ = ("scopes", "returned", "ret", "ret_type")

    def __init__(self, n, ret_type):
        self.scopes = [{}]
        self.returned = np.zeros(n, dtype=bool)
        self.ret = None
        self.ret_type = ret_type


class Loop:
    __slots__ = ("broken", "continued")

    def __init__(self, n):
        self.broken = np.zeros(n, dtype=bool)
        self.continued = np.zeros(n, dtype=bool)


MAX_LOOP_TRIPS = 100000


class Interp:
    """Executes parsed GLSL on `n` lanes.  `globals_` maps names to values (uniforms, material
    ids, constants); `natives` maps names to Python callables f(interp, args, mask) -> value."""

    def __init__(self, n: int, program: "Interp | None" = None):
        self.n = n
        if program is not None:  # share the parsed program, run it on a different lane count
            self.globals, self.natives, self.funcs, self.structs = program.globals, program.natives, program.funcs, program.structs
            self.out_globals = program.out_globals
        else:
            self.globals = {}
            self.natives = {}
            self.funcs = {}    # name -> [(param_types, ast)]
            self.structs = {}  # name -> [(type, field)]
            self.out_globals = set()  # `out` globals of a whole shader (FragColor): the only globals code may assign
        self.frames = []
        self.loops = []

    # ---- program loading
    def known_types(self):
        return set(self.structs)

    def load_unit(self, src: str):
        p = Parser(src, self.known_types())
        for item in p.parse_unit():
            self.declare(item)
            if item[0] == "struct":
                p.types.add(item[1])

    def declare(self, item):
        if item[0] == "struct":
            self.structs[item[1]] = item[2]
        elif item[0] == "func":
            self.funcs.setdefault(item[2], []).append((tuple(t for _, t, _ in item[3]), item))
        elif item[0] == "global":
            _, ty, name, init, quals = item
            if "out" in quals:
                self.out_globals.add(name)
            if ("uniform" in quals or "in" in quals) and name in self.globals:
                return  # the host bound a value before the shader text was loaded
            frame = Frame(self.n, ty)
            self.frames.append(frame)
            try:
                val = self.eval(init, np.ones(self.n, bool)) if init is not None else self.default_value(ty)
            finally:
                self.frames.pop()
            val = self.convert(val, ty)
            self.globals[name] = V.take(V.expand(val, self.n), 0)  # globals are uniform: keep them 0-d

    def define_function(self, ret, name, params, body_src):
        """A snippet stored without signature/braces: wrap it (scene.rs:857-877)."""
        p = Parser(body_src, self.known_types())
        body = p.parse_body_until_eof()
        item = ("func", ret, name, [("in", t, n) for t, n in params], body)
        self.declare(item)

    # ---- values
    def default_value(self, ty):
        if ty == "float":
            return F32(0)
        if ty in ("int", "uint"):
            return I32(0)
        if ty == "bool":
            return np.bool_(False)
        if ty.startswith("vec"):
            return Vec([F32(0)] * int(ty[3]))
        if ty.startswith("mat"):
            k = int(ty[3])
            return Mat([Vec([F32(0)] * k) for _ in range(k)])
        if ty in self.structs:
            return Struct(ty, {f: self.default_value(t) for t, f in self.structs[ty]})
        raise GlslError(f"unknown type {ty}")

    def convert(self, v, ty):
        """Implicit conversions at initialisation / argument passing (int -> float only)."""
        if ty == "float" and V.is_int(v):
            return V.to_float(v)
        return v

    def construct(self, ty, args, line):
        if ty == "float":
            return V.to_float(args[0].c[0] if isinstance(args[0], Vec) else args[0]) if not V.is_bool(args[0]) else np.asarray(args[0]).astype(F32)
        if ty in ("int", "uint"):
            return V.to_int(args[0])
        if ty == "bool":
            a = np.asarray(args[0])
            return a if a.dtype == bool else (a != 0)
        if ty.startswith("vec"):
            return V.make_vec(int(ty[3]), args)
        if ty.startswith("mat"):
            return V.make_mat(int(ty[3]), args)
        if ty in self.structs:
            fields = self.structs[ty]
            if len(args) != len(fields):
                raise GlslError(f"line {line}: {ty} constructor needs {len(fields)} arguments")
            return Struct(ty, {f: self.convert(a, t) for (t, f), a in zip(fields, args)})
        raise GlslError(f"line {line}: cannot construct {ty}")

    # ---- variables
    def lookup(self, name, line=0):
        for scope in reversed(self.frames[-1].scopes):
            if name in scope:
                return scope[name]
        if name in self.globals:
            return self.globals[name]
        raise GlslError(f"line {line}: undefined identifier `{name}`")

    def store(self, name, value, mask):
        for scope in reversed(self.frames[-1].scopes):
            if name in scope:
                scope[name] = V.select(mask, value, scope[name])
                return
        raise GlslError(f"assignment to undeclared or global variable `{name}`")

    # ---- masks
    def live(self, mask):
        m = mask & ~self.frames[-1].returned
        if self.loops:
            m = m & ~self.loops[-1].broken & ~self.loops[-1].continued
        return m

    # ---- statements
    def exec(self, st, mask):
        mask = self.live(mask)
        if not mask.any():
            return
        kind = st[0]
        if kind == "block":
            self.frames[-1].scopes.append({})
            try:
                for s in st[1]:
                    self.exec(s, mask)
            finally:
                self.frames[-1].scopes.pop()
        elif kind == "decl":
            _, ty, decls = st
            for name, init in decls:
                M.set_active(mask.sum())
                val = self.convert(self.eval(init, mask), ty) if init is not None else self.default_value(ty)
                self.frames[-1].scopes[-1][name] = V.expand(val, self.n)
        elif kind == "expr":
            M.set_active(mask.sum())
            self.eval(st[1], mask)
        elif kind == "if":
            M.set_active(mask.sum())
            c = np.broadcast_to(np.asarray(self.eval(st[1], mask), bool), (self.n,))
            self.exec(st[2], mask & c)
            if st[3] is not None:
                self.exec(st[3], mask & ~c)
        elif kind == "for":
            self.exec_for(st, mask)
        elif kind == "return":
            frame = self.frames[-1]
            if st[1] is not None:
                M.set_active(mask.sum())
                val = V.expand(self.convert(self.eval(st[1], mask), frame.ret_type), self.n)
                frame.ret = val if frame.ret is None else V.select(mask, val, frame.ret)
            frame.returned = frame.returned | mask
        elif kind == "break":
            self.loops[-1].broken |= mask
        elif kind == "continue":
            self.loops[-1].continued |= mask
        else:
            raise GlslError(f"unknown statement {kind}")

    def exec_for(self, st, mask):
        _, init, cond, step, body = st
        frame = self.frames[-1]
        frame.scopes.append({})
        loop = Loop(self.n)
        try:
            if init is not None:
                self.exec(init, mask)
            in_loop = mask.copy()
            self.loops.append(loop)
            try:
                for _ in range(MAX_LOOP_TRIPS):
                    loop.continued[:] = False
                    active = in_loop & ~loop.broken & ~frame.returned
                    if not active.any():
                        break
                    if cond is not None:
                        M.set_active(active.sum())
                        c = np.broadcast_to(np.asarray(self.eval(cond, active), bool), (self.n,))
                        in_loop = in_loop & (c | ~active)
                        active = active & c
                        if not active.any():
                            break
                    self.exec(body, active)
                    if step is not None:
                        loop.continued[:] = False
                        s_mask = active & ~loop.broken & ~frame.returned
                        if s_mask.any():
                            M.set_active(s_mask.sum())
                            self.eval(step, s_mask)
                else:
                    raise GlslError("loop did not terminate")
            finally:
                self.loops.pop()
        finally:
            frame.scopes.pop()

    # ---- expressions
    def eval(self, e, mask):
        kind = e[0]
        if kind == "num":
            return F32(e[1]) if e[2] else I32(e[1])
        if kind == "bool":
            return np.bool_(e[1])
        if kind == "var":
            return self.lookup(e[1], e[2])
        if kind == "comma":
            self.eval(e[1], mask)
            return self.eval(e[2], mask)
        if kind == "unop":
            v = self.eval(e[2], mask)
            if e[1] == "-":
                return V.neg(v)
            if e[1] == "+":
                return v
            if e[1] == "!":
                return ~np.asarray(v, bool)
            raise GlslError(f"unsupported unary {e[1]}")
        if kind == "binop":
            op = e[1]
            if op in ("&&", "||"):
                # GLSL short-circuits; the right operand is evaluated only on lanes that need it
                a = np.broadcast_to(np.asarray(self.eval(e[2], mask), bool), (self.n,))
                need = mask & (a if op == "&&" else ~a)
                if not need.any():
                    return a.copy()
                saved = M._active
                M.set_active(need.sum())
                b = np.broadcast_to(np.asarray(self.eval(e[3], need), bool), (self.n,))
                M.set_active(saved)
                return (a & b) if op == "&&" else (a | b)
            a, b = self.eval(e[2], mask), self.eval(e[3], mask)
            if op in ("<", ">", "<=", ">=", "==", "!="):
                return V.compare(op, a, b)
            if op == "^^":
                return np.asarray(a, bool) ^ np.asarray(b, bool)
            return V.binop(op, a, b)
        if kind == "ternary":
            c = np.broadcast_to(np.asarray(self.eval(e[1], mask), bool), (self.n,))
            a = self.eval(e[2], mask & c) if (mask & c).any() else None
            b = self.eval(e[3], mask & ~c) if (mask & ~c).any() else None
            if a is None:
                return b
            if b is None:
                return a
            return V.select(c, V.expand(a, self.n), V.expand(b, self.n))
        if kind == "field":
            base = self.eval(e[1], mask)
            return self.get_field(base, e[2])
        if kind == "index":
            base, idx = self.eval(e[1], mask), self.eval(e[2], mask)
            return self.get_index(base, idx)
        if kind == "call":
            return self.call(e[1], [self.eval(a, mask) for a in e[2]], mask, e[3], e[2])
        if kind == "assign":
            return self.assign(e, mask)
        if kind == "postinc":
            old = self.eval(e[2], mask)
            self.assign(("assign", e[1], e[2], ("num", 1, False)), mask)
            return old
        raise GlslError(f"unknown expression {kind}")

    def get_field(self, base, name):
        if isinstance(base, Struct):
            if name not in base.f:
                raise GlslError(f"{base.tname} has no field `{name}`")
            return base.f[name]
        if isinstance(base, Vec):
            idx = V.swizzle_indices(name)
            if idx is None or max(idx) >= base.n:
                raise GlslError(f"bad swizzle .{name} on vec{base.n}")
            return base.c[idx[0]] if len(idx) == 1 else Vec([base.c[i] for i in idx])
        raise GlslError(f"field access .{name} on {V.type_of(base)}")

    def get_index(self, base, idx):
        k = np.asarray(idx)
        if k.ndim != 0 and not (k == k.flat[0]).all():
            raise GlslError("per-lane dynamic indexing is not supported by the oracle")
        k = int(k.flat[0]) if k.ndim else int(k)
        if isinstance(base, Mat):
            return base.cols[k]
        if isinstance(base, Vec):
            return base.c[k]
        raise GlslError(f"indexing {V.type_of(base)}")

    def assign(self, e, mask):
        _, op, target, rhs_e = e
        rhs = self.eval(rhs_e, mask)
        if op != "=":
            cur = self.eval(target, mask)
            rhs = V.binop(op[0], cur, rhs)
        # walk the lvalue: collect the access path down to a variable
        path, node = [], target
        while node[0] in ("field", "index"):
            path.append(node)
            node = node[1]
        if node[0] != "var":
            raise GlslError("assignment to a non-lvalue")
        name = node[1]
        cur = self.lookup(name, node[2])

        def update(value, steps):
            if not steps:
                new = rhs
                if V.is_float(value) and V.is_int(new):
                    new = V.to_float(new)
                return V.select(mask, V.expand(new, self.n), V.expand(value, self.n))
            step = steps[-1]
            if step[0] == "field":
                fname = step[2]
                if isinstance(value, Struct):
                    return value.with_field(fname, update(value.f[fname], steps[:-1]))
                if isinstance(value, Vec):
                    idx = V.swizzle_indices(fname)
                    if idx is None:
                        raise GlslError(f"bad swizzle .{fname}")
                    comps = list(value.c)
                    if len(idx) == 1:
                        comps[idx[0]] = update(comps[idx[0]], steps[:-1])
                    else:
                        if len(steps) != 1:
                            raise GlslError("nested access below a multi-component swizzle")
                        new = rhs if isinstance(rhs, Vec) else V.make_vec(len(idx), [rhs])
                        for j, i in enumerate(idx):
                            comps[i] = np.where(mask, new.c[j], comps[i]).astype(F32)
                    return Vec(comps)
                raise GlslError(f"field assignment on {V.type_of(value)}")
            k = int(np.asarray(self.eval(step[2], mask)).flat[0])
            if isinstance(value, Mat):
                cols = list(value.cols)
                cols[k] = update(cols[k], steps[:-1])
                return Mat(cols)
            if isinstance(value, Vec):
                comps = list(value.c)
                comps[k] = update(comps[k], steps[:-1])
                return Vec(comps)
            raise GlslError(f"index assignment on {V.type_of(value)}")

        new_value = update(cur, path)
        for scope in reversed(self.frames[-1].scopes):
            if name in scope:
                scope[name] = new_value
                return rhs
        if name in self.out_globals:
            self.globals[name] = new_value
            return rhs
        raise GlslError(f"assignment to `{name}` which is not a local variable")

    # ---- calls
    def call(self, name, args, mask, line=0, arg_exprs=None):
        if name in TYPE_NAMES or name in self.structs:
            return self.construct(name, args, line)
        if name in self.funcs:
            return self.call_user(name, args, mask, line, arg_exprs)
        if name in self.natives:
            return self.natives[name](self, args, mask)
        if name in BUILTINS:
            return BUILTINS[name](*args)
        raise GlslError(f"line {line}: unknown function `{name}`")

    def call_user(self, name, args, mask, line=0, arg_exprs=None):
        arg_types = tuple(V.type_of(a) for a in args)
        chosen = None
        for ptypes, item in self.funcs[name]:
            if len(ptypes) != len(args):
                continue
            if all(p == a or (p == "float" and a == "int") for p, a in zip(ptypes, arg_types)):
                chosen = item
                if ptypes == arg_types:
                    break
        if chosen is None:
            raise GlslError(f"line {line}: no overload of `{name}` for ({', '.join(arg_types)})")
        _, ret, _, params, body = chosen
        frame = Frame(self.n, ret)
        for (qual, ty, pname), a in zip(params, args):
            frame.scopes[0][pname] = V.expand(self.convert(a, ty), self.n)
        saved_loops = self.loops
        self.loops = []
        self.frames.append(frame)
        try:
            self.exec(body, mask)
        finally:
            self.frames.pop()
            self.loops = saved_loops
        # copy-out for out / inout parameters
        if arg_exprs is not None:
            for (qual, ty, pname), ex in zip(params, arg_exprs):
                if qual in ("out", "inout"):
                    self.assign(("assign", "=", ex, ("value", frame.scopes[0][pname])), mask)
        M.set_active(self.live(mask).sum() if self.frames else mask.sum())
        if ret == "void":
            return None
        if frame.ret is None:
            return self.default_value(ret)
        return frame.ret

    def run_function(self, name, args, mask=None):
        """Entry point for the host: call a user function on all (or masked) lanes."""
        mask = np.ones(self.n, bool) if mask is None else mask
        self.frames.append(Frame(self.n, "void"))
        try:
            M.set_active(mask.sum())
            return self.call(name, args, mask)
        finally:
            self.frames.pop()


# the ("value", v) pseudo-expression used for out-parameter copy-back
_orig_eval = Interp.eval


def _eval_with_value(self, e, mask):
    if e[0] == "value":
        return e[1]
    return _orig_eval(self, e, mask)


Interp.eval = _eval_with_value


# =============================================================================================
# GLSL builtins (GLSL ES 3.00 chapter 8) on the value model
# =============================================================================================
def _atan(*a):
    return V.map1(M.atan, a[0]) if len(a) == 1 else V.map2(M.atan2, a[0], a[1])


def _mix(a, b, t):
    return V.map3(M.mix, a, b, t)


def _reflect(i, n):
    return V.binop("-", i, V.binop("*", n, M.mul(F32(2), V.dot(n, i))))


BUILTINS = {
    "sin": lambda a: V.map1(M.sin, a),
    "cos": lambda a: V.map1(M.cos, a),
    "tan": lambda a: V.map1(M.tan, a),
    "asin": lambda a: V.map1(M.asin, a),
    "acos": lambda a: V.map1(M.acos, a),
    "atan": _atan,
    "exp": lambda a: V.map1(M.exp, a),
    "log": lambda a: V.map1(M.log, a),
    "exp2": lambda a: V.map1(M.exp2, a),
    "log2": lambda a: V.map1(M.log2, a),
    "pow": lambda a, b: V.map2(M.pow, a, b),
    "sqrt": lambda a: V.map1(M.sqrt, a),
    "inversesqrt": lambda a: V.map1(M.inversesqrt, a),
    "abs": lambda a: (np.abs(np.asarray(a, I32)).astype(I32) if V.is_int(a) else V.map1(M.absf, a)),
    "sign": lambda a: V.map1(M.sign, a),
    "floor": lambda a: V.map1(M.floor, a),
    "ceil": lambda a: V.map1(M.ceil, a),
    "trunc": lambda a: V.map1(M.trunc, a),
    "round": lambda a: V.map1(M.rint, a),
    "roundEven": lambda a: V.map1(M.rint, a),
    "fract": lambda a: V.map1(M.fract, a),
    "radians": lambda a: V.map1(M.radians, a),
    "degrees": lambda a: V.map1(M.degrees, a),
    "mod": lambda a, b: V.map2(M.mod, a, b),
    "min": lambda a, b: (np.minimum(a, b).astype(I32) if V.is_int(a) and V.is_int(b) else V.map2(M.fmin, a, b)),
    "max": lambda a, b: (np.maximum(a, b).astype(I32) if V.is_int(a) and V.is_int(b) else V.map2(M.fmax, a, b)),
    "clamp": lambda x, lo, hi: V.map3(M.clamp, x, lo, hi),
    "mix": _mix,
    "step": lambda e, x: V.map2(M.step, e, x),
    "smoothstep": lambda a, b, x: V.map3(M.smoothstep, a, b, x),
    "length": V.length,
    "distance": lambda a, b: V.length(V.binop("-", a, b)),
    "dot": lambda a, b: V.dot(a, b) if isinstance(a, Vec) else M.mul(a, b),
    "cross": V.cross,
    "normalize": V.normalize,
    "reflect": _reflect,
    "texture": V.texture,
    "isnan": lambda a: np.isnan(V.to_float(a)),
    "isinf": lambda a: np.isinf(V.to_float(a)),
}
