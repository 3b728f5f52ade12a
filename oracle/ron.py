"""RON (Rusty Object Notation) reader for optozorax/portal scene files.

TEST INFRASTRUCTURE ONLY (part of the CPU oracle): nothing under portal_amd/
may import this module.  It is an independent restatement of what the
reference gets from the third-party `ron 0.10.1` crate (Cargo.lock:1489,
call site src/main.rs:2882) for the subset of RON the scene corpus uses
(SURVEY.md Appendix C).

Value model
    Struct(name|None, {field: value})   `Name(a: 1, b: 2)` / `(a: 1)`
    Tuple(name|None, [values])          `Name(1, 2)` / `(1, 2)`; `Some(x)` is Tuple('Some',[x])
    Unit(name)                          `Name`  (`None`, `Normal`, `FromDev` ...)
    list, dict, str, bool, int, float   as in Python
"""
from __future__ import annotations
from dataclasses import dataclass, field


@dataclass
class Struct:
    name: str | None
    fields: dict

    def get(self, key, default=None):
        return self.fields.get(key, default)

    def __getitem__(self, key):
        return self.fields[key]

    def __contains__(self, key):
        return key in self.fields


@dataclass
class Tuple:
    name: str | None
    items: list

    def __getitem__(self, i):
        return self.items[i]

    def __len__(self):
        return len(self.items)


@dataclass
class Unit:
    name: str


class RonError(ValueError):
    pass


class _P:
    def __init__(self, text: str):
        self.s = text
        self.i = 0
        self.n = len(text)

    def err(self, msg):
        line = self.s.count("\n", 0, self.i) + 1
        raise RonError(f"RON parse error at line {line}: {msg}")

    def ws(self):
        s, n = self.s, self.n
        while self.i < n:
            c = s[self.i]
            if c in " \t\r\n":
                self.i += 1
            elif s.startswith("//", self.i):
                j = s.find("\n", self.i)
                self.i = n if j < 0 else j + 1
            elif s.startswith("/*", self.i):
                j = s.find("*/", self.i + 2)
                if j < 0:
                    self.err("unterminated block comment")
                self.i = j + 2
            else:
                break

    def peek(self):
        self.ws()
        return self.s[self.i] if self.i < self.n else ""

    def eat(self, c):
        if self.peek() != c:
            self.err(f"expected {c!r}, found {self.s[self.i:self.i+10]!r}")
        self.i += 1

    def ident(self):
        s = self.s
        j = self.i
        while j < self.n and (s[j].isalnum() or s[j] == "_"):
            j += 1
        out = s[self.i:j]
        self.i = j
        return out

    def string(self):
        s = self.s
        assert s[self.i] == '"'
        self.i += 1
        out = []
        while True:
            if self.i >= self.n:
                self.err("unterminated string")
            c = s[self.i]
            if c == '"':
                self.i += 1
                return "".join(out)
            if c == "\\":
                self.i += 1
                e = s[self.i]
                self.i += 1
                if e == "n":
                    out.append("\n")
                elif e == "t":
                    out.append("\t")
                elif e == "r":
                    out.append("\r")
                elif e == "0":
                    out.append("\0")
                elif e in "\\\"'/":
                    out.append(e)
                elif e == "x":
                    out.append(chr(int(s[self.i:self.i + 2], 16)))
                    self.i += 2
                elif e == "u":
                    j = s.index("}", self.i)
                    out.append(chr(int(s[self.i + 1:j], 16)))
                    self.i = j + 1
                else:
                    self.err(f"bad escape \\{e}")
            else:
                out.append(c)
                self.i += 1

    def raw_string(self):
        s = self.s
        assert s[self.i] == "r"
        j = self.i + 1
        hashes = 0
        while s[j] == "#":
            hashes += 1
            j += 1
        if s[j] != '"':
            self.err("bad raw string")
        end = '"' + "#" * hashes
        k = s.find(end, j + 1)
        if k < 0:
            self.err("unterminated raw string")
        self.i = k + len(end)
        return s[j + 1:k]

    def number(self):
        s = self.s
        j = self.i
        if s[j] in "+-":
            j += 1
        isf = False
        while j < self.n and (s[j].isdigit() or s[j] in "._eE" or (s[j] in "+-" and s[j - 1] in "eE")):
            if s[j] in ".eE":
                isf = True
            j += 1
        tok = s[self.i:j].replace("_", "")
        self.i = j
        if tok in ("inf", "+inf"):
            return float("inf")
        return float(tok) if isf else int(tok)

    def parens(self, name):
        """after '(' -- decide struct vs tuple by `ident :` lookahead"""
        self.eat("(")
        if self.peek() == ")":
            self.i += 1
            return Tuple(name, [])
        save = self.i
        is_struct = False
        c = self.peek()
        if c.isalpha() or c == "_":
            self.ident()
            if self.peek() == ":":
                is_struct = True
        self.i = save
        if is_struct:
            fields = {}
            while True:
                if self.peek() == ")":
                    self.i += 1
                    break
                k = self.ident()
                if not k:
                    self.err("expected field name")
                self.eat(":")
                fields[k] = self.value()
                if self.peek() == ",":
                    self.i += 1
            return Struct(name, fields)
        items = []
        while True:
            if self.peek() == ")":
                self.i += 1
                break
            items.append(self.value())
            if self.peek() == ",":
                self.i += 1
        return Tuple(name, items)

    def value(self):
        c = self.peek()
        s = self.s
        if c == "":
            self.err("unexpected end of input")
        if c == '"':
            return self.string()
        if c == "r" and self.i + 1 < self.n and s[self.i + 1] in '#"':
            return self.raw_string()
        if c == "(":
            return self.parens(None)
        if c == "[":
            self.i += 1
            out = []
            while True:
                if self.peek() == "]":
                    self.i += 1
                    return out
                out.append(self.value())
                if self.peek() == ",":
                    self.i += 1
        if c == "{":
            self.i += 1
            out = {}
            while True:
                if self.peek() == "}":
                    self.i += 1
                    return out
                k = self.value()
                self.eat(":")
                out[k] = self.value()
                if self.peek() == ",":
                    self.i += 1
        if c.isdigit() or c in "+-.":
            return self.number()
        if c.isalpha() or c == "_":
            name = self.ident()
            if name == "true":
                return True
            if name == "false":
                return False
            if name in ("inf", "NaN"):
                return float(name)
            if self.peek() == "(":
                return self.parens(name)
            return Unit(name)
        self.err(f"unexpected character {c!r}")


def loads(text: str):
    p = _P(text)
    v = p.value()
    p.ws()
    if p.i != p.n:
        p.err("trailing characters")
    return v


def load(path: str):
    with open(path, encoding="utf-8") as f:
        return loads(f.read())


def is_none(v):
    return isinstance(v, Unit) and v.name == "None"


def unwrap_some(v):
    """Option<T>: `Some(x)` -> x, `None` -> None."""
    if is_none(v):
        return None
    if isinstance(v, Tuple) and v.name == "Some":
        return v.items[0]
    return v
