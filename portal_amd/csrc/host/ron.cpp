// ron.cpp -- recursive-descent RON reader (see ron.h).
#include "ron.h"

#include <algorithm>
#include <cctype>
#include <charconv>
#include <cstdlib>
#include <cstring>
#include <cmath>

namespace ptl::ron {
namespace {

struct Parser {
    const std::string& s;
    size_t i = 0;
    explicit Parser(const std::string& text) : s(text) {}

    [[noreturn]] void fail(const std::string& msg) const {
        int line = 1;
        for (size_t k = 0; k < i && k < s.size(); ++k)
            if (s[k] == '\n') ++line;
        throw ParseError(line, msg);
    }
    bool starts(const char* lit) const { return s.compare(i, std::strlen(lit), lit) == 0; }

    void skip_ws() {
        while (i < s.size()) {
            char c = s[i];
            if (c == ' ' || c == '\t' || c == '\r' || c == '\n') {
                ++i;
            } else if (starts("//")) {
                size_t j = s.find('\n', i);
                i = j == std::string::npos ? s.size() : j + 1;
            } else if (starts("/*")) {
                size_t j = s.find("*/", i + 2);
                if (j == std::string::npos) fail("unterminated block comment");
                i = j + 2;
            } else {
                break;
            }
        }
    }
    char peek() {
        skip_ws();
        return i < s.size() ? s[i] : '\0';
    }
    void expect(char c) {
        if (peek() != c) fail(std::string("expected `") + c + "`, found `" + s.substr(i, 12) + "`");
        ++i;
    }
    static bool ident_char(char c) { return std::isalnum((unsigned char)c) || c == '_'; }
    std::string ident() {
        size_t j = i;
        while (j < s.size() && ident_char(s[j])) ++j;
        std::string out = s.substr(i, j - i);
        i = j;
        return out;
    }
    static void append_utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) {
            out += (char)cp;
        } else if (cp < 0x800) {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }
    std::string quoted() {
        ++i;  // opening quote
        std::string out;
        for (;;) {
            if (i >= s.size()) fail("unterminated string");
            char c = s[i++];
            if (c == '"') return out;
            if (c != '\\') {
                out += c;
                continue;
            }
            if (i >= s.size()) fail("dangling escape");
            char e = s[i++];
            switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case '0': out += '\0'; break;
                case '\\': case '"': case '\'': case '/': out += e; break;
                case 'x': {
                    out += (char)std::strtoul(s.substr(i, 2).c_str(), nullptr, 16);
                    i += 2;
                    break;
                }
                case 'u': {
                    size_t j = s.find('}', i);
                    if (j == std::string::npos || s[i] != '{') fail("bad \\u escape");
                    append_utf8(out, (unsigned)std::strtoul(s.substr(i + 1, j - i - 1).c_str(), nullptr, 16));
                    i = j + 1;
                    break;
                }
                default: fail(std::string("bad escape \\") + e);
            }
        }
    }
    std::string raw_string() {
        size_t j = i + 1;
        size_t hashes = 0;
        while (j < s.size() && s[j] == '#') {
            ++hashes;
            ++j;
        }
        if (j >= s.size() || s[j] != '"') fail("bad raw string");
        std::string end = "\"" + std::string(hashes, '#');
        size_t k = s.find(end, j + 1);
        if (k == std::string::npos) fail("unterminated raw string");
        std::string out = s.substr(j + 1, k - j - 1);
        i = k + end.size();
        return out;
    }
    Value number() {
        size_t j = i;
        if (s[j] == '+' || s[j] == '-') ++j;
        bool is_float = false;
        while (j < s.size()) {
            char c = s[j];
            if (std::isdigit((unsigned char)c) || c == '_') {
            } else if (c == '.' || c == 'e' || c == 'E') {
                is_float = true;
            } else if ((c == '+' || c == '-') && (s[j - 1] == 'e' || s[j - 1] == 'E')) {
            } else {
                break;
            }
            ++j;
        }
        std::string tok;
        for (size_t k = i; k < j; ++k)
            if (s[k] != '_') tok += s[k];
        if (tok.empty() || tok == "+" || tok == "-") fail("bad number");
        i = j;
        Value v;
        if (is_float) {
            v.kind = Value::Float;
            v.f = std::strtod(tok.c_str(), nullptr);
        } else {
            v.kind = Value::Int;
            v.i = std::strtoll(tok.c_str(), nullptr, 10);
            v.f = (double)v.i;
        }
        return v;
    }
    // after an optional name: `( field: v, ... )` or `( v, ... )`
    Value parens(const std::string& name) {
        expect('(');
        Value out;
        out.s = name;
        if (peek() == ')') {
            ++i;
            out.kind = Value::Tuple;
            return out;
        }
        size_t save = i;
        bool is_struct = false;
        char c = peek();
        if (std::isalpha((unsigned char)c) || c == '_') {
            ident();
            if (peek() == ':') is_struct = true;
        }
        i = save;
        if (is_struct) {
            out.kind = Value::Struct;
            for (;;) {
                if (peek() == ')') {
                    ++i;
                    break;
                }
                std::string key = ident();
                if (key.empty()) fail("expected field name");
                expect(':');
                out.fields.emplace_back(key, value());
                if (peek() == ',') ++i;
            }
        } else {
            out.kind = Value::Tuple;
            for (;;) {
                if (peek() == ')') {
                    ++i;
                    break;
                }
                out.items.push_back(value());
                if (peek() == ',') ++i;
            }
        }
        return out;
    }
    Value value() {
        char c = peek();
        if (c == '\0') fail("unexpected end of input");
        Value v;
        if (c == '"') {
            v.kind = Value::String;
            v.s = quoted();
            return v;
        }
        if (c == 'r' && i + 1 < s.size() && (s[i + 1] == '#' || s[i + 1] == '"')) {
            v.kind = Value::String;
            v.s = raw_string();
            return v;
        }
        if (c == '(') return parens("");
        if (c == '[') {
            ++i;
            v.kind = Value::List;
            for (;;) {
                if (peek() == ']') {
                    ++i;
                    return v;
                }
                v.items.push_back(value());
                if (peek() == ',') ++i;
            }
        }
        if (c == '{') {
            ++i;
            v.kind = Value::Map;
            for (;;) {
                if (peek() == '}') {
                    ++i;
                    return v;
                }
                Value k = value();
                expect(':');
                Value val = value();
                v.entries.emplace_back(std::move(k), std::move(val));
                if (peek() == ',') ++i;
            }
        }
        if (std::isdigit((unsigned char)c) || c == '+' || c == '-' || c == '.') return number();
        if (std::isalpha((unsigned char)c) || c == '_') {
            std::string name = ident();
            if (name == "true" || name == "false") {
                v.kind = Value::Bool;
                v.b = name == "true";
                return v;
            }
            if (name == "inf" || name == "NaN") {
                v.kind = Value::Float;
                v.f = name == "inf" ? HUGE_VAL : std::strtod("nan", nullptr);
                return v;
            }
            if (peek() == '(') return parens(name);
            v.kind = Value::Unit;
            v.s = name;
            return v;
        }
        fail(std::string("unexpected character `") + c + "`");
    }
};

}  // namespace

namespace {

void write_float(std::string& out, double f) {
    if (std::isnan(f)) {
        out += "NaN";
        return;
    }
    if (std::isinf(f)) {
        out += f > 0 ? "inf" : "-inf";
        return;
    }
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, f, std::chars_format::fixed);  // shortest digits that read back as f, no exponent
    std::string t(buf, r.ptr);
    size_t dot = t.find('.');
    if (dot == std::string::npos) {
        t += ".0";
    } else {
        // Rust's float printing (what ron uses) resolves an exact tie between two equally short candidates upwards in
        // magnitude, std::to_chars to the even digit: 0.99658966064453125 is "...313" there, "...312" here.  Detect the tie
        // on the exact expansion (every binary64 has a finite one) and bump the last digit.
        size_t decimals = t.size() - dot - 1;
        char exact[1200];
        auto e = std::to_chars(exact, exact + sizeof exact, f, std::chars_format::fixed, 1100);
        std::string full(exact, e.ptr);
        size_t fdot = full.find('.');
        std::string rest = full.substr(fdot + 1 + decimals);
        bool tie = !rest.empty() && rest[0] == '5' && rest.find_first_not_of('0', 1) == std::string::npos;
        std::string truncated = full.substr(0, fdot + 1 + decimals);
        if (tie && truncated == t) {  // to_chars rounded down (to even): take the upper neighbour
            size_t k = t.size();
            while (k > 0) {
                --k;
                if (t[k] == '.' || t[k] == '-') continue;
                if (t[k] == '9') {
                    t[k] = '0';
                } else {
                    ++t[k];
                    break;
                }
            }
        }
    }
    out += t;
}

void write_string(std::string& out, const std::string& s) {
    if (s.find('"') == std::string::npos && s.find('\\') == std::string::npos) {
        out += '"';
        out += s;
        out += '"';
        return;
    }
    // raw string: one more hash than the longest run of hashes anywhere in the text (ron counts runs, not just those after a quote)
    size_t need = 1, run = 0;
    for (char c : s) {
        run = c == '#' ? run + 1 : 0;
        need = std::max(need, run + 1);
    }
    out += 'r';
    out.append(need, '#');
    out += '"';
    out += s;
    out += '"';
    out.append(need, '#');
}

void write_value(std::string& out, const Value& v, int level) {
    auto indent = [&](int l) { out.append((size_t)l * 4, ' '); };
    switch (v.kind) {
        case Value::Int: out += std::to_string(v.i); break;
        case Value::Float: write_float(out, v.f); break;
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::String: write_string(out, v.s); break;
        case Value::Unit: out += v.s; break;
        case Value::Tuple:
            out += v.s;
            out += '(';
            for (size_t k = 0; k < v.items.size(); ++k) {
                if (k) out += ", ";
                write_value(out, v.items[k], level);
            }
            out += ')';
            break;
        case Value::Struct:
            out += v.s;
            out += "(\n";
            for (auto& kv : v.fields) {
                indent(level + 1);
                out += kv.first;
                out += ": ";
                write_value(out, kv.second, level + 1);
                out += ",\n";
            }
            indent(level);
            out += ')';
            break;
        case Value::List:
            if (v.items.empty()) {
                out += "[]";
                break;
            }
            out += "[\n";
            for (auto& item : v.items) {
                indent(level + 1);
                write_value(out, item, level + 1);
                out += ",\n";
            }
            indent(level);
            out += ']';
            break;
        case Value::Map:
            if (v.entries.empty()) {
                out += "{}";
                break;
            }
            out += "{\n";
            for (auto& kv : v.entries) {
                indent(level + 1);
                write_value(out, kv.first, level + 1);
                out += ": ";
                write_value(out, kv.second, level + 1);
                out += ",\n";
            }
            indent(level);
            out += '}';
            break;
    }
}

}  // namespace

std::string to_string(const Value& v) {
    std::string out;
    write_value(out, v, 0);
    return out;
}

Value parse(const std::string& text) {
    Parser p(text);
    Value v = p.value();
    p.skip_ws();
    if (p.i != text.size()) p.fail("trailing characters after document");
    return v;
}

}  // namespace ptl::ron
