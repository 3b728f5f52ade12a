// glsl_translate.cpp -- see glsl_translate.h.
#include "glsl_translate.h"
#include "glsl_tokens.h"

#include <stdexcept>

#include <algorithm>
#include <cctype>
#include <cstring>
#include <map>
#include <set>
#include <vector>

namespace ptl {

std::string filter_tagged_lines(const std::string& text, const CodegenFlags& f) {
    std::string out;
    out.reserve(text.size());
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t eol = text.find('\n', pos);
        bool last = eol == std::string::npos;
        std::string line = text.substr(pos, last ? std::string::npos : eol - pos);
        auto has = [&](const char* tag) { return line.find(tag) != std::string::npos; };
        bool skip = (has("!FOR_NUMBER!") && f.for_prefer_variable) || (has("!FOR_VARIABLE!") && !f.for_prefer_variable) ||
                    (has("!ANTIALIASING!") && f.disable_antialiasing) || (has("!ANAGLYPH!") && f.disable_anaglyph) ||
                    (has("!CAMERA_TELEPORTATION!") && f.disable_camera_teleportation) || (has("!GLSL100!") && f.use_300_version) ||
                    (has("!GLSL300!") && !f.use_300_version);
        if (!skip) out += line;
        if (last) break;
        out += '\n';
        pos = eol + 1;
    }
    return out;
}

std::vector<Token> tokenize_glsl(const std::string& s) {
    std::vector<Token> out;
    size_t i = 0, n = s.size();
    bool line_start = true;
    while (i < n) {
        char c = s[i];
        if (c == '\n') {
            out.push_back({Token::Space, "\n"});
            ++i;
            line_start = true;
            continue;
        }
        if (c == ' ' || c == '\t' || c == '\r') {
            size_t j = i;
            while (j < n && (s[j] == ' ' || s[j] == '\t' || s[j] == '\r')) ++j;
            out.push_back({Token::Space, s.substr(i, j - i)});
            i = j;
            continue;
        }
        if (c == '#' && line_start) {  // preprocessor directive: keep the line verbatim ...
            size_t j = s.find('\n', i);
            if (j == std::string::npos) j = n;
            // ... except the replacement text of a `#define`, which is GLSL like any other: its float literals need their suffix and its
            // divisions are the contract's (rewrite_divisions), e.g. the reference's own `#define PI2 (acos(-1.) / 2.0)`.  The head
            // (`#define NAME` or `#define NAME(params)`) stays one verbatim token, the rest of the line is tokenised as code.
            size_t h = i + 1;
            while (h < j && (s[h] == ' ' || s[h] == '\t')) ++h;
            if (s.compare(h, 6, "define") == 0 && h + 6 < j && (s[h + 6] == ' ' || s[h + 6] == '\t') && s.find('\\', i) >= j) {
                h += 6;
                while (h < j && (s[h] == ' ' || s[h] == '\t')) ++h;
                while (h < j && (std::isalnum((unsigned char)s[h]) || s[h] == '_')) ++h;
                if (h < j && s[h] == '(') {  // function-like: the parameter list belongs to the head
                    size_t close = s.find(')', h);
                    h = (close == std::string::npos || close >= j) ? j : close + 1;
                }
                out.push_back({Token::Preproc, s.substr(i, h - i)});
                i = h;
                line_start = false;
                continue;
            }
            out.push_back({Token::Preproc, s.substr(i, j - i)});
            i = j;
            continue;
        }
        line_start = false;
        if (c == '/' && i + 1 < n && s[i + 1] == '/') {
            size_t j = s.find('\n', i);
            if (j == std::string::npos) j = n;
            out.push_back({Token::Comment, s.substr(i, j - i)});
            i = j;
            continue;
        }
        if (c == '/' && i + 1 < n && s[i + 1] == '*') {
            size_t j = s.find("*/", i + 2);
            j = j == std::string::npos ? n : j + 2;
            out.push_back({Token::Comment, s.substr(i, j - i)});
            i = j;
            continue;
        }
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t j = i;
            while (j < n && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
            out.push_back({Token::Ident, s.substr(i, j - i)});
            i = j;
            continue;
        }
        if (std::isdigit((unsigned char)c) || (c == '.' && i + 1 < n && std::isdigit((unsigned char)s[i + 1]))) {
            size_t j = i;
            if (c == '0' && j + 1 < n && (s[j + 1] == 'x' || s[j + 1] == 'X')) {
                j += 2;
                while (j < n && std::isxdigit((unsigned char)s[j])) ++j;
            } else {
                while (j < n && std::isdigit((unsigned char)s[j])) ++j;
                if (j < n && s[j] == '.') {
                    ++j;
                    while (j < n && std::isdigit((unsigned char)s[j])) ++j;
                }
                if (j < n && (s[j] == 'e' || s[j] == 'E')) {
                    size_t k = j + 1;
                    if (k < n && (s[k] == '+' || s[k] == '-')) ++k;
                    if (k < n && std::isdigit((unsigned char)s[k])) {
                        while (k < n && std::isdigit((unsigned char)s[k])) ++k;
                        j = k;
                    }
                }
            }
            while (j < n && (s[j] == 'f' || s[j] == 'F' || s[j] == 'u' || s[j] == 'U')) ++j;  // suffixes
            out.push_back({Token::Number, s.substr(i, j - i)});
            i = j;
            continue;
        }
        // two-character operators that matter for the lvalue-swizzle test
        static const char* two[] = {"==", "!=", "<=", ">=", "+=", "-=", "*=", "/=", "&&", "||", "++", "--", "<<", ">>"};
        bool matched = false;
        for (const char* t : two) {
            if (s.compare(i, 2, t) == 0) {
                out.push_back({Token::Punct, t});
                i += 2;
                matched = true;
                break;
            }
        }
        if (matched) continue;
        out.push_back({Token::Punct, std::string(1, c)});
        ++i;
    }
    return out;
}

namespace {

std::vector<Token> tokenize(const std::string& s) { return tokenize_glsl(s); }

bool is_float_literal(const std::string& t) {
    if (t.size() > 1 && t[0] == '0' && (t[1] == 'x' || t[1] == 'X')) return false;
    for (char c : t)
        if (c == '.' || c == 'e' || c == 'E') return true;
    return false;
}

// .xyzw / .rgba / .stpq selectors -> component indices; empty if `t` is not a swizzle
std::vector<int> swizzle_indices(const std::string& t) {
    static const char* sets[] = {"xyzw", "rgba", "stpq"};
    if (t.size() < 2 || t.size() > 4) return {};
    for (const char* set : sets) {
        std::vector<int> idx;
        for (char c : t) {
            const char* p = std::strchr(set, c);
            if (!p) {
                idx.clear();
                break;
            }
            idx.push_back((int)(p - set));
        }
        if (idx.size() == t.size()) return idx;
    }
    return {};
}

const std::set<std::string>& cpp_only_keywords() {
    static const std::set<std::string> k = {
        "alignas", "alignof", "and", "and_eq", "asm", "auto", "bitand", "bitor", "catch", "char", "class", "compl", "concept",
        "const_cast", "consteval", "constexpr", "constinit", "co_await", "co_return", "co_yield", "decltype", "delete",
        "dynamic_cast", "enum", "explicit", "export", "extern", "friend", "goto", "inline", "long", "mutable", "namespace", "new",
        "noexcept", "not", "not_eq", "nullptr", "operator", "or", "or_eq", "private", "protected", "public", "register",
        "reinterpret_cast", "requires", "short", "signed", "sizeof", "static", "static_assert", "static_cast", "template", "this",
        "thread_local", "throw", "try", "typedef", "typeid", "typename", "union", "unsigned", "using", "virtual", "volatile",
        "wchar_t", "xor", "xor_eq", "double"};
    return k;
}

bool looks_like_a_uniform(const std::string& name) {
    auto ends = [&](const char* suffix) {
        size_t n = std::strlen(suffix);
        return name.size() > n && name.compare(name.size() - n, n, suffix) == 0;
    };
    return ends("_mat") || ends("_mat_inv") || ends("_mat_teleport") || ends("_u") || (name.size() > 1 && name[0] == '_');
}

bool is_assign_op(const Token& t) {
    return t.kind == Token::Punct && (t.text == "=" || t.text == "+=" || t.text == "-=" || t.text == "*=" || t.text == "/=" || t.text == "++" || t.text == "--");
}

// See glsl_translate.h (`defer_loop_updates`).  Works on the significant tokens; every check that fails leaves the loop untouched.
void defer_loop_carried_updates(std::vector<Token>& toks) {
    std::vector<size_t> sig;  // indices of the significant tokens
    for (size_t k = 0; k < toks.size(); ++k)
        if (toks[k].kind != Token::Space && toks[k].kind != Token::Comment && toks[k].kind != Token::Preproc) sig.push_back(k);
    const size_t n = sig.size();
    auto T = [&](size_t i) -> const Token& { return toks[sig[i]]; };
    auto is = [&](size_t i, const char* text) { return i < n && T(i).kind == Token::Punct && T(i).text == text; };
    auto ident = [&](size_t i, const char* text) { return i < n && T(i).kind == Token::Ident && T(i).text == text; };
    auto match = [&](size_t open, const char* a, const char* b) -> size_t {  // index of the bracket closing the one at `open`, or n
        int depth = 0;
        for (size_t i = open; i < n; ++i) {
            if (is(i, a)) ++depth;
            else if (is(i, b) && --depth == 0) return i;
        }
        return n;
    };
    // identifiers the snippet assigns or declares somewhere: never treated as loop-invariant
    std::set<std::string> written;
    for (size_t i = 0; i < n; ++i) {
        if (T(i).kind != Token::Ident) continue;
        bool target = (i + 1 < n && is_assign_op(T(i + 1))) || (i > 0 && (is(i - 1, "++") || is(i - 1, "--")));
        bool declared = i > 0 && T(i - 1).kind == Token::Ident && i + 1 < n && (is(i + 1, "=") || is(i + 1, ";") || is(i + 1, ",") || is(i + 1, ")") || is(i + 1, "["));
        if (target || declared) written.insert(T(i).text);
    }
    for (size_t i = 0; i < n; ++i)
        if (ident(i, "do") || ident(i, "goto")) return;  // do-while bodies are not tracked as loops: leave such code alone
    struct Loop {
        size_t kw, header_open, header_close, body_open, body_close;
    };
    std::vector<Loop> loops;  // every loop with a braced body, `for` and `while`
    for (size_t i = 0; i + 1 < n; ++i) {
        if (!(ident(i, "for") || ident(i, "while")) || !is(i + 1, "(")) continue;
        size_t close = match(i + 1, "(", ")");
        if (close >= n || !is(close + 1, "{")) continue;
        size_t end = match(close + 1, "{", "}");
        if (end >= n) continue;
        loops.push_back({i, i + 1, close, close + 1, end});
    }
    struct Insert {
        size_t at;  // significant-token index the text goes in front of
        std::string text;
    };
    std::vector<Insert> inserts;
    std::vector<std::pair<size_t, size_t>> removed;  // [first, last] significant tokens replaced by the text of the insert at `first`
    int counter = 0;
    for (const Loop& L : loops) {
        if (!ident(L.kw, "for")) continue;
        bool nested = false;
        for (const Loop& O : loops) nested = nested || (O.body_open < L.kw && L.kw < O.body_close);
        if (nested) continue;
        if (L.kw > 0 && !(is(L.kw - 1, ";") || is(L.kw - 1, "{") || is(L.kw - 1, "}"))) continue;  // `if (c) for ...`: nowhere to declare the counter
        // the enclosing function: for a body-only snippet (loop at brace depth 0) the whole text, else from the brace that opens the
        // function body to the one that closes it
        size_t fn_begin = 0, fn_end = n;
        {
            int depth = 0;
            for (size_t i = 0; i < L.kw; ++i) {
                if (is(i, "{") && depth++ == 0) fn_begin = i + 1;
                else if (is(i, "}")) --depth;
            }
            if (depth == 0) fn_begin = 0;
            if (depth > 0) {
                int d = depth;
                for (size_t i = L.body_close + 1; i < n; ++i) {
                    d += is(i, "{") ? 1 : is(i, "}") ? -1 : 0;
                    if (d == 0) {
                        fn_end = i;
                        break;
                    }
                }
            }
        }
        // statement starts inside the body, by paren depth 0 delimiters
        int brace = 0, paren = 0;
        size_t stmt_start = L.body_open + 1;
        std::vector<size_t> start_of(n, 0);
        std::vector<int> depth_of(n, 0);
        for (size_t i = L.body_open + 1; i < L.body_close; ++i) {
            start_of[i] = stmt_start;
            depth_of[i] = brace;
            if (is(i, "(")) ++paren;
            else if (is(i, ")")) --paren;
            else if (is(i, "{")) { ++brace; if (paren == 0) stmt_start = i + 1; }
            else if (is(i, "}")) { --brace; if (paren == 0) stmt_start = i + 1; }
            else if (is(i, ";") && paren == 0) stmt_start = i + 1;
        }
        for (size_t i = L.body_open + 1; i < L.body_close; ++i) {
            if (depth_of[i] != 0 || start_of[i] != i || T(i).kind != Token::Ident || !is(i + 1, "=")) continue;
            const std::string X = T(i).text;
            size_t semi = i + 2;
            int p = 0, x_in_rhs = 0;
            bool ok = true;
            for (; semi < L.body_close && !(is(semi, ";") && p == 0); ++semi) {
                const Token& t = T(semi);
                if (is(semi, "(")) ++p;
                else if (is(semi, ")")) --p;
                else if (is(semi, ",") || is(semi, "*")) {}
                else if (t.kind == Token::Ident && t.text == X) ++x_in_rhs;
                else if (t.kind == Token::Ident && t.text == "transform" && is(semi + 1, "(")) {}
                else if (t.kind == Token::Ident && looks_like_a_uniform(t.text) && !written.count(t.text) && !(semi > 0 && is(semi - 1, "."))) {}
                else ok = false;
            }
            if (!ok || x_in_rhs != 1 || semi >= L.body_close || p != 0) continue;
            // X elsewhere: not in the header, only in nested blocks of the body, never written, never a member name
            std::vector<size_t> uses;
            for (size_t j = L.header_open; j <= L.header_close && ok; ++j) ok = !(T(j).kind == Token::Ident && T(j).text == X);
            for (size_t j = L.body_open + 1; j < L.body_close && ok; ++j) {
                if (j >= i && j <= semi) continue;
                if (T(j).kind != Token::Ident || T(j).text != X) continue;
                if (j > 0 && is(j - 1, ".")) { ok = false; break; }
                bool target = is_assign_op(T(j + 1)) || is(j - 1, "++") || is(j - 1, "--");
                if (depth_of[j] < 1 || target) { ok = false; break; }
                if (ident(start_of[j], "else") || ident(start_of[j], "case") || ident(start_of[j], "default")) { ok = false; break; }
                uses.push_back(start_of[j]);
            }
            if (!ok || uses.empty()) continue;
            // X must be a LOCAL of this function, declared before the loop: an `out` / `inout` parameter or a global would have to carry
            // its final value out of the function, which a deferred update does not guarantee
            bool local = false;
            for (size_t j = fn_begin + 1; j + 1 < L.kw && !local; ++j)
                local = T(j).kind == Token::Ident && T(j).text == X && T(j - 1).kind == Token::Ident && (is(j + 1, "=") || is(j + 1, ";") || is(j + 1, ","));
            if (!local) continue;
            bool used_after = false;
            for (size_t j = L.body_close + 1; j < fn_end; ++j) used_after = used_after || (T(j).kind == Token::Ident && T(j).text == X);
            // the update statement as text, the counter, the flush
            std::string update;
            for (size_t j = i; j <= semi; ++j) update += T(j).text + (j < semi && T(j).kind == Token::Ident && T(j + 1).kind == Token::Ident ? " " : "");
            const std::string pend = "ptl_pend_" + std::to_string(counter++);
            const std::string flush = "for (; " + pend + " > 0; --" + pend + ") " + update + " ";
            inserts.push_back({L.kw, "int " + pend + " = 0; "});
            inserts.push_back({i, "++" + pend + ";"});
            removed.push_back({i, semi});
            std::sort(uses.begin(), uses.end());
            uses.erase(std::unique(uses.begin(), uses.end()), uses.end());
            for (size_t u : uses) inserts.push_back({u, flush});
            if (used_after) inserts.push_back({L.body_close + 1, flush});
            i = semi;
        }
    }
    if (inserts.empty()) return;
    // rebuild the token list: inserts go in front of their anchor, removed ranges keep only their line breaks
    std::vector<bool> drop(toks.size(), false);
    for (auto& r : removed)
        for (size_t k = sig[r.first]; k <= sig[r.second]; ++k) drop[k] = true;
    std::stable_sort(inserts.begin(), inserts.end(), [](const Insert& a, const Insert& b) { return a.at < b.at; });
    std::vector<Token> out;
    size_t next = 0;
    for (size_t k = 0; k <= toks.size(); ++k) {
        while (next < inserts.size() && (inserts[next].at >= n ? k == toks.size() : sig[inserts[next].at] == k)) out.push_back({Token::Raw, inserts[next++].text});
        if (k == toks.size()) break;
        if (!drop[k]) out.push_back(toks[k]);
        else if (toks[k].kind == Token::Space && toks[k].text == "\n") out.push_back(toks[k]);  // line-preserving
    }
    toks.swap(out);
}

// `a / b` -> `ptl_div(a, b)`, `a /= b` -> `ptl_div_assign(a, b)` (device/ptl_glsl.h, "CONTRACT 2"): C++ cannot overload the division
// of two scalars, and the numerics contract defines it (a * (1/b) with the contract's reciprocal), so every division of a snippet
// becomes a call; the overload set of ptl_div keeps int / int an integer division and sends vector operands to the vector operators.
// Operand extents follow GLSL's grammar: the right operand is ONE unary expression (prefix operators, a primary, its postfix chain),
// the left operand the whole multiplicative chain in front (`a * b / c` is `(a * b) / c`; `*`, `/`, `%` associate to the left --
// earlier divisions of the chain have already been rewritten when a later one is reached, so their text is part of that operand).
// `x.yz /= e` (a multi-component swizzle as target) stays: the swizzle proxy's own operator/= does the same arithmetic.
void rewrite_divisions(std::vector<Token>& toks) {
    std::vector<size_t> sig;
    for (size_t k = 0; k < toks.size(); ++k)
        if (toks[k].kind != Token::Space && toks[k].kind != Token::Comment && toks[k].kind != Token::Preproc) sig.push_back(k);
    const size_t n = sig.size();
    auto T = [&](size_t i) -> const Token& { return toks[sig[i]]; };
    auto is = [&](size_t i, const char* text) { return i < n && T(i).kind == Token::Punct && T(i).text == text; };
    auto operand_end = [&](size_t i) {  // can a (sub)expression end with this token?
        if (i >= n) return false;
        const Token& t = T(i);
        if (t.kind == Token::Ident) {
            static const std::set<std::string> kw = {"return", "if", "else", "for", "while", "do", "case", "in", "out", "inout", "const"};
            return kw.count(t.text) == 0;
        }
        if (t.kind == Token::Number) return true;
        return t.kind == Token::Punct && (t.text == ")" || t.text == "]");
    };
    auto open_of = [&](size_t close) -> size_t {  // index of the bracket the one at `close` closes, or n
        const std::string c = T(close).text, o = c == ")" ? "(" : "[";
        int depth = 0;
        for (size_t i = close + 1; i-- > 0;) {
            if (is(i, c.c_str())) ++depth;
            else if (is(i, o.c_str()) && --depth == 0) return i;
        }
        return n;
    };
    auto close_of = [&](size_t open) -> size_t {
        const std::string o = T(open).text, c = o == "(" ? ")" : "]";
        int depth = 0;
        for (size_t i = open; i < n; ++i) {
            if (is(i, o.c_str())) ++depth;
            else if (is(i, c.c_str()) && --depth == 0) return i;
        }
        return n;
    };
    auto is_prefix_op = [&](size_t i) {  // a unary operator: one of - + ! ~ ++ -- that does not follow an operand
        if (i >= n || T(i).kind != Token::Punct) return false;
        const std::string& t = T(i).text;
        if (!(t == "-" || t == "+" || t == "!" || t == "~" || t == "++" || t == "--")) return false;
        return i == 0 || !operand_end(i - 1);
    };
    // start of the unary expression that ends at `e` (inclusive); n when the text is not understood
    auto unary_start = [&](size_t e) -> size_t {
        size_t p = e;
        if ((is(p, "++") || is(p, "--")) && p > 0 && operand_end(p - 1)) --p;  // postfix increment
        for (;;) {
            if (p >= n) return n;
            const Token& t = T(p);
            if (t.kind == Token::Punct && (t.text == ")" || t.text == "]")) {
                const size_t o = open_of(p);
                if (o == n) return n;
                if (t.text == "]") {  // base[...]: go on with the base
                    if (o == 0) return n;
                    p = o - 1;
                    continue;
                }
                p = o;
                if (o > 0 && T(o - 1).kind == Token::Ident && operand_end(o - 1)) p = o - 1;  // a call or a constructor
                else break;  // a parenthesised expression
            } else if (t.kind != Token::Ident && t.kind != Token::Number) {
                return n;
            }
            if (p > 0 && is(p - 1, ".")) {  // member of something: go on with the object
                if (p < 2) return n;
                p -= 2;
                continue;
            }
            break;
        }
        while (p > 0 && is_prefix_op(p - 1)) --p;
        return p;
    };
    // end (inclusive) of the unary expression that starts at `b`
    auto unary_end = [&](size_t b) -> size_t {
        size_t p = b;
        while (is_prefix_op(p)) ++p;
        if (p >= n) return n;
        if (is(p, "(")) {
            p = close_of(p);
            if (p == n) return n;
        } else if (T(p).kind != Token::Ident && T(p).kind != Token::Number) {
            return n;
        }
        for (;;) {  // postfix chain
            if (is(p + 1, "(") || is(p + 1, "[")) {
                p = close_of(p + 1);
                if (p == n) return n;
            } else if (is(p + 1, ".") && p + 2 < n && T(p + 2).kind == Token::Ident) {
                p += 2;
            } else if ((is(p + 1, "++") || is(p + 1, "--"))) {
                p += 1;
            } else {
                return p;
            }
        }
    };
    struct Insert { size_t at; int order; std::string text; };  // `at`: significant-token index the text goes in FRONT of (n: the end)
    std::vector<Insert> inserts;
    std::vector<std::pair<size_t, std::string>> retext;  // significant-token index -> replacement text of that token
    for (size_t k = 0; k < n; ++k) {
        auto fail = [&](const char* what) {
            std::string near;
            for (size_t i = k > 4 ? k - 4 : 0; i < n && i < k + 5; ++i) near += T(i).text + " ";
            throw std::runtime_error(std::string("GLSL -> HIP translation: cannot find the ") + what + " operand of the division near `" + near + "`");
        };
        auto ends_operand = [&](size_t i) { return operand_end(i) || ((is(i, "++") || is(i, "--")) && i > 0 && operand_end(i - 1)); };
        if (is(k, "/")) {
            if (k == 0 || !ends_operand(k - 1)) fail("left");
            size_t left = unary_start(k - 1);
            if (left == n) fail("left");
            while (left > 1 && (is(left - 1, "*") || is(left - 1, "/") || is(left - 1, "%")) && operand_end(left - 2)) {
                const size_t prev = unary_start(left - 2);
                if (prev == n) break;
                left = prev;
            }
            const size_t right = unary_end(k + 1);
            if (right == n) fail("right");
            inserts.push_back({left, 0, "ptl_div("});
            retext.push_back({k, ","});
            inserts.push_back({right + 1, 1, ")"});
        } else if (is(k, "/=")) {
            if (k == 0 || !operand_end(k - 1)) fail("left");
            if (T(k - 1).kind == Token::Ident && k >= 2 && is(k - 2, ".") && !swizzle_indices(T(k - 1).text).empty()) continue;  // v.xy /= e
            const size_t left = unary_start(k - 1);
            if (left == n) fail("left");
            size_t end = k + 1;  // the assignment expression: up to the `;`, `,` or closing bracket of this nesting level
            int depth = 0;
            for (; end < n; ++end) {
                if (is(end, "(") || is(end, "[")) ++depth;
                else if (is(end, ")") || is(end, "]")) {
                    if (depth == 0) break;
                    --depth;
                } else if (depth == 0 && (is(end, ";") || is(end, ","))) break;
            }
            if (end == k + 1) fail("right");
            inserts.push_back({left, 0, "ptl_div_assign("});
            retext.push_back({k, ","});
            inserts.push_back({end, 1, ")"});
        }
    }
    if (inserts.empty()) return;
    // an opener and a closer never share a position; several openers or several closers at one place keep their order
    std::stable_sort(inserts.begin(), inserts.end(), [](const Insert& a, const Insert& b) { return a.at < b.at || (a.at == b.at && a.order > b.order); });
    for (auto& r : retext) toks[sig[r.first]].text = r.second;
    std::vector<Token> out;
    out.reserve(toks.size() + inserts.size());
    size_t next = 0;
    for (size_t k = 0; k <= toks.size(); ++k) {
        // a closer belongs right behind the last token of its operand, not behind the white space and comments that follow it
        while (next < inserts.size() && (inserts[next].at >= n ? k == toks.size() : (inserts[next].order == 1 ? sig[inserts[next].at - 1] + 1 == k : sig[inserts[next].at] == k)))
            out.push_back({Token::Raw, inserts[next++].text});
        if (k < toks.size()) out.push_back(toks[k]);
    }
    toks.swap(out);
}

}  // namespace

// GLSL passes `out` / `inout` arguments by value-result: copied in at the call, copied out at the return (GLSL ES 3.00 section 6.1.1).
// The translation passes C++ references.  The two agree unless the callee can reach the argument under another name while it runs:
//   (a) the argument is a mutable global that the callee (or something it calls) also names -- GLSL keeps the global's old value
//       until the return, a reference changes it at once;
//   (b) the same variable is given to two out / inout parameters of one call -- GLSL leaves the order of the copies back undefined, so
//       the reference's picture would depend on its GL driver.
// Neither occurs in the reference's scenes (tests/test_host_logic.py runs the corpus through this).  A scene that does either is
// refused here instead of silently drawing something the reference might not: VERDICT r2 "semantic gaps" #8.
void check_out_argument_aliasing(const std::vector<std::string>& sources, const std::vector<std::string>& bodies) {
    struct Sig { const Token* tok; };
    struct Function {
        std::vector<char> out;          // per parameter: out / inout?
        std::set<std::string> names;    // identifiers of the body
        std::set<std::string> calls;
    };
    static const std::set<std::string> not_a_type = {"return", "if", "else", "for", "while", "do", "switch", "case", "in", "out", "inout", "const", "uniform", "struct", "break", "continue", "discard"};
    std::multimap<std::string, Function> functions;
    std::set<std::string> globals;
    std::vector<std::vector<Token>> streams;
    std::vector<std::vector<size_t>> sigs;        // indices of the tokens that are not white space, per stream
    std::vector<std::set<size_t>> definitions;    // positions (in sigs) of function names at their definition
    std::vector<const std::string*> texts;
    for (const std::string& text : sources) texts.push_back(&text);
    for (const std::string& text : bodies) texts.push_back(&text);  // statements of a generated function: calls, no definitions
    for (const std::string* text : texts) {
        streams.push_back(tokenize(*text));
        std::vector<size_t> sig;
        for (size_t k = 0; k < streams.back().size(); ++k) {
            Token::Kind kind = streams.back()[k].kind;
            if (kind != Token::Space && kind != Token::Comment && kind != Token::Preproc) sig.push_back(k);
        }
        sigs.push_back(std::move(sig));
        definitions.emplace_back();
    }
    auto matching = [](const std::vector<Token>& toks, const std::vector<size_t>& sig, size_t open, const char* o, const char* c) {
        int depth = 0;
        for (size_t k = open; k < sig.size(); ++k) {
            const Token& t = toks[sig[k]];
            if (t.kind != Token::Punct) continue;
            if (t.text == o) ++depth;
            else if (t.text == c && --depth == 0) return k;
        }
        return sig.size();
    };
    // pass 1: what is defined at file scope
    for (size_t f = 0; f < sources.size(); ++f) {
        const std::vector<Token>& toks = streams[f];
        const std::vector<size_t>& sig = sigs[f];
        auto tk = [&](size_t k) -> const Token& { return toks[sig[k]]; };
        for (size_t k = 0; k < sig.size();) {
            if (tk(k).kind == Token::Punct && tk(k).text == "{") {  // a struct body (or anything else braced at file scope)
                k = matching(toks, sig, k, "{", "}") + 1;
                continue;
            }
            bool decl = tk(k).kind == Token::Ident && !not_a_type.count(tk(k).text) && k + 2 < sig.size() && tk(k + 1).kind == Token::Ident && tk(k + 2).kind == Token::Punct;
            if (!decl) {
                ++k;
                continue;
            }
            bool qualified = false;  // `const float x`, `uniform ...`: nothing a function could write
            for (size_t b = k; b-- > 0;) {
                if (tk(b).kind == Token::Punct && (tk(b).text == ";" || tk(b).text == "}")) break;
                if (tk(b).kind == Token::Ident && (tk(b).text == "const" || tk(b).text == "uniform" || tk(b).text == "in")) qualified = true;
            }
            const std::string& name = tk(k + 1).text;
            const std::string& after = tk(k + 2).text;
            if (after == "(") {
                size_t close = matching(toks, sig, k + 2, "(", ")");
                if (close + 1 < sig.size() && tk(close + 1).kind == Token::Punct && tk(close + 1).text == "{") {
                    Function fn;
                    bool is_out = false, any = false;
                    int depth = 0;
                    for (size_t a = k + 3; a < close; ++a) {
                        const Token& t = tk(a);
                        if (t.kind == Token::Punct && (t.text == "(" || t.text == "[")) ++depth;
                        else if (t.kind == Token::Punct && (t.text == ")" || t.text == "]")) --depth;
                        else if (t.kind == Token::Punct && t.text == "," && depth == 0) {
                            fn.out.push_back(is_out);
                            is_out = false;
                            continue;
                        } else if (t.kind == Token::Ident && (t.text == "out" || t.text == "inout")) is_out = true;
                        any = true;
                    }
                    if (any && !(close == k + 4 && tk(k + 3).text == "void")) fn.out.push_back(is_out);
                    size_t end = matching(toks, sig, close + 1, "{", "}");
                    for (size_t a = close + 2; a < end; ++a)
                        if (tk(a).kind == Token::Ident) {
                            fn.names.insert(tk(a).text);
                            if (a + 1 < end && tk(a + 1).kind == Token::Punct && tk(a + 1).text == "(") fn.calls.insert(tk(a).text);
                        }
                    definitions[f].insert(k + 1);
                    functions.emplace(name, std::move(fn));
                    k = end + 1;
                    continue;
                }
                k = close + 1;
                continue;
            }
            if (!qualified && (after == ";" || after == "=" || after == "[" || after == ",")) {
                // `float a, b = 1.0, c;` : every declarator up to the `;`
                globals.insert(name);
                int depth = 0;
                size_t a = k + 2;
                for (; a < sig.size(); ++a) {
                    const Token& t = tk(a);
                    if (t.kind != Token::Punct) continue;
                    if (t.text == "(" || t.text == "[" || t.text == "{") ++depth;
                    else if (t.text == ")" || t.text == "]" || t.text == "}") --depth;
                    else if (t.text == ";" && depth == 0) break;
                    else if (t.text == "," && depth == 0 && a + 1 < sig.size() && tk(a + 1).kind == Token::Ident) globals.insert(tk(a + 1).text);
                }
                k = a + 1;
                continue;
            }
            ++k;
        }
    }
    bool any_out = false;
    for (auto& [name, fn] : functions)
        for (char o : fn.out) any_out = any_out || o;
    if (!any_out) return;
    // which mutable globals can a call of `name` touch (through anything it calls)
    auto reach = [&](const std::string& name) {
        std::set<std::string> seen_fn, found;
        std::vector<std::string> todo = {name};
        while (!todo.empty()) {
            std::string cur = todo.back();
            todo.pop_back();
            if (!seen_fn.insert(cur).second) continue;
            auto range = functions.equal_range(cur);
            for (auto it = range.first; it != range.second; ++it) {
                for (const std::string& g : globals)
                    if (it->second.names.count(g)) found.insert(g);
                for (const std::string& c : it->second.calls) todo.push_back(c);
            }
        }
        return found;
    };
    // pass 2: every call of a function with out / inout parameters
    for (size_t f = 0; f < streams.size(); ++f) {
        const std::vector<Token>& toks = streams[f];
        const std::vector<size_t>& sig = sigs[f];
        auto tk = [&](size_t k) -> const Token& { return toks[sig[k]]; };
        for (size_t k = 0; k + 1 < sig.size(); ++k) {
            if (tk(k).kind != Token::Ident || !functions.count(tk(k).text) || definitions[f].count(k)) continue;
            if (!(tk(k + 1).kind == Token::Punct && tk(k + 1).text == "(")) continue;
            if (k > 0 && tk(k - 1).kind == Token::Punct && tk(k - 1).text == ".") continue;
            size_t close = matching(toks, sig, k + 1, "(", ")");
            std::vector<std::string> roots, whole;  // first identifier of every argument; the argument without white space
            {
                int depth = 0;
                std::string root, text;
                bool have = false;
                for (size_t a = k + 2; a < close; ++a) {
                    const Token& t = tk(a);
                    if (t.kind == Token::Punct && (t.text == "(" || t.text == "[")) ++depth;
                    else if (t.kind == Token::Punct && (t.text == ")" || t.text == "]")) --depth;
                    else if (t.kind == Token::Punct && t.text == "," && depth == 0) {
                        roots.push_back(root);
                        whole.push_back(text);
                        root.clear();
                        text.clear();
                        have = false;
                        continue;
                    }
                    text += t.text;
                    if (!have && t.kind == Token::Ident) {
                        root = t.text;
                        have = true;
                    }
                }
                if (close > k + 2) {
                    roots.push_back(root);
                    whole.push_back(text);
                }
            }
            // may two lvalues name the same storage?  `v.x` / `v.y` and `s.a` / `s.b` do not; `v` / `v.x`, `v.xy` / `v.yz`, anything indexed may
            auto overlap = [](const std::string& a, const std::string& b) {
                if (a == b) return true;
                if (a.find('[') != std::string::npos || b.find('[') != std::string::npos) return true;
                auto parts = [](const std::string& t) {
                    std::vector<std::string> out(1);
                    for (char c : t)
                        if (c == '.') out.emplace_back();
                        else out.back() += c;
                    return out;
                };
                std::vector<std::string> pa = parts(a), pb = parts(b);
                for (size_t c = 0;; ++c) {
                    if (c == pa.size() || c == pb.size()) return true;  // one is the whole of which the other is a part
                    if (pa[c] == pb[c]) continue;
                    std::vector<int> ia = swizzle_indices(pa[c]), ib = swizzle_indices(pb[c]);
                    if (ia.empty() || ib.empty()) return false;         // two different fields
                    for (int x : ia)
                        for (int y : ib)
                            if (x == y) return true;
                    return false;
                }
            };
            auto range = functions.equal_range(tk(k).text);
            for (auto it = range.first; it != range.second; ++it) {
                const Function& fn = it->second;
                if (fn.out.size() != roots.size()) continue;
                std::set<std::string> touched;
                bool computed = false;
                for (size_t j = 0; j < roots.size(); ++j) {
                    if (!fn.out[j] || roots[j].empty()) continue;
                    for (size_t j2 = j + 1; j2 < roots.size(); ++j2)
                        if (fn.out[j2] && roots[j2] == roots[j] && overlap(whole[j], whole[j2]))
                            throw std::runtime_error("`" + tk(k).text + "(...)`: `" + roots[j] + "` is given to two out / inout parameters of one call; GLSL leaves the order "
                                                     "in which they are copied back undefined, so the picture would depend on the GL driver");
                    if (!globals.count(roots[j])) continue;
                    if (!computed) {
                        touched = reach(tk(k).text);
                        computed = true;
                    }
                    if (touched.count(roots[j]))
                        throw std::runtime_error("`" + tk(k).text + "(...)`: the global `" + roots[j] + "` is an out / inout argument of a function that also names it; "
                                                 "GLSL copies the argument back at the return, this translation passes a reference -- the two would differ.  Pass a local "
                                                 "and assign it afterwards");
                }
            }
        }
    }
}

// Token indices of the return types of the function DEFINITIONS of a file-scope GLSL text: `type name ( ... ) {` at brace depth 0.
// (Prototypes, struct heads, globals with initialisers and whatever follows a #define are not definitions.)
static std::set<size_t> definition_heads(const std::vector<Token>& toks) {
    std::set<size_t> heads;
    auto sig = [&](size_t k) {  // the next token that is neither space nor comment, or toks.size()
        while (k < toks.size() && (toks[k].kind == Token::Space || toks[k].kind == Token::Comment)) ++k;
        return k;
    };
    int brace = 0;
    bool in_define = false;
    for (size_t k = 0; k < toks.size(); ++k) {
        const Token& t = toks[k];
        if (t.kind == Token::Preproc) in_define = true;
        if (t.kind == Token::Space && t.text.find('\n') != std::string::npos) in_define = false;
        if (t.kind == Token::Punct && t.text == "{") ++brace;
        if (t.kind == Token::Punct && t.text == "}") --brace;
        if (in_define || brace != 0 || t.kind != Token::Ident) continue;
        const size_t name = sig(k + 1);
        if (name >= toks.size() || toks[name].kind != Token::Ident) continue;
        size_t open = sig(name + 1);
        if (open >= toks.size() || toks[open].kind != Token::Punct || toks[open].text != "(") continue;
        int depth = 0;
        size_t close = open;
        for (; close < toks.size(); ++close) {
            if (toks[close].kind != Token::Punct) continue;
            if (toks[close].text == "(") ++depth;
            if (toks[close].text == ")" && --depth == 0) break;
        }
        const size_t body = sig(close + 1);
        if (body < toks.size() && toks[body].kind == Token::Punct && toks[body].text == "{") heads.insert(k);
    }
    return heads;
}

std::string translate_glsl(const std::string& glsl, bool defer_loop_updates, bool force_inline_definitions) {
    std::vector<Token> toks = tokenize(glsl);
    if (defer_loop_updates) defer_loop_carried_updates(toks);
    rewrite_divisions(toks);
    const std::set<size_t> heads = force_inline_definitions ? definition_heads(toks) : std::set<size_t>();
    std::string out;
    out.reserve(glsl.size() + glsl.size() / 8);

    auto prev_sig = [&](size_t k) -> const Token* {
        while (k-- > 0)
            if (toks[k].kind != Token::Space && toks[k].kind != Token::Comment) return &toks[k];
        return nullptr;
    };
    auto next_sig = [&](size_t k) -> const Token* {
        for (++k; k < toks.size(); ++k)
            if (toks[k].kind != Token::Space && toks[k].kind != Token::Comment) return &toks[k];
        return nullptr;
    };

    // The translation works on tokens, not on types: `.xy` after ANY expression becomes a swizzle call.  A user struct with a field
    // that spells a swizzle (`struct S { vec2 st; }` ... `s.st`) would silently turn into s.sw<0,1>(): refuse it with a message
    // that names the field instead of leaving the scene author with a hiprtc error (or none).  The reference's corpus has no such field.
    {
        int brace = 0, struct_brace = -1;
        bool struct_head = false;
        for (size_t k = 0; k < toks.size(); ++k) {
            const Token& t = toks[k];
            if (t.kind == Token::Ident && t.text == "struct") struct_head = true;
            if (t.kind != Token::Punct && t.kind != Token::Ident) continue;
            if (t.kind == Token::Punct && t.text == "{") {
                ++brace;
                if (struct_head) {
                    struct_brace = brace;
                    struct_head = false;
                }
            } else if (t.kind == Token::Punct && t.text == "}") {
                if (brace == struct_brace) struct_brace = -1;
                --brace;
            } else if (t.kind == Token::Punct && t.text == ";") {
                struct_head = false;  // `struct S;` or a variable of struct type: no body
            } else if (t.kind == Token::Ident && struct_brace == brace && brace > 0) {
                const Token* nx = next_sig(k);
                if (nx && nx->kind == Token::Punct && (nx->text == ";" || nx->text == "," || nx->text == "[") && !swizzle_indices(t.text).empty())
                    throw std::runtime_error("struct field `" + t.text + "` spells a vector swizzle (xyzw / rgba / stpq): the GLSL -> HIP translation "
                                             "cannot tell `value." + t.text + "` from a swizzle; rename the field");
            }
        }
    }

    int paren_depth = 0;
    bool pending_ref = false;  // an `out` / `inout` qualifier was seen: next type name gets `&`
    for (size_t k = 0; k < toks.size(); ++k) {
        const Token& t = toks[k];
        switch (t.kind) {
            case Token::Space:
            case Token::Comment:
            case Token::Preproc:
            case Token::Raw:
                out += t.text;
                break;
            case Token::Number: {
                std::string s = t.text;
                if (is_float_literal(s)) {
                    char last = s.back();
                    if (last == 'F') s.back() = 'f';
                    else if (last != 'f') s += 'f';
                }
                out += s;
                break;
            }
            case Token::Punct:
                if (t.text == "(") ++paren_depth;
                if (t.text == ")") --paren_depth;
                out += t.text;
                break;
            case Token::Ident: {
                if (heads.count(k)) out += "PTL_FN ";  // on the line of the definition: the scene's line numbers stay what they are
                const Token* p = prev_sig(k);
                const Token* nx = next_sig(k);
                bool after_dot = p && p->kind == Token::Punct && p->text == ".";
                if (after_dot) {
                    std::vector<int> idx = swizzle_indices(t.text);
                    if (!idx.empty()) {
                        bool lvalue = nx && nx->kind == Token::Punct &&
                                      (nx->text == "=" || nx->text == "+=" || nx->text == "-=" || nx->text == "*=" || nx->text == "/=");
                        out += lvalue ? "swr<" : "sw<";
                        for (size_t c = 0; c < idx.size(); ++c) {
                            if (c) out += ',';
                            out += std::to_string(idx[c]);
                        }
                        out += ">()";
                        break;
                    }
                    out += t.text;  // ordinary member (or single component)
                    break;
                }
                // parameter qualifiers: only meaningful right after `(` or `,` inside parentheses
                bool param_pos = paren_depth > 0 && p && ((p->kind == Token::Punct && (p->text == "(" || p->text == ",")) ||
                                                          (p->kind == Token::Ident && p->text == "const"));  // `const in vec3 v`
                bool next_is_type = nx && nx->kind == Token::Ident;
                if (param_pos && next_is_type && (t.text == "in" || t.text == "out" || t.text == "inout")) {
                    if (t.text != "in") pending_ref = true;
                    break;  // drop the qualifier
                }
                if (t.text == "highp" || t.text == "mediump" || t.text == "lowp") break;
                if (cpp_only_keywords().count(t.text)) {
                    out += t.text + "_";
                    break;
                }
                out += t.text;
                if (pending_ref && next_is_type) {  // this identifier is the parameter's type
                    out += "&";
                    pending_ref = false;
                }
                break;
            }
        }
    }
    return out;
}

}  // namespace ptl
