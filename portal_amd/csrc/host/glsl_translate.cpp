// glsl_translate.cpp -- see glsl_translate.h.
#include "glsl_translate.h"
#include "glsl_tokens.h"

#include <stdexcept>

#include <algorithm>
#include <cctype>
#include <cstring>
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
        if (c == '#' && line_start) {  // preprocessor directive: keep the line verbatim
            size_t j = s.find('\n', i);
            if (j == std::string::npos) j = n;
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

}  // namespace

std::string translate_glsl(const std::string& glsl, bool defer_loop_updates) {
    std::vector<Token> toks = tokenize(glsl);
    if (defer_loop_updates) defer_loop_carried_updates(toks);
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
