// glsl_translate.cpp -- see glsl_translate.h.
#include "glsl_translate.h"

#include <stdexcept>

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

namespace {

struct Token {
    enum Kind { Space, Comment, Ident, Number, Punct, Preproc } kind;
    std::string text;
};

std::vector<Token> tokenize(const std::string& s) {
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

}  // namespace

std::string translate_glsl(const std::string& glsl) {
    std::vector<Token> toks = tokenize(glsl);
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
