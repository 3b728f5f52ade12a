// glsl_hoist.cpp -- see glsl_hoist.h.
#include "glsl_hoist.h"

#include <algorithm>
#include <cstring>

#include "glsl_tokens.h"

namespace ptl {
namespace {

struct ParseError {};

struct Node {
    enum Kind { Lit, Ident, Paren, Unary, Binary, Ternary, Assign, Call, Member, Index, Post } kind = Lit;
    size_t b = 0, e = 0;  // significant-token span [b, e)
    std::vector<int> kids;
    std::string text;  // operator, identifier, function or member name
    // classification
    std::string type;  // GLSL type, "" = unknown
    bool uniform = false;
    bool leaf = false;  // reads at least one run-time uniform (directly or through a local)
    bool op = false;    // does arithmetic
    int cost = 0;       // ... roughly this many VALU instructions of it
    bool half = false;  // a Ray whose origin is uniform and whose direction is not (first-trip variants); then `uniform` is false
    int loop = 0;       // > 0: depends on the carried variables of that loop ...
    int phase = 0;      // ... 0 as they are before their updates in the iteration, 1 after, -1 mixed
};

bool float_type(const std::string& t) { return t == "float" || t == "vec2" || t == "vec3" || t == "vec4" || t == "mat2" || t == "mat3" || t == "mat4"; }
int vec_size(const std::string& t) { return t == "vec2" ? 2 : t == "vec3" ? 3 : t == "vec4" ? 4 : 0; }
int mat_size(const std::string& t) { return t == "mat2" ? 2 : t == "mat3" ? 3 : t == "mat4" ? 4 : 0; }
std::string vec_of(int n) { return n == 1 ? "float" : n == 2 ? "vec2" : n == 3 ? "vec3" : n == 4 ? "vec4" : ""; }
bool type_name(const std::string& t) {
    static const std::set<std::string> names = {"float", "int", "uint", "bool", "vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3",
                                                "uvec4", "bvec2", "bvec3", "bvec4", "mat2", "mat3", "mat4"};
    return names.count(t) != 0;
}
bool swizzle(const std::string& s) {
    if (s.empty() || s.size() > 4) return false;
    for (const char* set : {"xyzw", "rgba", "stpq"}) {
        bool all = true;
        for (char c : s) all = all && std::strchr(set, c) != nullptr;
        if (all) return true;
    }
    return false;
}
bool float_literal(const std::string& t) {
    if (t.size() > 1 && t[0] == '0' && (t[1] == 'x' || t[1] == 'X')) return false;
    for (char c : t)
        if (c == '.' || c == 'e' || c == 'E' || c == 'f' || c == 'F') return true;
    return false;
}

// GLSL built-ins (and the two pure helpers of the reference's library every scene uses) with the rule that types their result
enum class Ret { Arg0, Arg1, Arg2, Float, Vec3 };
const std::map<std::string, Ret>& pure_functions() {
    static const std::map<std::string, Ret> f = {
        {"normalize", Ret::Arg0}, {"abs", Ret::Arg0},   {"sign", Ret::Arg0},    {"floor", Ret::Arg0},       {"ceil", Ret::Arg0},     {"fract", Ret::Arg0},
        {"sqrt", Ret::Arg0},      {"inversesqrt", Ret::Arg0}, {"sin", Ret::Arg0}, {"cos", Ret::Arg0},       {"tan", Ret::Arg0},      {"asin", Ret::Arg0},
        {"acos", Ret::Arg0},      {"atan", Ret::Arg0},  {"exp", Ret::Arg0},     {"log", Ret::Arg0},         {"exp2", Ret::Arg0},     {"log2", Ret::Arg0},
        {"radians", Ret::Arg0},   {"degrees", Ret::Arg0}, {"pow", Ret::Arg0},   {"mod", Ret::Arg0},         {"min", Ret::Arg0},      {"max", Ret::Arg0},
        {"clamp", Ret::Arg0},     {"mix", Ret::Arg0},   {"step", Ret::Arg1},    {"smoothstep", Ret::Arg2},  {"length", Ret::Float},  {"distance", Ret::Float},
        {"dot", Ret::Float},      {"cross", Ret::Vec3}, {"reflect", Ret::Arg0}, {"transpose", Ret::Arg0},   {"inverse", Ret::Arg0},  {"determinant", Ret::Float},
        {"get_normal", Ret::Vec3}, {"sqr", Ret::Arg0},
    };
    return f;
}

struct Site {  // one parsed expression and where it stands
    enum Kind { Expr, DeclInit, Decl } kind = Expr;
    int root = -1;
    int depth = 0;        // blocks / controlled statements around it inside the function body
    int loop = 0;         // innermost loop it is in (its header included), 0 = none
    bool header = false;  // part of a loop header
    size_t stmt_b = 0, stmt_e = 0;  // for expression statements: the whole statement with its `;`
    std::string decl_type, decl_name;
    size_t name_at = 0;   // Decl / DeclInit: significant-token index of the declared name
};

struct LoopInfo {
    int id = 0, outer = 0, depth = 0;
    size_t kw = 0, header_open = 0, header_close = 0, body_b = 0, body_e = 0;  // body_b/e: first token of the body, one past its last
    bool canonical = false, has_continue = false, braced = false;
    std::string var, bound;
};

struct Carried {  // a local that is a function of the iteration number alone
    bool leaf = true;  // ... and of at least one run-time uniform (a chain of baked constants is the compiler's business)
    std::string name, type;
    int loop = 0;
    int update_site = -1;
};

struct Replacement {
    size_t b, e;  // significant tokens [b, e) replaced by `text` (b == e: an insertion in front of b)
    std::string text;
};

struct PrologueItem {
    size_t at = 0;
    int loop = 0;  // > 0: belongs into the table loop of that loop
    std::string text;
};

class Hoister {
  public:
    Hoister(const std::string& glsl, const HoistParams& params, int& next_member) : P(params), next_member_(next_member) {
        toks_ = tokenize_glsl(glsl);
        for (size_t k = 0; k < toks_.size(); ++k)
            if (toks_[k].kind != Token::Space && toks_[k].kind != Token::Comment) sig_.push_back(k);
        n_ = sig_.size();
    }

    HoistResult run(const std::string& original) {
        HoistResult out;
        out.glsl = original;
        for (size_t i = 0; i < n_; ++i)
            if (T(i).kind == Token::Preproc || T(i).kind == Token::Raw) return out;  // macros: the text is not what the compiler will see
        struct Fn {
            size_t body_b, body_e;
            std::vector<std::string> params;
        };
        std::vector<Fn> fns;
        if (P.body_only) {
            fns.push_back({0, n_, P.body_params});
        } else {
            int depth = 0;
            for (size_t i = 0; i < n_; ++i) {
                if (is(i, "{")) ++depth;
                else if (is(i, "}")) --depth;
                else if (depth == 0 && ident(i) && i + 2 < n_ && ident(i + 1) && is(i + 2, "(") && T(i).text != "return") {
                    size_t close = match(i + 2, "(", ")");
                    if (close >= n_ || !is(close + 1, "{")) continue;
                    size_t end = match(close + 1, "{", "}");
                    if (end >= n_) continue;
                    Fn f{close + 2, end, {}};
                    size_t last_ident = n_;
                    for (size_t j = i + 3; j <= close; ++j) {
                        if (ident(j)) last_ident = j;
                        if ((is(j, ",") || j == close) && last_ident < n_) {
                            f.params.push_back(T(last_ident).text);
                            last_ident = n_;
                        }
                    }
                    fns.push_back(f);
                    i = end;  // (depth stays 0)
                }
            }
        }
        for (const Fn& f : fns) {
            const size_t replacements_before = replacements_.size(), members_before = members_.size(), items_before = items_.size();
            const int counter_before = next_member_;
            try {
                function(f.body_b, f.body_e, f.params);
            } catch (const ParseError&) {  // leave this function exactly as written
                replacements_.resize(replacements_before);
                members_.resize(members_before);
                items_.resize(items_before);
                next_member_ = counter_before;
            }
        }
        if (members_.empty()) return out;
        out.glsl = rebuild();
        out.members = members_;
        out.prologue = prologue_;
        return out;
    }

  private:
    const HoistParams& P;
    int& next_member_;
    std::vector<Token> toks_;
    std::vector<size_t> sig_;
    size_t n_ = 0;
    std::vector<Node> nodes_;
    std::vector<Replacement> replacements_;
    std::vector<HoistedMember> members_;
    std::vector<PrologueItem> items_;
    std::string prologue_;

    // per function
    std::vector<Site> sites_;
    std::vector<LoopInfo> loops_;
    std::set<std::string> declared_;
    std::map<std::string, int> writes_;
    struct UniformLocal {
        std::string type;
        size_t name_at;  // significant-token index of the declared name
        bool leaf;       // its value depends on a run-time uniform (not just on literals)
    };
    std::map<std::string, UniformLocal> ulocals_;
    std::map<std::string, Carried> carried_;
    std::map<std::string, size_t> hlocals_;   // Ray locals with a uniform origin, written once: name -> index of the declared name
    std::map<std::string, Carried> hcarried_;  // ... and those a canonical loop advances by `V = transform(U, ... V ...)`

    const Token& T(size_t i) const { return toks_[sig_[i]]; }
    bool is(size_t i, const char* text) const { return i < n_ && T(i).kind == Token::Punct && T(i).text == text; }
    bool ident(size_t i) const { return i < n_ && T(i).kind == Token::Ident; }
    bool ident(size_t i, const char* text) const { return ident(i) && T(i).text == text; }
    size_t match(size_t open, const char* a, const char* b) const {
        int depth = 0;
        for (size_t i = open; i < n_; ++i) {
            if (is(i, a)) ++depth;
            else if (is(i, b) && --depth == 0) return i;
        }
        return n_;
    }
    std::string text_of(size_t b, size_t e) const {  // source text of significant tokens [b, e), on one line
        std::string s;
        if (b >= e) return s;
        for (size_t k = sig_[b]; k <= sig_[e - 1]; ++k) {
            const Token& t = toks_[k];
            if (t.kind == Token::Comment) s += ' ';
            else if (t.kind == Token::Space) s += ' ';
            else s += t.text;
        }
        return s;
    }

    // ---- expressions ---------------------------------------------------------------------------------------------------------
    int add(Node nd) {
        nodes_.push_back(std::move(nd));
        return (int)nodes_.size() - 1;
    }
    static int precedence(const std::string& op) {
        static const std::map<std::string, int> p = {{"*", 12}, {"/", 12}, {"%", 12}, {"+", 11}, {"-", 11}, {"<<", 10}, {">>", 10}, {"<", 9},  {">", 9},
                                                     {"<=", 9}, {">=", 9}, {"==", 8}, {"!=", 8}, {"&", 7},  {"^", 6},   {"|", 5},   {"&&", 4}, {"||", 3},
                                                     {"?", 2},  {"=", 1},  {"+=", 1}, {"-=", 1}, {"*=", 1}, {"/=", 1}};
        auto it = p.find(op);
        return it == p.end() ? 0 : it->second;
    }
    int parse_expr(size_t& pos, size_t end, int min_prec) {
        int lhs = parse_unary(pos, end);
        while (pos < end && T(pos).kind == Token::Punct) {
            const std::string op = T(pos).text;
            const int prec = precedence(op);
            if (prec == 0 || prec < min_prec) break;
            ++pos;
            Node nd;
            nd.b = nodes_[lhs].b;
            nd.text = op;
            if (op == "?") {
                int mid = parse_expr(pos, end, 1);
                if (!is(pos, ":")) throw ParseError{};
                ++pos;
                int rhs = parse_expr(pos, end, 2);
                nd.kind = Node::Ternary;
                nd.kids = {lhs, mid, rhs};
            } else if (prec == 1) {
                int rhs = parse_expr(pos, end, 1);
                nd.kind = Node::Assign;
                nd.kids = {lhs, rhs};
            } else {
                int rhs = parse_expr(pos, end, prec + 1);
                nd.kind = Node::Binary;
                nd.kids = {lhs, rhs};
            }
            nd.e = pos;
            lhs = add(nd);
        }
        return lhs;
    }
    int parse_unary(size_t& pos, size_t end) {
        if (pos >= end) throw ParseError{};
        if (T(pos).kind == Token::Punct) {
            const std::string op = T(pos).text;
            if (op == "-" || op == "+" || op == "!" || op == "~" || op == "++" || op == "--") {
                Node nd;
                nd.kind = Node::Unary;
                nd.b = pos;
                nd.text = op;
                ++pos;
                nd.kids = {parse_unary(pos, end)};
                nd.e = pos;
                return add(nd);
            }
        }
        return parse_postfix(pos, end);
    }
    int parse_postfix(size_t& pos, size_t end) {
        int cur = parse_primary(pos, end);
        while (pos < end) {
            Node nd;
            nd.b = nodes_[cur].b;
            if (is(pos, ".")) {
                if (!ident(pos + 1) || is(pos + 2, "(")) throw ParseError{};  // (a method call: .length())
                nd.kind = Node::Member;
                nd.text = T(pos + 1).text;
                nd.kids = {cur};
                pos += 2;
            } else if (is(pos, "[")) {
                ++pos;
                int idx = parse_expr(pos, end, 1);
                if (!is(pos, "]")) throw ParseError{};
                ++pos;
                nd.kind = Node::Index;
                nd.kids = {cur, idx};
            } else if (is(pos, "++") || is(pos, "--")) {
                nd.kind = Node::Post;
                nd.text = T(pos).text;
                nd.kids = {cur};
                ++pos;
            } else {
                break;
            }
            nd.e = pos;
            cur = add(nd);
        }
        return cur;
    }
    int parse_primary(size_t& pos, size_t end) {
        if (pos >= end) throw ParseError{};
        Node nd;
        nd.b = pos;
        const Token& t = T(pos);
        if (t.kind == Token::Number) {
            nd.kind = Node::Lit;
            nd.text = t.text;
            nd.e = ++pos;
            return add(nd);
        }
        if (t.kind == Token::Ident) {
            nd.text = t.text;
            if (is(pos + 1, "(")) {
                nd.kind = Node::Call;
                pos += 2;
                if (is(pos, ")")) {
                    ++pos;
                } else {
                    for (;;) {
                        nd.kids.push_back(parse_expr(pos, end, 1));
                        if (is(pos, ",")) {
                            ++pos;
                            continue;
                        }
                        if (!is(pos, ")")) throw ParseError{};
                        ++pos;
                        break;
                    }
                }
                nd.e = pos;
                return add(nd);
            }
            nd.kind = Node::Ident;
            nd.e = ++pos;
            return add(nd);
        }
        if (is(pos, "(")) {
            ++pos;
            nd.kind = Node::Paren;
            nd.kids = {parse_expr(pos, end, 1)};
            if (!is(pos, ")")) throw ParseError{};
            nd.e = ++pos;
            return add(nd);
        }
        throw ParseError{};
    }

    // ---- statements ----------------------------------------------------------------------------------------------------------
    // parses one expression that must end exactly at `stop` (a `;`, `)` ... the caller has found)
    int expression_until(size_t b, size_t stop) {
        size_t pos = b;
        int root = parse_expr(pos, stop, 1);
        if (pos != stop) throw ParseError{};
        return root;
    }
    size_t statement_end(size_t pos, size_t end) const {  // the `;` that ends the statement starting at pos (brackets balanced)
        int depth = 0;
        for (size_t i = pos; i < end; ++i) {
            if (is(i, "(") || is(i, "[")) ++depth;
            else if (is(i, ")") || is(i, "]")) --depth;
            else if (is(i, "{") || is(i, "}")) throw ParseError{};
            else if (is(i, ";") && depth == 0) return i;
        }
        throw ParseError{};
    }
    void controlled(size_t& pos, size_t end, int depth, int loop) {  // the statement an if / else / for / while governs
        if (is(pos, "{")) {
            size_t close = match(pos, "{", "}");
            if (close >= end) throw ParseError{};
            block(pos + 1, close, depth + 1, loop);
            pos = close + 1;
        } else {
            statement(pos, end, depth + 1, loop);
        }
    }
    void block(size_t pos, size_t end, int depth, int loop) {
        while (pos < end) statement(pos, end, depth, loop);
    }
    void declaration_or_expression(size_t& pos, size_t end, int depth, int loop, bool header) {
        size_t p = pos;
        while (ident(p, "const") || ident(p, "highp") || ident(p, "mediump") || ident(p, "lowp")) ++p;
        const bool declares = ident(p) && ident(p + 1) && (is(p + 2, "=") || is(p + 2, ";") || is(p + 2, ",") || is(p + 2, "["));
        const size_t semi = statement_end(pos, end);
        if (declares) {
            const std::string type = T(p).text;
            size_t q = p + 1;
            for (;;) {
                if (!ident(q)) throw ParseError{};
                Site s;
                s.depth = depth;
                s.loop = loop;
                s.header = header;
                s.decl_type = type;
                s.decl_name = T(q).text;
                s.name_at = q;
                declared_.insert(s.decl_name);
                ++writes_[s.decl_name];
                ++q;
                if (is(q, "[")) throw ParseError{};  // arrays: not handled
                if (is(q, "=")) {
                    ++q;
                    s.kind = Site::DeclInit;
                    s.root = parse_expr(q, semi, 1);
                } else {
                    s.kind = Site::Decl;
                }
                sites_.push_back(s);
                if (is(q, ",")) {
                    ++q;
                    continue;
                }
                if (q != semi) throw ParseError{};
                break;
            }
        } else {
            Site s;
            s.depth = depth;
            s.loop = loop;
            s.header = header;
            s.stmt_b = pos;
            s.stmt_e = semi + 1;
            s.root = expression_until(pos, semi);
            sites_.push_back(s);
        }
        pos = semi + 1;
    }
    void statement(size_t& pos, size_t end, int depth, int loop) {
        if (is(pos, ";")) {
            ++pos;
            return;
        }
        if (is(pos, "{")) {
            size_t close = match(pos, "{", "}");
            if (close >= end) throw ParseError{};
            block(pos + 1, close, depth + 1, loop);
            pos = close + 1;
            return;
        }
        if (ident(pos, "if")) {
            if (!is(pos + 1, "(")) throw ParseError{};
            size_t close = match(pos + 1, "(", ")");
            if (close >= end) throw ParseError{};
            Site s;
            s.depth = depth;
            s.loop = loop;
            s.root = expression_until(pos + 2, close);
            sites_.push_back(s);
            pos = close + 1;
            controlled(pos, end, depth, loop);
            if (ident(pos, "else")) {
                ++pos;
                controlled(pos, end, depth, loop);
            }
            return;
        }
        if (ident(pos, "for") || ident(pos, "while")) {
            const bool is_for = T(pos).text == "for";
            if (!is(pos + 1, "(")) throw ParseError{};
            size_t close = match(pos + 1, "(", ")");
            if (close >= end) throw ParseError{};
            LoopInfo L;
            L.id = (int)loops_.size() + 1;
            L.outer = loop;
            L.depth = depth;
            L.kw = pos;
            L.header_open = pos + 1;
            L.header_close = close;
            loops_.push_back(L);
            const size_t slot = loops_.size() - 1;
            if (is_for) {
                size_t p = pos + 2;
                // `for (int I = 0; I < BOUND; I++)`, token for token: the only loop shape whose trip number is known to be I
                if (ident(p, "int") && ident(p + 1) && is(p + 2, "=") && T(p + 3).kind == Token::Number && T(p + 3).text == "0" && is(p + 4, ";") &&
                    ident(p + 5) && T(p + 5).text == T(p + 1).text && is(p + 6, "<") && (ident(p + 7) || T(p + 7).kind == Token::Number) && is(p + 8, ";") &&
                    ident(p + 9) && T(p + 9).text == T(p + 1).text && is(p + 10, "++") && p + 11 == close) {
                    loops_[slot].canonical = true;
                    loops_[slot].var = T(p + 1).text;
                    loops_[slot].bound = T(p + 7).text;
                }
                if (is(p, ";")) ++p;
                else declaration_or_expression(p, close, depth + 1, L.id, true);  // init (ends at its `;`)
                size_t semi = statement_end(p, close);
                if (semi > p) {
                    Site s;
                    s.depth = depth + 1;
                    s.loop = L.id;
                    s.header = true;
                    s.root = expression_until(p, semi);
                    sites_.push_back(s);
                }
                p = semi + 1;
                if (p < close) {
                    Site s;
                    s.depth = depth + 1;
                    s.loop = L.id;
                    s.header = true;
                    s.root = expression_until(p, close);
                    sites_.push_back(s);
                }
            } else {
                Site s;
                s.depth = depth + 1;
                s.loop = L.id;
                s.header = true;
                s.root = expression_until(pos + 2, close);
                sites_.push_back(s);
            }
            pos = close + 1;
            loops_[slot].braced = is(pos, "{");
            loops_[slot].body_b = pos;
            controlled(pos, end, depth, L.id);
            loops_[slot].body_e = pos;
            return;
        }
        if (ident(pos, "return")) {
            const size_t semi = statement_end(pos, end);
            if (semi > pos + 1) {
                Site s;
                s.depth = depth;
                s.loop = loop;
                s.root = expression_until(pos + 1, semi);
                sites_.push_back(s);
            }
            pos = semi + 1;
            return;
        }
        if (ident(pos, "break") || ident(pos, "continue") || ident(pos, "discard")) {
            if (T(pos).text == "continue")
                for (int l = loop; l > 0; l = loops_[l - 1].outer) loops_[l - 1].has_continue = true;  // (only the innermost is skipped, but be strict)
            if (!is(pos + 1, ";")) throw ParseError{};
            pos += 2;
            return;
        }
        if (ident(pos, "do") || ident(pos, "switch") || ident(pos, "case") || ident(pos, "default") || ident(pos, "struct") || ident(pos, "goto") ||
            ident(pos, "else"))
            throw ParseError{};
        declaration_or_expression(pos, end, depth, loop, false);
    }

    // ---- analysis ------------------------------------------------------------------------------------------------------------
    std::string base_identifier(int id) const {
        const Node& nd = nodes_[id];
        if (nd.kind == Node::Ident) return nd.text;
        if (nd.kind == Node::Member || nd.kind == Node::Index || nd.kind == Node::Paren) return base_identifier(nd.kids[0]);
        return "";
    }
    void count_writes(int id) {
        const Node& nd = nodes_[id];
        if (nd.kind == Node::Assign) ++writes_[base_identifier(nd.kids[0])];
        if (nd.kind == Node::Post || (nd.kind == Node::Unary && (nd.text == "++" || nd.text == "--"))) ++writes_[base_identifier(nd.kids[0])];
        // ... and the GLSL built-ins that write through an argument (GLSL ES 3.00 8.3, 8.8): `float ip = 0.0; f = modf(x, ip);` leaves
        // `ip` written twice, not a write-once uniform local (the prelude has no modf / frexp today; the day it gets one this must hold)
        static const std::set<std::string> builtin_out = {"modf", "frexp", "umulExtended", "imulExtended", "uaddCarry", "usubBorrow"};
        if (nd.kind == Node::Call && (P.functions_with_out_params.count(nd.text) || builtin_out.count(nd.text)))
            for (int k : nd.kids) ++writes_[base_identifier(k)];
        for (int k : nd.kids) count_writes(k);
    }
    bool mentions(int id, const std::string& name) const {
        const Node& nd = nodes_[id];
        if (nd.kind == Node::Ident && nd.text == name) return true;
        for (int k : nd.kids)
            if (mentions(k, name)) return true;
        return false;
    }
    bool tokens_mention(size_t b, size_t e, const std::string& name) const {
        for (size_t i = b; i < e && i < n_; ++i)
            if (ident(i) && T(i).text == name && !(i > 0 && is(i - 1, "."))) return true;
        return false;
    }

    static void combine(Node& nd, const Node& kid) {
        nd.leaf = nd.leaf || kid.leaf;
        nd.op = nd.op || kid.op;
        nd.cost += kid.cost;
        if (kid.loop > 0) {
            if (nd.loop == 0) {
                nd.loop = kid.loop;
                nd.phase = kid.phase;
            } else if (nd.loop != kid.loop || nd.phase != kid.phase) {
                nd.phase = -1;
            }
        }
    }
    void classify(int id) {
        Node& nd = nodes_[id];
        for (int k : nd.kids) classify(k);
        nd.type.clear();
        nd.uniform = nd.leaf = nd.op = false;
        nd.loop = 0;
        nd.phase = 0;
        nd.cost = 0;
        nd.half = false;
        auto kid = [&](int i) -> const Node& { return nodes_[nodes_[id].kids[i]]; };
        switch (nd.kind) {
            case Node::Lit:
                nd.uniform = true;
                nd.type = float_literal(nd.text) ? "float" : ((nd.text.back() == 'u' || nd.text.back() == 'U') ? "uint" : "int");
                break;
            case Node::Ident: {
                if (nd.text == "true" || nd.text == "false") {
                    nd.uniform = true;
                    nd.type = "bool";
                    break;
                }
                auto c = carried_.find(nd.text);
                if (c != carried_.end()) {
                    const LoopInfo& L = loops_[c->second.loop - 1];
                    if (nd.b >= L.body_b && nd.b < L.body_e) {
                        const Site& upd = sites_[c->second.update_site];
                        nd.uniform = true;
                        nd.leaf = c->second.leaf;
                        nd.type = c->second.type;
                        nd.loop = c->second.loop;
                        nd.phase = nd.b < upd.stmt_b ? 0 : (nd.b >= upd.stmt_e ? 1 : 0);
                        break;
                    }
                }
                auto hc = hcarried_.find(nd.text);
                if (hc != hcarried_.end()) {
                    const LoopInfo& L = loops_[hc->second.loop - 1];
                    if (nd.b >= L.body_b && nd.b < L.body_e) {
                        const Site& upd = sites_[hc->second.update_site];
                        nd.half = nd.leaf = true;
                        nd.type = "Ray";
                        nd.loop = hc->second.loop;
                        nd.phase = nd.b < upd.stmt_b ? 0 : (nd.b >= upd.stmt_e ? 1 : 0);
                        break;
                    }
                }
                auto hl = hlocals_.find(nd.text);
                if (hl != hlocals_.end() && nd.b > hl->second) {
                    nd.half = nd.leaf = true;
                    nd.type = "Ray";
                    break;
                }
                if (std::find(P.origin_uniform_rays.begin(), P.origin_uniform_rays.end(), nd.text) != P.origin_uniform_rays.end() && writes_[nd.text] == 1 &&
                    !hlocals_.count(nd.text)) {
                    nd.half = nd.leaf = true;  // a parameter whose origin is the camera's
                    nd.type = "Ray";
                    break;
                }
                auto u = ulocals_.find(nd.text);
                if (u != ulocals_.end() && nd.b > u->second.name_at) {
                    nd.uniform = true;
                    nd.leaf = u->second.leaf;
                    nd.type = u->second.type;
                    break;
                }
                if (declared_.count(nd.text)) break;
                auto g = P.uniforms.find(nd.text);
                if (g != P.uniforms.end()) {
                    nd.uniform = nd.leaf = true;
                    nd.type = g->second;
                    break;
                }
                auto k = P.constants.find(nd.text);
                if (k != P.constants.end()) {
                    nd.uniform = true;
                    nd.type = k->second;
                }
                break;
            }
            case Node::Paren:
                nd.uniform = kid(0).uniform;
                nd.half = kid(0).half;
                nd.type = kid(0).type;
                combine(nd, kid(0));
                break;
            case Node::Unary:
                if (nd.text == "++" || nd.text == "--") break;
                nd.uniform = kid(0).uniform;
                nd.type = nd.text == "!" ? "bool" : kid(0).type;
                combine(nd, kid(0));
                break;
            case Node::Post:
            case Node::Assign:
                break;
            case Node::Binary: {
                const Node &a = kid(0), &b = kid(1);
                nd.uniform = a.uniform && b.uniform;
                combine(nd, a);
                combine(nd, b);
                const std::string& op = nd.text;
                if (op == "*" || op == "/" || op == "+" || op == "-") {
                    if (a.type == b.type) nd.type = a.type;
                    else if (a.type == "float" && (vec_size(b.type) || mat_size(b.type))) nd.type = b.type;
                    else if (b.type == "float" && (vec_size(a.type) || mat_size(a.type))) nd.type = a.type;
                    else if (op == "*" && mat_size(a.type) && mat_size(a.type) == vec_size(b.type)) nd.type = b.type;
                    else if (op == "*" && mat_size(b.type) && mat_size(b.type) == vec_size(a.type)) nd.type = a.type;
                    if (float_type(nd.type)) {
                        nd.op = true;
                        const int ma = mat_size(a.type), mb = mat_size(b.type), width = vec_size(nd.type) ? vec_size(nd.type) : (mat_size(nd.type) ? mat_size(nd.type) * mat_size(nd.type) : 1);
                        nd.cost += (op == "*" && ma && mb) ? ma * ma * ma : (op == "*" && (ma || mb) && vec_size(nd.type)) ? width * width : (op == "/" ? 10 : 1) * width;
                    } else if ((op == "/") && !float_type(nd.type)) nd.uniform = false;  // an integer quotient may trap where the snippet guards it
                } else if (op == "%" || op == "<<" || op == ">>" || op == "&" || op == "|" || op == "^") {
                    nd.type = a.type == b.type ? a.type : "";
                    if (op == "%") nd.uniform = false;
                } else {
                    nd.type = "bool";
                }
                break;
            }
            case Node::Ternary:
                nd.uniform = kid(0).uniform && kid(1).uniform && kid(2).uniform;
                for (int i = 0; i < 3; ++i) combine(nd, kid(i));
                nd.type = kid(1).type == kid(2).type ? kid(1).type : "";
                break;
            case Node::Call: {
                bool all = true;
                for (size_t i = 0; i < nd.kids.size(); ++i) {
                    all = all && kid((int)i).uniform;
                    combine(nd, kid((int)i));
                }
                if (nd.text == "transform" && nd.kids.size() == 2 && !declared_.count("transform") && !P.scene_functions.count("transform") &&
                    kid(0).uniform && kid(0).type == "mat4" && kid(1).half) {
                    nd.half = true;  // (combine() above has merged leaf / loop / phase / cost of both arguments)
                    nd.type = "Ray";
                    nd.op = true;
                    nd.cost += 16;
                    break;
                }
                if (type_name(nd.text)) {
                    nd.uniform = all && !nd.kids.empty();
                    nd.type = nd.text;
                    break;
                }
                auto f = pure_functions().find(nd.text);
                if (f == pure_functions().end() || declared_.count(nd.text) || P.scene_functions.count(nd.text) || nd.kids.empty()) {
                    nd.leaf = nd.op = false;
                    nd.loop = 0;
                    nd.cost = 0;
                    break;
                }
                nd.uniform = all;
                nd.op = true;
                nd.cost += 8;
                const size_t want = f->second == Ret::Arg1 ? 1 : f->second == Ret::Arg2 ? 2 : 0;
                if (f->second == Ret::Float) nd.type = "float";
                else if (f->second == Ret::Vec3) nd.type = "vec3";
                else if (want < nd.kids.size()) nd.type = kid((int)want).type;
                if (!float_type(nd.type)) nd.type.clear();  // (an integer overload: not ours to type)
                break;
            }
            case Node::Member:
                if (kid(0).half && nd.text == "o") {
                    nd.uniform = true;
                    nd.type = "vec4";
                    combine(nd, kid(0));
                    break;
                }
                if (kid(0).uniform && vec_size(kid(0).type) && swizzle(nd.text)) {
                    nd.uniform = true;
                    nd.type = vec_of((int)nd.text.size());
                    combine(nd, kid(0));
                }
                break;
            case Node::Index:
                if (kid(0).uniform && kid(1).uniform && (mat_size(kid(0).type) || vec_size(kid(0).type))) {
                    nd.uniform = true;
                    nd.type = mat_size(kid(0).type) ? vec_of(mat_size(kid(0).type)) : "float";
                    combine(nd, kid(0));
                    combine(nd, kid(1));
                }
                break;
        }
        if (!nd.uniform && !nd.half) {
            nd.leaf = nd.op = false;
            nd.loop = 0;
            nd.phase = 0;
            nd.cost = 0;
        }
    }
    static bool placeable(const Node& nd) { return nd.uniform && nd.leaf && (nd.loop == 0 || nd.phase >= 0); }
    // a Ray with a uniform origin that is more than a name: its origin arithmetic can come from the prologue
    static bool half_hoistable(const Node& nd) { return nd.half && nd.leaf && nd.op && (nd.loop == 0 || nd.phase >= 0); }
    // worth a member: a scalar load replaces it, so one multiplication alone is not
    static bool hoistable(const Node& nd) { return placeable(nd) && nd.op && nd.cost >= 3 && float_type(nd.type); }

    // ---- rewriting -----------------------------------------------------------------------------------------------------------
    std::string guard_name(int loop) const { return "ptl_tab_ok_" + std::to_string(loop_tags_.at(loop)); }
    std::map<int, int> loop_tags_;  // loop id (per function) -> number unique in the kernel
    bool half_used_ = false;        // this function got members that hold ray origins: the prologue needs the rays themselves

    // a new member holding `expr_text` (evaluated where `like` stands); returns the expression that reads it
    std::string member_for(const std::string& type, const Node& like, const std::string& expr_text) {
        HoistedMember m;
        m.type = type;
        m.name = "ptl_hv" + std::to_string(next_member_++);
        m.length = like.loop > 0 ? kTableEntries : 0;
        members_.push_back(m);
        PrologueItem it;
        it.at = like.b;
        it.loop = like.loop;
        if (like.loop > 0) {
            it.text = "PTL_DV_OUT." + m.name + "[ptl_k] = " + expr_text + ";";
            const std::string index = loops_[like.loop - 1].var + (like.phase == 1 ? " + 1" : "");
            items_.push_back(it);
            return "(" + guard_name(like.loop) + " ? PTL_U." + m.name + "[" + index + "] : (" + expr_text + "))";
        }
        it.text = "PTL_DV_OUT." + m.name + " = " + expr_text + ";";
        items_.push_back(it);
        return "PTL_U." + m.name;
    }
    void hoist_in(int id) {
        const Node& nd = nodes_[id];
        if (half_hoistable(nd)) {
            const std::string text = text_of(nd.b, nd.e);
            HoistedMember m;
            m.type = "vec4";
            m.name = "ptl_hv" + std::to_string(next_member_++);
            m.length = nd.loop > 0 ? kTableEntries : 0;
            members_.push_back(m);
            half_used_ = true;
            PrologueItem it;
            it.at = nd.b;
            it.loop = nd.loop;
            if (nd.loop > 0) {
                it.text = "PTL_DV_OUT." + m.name + "[ptl_k] = (" + text + ").o;";
                items_.push_back(it);
                const std::string index = loops_[nd.loop - 1].var + (nd.phase == 1 ? " + 1" : "");
                replacements_.push_back({nd.b, nd.e, "(" + guard_name(nd.loop) + " ? ptl_ray_o(" + text + ", PTL_U." + m.name + "[" + index + "]) : (" + text + "))"});
            } else {
                it.text = "PTL_DV_OUT." + m.name + " = (" + text + ").o;";
                items_.push_back(it);
                replacements_.push_back({nd.b, nd.e, "ptl_ray_o(" + text + ", PTL_U." + m.name + ")"});
            }
            return;
        }
        if (hoistable(nd)) {
            const std::string text = text_of(nd.b, nd.e);
            replacements_.push_back({nd.b, nd.e, member_for(nd.type, nd, text)});
            return;
        }
        if (nd.kind == Node::Call && !declared_.count(nd.text) && !P.scene_functions.count(nd.text)) {
            auto staged_arg = [&](int k) { return k < (int)nd.kids.size() && placeable(nodes_[nd.kids[k]]) && nodes_[nd.kids[k]].type == "vec3"; };
            if ((nd.text == "normalize_normal" && nd.kids.size() == 2 && staged_arg(0)) || (nd.text == "plane_intersect" && nd.kids.size() == 3 && staged_arg(2))) {
                const int which = nd.text == "normalize_normal" ? 0 : 2;
                const Node& arg = nodes_[nd.kids[which]];
                replacements_.push_back({nd.b, nd.b + 1, nd.text == "normalize_normal" ? "ptl_normalize_normal_unit" : "ptl_plane_intersect_unit"});
                replacements_.push_back({arg.b, arg.e, member_for("vec3", arg, "normalize(" + text_of(arg.b, arg.e) + ")")});
                for (int k = 0; k < (int)nd.kids.size(); ++k)
                    if (k != which) hoist_in(nd.kids[k]);
                return;
            }
            if (nd.text == "is_collinear" && nd.kids.size() == 2 && (staged_arg(1) || staged_arg(0))) {
                const int which = staged_arg(1) ? 1 : 0;
                const Node& arg = nodes_[nd.kids[which]];
                replacements_.push_back({nd.b, nd.b + 1, which == 1 ? "ptl_is_collinear_len" : "ptl_is_collinear_len0"});
                replacements_.push_back({nd.e - 1, nd.e - 1, ", " + member_for("float", arg, "length(" + text_of(arg.b, arg.e) + ")")});
                for (int k : nd.kids) hoist_in(k);
                return;
            }
        }
        for (int k : nd.kids) hoist_in(k);
    }

    // `Ray V = <ray with a uniform origin>;` at the top level of the function: V keeps that property if nothing else writes it, or if
    // the one other write is `V = <such a ray built from V>;` directly in the body of a canonical loop that also READS V directly in
    // its body (a chain that is only read in nested blocks is left to the translator's deferred updates, which skip it altogether).
    void half_ray_local(size_t si, size_t body_e) {
        const Site& s = sites_[si];
        const int w = writes_[s.decl_name];
        if (w != 1 && w != 2) return;
        classify(s.root);
        if (!nodes_[s.root].half || nodes_[s.root].loop != 0) return;
        if (w == 1) {
            hlocals_[s.decl_name] = s.name_at;
            return;
        }
        for (size_t ui = si + 1; ui < sites_.size(); ++ui) {
            const Site& u = sites_[ui];
            if (u.kind != Site::Expr || u.header || u.loop == 0 || u.stmt_e == 0) continue;
            const Node& root = nodes_[u.root];
            if (root.kind != Node::Assign || root.text != "=" || nodes_[root.kids[0]].kind != Node::Ident || nodes_[root.kids[0]].text != s.decl_name) continue;
            const LoopInfo& L = loops_[u.loop - 1];
            if (!L.canonical || !L.braced || L.outer != 0 || L.depth != 0 || L.has_continue || u.depth != L.depth + 1) return;
            if (writes_[L.var] != 2 || (declared_.count(L.bound) && writes_[L.bound] != 1)) return;
            if (!(is(L.kw - 1, ";") || is(L.kw - 1, "{") || is(L.kw - 1, "}"))) return;
            if (tokens_mention(L.header_open, L.header_close, s.decl_name) || tokens_mention(L.body_e, body_e, s.decl_name)) return;
            bool read_in_the_body = false;
            for (size_t k = 0; k < sites_.size(); ++k)
                read_in_the_body = read_in_the_body || (k != ui && sites_[k].root >= 0 && !sites_[k].header && sites_[k].loop == u.loop &&
                                                        sites_[k].depth == L.depth + 1 && mentions(sites_[k].root, s.decl_name));
            if (!read_in_the_body) return;
            Carried c;
            c.name = s.decl_name;
            c.type = "Ray";
            c.loop = u.loop;
            c.update_site = (int)ui;
            hcarried_[c.name] = c;
            hlocals_[c.name] = s.name_at;  // between its declaration and the loop: an ordinary ray with a uniform origin
            classify(root.kids[1]);
            const Node& rhs = nodes_[root.kids[1]];
            if (!rhs.half || (rhs.loop != 0 && rhs.loop != u.loop) || rhs.phase < 0 || mentions(root.kids[1], L.var)) {
                hcarried_.erase(c.name);
                hlocals_.erase(c.name);
            }
            return;
        }
    }

    void function(size_t body_b, size_t body_e, const std::vector<std::string>& params) {
        sites_.clear();
        loops_.clear();
        declared_.clear();
        writes_.clear();
        ulocals_.clear();
        carried_.clear();
        hlocals_.clear();
        hcarried_.clear();
        loop_tags_.clear();
        half_used_ = false;
        const size_t first_item = items_.size();
        for (auto& p : params) {
            declared_.insert(p);
            ++writes_[p];
        }
        block(body_b, body_e, 0, 0);
        for (const Site& s : sites_)
            if (s.root >= 0) count_writes(s.root);

        // locals that are uniform values (written once: their declaration) or uniform sequences (once more: their update in a loop)
        for (size_t si = 0; si < sites_.size(); ++si) {
            const Site& s = sites_[si];
            if (s.kind == Site::DeclInit && s.depth == 0 && s.loop == 0 && s.decl_type == "Ray" && !P.origin_uniform_rays.empty()) {
                half_ray_local(si, body_e);
                continue;
            }
            if (s.kind != Site::DeclInit || s.depth != 0 || s.loop != 0 || !type_name(s.decl_type)) continue;
            const int w = writes_[s.decl_name];
            if (w != 1 && w != 2) continue;
            classify(s.root);
            const Node& init = nodes_[s.root];
            if (!init.uniform || init.loop != 0) continue;
            if (!init.type.empty() && init.type != s.decl_type) continue;  // (an implicit conversion we do not model)
            if (w == 1) {
                ulocals_[s.decl_name] = {s.decl_type, s.name_at, init.leaf};
                continue;
            }
            // w == 2: the other write must be `V = <uniform of V>;` directly in the body of a canonical loop of the function's top level
            for (size_t ui = si + 1; ui < sites_.size(); ++ui) {
                const Site& u = sites_[ui];
                if (u.kind != Site::Expr || u.header || u.loop == 0 || u.stmt_e == 0) continue;
                const Node& root = nodes_[u.root];
                if (root.kind != Node::Assign || root.text != "=" || nodes_[root.kids[0]].kind != Node::Ident || nodes_[root.kids[0]].text != s.decl_name) continue;
                const LoopInfo& L = loops_[u.loop - 1];
                if (!L.canonical || !L.braced || L.outer != 0 || L.depth != 0 || L.has_continue || u.depth != L.depth + 1) break;
                if (writes_[L.var] != 2 || (declared_.count(L.bound) && writes_[L.bound] != 1)) break;
                if (!(is(L.kw - 1, ";") || is(L.kw - 1, "{") || is(L.kw - 1, "}") || L.kw == body_b)) break;
                if (tokens_mention(L.header_open, L.header_close, s.decl_name) || tokens_mention(L.body_e, body_e, s.decl_name)) break;
                Carried c;
                c.name = s.decl_name;
                c.type = s.decl_type;
                c.loop = u.loop;
                c.update_site = (int)ui;
                c.leaf = false;  // (first without: does the chain read a run-time uniform besides itself?)
                carried_[c.name] = c;
                ulocals_[c.name] = {s.decl_type, s.name_at, init.leaf};  // between its declaration and the loop it is an ordinary uniform local
                classify(root.kids[1]);
                const Node& rhs = nodes_[root.kids[1]];
                if (!rhs.uniform || rhs.type != s.decl_type || (rhs.loop != 0 && rhs.loop != u.loop) || rhs.phase < 0 || mentions(root.kids[1], L.var) ||
                    !(init.leaf || rhs.leaf)) {
                    carried_.erase(c.name);
                    ulocals_.erase(c.name);
                } else {
                    carried_[c.name].leaf = true;
                }
                break;
            }
        }
        // an update may read another carried variable: every one of them has to have survived
        for (bool again = true; again;) {
            again = false;
            for (auto it = carried_.begin(); it != carried_.end(); ++it) {
                const Node& root = nodes_[sites_[it->second.update_site].root];
                classify(root.kids[1]);
                if (!nodes_[root.kids[1]].uniform) {
                    ulocals_.erase(it->first);
                    carried_.erase(it);
                    again = true;
                    break;
                }
            }
        }
        for (auto& c : carried_)
            if (!loop_tags_.count(c.second.loop)) loop_tags_[c.second.loop] = next_member_++;  // (a number unique in the kernel)
        for (auto& c : hcarried_)
            if (!loop_tags_.count(c.second.loop)) loop_tags_[c.second.loop] = next_member_++;

        // the hoist itself
        const size_t members_before = members_.size();
        for (size_t si = 0; si < sites_.size(); ++si) {
            const Site& s = sites_[si];
            if (s.root < 0 || s.header) continue;
            bool is_update = false;
            for (auto& c : carried_) is_update = is_update || c.second.update_site == (int)si;
            for (auto& c : hcarried_) is_update = is_update || c.second.update_site == (int)si;
            if (is_update) continue;
            classify(s.root);
            hoist_in(s.root);
        }
        if (members_.size() == members_before && carried_.empty() && hcarried_.empty()) {  // nothing to do here: drop what the analysis has prepared
            items_.resize(first_item);
            return;
        }
        // carried variables: their own tables, the guarded updates, the guards
        std::set<int> used_loops;
        for (size_t k = first_item; k < items_.size(); ++k)
            if (items_[k].loop > 0) used_loops.insert(items_[k].loop);
        for (auto& entry : carried_) {
            const Carried& c = entry.second;
            const Site& u = sites_[c.update_site];
            const LoopInfo& L = loops_[c.loop - 1];
            const Node& root = nodes_[u.root];
            const std::string rhs = text_of(nodes_[root.kids[1]].b, nodes_[root.kids[1]].e);
            HoistedMember m;
            m.type = c.type;
            m.name = "ptl_hv" + std::to_string(next_member_++);
            m.length = kTableEntries;
            members_.push_back(m);
            used_loops.insert(c.loop);
            PrologueItem table;  // first in the table loop: the value this iteration starts with
            table.at = L.body_b;
            table.loop = c.loop;
            table.text = "PTL_DV_OUT." + m.name + "[ptl_k] = " + c.name + ";";
            items_.push_back(table);
            PrologueItem step;  // last: the update itself
            step.at = u.stmt_b;
            step.loop = -c.loop;
            step.text = c.name + " = " + rhs + ";";
            items_.push_back(step);
            replacements_.push_back({u.stmt_b, u.stmt_e, "if (" + guard_name(c.loop) + ") " + c.name + " = PTL_U." + m.name + "[" + L.var + " + 1]; else " + c.name + " = " + rhs + ";"});
        }
        for (auto& entry : hcarried_) {  // the same for rays that carry a uniform origin: a table of origins, the update keeps its direction half
            const Carried& c = entry.second;
            const Site& u = sites_[c.update_site];
            const LoopInfo& L = loops_[c.loop - 1];
            const Node& root = nodes_[u.root];
            const std::string rhs = text_of(nodes_[root.kids[1]].b, nodes_[root.kids[1]].e);
            HoistedMember m;
            m.type = "vec4";
            m.name = "ptl_hv" + std::to_string(next_member_++);
            m.length = kTableEntries;
            members_.push_back(m);
            used_loops.insert(c.loop);
            half_used_ = true;
            PrologueItem table;
            table.at = L.body_b;
            table.loop = c.loop;
            table.text = "PTL_DV_OUT." + m.name + "[ptl_k] = " + c.name + ".o;";
            items_.push_back(table);
            PrologueItem step;
            step.at = u.stmt_b;
            step.loop = -c.loop;
            step.text = c.name + " = " + rhs + ";";
            items_.push_back(step);
            replacements_.push_back({u.stmt_b, u.stmt_e, "if (" + guard_name(c.loop) + ") " + c.name + " = ptl_ray_o(" + rhs + ", PTL_U." + m.name + "[" + L.var + " + 1]); else " + c.name + " = " + rhs + ";"});
        }
        for (int l : used_loops) {
            const LoopInfo& L = loops_[l - 1];
            replacements_.push_back({L.kw, L.kw, "const bool " + guard_name(l) + " = (" + L.bound + ") <= " + std::to_string(kTableLoop) + "; "});
        }
        // uniform locals: the prologue needs them wherever a hoisted expression names one
        for (auto& u : ulocals_) {
            for (const Site& s : sites_) {
                if (s.kind != Site::DeclInit || s.decl_name != u.first || s.name_at != u.second.name_at) continue;
                PrologueItem decl;
                decl.at = s.name_at;
                decl.text = s.decl_type + " " + s.decl_name + " = " + text_of(nodes_[s.root].b, nodes_[s.root].e) + ";";
                items_.push_back(decl);
            }
        }
        // The rays themselves: the parameters as dummies with the right origin, the locals as declared.  Not only when a ray
        // expression was rewritten (half_used_): a plain uniform expression or uniform local may read `r.o` -- or `q.o` of a copy
        // `Ray q = r;` -- of an origin-uniform ray (Member-on-half is classed uniform), and then the prologue names the ray too.
        if (half_used_ || (!P.origin_uniform_rays.empty() && items_.size() > first_item)) {
            for (const std::string& p : P.origin_uniform_rays) {
                PrologueItem param;
                param.at = body_b;
                param.text = "Ray " + p + " = Ray(" + P.origin_expr + ", vec4(0.0), 1.0, false);";
                items_.push_back(param);
            }
            for (auto& h : hlocals_) {
                for (const Site& s : sites_) {
                    if (s.kind != Site::DeclInit || s.decl_name != h.first || s.name_at != h.second) continue;
                    PrologueItem decl;
                    decl.at = s.name_at;
                    decl.text = "Ray " + s.decl_name + " = " + text_of(nodes_[s.root].b, nodes_[s.root].e) + ";";
                    items_.push_back(decl);
                }
            }
        }
        // assemble this function's block: source order; the items of a loop inside one table loop at the loop's place, updates last
        std::vector<PrologueItem> mine(items_.begin() + (long)first_item, items_.end());
        items_.resize(first_item);
        std::stable_sort(mine.begin(), mine.end(), [](const PrologueItem& a, const PrologueItem& b) { return a.at < b.at; });
        std::string text = "{\n";
        std::set<int> emitted;
        for (const PrologueItem& it : mine) {
            if (it.loop == 0) {
                text += "    " + it.text + "\n";
                continue;
            }
            const int l = it.loop > 0 ? it.loop : -it.loop;
            if (emitted.count(l)) continue;
            emitted.insert(l);
            text += "    for (int ptl_k = 0; ptl_k < " + std::to_string(kTableEntries) + "; ptl_k++) {\n";
            for (const PrologueItem& in : mine)
                if (in.loop == l) text += "        " + in.text + "\n";
            for (const PrologueItem& in : mine)
                if (in.loop == -l) text += "        " + in.text + "\n";
            text += "    }\n";
        }
        text += "}\n";
        prologue_ += text;
    }

    std::string rebuild() {
        std::stable_sort(replacements_.begin(), replacements_.end(), [](const Replacement& a, const Replacement& b) { return a.b < b.b || (a.b == b.b && a.e < b.e); });
        std::vector<bool> drop(toks_.size(), false);
        for (const Replacement& r : replacements_)
            if (r.e > r.b)
                for (size_t k = sig_[r.b]; k <= sig_[r.e - 1]; ++k) drop[k] = true;
        std::string out;
        size_t next = 0;
        for (size_t k = 0; k <= toks_.size(); ++k) {
            while (next < replacements_.size() && (replacements_[next].b >= n_ ? k == toks_.size() : sig_[replacements_[next].b] == k)) out += replacements_[next++].text;
            if (k == toks_.size()) break;
            if (!drop[k]) out += toks_[k].text;
            else if (toks_[k].kind == Token::Space && toks_[k].text == "\n") out += "\n";  // line for line
        }
        return out;
    }
};

}  // namespace

HoistResult hoist_uniform_work(const std::string& glsl, const HoistParams& params, int& next_member) {
    const int before = next_member;
    try {
        Hoister h(glsl, params, next_member);
        return h.run(glsl);
    } catch (...) {
        next_member = before;
        HoistResult r;
        r.glsl = glsl;
        return r;
    }
}

}  // namespace ptl
