// formula.cpp -- see formula.h.
#include "formula.h"

#include <cctype>
#include <cmath>
#include <cstdlib>

namespace ptl {
namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kE = 2.71828182845904523536028747135266250;

// Binary operators in ascending binding strength (fasteval's BinaryOp order).
enum Op { OR = 1, AND, NE, EQ, GTE, LTE, GT, LT, ADD, SUB, MUL, DIV, MOD, EXP };
bool is_comparison(Op o) { return o >= NE && o <= LT; }

struct PExpr;
struct PValue {
    enum Kind { Const, Neg, Pos, Not, Paren, Var, Func } kind = Const;
    double c = 0.0;
    std::string name;
    std::shared_ptr<PValue> inner;             // Neg / Pos / Not
    std::shared_ptr<PExpr> paren;              // Paren
    std::vector<std::shared_ptr<PExpr>> args;  // Func
};
struct PExpr {
    PValue first;
    std::vector<std::pair<Op, PValue>> pairs;
};

struct Parser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    explicit Parser(const std::string& t) : s(t) {}

    void ws() {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    }
    bool fail(const std::string& m) {
        if (err.empty()) err = m + " at offset " + std::to_string(i);
        return false;
    }
    bool parse_expr(PExpr& out, int depth) {
        if (depth > 64) return fail("expression too deep");
        if (!parse_value(out.first, depth)) return false;
        for (;;) {
            ws();
            Op op;
            if (!peek_op(op)) break;
            PValue v;
            if (!parse_value(v, depth)) return false;
            out.pairs.emplace_back(op, std::move(v));
        }
        return true;
    }
    bool peek_op(Op& op) {
        if (i >= s.size()) return false;
        auto two = [&](const char* lit) { return s.compare(i, 2, lit) == 0; };
        char c = s[i];
        if (two("||")) { op = OR; i += 2; return true; }
        if (two("&&")) { op = AND; i += 2; return true; }
        if (two("!=")) { op = NE; i += 2; return true; }
        if (two("==")) { op = EQ; i += 2; return true; }
        if (two(">=")) { op = GTE; i += 2; return true; }
        if (two("<=")) { op = LTE; i += 2; return true; }
        if (s.compare(i, 2, "or") == 0 && (i + 2 >= s.size() || !std::isalnum((unsigned char)s[i + 2]))) { op = OR; i += 2; return true; }
        if (s.compare(i, 3, "and") == 0 && (i + 3 >= s.size() || !std::isalnum((unsigned char)s[i + 3]))) { op = AND; i += 3; return true; }
        switch (c) {
            case '>': op = GT; break;
            case '<': op = LT; break;
            case '+': op = ADD; break;
            case '-': op = SUB; break;
            case '*': op = MUL; break;
            case '/': op = DIV; break;
            case '%': op = MOD; break;
            case '^': op = EXP; break;
            default: return false;
        }
        ++i;
        return true;
    }
    bool parse_value(PValue& out, int depth) {
        ws();
        if (i >= s.size()) return fail("unexpected end of formula");
        char c = s[i];
        if (std::isdigit((unsigned char)c) || c == '.') return parse_const(out);
        if (c == '-' || c == '+' || c == '!') {
            ++i;
            out.kind = c == '-' ? PValue::Neg : (c == '+' ? PValue::Pos : PValue::Not);
            out.inner = std::make_shared<PValue>();
            return parse_value(*out.inner, depth + 1);
        }
        if (c == '(' || c == '[') {
            char close = c == '(' ? ')' : ']';
            ++i;
            out.kind = PValue::Paren;
            out.paren = std::make_shared<PExpr>();
            if (!parse_expr(*out.paren, depth + 1)) return false;
            ws();
            if (i >= s.size() || s[i] != close) return fail("missing closing bracket");
            ++i;
            return true;
        }
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t j = i;
            while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
            out.name = s.substr(i, j - i);
            i = j;
            ws();
            if (i < s.size() && (s[i] == '(' || s[i] == '[')) {
                char close = s[i] == '(' ? ')' : ']';
                ++i;
                out.kind = PValue::Func;
                for (;;) {
                    ws();
                    if (i < s.size() && s[i] == close) {
                        ++i;
                        break;
                    }
                    auto e = std::make_shared<PExpr>();
                    if (!parse_expr(*e, depth + 1)) return false;
                    out.args.push_back(e);
                    ws();
                    if (i < s.size() && (s[i] == ',' || s[i] == ';')) ++i;
                    else if (i >= s.size() || s[i] != close) return fail("bad argument list");
                }
            } else {
                out.kind = PValue::Var;
            }
            return true;
        }
        return fail(std::string("unexpected character `") + c + "`");
    }
    bool parse_const(PValue& out) {
        const char* begin = s.c_str() + i;
        char* end = nullptr;
        double v = std::strtod(begin, &end);
        if (end == begin) return fail("bad number");
        i += (size_t)(end - begin);
        // fasteval's SI suffixes
        if (i < s.size()) {
            double mul = 0.0;
            switch (s[i]) {
                case 'k': case 'K': mul = 1e3; break;
                case 'M': mul = 1e6; break;
                case 'G': mul = 1e9; break;
                case 'T': mul = 1e12; break;
                case 'm': mul = 1e-3; break;
                case 'u': mul = 1e-6; break;
                case 'n': mul = 1e-9; break;
                case 'p': mul = 1e-12; break;
                default: break;
            }
            if (mul != 0.0 && (i + 1 >= s.size() || !std::isalnum((unsigned char)s[i + 1]))) {
                v *= mul;
                ++i;
            }
        }
        out.kind = PValue::Const;
        out.c = v;
        return true;
    }
};

bool approx_eq(double l, double r) { return std::fabs(l - r) <= 8.0 * 2.220446049250313e-16; }

}  // namespace

struct Formula::Node {
    enum Kind {
        Const, Neg, Not, Inv, Add, Mul, Mod, Exp, Cmp, Or, And, Var, Func,
        // fasteval builtins
        FInt, FCeil, FFloor, FAbs, FSign, FLog, FRound, FMin, FMax,
        FSin, FCos, FTan, FAsin, FAcos, FAtan, FSinh, FCosh, FTanh, FAsinh, FAcosh, FAtanh
    } kind = Const;
    double c = 0.0;
    Op cmp = EQ;
    std::string name;
    std::vector<std::unique_ptr<Node>> kids;
};

namespace {
using Node = Formula::Node;
using NodeP = std::unique_ptr<Node>;

NodeP mk(Node::Kind k) {
    auto n = std::make_unique<Node>();
    n->kind = k;
    return n;
}
NodeP mk_const(double c) {
    auto n = mk(Node::Const);
    n->c = c;
    return n;
}
NodeP mk2(Node::Kind k, NodeP a, NodeP b) {
    auto n = mk(k);
    n->kids.push_back(std::move(a));
    n->kids.push_back(std::move(b));
    return n;
}
bool is_const(const NodeP& n) { return n->kind == Node::Const; }

struct Slice {
    const PValue* first;
    const std::pair<Op, PValue>* pairs;
    size_t n;
};

NodeP compile_value(const PValue& v);
NodeP compile_slice(const Slice& sl);
NodeP compile_expr(const PExpr& e) { return compile_slice(Slice{&e.first, e.pairs.data(), e.pairs.size()}); }

std::vector<Slice> split(const Slice& sl, const std::function<bool(Op)>& at) {
    std::vector<Slice> out;
    const PValue* first = sl.first;
    size_t start = 0;
    for (size_t k = 0; k < sl.n; ++k) {
        if (at(sl.pairs[k].first)) {
            out.push_back(Slice{first, sl.pairs + start, k - start});
            first = &sl.pairs[k].second;
            start = k + 1;
        }
    }
    out.push_back(Slice{first, sl.pairs + start, sl.n - start});
    return out;
}

NodeP neg_wrap(NodeP n) {
    if (is_const(n)) return mk_const(-n->c);
    if (n->kind == Node::Neg) return std::move(n->kids[0]);
    auto o = mk(Node::Neg);
    o->kids.push_back(std::move(n));
    return o;
}
NodeP inv_wrap(NodeP n) {
    if (is_const(n)) return mk_const(1.0 / n->c);
    if (n->kind == Node::Inv) return std::move(n->kids[0]);
    auto o = mk(Node::Inv);
    o->kids.push_back(std::move(n));
    return o;
}
NodeP combine(std::vector<NodeP> terms, Node::Kind kind, double identity) {
    NodeP out;
    double folded = identity;
    for (auto& t : terms) {
        if (is_const(t)) {
            folded = kind == Node::Add ? folded + t->c : folded * t->c;
        } else if (out) {
            out = mk2(kind, std::move(out), std::move(t));
        } else {
            out = std::move(t);
        }
    }
    if (!out) return mk_const(folded);
    if (folded != identity) out = mk2(kind, std::move(out), mk_const(folded));
    return out;
}

double eval_cmp(Op op, double l, double r) {
    switch (op) {
        case NE: return approx_eq(l, r) ? 0.0 : 1.0;
        case EQ: return approx_eq(l, r) ? 1.0 : 0.0;
        case GTE: return l >= r ? 1.0 : 0.0;
        case LTE: return l <= r ? 1.0 : 0.0;
        case GT: return l > r ? 1.0 : 0.0;
        case LT: return l < r ? 1.0 : 0.0;
        default: return 0.0;
    }
}

NodeP compile_slice(const Slice& sl) {
    if (sl.n == 0) return compile_value(*sl.first);
    Op lowest = sl.pairs[0].first;
    for (size_t k = 0; k < sl.n; ++k)
        if (sl.pairs[k].first < lowest) lowest = sl.pairs[k].first;

    if (is_comparison(lowest)) {  // all comparisons share one level, left to right
        const PValue* first = sl.first;
        size_t start = 0;
        NodeP acc;
        Op pending = EQ;
        for (size_t k = 0; k <= sl.n; ++k) {
            if (k == sl.n || is_comparison(sl.pairs[k].first)) {
                NodeP part = compile_slice(Slice{first, sl.pairs + start, k - start});
                if (!acc) {
                    acc = std::move(part);
                } else if (is_const(acc) && is_const(part)) {
                    acc = mk_const(eval_cmp(pending, acc->c, part->c));
                } else {
                    acc = mk2(Node::Cmp, std::move(acc), std::move(part));
                    acc->cmp = pending;
                }
                if (k < sl.n) {
                    pending = sl.pairs[k].first;
                    first = &sl.pairs[k].second;
                    start = k + 1;
                }
            }
        }
        return acc;
    }

    auto parts = split(sl, [&](Op o) { return o == lowest; });
    std::vector<NodeP> nodes;
    nodes.reserve(parts.size());
    for (auto& p : parts) nodes.push_back(compile_slice(p));
    switch (lowest) {
        case OR:
        case AND: {
            NodeP acc = std::move(nodes[0]);
            for (size_t k = 1; k < nodes.size(); ++k) acc = mk2(lowest == OR ? Node::Or : Node::And, std::move(acc), std::move(nodes[k]));
            return acc;
        }
        case ADD: return combine(std::move(nodes), Node::Add, 0.0);
        case SUB:
            for (size_t k = 1; k < nodes.size(); ++k) nodes[k] = neg_wrap(std::move(nodes[k]));
            return combine(std::move(nodes), Node::Add, 0.0);
        case MUL: return combine(std::move(nodes), Node::Mul, 1.0);
        case DIV:
            for (size_t k = 1; k < nodes.size(); ++k) nodes[k] = inv_wrap(std::move(nodes[k]));
            return combine(std::move(nodes), Node::Mul, 1.0);
        case MOD: {
            NodeP acc = std::move(nodes[0]);
            for (size_t k = 1; k < nodes.size(); ++k) {
                if (is_const(acc) && is_const(nodes[k])) acc = mk_const(std::fmod(acc->c, nodes[k]->c));
                else acc = mk2(Node::Mod, std::move(acc), std::move(nodes[k]));
            }
            return acc;
        }
        case EXP: {  // right to left
            NodeP acc = std::move(nodes.back());
            for (size_t k = nodes.size() - 1; k-- > 0;) {
                if (is_const(acc) && is_const(nodes[k])) acc = mk_const(std::pow(nodes[k]->c, acc->c));
                else acc = mk2(Node::Exp, std::move(nodes[k]), std::move(acc));
            }
            return acc;
        }
        default: return mk_const(std::nan(""));
    }
}

struct Builtin {
    const char* name;
    Node::Kind kind;
    int min_args, max_args;
};
const Builtin kBuiltins[] = {
    {"int", Node::FInt, 1, 1},     {"ceil", Node::FCeil, 1, 1},   {"floor", Node::FFloor, 1, 1},
    {"abs", Node::FAbs, 1, 1},     {"sign", Node::FSign, 1, 1},   {"log", Node::FLog, 1, 2},
    {"round", Node::FRound, 1, 2}, {"min", Node::FMin, 1, 64},    {"max", Node::FMax, 1, 64},
    {"sin", Node::FSin, 1, 1},     {"cos", Node::FCos, 1, 1},     {"tan", Node::FTan, 1, 1},
    {"asin", Node::FAsin, 1, 1},   {"acos", Node::FAcos, 1, 1},   {"atan", Node::FAtan, 1, 1},
    {"sinh", Node::FSinh, 1, 1},   {"cosh", Node::FCosh, 1, 1},   {"tanh", Node::FTanh, 1, 1},
    {"asinh", Node::FAsinh, 1, 1}, {"acosh", Node::FAcosh, 1, 1}, {"atanh", Node::FAtanh, 1, 1},
};

double eval_builtin(Node::Kind k, const std::vector<double>& a) {
    switch (k) {
        case Node::FInt: return std::trunc(a[0]);
        case Node::FCeil: return std::ceil(a[0]);
        case Node::FFloor: return std::floor(a[0]);
        case Node::FAbs: return std::fabs(a[0]);
        case Node::FSign: return std::isnan(a[0]) ? a[0] : (std::signbit(a[0]) ? -1.0 : 1.0);  // f64::signum
        case Node::FLog: {
            double base = a.size() == 2 ? a[0] : 10.0;
            double n = a.back();
            if (base == 2.0) return std::log2(n);
            if (base == 10.0) return std::log10(n);
            return std::log(n) / std::log(base);
        }
        case Node::FRound: {
            double modulus = a.size() == 2 ? a[0] : 1.0;
            return std::round(a.back() / modulus) * modulus;
        }
        // f64::min / max folded over the arguments: a NaN operand is ignored; which of +0 / -0 wins a tie is not defined by
        // Rust (llvm.minnum) -- here, as in the oracle, the earlier argument stays
        case Node::FMin: {
            double m = a[0];
            for (size_t i = 1; i < a.size(); ++i)
                if (std::isnan(m) || a[i] < m) m = a[i];
            return m;
        }
        case Node::FMax: {
            double m = a[0];
            for (size_t i = 1; i < a.size(); ++i)
                if (std::isnan(m) || a[i] > m) m = a[i];
            return m;
        }
        case Node::FSin: return std::sin(a[0]);
        case Node::FCos: return std::cos(a[0]);
        case Node::FTan: return std::tan(a[0]);
        case Node::FAsin: return std::asin(a[0]);
        case Node::FAcos: return std::acos(a[0]);
        case Node::FAtan: return std::atan(a[0]);
        case Node::FSinh: return std::sinh(a[0]);
        case Node::FCosh: return std::cosh(a[0]);
        case Node::FTanh: return std::tanh(a[0]);
        case Node::FAsinh: return std::asinh(a[0]);
        case Node::FAcosh: return std::acosh(a[0]);
        case Node::FAtanh: return std::atanh(a[0]);
        default: return std::nan("");
    }
}

NodeP compile_value(const PValue& v) {
    switch (v.kind) {
        case PValue::Const: return mk_const(v.c);
        case PValue::Pos: return compile_value(*v.inner);
        case PValue::Neg: return neg_wrap(compile_value(*v.inner));
        case PValue::Not: {
            NodeP in = compile_value(*v.inner);
            if (is_const(in)) return mk_const(approx_eq(in->c, 0.0) ? 1.0 : 0.0);
            auto n = mk(Node::Not);
            n->kids.push_back(std::move(in));
            return n;
        }
        case PValue::Paren: return compile_expr(*v.paren);
        case PValue::Var: {
            auto n = mk(Node::Var);
            n->name = v.name;
            return n;
        }
        case PValue::Func: {
            if (v.name == "pi" && v.args.empty()) return mk_const(kPi);
            if (v.name == "e" && v.args.empty()) return mk_const(kE);
            for (const Builtin& b : kBuiltins) {
                if (v.name != b.name) continue;
                if ((int)v.args.size() < b.min_args || (int)v.args.size() > b.max_args) break;
                auto n = mk(b.kind);
                bool all_const = true;
                std::vector<double> consts;
                for (auto& a : v.args) {
                    n->kids.push_back(compile_expr(*a));
                    all_const = all_const && is_const(n->kids.back());
                    if (all_const) consts.push_back(n->kids.back()->c);
                }
                if (all_const) return mk_const(eval_builtin(b.kind, consts));
                return n;
            }
            auto n = mk(Node::Func);
            n->name = v.name;
            for (auto& a : v.args) n->kids.push_back(compile_expr(*a));
            return n;
        }
    }
    return mk_const(std::nan(""));
}

std::optional<double> eval_node(const Node& n, const FormulaNamespace& ns) {
    auto kid = [&](size_t k) { return eval_node(*n.kids[k], ns); };
    switch (n.kind) {
        case Node::Const: return n.c;
        case Node::Neg: { auto a = kid(0); if (!a) return std::nullopt; return -*a; }
        case Node::Inv: { auto a = kid(0); if (!a) return std::nullopt; return 1.0 / *a; }
        case Node::Not: { auto a = kid(0); if (!a) return std::nullopt; return approx_eq(*a, 0.0) ? 1.0 : 0.0; }
        case Node::Add: { auto a = kid(0), b = kid(1); if (!a || !b) return std::nullopt; return *a + *b; }
        case Node::Mul: { auto a = kid(0), b = kid(1); if (!a || !b) return std::nullopt; return *a * *b; }
        case Node::Mod: { auto a = kid(0), b = kid(1); if (!a || !b) return std::nullopt; return std::fmod(*a, *b); }
        case Node::Exp: { auto a = kid(0), b = kid(1); if (!a || !b) return std::nullopt; return std::pow(*a, *b); }
        case Node::Cmp: { auto a = kid(0), b = kid(1); if (!a || !b) return std::nullopt; return eval_cmp(n.cmp, *a, *b); }
        case Node::Or: {
            auto a = kid(0);
            if (!a) return std::nullopt;
            if (!approx_eq(*a, 0.0)) return *a;
            return kid(1);
        }
        case Node::And: {
            auto a = kid(0);
            if (!a) return std::nullopt;
            if (approx_eq(*a, 0.0)) return *a;
            return kid(1);
        }
        case Node::Var: return ns(n.name, {});
        case Node::Func: {
            std::vector<double> args;
            for (size_t k = 0; k < n.kids.size(); ++k) {
                auto a = kid(k);
                if (!a) return std::nullopt;
                args.push_back(*a);
            }
            return ns(n.name, args);
        }
        default: {
            std::vector<double> args;
            for (size_t k = 0; k < n.kids.size(); ++k) {
                auto a = kid(k);
                if (!a) return std::nullopt;
                args.push_back(*a);
            }
            return eval_builtin(n.kind, args);
        }
    }
}

}  // namespace

Formula::~Formula() = default;

std::shared_ptr<Formula> Formula::compile(const std::string& text, std::string* error) {
    Parser p(text);
    PExpr e;
    bool ok = p.parse_expr(e, 0);
    if (ok) {
        p.ws();
        if (p.i != text.size()) ok = p.fail("trailing characters");
    }
    if (!ok) {
        if (error) *error = p.err;
        return nullptr;
    }
    std::shared_ptr<Formula> f(new Formula());
    f->root_ = compile_expr(e);
    return f;
}

std::optional<double> Formula::eval(const FormulaNamespace& ns) const { return eval_node(*root_, ns); }

// --- the reference's custom callback table: src/gui/uniform.rs:1014-1124, src/gui/easing.rs:6-45 ---
namespace {
double easing_in(double t) { return 1.0 - std::cos(t * kPi * 0.5); }
double easing_in_out(double t) { return (1.0 - std::cos(t * kPi)) * 0.5; }
bool is_one(double v) { return std::fabs(v - 1.0) < 1e-6; }
}  // namespace

std::optional<double> formula_custom_function(const std::string& name, const std::vector<double>& a, bool* known) {
    *known = true;
    auto need = [&](size_t n) { return a.size() >= n; };
    if (name == "if") { if (!need(1)) return std::nullopt; if (is_one(a[0])) { if (!need(2)) return std::nullopt; return a[1]; } if (!need(3)) return std::nullopt; return a[2]; }
    if (name == "and") { if (!need(1)) return std::nullopt; if (!is_one(a[0])) return 0.0; if (!need(2)) return std::nullopt; return is_one(a[1]) ? 1.0 : 0.0; }
    if (name == "or") { if (!need(1)) return std::nullopt; if (is_one(a[0])) return 1.0; if (!need(2)) return std::nullopt; return is_one(a[1]) ? 1.0 : 0.0; }
    if (name == "not") { if (!need(1)) return std::nullopt; return is_one(a[0]) ? 0.0 : 1.0; }
    if (name == "deg2rad") { if (!need(1)) return std::nullopt; return a[0] / 180.0 * kPi; }
    if (name == "rad2deg") { if (!need(1)) return std::nullopt; return a[0] * 180.0 / kPi; }
    if (name == "switch") {
        if (!need(1)) return std::nullopt;
        double idx = a[0];
        size_t k = idx > 0.0 ? (idx >= 1e18 ? (size_t)-1 : (size_t)idx) : 0;  // Rust `as usize` saturates
        if (k >= a.size()) return std::nullopt;
        return a[k];
    }
    if (name == "on") {
        if (!need(3)) return std::nullopt;
        double v = a[0], lo = a[1], hi = a[2];
        if (v < lo) return 0.0;
        if (v > hi) return 1.0;
        return (v - lo) / (hi - lo);
    }
    if (name == "inv") { if (!need(1)) return std::nullopt; return 1.0 - a[0]; }
    if (name == "sqrt") { if (!need(1)) return std::nullopt; return std::sqrt(a[0]); }
    if (name == "atan2") { if (!need(2)) return std::nullopt; return std::atan2(a[0], a[1]); }
    if (name == "easing_linear") { if (!need(1)) return std::nullopt; return a[0]; }
    if (name == "easing_in") { if (!need(1)) return std::nullopt; return easing_in(a[0]); }
    if (name == "easing_out") { if (!need(1)) return std::nullopt; return 1.0 - easing_in(1.0 - a[0]); }
    if (name == "easing_in_out") { if (!need(1)) return std::nullopt; return easing_in_out(a[0]); }
    if (name == "easing_in_out_fast") { if (!need(1)) return std::nullopt; return easing_in_out(easing_in_out(a[0])); }
    if (name == "easing_plus_minus") {
        if (!need(1)) return std::nullopt;
        double t = a[0] * (2.0 * kPi);
        double t2 = 2.0 * t;
        return std::sin(t) * (3.0 - std::cos(t) - std::cos(t2) - std::cos(t) * std::cos(t2)) / 4.0;
    }
    if (name == "easing_elastic_out") {
        if (!need(1)) return std::nullopt;
        double x = a[0];
        double c4 = (2.0 * kPi) / 3.0;
        if (x == 0.0) return 0.0;
        if (x == 1.0) return 1.0;
        return std::pow(2.0, -10.0 * x) * std::sin((x * 10.0 - 0.75) * c4) + 1.0;
    }
    if (name == "bump") {
        if (!need(3)) return std::nullopt;
        double x = (a[0] - a[1]) / a[2];
        if (std::fabs(x) < 1.0) return 0.5 * (1.0 + std::cos(kPi * x));
        return 0.0;
    }
    if (name == "later_start") {
        if (!need(2)) return std::nullopt;
        double t = a[0], time = 1.0 - a[1];
        double v = t / time - (1.0 - time) / time;
        return v > 0.0 ? v : 0.0;  // f64::max(0.0, v): NaN -> 0, a +-0 tie keeps the first operand
    }
    if (name == "early_finish") {
        if (!need(2)) return std::nullopt;
        double v = a[0] / a[1];
        return v < 1.0 ? v : 1.0;  // f64::min(1.0, v)
    }
    if (name == "lerp") {  // egui::lerp(a..=b, t)
        if (!need(3)) return std::nullopt;
        return (1.0 - a[2]) * a[0] + a[2] * a[1];
    }
    *known = false;
    return std::nullopt;
}

}  // namespace ptl
