// glsl_bound.cpp -- bound_nearer_blocks (glsl_translate.h): the caller's distance bound inside intersection-material snippets.
#include <set>
#include <string>
#include <vector>

#include "glsl_tokens.h"
#include "glsl_translate.h"

namespace ptl {
namespace {

struct Sig {  // the significant tokens of a snippet (no white space, comments, directives)
    std::vector<Token>& toks;
    std::vector<size_t> at;
    explicit Sig(std::vector<Token>& t) : toks(t) {
        for (size_t k = 0; k < t.size(); ++k)
            if (t[k].kind != Token::Space && t[k].kind != Token::Comment && t[k].kind != Token::Preproc) at.push_back(k);
    }
    size_t n() const { return at.size(); }
    const Token& operator[](size_t i) const { return toks[at[i]]; }
    bool is(size_t i, const char* text) const { return i < n() && (*this)[i].kind == Token::Punct && (*this)[i].text == text; }
    bool ident(size_t i, const std::string& text) const { return i < n() && (*this)[i].kind == Token::Ident && (*this)[i].text == text; }
    bool any_ident(size_t i) const { return i < n() && (*this)[i].kind == Token::Ident; }
    size_t close_of(size_t open) const {  // matching bracket of the one at `open`, or n()
        const std::string o = (*this)[open].text, c = o == "(" ? ")" : (o == "{" ? "}" : "]");
        int depth = 0;
        for (size_t i = open; i < n(); ++i) {
            if (is(i, o.c_str())) ++depth;
            else if (is(i, c.c_str()) && --depth == 0) return i;
        }
        return n();
    }
    // does the token sequence starting at i spell `words` (identifiers and punctuation alternately, as given)?
    bool spells(size_t i, std::initializer_list<const char*> words) const {
        for (const char* w : words) {
            if (i >= n() || (*this)[i].text != w) return false;
            ++i;
        }
        return true;
    }
};

bool is_assignment_op(const Token& t) {
    if (t.kind != Token::Punct) return false;
    return t.text == "=" || t.text == "+=" || t.text == "-=" || t.text == "*=" || t.text == "/=" || t.text == "++" || t.text == "--";
}

}  // namespace

// See glsl_translate.h.  Every condition is checked on the token stream; anything not recognised leaves the snippet as written.
std::string bound_nearer_blocks(const std::string& glsl_body, const std::set<std::string>& functions_with_out_params, int* bounded) {
    if (bounded) *bounded = 0;
    std::vector<Token> toks = tokenize_glsl(glsl_body);
    Sig s(toks);
    const size_t n = s.n();
    // (1) the accumulator: exactly one `SceneIntersectionWithMaterial R = ...;`, at brace depth 0, and the body ends in `return R;`
    std::string R;
    size_t decl = n;
    {
        int depth = 0;
        for (size_t i = 0; i < n; ++i) {
            if (s.is(i, "{")) ++depth;
            else if (s.is(i, "}")) --depth;
            else if (s.ident(i, "SceneIntersectionWithMaterial") && s.any_ident(i + 1) && (s.is(i + 2, "=") || s.is(i + 2, ";"))) {
                if (!R.empty() || depth != 0) return glsl_body;  // a second accumulator, or one that lives in a block
                R = s[i + 1].text;
                decl = i + 1;
            }
        }
    }
    if (R.empty() || n < 3 || !(s.ident(n - 3, "return") && s.ident(n - 2, R) && s.is(n - 1, ";"))) return glsl_body;
    // (the initial value must be "no hit": the invariant below starts from it)
    if (!(s.is(decl + 1, "=") && s.spells(decl + 2, {"SceneIntersectionWithMaterial", "(", "scene_intersection_none", ",", "material_empty", "(", ")", ")", ";"}))) return glsl_body;

    // (2) the blocks `if (nearer(R.scene.hit, H)) { ... }` without an else
    struct Block { size_t cond_close, open, close; std::string H; };
    std::vector<Block> blocks;
    for (size_t i = 0; i + 12 < n; ++i) {
        if (!(s.ident(i, "if") && s.is(i + 1, "(") && s.ident(i + 2, "nearer") && s.is(i + 3, "(") && s.ident(i + 4, R))) continue;
        if (!(s.is(i + 5, ".") && s.ident(i + 6, "scene") && s.is(i + 7, ".") && s.ident(i + 8, "hit") && s.is(i + 9, ",") && s.any_ident(i + 10) && s.is(i + 11, ")") && s.is(i + 12, ")")))
            return glsl_body;  // R inside a nearer() of another shape
        const size_t open = i + 13;
        if (!s.is(open, "{")) return glsl_body;
        const size_t close = s.close_of(open);
        if (close == n || s.ident(close + 1, "else")) return glsl_body;
        if (s[i + 10].text == R) return glsl_body;
        blocks.push_back({i + 12, open, close, s[i + 10].text});
    }
    if (blocks.empty()) return glsl_body;
    auto block_of = [&](size_t i) -> const Block* {
        for (const Block& b : blocks)
            if (i > b.open && i < b.close) return &b;
        return nullptr;
    };

    // (3) every other occurrence of R: inside one of the blocks, in one of three shapes
    std::vector<std::pair<size_t, size_t>> stores, reads;  // (block open, position)
    for (size_t i = 0; i < n; ++i) {
        if (!s.ident(i, R) || (i > 0 && s.is(i - 1, "."))) continue;
        if (i == decl || i == n - 2) continue;
        bool in_condition = false;
        for (const Block& b : blocks) in_condition = in_condition || (i + 8 == b.cond_close);  // the R of `nearer(R.scene.hit, H))`
        if (in_condition) continue;
        const Block* b = block_of(i);
        if (!b) return glsl_body;  // read or written outside the blocks: its value there may depend on candidates the bound skips
        // R.material = ...            (anything may be stored; who reads it is checked in (5))
        if (s.is(i + 1, ".") && s.ident(i + 2, "material") && s.is(i + 3, "=")) continue;
        // R.scene.material == / != X   (a read of what the block itself has just stored)
        if (s.is(i + 1, ".") && s.ident(i + 2, "scene") && s.is(i + 3, ".") && s.ident(i + 4, "material") && (s.is(i + 5, "==") || s.is(i + 5, "!="))) {
            reads.push_back({b->open, i});
            continue;
        }
        // R.scene = process_{portal,plane}_intersection(R.scene, H, ...): the library's conditional store of H (library.glsl:560-589)
        if (s.is(i + 1, ".") && s.ident(i + 2, "scene") && s.is(i + 3, "=") && (s.ident(i + 4, "process_portal_intersection") || s.ident(i + 4, "process_plane_intersection")) &&
            s.is(i + 5, "(") && s.ident(i + 6, R) && s.is(i + 7, ".") && s.ident(i + 8, "scene") && s.is(i + 9, ",") && s.ident(i + 10, b->H) && s.is(i + 11, ",")) {
            stores.push_back({b->open, i});
            continue;
        }
        if (i >= 6 && s.ident(i - 6, R) && s.is(i - 5, ".") && s.ident(i - 4, "scene") && s.is(i - 3, "=") && s.is(i - 1, "(") && s.is(i + 1, ".") && s.ident(i + 2, "scene") && s.is(i + 3, ","))
            continue;  // (the second R of that statement)
        return glsl_body;
    }
    // ... and a block looks at what R holds (`R.scene.material ==`) only behind its own stores: in front of them it would see the previous
    // candidate's id, which is the one thing the bounded accumulator may not share with the one as written (a store guarded by it could
    // then go the other way)
    for (auto& rd : reads)
        for (auto& st : stores)
            if (rd.first == st.first && rd.second < st.second) return glsl_body;
    for (auto& rd : reads) {
        bool behind_a_store = false;
        for (auto& st : stores) behind_a_store = behind_a_store || (rd.first == st.first && st.second < rd.second);
        if (!behind_a_store) return glsl_body;
    }

    // (4) a block changes nothing but R and its own locals (a skipped block must leave no other trace), and H is not assigned in it
    for (const Block& b : blocks) {
        std::set<std::string> locals;
        for (size_t i = b.open + 1; i + 1 < b.close; ++i)  // `type name =` / `type name;` at the start of a statement
            if (s.any_ident(i) && s.any_ident(i + 1) && (s.is(i + 2, "=") || s.is(i + 2, ";")) && (s.is(i - 1, ";") || s.is(i - 1, "{") || s.is(i - 1, "}") || s.is(i - 1, ")")) && s[i].text != "return")
                locals.insert(s[i + 1].text);
        for (size_t i = b.open + 1; i < b.close; ++i) {
            if (!is_assignment_op(s[i])) continue;
            // the target: `++x` names it behind the operator, everything else in front -- walk back over `.member` and `[index]` to the base name
            const bool prefix = (s.is(i, "++") || s.is(i, "--")) && !(i > 0 && (s.any_ident(i - 1) || s.is(i - 1, "]")));
            size_t t = prefix ? i + 1 : i - 1;
            while (!prefix && t > 0) {
                if (s.is(t, "]")) {
                    int depth = 0;
                    while (t > 0) {
                        if (s.is(t, "]")) ++depth;
                        else if (s.is(t, "[") && --depth == 0) break;
                        --t;
                    }
                    if (t == 0) return glsl_body;
                    --t;
                } else if (s.any_ident(t) && t >= 2 && s.is(t - 1, ".")) {
                    t -= 2;
                } else {
                    break;
                }
            }
            if (!s.any_ident(t)) return glsl_body;
            const std::string& base = s[t].text;
            if (base == b.H) return glsl_body;
            if (base != R && !locals.count(base)) return glsl_body;
        }
        // ... nor through an `out` / `inout` argument: no function that has such parameters is called in a block
        for (size_t i = b.open + 1; i + 1 < b.close; ++i)
            if (s.any_ident(i) && s.is(i + 1, "(") && functions_with_out_params.count(s[i].text)) return glsl_body;
        // ... nor through control flow: a block that leaves a loop or the function (`break`, `continue`, `return`, `discard`) changes which LATER
        // candidates are reached at all, so skipping it could let a candidate through that the code as written never saw
        for (size_t i = b.open + 1; i < b.close; ++i)
            if (s.ident(i, "break") || s.ident(i, "continue") || s.ident(i, "return") || s.ident(i, "discard")) return glsl_body;
    }

    // (5) who reads R.material: the caller, and only when R.scene.material == CUSTOM_MATERIAL (src/frag.glsl:118-122).  A block that may leave
    // that id behind must store the material itself -- otherwise the caller would read what an EARLIER candidate stored, which the bound may
    // have skipped.  Required shape: behind every `R.scene = process_*(...)` statement, at the same nesting level, either `R.material = ...;`
    // or `if (R.scene.material == CUSTOM_MATERIAL) { ... R.material = ...; ... }`.
    for (const Block& b : blocks) {
        for (size_t i = b.open + 1; i < b.close; ++i) {
            if (!(s.ident(i, R) && s.is(i + 1, ".") && s.ident(i + 2, "scene") && s.is(i + 3, "="))) continue;
            size_t j = i;
            while (j < b.close && !s.is(j, ";")) ++j;  // end of the store statement (its call has no `;` inside)
            bool stored = false;
            int depth = 0;
            for (size_t k = j + 1; k < b.close && depth >= 0 && !stored; ++k) {
                if (s.is(k, "{")) ++depth;
                else if (s.is(k, "}")) --depth;
                else if (depth == 0 && s.ident(k, R) && s.is(k + 1, ".") && s.ident(k + 2, "material") && s.is(k + 3, "=") && (s.is(k - 1, ";") || s.is(k - 1, "{") || s.is(k - 1, "}")))
                    stored = true;
                else if (depth == 0 && s.ident(k, "if") && s.spells(k + 1, {"("}) && s.ident(k + 2, R) &&
                         s.spells(k + 3, {".", "scene", ".", "material", "==", "CUSTOM_MATERIAL", ")", "{"})) {
                    const size_t open = k + 10, close = s.close_of(open);
                    int d2 = 0;
                    for (size_t m = open + 1; m < close; ++m) {
                        if (s.is(m, "{")) ++d2;
                        else if (s.is(m, "}")) --d2;
                        else if (d2 == 0 && s.ident(m, R) && s.is(m + 1, ".") && s.ident(m + 2, "material") && s.is(m + 3, "=")) stored = true;
                    }
                }
            }
            if (!stored) return glsl_body;
        }
    }

    // (6) the rewrite: `nearer(R.scene.hit, H)` -> `nearer(R.scene.hit, H) && !(H.t > ptl_far)` (inserted in front of the condition's `)`)
    std::vector<std::pair<size_t, std::string>> inserts;  // raw token index -> text in front of it
    for (const Block& b : blocks) inserts.push_back({s.at[b.cond_close], " && !(" + b.H + ".t > ptl_far)"});
    std::string out;
    size_t next = 0;
    for (size_t k = 0; k < toks.size(); ++k) {
        while (next < inserts.size() && inserts[next].first == k) out += inserts[next++].second;
        out += toks[k].text;
    }
    if (bounded) *bounded = (int)blocks.size();
    return out;
}

}  // namespace ptl
