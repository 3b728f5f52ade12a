// ron.h -- reader for the subset of RON (Rusty Object Notation) the portal scene corpus uses.
//
// Replaces the third-party `ron 0.10.1` crate at the reference call site src/main.rs:2882
// (`ron::from_str::<SerializedScene>`).  Only the read side is needed by the hot path;
// scenes are written by the reference GUI with `pretty_config().escape_strings(false)`
// (src/gui/scene_serialized.rs:22-24), so strings are multi-line and mostly unescaped.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ptl::ron {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

// One node of the parsed document.
struct Value {
    enum Kind { Int, Float, Bool, String, List, Map, Struct, Tuple, Unit } kind = Unit;
    // Int / Float / Bool
    long long i = 0;
    double f = 0.0;
    bool b = false;
    // String: text.  Struct / Tuple / Unit: optional type or variant name.
    std::string s;
    // List / Tuple items
    std::vector<Value> items;
    // Struct fields in declaration order; Map entries (key, value) in declaration order
    std::vector<std::pair<std::string, Value>> fields;
    std::vector<std::pair<Value, Value>> entries;

    bool is_number() const { return kind == Int || kind == Float; }
    double number() const { return kind == Int ? (double)i : f; }
    bool is_named(const char* n) const { return (kind == Struct || kind == Tuple || kind == Unit) && s == n; }
    bool is_none() const { return kind == Unit && s == "None"; }
    // Option<T>: Some(x) -> &x, None -> nullptr, anything else -> itself
    const Value* some() const {
        if (is_none()) return nullptr;
        if (kind == Tuple && s == "Some" && items.size() == 1) return &items[0];
        return this;
    }
    const Value* find(const std::string& key) const {
        for (auto& kv : fields)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Value& at(const std::string& key) const {
        const Value* v = find(key);
        if (!v) throw std::runtime_error("RON: missing field `" + key + "`");
        return *v;
    }
    // newtype chains like (("text")) -> the innermost value
    const Value& unwrap_newtypes() const {
        const Value* v = this;
        while (v->kind == Tuple && v->items.size() == 1) v = &v->items[0];
        return *v;
    }
};

struct ParseError : std::runtime_error {
    int line;
    ParseError(int line_, const std::string& msg)
        : std::runtime_error("RON parse error at line " + std::to_string(line_) + ": " + msg), line(line_) {}
};

Value parse(const std::string& text);

// The writer: the layout the reference's scene files have (ron 0.10 `PrettyConfig::default().escape_strings(false)`,
// src/gui/scene_serialized.rs:22-24,654-666): structs, lists and maps one member per line with a trailing comma and four
// spaces per level, tuples and newtypes inline, floats in positional shortest round-trip form with at least one decimal,
// strings verbatim (raw `r#".."#` with as many hashes as needed when they contain a quote or a backslash).
std::string to_string(const Value& v);

}  // namespace ptl::ron
