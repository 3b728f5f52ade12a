// formula.h -- evaluator for the scalar formulas stored in scene uniforms.
//
// Replaces the third-party `fasteval 0.2.4` crate (Cargo.lock:636) at the reference call
// sites src/gui/uniform.rs:602-634 (parse + compile) and :1133-1136 (eval with the custom
// callback of :1009-1124).  fasteval is not vendored in the reference tree, so this is a
// restatement of its documented grammar and of its compile step's evaluation order:
//   precedence, highest first:  ^  %  /  *  -  +  (== != < <= >= >)  &&  ||
//   a - b  is compiled as  a + (-b);   a / b  as  a * (1/b);
//   in a chain of + (or *) the non-constant terms are combined left to right and the
//   folded constant is applied last.
// All arithmetic is IEEE binary64, like the reference.
#pragma once
#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <vector>

namespace ptl {

// Resolver for everything that is not fasteval syntax or a fasteval builtin: the reference's
// custom functions (`if`, `deg2rad`, `lerp`, easings ...) and free variables (other uniforms).
// Returns nullopt when the name cannot be resolved -> the whole formula evaluates to nullopt.
using FormulaNamespace = std::function<std::optional<double>(const std::string& name, const std::vector<double>& args)>;

class Formula {
public:
    // Parses and compiles; returns nullptr (and fills *error) on a syntax error.
    static std::shared_ptr<Formula> compile(const std::string& text, std::string* error = nullptr);
    std::optional<double> eval(const FormulaNamespace& ns) const;
    ~Formula();

    struct Node;

private:
    Formula() = default;
    std::unique_ptr<Node> root_;
};

// The reference's custom function table (src/gui/uniform.rs:1014-1124) minus `time`,
// `total_time` and free variables, which the caller supplies.  Returns nullopt if `name`
// is not one of them or an argument is missing.
std::optional<double> formula_custom_function(const std::string& name, const std::vector<double>& args, bool* known);

}  // namespace ptl
