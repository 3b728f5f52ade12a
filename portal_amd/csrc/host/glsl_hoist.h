// glsl_hoist.h -- binding-time analysis of scene snippets: work that depends on nothing but run-time uniforms is moved out of the
// tracer into the module's prologue kernel (ptl_derive_kernel, device/ptl_entry.h), which runs once per uniform upload.
//
// A GLSL driver compiles the reference's shader with the uniforms as run-time values too, and evaluates every expression of a
// snippet for every ray -- including `b0_mat * (a_mat_inv * normal_b)`, a product of two scene matrices and a vector that no ray
// ever changes (scenes/portal_in_portal.ron:1181).  gfx950 has no scalar floating-point unit: such an expression costs the same
// VALU issue slots as ray arithmetic, in every wave, on every trip of the bounce loop.  Here the snippet is parsed (expressions
// only; anything the parser does not understand leaves the function untouched), every expression is classified bottom-up as
// UNIFORM (literals, run-time uniforms, locals initialised from such and never written again, pure GLSL built-ins of those) or
// VARYING, and each maximal uniform expression that does arithmetic becomes a member of the uniform block behind the uploaded
// part: the snippet reads it (a scalar load), the prologue kernel computes it -- from the SAME expression text, compiled in the
// same module with the same numerics contract, so the value has the same bits as the one every ray used to compute.
//
// Loop-carried uniform chains are tabulated: `V = init; for (int i = 0; i < N; i++) { ... V = f(V, uniforms); }` makes V a
// function of i alone, so V and the uniform expressions of V become arrays indexed by i (kTableEntries long; a wave-uniform guard
// `N <= kTableLoop` keeps the original code for longer loops).
//
// Three library functions mix a uniform argument with a varying one and spend most of their time on the uniform one; their calls
// are re-targeted to staged forms (device/ptl_library.h) that take the precomputed part:
//   normalize_normal(U, d)       -> ptl_normalize_normal_unit(normalize(U), d)
//   plane_intersect(r, M, U)     -> ptl_plane_intersect_unit(r, M, normalize(U))
//   is_collinear(a, U) / (U, b)  -> ptl_is_collinear_len(a, U, length(U)) / (U, b, length(U))
#pragma once
#include <map>
#include <set>
#include <string>
#include <vector>

namespace ptl {

constexpr int kTableLoop = 64;                  // longest loop whose uniform chains are tabulated
constexpr int kTableEntries = kTableLoop + 2;   // values before iteration 0 .. after iteration kTableLoop

struct HoistedMember {
    std::string type;  // GLSL type: float, vec2..4, mat2..4
    std::string name;  // member of ptl_uniform_block
    int length = 0;    // 0: one value; otherwise an array of that many
};

struct HoistParams {
    std::map<std::string, std::string> uniforms;      // uniforms READ AT RUN TIME: name -> GLSL type (baked ones must not be listed)
    std::map<std::string, std::string> constants;     // uniforms BAKED into the source as literals: uniform values the compiler folds by itself
                                                      // (an expression of these alone is never hoisted; one that also reads `uniforms` is)
    std::set<std::string> functions_with_out_params;  // functions of the scene that may write through an argument
    std::set<std::string> scene_functions;            // every function the scene defines itself: never taken for a built-in of that name
    // First-trip variant of a snippet (KernelOptions::first_trip): these parameters are rays whose ORIGIN is the uniform value
    // `origin_expr` (every primary ray starts at the camera).  A `Ray` expression built from them with `transform(uniform matrix, ray)`
    // keeps a uniform origin: that half -- and a loop-carried chain of it -- comes from the prologue (the direction half stays per ray),
    // the expression is wrapped as ptl_ray_o(<expression>, <member>) and the compiler drops the now dead origin arithmetic.
    std::vector<std::string> origin_uniform_rays;
    std::string origin_expr;                          // in the prologue's terms, e.g. "PTL_DV_OUT.ptl_dv_origin"
    bool body_only = false;                           // the text is the body of a function the code generator wraps, not definitions
    std::vector<std::string> body_params;             // that function's parameter names
};

struct HoistResult {
    std::string glsl;      // the snippet, line for line, with the hoisted work replaced by reads of the members
    std::string prologue;  // GLSL statements (`out->NAME = ...;`, one block per function) for ptl_tracer::derive; empty: nothing hoisted
    std::vector<HoistedMember> members;
};

// `next_member` numbers the members across the snippets of one kernel.  Never throws: on anything unexpected the text comes back unchanged.
HoistResult hoist_uniform_work(const std::string& glsl, const HoistParams& params, int& next_member);

}  // namespace ptl
