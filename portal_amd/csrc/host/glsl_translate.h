// glsl_translate.h -- token-level rewrite of the GLSL ES 3.00 snippets stored in scene files
// into C++ that compiles against device/ptl_glsl.h (for hiprtc and for the host build).
//
// The reference hands these snippets verbatim to the GL driver's GLSL compiler
// (src/gui/scene.rs:776,865,877,1020,1041).  A HIP kernel has no GLSL front end, so the
// differences between the two languages are bridged here, lexically and line-preserving
// (the output has exactly as many lines as the input, so compiler diagnostics still point
// at the right snippet line -- the job of src/code_generation.rs's LineNumbersByKey):
//   * floating literals get an `f` suffix (GLSL literals are binary32, C++'s are binary64)
//   * multi-component swizzles  v.xyz -> v.sw<0,1,2>(),  v.xy = e -> v.swr<0,1>() = e
//   * `in` / `out` / `inout` / precision qualifiers in parameter lists
//   * identifiers that are C++ keywords are renamed
//   * the tagged-line filter of src/gui/scene.rs:1065-1107 (!FOR_NUMBER! etc.)
#pragma once
#include <string>
#include <vector>

namespace ptl {

// Feature switches of the reference's `Data` (src/gui/common.rs:63-86) with the native
// (non-wasm) defaults of src/main.rs:935-941.
struct CodegenFlags {
    bool for_prefer_variable = true;
    bool disable_antialiasing = false;
    bool disable_anaglyph = true;
    bool disable_camera_teleportation = false;
    bool use_300_version = true;
    bool defer_loop_updates = true;  // translate_glsl's deferred loop-carried ray transforms (KernelOptions can switch it off)
};

// Drops (blanks) the lines whose tags are switched off; keeps the line count.
std::string filter_tagged_lines(const std::string& text, const CodegenFlags& flags);

// GLSL snippet -> C++ (line-preserving).
//
// `defer_loop_updates` (default on): one optimisation on the way, exact by construction.  Scene snippets advance rays through a
// pair of portal matrices on EVERY iteration of a loop -- `X = transform(A_mat, transform(B_mat_inv, X));` -- but read X only inside
// nested blocks that few rays enter (where a portal hit is recorded: scenes/portal_in_portal.ron:1146-1147,1157,1178).  Such an
// update is replaced by a counter increment, and the pending updates are applied right before every statement that reads X (and
// behind the loop, if X is read there): the same operations on the same values in the same order for every ray that reads X,
// none for the rays that never do.  Applied only when all of this holds -- otherwise the code is left exactly as written:
//   * the update is a statement of a `for` body (not nested deeper, loop not inside another loop), of the form above: `transform`
//     calls and `*` over X (once) and identifiers that name uniforms (`*_mat`, `*_mat_inv`, `*_mat_teleport`, `*_u`, `_*`) which the
//     snippet never assigns or declares;
//   * every other occurrence of X in the loop body sits in a nested block, is not an assignment target, and X does not appear in
//     the loop header.
std::string translate_glsl(const std::string& glsl, bool defer_loop_updates = true);

// Refuses (std::runtime_error) a scene in which passing `out` / `inout` arguments by reference could differ from GLSL's copy in /
// copy out: a mutable global handed to a function that also names it, or one variable handed to two out parameters of a call.
// `sources`: file-scope GLSL (the scene's library); `bodies`: statement lists that call into it (material / object / intersection snippets).
void check_out_argument_aliasing(const std::vector<std::string>& sources, const std::vector<std::string>& bodies);

}  // namespace ptl
