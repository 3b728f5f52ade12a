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
#include <set>
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
// `force_inline_definitions`: the text is a file-scope library (scene.rs:1037-1044) -- its function definitions get PTL_FN like every other function
// of the kernel.  Without it they are plain inline members that LLVM's bottom-up inliner happens to inline and its module inliner does not
// (a call with `this` then keeps the tracer object in scratch).
std::string translate_glsl(const std::string& glsl, bool defer_loop_updates = true, bool force_inline_definitions = false);

// Refuses (std::runtime_error) a scene in which passing `out` / `inout` arguments by reference could differ from GLSL's copy in /
// copy out: a mutable global handed to a function that also names it, or one variable handed to two out parameters of a call.
// `sources`: file-scope GLSL (the scene's library); `bodies`: statement lists that call into it (material / object / intersection snippets).
void check_out_argument_aliasing(const std::vector<std::string>& sources, const std::vector<std::string>& bodies);

// The caller's distance bound inside an intersection-material snippet (KernelOptions::bound_snippets).  The bounce loop takes the nearest
// of scene_intersect() and the scene's snippets (src/frag.glsl:114-125); a snippet candidate farther than the hit scene_intersect() has
// already found can never be that nearest hit, yet the snippets of the reference's scenes spend most of their time deciding whether such
// candidates lie inside a portal (scenes/portal_in_portal.ron:1133-1191: ten nested copies of two portals, each with an O(copy) walk).
// For a snippet of the shape
//     SceneIntersectionWithMaterial R = SceneIntersectionWithMaterial(scene_intersection_none, material_empty());
//     ... if (nearer(R.scene.hit, H)) { ... R.scene = process_portal_intersection(R.scene, H, ...); ... R.material = ...; } ...
//     return R;
// every such condition becomes `nearer(R.scene.hit, H) && !(H.t > ptl_far)` (`ptl_far`: a new parameter of the snippet function).
// Exact by construction: with S_k the accumulator of the snippet as written after block k and T_k the bounded one's, T_k = S_k whenever
// S_k.t <= ptl_far and "no hit" otherwise (induction over the blocks: a candidate within the bound meets the same `nearer` verdict in
// both, one beyond it is accepted as written only when the accumulator holds no nearer hit -- and a hit beyond the bound loses against
// scene_intersect()'s in the caller either way).  Applied only when the token stream shows that nothing else can carry a skipped block's
// effect out of it: R appears only in the declaration, the conditions, `R.scene = process_{portal,plane}_intersection(R.scene, H, ..)`,
// `R.scene.material ==/!=`, `R.material =` and the final `return R;`; a block assigns only R and its own locals, calls no function with
// out / inout parameters, and stores R.material itself whenever it may leave CUSTOM_MATERIAL behind (the one case in which the caller reads
// it).  The caller additionally requires that no GLSL of the scene names TELEPORT_SUBSPACE / in_subspace: process_portal_intersection never
// resets SceneIntersection::in_subspace, so with subspace portals a skipped candidate could leave that flag behind.  Returns the text
// unchanged (and *bounded = 0) when any of this fails.
std::string bound_nearer_blocks(const std::string& glsl_body, const std::set<std::string>& functions_with_out_params, int* bounded);

}  // namespace ptl
