// codegen.h -- scene -> HIP C++ kernel source.
//
// Host-side mirror of the reference's L3 "scene -> kernel codegen":
//   StringStorage / LineNumbersByKey / apply_template   src/code_generation.rs:10-98
//   Scene::uniforms, generate_uniforms_declarations      src/gui/scene.rs:424-543, 661-691
//   Scene::generate_shader_code (slot generators)        src/gui/scene.rs:693-1110
// The reference emits GLSL for the GL driver; this emits one self-contained HIP C++
// translation unit (device/ptl_glsl.h + ptl_library.h + filled ptl_trace.tpl + ptl_entry.h)
// for hiprtc, which the host build (oracle/host_build) compiles unchanged with g++.
#pragma once
#include <map>
#include <set>
#include <string>
#include <vector>

#include "glsl_translate.h"
#include "scene.h"

namespace ptl {

// Which scene element a span of generated lines came from (reference: (TypeId, UniqueId)).
struct ElementKey {
    std::string kind;  // "object", "material", "intersection_material", "library"
    std::string name;
    bool operator<(const ElementKey& o) const { return kind != o.kind ? kind < o.kind : name < o.name; }
    bool operator==(const ElementKey& o) const { return kind == o.kind && name == o.name; }
};
struct LineRange {
    int start = 0, end = 0;  // [start, end), 1-based like the reference
};

class LineNumbersByKey {
public:
    std::map<ElementKey, LineRange> ranges;
    void offset(int lines);
    void add(const ElementKey& key, LineRange r);
    void extend(const LineNumbersByKey& other);
    // element containing `line_no` and the line number local to that element (1-based)
    bool get_identifier(int line_no, ElementKey* key, int* local_line) const;
};

class StringStorage {
public:
    std::string storage;
    LineNumbersByKey line_numbers;
    int current_line_no = 1;
    void add_string(const std::string& s);
    void add_identifier_string(const ElementKey& id, const std::string& s);
    void add_string_storage(StringStorage other);
};

// Splits `tmpl` on "//%"; odd pieces are slot names looked up in `storages`.
StringStorage apply_template(const std::string& tmpl, std::map<std::string, StringStorage> storages);

// --- uniforms ------------------------------------------------------------------------------
enum class UniformType { Mat4 = 0, Float1 = 1, Int1 = 2, Float2 = 3, Float3 = 4, Sampler = 5 };
struct UniformDesc {
    std::string name;
    UniformType type;
    size_t offset = 0;  // byte offset inside the kernel's uniform block
};
size_t uniform_type_size(UniformType t);

// One evaluated uniform value, as Scene::set_uniforms (scene.rs:545-657) would upload it.
struct UniformUpload {
    std::string name;
    UniformType type;
    float f[16] = {0};
    int i = 0;
    bool animated = false;  // its evaluation read time / total_time / the camera matrix: changes from frame to frame
    bool same_value(const UniformUpload& o) const;
};

// What a kernel may assume about a matrix uniform that stays a run-time value (KernelOptions::mask_zero_elements), element k = 4 * column + row:
//   bit k        the element may be non-zero (clear: it is zero -- its term is skipped, device/ptl_glsl.h `ptl_row_m`);
//   bit 16 + k   it is exactly +1;   bit 32 + k   it is exactly -1   (the term is `x + acc` / `acc - x`: the same operation, the value known).
// 0xffff = nothing known.  Patterns of several states of a matrix (probes over a clip) combine: may-be-non-zero bits by OR, unit bits by AND.
typedef unsigned long long MatrixPattern;
MatrixPattern matrix_pattern(const float elements[16]);
inline MatrixPattern combine_patterns(MatrixPattern a, MatrixPattern b) { return ((a | b) & 0xffffull) | (a & b & ~0xffffull); }
// does a matrix with these elements have (at least) what `assumed` says?
inline bool pattern_holds(MatrixPattern assumed, const float elements[16]) {
    const MatrixPattern now = matrix_pattern(elements);
    return ((now & 0xffffull) & ~assumed) == 0 && ((assumed & ~0xffffull) & ~now) == 0;
}

// What the zero-pattern probing of generate_kernel_source found last time, and for which scene state (KernelOptions::mask_cache): a renderer
// with baked Bool / Int uniforms regenerates its source on every scene-version bump -- every camera move -- and the probing (two scene
// copies, up to 33 Scene::update + 34 evaluations) is by far the most expensive part of a generation that nearly always ends in "unchanged".
struct ZeroMaskCache {
    std::string key;  // stage, clip, and every uniform's name with its value (animated ones: the name only -- their patterns come from the probes)
    std::vector<std::pair<std::string, MatrixPattern>> masked;
    int hits = 0, misses = 0;
};

struct KernelOptions {
    bool specialize_ints = false;  // bake current Bool/Int uniform values in as literals (recompile when they change)
    bool specialize_all = false;   // also bake Float / matrix scene uniforms (not the camera / builtins)
    bool count_segments = false;   // compile with PTL_COUNT_SEGMENTS
    bool anaglyph = false;         // compile the !ANAGLYPH! code in (the reference's `disable_anaglyph = false`)
    // The scene's own SWITCHES: Bool / Int uniforms whose evaluation reads no per-frame input (time, the camera) are GUI toggles and
    // counters -- `filter_teleported`, `shape`, `show_teleported` -- that stay put while Float / Angle / matrix values animate.  The patterns build
    // (PTL_FLAG_SPECIALIZE_PATTERNS) compiles them in like the renderer's mode switches (round 5: the headline's patterns kernel 0.50 -> 0.30 ms --
    // dead branches, constant loop bounds, unrolled snippet loops); the renderer checks them before every draw like a clip-constant value,
    // demotes the one that moved and rebuilds.
    bool specialize_static_ints = false;
    bool specialize_static = false;  // bake every scene uniform whose evaluation does not read a per-frame input ...
    std::set<std::string> keep_dynamic;  // ... except these (values that changed after all: demoted by the renderer)
    // The renderer's own mode switches (`_use_panini_projection`, `_use_360_camera`, `_use_180_camera`, `_draw_depth_map`, `_draw_anaglyph`,
    // `_draw_side_by_side`) as literals, name -> value: with any specialisation on, the camera models and output modes a frame does not use
    // are not compiled in at all -- 7-10 % of the kernel time of the BASELINE scenes (profiles/r03/stub_bake_switches.jsonl), although they
    // are read once per ray: what they cost is registers and code around the bounce loop.  The renderer rebuilds when one changes.
    std::map<std::string, int> baked_options;
    // A matrix uniform that stays a run-time value (nothing baked, only Bool / Int baked, animated within the clip, demoted) still has a ZERO
    // PATTERN -- portal matrices are mostly translations and quarter turns -- and the pattern survives where the values move.  With this on
    // the generator records the pattern of every such matrix (GeneratedKernel::masked, `#define PTL_MASK_<name>`) and writes the
    // `transform(<name>, ..)` calls of the snippets and the generated plane tests in their masked forms (device/ptl_glsl.h `ptl_mul_m`):
    // the terms with a zero element are neither executed nor loaded -- what a baked matrix gets from ptl_mterm.  Same deviation as there
    // (non-finite vector components); the renderer rebuilds when a masked element stops being zero.  Off for contract 1 and the tolerance mode.
    bool mask_zero_elements = false;
    std::set<std::string> keep_unmasked;  // ... except these (a pattern that did not hold: demoted by the renderer)
    ZeroMaskCache* mask_cache = nullptr;  // optional: reuse the probed patterns while the scene state they depend on is the same
    // Shortened products (ptl_mterm / masks) equal the full chains for every FINITE vector.  A scene with a matrix that holds infinities, or
    // NaN beside numbers (matrix_breaks_short_chains), sends +-inf components down its rays, so the generator keeps every full chain for it
    // (GeneratedKernel::full_chains: no PTL_DROP_ZERO_TERMS, no masks); the renderer sets this when a run-time matrix turns so later.
    bool full_chains = false;
    // Ray-independent work of the generated plane code (normalize(get_normal(X_mat)), both possible is_collinear verdicts) is
    // evaluated once per uniform upload by the module's prologue kernel `ptl_derive_kernel` and read back as extra uniforms,
    // instead of once per bounce-loop trip by every lane.  Same functions, same binary32 operations: identical frames.
    // Applies to matrices that are run-time uniforms (a baked matrix folds at JIT time anyway).
    bool derived_uniforms = true;
    // ... and the uniform-only work of the scene snippets with them (glsl_hoist.h); no effect without derived_uniforms
    bool hoist_uniform_work = true;
    // first-trip variants of the intersection-material snippets (ptl_trace.tpl PTL_FIRST_TRIP): the origin half of their ray arithmetic
    // comes from the prologue while the ray still starts at the camera.  Needs derived_uniforms and hoist_uniform_work.
    bool first_trip = true;
    // snippet loops whose bound is a baked Int uniform (<= 16) are unrolled (codegen.cpp SnippetTranslator::unrolled): identical frames
    bool unroll_baked_loops = true;
    bool exact_cr = false;   // PTL_CONTRACT_V1: the rounds-1-2 numerics contract (IEEE correctly rounded / and sqrt for every input) instead of contract 2 (device/ptl_glsl.h)
    // first-trip form of the GENERATED plane tests (scene.rs:912-948): a second copy of scene_intersect for the trip on which every ray of the
    // wave still starts at the camera takes `plane_inv * r.o` of every Flat object from the prologue kernel (a vec4 per plane behind the
    // derived uniforms) -- the same product of the same values, computed once per upload instead of per lane.  Needs derived_uniforms.
    bool first_trip_planes = true;
    // The caller's distance bound inside the intersection-material snippets (glsl_translate.h `bound_nearer_blocks`): the bounce loop evaluates
    // scene_intersect() first and hands its hit distance to the snippets, whose `if (nearer(result.scene.hit, H))` blocks then skip candidates
    // that could never be the nearest hit.  Exact by construction (identical frames); applied per snippet where its shape allows.
    // OFF by default: measured on the headline scene from five views (profiles/r04/ab_bounded_snippets.jsonl) it gains nothing -- the blocks it
    // skips are entered by few lanes anyway, and keeping scene_intersect's hit alive across the snippet costs what the skipped work saves
    // (-2 ... +13 % kernel time).  Kept as an option for scenes whose snippets are dominated by their inside tests.
    bool bound_snippets = false;
    // The render entry reads its uniform block from a device BUFFER of blocks instead of the module's one global, indexed by blockIdx.z
    // (`ptl_render_slices_kernel`): ONE launch then traces several frames that differ in their uniforms -- the motion-blur sub-frames of a clip
    // frame (src/main.rs:1798 re-draws with `_aa_start` windows one after the other) -- so that their ramps and tails overlap.  Applied to the
    // generated text by substitution (codegen.cpp `apply_slices_entry`): kernels built without it are byte for byte what they were.
    bool slices_entry = false;
    // Affine rays (round 5): in a kernel whose every scene matrix is KNOWN to have the bottom row 0 0 0 1 (baked, or through its pattern) and
    // whose scene snippets never write a ray's w, every origin has w = 1 and every direction w = 0, and the products of a matrix with a ray
    // say so (device/ptl_glsl.h PTL_AFFINE_RAYS): the translation column costs a direction nothing, the w row folds to a constant.  Exact for
    // finite rays -- the guard and the stated deviation of the shortened products (full_chains).  The renderer switches it off, and rebuilds,
    // when a CAMERA matrix (a run-time value in every build) is not affine (GeneratedKernel::affine_rays says whether the kernel has it).
    bool affine_rays = true;
    // Round 6, the dynamic belt behind the scan of the snippets: the same build WITHOUT affine rays (the general products) whose assumption sites --
    // the matrix-times-ray products, the bounce loop -- count the ray halves that arrive with another w than 1 / 0 into the `segments` counter
    // (device/ptl_glsl.h PTL_CHECK_AFFINE).  Its frame is right either way; a non-zero count says an affine-rays kernel's would not be.
    bool check_affine = false;
    // Round 6: the Simple materials' literals in a per-workgroup LDS table, one material_simple2 call for all of them (codegen.cpp, materials).
    // Identical frames; PTL_FLAG_NO_MATERIAL_TABLE (bit 26) keeps the reference's chain of one inlined call per material (A/B measurements, tests).
    // MEASURED (profiles/r06/ab_material_table.jsonl): the LDS table is SLOWER than the reference's chain -- headline 0.1925 against 0.1873 ms, C2 0.0378
    // against 0.0361 -- a workgroup cannot trace before its table is staged (a global load and a barrier at the head of a 3 us workgroup), and one body
    // with run-time grid flags costs a wave of ONE material (the usual case) more than its own copy with the literals folded.  So: off by default;
    // 1 = the LDS table (PTL_FLAG_MATERIAL_TABLE_LDS, bit 26), 2 = the same table in constant memory, read with SCALAR loads per distinct material
    // of the wave (a waterfall loop; PTL_FLAG_MATERIAL_TABLE_SCALAR, bit 27).
    int material_table = 0;
    // A/B switch (PTL_FLAG_KEEP_TRANSFORM_DODGES): a kernel with affine rays still gets the deferred loop updates and the first-trip snippet copies
    // -- round 4's shape, for measurements; by default it gets neither (codegen.cpp: a transform is then a few additions, cheaper than its dodge)
    bool keep_transform_dodges = false;
    bool quick_jit = false;  // PTL_QUICK_JIT: compile at -O1 instead of the shipped -O3 (half the JIT time, a 5-20 % slower kernel)
    bool fast_math = false;  // PTL_FAST_MATH: hardware rcp / sqrt / rsq (1 ulp), a/b = a * rcp(b), FMA contraction: tolerance mode, not bit-exact
};

// One plane test of a Flat object whose ray-independent part is evaluated by the prologue kernel (KernelOptions::derived_uniforms).
struct DerivedPlane {
    int object = 0, side = 0;   // scene object index; 0 = the object's (first) matrix, 1 = a portal's second matrix
    std::string member;         // ptl_uniform_block members `<member>_nrm` (vec3) and `<member>_col` (int, two verdict bits)
    std::string normal_expr;    // the generated code's `normal` (what is_collinear compares hit.n with)
    std::string arg_expr;       // the vector plane_intersect normalises
};

struct GeneratedKernel {
    std::string source;                 // complete translation unit
    LineNumbersByKey line_numbers;      // lines that came from scene snippets
    std::vector<UniformDesc> uniforms;  // block layout: samplers first, then Scene::uniforms() order
    size_t uniform_block_size = 0;
    std::vector<std::string> defines;   // e.g. "PTL_COUNT_SEGMENTS"
    std::vector<UniformUpload> baked;   // the values compiled in as literals (specialised builds)
    int first_trip_plane_tests = 0;     // generated plane tests that have a first-trip form (ptl_dvo_<object>_<side> members)
    bool first_trip_variants = false;   // the kernel has first-trip copies of its intersection-material snippets (define PTL_FIRST_TRIP)
    bool looped_snippets = false;       // an intersection-material snippet has a force-unrolled loop (define PTL_JIT_MODULE_INLINER: kernel.cpp compiles it with the module inliner)
    int hoisted_members = 0;            // ... plus this many members holding uniform-only work of the scene snippets (glsl_hoist.h)
    std::vector<DerivedPlane> derived;  // members appended to the block behind uniform_block_size, written on the device
    std::vector<std::pair<std::string, MatrixPattern>> masked;  // run-time matrices whose pattern is compiled in (MatrixPattern)
    int bounded_snippet_blocks = 0;     // `nearer` blocks of intersection-material snippets that take the caller's distance bound (define PTL_BOUNDED_SNIPPETS)
    bool full_chains = false;           // a matrix of the scene is not finite (or KernelOptions::full_chains): no product was shortened
    std::string affine_rays_refused_because;  // what `snippets_keep_rays_affine` said when it switched the assumption off ("" otherwise)
    bool affine_rays = false;           // generated with PTL_AFFINE_RAYS: valid while the camera matrices have the bottom row 0 0 0 1
};

// Scene::uniforms (scene.rs:424-543): names and types, in the reference's order.
std::vector<UniformDesc> scene_uniform_list(const Scene& scene);
// Sampler names "{name}_tex" for textures (scene.rs:394-412), sorted and de-duplicated.
std::vector<std::string> scene_texture_list(const Scene& scene);

GeneratedKernel generate_kernel_source(const Scene& scene, const CodegenFlags& flags, const KernelOptions& opts);
// A matrix with infinite elements, or with NaN beside numbers: its products carry +-inf components, for which a product that skips zero
// terms differs from the full chain (KernelOptions::full_chains).  An all-NaN matrix does not (all-NaN vectors stay NaN either way).
bool matrix_breaks_short_chains(const float m[16]);
// Bottom row exactly 0 0 0 1 (column-major elements 3, 7, 11, 15): an affine map -- it keeps the w of a point at 1 and of a direction at 0.
bool matrix_is_affine(const float m[16]);
inline bool pattern_is_affine(MatrixPattern p) { return (p & ((1ull << 3) | (1ull << 7) | (1ull << 11))) == 0 && ((p >> (16 + 15)) & 1ull) != 0; }
// Do the scene's GLSL snippets keep rays affine?  True unless one of them builds a Ray from parts that are not spelled `vec4(.., 1.)` /
// `vec4(.., 0.)`, assigns a ray's `.o` / `.d` in another than a whitelisted form, calls transform() with a matrix that is not a scene uniform
// (`X_mat`, `X_mat_inv`, `A_to_B_mat_teleport`: the ones whose values are checked), or has an out / inout parameter of type Ray or vec4
// (KernelOptions::affine_rays; conservative: a refusal only costs the optimisation).  `why` receives the offending text.
bool snippets_keep_rays_affine(const std::vector<std::string>& codes, std::string* why);

// All scene-derived uniforms: X_mat, X_mat_inv, A_to_B_mat_teleport, user uniforms.
// `errors` receives the reference's "matrix `x` can't be getted" style messages.
std::vector<UniformUpload> evaluate_scene_uniforms(const Scene& scene, std::vector<std::string>* errors);

// Rust `{:e}` formatting of an f64 (shortest round-trip digits, exponent without padding).
std::string format_lower_exp(double v);

// The fixed device sources (embedded at build time from csrc/device/).
const char* device_source_glsl();
const char* device_source_library();
const char* device_source_trace_template();
const char* device_source_entry();

}  // namespace ptl
