/* portal_amd.h -- C ABI of libportal_amd.so, the MI355X backend for optozorax/portal scenes.
 *
 * Two layers, both plain C (pointers and sizes only):
 *
 *  1. ptl_kernel_*  -- the drop-in boundary.  Replaces exactly what the reference asks of
 *     macroquad/miniquad for its trace shader:
 *         load_material(ShaderSource::Glsl{..}, MaterialParams{uniforms, textures})
 *                                                   src/gui/scene.rs:1132-1143  -> ptl_kernel_compile
 *         material.set_uniform(name, value)         src/gui/scene.rs:587-588,624-632,641-649,
 *                                                   src/main.rs:1269-1358      -> ptl_kernel_set_uniform
 *         material.set_texture(name, texture)       src/main.rs:1077-1078      -> ptl_kernel_set_texture
 *         render_target(w,h) + gl_use_material + draw_rectangle(0,0,w,h)
 *                                                   src/main.rs:1041-1042,1424-1425 -> ptl_kernel_render
 *         texture.get_texture_data()                src/main.rs:2939-2943      -> ptl_kernel_render_to_host
 *     A Rust host binds these with `extern "C"` (INTEGRATION.md shows the stub).
 *
 *  2. ptl_scene_* / ptl_renderer_*  -- the host side above that boundary, written in C++
 *     because no Rust toolchain exists in the build image: .ron loader, uniform/matrix
 *     evaluator, scene -> HIP source generator, and the SceneRenderer driver
 *     (src/main.rs:934-1064,1266-1359,1411-1428).  Python (portal_amd/__init__.py), the CLI and
 *     the tests call this layer through ctypes.
 *
 * Threading: a handle is not thread-safe; different handles may be used from different threads
 * (one per GPU).  Every function returns 0 on success unless stated otherwise.
 */
#ifndef PORTAL_AMD_H
#define PORTAL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---------------------------------------------------------------------- */
#define PTL_OK 0
#define PTL_UNKNOWN_UNIFORM 1 /* set_uniform: the kernel has no such uniform; ignored, like macroquad */
#define PTL_ERR_INVALID (-1)
#define PTL_ERR_COMPILE (-2) /* hiprtc diagnostics are in the log buffer */
#define PTL_ERR_HIP (-3)     /* HIP runtime call failed; see ptl_last_error() */
#define PTL_ERR_TYPE (-4)    /* set_uniform: type differs from the declared one */
#define PTL_ERR_SCENE (-5)   /* scene file / evaluation error */
#define PTL_ERR_NO_DEVICE (-6)

/* Last error message of the calling thread ("" if none). */
const char* ptl_last_error(void);

/* Library / toolchain description: "portal_amd <ver>; hip=<path>; hiprtc=<path>". */
const char* ptl_version(void);

/* Number of HIP devices visible (0 when the HIP runtime cannot be loaded). */
int ptl_device_count(void);

/* ---- layer 1: kernel handle ---------------------------------------------------------------- */
typedef struct ptl_kernel ptl_kernel;

typedef enum { PTL_MAT4 = 0, PTL_F32 = 1, PTL_I32 = 2, PTL_VEC2 = 3, PTL_VEC3 = 4, PTL_SAMPLER = 5 } ptl_type;

typedef struct {
    const char* name;
    ptl_type type;
    size_t offset; /* byte offset in the kernel's `ptl_uniform_block` (see ptl_scene_uniform_layout) */
} ptl_uniform_desc;

/* Compile `hip_source` (a translation unit produced by ptl_scene_generate_source, or written
 * by hand against the same conventions) for `device` with hiprtc and load it.
 *   uniforms/n_uniforms : every uniform of the block incl. samplers, with offsets
 *   defines/n_defines   : preprocessor symbols, e.g. "PTL_COUNT_SEGMENTS"
 *   log/log_cap         : receives the hiprtc log (NUL-terminated, truncated to fit)
 * device = -1 compiles only (no GPU needed): the handle can then only be queried with
 * ptl_kernel_code_object. */
int ptl_kernel_compile(int device, const char* hip_source, const ptl_uniform_desc* uniforms, int n_uniforms,
                       size_t uniform_block_size, const char* const* defines, int n_defines, ptl_kernel** out, char* log,
                       size_t log_cap);

/* A second instance of a compiled kernel on the same device: the code object loaded once more, so it has a uniform block of its own and
 * can be in flight on another stream with other uniform values (one module's block is a single global: consecutive launches of ONE handle
 * with different uniforms serialise on the upload).  Nothing is compiled.  ptl_kernel_copy_uniforms hands the original's current values
 * (sampler records included: the clone reads the original's texel buffers, so the original must outlive it) to the clone.  What
 * ptl_renderer_set_option("concurrent_draws", K) is built on. */
int ptl_kernel_clone(ptl_kernel* src, ptl_kernel** out);
int ptl_kernel_copy_uniforms(ptl_kernel* dst, const ptl_kernel* src);
/* The compiled gfx950 code object (valid until ptl_kernel_destroy). */
int ptl_kernel_code_object(ptl_kernel* k, const void** data, size_t* size);

/* Kernel metadata of a gfx950 code object (ptl_kernel_code_object, or a file of the code-object cache) without a device: the largest value of
 * `key` (".vgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size", ".sgpr_count") over the kernels whose name starts with
 * `kernel_prefix` (NULL or "": every kernel of the module); -1 when the key does not occur -- and, with a prefix, for a key that sorts before
 * ".name" (".kernarg_segment_size", ".group_segment_fixed_size" ...: the per-kernel attribution reads the metadata map in its alphabetical key
 * order and is only right behind ".name").  The JIT's occupancy retry decides on the render
 * entries alone ("ptl_render"): the one-wave teleport and prologue entries of a one-module build say nothing about the render kernel. */
int ptl_code_object_note(const void* code, size_t size, const char* key, const char* kernel_prefix);
/* Per-lane resources of the loaded render kernel (hipFuncGetAttribute): vector registers, scratch
 * (private segment) bytes -- non-zero means register spills that travel through the memory hierarchy --
 * and static LDS bytes per workgroup.  -1 where the runtime cannot tell. */
int ptl_kernel_resources(ptl_kernel* k, int* registers, int* scratch_bytes, int* lds_bytes);

/* value points at 16 floats (column-major) / 1 float / 1 int32 / 2 floats / 3 floats. */
int ptl_kernel_set_uniform(ptl_kernel* k, const char* name, ptl_type type, const void* value);

/* RGBA8 texels, row 0 first, are copied to device memory; sampled bilinear, clamp-to-edge. */
int ptl_kernel_set_texture(ptl_kernel* k, const char* sampler, const uint8_t* rgba8, int width, int height);

/* Row-block sharding of one frame: the frame is cut into blocks of 8 pixel rows; this launch
 * renders blocks phase, phase+stride, phase+2*stride, ...  (1 GPU: phase 0, stride 1). */
typedef struct {
    int width, height;       /* full frame, used for the pixel -> ray mapping */
    int rb_phase, rb_stride; /* row-block interleave */
    int in_place;            /* 0: the shard's rows are stored packed, in block order (out = shard_rows*width pixels);
                              * 1: the out pointers address a FULL frame (height*width pixels) and every row is stored at its
                              *    own position -- N launches with phase 0..N-1 fill one buffer between them, which may be
                              *    another GPU's memory mapped with ptl_ipc_open */
} ptl_frame;

/* Rows rendered by a shard (with in_place = 0 they are stored packed, in block order, in the output buffers). */
int ptl_frame_shard_rows(const ptl_frame* f);

/* Launch on `stream` (a hipStream_t, NULL = default stream).  Outputs are DEVICE pointers, either
 * may be NULL: out_rgba8 = shard_rows*width*4 bytes, out_rgba32f = shard_rows*width*16 bytes
 * (frame->in_place: height*width*4 and height*width*16 bytes).
 * segments (DEVICE pointer to one uint64, may be NULL) accumulates bounce-loop trips when the
 * kernel was compiled with PTL_COUNT_SEGMENTS.  If elapsed_ms is non-NULL the launch is bracketed
 * by HIP events on `stream` and the call waits for completion. */
int ptl_kernel_render(ptl_kernel* k, const ptl_frame* frame, void* out_rgba8, void* out_rgba32f, void* segments,
                      void* stream, float* elapsed_ms);

/* Convenience: render the shard and copy it to HOST buffers (either may be NULL). */
int ptl_kernel_render_to_host(ptl_kernel* k, const ptl_frame* frame, uint8_t* host_rgba8, float* host_rgba32f,
                              uint64_t* host_segments, float* elapsed_ms);
/* Slices: ONE launch for several frames that differ in their uniforms only (the motion-blur sub-frames of a clip frame, src/main.rs:1798).
 * A source generated with flag bit 22 (PTL_FLAG_SLICES) has the render entry `ptl_render_slices_kernel`, which reads the uniform block of
 * slice blockIdx.z from a device buffer of blocks instead of the module's one global -- through the same scalar loads -- and writes slice z's
 * frame behind slice z - 1's.  ptl_kernel_stage_slice(k, j) makes the values set so far (ptl_kernel_set_uniform ...) slice j of the next
 * batch; ptl_kernel_render_slices uploads slices 0 .. n-1 with one copy, runs the prologue of every slice and launches once with
 * grid.z = n: slice z lands at out_rgba8 + z * slice_pixels pixels (out_rgba32f + 4 * z * slice_pixels floats; either may be NULL).  The
 * ramp and tail of a small frame (~10 us of a 50 us 1080p launch) overlap with its neighbours' instead of adding up.  A single
 * ptl_kernel_render of such a module is a batch of one; the camera-teleport query keeps using the module's own block.
 * ptl_kernel_max_slices: 16 for such a module, 0 otherwise. */
int ptl_kernel_max_slices(ptl_kernel* k);
int ptl_kernel_stage_slice(ptl_kernel* k, int index);
/* ... or a block saved earlier (ptl_kernel_snapshot_uniforms: the host copy of the uniform block, ptl_kernel_uniform_block_size bytes), also one
 * saved from ANOTHER kernel of the same scene -- every build of a scene has the same block layout.  A sampler record that names a texel buffer
 * of `k` itself (bound now, or bound when the block was taken) is kept; any other record is replaced by `k`'s current one.
 * Textures and slices: a staged slice names the texel buffers that were bound when it was staged.  Re-binding a sampler
 * (ptl_kernel_set_texture: a video texture stepping to its next frame) between two stage calls does NOT free the buffer the earlier slice
 * reads: it is retired and freed behind the launch.  A caller that keeps snapshots instead of staging at once brackets them with
 * ptl_kernel_hold_textures(k, 1) ... (k, 0) for the same guarantee.
 * A block is only meaningful for a kernel whose compiled-in values agree with it: a value-baked build (flag bits 0 / 2 / 3 / 20) that had to
 * be rebuilt between two stage calls must not trace the earlier blocks -- layer 2 (ptl_renderer_stage_slice / _draw_slices) launches each
 * slice on the kernel it was staged with; a layer-1 caller that rebuilds has to do the same. */
int ptl_kernel_stage_slice_from(ptl_kernel* k, int index, const void* block, size_t size);
int ptl_kernel_hold_textures(ptl_kernel* k, int hold);
size_t ptl_kernel_uniform_block_size(ptl_kernel* k);
int ptl_kernel_snapshot_uniforms(ptl_kernel* k, void* dst, size_t cap);
int ptl_kernel_render_slices(ptl_kernel* k, const ptl_frame* frame, int n, void* out_rgba8, void* out_rgba32f, unsigned long long slice_pixels,
                             void* stream, float* elapsed_ms);

/* SceneRenderer::teleport_external_ray (src/main.rs:1361-1409): where does point b end up when the
 * segment a -> b is carried through the scene's portals (at most 10)?  One-thread launch of the
 * kernel's second entry point; synchronous.  teleported = 0 means "no portal crossed" (the
 * reference's `None`), out_pos is then (0,0,0).  Uses whatever uniforms are currently set. */
int ptl_kernel_teleport_ray(ptl_kernel* k, const float a[3], const float b[3], float out_pos[3], int* hit_object,
                            int* changed_subspace, int* teleported);
/* Split builds.  A source compiled with the define PTL_RENDER_MODULE has no teleport entry: that entry is a second copy of the whole
 * tracer, a fifth of every hiprtc build, and a camera that stands still never asks for it.  ptl_kernel_teleport_ray on such a kernel
 * compiles the other half on demand -- the same source with PTL_TELEPORT_MODULE: teleport entry + prologue, no render entry; its code
 * object is cached like any other -- and runs the query there with this kernel's uniform values.  ptl_kernel_prebuild_teleport does the
 * compile ahead of time (on a compile-only handle, device -1: fills the code-object cache).  Kernels compiled without either define
 * have every entry in one module, as before.  ptl_renderer_* builds split (environment PTL_ONE_MODULE=1: one module). */
int ptl_kernel_prebuild_teleport(ptl_kernel* k);

void ptl_kernel_destroy(ptl_kernel* k);

/* ---- layer 2: scene + renderer ------------------------------------------------------------- */
typedef struct ptl_scene ptl_scene;
typedef struct ptl_renderer ptl_renderer;

/* Scene::from_serialized(ron::from_str(..)) -- src/main.rs:2882, src/gui/scene.rs:142-146 */
int ptl_scene_load_file(const char* path, ptl_scene** out);
int ptl_scene_load_text(const char* ron_text, ptl_scene** out);
void ptl_scene_free(ptl_scene* s);

/* Override a named uniform's stored value (what a GUI slider / stage does). 1 = no such uniform. */
int ptl_scene_set_uniform(ptl_scene* s, const char* name, double value);
/* Trefoil uniforms in the reference's text form (TrefoilSpecial::decode / encode, src/gui/uniform.rs:23-98; its unit test:
 * "1a 2a G,1b 3b B,2a 1a S" round-trips): part 1a teleports to part 2a and is drawn in colour G, ...  Returns
 * PTL_UNKNOWN_UNIFORM (1) for an unknown name, a uniform of another kind, or text that does not decode. */
int ptl_scene_set_trefoil(ptl_scene* s, const char* name, const char* text);
int ptl_scene_get_trefoil(ptl_scene* s, const char* name, char* text, size_t cap);
/* Formula time inputs (FormulasCache::set_time / set_total_time). */
int ptl_scene_set_time(ptl_scene* s, double time, double total_time);

/* Scene::init_stage_by_name (src/gui/scene.rs:1237-1250, `render-frame --stage`): apply the named
 * animation stage's uniform / matrix overrides.  camera receives the name of the camera the stage
 * selects ("" = the scene's original camera; an inline stage camera is reported as "#<index>").
 * Returns 1 if the scene has no such stage. */
int ptl_scene_init_stage(ptl_scene* s, const char* stage, char* camera, size_t camera_cap);
/* Number of stages / cameras and their names (index past the end: returns 1). */
int ptl_scene_stage_name(ptl_scene* s, int index, char* name, size_t cap);
int ptl_scene_camera_name(ptl_scene* s, int index, char* name, size_t cap);

/* FormulasCache::set_camera_matrix (src/gui/uniform.rs:625-697): what Matrix::Camera evaluates to (column-major
 * binary64).  A renderer sets it to its camera's matrix before every draw (send_camera_object_matrix). */
int ptl_scene_set_camera_matrix(ptl_scene* s, const double m16[16]);
/* Real animations = the clips of the video pipeline (RealAnimation, src/gui/animation.rs:1014-1043).
 * ptl_scene_animation: name and duration (seconds) of clip `index`, 1 past the end.
 * ptl_scene_init_animation: Scene::init_animation_by_name (src/gui/scene.rs:1254-1267): the clip's base stage,
 * then its own replacements, then its start camera becomes the scene's current camera.  1 = no such clip. */
int ptl_scene_animation(ptl_scene* s, int index, char* name, size_t cap, double* duration);
int ptl_scene_init_animation(ptl_scene* s, const char* animation);
/* CalculatedCam (src/gui/camera.rs:22-32); matrix is column-major binary64 */
typedef struct ptl_calculated_cam {
    double look_at[3], alpha, beta, r;
    int free_movement, in_subspace, override_matrix;
    double matrix[16];
} ptl_calculated_cam;
/* Scene::update (src/gui/scene.rs:1353-1493): map wall-clock `seconds` to the formulas' `time` (fraction of the
 * current clip) and `total_time`, and compute the clip's interpolated camera ("OverrideCam") when it has a start
 * and an end camera (*has_cam = 1). */
int ptl_scene_update(ptl_scene* s, double seconds, double* time, double* total_time, int* has_cam, ptl_calculated_cam* cam);

/* AnyUniform::get: kind 0 = bool, 1 = int, 2 = float.  Returns 1 if the uniform cannot be evaluated. */
int ptl_scene_eval_uniform(ptl_scene* s, const char* name, int* kind, double* value);
/* Matrix::get as binary64, column-major.  Returns 1 if it cannot be evaluated. */
int ptl_scene_eval_matrix(ptl_scene* s, const char* name, double out16[16]);
/* Scene `cam` block: look_at xyz, alpha, beta, r, offset_after_material. */
int ptl_scene_cam(ptl_scene* s, double out7[7]);

/* Texture `index` of the scene: name ("monoportal") and path ("scenes/img/monoportal.png").
 * Returns 1 when index is past the end. */
int ptl_scene_texture(ptl_scene* s, int index, char* name, size_t name_cap, char* path, size_t path_cap);

/* Scene::generate_shader_code: returns a malloc'ed NUL-terminated HIP C++ source (free with
 * ptl_free).  flags: bit0 = bake Bool/Int scene uniforms as literals, bit1 = count segments,
 * bit2 = bake every scene uniform (ints, floats, matrices; camera and other builtins stay dynamic),
 * bit3 = clip-constant specialisation: bake every scene uniform whose evaluation reads no per-frame input (time,
 * total_time, the camera matrix) -- what stays fixed while a clip plays; a renderer checks the compiled-in values before
 * every draw and rebuilds (demoting what moved) if one no longer holds, so results never depend on the guess.
 * A RENDERER built with bit0, bit2 or bit3 also compiles its own mode switches in -- "use_panini_projection", "use_360_camera",
 * "use_180_camera", "draw_depth_map", "draw_anaglyph", "draw_side_by_side": the camera models and output modes a frame does not use
 * are then not in the kernel at all (7-10 % of the kernel time) -- and rebuilds when ptl_renderer_set_option flips one (counted by
 * ptl_renderer_rejit_count; the code-object cache keeps both).  ptl_scene_generate_source leaves them run-time values,
 * bit4 = compile the anaglyph stereo mode in (the reference's `disable_anaglyph = false`, src/main.rs:939),
 * bit5 = NO derived uniforms: by default the ray-independent half of every generated plane test whose matrix is a run-time
 * uniform (normalize(get_normal(X_mat)), both possible is_collinear verdicts) is evaluated once per uniform upload by the
 * module's prologue kernel `ptl_derive_kernel` and read back as extra uniforms -- same operations, identical frames;
 * this bit keeps the reference's per-call form (A/B measurements, tests),
 * bit14 = EXACT CR (`--exact-cr`): numerics contract 1 of rounds 1-2 -- `/`, 1/x and sqrt IEEE correctly rounded on every input --
 * instead of contract 2 (device/ptl_glsl.h: 1/x correctly rounded with the extremes flushed, a / b = a * (1/b), sqrt correctly rounded
 * with |x| < 2^-100 flushed); both are bit-exact against the oracle run with the same contract,
 * bit6 = FAST MATH, the tolerance mode (`--fast`): hardware rcp / sqrt / rsq estimates (1 ulp), a / b = a * rcp(b), FMA
 * contraction, plane tests as t = -o'.z * rcp(d'.z) without normalising the transformed direction, products with a literal zero folded
 * (-fno-signed-zeros -fno-honor-nans: with the scene state baked in, most portal matrices are mostly zeros).  Frames agree with the exact kernel to ~1e-6 per channel except at pixels where the last bit decides a path
 * (object edges); the exact kernel stays the default and the parity reference.
 * bit7 = NO deferred loop updates: by default a loop-carried ray transform in a scene snippet (`X = transform(A_mat, transform(B_mat_inv, X));`
 * on every iteration, X read only where a hit is recorded) is replaced by a counter and applied right before X is read -- the same
 * operations on the same values for the rays that read X, none for the others; identical frames (glsl_translate.h has the conditions).
 * bit12 = NO hoisted uniform work: by default every expression of a scene snippet that depends on run-time uniforms alone
 * (`b0_mat * (a_mat_inv * normal_b)`, `d_mat * c_mat_inv`, normalize() of such a normal ...), loop-carried chains of them
 * included, is computed by the prologue kernel of bit5 into members behind the derived uniforms, and the snippet reads it
 * from there -- the same expression text compiled in the same module, identical frames (host/glsl_hoist.h).  Off with bit5.
 * bit13 = NO first-trip variants: with the first-trip forms (below: every specialised build; the un-specialised kernel with bit24) every intersection-material snippet is compiled twice, and
 * the copy that runs while a ray still starts at the camera (the first trip of the bounce loop: nine trips in ten) takes the ORIGIN half of its
 * `transform(uniform matrix, ray)` chains from the prologue kernel -- the origin of every primary ray is the same uniform value --
 * while the direction half stays per ray; same operations on the same values, identical frames.  Off with bit5 / bit12.
 * bit16 = NO first-trip form of the generated plane tests: with the first-trip forms, where the scene's matrices are
 * run-time uniforms, a second copy of scene_intersect serves the trip on which every ray of a wave still starts at the camera and takes
 * `plane_inv * r.o` of every Flat object from the prologue kernel (ptl_dvo_<object>_<side>) -- the same product of the same values, identical frames.
 * Round 6: in the UN-SPECIALISED kernel (none of bit0 / bit2 / bit3 / bit20) both first-trip forms are opt-in (bit24).  Rounds 3-5 built them in
 * every kernel; the un-specialised one loses with either -- headline 0.697 -> 0.664 ms without them (-3 ... -9 % over five views), triple_portal
 * 0.445 -> 0.434, mobius 1.108 -> 1.046, monoportal equal (profiles/r06/ab_unspec_code_size*.jsonl) -- and is 12 535 instead of 17 090 instructions
 * and compiles in a quarter of the time.  The specialised builds keep them as before (the plane form still gains 1 ... 4 % in the patterns and
 * Int-baked builds, profiles/r06/ab_first_trip_planes.jsonl; the snippet copies go where rays are affine, bit24).
 * bit17 = ASYNC REJIT (read by ptl_renderer_create only; the "specialize_static" option may be switched on such a renderer: the pair
 * of kernels is rebuilt synchronously, like at creation): see ptl_renderer_rejit_pending.
 * bit18 = QUICK JIT: compile at -O1 instead of the shipped -O3 without SLP: half the hiprtc time for a 5-20 % slower kernel, identical
 * frames -- for a build that is wanted now and used briefly (the CLI's render-frame; the kernel a clip starts on).  Implies bit15;
 * ignored together with bit3 (a clip-constant build is asked for because many frames follow).
 * bit19 = NO ZERO MASKS: by default a build with bit0, bit2 or bit3 compiles in the ZERO PATTERN of every matrix uniform that stays a
 * run-time value (only Bool / Int baked; animated within the clip -- its pattern then taken over probes of the clip's `time`; demoted):
 * `transform(X_mat, ..)` of the scene snippets and the generated plane tests skip the terms whose element is zero, as a baked matrix's do
 * (same results for finite operands); a renderer rebuilds when a masked element stops being zero.  Since round 4 the pattern also names the
 * elements that are exactly +1 or -1 (their terms become `x + acc` / `acc - x`: the same operation, no bit moves), with the same rebuild rule.
 * This bit keeps the full products (A/B),
 * bit20 = SPECIALIZE PATTERNS: no animated VALUE of the scene is compiled in, only what survives while the values move -- the zero / unit
 * patterns of the matrix uniforms (as with bit19 clear), in a renderer its mode switches, and (round 5) the scene's own switches: the Bool /
 * Int uniforms whose evaluation reads no per-frame input (GUI toggles and counters such as `filter_teleported`, `show_teleported`; dead
 * branches, constant loop bounds and unrolled snippet loops follow: headline 0.50 -> 0.30 ms).  The kernel for scenes whose uniforms move
 * every frame (the reference uploads them every frame and never recompiles, src/main.rs:1266-1359): a renderer rebuilds only when a matrix
 * element leaves its pattern (then without masks, at most once per stage), a switch flips (the one that moved becomes a run-time uniform)
 * or a matrix stops being affine (bit23); with bit17 the un-specialised kernel draws meanwhile.
 * bit21 = BOUNDED SNIPPETS (opt-in): the bounce loop evaluates scene_intersect() first and hands its hit distance to the scene's
 * intersection-material snippets; a snippet of the usual shape (an accumulator filled in `if (nearer(result.scene.hit, H)) { ... }` blocks)
 * then skips the candidates beyond it, which could never be the nearest hit -- exact by construction (host/glsl_translate.h
 * `bound_nearer_blocks` has the proof and the conditions), identical frames.  Off by default: on the headline scene it measures -2 ... +13 %
 * kernel time over five views (profiles/r04/ab_bounded_snippets.jsonl),
 * bit22 = SLICES: the render entry takes its uniform block from a buffer of blocks, one per blockIdx.z (ptl_kernel_render_slices /
 * ptl_renderer_draw_slices: one launch for the blur sub-frames of a clip frame).  Same arithmetic, same scalar loads, identical frames; a
 * single draw is a batch of one,
 * bit23 = NO AFFINE RAYS: by default (round 5) a build that may shorten products (bit0, bit2, bit3 or bit20; not with bit14 / bit6) whose scene
 * matrices all have the bottom row 0 0 0 1 -- or are NaN throughout: a switched-off object -- and whose snippets never write a ray's w says in
 * its matrix-times-ray products what every ray of the reference satisfies anyway: an origin has w = 1, a direction w = 0.  The translation
 * column then costs a direction nothing and the w row folds to a constant (headline kernel -16 %); the same operations on the same values
 * for finite rays, identical frames.  A renderer rebuilds without it when a matrix -- the camera's included -- stops being affine.
 * This bit keeps the general products (A/B measurements, tests),
 * bit26 = MATERIAL TABLE IN LDS, bit27 = MATERIAL TABLE BEHIND SCALAR LOADS (round 6, A/B; both measured and left off): the reference prints one
 * `else if (i.material == X_M) return material_simple2(hit, r, <nine literals>);` per Simple material (src/gui/scene.rs:736-760).  With one of these bits
 * the nine literals of every Simple material (and of DEBUG_RED / GREEN / BLUE) sit in a table indexed by the material id and ONE call of
 * material_simple2 serves them all -- bit26: the table staged in LDS per workgroup, two ds_read_b128 per lane; bit27: the table in constant memory,
 * one scalar load per DISTINCT material of the wave (a readfirstlane loop), the grid flags on the scalar unit.  Same function, same argument
 * values: identical frames.  Headline 0.1869 ms without, 0.1905 (LDS) / 0.1872 (scalar); C2 0.0369 / 0.0383 / 0.0376; C3 0.2133 / 0.2118 / 0.2086
 * (profiles/r06/ab_material_table2.jsonl): a wave usually holds ONE material and its own copy with the literals folded beats a generic body.
 * bit25 = CHECK AFFINE (round 6, diagnostics): never affine rays; the kernel is generated with PTL_CHECK_AFFINE and its `segments` counter (bit1 is
 * implied) counts, instead of bounce-loop trips, the ray halves that reach a matrix-times-ray product or the bounce loop with a w that is
 * not 1 (origin) / 0 (direction) -- what a kernel with affine rays would have assumed wrongly.  Same frames as bit23.  See ptl_renderer_check_affine.
 * bit24 = KEEP TRANSFORM DODGES (A/B): round 4's shape of a kernel.  (a) The un-specialised kernel gets the first-trip forms of bit13 / bit16 (opt-in there
 * since round 6, see there), a kernel with affine rays its first-trip snippet copies.  (b) A kernel with affine rays (bit23 clear and everything affine) is by default generated WITHOUT the deferred loop updates (bit7) --
 * they dodge `transform(uniform matrix, ray)`, which is a handful of additions there and cheaper than the bookkeeping around it (headline baked
 * 0.2305 -> 0.2046 ms, Int-baked 0.272 -> 0.239, patterns 0.274 -> 0.239; identical frames); this bit keeps them.  Kernels without affine rays (the
 * un-specialised one above all: 0.70 against 0.89 ms) keep the deferral anyway,
 * bit15 = NO unrolling of baked loops: by default (with bit0) a counting loop of a scene snippet whose bound is a baked Int uniform
 * (<= 16 iterations) is unrolled -- the same operations in the same order, identical frames; every iteration then has its own
 * constants (scenes/portal_in_portal.ron's `size` drives an inner loop and a material index).
 * ptl_renderer_create additionally reads bits 8-11 as an occupancy hint n (0 = none):
 * the kernel is built with __launch_bounds__(256, n), i.e. at least n waves per SIMD. */
int ptl_scene_generate_source(ptl_scene* s, unsigned flags, char** source);
/* The preprocessor defines that belong to the source generated last (ptl_scene_generate_source / ptl_renderer_*), space separated: what
 * ptl_kernel_compile has to be given with it ("PTL_FIRST_TRIP", "PTL_DROP_ZERO_TERMS" -- absent when a matrix of the scene is infinite or
 * NaN: shortened products are exact for finite vectors only, so such a scene keeps every full chain -- "PTL_COUNT_SEGMENTS" ...). */
int ptl_scene_generated_defines(ptl_scene* s, char* out, size_t cap);
/* Diagnostics: how often a generation with zero patterns (bit0 / bit2 / bit3 without bit19) probed the scene for them (copies of the scene
 * stepped through the clip) and how often it reused the last result because the state the patterns depend on had not changed -- a renderer
 * with baked Bool / Int uniforms regenerates on every camera move. */
int ptl_scene_zero_mask_probes(ptl_scene* s, int* reused, int* probed);
/* Scene::uniforms + layout: descs are owned by the scene handle and stay valid until the next
 * call of this function or ptl_scene_free. */
int ptl_scene_uniform_layout(ptl_scene* s, const ptl_uniform_desc** descs, int* n, size_t* block_size);
/* Scene::set_uniforms: evaluate and upload X_mat / X_mat_inv / *_mat_teleport / user uniforms. */
int ptl_scene_set_uniforms(ptl_scene* s, ptl_kernel* k);
/* The same values without a kernel (tests): callback per uniform. */
typedef void (*ptl_uniform_cb)(void* user, const char* name, ptl_type type, const void* value);
int ptl_scene_visit_uniforms(ptl_scene* s, ptl_uniform_cb cb, void* user);
/* Which scene element produced generated line `line` (1-based) of the last generated source:
 * kind/name are written into caller buffers.  Returns 1 if the line is template text. */
int ptl_scene_source_line_owner(ptl_scene* s, int line, char* kind, size_t kind_cap, char* name, size_t name_cap,
                                int* local_line);
void ptl_free(void* p);

/* SceneRenderer::new (src/main.rs:934-1064): generate + compile + upload scene uniforms and
 * textures (paths are resolved against `asset_root`).  device = -1: no GPU, only
 * ptl_renderer_uniform_* queries work. */
int ptl_renderer_create(ptl_scene* s, int device, const char* asset_root, unsigned flags, ptl_renderer** out, char* log,
                        size_t log_cap);
/* The same with renderer options (ptl_renderer_set_option names, "specialize_static" excluded) applied BEFORE the first build: a
 * specialised renderer (flags bit0 / bit2 / bit3) compiles its mode switches in, so a caller that draws side by side, with Panini, a
 * 360 camera ... gets the kernel it will draw with from the start -- and a compile-only handle (device = -1) warms the code-object cache
 * with exactly that kernel (`portal-amd render --stereoimage` prefetches its clips' kernels this way).  An unknown name is
 * PTL_ERR_INVALID. */
int ptl_renderer_create_with_options(ptl_scene* s, int device, const char* asset_root, unsigned flags, const char* const* option_names,
                                     const double* option_values, int n_options, ptl_renderer** out, char* log, size_t log_cap);
/* CLI / GUI knobs of SceneRenderer, by the reference's field names: "render_depth", "aa_count",
 * "aa_start", "view_angle", "use_panini_projection", "panini_param", "use_360_camera",
 * "use_180_camera", "darken_by_distance", "gray_t_start", "gray_t_size", "draw_depth_map",
 * "depth_map_min", "depth_map_max", "angle_color_disable", "grid_disable",
 * "black_border_disable", "offset_after_material", "draw_side_by_side". */
/* The translation unit the renderer's current kernel was compiled from (malloc'ed, free with ptl_free): ptl_scene_generate_source's text
 * with the renderer's own mode switches compiled in where it is a specialised build -- what tools/isa_hist.py attributes instructions to. */
int ptl_renderer_kernel_source(ptl_renderer* r, char** source);
int ptl_renderer_set_option(ptl_renderer* r, const char* name, double value);
/* "concurrent_draws" K (1 = off, up to 8; round 4): draws issued with stream = NULL / the caller's stream and without a request for the
 * kernel time go round-robin to K internal streams, each with its own instance of the kernel (ptl_kernel_clone), so consecutive draws with
 * DIFFERENT uniforms -- the motion-blur sub-frames of a clip frame (src/main.rs:1798) -- overlap on the GPU instead of serialising on the one
 * uniform block a module has.  Each launch waits for what the caller's stream had queued when it was issued; nothing waits for the launches
 * until ptl_renderer_join(r, stream), which puts `stream` (NULL = the default stream) behind all of them -- call it before the frames are
 * consumed (averaged, downloaded).  Timed draws, counting draws, draws to host memory and the teleport query join by themselves.
 * Identical frames (tests/test_gpu_round2.py); measured on one MI355X it buys nothing (profiles/r04/concurrent_draws.jsonl: 1080p
 * 0.0519 ms per sub-frame with one instance, 0.0522 with two, 0.0559 with four) -- the cross-stream waits cost what the overlap saves --
 * so nothing switches it on by default.
 * "lane_fence" 0 (round 6; default 1): a draw on a lane is the kernel's packet and nothing else -- the launch does NOT wait for what the
 * caller's stream holds, and no event is recorded behind it (a lane's completion event is recorded when ptl_renderer_join or a host-side
 * wait asks for it).  The caller orders the reuse of a target itself: ptl_renderer_join before a target is read or drawn into again.
 * With two lanes and a target per lane this is two frames in flight -- frame n + 1's ramp under frame n's tail -- for a caller that queues
 * frames and looks at them later (`bench.py`'s timed region at one GPU; profiles/r06/two_streams.jsonl: headline 4K 0.187 -> 0.177 ms per
 * frame, monoportal 1080p 0.035 -> 0.027, identical bytes).
 * "lane_stagger_us" T (default 0): lanes that start together stay together -- two launches queued at the same moment share the chip evenly and
 * drain at the same moment, so only one drain per PAIR is hidden.  With T > 0 the first draw of every lane but the first, counted from the last
 * ptl_renderer_join (or host-side wait), is issued T x 2 / K microseconds after the previous lane's: a host-side spin while the GPU is busy with
 * the first launch.  Half a launch is the natural T (bench.py sets it; profiles/r06/stagger.jsonl: 20-frame batches 0.1823 -> 0.180 ms per 4K frame). */
int ptl_renderer_join(ptl_renderer* r, void* stream);
/* One launch for several draws of a renderer created with flag bit 22 (PTL_FLAG_SLICES): ptl_renderer_stage_slice does everything a draw
 * does short of launching -- camera, rebuild checks, uniform evaluation for `frame` -- and keeps the resulting uniform block as slice `index`;
 * ptl_renderer_draw_slices launches slices 0 .. n-1 at once (ptl_kernel_render_slices).  Between two stage calls: ptl_renderer_update,
 * ptl_renderer_set_option("aa_start", j), camera moves ... as between two draws.  A slice is a snapshot of the uniform block TOGETHER WITH the
 * kernel it was staged with: when the kernel is rebuilt between two stage calls (a value-baked build whose value moved, a mode switch, an
 * adopted background build) the earlier slices are still traced by the earlier kernel -- draw_slices then issues one launch per run of
 * slices that share a kernel, same frames as draws one by one -- and the texel buffers a slice names (a video texture that steps between
 * two sub-frames) live until its launch.  draw_slices without slices 0 .. n-1 staged since the last launch is PTL_ERR_INVALID.
 * draw_slices is asynchronous like a draw EXCEPT when a kernel was rebuilt or a video texture stepped between two stage calls: it then releases
 * the parked kernels and retired texel buffers behind its launches, which waits for those launches (hipFree / the kernels' completion events);
 * on an error return `elapsed_ms` covers only the runs launched so far. */
int ptl_renderer_stage_slice(ptl_renderer* r, const ptl_frame* frame, int index);
int ptl_renderer_draw_slices(ptl_renderer* r, const ptl_frame* frame, int n, void* out_rgba8, void* out_rgba32f, unsigned long long slice_pixels,
                             void* stream, float* elapsed_ms);
/* Camera (RotateAroundCam): look_at xyz, alpha, beta, r. */
int ptl_renderer_set_camera(ptl_renderer* r, const double look_at[3], double alpha, double beta, double radius);
/* `render-frame --camera NAME` (src/main.rs:2918-2926, 1442-1478): take look_at / alpha / beta / r /
 * in_subspace / free_movement / teleport matrix from a named scene camera ("#<index>" for an inline stage
 * camera); "" restores the scene's `cam` block.  Returns 1 if there is no such camera. */
int ptl_renderer_use_camera(ptl_renderer* r, const char* camera);
/* The value draw_texture would upload for a builtin or scene uniform (floats; ints as 1 float). */
int ptl_renderer_uniform_value(ptl_renderer* r, int width, int height, const char* name, float out16[16], int* n_values);
/* SceneRenderer::draw_texture (src/main.rs:1411-1428): scene.set_uniforms + set_uniforms(w,h) +
 * one launch.  Same output conventions as ptl_kernel_render. */
int ptl_renderer_draw(ptl_renderer* r, const ptl_frame* frame, void* out_rgba8, void* out_rgba32f, void* segments,
                      void* stream, float* elapsed_ms);
int ptl_renderer_draw_to_host(ptl_renderer* r, const ptl_frame* frame, uint8_t* host_rgba8, float* host_rgba32f,
                              uint64_t* host_segments, float* elapsed_ms);
/* The same through the renderer: uploads scene + builtin uniforms as teleport_external_ray does
 * (set_uniforms(0,0), teleport_light_u = 1), then ptl_kernel_teleport_ray. */
int ptl_renderer_teleport_ray(ptl_renderer* r, const double a[3], const double b[3], double out_pos[3], int* hit_object,
                              int* changed_subspace, int* teleported);
/* Build (or load from the cache) the teleport half of the renderer's current kernel now instead of at the first query. */
int ptl_renderer_prebuild_teleport(ptl_renderer* r);
/* Move the camera the way the interactive reference does every frame: set the new orbit parameters, then
 * SceneRenderer::teleport_camera (src/main.rs:1217-1264): if the straight segment from the previous camera
 * position to the new one crosses a portal, the camera's teleport matrix becomes the portal map's
 * finite-difference Jacobian (teleport_matrix, src/main.rs:1174-1215; four ray queries) so that the view
 * continues seamlessly on the other side.  *teleported = 1 if that happened, *blocked = 1 if the move was
 * undone (stop_at_objects, or no Jacobian could be formed).  ptl_renderer_set_camera, by contrast, *places*
 * the camera without looking for a crossing. */
int ptl_renderer_move_camera(ptl_renderer* r, const double look_at[3], double alpha, double beta, double radius, int* teleported,
                             int* blocked);
/* SceneRenderer::update (src/main.rs:1430-1538), the per-frame step of `render` and `render-frame`:
 * ptl_scene_update(seconds); Matrix::Camera := the camera's matrix; switch to / follow the scene's current
 * camera; apply the clip's interpolated camera; if the camera matrix moved, teleport_camera (portal crossing). */
int ptl_renderer_update(ptl_renderer* r, double seconds, int* teleported, int* blocked);

/* Current teleport matrix (binary64, column-major), subspace flag and world position of the camera. */
int ptl_renderer_camera_state(ptl_renderer* r, double teleport16[16], int* in_subspace, double position[3]);
ptl_kernel* ptl_renderer_kernel(ptl_renderer* r);
/* The binary64 primitives behind the scene's constants, exposed one by one for tests (tests/test_matrix_exact.py compares each with exact
 * arithmetic): op = "inverse" (a), "mul" (a * b), "teleport" (b * a^-1: `A_to_B_mat_teleport`, src/gui/scene.rs:624-632), "srt" (a = scale
 * xyz, b = rotate xyz, c = offset xyz: Matrix::Simple / Parametrized, src/gui/matrix.rs:555-569), "camera" (a = look_at xyz, alpha, beta, r;
 * b = the teleport matrix: RotateAroundCam::get_matrix, src/main.rs:278-304).  Matrices are 16 doubles, column-major. */
int ptl_dmath(const char* op, const double* a, const double* b, const double* c, double* out);
/* 1 when the renderer's current kernel was generated with affine rays (flag bit23 clear, a build that may shorten products, every scene
 * matrix and the camera affine, no snippet that writes a ray's w): its matrix-times-ray products spell o.w = 1 / d.w = 0.  A matrix or a
 * camera that stops being affine makes the next draw rebuild without it (counted by ptl_renderer_rejit_count); 0 then, and for every other build. */
int ptl_renderer_affine_rays(ptl_renderer* r);
/* Round 6 -- the dynamic belt behind that decision (the reference has nothing to check: `transform` multiplies all four components,
 * /root/reference/src/library.glsl:95-120).  Builds the checking kernel of the renderer's CURRENT state (flag bit25: general products, every
 * place an affine-rays kernel assumes a w counts the ray halves that arrive with another one), draws width x height with the renderer's camera
 * and options, and reports the count.  A count above zero on a renderer whose kernel has affine rays switches them off for the stage and
 * rebuilds (ptl_renderer_rejit_count goes up, ptl_last_error says why); its frames were and stay those of the general products only from then on.
 * Needs a device.  The renderer option "check_affine" = 1 (or PTL_CHECK_AFFINE=1 in the environment) runs this at 64 x 36 before the first
 * draw with every new affine-rays source; `portal-amd check` runs it when a GPU is present. */
int ptl_renderer_check_affine(ptl_renderer* r, int width, int height, unsigned long long* violations);
/* The scan behind that decision, exposed for tests: 1 when the GLSL text keeps rays affine (no Ray built from halves whose w is not spelled
 * `vec4(.., 1.)` / `vec4(.., 0.)`, no `.o` / `.d` assigned in another than the whitelisted forms, no out / inout parameter of type Ray or
 * vec4, no transform() by a matrix that is not a scene uniform), 0 with the offending text in `why`, -1 on malformed input (ptl_last_error).
 * Round 6: a whitelist over EVERY write to a ray half (x / y / z alone, or the whole half in five spelled forms), no preprocessor directive, no
 * out / inout parameter of any type, no modf / frexp, no `ray_none` (codegen.cpp; hunted by tests/test_affine_guard_fuzz.py). */
int ptl_snippets_keep_rays_affine(const char* glsl, char* why, size_t why_cap);
/* how many times a draw had to rebuild a specialised kernel since the renderer was created: a clip-constant value that moved (flags bit3),
 * a mode switch that was flipped (flags bit0 / bit2 / bit3) */
int ptl_renderer_rejit_count(ptl_renderer* r);
/* flags bit17 (ASYNC REJIT) on a specialised renderer (bits 0/2 or 3): a draw that finds its compiled-in values stale does not wait for
 * the rebuild (1-3 s of hiprtc) -- it draws with the un-specialised kernel of the scene (every build draws the same bits) while a worker
 * thread compiles the specialised source of the current state; the first draw that finds it finished and still matching loads it and
 * switches (counted by ptl_renderer_rejit_count).  Returns 1 while such a build is in flight or the un-specialised kernel is in use.
 * Reference seam: the reference recompiles its shader synchronously when the scene changes (src/gui/scene.rs:1112-1176). */
int ptl_renderer_rejit_pending(ptl_renderer* r);
void ptl_renderer_destroy(ptl_renderer* r);

/* Place the packed rows of shard (phase, stride) into a full-frame RGBA8 image (host memory). */
int ptl_deinterleave_rows(const uint8_t* shard_rgba8, const ptl_frame* frame, uint8_t* full_rgba8);

/* average_images (src/main.rs:645-722), the motion-blur step of the video pipeline, on the GPU: N RGBA8
 * sub-frames (DEVICE pointers, 16-byte aligned, any width x height) -> one RGBA8 frame:
 * per channel mean of c*c over the sub-frames (integer division), then (u8)(sqrt(mean) + 0.5); alpha = 255.
 * HBM-bound: reads 4*N bytes and writes 4 bytes per pixel.  1 <= n_frames <= 256 (up to 64 the pointers travel in the kernel
 * arguments, beyond in a device table; 256 is where the exact one-multiply integer mean ends: 65025 * 256 * 256 < 2^32).  (For one image the reference
 * hands it back untouched; callers skip the call then -- the kernel would still force alpha to 255.)  Launched on `stream`;
 * with elapsed_ms != NULL it is bracketed by HIP events and the call waits. */
int ptl_average_images(int device, const void* const* frames_rgba8, int n_frames, void* out_rgba8, int width, int height, void* stream,
                       float* elapsed_ms);

/* Device frame buffers for callers that keep frames on the GPU between kernels (sub-frames -> ptl_average_images ->
 * one download); the reference's counterpart is the macroquad render target + Texture2D::get_texture_data()
 * (src/main.rs:1041-1042,1803-1816).  ptl_device_download waits for `stream` first (it is a stream-ordered copy). */
int ptl_device_alloc(int device, size_t bytes, void** out);
int ptl_device_free(void* p);
int ptl_device_download(void* host_dst, const void* device_src, size_t bytes, void* stream);
/* Streams and events, 1:1 over HIP, for callers that overlap the download of frame i (on its own non-blocking stream, into
 * page-locked memory) with the tracing of frame i+1: record an event behind the producer, make the copy stream wait for it,
 * ptl_device_download_async, record a second event behind the copy and ptl_event_synchronize on it where the pixels are
 * consumed.  (`portal-amd render` does exactly this.) */
int ptl_stream_create(int device, void** stream);
int ptl_stream_destroy(void* stream);
int ptl_event_create(int device, void** event);
int ptl_event_destroy(void* event);
int ptl_event_record(void* event, void* stream);
int ptl_event_synchronize(void* event);
int ptl_stream_wait_event(void* stream, void* event);
int ptl_device_download_async(void* host_dst, const void* device_src, size_t bytes, void* stream);
/* Strided device-to-device copy on `stream` (1:1 over hipMemcpy2DAsync, pitches in bytes).  A rank's packed shard (source pitch =
 * one 8-row block) into another GPU's frame (destination pitch = G blocks; a pointer from ptl_ipc_open or a peer-accessible one):
 * the gather of that shard and its de-interleave in ONE transfer over the rank's own xGMI link. */
int ptl_device_copy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, void* stream);
/* A frame buffer shared by the processes of one node (one process per GPU; the reference is single-GPU): the
 * destination rank exports a ptl_device_alloc'ed buffer, the others map it and render into it with ptl_frame.in_place = 1
 * (stores go over xGMI into the destination's HBM).  A handle is PTL_IPC_HANDLE_BYTES opaque bytes to hand to the other
 * processes by any means; it cannot be opened in the exporting process.  Completion is the caller's business (a barrier
 * behind the launches). */
#define PTL_IPC_HANDLE_BYTES 64
int ptl_ipc_export(void* device_ptr, unsigned char handle[PTL_IPC_HANDLE_BYTES]);
int ptl_ipc_open(int device, const unsigned char handle[PTL_IPC_HANDLE_BYTES], void** device_ptr);
int ptl_ipc_close(void* device_ptr);
/* ---- layer 3: ONE frame across the GPUs of one node from ONE host process -------------------------------
 * The draw it shards is SceneRenderer::draw_texture (src/main.rs:1411-1428): rank g of n renders the 8-row blocks b with
 * b % n == g, with its own renderer (same scene, same flags) on devices[g]; the frame is assembled in devices[0]'s memory by
 *   PTL_GROUP_PEER_STORES  the kernels' own row stores over xGMI (peer access, ptl_frame.in_place = 1): no staging, no gather;
 *   PTL_GROUP_COPY_GATHER  a packed shard per rank + ONE strided peer copy per rank (hipMemcpy2DAsync) that gathers and
 *                          de-interleaves in the same transfer -- what an RCCL gather to one root decomposes into;
 *   PTL_GROUP_RCCL_GATHER  the packed shards gathered by RCCL itself (ONE group of ncclSend x n / ncclRecv x n into devices[0], the
 *                          "single RCCL gather" of BASELINE.json's north star), then one strided device-local copy per shard.
 *                          librccl is bound at run time (dlopen; PTL_RCCL_LIB overrides the search): create fails with
 *                          PTL_ERR_NO_DEVICE when it is absent.  Devices must be distinct for this transport.
 * No torch, no second process: a Rust host binds exactly these.  One thread drives all ranks;
 * the handle is not thread-safe.  A device may be listed more than once with the first two transports (rehearsal on a one-GPU machine).
 * Options / camera / update are forwarded to every rank's renderer (ptl_frame_group_renderer gives the individual ones;
 * keep them in step).  ptl_frame_group_draw returns after every rank has finished; *device_rgba8 is the width x height
 * RGBA8 frame on devices[0] (owned by the group, valid until the next draw with another size or destroy);
 * kernel_ms (n floats, may be NULL) receives each rank's trace-kernel time. */
typedef struct ptl_frame_group ptl_frame_group;
enum { PTL_GROUP_PEER_STORES = 0, PTL_GROUP_COPY_GATHER = 1, PTL_GROUP_RCCL_GATHER = 2 };
int ptl_frame_group_create(ptl_scene* s, const int* devices, int n_devices, const char* asset_root, unsigned flags, int transport,
                           ptl_frame_group** out, char* log, size_t log_cap);
int ptl_frame_group_size(const ptl_frame_group* g);
ptl_renderer* ptl_frame_group_renderer(ptl_frame_group* g, int rank);
int ptl_frame_group_set_option(ptl_frame_group* g, const char* name, double value);
int ptl_frame_group_use_camera(ptl_frame_group* g, const char* camera);
int ptl_frame_group_set_camera(ptl_frame_group* g, const double look_at[3], double alpha, double beta, double radius);
int ptl_frame_group_update(ptl_frame_group* g, double seconds);
int ptl_frame_group_draw(ptl_frame_group* g, int width, int height, void** device_rgba8, float* kernel_ms);
/* The pipelined form for clips ("gather of frame n overlaps tracing of frame n+1 on a second stream"): _submit enqueues the
 * trace of one frame on every rank's compute stream and its transfer to devices[0] behind it on the rank's second stream, and returns at
 * once with a ticket; _wait blocks until THAT frame is assembled and hands out its buffer (valid until the frame after the next is
 * submitted).  Two frames may be in flight -- shards, gather buffer and frame are double-buffered -- so while frame n travels over xGMI
 * the compute streams already trace frame n + 1 (set the camera / options / update for it between the two calls: uniform uploads are
 * stream-ordered behind the previous frame's kernel).  A third _submit before the oldest _wait is PTL_ERR_INVALID, and so is a _submit with
 * another frame size while a frame is in flight (the buffers are re-allocated); ptl_frame_group_draw is _submit + _wait and finishes
 * whatever is in flight first. */
int ptl_frame_group_submit(ptl_frame_group* g, int width, int height, int* ticket);
int ptl_frame_group_wait(ptl_frame_group* g, int ticket, void** device_rgba8, float* kernel_ms);
int ptl_frame_group_download(ptl_frame_group* g, uint8_t* host_rgba8);
void ptl_frame_group_destroy(ptl_frame_group* g);

/* Page-locked host memory for those downloads (PCIe-rate copies; pageable memory works too, several times slower). */
int ptl_host_alloc(size_t bytes, void** out);
int ptl_host_free(void* p);

/* The scene writer (serialize_scene_new_format + ron::ser::to_string_pretty, src/gui/scene_serialized.rs:22-24,654-1100):
 * the scene as a .ron document in the reference's own layout, malloc'ed (ptl_free).  The document the scene was loaded from is
 * kept whole and edited in step with the model (uniform values set through ptl_scene_set_uniform, the current stage, the cam
 * block), so an unmodified scene comes back byte for byte -- checked on every scene file of the reference.
 * ptl_ron_format: parse any RON text and write it back in that layout (NULL + ptl_last_error() on a syntax error). */
int ptl_scene_to_ron(ptl_scene* s, char** text);
char* ptl_ron_format(const char* text);

/* PNG I/O (RGBA8): the reference's Texture2D::from_file_with_format / Image::export_png. */
int ptl_png_read(const char* path, uint8_t** rgba8, int* width, int* height); /* free with ptl_free */
int ptl_png_write(const char* path, const uint8_t* rgba8, int width, int height); /* deflate level 6 */
/* the same with an explicit deflate level 0..9 (frame sequences that an encoder consumes and deletes: level 3 is 2.3x faster) */
int ptl_png_write_level(const char* path, const uint8_t* rgba8, int width, int height, int level);

/* ---- template engine test hooks (src/code_generation.rs) ----------------------------------- */
typedef struct ptl_strstore ptl_strstore;
ptl_strstore* ptl_strstore_new(void);
void ptl_strstore_free(ptl_strstore* s);
void ptl_strstore_add_string(ptl_strstore* s, const char* text);
void ptl_strstore_add_identifier_string(ptl_strstore* s, const char* kind, const char* name, const char* text);
/* consumes (frees) the storages it is given */
ptl_strstore* ptl_apply_template(const char* tmpl, const char* const* slot_names, ptl_strstore* const* storages, int n);
const char* ptl_strstore_text(const ptl_strstore* s);
int ptl_strstore_current_line(const ptl_strstore* s);
/* range of element (kind, name): 0 found, 1 missing; [start, end) 1-based */
int ptl_strstore_range(const ptl_strstore* s, const char* kind, const char* name, int* start, int* end);
int ptl_strstore_get_identifier(const ptl_strstore* s, int line, char* kind, size_t kind_cap, char* name, size_t name_cap,
                                int* local_line);

/* The fixed device sources embedded in the library: "glsl" (types + numerics contract),
 * "library" (prelude), "trace" (kernel template), "entry" (launchable entry points).
 * Returns NULL for an unknown name.  Lets a caller write its own kernel against the same
 * conventions and hand it to ptl_kernel_compile. */
const char* ptl_device_source(const char* which);

/* GLSL snippet -> C++ (malloc'ed, ptl_free) and the formula evaluator, exposed for tests. */
char* ptl_translate_glsl(const char* glsl);
/* The same for a file-scope library text (scene.rs:1037-1044): every function DEFINITION gets the kernel's force-inline attribute PTL_FN. */
char* ptl_translate_library_glsl(const char* glsl);
/* The distance-bound rewrite of an intersection-material snippet on its own (host/glsl_translate.h `bound_nearer_blocks`; tests): the GLSL
 * body with `&& !(H.t > ptl_far)` added to the conditions it recognises (malloc'ed, ptl_free), *bounded = how many; `out_functions`:
 * comma-separated names of scene functions with out / inout parameters.  Unchanged text and 0 when the snippet's shape does not allow it. */
char* ptl_bound_glsl(const char* glsl_body, const char* out_functions, int* bounded);
/* The uniform-work hoister on one snippet, exposed for tests (the code generator runs it on every scene snippet unless flags
 * bit12 / bit5 say otherwise).  `uniforms` lists the run-time uniforms as "type name;type name;..." (GLSL types), `out_functions`
 * the functions that write through an argument ("f;g"; "=f" marks a function the scene merely defines itself, which is then never
 * taken for the built-in of that name), `body_only` != 0 says the text is a function BODY whose parameters are
 * `params` ("r;first"; "@r" marks a ray whose origin is the camera's, as in the first-trip variant of a snippet).  Returns the rewritten GLSL (malloc'ed, ptl_free; the input itself when nothing was hoisted) and, in
 * *prologue (may be NULL), the GLSL statements for the prologue kernel, one line per created member in front as
 * "// member: type name[count]". */
char* ptl_hoist_glsl(const char* glsl, const char* uniforms, const char* out_functions, int body_only, const char* params, char** prologue);
/* names/values: free variables; returns 0 and *out, 1 if the formula is invalid or unresolvable */
int ptl_formula_eval(const char* text, const char* const* names, const double* values, int n, double time, double* out);

#ifdef __cplusplus
}
#endif
#endif /* PORTAL_AMD_H */
