// capi.cpp -- layer 2 of the C ABI: scene handles and the SceneRenderer driver.
//
// Host-side mirror of SceneRenderer (src/main.rs:732-1544) for the offline image path:
//   SceneRenderer::new          src/main.rs:934-1064   -> ptl_renderer_create
//   SceneRenderer::set_uniforms src/main.rs:1266-1359  -> builtin_uniforms()
//   RotateAroundCam::get_matrix src/main.rs:278-304    -> Camera::matrix()
//   SceneRenderer::draw_texture src/main.rs:1411-1428  -> ptl_renderer_draw
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <dirent.h>

#include <algorithm>
#include <atomic>
#include <set>
#include <thread>
#include <string>
#include <vector>

#include "../../../include/portal_amd.h"
#include "codegen.h"
#include "formula.h"
#include "glsl_hoist.h"
#include "glsl_translate.h"
#include "internal.h"
#include "scene.h"

using namespace ptl;

extern "C" int ptl_kernel_compile_prebuilt(int device, const char* hip_source, const ptl_uniform_desc* uniforms, int n_uniforms, size_t uniform_block_size,
                                           const char* const* defines, int n_defines, const void* code, size_t code_size, ptl_kernel** out, char* log,
                                           size_t log_cap);  // kernel.cpp

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;
double deg2rad(double deg) { return deg / 180.0 * kPi; }  // src/gui/common.rs:23-25

// RotateAroundCam (src/main.rs:35-340), the fields the offline path reads
struct Camera {
    DVec3 look_at;
    double alpha = deg2rad(81.0), beta = deg2rad(64.0), r = 3.5;
    double view_angle = deg2rad(90.0);
    bool use_panini_projection = false;
    double panini_param = 1.0;
    bool use_360_camera = false, use_180_camera = false;
    DMat4 teleport_matrix = DMat4::identity();
    bool in_subspace = false, free_movement = false;
    bool allow_teleport = true, stop_at_objects = false;  // src/main.rs:136-137
    DVec3 prev_cam_pos;
    bool do_not_teleport_one_frame = false;  // src/main.rs:86,1218-1222
    int from = -1;                           // RotateAroundCam::from: scene camera in use, -1 = original
    DMat4 left_eye_matrix = DMat4::identity(), right_eye_matrix = DMat4::identity();  // src/main.rs:87-90,149-152
    bool left_eye_in_subspace = false, right_eye_in_subspace = false;

    DVec3 pos_vec() const { return DVec3(std::sin(beta) * std::cos(alpha), std::cos(beta), std::sin(beta) * std::sin(alpha)) * r; }
    DMat4 matrix() const {  // src/main.rs:286-304
        DVec3 pos = pos_vec() + look_at;
        DVec3 k = (look_at - pos).normalize();
        DVec3 i = k.cross(DVec3(0.0, 1.0, 0.0)).normalize();
        DVec3 j = k.cross(i).normalize();
        DVec3 p = free_movement ? look_at : pos;
        return teleport_matrix * DMat4::from_cols({i.x, i.y, i.z, 0.0}, {j.x, j.y, j.z, 0.0}, {k.x, k.y, k.z, 0.0}, {p.x, p.y, p.z, 1.0});
    }
};

DVec3 cam_pos(const Camera& c) {  // RotateAroundCam::get_cam_pos (src/main.rs:316-318)
    DVec4 p = c.matrix().mul_vec4(DVec4(0.0, 0.0, 0.0, 1.0));
    return DVec3(p.x, p.y, p.z);
}

double calc_scale(const DMat4& m) {  // src/main.rs:1325-1333
    return (m.c[0].length() + m.c[1].length() + m.c[2].length()) / 3.0;
}

void copy_str(char* dst, size_t cap, const std::string& s) {
    if (!dst || !cap) return;
    std::strncpy(dst, s.c_str(), cap - 1);
    dst[cap - 1] = '\0';
}

ptl_type to_c_type(UniformType t) { return (ptl_type)(int)t; }

}  // namespace

struct ptl_scene {
    std::shared_ptr<Scene> scene;
    GeneratedKernel last;  // most recent generate_kernel_source() result
    std::vector<ptl_uniform_desc> descs;
    std::vector<std::string> desc_names;
    ZeroMaskCache mask_cache;  // zero patterns of the run-time matrices as probed last, and the scene state they belong to (codegen.h)
};

struct ptl_renderer {
    ptl_scene* owner = nullptr;
    std::shared_ptr<Scene> scene;
    ptl_kernel* kernel = nullptr;
    Camera cam;
    // SceneRenderer defaults, src/main.rs:1021-1040
    double offset_after_material = 0.005, gray_t_start = 10.0, gray_t_size = 200.0;
    int render_depth = 100, aa_count = 1, aa_start = 0;
    bool draw_side_by_side = false, draw_depth_map = false, angle_color_disable = false, grid_disable = false,
         black_border_disable = false, darken_by_distance = true;
    double depth_map_min = 0.0, depth_map_max = 10.0, anaglyph_p = 0.29, anaglyph_q = 0.06;
    bool draw_anaglyph = false, anaglyph_mode = false;  // anaglyph_mode = "colorful" (src/main.rs:1551-1556)
    double eye_distance = 0.07;  // src/main.rs:1028-1029
    bool swap_eyes = false;
    // draw-to-draw caching of the uploads: the reference re-evaluates and re-uploads every uniform on
    // every draw (src/main.rs:1413-1414); the values only change when the scene, an option, the camera
    // or the frame size does, so a draw of an unchanged state is just the kernel launch
    unsigned long long options_version = 1, uploaded_scene = 0, uploaded_options = 0;
    int uploaded_w = -1, uploaded_h = -1;
    // how the kernel was built (needed to re-JIT a specialised kernel when the scene changes)
    int device = -1;
    unsigned flags = 0;
    std::string asset_root;
    unsigned long long kernel_scene_version = 0;
    std::string kernel_source;
    std::map<std::string, int> kernel_switches;  // the mode switches the current specialised kernel has compiled in (KernelOptions::baked_options)
    std::vector<std::pair<std::string, MatrixPattern>> masked;  // run-time matrices whose pattern (zeros, +-1) the current kernel has compiled in (GeneratedKernel::masked)
    std::set<std::string> keep_unmasked;                   // ... and those whose pattern did not hold (clip-constant builds: demoted like keep_dynamic)
    bool shortened = false;    // the current kernel skips zero terms of matrix products (PTL_DROP_ZERO_TERMS and / or masks): exact for finite vectors
    bool full_chains = false;  // a run-time matrix turned non-finite under such a kernel: every later build of this stage keeps the full chains
    // Affine rays (codegen.h KernelOptions::affine_rays): the current kernel assumes o.w = 1 / d.w = 0, which holds while every matrix that
    // meets a ray -- the scene's (checked where the zero patterns are, and by the generator) and the CAMERA's (a run-time value in every
    // build: checked before every draw) -- has the bottom row 0 0 0 1.  One that does not switches the assumption off for this stage.
    bool affine_rays = false;  // the current kernel was generated with PTL_AFFINE_RAYS
    bool no_affine = false;    // ... and must not be any more
    // Round 6: option "check_affine" (or PTL_CHECK_AFFINE=1 in the environment): the first draw with every NEW affine-rays source first runs the
    // checking build of the same state at 64 x 36 (ptl_renderer_check_affine); a ray that met a product with another w switches the assumption off.
    int affine_returns = 0;           // returns to affine rays in this stage (at most one: build_kernel)
    bool no_affine_just_set = false;  // the rebuild in progress is the one that switches them off
    bool check_affine_on_new_source = false;
    std::string checked_source;
    unsigned long long affine_violations_seen = 0;
    // SceneRenderer::update state (src/main.rs:1430-1538)
    Camera prev_cam;
    bool has_prev_cam = false;
    CalculatedCam original_cam;  // egui memory "OriginalCam"
    // FLAG_SPECIALIZE_STATIC: what the current kernel has compiled in, for which stage, and what had to be demoted
    std::vector<UniformUpload> baked;
    std::set<std::string> keep_dynamic;
    StageRef kernel_stage;
    int rejit_count = 0;
    // PTL_FLAG_ASYNC_REJIT (bit 17): a specialised renderer whose baked values went stale does not stall the draw for the 1-3 s of a rebuild.
    // It keeps two kernels -- `spec_kernel` (the specialised build of some scene state) and `dyn_kernel` (the un-specialised build: valid
    // for every state) -- `kernel` points at the one in use, and a worker thread compiles the specialised source of the current state
    // (hiprtc, no device); the draw that finds it finished, still matching the scene, loads the code object and switches.  Every
    // build draws the same bits, so the pictures do not change with the switch -- only the kernel time does.
    struct Build {  // everything a compile needs, detached from the scene handle (which the caller keeps changing)
        std::string source;
        std::vector<std::string> defines, desc_names;
        std::vector<ptl_uniform_desc> descs;
        size_t block_size = 0;
        std::vector<UniformUpload> baked;
    };
    struct Job {
        Build build;
        std::atomic<int> state{1};  // 1 running, 2 done, 3 failed
        std::vector<char> code;
        std::thread worker;
    };
    ptl_kernel* spec_kernel = nullptr;
    ptl_kernel* dyn_kernel = nullptr;
    std::string spec_source, failed_source;
    Build want;                 // the specialised build of the scene state seen by the last draw
    std::shared_ptr<Job> job;
    // "concurrent_draws" K > 1: draws on the caller's default stream go round-robin to K internal streams, each with its OWN instance of the
    // kernel (ptl_kernel_clone: the same code object, another uniform block), so that consecutive draws with different uniforms -- the blur
    // sub-frames of a clip frame -- overlap on the GPU (tail of one under the ramp of the next) instead of serialising on the one uniform
    // block a module has.  Lane 0 draws with `kernel` itself.  ptl_renderer_join orders a stream behind everything issued so far.
    // "lane_fence" 0 (round 6): a draw on a lane is the kernel's packet and nothing else -- no event on the caller's stream for the lane to
    // wait on.  The caller then orders the reuse of a target buffer itself (ptl_renderer_join before it reads or overwrites one); what it
    // gets is two frames in flight: frame n + 1's ramp under frame n's tail (tools/two_streams.py: headline 0.187 -> 0.177 ms, 1080p 0.035 -> 0.027).
    struct Lane {
        ptl_kernel* clone = nullptr;
        void* stream = nullptr;
        void* done = nullptr;
        bool busy = false;
    };
    std::vector<std::vector<unsigned char>> staged_blocks;  // ptl_renderer_stage_slice: snapshots of the uniform block, one per slice ...
    unsigned staged_mask = 0;                               // ... and which of them are staged since the last launch
    // A slice is traced by the kernel it was staged with: a rebuild between two stage calls (a value-baked build whose value moved, a mode
    // switch, an adopted background build) compiles ANOTHER state in, and the earlier blocks are only right for the earlier kernel.
    // A kernel that is replaced while staged slices name it is parked here (with its texel buffers) until those slices are launched.
    std::vector<ptl_kernel*> staged_kernels;
    std::vector<ptl_kernel*> parked_kernels;
    int concurrent = 1;
    bool lane_fence = true;
    // "lane_stagger_us" (round 6): lanes that start together stay together -- two launches queued at the same moment share the chip evenly, end at the
    // same moment, and their drains coincide (one drain per PAIR hidden instead of one per frame).  With this option the first draw of every lane but the
    // first, counted from the last join / host-side wait, is issued that many microseconds (x 2 / K) after the previous lane's: a host-side spin while the GPU
    // is busy with the first launch.  Half a launch is the natural value (bench.py sets it; profiles/r06/stagger.jsonl: 20-frame batches 0.1823 -> 0.180 ms).
    double lane_stagger_us = 0.0;
    unsigned lane_draws_since_join = 0;
    std::vector<Lane> lanes;
    ptl_kernel* lanes_of = nullptr;  // the kernel the clones were made from
    unsigned next_lane = 0;
    void* fence = nullptr;
    // VideoRuntime (src/main.rs:771-925): per video, the sorted frame files and the frame currently bound
    struct VideoState {
        bool scanned = false;
        std::vector<std::string> frames;
        long bound = -1;
    };
    std::vector<VideoState> videos;
};

namespace {

int guarded(const std::function<int()>& fn) {
    try {
        return fn();
    } catch (const ron::ParseError& e) {
        set_last_error(e.what());
        return PTL_ERR_SCENE;
    } catch (const SceneError& e) {
        set_last_error(e.what());
        return PTL_ERR_SCENE;
    } catch (const std::exception& e) {
        set_last_error(std::string("internal error: ") + e.what());
        return PTL_ERR_INVALID;
    }
}

// The three camera matrices as the kernel gets them (binary32): bottom row 0 0 0 1?  (RotateAroundCam::get_matrix builds an affine basis,
// src/main.rs:278-304; the accumulated portal matrix in front of it is affine while the portals are.)
bool camera_is_affine(const ptl_renderer& r) {
    float f[16];
    r.cam.matrix().to_f32(f);
    if (!matrix_is_affine(f)) return false;
    if (r.draw_side_by_side || r.draw_anaglyph) {
        r.cam.left_eye_matrix.to_f32(f);
        if (!matrix_is_affine(f)) return false;
        r.cam.right_eye_matrix.to_f32(f);
        if (!matrix_is_affine(f)) return false;
    }
    return true;
}

std::vector<UniformUpload> builtin_uniforms(const ptl_renderer& r, int width, int height) {
    std::vector<UniformUpload> out;
    auto f1 = [&](const char* n, double v) {
        UniformUpload u;
        u.name = n;
        u.type = UniformType::Float1;
        u.f[0] = (float)v;
        out.push_back(u);
    };
    auto i1 = [&](const char* n, int v) {
        UniformUpload u;
        u.name = n;
        u.type = UniformType::Int1;
        u.i = v;
        out.push_back(u);
    };
    auto m4 = [&](const char* n, const DMat4& m) {
        UniformUpload u;
        u.name = n;
        u.type = UniformType::Mat4;
        m.to_f32(u.f);
        out.push_back(u);
    };
    {
        UniformUpload u;
        u.name = "_resolution";
        u.type = UniformType::Float2;
        u.f[0] = (float)width;
        u.f[1] = (float)height;
        out.push_back(u);
    }
    DMat4 cam = r.cam.matrix();
    m4("_camera", cam);
    m4("_camera_left_eye", r.cam.left_eye_matrix);
    m4("_camera_right_eye", r.cam.right_eye_matrix);
    i1("_left_eye_in_subspace", r.cam.left_eye_in_subspace ? 1 : 0);
    i1("_right_eye_in_subspace", r.cam.right_eye_in_subspace ? 1 : 0);
    m4("_camera_mul_inv", r.cam.teleport_matrix.inverse());
    i1("_camera_in_subspace", r.cam.in_subspace ? 1 : 0);
    f1("_view_angle", r.cam.view_angle);
    f1("_panini_param", r.cam.panini_param);
    i1("_use_panini_projection", r.cam.use_panini_projection ? 1 : 0);
    i1("_use_360_camera", r.cam.use_360_camera ? 1 : 0);
    i1("_use_180_camera", r.cam.use_180_camera ? 1 : 0);
    i1("_ray_tracing_depth", r.render_depth);
    i1("_aa_count", r.aa_count);
    i1("_aa_start", r.aa_start);
    i1("_draw_side_by_side", r.draw_side_by_side ? 1 : 0);
    i1("_draw_anaglyph", r.draw_anaglyph ? 1 : 0);
    f1("_anaglyph_p", r.anaglyph_p);
    f1("_anaglyph_q", r.anaglyph_q);
    i1("_anaglyph_mode", r.anaglyph_mode ? 1 : 0);
    i1("_draw_depth_map", r.draw_depth_map ? 1 : 0);
    f1("_depth_map_min", r.depth_map_min);
    f1("_depth_map_max", r.depth_map_max);
    f1("_offset_after_material", r.offset_after_material);
    f1("_t_start", r.gray_t_start);
    f1("_t_end", r.gray_t_start + r.gray_t_size);
    f1("_camera_scale", calc_scale(cam));
    f1("_left_eye_scale", calc_scale(r.cam.left_eye_matrix));
    f1("_right_eye_scale", calc_scale(r.cam.right_eye_matrix));
    i1("_angle_color_disable", r.angle_color_disable ? 1 : 0);
    i1("_grid_disable", r.grid_disable ? 1 : 0);
    i1("_black_border_disable", r.black_border_disable ? 1 : 0);
    i1("_darken_by_distance", r.darken_by_distance ? 1 : 0);
    i1("_teleport_external_ray", 0);
    return out;
}

int upload(ptl_kernel* k, const std::vector<UniformUpload>& ups) {
    for (const UniformUpload& u : ups) {
        const void* v = u.type == UniformType::Int1 ? (const void*)&u.i : (const void*)u.f;
        int rc = ptl_kernel_set_uniform(k, u.name.c_str(), to_c_type(u.type), v);
        if (rc < 0) return rc;  // unknown names are tolerated (macroquad looks names up at upload time)
    }
    return PTL_OK;
}

}  // namespace

// ---- scenes -----------------------------------------------------------------------------------
extern "C" int ptl_scene_load_file(const char* path, ptl_scene** out) {
    if (!path || !out) return PTL_ERR_INVALID;
    return guarded([&] {
        auto s = std::make_unique<ptl_scene>();
        s->scene = Scene::from_file(path);
        *out = s.release();
        return PTL_OK;
    });
}
extern "C" int ptl_scene_load_text(const char* text, ptl_scene** out) {
    if (!text || !out) return PTL_ERR_INVALID;
    return guarded([&] {
        auto s = std::make_unique<ptl_scene>();
        s->scene = Scene::from_ron_text(text);
        *out = s.release();
        return PTL_OK;
    });
}
extern "C" void ptl_scene_free(ptl_scene* s) { delete s; }
extern "C" void ptl_free(void* p) { std::free(p); }

extern "C" int ptl_scene_set_uniform(ptl_scene* s, const char* name, double value) {
    if (!s || !name) return PTL_ERR_INVALID;
    return s->scene->set_uniform_value(name, value) ? PTL_OK : PTL_UNKNOWN_UNIFORM;
}
extern "C" int ptl_scene_set_trefoil(ptl_scene* s, const char* name, const char* text) {
    if (!s || !name || !text) return PTL_ERR_INVALID;
    return guarded([&] { return s->scene->set_trefoil(name, text) ? PTL_OK : PTL_UNKNOWN_UNIFORM; });
}
extern "C" int ptl_scene_get_trefoil(ptl_scene* s, const char* name, char* text, size_t cap) {
    if (!s || !name) return PTL_ERR_INVALID;
    return guarded([&] {
        auto t = s->scene->get_trefoil(name);
        if (!t) return (int)PTL_UNKNOWN_UNIFORM;
        copy_str(text, cap, *t);
        return (int)PTL_OK;
    });
}
extern "C" int ptl_scene_set_time(ptl_scene* s, double time, double total_time) {
    if (!s) return PTL_ERR_INVALID;
    if (s->scene->time != time || s->scene->total_time != total_time) ++s->scene->version;
    s->scene->time = time;
    s->scene->total_time = total_time;
    return PTL_OK;
}
extern "C" int ptl_scene_set_camera_matrix(ptl_scene* s, const double m16[16]) {
    if (!s || !m16) return PTL_ERR_INVALID;
    DMat4 m = DMat4::from_cols({m16[0], m16[1], m16[2], m16[3]}, {m16[4], m16[5], m16[6], m16[7]}, {m16[8], m16[9], m16[10], m16[11]},
                               {m16[12], m16[13], m16[14], m16[15]});
    s->scene->camera_matrix = m;
    ++s->scene->version;
    return PTL_OK;
}
extern "C" int ptl_scene_init_stage(ptl_scene* s, const char* stage, char* camera, size_t camera_cap) {
    if (!s || !stage) return PTL_ERR_INVALID;
    return guarded([&] {
        int cam = -1;
        if (!s->scene->init_stage_by_name(stage, &cam)) return 1;
        std::string name;
        if (cam >= 0) name = s->scene->cameras[cam].name.empty() ? "#" + std::to_string(cam) : s->scene->cameras[cam].name;
        copy_str(camera, camera_cap, name);
        return PTL_OK;
    });
}
extern "C" int ptl_scene_stage_name(ptl_scene* s, int index, char* name, size_t cap) {
    if (!s || index < 0) return PTL_ERR_INVALID;
    if (index >= (int)s->scene->stages.size()) return 1;
    copy_str(name, cap, s->scene->stages[index].name);
    return PTL_OK;
}
extern "C" int ptl_scene_camera_name(ptl_scene* s, int index, char* name, size_t cap) {
    if (!s || index < 0) return PTL_ERR_INVALID;
    if (index >= (int)s->scene->cameras.size()) return 1;
    copy_str(name, cap, s->scene->cameras[index].name.empty() ? "#" + std::to_string(index) : s->scene->cameras[index].name);
    return PTL_OK;
}

extern "C" int ptl_scene_animation(ptl_scene* s, int index, char* name, size_t cap, double* duration) {
    if (!s || index < 0) return PTL_ERR_INVALID;
    if (index >= (int)s->scene->animations.size()) return 1;
    copy_str(name, cap, s->scene->animations[index].name);
    if (duration) *duration = s->scene->animations[index].duration;
    return PTL_OK;
}
extern "C" int ptl_scene_init_animation(ptl_scene* s, const char* animation) {
    if (!s || !animation) return PTL_ERR_INVALID;
    return guarded([&] { return s->scene->init_animation_by_name(animation) ? PTL_OK : 1; });
}
static void fill_cam(const CalculatedCam& c, ptl_calculated_cam* out) {
    out->look_at[0] = c.look_at.x;
    out->look_at[1] = c.look_at.y;
    out->look_at[2] = c.look_at.z;
    out->alpha = c.alpha;
    out->beta = c.beta;
    out->r = c.r;
    out->in_subspace = c.in_subspace;
    out->free_movement = c.free_movement;
    out->override_matrix = c.override_matrix;
    for (int k = 0; k < 4; ++k) {
        out->matrix[4 * k + 0] = c.matrix.c[k].x;
        out->matrix[4 * k + 1] = c.matrix.c[k].y;
        out->matrix[4 * k + 2] = c.matrix.c[k].z;
        out->matrix[4 * k + 3] = c.matrix.c[k].w;
    }
}
extern "C" int ptl_scene_update(ptl_scene* s, double seconds, double* time, double* total_time, int* has_cam, ptl_calculated_cam* cam) {
    if (!s) return PTL_ERR_INVALID;
    return guarded([&] {
        auto c = s->scene->update(seconds);
        if (time) *time = s->scene->time;
        if (total_time) *total_time = s->scene->total_time;
        if (has_cam) *has_cam = c ? 1 : 0;
        if (c && cam) fill_cam(*c, cam);
        return PTL_OK;
    });
}

extern "C" int ptl_scene_eval_uniform(ptl_scene* s, const char* name, int* kind, double* value) {
    if (!s || !name) return PTL_ERR_INVALID;
    return guarded([&] {
        auto v = s->scene->eval_uniform(s->scene->find_uniform(name));
        if (!v) return 1;
        if (kind) *kind = (int)v->kind;
        if (value) *value = v->as_f64();
        return PTL_OK;
    });
}
extern "C" int ptl_scene_eval_matrix(ptl_scene* s, const char* name, double out16[16]) {
    if (!s || !name || !out16) return PTL_ERR_INVALID;
    return guarded([&] {
        auto m = s->scene->eval_matrix(s->scene->find_matrix(name));
        if (!m) return 1;
        for (int k = 0; k < 4; ++k) {
            out16[4 * k + 0] = m->c[k].x;
            out16[4 * k + 1] = m->c[k].y;
            out16[4 * k + 2] = m->c[k].z;
            out16[4 * k + 3] = m->c[k].w;
        }
        return PTL_OK;
    });
}
extern "C" int ptl_scene_cam(ptl_scene* s, double out7[7]) {
    if (!s || !out7) return PTL_ERR_INVALID;
    const CamSettings& c = s->scene->cam;
    double v[7] = {c.look_at.x, c.look_at.y, c.look_at.z, c.alpha, c.beta, c.r, c.offset_after_material};
    std::memcpy(out7, v, sizeof v);
    return PTL_OK;
}

extern "C" int ptl_scene_texture(ptl_scene* s, int index, char* name, size_t name_cap, char* path, size_t path_cap) {
    if (!s || index < 0) return PTL_ERR_INVALID;
    if (index >= (int)s->scene->textures.size()) return 1;
    copy_str(name, name_cap, s->scene->textures[index].name);
    copy_str(path, path_cap, s->scene->textures[index].path);
    return PTL_OK;
}

static KernelOptions options_from_flags(unsigned flags) {
    KernelOptions o;
    o.specialize_ints = (flags & 1u) != 0;
    o.count_segments = (flags & 2u) != 0;
    o.specialize_all = (flags & 4u) != 0;
    o.anaglyph = (flags & 16u) != 0;
    o.specialize_static = (flags & 8u) != 0;
    o.specialize_static_ints = (flags & (1u << 20)) != 0;  // PTL_FLAG_SPECIALIZE_PATTERNS: + the scene's own switches (Bool / Int uniforms that read no per-frame input)
    o.derived_uniforms = (flags & 32u) == 0;  // PTL_FLAG_NO_DERIVED_UNIFORMS: the plain plane tests (A/B measurements, tests)
    o.fast_math = (flags & 64u) != 0;         // PTL_FLAG_FAST_MATH: tolerance mode
    o.exact_cr = (flags & 16384u) != 0;       // PTL_FLAG_EXACT_CR: numerics contract 1 (IEEE / and sqrt on every input), `--exact-cr`
    // PTL_FLAG_QUICK_JIT: -O1 instead of -O3 (a build that is wanted now and used briefly).  Not for a clip-constant build (bit 3 /
    // "specialize_static"): that one is asked for because many frames will run on it, so it keeps the full optimisation level
    o.quick_jit = (flags & 262144u) != 0 && !o.specialize_static;
    // zero patterns of the matrices that stay run-time values: with any specialisation (the un-specialised kernel has to be valid for every
    // state of the scene -- the background re-JIT draws with it meanwhile); PTL_FLAG_NO_ZERO_MASKS (bit 19) for A/B measurements and tests
    o.mask_zero_elements = (flags & (13u | (1u << 20))) != 0 && (flags & 524288u) == 0;
    o.slices_entry = (flags & (1u << 22)) != 0;  // PTL_FLAG_SLICES: the render entry reads its uniform block from a buffer of blocks, one per blockIdx.z
    o.bound_snippets = (flags & (1u << 21)) != 0;  // PTL_FLAG_BOUNDED_SNIPPETS: scene_intersect first, its distance bounds the intersection-material snippets (opt-in: measured, no gain)
    // Round 6: the first-trip forms -- a second copy of scene_intersect and of every intersection-material snippet for the trip on which all rays of a
    // wave still start at the camera -- are OPT-IN IN THE UN-SPECIALISED KERNEL (bit 24, PTL_FLAG_KEEP_TRANSFORM_DODGES) and stay the default of the
    // specialised builds (where a kernel with affine rays drops the snippet copies by itself, codegen.cpp).  Round 3 measured +4 % for them on kernels
    // whose transforms were 32 FMAs.  Today: the un-specialised headline 0.697 -> 0.664 ms without them (five views: -3 ... -9 %), triple_portal 0.445 ->
    // 0.434, mobius 1.108 -> 1.046, monoportal equal (profiles/r06/ab_unspec_code_size*.jsonl) -- and that kernel is 12 535 instead of 17 090
    // instructions, compiles in 3.5 s instead of 15.8 on this container's cores and needs no occupancy retry; the specialised builds that keep run-time
    // matrices (patterns, Int-baked) still gain 1 ... 4 % from the plane form (profiles/r06/ab_first_trip_planes.jsonl).  Identical frames either way.
    const bool first_trip_forms = (flags & (1u << 24)) != 0 || (flags & (13u | (1u << 20))) != 0;
    o.first_trip_planes = first_trip_forms && (flags & 65536u) == 0;   // PTL_FLAG_NO_FIRST_TRIP_PLANES: one scene_intersect for every trip
    if (const char* ab = std::getenv("PTL_AB_FIRST_TRIP_PLANES"); ab && (ab[0] == '0' || ab[0] == '1')) o.first_trip_planes = ab[0] == '1' && (flags & 65536u) == 0;  // (A/B hook)
    // PTL_FLAG_NO_UNROLL: keep snippet loops with baked bounds as loops (A/B measurements).  The quick build keeps them too: unrolling
    // is half of its hiprtc time for the headline scene (3.4 -> 1.8 s on this container's cores) and buys 0.05 ms of kernel
    o.unroll_baked_loops = (flags & 32768u) == 0 && !o.quick_jit;
    o.keep_transform_dodges = (flags & (1u << 24)) != 0;  // PTL_FLAG_KEEP_TRANSFORM_DODGES: deferred updates + first-trip snippet copies also with affine rays (A/B)
    // the Simple materials' literals from a table instead of one inlined material_simple2 per material (A/B: measured slower, off by default -- codegen.h):
    // PTL_FLAG_MATERIAL_TABLE_LDS (bit 26) staged in LDS, PTL_FLAG_MATERIAL_TABLE_SCALAR (bit 27) in constant memory behind scalar loads
    o.material_table = (flags & (1u << 27)) != 0 ? 2 : ((flags & (1u << 26)) != 0 ? 1 : 0);
    o.check_affine = (flags & (1u << 25)) != 0;   // PTL_FLAG_CHECK_AFFINE: general products, and `segments` counts the ray halves whose w is not 1 / 0
    o.affine_rays = (flags & (1u << 23)) == 0;    // PTL_FLAG_NO_AFFINE_RAYS: matrix-times-ray products never assume o.w = 1 / d.w = 0 (A/B measurements, tests)
    o.first_trip = first_trip_forms && (flags & 8192u) == 0;  // PTL_FLAG_NO_FIRST_TRIP: no first-trip copies of the intersection-material snippets
    o.hoist_uniform_work = (flags & 4096u) == 0;  // PTL_FLAG_NO_UNIFORM_HOIST: snippets evaluate their uniform-only expressions per ray
    return o;
}

// The renderer's mode switches that a specialised build compiles in (KernelOptions::baked_options): the camera models and output modes.
static std::map<std::string, int> mode_switches(const ptl_renderer& r) {
    // (Round 6 measured the four display toggles -- `_grid_disable`, `_black_border_disable`, `_angle_color_disable`, `_darken_by_distance` -- and the scene's
    // `teleport_light_u` compiled in as well: 10 % fewer static instructions (every scalar load with its address arithmetic and wait inside the headline's ~20
    // inlined portal tests gone), 0.1873 -> 0.1869 ms: nothing -- the kernel is bound by VALU issue, scalar work hides behind it.  profiles/r06/README.md)
    return {{"_use_panini_projection", r.cam.use_panini_projection ? 1 : 0}, {"_use_360_camera", r.cam.use_360_camera ? 1 : 0},
            {"_use_180_camera", r.cam.use_180_camera ? 1 : 0},               {"_draw_depth_map", r.draw_depth_map ? 1 : 0},
            {"_draw_anaglyph", r.draw_anaglyph ? 1 : 0},                     {"_draw_side_by_side", r.draw_side_by_side ? 1 : 0}};
}

static void refresh_generated(ptl_scene* s, unsigned flags, const std::set<std::string>* keep_dynamic = nullptr, const std::map<std::string, int>* switches = nullptr,
                              const std::set<std::string>* keep_unmasked = nullptr, bool full_chains = false, bool no_affine = false) {
    KernelOptions opts = options_from_flags(flags);
    opts.full_chains = full_chains;
    if (no_affine) opts.affine_rays = false;
    opts.mask_cache = &s->mask_cache;
    if (keep_dynamic) opts.keep_dynamic = *keep_dynamic;
    if (keep_unmasked) opts.keep_unmasked = *keep_unmasked;
    if (switches && (flags & (13u | (1u << 20))) != 0) opts.baked_options = *switches;
    CodegenFlags cg;
    cg.defer_loop_updates = (flags & 128u) == 0;  // PTL_FLAG_NO_DEFERRED_UPDATES: the snippets exactly as written (A/B measurements, tests)
    s->last = generate_kernel_source(*s->scene, cg, opts);
    s->desc_names.clear();
    s->descs.clear();
    for (auto& u : s->last.uniforms) s->desc_names.push_back(u.name);
    for (size_t k = 0; k < s->last.uniforms.size(); ++k)
        s->descs.push_back(ptl_uniform_desc{s->desc_names[k].c_str(), to_c_type(s->last.uniforms[k].type), s->last.uniforms[k].offset});
}

extern "C" int ptl_scene_generate_source(ptl_scene* s, unsigned flags, char** source) {
    if (!s || !source) return PTL_ERR_INVALID;
    return guarded([&] {
        refresh_generated(s, flags);
        *source = (char*)std::malloc(s->last.source.size() + 1);
        std::memcpy(*source, s->last.source.c_str(), s->last.source.size() + 1);
        return PTL_OK;
    });
}
extern "C" int ptl_scene_zero_mask_probes(ptl_scene* s, int* reused, int* probed) {
    if (!s) return PTL_ERR_INVALID;
    if (reused) *reused = s->mask_cache.hits;
    if (probed) *probed = s->mask_cache.misses;
    return PTL_OK;
}
extern "C" int ptl_scene_generated_defines(ptl_scene* s, char* out, size_t cap) {
    if (!s || !out || cap == 0) return PTL_ERR_INVALID;
    std::string joined;
    for (const std::string& d : s->last.defines) joined += (joined.empty() ? "" : " ") + d;
    if (joined.size() + 1 > cap) return PTL_ERR_INVALID;
    std::memcpy(out, joined.c_str(), joined.size() + 1);
    return PTL_OK;
}
extern "C" int ptl_scene_uniform_layout(ptl_scene* s, const ptl_uniform_desc** descs, int* n, size_t* block_size) {
    if (!s) return PTL_ERR_INVALID;
    return guarded([&] {
        if (s->last.source.empty()) refresh_generated(s, 0);
        if (descs) *descs = s->descs.data();
        if (n) *n = (int)s->descs.size();
        if (block_size) *block_size = s->last.uniform_block_size;
        return PTL_OK;
    });
}
extern "C" int ptl_scene_set_uniforms(ptl_scene* s, ptl_kernel* k) {
    if (!s || !k) return PTL_ERR_INVALID;
    return guarded([&] {
        std::vector<std::string> errors;
        int rc = upload(k, evaluate_scene_uniforms(*s->scene, &errors));
        if (!errors.empty()) set_last_error(errors[0]);
        return rc;
    });
}
extern "C" int ptl_scene_visit_uniforms(ptl_scene* s, ptl_uniform_cb cb, void* user) {
    if (!s || !cb) return PTL_ERR_INVALID;
    return guarded([&] {
        for (const UniformUpload& u : evaluate_scene_uniforms(*s->scene, nullptr))
            cb(user, u.name.c_str(), to_c_type(u.type), u.type == UniformType::Int1 ? (const void*)&u.i : (const void*)u.f);
        return PTL_OK;
    });
}
extern "C" int ptl_scene_source_line_owner(ptl_scene* s, int line, char* kind, size_t kind_cap, char* name, size_t name_cap, int* local_line) {
    if (!s) return PTL_ERR_INVALID;
    ElementKey key;
    int local = 0;
    if (!s->last.line_numbers.get_identifier(line, &key, &local)) return 1;
    copy_str(kind, kind_cap, key.kind);
    copy_str(name, name_cap, key.name);
    if (local_line) *local_line = local;
    return PTL_OK;
}

// ---- renderer ---------------------------------------------------------------------------------
// generate + compile + load textures: the JIT step of SceneRenderer::new (main.rs:946-1010,1066-1083)
namespace {

CalculatedCam calculated_of(const Camera& c) {  // RotateAroundCam::get_calculated_cam (src/main.rs:156-167)
    CalculatedCam out;
    out.look_at = c.look_at;
    out.alpha = c.alpha;
    out.beta = c.beta;
    out.r = c.r;
    out.in_subspace = c.in_subspace;
    out.free_movement = c.free_movement;
    out.matrix = c.teleport_matrix;
    return out;
}

bool same_matrix(const DMat4& a, const DMat4& b) {
    for (int k = 0; k < 4; ++k)
        if (a.c[k].x != b.c[k].x || a.c[k].y != b.c[k].y || a.c[k].z != b.c[k].z || a.c[k].w != b.c[k].w) return false;
    return true;
}

// `send_camera_object_matrix` (src/main.rs:147,1432-1436,1530-1534): Matrix::Camera evaluates to the camera's matrix
void send_camera_matrix(ptl_renderer* r) {
    DMat4 m = r->cam.matrix();
    if (!same_matrix(m, r->scene->camera_matrix)) {
        r->scene->camera_matrix = m;
        ++r->scene->version;
    }
}

}  // namespace

namespace {

// VideoRuntime::update (src/main.rs:849-924): frame index = round((count - 1) * clamp(uniform, 0, 1)); a changed index loads
// that PNG and binds it to the video's sampler.  IO errors are ignored like there (the sampler keeps its previous frame).
int update_videos(ptl_renderer* r) {
    const Scene& scene = *r->scene;
    if (scene.video_sources.empty() || r->device < 0) return PTL_OK;
    r->videos.resize(scene.video_sources.size());
    for (size_t k = 0; k < scene.video_sources.size(); ++k) {
        const Scene::Video& v = scene.video_sources[k];
        ptl_renderer::VideoState& st = r->videos[k];
        if (v.path.empty() || v.uniform < 0) continue;
        if (!st.scanned) {  // video_collect_frame_files: video_png/<file stem>/*.png, sorted
            st.scanned = true;
            std::string base = v.path.substr(v.path.rfind('/') == std::string::npos ? 0 : v.path.rfind('/') + 1);
            std::string stem = base.substr(0, base.rfind('.') == std::string::npos ? base.size() : base.rfind('.'));
            std::string dir = (r->asset_root.empty() ? std::string() : r->asset_root + "/") + "video_png/" + stem;
            if (DIR* d = opendir(dir.c_str())) {
                while (dirent* e = readdir(d)) {
                    std::string name = e->d_name;
                    if (name.size() > 4 && name.compare(name.size() - 4, 4, ".png") == 0) st.frames.push_back(dir + "/" + name);
                }
                closedir(d);
            }
            std::sort(st.frames.begin(), st.frames.end());
        }
        if (st.frames.empty()) continue;
        auto value = scene.eval_uniform(v.uniform);
        if (!value) continue;
        double x = value->as_f64();
        x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);  // f64::clamp; NaN stays NaN and ends up as index 0 below
        double last = (double)(st.frames.size() - 1);
        double target = std::round(last * x);
        long index = std::isnan(target) ? 0 : (long)(target < 0.0 ? 0.0 : (target > last ? last : target));
        if (index == st.bound) continue;
        uint8_t* px = nullptr;
        int w = 0, h = 0;
        if (ptl_png_read(st.frames[index].c_str(), &px, &w, &h) != PTL_OK) continue;
        int rc = ptl_kernel_set_texture(r->kernel, (v.name + "_tex").c_str(), px, w, h);
        std::free(px);
        if (rc < 0) return rc;
        st.bound = index;
    }
    return PTL_OK;
}

}  // namespace

static constexpr unsigned kAsyncRejit = 1u << 17;  // PTL_FLAG_ASYNC_REJIT
// PTL_FLAG_SPECIALIZE_PATTERNS (bit 20): no VALUE of the scene is compiled in, only what survives while values move -- the zero patterns of
// the matrices and the renderer's mode switches.  kSpecialised = the builds that compile in something a later state can invalidate.
static constexpr unsigned kPatterns = 1u << 20;
static constexpr unsigned kSpecialised = 13u | kPatterns;

// reload_textures (main.rs:1066-1083) into a freshly built kernel
static int bind_textures(ptl_renderer* r, ptl_kernel* k) {
    if (r->device < 0) return PTL_OK;
    for (const Texture& t : r->scene->textures) {
        std::string path = r->asset_root.empty() ? t.path : r->asset_root + "/" + t.path;
        uint8_t* px = nullptr;
        int w = 0, h = 0;
        if (ptl_png_read(path.c_str(), &px, &w, &h) != PTL_OK) continue;  // like the reference: the sampler stays unbound (see build_kernel)
        int trc = ptl_kernel_set_texture(k, (t.name + "_tex").c_str(), px, w, h);
        std::free(px);
        if (trc < 0) return trc;
    }
    return PTL_OK;
}

// the kernel the draws use from now on: a different one starts from a blank uniform block and blank samplers
static int activate_kernel(ptl_renderer* r, ptl_kernel* k);

// The renderer's kernels are built in two halves (kernel.cpp `ptl_kernel::split`): the render module now, the camera-teleport module when a
// query first asks for it -- a fifth of every build that a still camera never needs.  PTL_ONE_MODULE=1: one module with every entry (A/B).
static bool split_builds() {
    const char* e = std::getenv("PTL_ONE_MODULE");
    return !(e && e[0] == '1');
}

// What a renderer's build adds to the defines that belong to the generated source: the occupancy hint of the flags (bits 8-11:
// __launch_bounds__(256, waves)) and the split.
static std::vector<std::string> build_defines(const GeneratedKernel&, unsigned flags) {
    std::vector<std::string> d;
    unsigned waves = (flags >> 8) & 0xFu;
    if (waves) d.push_back("PTL_WAVES_PER_EU=" + std::to_string(waves));
    if (split_builds()) d.push_back("PTL_RENDER_MODULE");
    return d;
}

static ptl_renderer::Build snapshot_build(ptl_scene* s, unsigned flags) {
    ptl_renderer::Build b;
    b.source = s->last.source;
    b.defines = s->last.defines;
    for (auto& d : build_defines(s->last, flags)) b.defines.push_back(d);
    b.desc_names = s->desc_names;
    b.descs = s->descs;
    for (size_t k = 0; k < b.descs.size(); ++k) b.descs[k].name = b.desc_names[k].c_str();
    b.block_size = s->last.uniform_block_size;
    b.baked = s->last.baked;
    return b;
}

static int compile_build(const ptl_renderer::Build& b, int device, const std::vector<char>* prebuilt, ptl_kernel** out, char* log, size_t log_cap) {
    std::vector<const char*> defines;
    for (auto& d : b.defines) defines.push_back(d.c_str());
    // (the desc names point into b.desc_names: re-seat them, `b` may have been moved since the snapshot)
    std::vector<ptl_uniform_desc> descs = b.descs;
    for (size_t k = 0; k < descs.size(); ++k) descs[k].name = b.desc_names[k].c_str();
    if (prebuilt)
        return ptl_kernel_compile_prebuilt(device, b.source.c_str(), descs.data(), (int)descs.size(), b.block_size, defines.data(), (int)defines.size(), prebuilt->data(),
                                           prebuilt->size(), out, log, log_cap);
    return ptl_kernel_compile(device, b.source.c_str(), descs.data(), (int)descs.size(), b.block_size, defines.data(), (int)defines.size(), out, log, log_cap);
}

// ---- concurrent draws (ptl_renderer::Lane) ----
// The lanes' streams are shared by every renderer of a process (per device, created on demand, never destroyed): the HIP runtime multiplexes
// streams onto a handful of hardware queues (four by default), and two streams that land on the same queue serialise.  A stream pair per
// renderer worked for the first renderer of a process and overlapped nothing for the fourth (bench.py's other workloads, round 6: 0.0364 ms
// per 1080p frame with and without lanes; 0.0304 when the same renderer was the first).  Draws of different renderers on one lane stream
// just follow each other.
static void* pooled_lane_stream(int device, size_t index) {
    static std::mutex guard;
    static std::map<int, std::vector<void*>> pool;
    std::lock_guard<std::mutex> lock(guard);
    std::vector<void*>& streams = pool[device];
    while (streams.size() <= index) {
        void* s = nullptr;
        if (ptl_stream_create(device, &s) != PTL_OK) return nullptr;
        streams.push_back(s);
    }
    return streams[index];
}
// (a lane's `done` event is recorded when somebody asks -- here and in join_lanes -- not behind every launch: one packet less per draw)
static void wait_for_lanes(ptl_renderer* r) {  // host-side: everything issued on the lanes has finished
    r->lane_draws_since_join = 0;
    for (auto& l : r->lanes)
        if (l.busy && l.done) {
            if (ptl_event_record(l.done, l.stream) == PTL_OK) ptl_event_synchronize(l.done);
            l.busy = false;
        }
}
static void drop_lane_clones(ptl_renderer* r) {  // before the kernel they were cloned from goes away (they read its texel buffers)
    wait_for_lanes(r);
    for (auto& l : r->lanes) {
        ptl_kernel_destroy(l.clone);
        l.clone = nullptr;
    }
    r->lanes_of = nullptr;
}
static int join_lanes(ptl_renderer* r, void* stream) {  // GPU-side: `stream` continues behind every draw issued so far
    r->lane_draws_since_join = 0;
    for (auto& l : r->lanes)
        if (l.busy && l.done) {
            if (int rc = ptl_event_record(l.done, l.stream); rc != PTL_OK) return rc;
            if (int rc = ptl_stream_wait_event(stream, l.done); rc != PTL_OK) return rc;
            l.busy = false;
        }
    return PTL_OK;
}

// The renderer lets go of kernel `k` (replaced by a rebuild): destroyed, unless a staged slice still has to be traced by it.
static void retire_kernel(ptl_renderer* r, ptl_kernel* k) {
    if (!k) return;
    bool named = false;
    for (size_t j = 0; j < r->staged_kernels.size(); ++j) named = named || ((r->staged_mask >> j & 1u) && r->staged_kernels[j] == k);
    if (named)
        r->parked_kernels.push_back(k);
    else
        ptl_kernel_destroy(k);
}
static void drop_staged_slices(ptl_renderer* r) {  // forget what was staged (after the launch; when the builds are torn down)
    for (size_t j = 0; j < r->staged_kernels.size(); ++j)
        if ((r->staged_mask >> j & 1u) && r->staged_kernels[j]) {
            bool parked = std::find(r->parked_kernels.begin(), r->parked_kernels.end(), r->staged_kernels[j]) != r->parked_kernels.end();
            if (!parked) ptl_kernel_hold_textures(r->staged_kernels[j], 0);
        }
    r->staged_mask = 0;
    for (ptl_kernel* k : r->parked_kernels) ptl_kernel_destroy(k);
    r->parked_kernels.clear();
}

static int build_kernel(ptl_renderer* r, char* log, size_t log_cap) {
    ptl_scene* s = r->owner;
    if (!(r->kernel_stage == r->scene->current_stage)) r->keep_dynamic.clear();  // another stage / clip: judge afresh what is constant
    if (!(r->kernel_stage == r->scene->current_stage)) r->keep_unmasked.clear();
    if (!(r->kernel_stage == r->scene->current_stage)) r->full_chains = false;
    if (!(r->kernel_stage == r->scene->current_stage)) {
        r->no_affine = !camera_is_affine(*r) || r->affine_violations_seen > 0;
        r->affine_returns = 0;
    } else if (r->no_affine && r->affine_violations_seen == 0 && r->affine_returns < 1 && camera_is_affine(*r) && !r->no_affine_just_set) {
        // one transient non-affine camera or matrix state cost the rest of the stage ~16 % of kernel time.  A rebuild that happens for another
        // reason may return to affine rays ONCE per stage when the camera is affine again (the generator re-checks the scene's matrices itself); a second
        // break keeps them off, so a state that flickers cannot rebuild per frame.  A violation the checking build counted stays off for good.
        r->no_affine = false;
        ++r->affine_returns;
    }
    r->no_affine_just_set = false;
    r->kernel_switches = mode_switches(*r);
    refresh_generated(s, r->flags, &r->keep_dynamic, &r->kernel_switches, &r->keep_unmasked, r->full_chains, r->no_affine);
    r->baked = s->last.baked;
    r->masked = s->last.masked;
    r->affine_rays = s->last.affine_rays;
    r->shortened = !s->last.full_chains && (s->last.masked.size() > 0 || std::find(s->last.defines.begin(), s->last.defines.end(), "PTL_DROP_ZERO_TERMS") != s->last.defines.end());
    r->kernel_stage = r->scene->current_stage;
    if (r->kernel && s->last.source == r->kernel_source) {  // nothing baked in changed
        r->kernel_scene_version = r->scene->version;
        return PTL_OK;
    }
    const std::vector<std::string> extra = build_defines(s->last, r->flags);  // (not into s->last.defines: ptl_scene_generated_defines describes the source, not this build)
    std::vector<const char*> defines;
    for (auto& d : s->last.defines) defines.push_back(d.c_str());
    for (auto& d : extra) defines.push_back(d.c_str());
    ptl_kernel* k = nullptr;
    int rc = ptl_kernel_compile(r->device, s->last.source.c_str(), s->descs.data(), (int)s->descs.size(), s->last.uniform_block_size, defines.data(),
                                (int)defines.size(), &k, log, log_cap);
    if (rc != PTL_OK) return rc;
    if (r->device >= 0) {  // reload_textures (main.rs:1066-1083)
        for (const Texture& t : r->scene->textures) {
            std::string path = r->asset_root.empty() ? t.path : r->asset_root + "/" + t.path;
            uint8_t* px = nullptr;
            int w = 0, h = 0;
            if (ptl_png_read(path.c_str(), &px, &w, &h) != PTL_OK) {
                // like the reference (data.texture_errors, src/main.rs:1082-1084): keep going, the sampler stays
                // unbound and reads as (0, 0, 0, 1); the message remains available through ptl_last_error()
                continue;
            }
            int trc = ptl_kernel_set_texture(k, (t.name + "_tex").c_str(), px, w, h);
            std::free(px);
            if (trc < 0) {
                ptl_kernel_destroy(k);
                return trc;
            }
        }
    }
    if (r->lanes_of == r->kernel) drop_lane_clones(r);
    wait_for_lanes(r);
    retire_kernel(r, r->kernel);
    r->kernel = k;
    r->kernel_source = s->last.source;
    r->kernel_scene_version = r->scene->version;
    r->uploaded_scene = 0;  // a fresh uniform block: upload everything again
    r->uploaded_options = 0;
    for (auto& v : r->videos) v.bound = -1;  // ... and fresh samplers: bind the current video frames again
    return update_videos(r);
}

// Background re-JIT bookkeeping around a synchronous build_kernel() (renderer creation; a switch of the specialisation bits):
// `kernel` aliases `spec_kernel` or `dyn_kernel` in that mode, so the pair is dropped as a whole before the rebuild (the worker joined
// first) and the freshly built specialised kernel seeds it again afterwards.
static void drop_async_kernels(ptl_renderer* r) {
    if (r->job) {
        if (r->job->worker.joinable()) r->job->worker.join();
        r->job.reset();
    }
    if (r->spec_kernel || r->dyn_kernel) {
        drop_lane_clones(r);
        drop_staged_slices(r);  // (a switch of the specialisation bits between stage calls: what was staged is gone with its kernels)
        ptl_kernel_destroy(r->spec_kernel);
        ptl_kernel_destroy(r->dyn_kernel);
        r->spec_kernel = r->dyn_kernel = r->kernel = nullptr;
        r->kernel_source.clear();
    }
    r->spec_source.clear();
    r->failed_source.clear();
    r->want = ptl_renderer::Build{};
}
static void seed_async_kernels(ptl_renderer* r) {
    if ((r->flags & kAsyncRejit) != 0 && (r->flags & kSpecialised) != 0 && r->device >= 0) {  // the first kernel is the specialised one of this state
        r->spec_kernel = r->kernel;
        r->spec_source = r->kernel_source;
        r->want = snapshot_build(r->owner, r->flags);
    }
}

static int set_plain_option(ptl_renderer* r, const std::string& n, double v);

extern "C" int ptl_renderer_create_with_options(ptl_scene* s, int device, const char* asset_root, unsigned flags, const char* const* option_names,
                                                const double* option_values, int n_options, ptl_renderer** out, char* log, size_t log_cap) {
    if (!s || !out || n_options < 0 || (n_options > 0 && (!option_names || !option_values))) return PTL_ERR_INVALID;
    if (log && log_cap) log[0] = '\0';
    return guarded([&] {
        auto r = std::make_unique<ptl_renderer>();
        r->owner = s;
        r->scene = s->scene;
        r->device = device;
        r->flags = flags;
        r->asset_root = asset_root ? asset_root : "";
        if (const char* e = std::getenv("PTL_CHECK_AFFINE"); e && e[0] == '1') r->check_affine_on_new_source = true;
        // options first: a specialised build compiles the mode switches in (mode_switches), so the FIRST build is already the one the
        // caller will draw with (`render --stereoimage`: draw_side_by_side) instead of a build nothing runs on plus a rebuild
        for (int k = 0; k < n_options; ++k) {
            if (!option_names[k]) return (int)PTL_ERR_INVALID;
            int orc = set_plain_option(r.get(), option_names[k], option_values[k]);
            if (orc != PTL_OK) {
                set_last_error(std::string("ptl_renderer_create_with_options: unknown option `") + option_names[k] + "`");
                return (int)PTL_ERR_INVALID;
            }
        }
        int rc = build_kernel(r.get(), log, log_cap);
        if (rc != PTL_OK) return rc;
        seed_async_kernels(r.get());
        // cam.set_cam(scene.cam); offset_after_material from the scene (main.rs:1057-1059)
        const CamSettings& c = s->scene->cam;
        r->cam.look_at = c.look_at;
        r->cam.alpha = c.alpha;
        r->cam.beta = c.beta;
        r->cam.r = c.r;
        r->offset_after_material = c.offset_after_material;
        r->cam.prev_cam_pos = cam_pos(r->cam);  // main.rs:1058
        r->original_cam = calculated_of(r->cam);  // render_frame inserts "OriginalCam" up front (main.rs:2893-2896)
        *out = r.release();
        return PTL_OK;
    });
}
extern "C" int ptl_renderer_create(ptl_scene* s, int device, const char* asset_root, unsigned flags, ptl_renderer** out, char* log,
                                   size_t log_cap) {
    return ptl_renderer_create_with_options(s, device, asset_root, flags, nullptr, nullptr, 0, out, log, log_cap);
}

// every option that is a plain field (no rebuild): PTL_OK, or PTL_UNKNOWN_UNIFORM for a name that is not one
static int set_plain_option(ptl_renderer* r, const std::string& n, double v) {
    bool b = v > 0.5;
    if (n == "check_affine") {
        r->check_affine_on_new_source = b;
        return PTL_OK;
    }
    if (n == "render_depth") r->render_depth = (int)v;
    else if (n == "aa_count") r->aa_count = (int)v;
    else if (n == "aa_start") r->aa_start = (int)v;
    else if (n == "view_angle") r->cam.view_angle = v;
    else if (n == "use_panini_projection") r->cam.use_panini_projection = b;
    else if (n == "panini_param") r->cam.panini_param = v;
    else if (n == "use_360_camera") r->cam.use_360_camera = b;
    else if (n == "use_180_camera") r->cam.use_180_camera = b;
    else if (n == "darken_by_distance") r->darken_by_distance = b;
    else if (n == "gray_t_start") r->gray_t_start = v;
    else if (n == "gray_t_size") r->gray_t_size = v;
    else if (n == "draw_depth_map") r->draw_depth_map = b;
    else if (n == "depth_map_min") r->depth_map_min = v;
    else if (n == "depth_map_max") r->depth_map_max = v;
    else if (n == "angle_color_disable") r->angle_color_disable = b;
    else if (n == "grid_disable") r->grid_disable = b;
    else if (n == "black_border_disable") r->black_border_disable = b;
    else if (n == "offset_after_material") r->offset_after_material = v;
    else if (n == "draw_side_by_side") r->draw_side_by_side = b;
    else if (n == "in_subspace") r->cam.in_subspace = b;
    else if (n == "draw_anaglyph") r->draw_anaglyph = b;
    else if (n == "anaglyph_mode") r->anaglyph_mode = b;
    else if (n == "anaglyph_p") r->anaglyph_p = v;
    else if (n == "anaglyph_q") r->anaglyph_q = v;
    else if (n == "eye_distance") r->eye_distance = v;
    else if (n == "swap_eyes") r->swap_eyes = b;
    else if (n == "allow_teleport") r->cam.allow_teleport = b;    // RotateAroundCam toggles, src/main.rs:136-137
    else if (n == "stop_at_objects") r->cam.stop_at_objects = b;
    else if (n == "concurrent_draws") {  // 1 = off (the default); K <= 8 kernel instances on K internal streams (ptl_renderer::Lane)
        int k = (int)v;
        if (k < 1 || k > 8) return PTL_ERR_INVALID;
        if (k != r->concurrent) {
            drop_lane_clones(r);
            r->concurrent = k;
        }
        return PTL_OK;
    }
    else if (n == "lane_stagger_us") {  // the first draw of lanes 2 .. K after a join comes this much (x 2 / K) later than the previous lane's (ptl_renderer::lane_stagger_us)
        if (!(v >= 0.0) || v > 1e6) return PTL_ERR_INVALID;
        r->lane_stagger_us = v;
        return PTL_OK;
    }
    else if (n == "lane_fence") {  // 1 (default): a lane's launch waits for what the caller's stream holds; 0: it does not (ptl_renderer::Lane)
        r->lane_fence = b;
        return PTL_OK;
    }
    else return PTL_UNKNOWN_UNIFORM;
    ++r->options_version;
    return PTL_OK;
}

extern "C" int ptl_renderer_set_option(ptl_renderer* r, const char* name, double v) {
    if (!r || !name) return PTL_ERR_INVALID;
    std::string n = name;
    if (n == "specialize_static") {  // switch clip-constant specialisation (flags bit3) on or off for what follows
        unsigned want = v > 0.5 ? (r->flags | 8u) : (r->flags & ~8u);
        if (want == r->flags) return PTL_OK;
        return guarded([&] {
            // With PTL_FLAG_ASYNC_REJIT `kernel` is one of spec_kernel / dyn_kernel and a worker may be compiling for the old flags: the
            // pair is dropped as a whole (build_kernel would free only the alias), rebuilt synchronously like at creation, and re-seeded.
            const bool async = (r->flags & kAsyncRejit) != 0 && r->device >= 0;
            if (async) drop_async_kernels(r);
            r->flags = want;
            r->keep_dynamic.clear();
            int rc = build_kernel(r, nullptr, 0);
            if (rc != PTL_OK) return rc;
            if (async) seed_async_kernels(r);
            ++r->options_version;
            return (int)PTL_OK;
        });
    }
    return set_plain_option(r, n, v);
}

extern "C" int ptl_renderer_set_camera(ptl_renderer* r, const double look_at[3], double alpha, double beta, double radius) {
    if (!r || !look_at) return PTL_ERR_INVALID;
    r->cam.look_at = DVec3(look_at[0], look_at[1], look_at[2]);
    r->cam.alpha = alpha;
    r->cam.beta = beta;
    r->cam.r = radius;
    r->cam.prev_cam_pos = cam_pos(r->cam);  // placing the camera is not a move: no portal crossing is looked for
    ++r->options_version;
    return PTL_OK;
}

extern "C" int ptl_renderer_use_camera(ptl_renderer* r, const char* camera) {
    if (!r || !camera) return PTL_ERR_INVALID;
    return guarded([&] {
        std::string name = camera;
        ++r->options_version;
        if (name.empty()) {  // original camera: scene.cam, teleport matrix = I (RotateAroundCam::set_cam)
            const CamSettings& c = r->scene->cam;
            r->cam.look_at = c.look_at;
            r->cam.alpha = c.alpha;
            r->cam.beta = c.beta;
            r->cam.r = c.r;
            r->cam.teleport_matrix = DMat4::identity();
            r->cam.in_subspace = false;
            r->cam.free_movement = false;
            r->cam.from = r->scene->current_cam = -1;
            return PTL_OK;
        }
        int idx = name[0] == '#' ? std::atoi(name.c_str() + 1) : r->scene->find_camera(name);
        if (idx < 0 || idx >= (int)r->scene->cameras.size()) return 1;
        const SceneCamera& c = r->scene->cameras[idx];
        auto look = r->scene->camera_look_at(c);
        if (!look) return 1;
        if (r->cam.from < 0) r->original_cam = calculated_of(r->cam);
        // SceneRenderer::update (src/main.rs:1465-1477)
        r->cam.alpha = c.alpha;
        r->cam.beta = c.beta;
        r->cam.r = c.r;
        r->cam.look_at = *look;
        r->cam.teleport_matrix = c.teleport;
        r->cam.in_subspace = c.in_subspace;
        r->cam.free_movement = c.free_movement;
        if (r->cam.free_movement) r->cam.look_at = r->cam.pos_vec() + r->cam.look_at;
        r->cam.from = r->scene->current_cam = idx;
        r->cam.do_not_teleport_one_frame = true;
        return PTL_OK;
    });
}

extern "C" int ptl_renderer_uniform_value(ptl_renderer* r, int width, int height, const char* name, float out16[16], int* n_values) {
    if (!r || !name || !out16) return PTL_ERR_INVALID;
    return guarded([&] {
        send_camera_matrix(r);  // Matrix::Camera uniforms: the value a draw would upload (prepare_draw does the same)
        auto all = builtin_uniforms(*r, width, height);
        auto scene_vals = evaluate_scene_uniforms(*r->scene, nullptr);
        all.insert(all.end(), scene_vals.begin(), scene_vals.end());
        for (const UniformUpload& u : all) {
            if (u.name != name) continue;
            int n = u.type == UniformType::Mat4 ? 16 : u.type == UniformType::Float2 ? 2 : u.type == UniformType::Float3 ? 3 : 1;
            if (u.type == UniformType::Int1) out16[0] = (float)u.i;
            else std::memcpy(out16, u.f, sizeof(float) * n);
            if (n_values) *n_values = n;
            return PTL_OK;
        }
        return (int)PTL_UNKNOWN_UNIFORM;
    });
}

// Zero patterns compiled into the current kernel (r->masked) against the values a draw is about to upload: a matrix with a non-zero where
// the pattern says zero is demoted (keep_unmasked) and reported.
static bool zero_patterns_broken(ptl_renderer* r, const std::vector<UniformUpload>& values) {
    bool broken = false;
    // a kernel with shortened products is exact for finite vectors: a matrix that got infinite elements, or NaN beside numbers, since the kernel was generated
    // (the generator checks the values it sees, codegen.cpp `full_chains`) asks for the full chains from here on
    if (r->shortened && !r->full_chains)
        for (const UniformUpload& v : values)
            if (v.type == UniformType::Mat4 && matrix_breaks_short_chains(v.f)) r->full_chains = broken = true;
    // ... and a kernel with affine rays is exact while every scene matrix maps w = 1 to 1 and w = 0 to 0 (or is NaN throughout: a switched-off object)
    if (r->affine_rays && !r->no_affine)
        for (const UniformUpload& v : values) {
            if (v.type != UniformType::Mat4 || matrix_is_affine(v.f)) continue;
            bool all_nan = true;
            for (int k = 0; k < 16; ++k) all_nan = all_nan && std::isnan(v.f[k]);
            if (!all_nan) r->no_affine = r->no_affine_just_set = broken = true;
        }
    for (auto& [name, mask] : r->masked)
        for (const UniformUpload& v : values) {
            if (v.name != name || v.type != UniformType::Mat4) continue;
            if (!pattern_holds(mask, v.f)) {  // a non-zero where the kernel skips a term, or a +-1 the kernel has as a literal and the matrix no longer holds
                r->keep_unmasked.insert(name);
                broken = true;
            }
            break;
        }
    // one broken pattern says the probes did not see this clip's motion: every mask goes, so that a clip costs at most ONE extra rebuild
    if (!r->keep_unmasked.empty())
        for (auto& m : r->masked) r->keep_unmasked.insert(m.first);
    return broken;
}

static int activate_kernel(ptl_renderer* r, ptl_kernel* k) {
    if (r->kernel == k) return PTL_OK;
    r->kernel = k;
    r->uploaded_scene = 0;
    r->uploaded_options = 0;
    r->uploaded_w = r->uploaded_h = -1;
    for (auto& v : r->videos) v.bound = -1;
    return update_videos(r);
}

// PTL_FLAG_ASYNC_REJIT: pick the kernel for this draw without ever waiting for a compile of the specialised source (see ptl_renderer::Job).
static int async_select_kernel(ptl_renderer* r) {
    ptl_scene* s = r->owner;
    const bool changed = r->kernel_scene_version != r->scene->version || !(r->kernel_stage == r->scene->current_stage) || mode_switches(*r) != r->kernel_switches ||
                         (r->affine_rays && !r->no_affine && !camera_is_affine(*r));  // (a camera that stopped being affine: the affine-rays kernel is no longer valid)
    if (!changed && !r->job && r->kernel == r->spec_kernel) return PTL_OK;
    if (changed) {
        if (!(r->kernel_stage == r->scene->current_stage)) {
            r->keep_dynamic.clear();
            r->keep_unmasked.clear();
            r->full_chains = false;
            r->no_affine = false;
        }
        if (!camera_is_affine(*r)) r->no_affine = true;
        if ((r->flags & (8u | kPatterns)) != 0 && (r->flags & 5u) == 0) {  // clip-constant specialisation: a compiled-in value that moved becomes a run-time uniform
            std::vector<UniformUpload> values = evaluate_scene_uniforms(*r->scene, nullptr);
            size_t at = 0;
            for (const UniformUpload& b : r->want.baked) {
                while (at < values.size() && values[at].name != b.name) ++at;
                if (at == values.size()) break;
                if (!values[at].same_value(b)) r->keep_dynamic.insert(b.name);
            }
            zero_patterns_broken(r, values);
        }
        r->kernel_switches = mode_switches(*r);
        refresh_generated(s, r->flags, &r->keep_dynamic, &r->kernel_switches, &r->keep_unmasked, r->full_chains, r->no_affine);  // the specialised source of the CURRENT state (generation is milliseconds)
        r->masked = s->last.masked;
        r->affine_rays = s->last.affine_rays;
        r->shortened = !s->last.full_chains && (s->last.masked.size() > 0 || std::find(s->last.defines.begin(), s->last.defines.end(), "PTL_DROP_ZERO_TERMS") != s->last.defines.end());
        r->want = snapshot_build(s, r->flags);
        r->kernel_scene_version = r->scene->version;
        r->kernel_stage = r->scene->current_stage;
    }
    if (r->spec_kernel && r->spec_source == r->want.source) {  // (also: the state moved back to what the specialised kernel was built for)
        r->baked = r->want.baked;
        return activate_kernel(r, r->spec_kernel);
    }
    if (r->job && r->job->state.load() != 1) {  // the worker has finished
        std::shared_ptr<ptl_renderer::Job> job = r->job;
        if (job->worker.joinable()) job->worker.join();
        r->job.reset();
        if (job->state.load() == 2 && job->build.source == r->want.source) {
            ptl_kernel* k = nullptr;
            int rc = compile_build(job->build, r->device, &job->code, &k, nullptr, 0);  // module load only: the code object is there
            if (rc == PTL_OK && (rc = bind_textures(r, k)) < 0) ptl_kernel_destroy(k);
            if (rc == PTL_OK) {
                ptl_kernel* old = r->spec_kernel;
                r->spec_kernel = k;
                r->spec_source = job->build.source;
                r->kernel_source = r->spec_source;
                r->baked = job->build.baked;
                ++r->rejit_count;
                rc = activate_kernel(r, k);
                if (r->lanes_of == old) drop_lane_clones(r);
                retire_kernel(r, old);  // (never the active one: `k` has just been activated; parked while a staged slice names it)
                return rc;
            }
            r->failed_source = job->build.source;
        } else if (job->state.load() == 3) {
            r->failed_source = job->build.source;  // does not compile: stay on the un-specialised kernel, do not try this source again
        }
    }
    if (!r->job && r->want.source != r->failed_source) {
        auto job = std::make_shared<ptl_renderer::Job>();
        job->build = r->want;
        job->worker = std::thread([job] {
            ptl_kernel* k = nullptr;
            if (compile_build(job->build, -1, nullptr, &k, nullptr, 0) == PTL_OK) {
                const void* data = nullptr;
                size_t size = 0;
                ptl_kernel_code_object(k, &data, &size);
                job->code.assign(static_cast<const char*>(data), static_cast<const char*>(data) + size);
                ptl_kernel_destroy(k);
                job->state.store(2);
            } else {
                job->state.store(3);
            }
        });
        r->job = job;
    }
    if (!r->dyn_kernel) {  // first need: built here, once (a cached code object makes it a module load)
        refresh_generated(s, r->flags & ~kSpecialised);
        ptl_renderer::Build dyn = snapshot_build(s, r->flags);
        ptl_kernel* k = nullptr;
        int rc = compile_build(dyn, r->device, nullptr, &k, nullptr, 0);
        if (rc == PTL_OK && (rc = bind_textures(r, k)) < 0) ptl_kernel_destroy(k);
        if (rc != PTL_OK) return rc;
        r->dyn_kernel = k;
    }
    return activate_kernel(r, r->dyn_kernel);
}

static int prepare_draw(ptl_renderer* r, const ptl_frame* frame) {
    send_camera_matrix(r);
    const bool async = (r->flags & kAsyncRejit) != 0 && (r->flags & kSpecialised) != 0 && r->device >= 0;
    if (async) {
        int rc = async_select_kernel(r);
        if (rc != PTL_OK) return rc;
    }
    if (!async && (r->flags & kSpecialised) != 0 && mode_switches(*r) != r->kernel_switches) {
        // a camera model / output mode was switched: the specialised kernel has the old one compiled in (and the new one compiled out)
        int rc = build_kernel(r, nullptr, 0);
        if (rc != PTL_OK) return rc;
        ++r->rejit_count;
    }
    if (!async && r->affine_rays && !r->no_affine && !camera_is_affine(*r)) {
        // the camera went through something that is not an affine map: the kernel's w = 1 / w = 0 no longer holds for primary rays
        r->no_affine = r->no_affine_just_set = true;
        int rc = build_kernel(r, nullptr, 0);
        if (rc != PTL_OK) return rc;
        ++r->rejit_count;
    }
    if (!async && (r->flags & 5u) != 0 && r->kernel_scene_version != r->scene->version) {
        // values are baked into a specialised kernel: the scene changed, so JIT again (cached by source hash)
        int rc = build_kernel(r, nullptr, 0);
        if (rc != PTL_OK) return rc;
    }
    if (r->uploaded_scene != r->scene->version) {
        std::vector<std::string> errors;
        std::vector<UniformUpload> values = evaluate_scene_uniforms(*r->scene, &errors);  // scene.set_uniforms
        if (!async && (r->flags & (8u | kPatterns)) != 0 && (r->flags & 5u) == 0) {
            // clip-constant specialisation: the kernel stays valid as long as every compiled-in value still holds; a value
            // that moved after all is demoted to a run-time uniform and the kernel is built again (cached by source hash)
            bool stale = !(r->kernel_stage == r->scene->current_stage);
            size_t at = 0;
            for (const UniformUpload& b : r->baked) {
                while (at < values.size() && values[at].name != b.name) ++at;  // same order as at generation time
                if (at == values.size()) {
                    stale = true;
                    break;
                }
                if (!values[at].same_value(b)) {
                    r->keep_dynamic.insert(b.name);
                    stale = true;
                }
            }
            if (zero_patterns_broken(r, values)) stale = true;  // an animated matrix left the zero pattern its products were shortened for
            if (stale) {
                int rc = build_kernel(r, nullptr, 0);
                if (rc != PTL_OK) return rc;
                ++r->rejit_count;
            }
        }
        int rc = upload(r->kernel, values);
        if (rc < 0) return rc;
        r->uploaded_scene = r->scene->version;
    }
    if (r->uploaded_options != r->options_version || r->uploaded_w != frame->width || r->uploaded_h != frame->height) {
        int rc = upload(r->kernel, builtin_uniforms(*r, frame->width, frame->height));  // self.set_uniforms(w, h)
        if (rc < 0) return rc;
        r->uploaded_options = r->options_version;
        r->uploaded_w = frame->width;
        r->uploaded_h = frame->height;
    }
    return PTL_OK;
}

// One draw of a renderer with "concurrent_draws" K > 1, issued on the caller's stream `stream` without a request for its time: it goes to
// the next of K lanes.  prepare_draw leaves the complete current state in the primary kernel's host copy of the uniform block; a clone
// takes that copy over and uploads it behind its own previous launch, on its own stream.  The launch waits (GPU-side) for what the
// caller's stream has queued so far -- the consumer of this target buffer from the previous round -- and nothing waits for the launch
// until ptl_renderer_join.
static int draw_on_a_lane(ptl_renderer* r, const ptl_frame* frame, void* out_rgba8, void* out_rgba32f, void* stream) {
    if ((int)r->lanes.size() != r->concurrent) {
        drop_lane_clones(r);
        for (auto& l : r->lanes)
            if (l.done) ptl_event_destroy(l.done);
        r->lanes.assign((size_t)r->concurrent, ptl_renderer::Lane{});
        for (size_t i = 0; i < r->lanes.size(); ++i) {
            ptl_renderer::Lane& l = r->lanes[i];
            l.stream = pooled_lane_stream(r->device, i);
            if (!l.stream) return PTL_ERR_HIP;
            if (int rc = ptl_event_create(r->device, &l.done); rc != PTL_OK) return rc;
        }
        if (!r->fence)
            if (int rc = ptl_event_create(r->device, &r->fence); rc != PTL_OK) return rc;
    }
    if (r->lanes_of != r->kernel) {  // first use, or the kernel was rebuilt / switched: instances of THIS code object
        drop_lane_clones(r);
        for (size_t i = 1; i < r->lanes.size(); ++i)
            if (int rc = ptl_kernel_clone(r->kernel, &r->lanes[i].clone); rc != PTL_OK) return rc;
        r->lanes_of = r->kernel;
    }
    const size_t idx = r->next_lane++ % r->lanes.size();
    ptl_renderer::Lane& lane = r->lanes[idx];
    ptl_kernel* k = idx == 0 ? r->kernel : lane.clone;
    if (idx != 0)
        if (int rc = ptl_kernel_copy_uniforms(k, r->kernel); rc != PTL_OK) return rc;
    if (r->lane_fence) {
        if (int rc = ptl_event_record(r->fence, stream); rc != PTL_OK) return rc;
        if (int rc = ptl_stream_wait_event(lane.stream, r->fence); rc != PTL_OK) return rc;
    }
    if (r->lane_stagger_us > 0.0 && r->lane_draws_since_join >= 1 && r->lane_draws_since_join < r->lanes.size()) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::nanoseconds((long long)(r->lane_stagger_us * 2000.0 / (double)r->lanes.size()));
        while (std::chrono::steady_clock::now() < until) {
        }
    }
    ++r->lane_draws_since_join;
    if (int rc = ptl_kernel_render(k, frame, out_rgba8, out_rgba32f, nullptr, lane.stream, nullptr); rc != PTL_OK) return rc;
    lane.busy = true;
    return PTL_OK;
}

extern "C" int ptl_renderer_stage_slice(ptl_renderer* r, const ptl_frame* frame, int index) {
    if (!r || !frame || index < 0 || index >= 16) return PTL_ERR_INVALID;
    return guarded([&] {
        int rc = prepare_draw(r, frame);
        if (rc < 0) return rc;
        // kept as a snapshot of the kernel's host copy of the uniform block: the block layout is the scene's, not the build's, so the
        // snapshot outlives a rebuild of the kernel between two stage calls (a clip-constant build whose compiled-in value moved)
        if (r->staged_blocks.size() < 16) r->staged_blocks.resize(16);
        if (r->staged_kernels.size() < 16) r->staged_kernels.resize(16, nullptr);
        std::vector<unsigned char>& b = r->staged_blocks[index];
        b.resize(ptl_kernel_uniform_block_size(r->kernel));
        rc = ptl_kernel_snapshot_uniforms(r->kernel, b.data(), b.size());
        if (rc != PTL_OK) return rc;
        if ((r->staged_mask >> index & 1u) && r->staged_kernels[index]) {  // staged twice: the earlier one is dropped
            bool parked = std::find(r->parked_kernels.begin(), r->parked_kernels.end(), r->staged_kernels[index]) != r->parked_kernels.end();
            if (!parked) ptl_kernel_hold_textures(r->staged_kernels[index], 0);
        }
        // the block names the texel buffers bound NOW (a video texture may step before the next stage call): they stay until the launch
        ptl_kernel_hold_textures(r->kernel, 1);
        r->staged_kernels[index] = r->kernel;
        r->staged_mask |= 1u << index;
        return (int)PTL_OK;
    });
}
extern "C" int ptl_renderer_draw_slices(ptl_renderer* r, const ptl_frame* frame, int n, void* out_rgba8, void* out_rgba32f, unsigned long long slice_pixels,
                                        void* stream, float* elapsed_ms) {
    if (!r || !frame || n < 1 || n > 16) return PTL_ERR_INVALID;
    return guarded([&] {
        const unsigned want = (1u << n) - 1u;
        if ((r->staged_mask & want) != want) {
            set_last_error("ptl_renderer_draw_slices: slices 0 .. n-1 are not all staged (ptl_renderer_stage_slice) since the last launch");
            return (int)PTL_ERR_INVALID;
        }
        if (int jrc = join_lanes(r, stream); jrc != PTL_OK) return jrc;
        // runs of consecutive slices staged with the same kernel go out as one launch each, on the kernel they were staged with -- one launch
        // for all n unless a rebuild fell between two stage calls (then the earlier slices keep the state THEIR kernel has compiled in)
        int rc = PTL_OK;
        float total_ms = 0.0f;
        for (int j0 = 0; j0 < n && rc == PTL_OK;) {
            ptl_kernel* k = r->staged_kernels[j0];
            int j1 = j0 + 1;
            while (j1 < n && r->staged_kernels[j1] == k) ++j1;
            for (int j = j0; j < j1 && rc == PTL_OK; ++j) rc = ptl_kernel_stage_slice_from(k, j - j0, r->staged_blocks[j].data(), r->staged_blocks[j].size());
            float ms = 0.0f;
            void* out8 = out_rgba8 ? static_cast<unsigned char*>(out_rgba8) + (size_t)j0 * slice_pixels * 4 : nullptr;
            void* out32 = out_rgba32f ? static_cast<float*>(out_rgba32f) + (size_t)j0 * slice_pixels * 4 : nullptr;
            if (rc == PTL_OK) rc = ptl_kernel_render_slices(k, frame, j1 - j0, out8, out32, slice_pixels, stream, elapsed_ms ? &ms : nullptr);
            total_ms += ms;
            j0 = j1;
        }
        if (elapsed_ms) *elapsed_ms = total_ms;
        drop_staged_slices(r);
        return rc;
    });
}

extern "C" int ptl_renderer_join(ptl_renderer* r, void* stream) {
    if (!r) return PTL_ERR_INVALID;
    return guarded([&] { return join_lanes(r, stream); });
}

// The dynamic belt behind `snippets_keep_rays_affine` (VERDICT r5 #2c).  A sibling renderer of the same scene handle and the same specialisation is
// built with PTL_FLAG_CHECK_AFFINE (general products; the assumption sites count what arrives with another w), gets this renderer's camera and
// options, and draws the current state once at width x height.  A count above zero means: a kernel that ASSUMES w = 1 / 0 would have computed with
// other values for this very state -- the renderer then switches affine rays off for its stage and rebuilds (one re-JIT, the frames stay right).
static int check_affine_now(ptl_renderer* r, int width, int height, unsigned long long* violations) {
    if (r->device < 0) return PTL_ERR_NO_DEVICE;
    const unsigned flags = (r->flags & ~(kAsyncRejit | (1u << 22) | (1u << 23))) | (1u << 25) | 2u | 262144u;  // not async, no slices entry; checking + counting, quick JIT
    std::vector<char> log(1 << 16);
    ptl_renderer* sib = nullptr;
    int rc = ptl_renderer_create(r->owner, r->device, r->asset_root.c_str(), flags, &sib, log.data(), log.size());
    if (rc != PTL_OK) return rc;
    sib->cam = r->cam;
    sib->offset_after_material = r->offset_after_material;
    sib->gray_t_start = r->gray_t_start;
    sib->gray_t_size = r->gray_t_size;
    sib->render_depth = r->render_depth;
    sib->aa_count = r->aa_count;
    sib->aa_start = r->aa_start;
    sib->draw_side_by_side = r->draw_side_by_side;
    sib->draw_depth_map = r->draw_depth_map;
    sib->angle_color_disable = r->angle_color_disable;
    sib->grid_disable = r->grid_disable;
    sib->black_border_disable = r->black_border_disable;
    sib->darken_by_distance = r->darken_by_distance;
    sib->depth_map_min = r->depth_map_min;
    sib->depth_map_max = r->depth_map_max;
    sib->anaglyph_p = r->anaglyph_p;
    sib->anaglyph_q = r->anaglyph_q;
    sib->draw_anaglyph = r->draw_anaglyph;
    sib->anaglyph_mode = r->anaglyph_mode;
    sib->eye_distance = r->eye_distance;
    sib->swap_eyes = r->swap_eyes;
    sib->check_affine_on_new_source = false;
    ++sib->options_version;
    ptl_frame f{};
    f.width = width;
    f.height = height;
    f.rb_phase = 0;
    f.rb_stride = 1;
    std::vector<uint8_t> pixels((size_t)width * height * 4);
    uint64_t count = 0;
    rc = ptl_renderer_draw_to_host(sib, &f, pixels.data(), nullptr, &count, nullptr);
    ptl_renderer_destroy(sib);
    if (rc != PTL_OK) return rc;
    if (violations) *violations = count;
    r->affine_violations_seen = count;
    r->checked_source = r->kernel_source;
    if (count > 0 && r->affine_rays && !r->no_affine) {
        r->no_affine = r->no_affine_just_set = true;
        drop_async_kernels(r);
        rc = build_kernel(r, nullptr, 0);
        if (rc != PTL_OK) return rc;
        seed_async_kernels(r);
        ++r->rejit_count;
        r->checked_source = r->kernel_source;
        set_last_error("check_affine: " + std::to_string(count) + " ray halves met a product with a w that is not 1 / 0: affine rays switched off for this stage");
    }
    return PTL_OK;
}
extern "C" int ptl_renderer_check_affine(ptl_renderer* r, int width, int height, unsigned long long* violations) {
    if (!r || width < 1 || height < 1) return PTL_ERR_INVALID;
    return guarded([&] {
        ptl_frame f{};
        f.width = width;
        f.height = height;
        f.rb_stride = 1;
        int rc = prepare_draw(r, &f);  // (the state a draw would see: pending rebuilds done, so that `kernel_source` is what would be launched)
        if (rc < 0) return rc;
        return check_affine_now(r, width, height, violations);
    });
}
// (before a draw: option "check_affine" / PTL_CHECK_AFFINE=1 -- once per new source that has affine rays)
static int check_affine_if_asked(ptl_renderer* r, const ptl_frame* frame) {
    if (!r->check_affine_on_new_source || !r->affine_rays || r->no_affine || r->device < 0 || r->checked_source == r->kernel_source) return PTL_OK;
    int rc = check_affine_now(r, 64, 36, nullptr);
    if (rc != PTL_OK) return rc;
    return prepare_draw(r, frame);  // (the rebuilt kernel, if any, wants its uniforms)
}

extern "C" int ptl_renderer_draw(ptl_renderer* r, const ptl_frame* frame, void* out_rgba8, void* out_rgba32f, void* segments, void* stream,
                                 float* elapsed_ms) {
    if (!r || !frame) return PTL_ERR_INVALID;
    return guarded([&] {
        int rc = prepare_draw(r, frame);
        if (rc < 0) return rc;
        if (rc = check_affine_if_asked(r, frame); rc < 0) return rc;
        if (r->concurrent > 1 && r->device >= 0 && !elapsed_ms && !segments) return draw_on_a_lane(r, frame, out_rgba8, out_rgba32f, stream);
        // (a timed or counting draw, and every draw of a renderer without lanes: on the caller's stream, behind what the lanes still hold)
        if (int jrc = join_lanes(r, stream); jrc != PTL_OK) return jrc;
        return ptl_kernel_render(r->kernel, frame, out_rgba8, out_rgba32f, segments, stream, elapsed_ms);
    });
}
extern "C" int ptl_renderer_draw_to_host(ptl_renderer* r, const ptl_frame* frame, uint8_t* host_rgba8, float* host_rgba32f,
                                         uint64_t* host_segments, float* elapsed_ms) {
    if (!r || !frame) return PTL_ERR_INVALID;
    return guarded([&] {
        int rc = prepare_draw(r, frame);
        if (rc < 0) return rc;
        if (rc = check_affine_if_asked(r, frame); rc < 0) return rc;
        wait_for_lanes(r);
        return ptl_kernel_render_to_host(r->kernel, frame, host_rgba8, host_rgba32f, host_segments, elapsed_ms);
    });
}
extern "C" int ptl_renderer_prebuild_teleport(ptl_renderer* r) {
    if (!r || !r->kernel) return PTL_ERR_INVALID;
    return guarded([&] { return ptl_kernel_prebuild_teleport(r->kernel); });
}
extern "C" int ptl_renderer_teleport_ray(ptl_renderer* r, const double a[3], const double b[3], double out_pos[3], int* hit_object,
                                         int* changed_subspace, int* teleported) {
    if (!r || !a || !b) return PTL_ERR_INVALID;
    return guarded([&] {
        ptl_frame zero{0, 0, 0, 1, 0};  // the reference calls self.set_uniforms(0., 0.) here
        int rc = prepare_draw(r, &zero);
        if (rc < 0) return rc;
        wait_for_lanes(r);  // (the query runs on the primary kernel's block, on the default stream)
        int one = 1;
        ptl_kernel_set_uniform(r->kernel, "teleport_light_u", PTL_I32, &one);  // src/main.rs:1367 (a scene without it: no-op)
        float fa[3] = {(float)a[0], (float)a[1], (float)a[2]}, fb[3] = {(float)b[0], (float)b[1], (float)b[2]}, pos[3] = {0, 0, 0};
        rc = ptl_kernel_teleport_ray(r->kernel, fa, fb, pos, hit_object, changed_subspace, teleported);
        r->uploaded_scene = 0;  // teleport_light_u was overridden: the next draw re-uploads the scene values
        r->uploaded_w = -1;
        if (rc == PTL_OK && out_pos)
            for (int k = 0; k < 3; ++k) out_pos[k] = (double)pos[k];
        return rc;
    });
}

namespace {

// SceneRenderer::teleport_external_ray as Option<DVec3> + flags
struct RayQuery {
    bool teleported = false, hit_object = false, changed_subspace = false;
    DVec3 pos;
};
int query_ray(ptl_renderer* r, const DVec3& a, const DVec3& b, RayQuery* q) {
    double pa[3] = {a.x, a.y, a.z}, pb[3] = {b.x, b.y, b.z}, out[3] = {0, 0, 0};
    int hit = 0, sub = 0, tel = 0;
    int rc = ptl_renderer_teleport_ray(r, pa, pb, out, &hit, &sub, &tel);
    if (rc != PTL_OK) return rc;
    q->teleported = tel != 0;
    q->hit_object = hit != 0;
    q->changed_subspace = sub != 0;
    q->pos = DVec3(out[0], out[1], out[2]);
    return PTL_OK;
}

// SceneRenderer::teleport_matrix (src/main.rs:1174-1215): finite-difference Jacobian of the portal map
// around the camera, three more ray queries with +-dx offsets along the camera's axes.
int teleport_matrix(ptl_renderer* r, const DMat4& matrix, const DVec3& start_pos, const DVec3& direction_pos, const DVec3& actual, double dx,
                    bool* ok, DMat4* out) {
    *ok = false;
    DVec4 cols[3];
    const DVec4 axes[3] = {DVec4(1, 0, 0, 0), DVec4(0, 1, 0, 0), DVec4(0, 0, 1, 0)};
    for (int k = 0; k < 3; ++k) {
        DVec4 v4 = matrix.mul_vec4(axes[k]) * dx;
        DVec3 v(v4.x, v4.y, v4.z);
        RayQuery q;
        int rc = query_ray(r, start_pos + v, direction_pos + v, &q);
        if (rc != PTL_OK) return rc;
        if (!q.teleported) return PTL_OK;  // `?` on None
        DVec3 d = q.pos - actual;
        cols[k] = DVec4(d.x / dx, d.y / dx, d.z / dx, 0.0);  // DVec4::from((i, 0.)) / dx
    }
    DMat4 new_mat = DMat4::from_cols(cols[0], cols[1], cols[2], DVec4(0, 0, 0, 1));
    DVec4 moved = (new_mat * matrix.inverse()).mul_vec4(DVec4(direction_pos.x, direction_pos.y, direction_pos.z, 1.0));
    DVec3 pos = actual - DVec3(moved.x, moved.y, moved.z);
    *out = DMat4::from_cols(cols[0], cols[1], cols[2], DVec4(pos.x, pos.y, pos.z, 1.0));
    *ok = true;
    return PTL_OK;
}

// SceneRenderer::teleport_camera (src/main.rs:1217-1264)
int teleport_camera(ptl_renderer* r, const Camera& prev_cam, int* teleported, int* blocked) {
    Camera& cam = r->cam;
    if (cam.do_not_teleport_one_frame) {
        cam.do_not_teleport_one_frame = false;
        cam.prev_cam_pos = cam_pos(cam);
        return PTL_OK;
    }
    if (!(cam.allow_teleport || cam.stop_at_objects)) return PTL_OK;
    DVec3 pos = cam_pos(cam);
    RayQuery q;
    int rc = query_ray(r, cam.prev_cam_pos, pos, &q);
    if (rc != PTL_OK) return rc;
    if (cam.stop_at_objects && q.hit_object) {
        cam = prev_cam;
        if (blocked) *blocked = 1;
        return PTL_OK;
    }
    if (!q.teleported) {
        cam.prev_cam_pos = pos;
        return PTL_OK;
    }
    if (!cam.allow_teleport) return PTL_OK;
    for (double dx : {0.001, 0.0001, 0.00001, 0.000001}) {
        bool ok = false;
        DMat4 m;
        rc = teleport_matrix(r, cam.teleport_matrix, cam.prev_cam_pos, pos, q.pos, dx, &ok, &m);
        if (rc != PTL_OK) return rc;
        if (!ok) continue;
        cam.teleport_matrix = m;
        if (q.changed_subspace) cam.in_subspace = !cam.in_subspace;
        cam.prev_cam_pos = cam_pos(cam);
        if (teleported) *teleported = 1;
        return PTL_OK;
    }
    cam = prev_cam;  // no step size produced a Jacobian: stay where we were
    if (blocked) *blocked = 1;
    return PTL_OK;
}

// SceneRenderer::teleport_eye_matrices (src/main.rs:1121-1172): each eye sits eye_distance to the side of the camera; if
// the segment camera -> eye crosses a portal, the eye gets its own teleported matrix (and subspace flag).
int teleport_eye_matrices(ptl_renderer* r) {
    Camera& cam = r->cam;
    if (!((r->draw_anaglyph || r->draw_side_by_side) && cam.allow_teleport)) return PTL_OK;
    double eye_distance = r->swap_eyes ? -r->eye_distance : r->eye_distance;
    auto one_eye = [&](double x, DMat4* out_m, bool* out_sub) -> int {
        DVec3 start_pos = cam_pos(cam);
        DMat4 m = cam.matrix();
        DVec4 d4 = m.mul_vec4(DVec4(x, 0.0, 0.0, 1.0));
        DVec3 direction_pos(d4.x, d4.y, d4.z);
        DVec3 shift = direction_pos - start_pos;
        DMat4 translation = DMat4::from_cols({1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {shift.x, shift.y, shift.z, 1});
        *out_m = translation * m;
        *out_sub = cam.in_subspace;
        RayQuery q;
        int rc = query_ray(r, start_pos, direction_pos, &q);
        if (rc != PTL_OK) return rc;
        if (!q.teleported) return PTL_OK;
        for (double dx : {0.001, 0.0001, 0.00001, 0.000001}) {
            bool ok = false;
            DMat4 tm;
            rc = teleport_matrix(r, *out_m, start_pos, direction_pos, q.pos, dx, &ok, &tm);
            if (rc != PTL_OK) return rc;
            if (!ok) continue;
            *out_m = tm;
            if (q.changed_subspace) *out_sub = !cam.in_subspace;
            break;
        }
        return PTL_OK;
    };
    int rc = one_eye(-eye_distance, &cam.left_eye_matrix, &cam.left_eye_in_subspace);
    if (rc == PTL_OK) rc = one_eye(eye_distance, &cam.right_eye_matrix, &cam.right_eye_in_subspace);
    return rc;
}

}  // namespace

extern "C" int ptl_renderer_move_camera(ptl_renderer* r, const double look_at[3], double alpha, double beta, double radius, int* teleported,
                                        int* blocked) {
    if (!r || !look_at) return PTL_ERR_INVALID;
    if (teleported) *teleported = 0;
    if (blocked) *blocked = 0;
    return guarded([&] {
        Camera prev = r->cam;
        r->cam.look_at = DVec3(look_at[0], look_at[1], look_at[2]);
        r->cam.alpha = alpha;
        r->cam.beta = beta;
        r->cam.r = radius;
        ++r->options_version;
        int rc = teleport_camera(r, prev, teleported, blocked);
        if (rc == PTL_OK) rc = teleport_eye_matrices(r);
        ++r->options_version;
        return rc;
    });
}

// SceneRenderer::update (src/main.rs:1430-1538): the per-frame step of the video pipeline and of render-frame
extern "C" int ptl_renderer_update(ptl_renderer* r, double seconds, int* teleported, int* blocked) {
    if (!r) return PTL_ERR_INVALID;
    if (teleported) *teleported = 0;
    if (blocked) *blocked = 0;
    return guarded([&] {
        Scene& scene = *r->scene;
        Camera& cam = r->cam;
        if (!r->has_prev_cam) {  // SceneRenderer::new: prev_cam = cam.clone()
            r->prev_cam = cam;
            r->has_prev_cam = true;
        }
        ++r->options_version;
        std::optional<CalculatedCam> override_cam = scene.update(seconds);
        send_camera_matrix(r);

        int current_cam = scene.current_cam;
        if (cam.from != current_cam) {
            CalculatedCam c;
            if (current_cam >= 0) {
                if (cam.from < 0) r->original_cam = calculated_of(cam);
                auto got = scene.calculated_cam(scene.cameras.at(current_cam));
                if (!got) throw SceneError("camera can't be evaluated");
                c = *got;
            } else {
                c = r->original_cam;
            }
            cam.from = current_cam;
            cam.alpha = c.alpha;
            cam.beta = c.beta;
            cam.r = c.r;
            cam.look_at = c.look_at;
            cam.teleport_matrix = c.matrix;
            cam.in_subspace = c.in_subspace;
            cam.free_movement = c.free_movement;
            if (cam.free_movement) cam.look_at = cam.pos_vec() + cam.look_at;
            cam.do_not_teleport_one_frame = true;
        } else if (cam.from >= 0) {
            auto got = scene.calculated_cam(scene.cameras.at(cam.from));
            if (!got) throw SceneError("camera can't be evaluated");
            if (!cam.free_movement) cam.look_at = got->look_at;
        }

        if (override_cam) {
            cam.alpha = override_cam->alpha;
            cam.beta = override_cam->beta;
            cam.r = override_cam->r;
            cam.look_at = override_cam->look_at;
            cam.free_movement = override_cam->free_movement;
            if (override_cam->override_matrix) {
                cam.teleport_matrix = override_cam->matrix;
                cam.in_subspace = override_cam->in_subspace;
                cam.do_not_teleport_one_frame = true;
            }
        }

        int rc = PTL_OK;
        if (!same_matrix(cam.matrix(), r->prev_cam.matrix())) {
            Camera prev = r->prev_cam;
            rc = teleport_camera(r, prev, teleported, blocked);
        }
        if (rc == PTL_OK) rc = teleport_eye_matrices(r);
        r->prev_cam = cam;
        send_camera_matrix(r);
        ++r->options_version;
        if (rc == PTL_OK) rc = update_videos(r);
        return rc;
    });
}

extern "C" int ptl_renderer_camera_state(ptl_renderer* r, double teleport16[16], int* in_subspace, double position[3]) {
    if (!r) return PTL_ERR_INVALID;
    if (teleport16)
        for (int k = 0; k < 4; ++k) {
            teleport16[4 * k + 0] = r->cam.teleport_matrix.c[k].x;
            teleport16[4 * k + 1] = r->cam.teleport_matrix.c[k].y;
            teleport16[4 * k + 2] = r->cam.teleport_matrix.c[k].z;
            teleport16[4 * k + 3] = r->cam.teleport_matrix.c[k].w;
        }
    if (in_subspace) *in_subspace = r->cam.in_subspace ? 1 : 0;
    if (position) {
        DVec3 p = cam_pos(r->cam);
        position[0] = p.x;
        position[1] = p.y;
        position[2] = p.z;
    }
    return PTL_OK;
}

extern "C" ptl_kernel* ptl_renderer_kernel(ptl_renderer* r) {
    if (!r) return nullptr;
    // the kernel the next draw would use: a specialised build follows the mode switches (also on a handle without a device, which never draws)
    const bool camera_left_the_affine_maps = r->affine_rays && !r->no_affine && !camera_is_affine(*r);
    if (camera_left_the_affine_maps && !((r->flags & kAsyncRejit) != 0 && r->device >= 0)) r->no_affine = r->no_affine_just_set = true;
    if ((r->flags & kSpecialised) != 0 && !((r->flags & kAsyncRejit) != 0 && r->device >= 0) && (mode_switches(*r) != r->kernel_switches || camera_left_the_affine_maps)) {
        int rc = guarded([&] {
            int rc2 = build_kernel(r, nullptr, 0);
            if (rc2 == PTL_OK) ++r->rejit_count;
            return rc2;
        });
        if (rc != PTL_OK) return nullptr;  // ptl_last_error() says why; the old kernel has other switches compiled in
    }
    return r->kernel;
}
extern "C" int ptl_renderer_kernel_source(ptl_renderer* r, char** source) {
    if (!r || !source) return PTL_ERR_INVALID;
    *source = (char*)std::malloc(r->kernel_source.size() + 1);
    if (!*source) return PTL_ERR_INVALID;
    std::memcpy(*source, r->kernel_source.c_str(), r->kernel_source.size() + 1);
    return PTL_OK;
}
// ---- the binary64 primitives behind the scene's constants, one by one (test hooks: tests/test_matrix_exact.py checks each against exact
// arithmetic).  Matrices are 16 doubles, column-major like glam's to_cols_array.
extern "C" int ptl_dmath(const char* op, const double* a, const double* b, const double* c, double* out) {
    if (!op || !a || !out) return PTL_ERR_INVALID;
    auto load = [](const double* v) {
        return DMat4::from_cols({v[0], v[1], v[2], v[3]}, {v[4], v[5], v[6], v[7]}, {v[8], v[9], v[10], v[11]}, {v[12], v[13], v[14], v[15]});
    };
    auto store = [&](const DMat4& m) {
        for (int k = 0; k < 4; ++k) {
            out[4 * k + 0] = m.c[k].x;
            out[4 * k + 1] = m.c[k].y;
            out[4 * k + 2] = m.c[k].z;
            out[4 * k + 3] = m.c[k].w;
        }
        return (int)PTL_OK;
    };
    const std::string what = op;
    if (what == "inverse") return store(load(a).inverse());  // glam DMat4::inverse (src/gui/scene.rs:587-588, matrix.rs:537-547)
    if (what == "mul" && b) return store(load(a) * load(b));
    if (what == "teleport" && b) return store(load(b) * load(a).inverse());  // a_to_b = B * A^-1 (src/gui/scene.rs:624-632)
    if (what == "srt" && b && c)  // Simple / Parametrized: T * (Rx * Ry * Rz) * S (src/gui/matrix.rs:555-569); a = scale xyz, b = rotate xyz, c = offset xyz
        return store(DMat4::from_scale_rotation_translation(DVec3(a[0], a[1], a[2]), DQuat::rotation_x(b[0]) * DQuat::rotation_y(b[1]) * DQuat::rotation_z(b[2]), DVec3(c[0], c[1], c[2])));
    if (what == "lerp" && b && c) {  // Matrix::Lerp (src/gui/matrix.rs:614-627): a = first, b = second, c[0] = t -- the very statements scene.cpp evaluates
        DVec3 fs, ft, ss, st;
        DQuat fr, sr;
        load(a).to_scale_rotation_translation(&fs, &fr, &ft);
        load(b).to_scale_rotation_translation(&ss, &sr, &st);
        return store(DMat4::from_scale_rotation_translation(fs.lerp(ss, c[0]), fr.lerp(sr, c[0]), ft.lerp(st, c[0])));
    }
    if (what == "camera" && b) {  // RotateAroundCam::get_matrix (src/main.rs:278-304): a = look_at xyz, alpha, beta, r; b = the teleport matrix
        Camera cam;
        cam.look_at = DVec3(a[0], a[1], a[2]);
        cam.alpha = a[3];
        cam.beta = a[4];
        cam.r = a[5];
        cam.teleport_matrix = load(b);
        return store(cam.matrix());
    }
    set_last_error(std::string("ptl_dmath: unknown operation `") + what + "`");
    return PTL_ERR_INVALID;
}

extern "C" int ptl_renderer_rejit_count(ptl_renderer* r) { return r ? r->rejit_count : -1; }
extern "C" int ptl_renderer_affine_rays(ptl_renderer* r) { return r ? (r->affine_rays ? 1 : 0) : -1; }
extern "C" int ptl_snippets_keep_rays_affine(const char* glsl, char* why, size_t why_cap) {
    if (!glsl) return -1;
    try {
        std::string reason;
        const bool ok = snippets_keep_rays_affine({glsl}, &reason);
        if (why && why_cap) {
            std::snprintf(why, why_cap, "%s", ok ? "" : reason.c_str());
        }
        return ok ? 1 : 0;
    } catch (const std::exception& e) {
        set_last_error(std::string("ptl_snippets_keep_rays_affine: ") + e.what());
        return -1;
    }
}
extern "C" int ptl_renderer_rejit_pending(ptl_renderer* r) {
    if (!r) return -1;
    return (r->job || (r->spec_kernel != nullptr && r->kernel != r->spec_kernel)) ? 1 : 0;
}
extern "C" void ptl_renderer_destroy(ptl_renderer* r) {
    if (!r) return;
    if (r->job && r->job->worker.joinable()) r->job->worker.join();  // (the worker owns nothing of ours, but a thread must be joined)
    drop_lane_clones(r);
    drop_staged_slices(r);
    for (auto& l : r->lanes)
        if (l.done) ptl_event_destroy(l.done);  // (the lanes' streams belong to the process-wide pool)
    if (r->fence) ptl_event_destroy(r->fence);
    if (r->spec_kernel || r->dyn_kernel) {  // background re-JIT: `kernel` is one of these two
        ptl_kernel_destroy(r->spec_kernel);
        ptl_kernel_destroy(r->dyn_kernel);
    } else {
        ptl_kernel_destroy(r->kernel);
    }
    delete r;
}

// ---- template engine hooks ----------------------------------------------------------------------
struct ptl_strstore {
    StringStorage s;
};
extern "C" ptl_strstore* ptl_strstore_new(void) { return new ptl_strstore(); }
extern "C" void ptl_strstore_free(ptl_strstore* s) { delete s; }
extern "C" void ptl_strstore_add_string(ptl_strstore* s, const char* text) {
    if (s && text) s->s.add_string(text);
}
extern "C" void ptl_strstore_add_identifier_string(ptl_strstore* s, const char* kind, const char* name, const char* text) {
    if (s && kind && name && text) s->s.add_identifier_string({kind, name}, text);
}
extern "C" ptl_strstore* ptl_apply_template(const char* tmpl, const char* const* slot_names, ptl_strstore* const* storages, int n) {
    std::map<std::string, StringStorage> m;
    for (int k = 0; k < n; ++k) {
        m[slot_names[k]] = std::move(storages[k]->s);
        delete storages[k];
    }
    try {
        auto* out = new ptl_strstore();
        out->s = apply_template(tmpl, std::move(m));
        return out;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return nullptr;
    }
}
extern "C" const char* ptl_strstore_text(const ptl_strstore* s) { return s ? s->s.storage.c_str() : ""; }
extern "C" int ptl_strstore_current_line(const ptl_strstore* s) { return s ? s->s.current_line_no : 0; }
extern "C" int ptl_strstore_range(const ptl_strstore* s, const char* kind, const char* name, int* start, int* end) {
    if (!s) return PTL_ERR_INVALID;
    auto it = s->s.line_numbers.ranges.find(ElementKey{kind, name});
    if (it == s->s.line_numbers.ranges.end()) return 1;
    if (start) *start = it->second.start;
    if (end) *end = it->second.end;
    return PTL_OK;
}
extern "C" int ptl_strstore_get_identifier(const ptl_strstore* s, int line, char* kind, size_t kind_cap, char* name, size_t name_cap,
                                           int* local_line) {
    if (!s) return PTL_ERR_INVALID;
    ElementKey key;
    int local = 0;
    if (!s->s.line_numbers.get_identifier(line, &key, &local)) return 1;
    copy_str(kind, kind_cap, key.kind);
    copy_str(name, name_cap, key.name);
    if (local_line) *local_line = local;
    return PTL_OK;
}

extern "C" const char* ptl_device_source(const char* which) {
    if (!which) return nullptr;
    std::string w = which;
    if (w == "glsl") return device_source_glsl();
    if (w == "library") return device_source_library();
    if (w == "trace") return device_source_trace_template();
    if (w == "entry") return device_source_entry();
    return nullptr;
}

extern "C" char* ptl_translate_glsl(const char* glsl) {
    if (!glsl) return nullptr;
    std::string out;
    try {
        out = translate_glsl(glsl);
    } catch (const std::exception& e) {  // e.g. a struct field that spells a swizzle: NULL + ptl_last_error()
        set_last_error(e.what());
        return nullptr;
    }
    char* p = (char*)std::malloc(out.size() + 1);
    std::memcpy(p, out.c_str(), out.size() + 1);
    return p;
}

extern "C" char* ptl_translate_library_glsl(const char* glsl) {  // a file-scope library text: function definitions get PTL_FN
    if (!glsl) return nullptr;
    std::string out;
    try {
        out = translate_glsl(glsl, true, true);
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return nullptr;
    }
    char* p = (char*)std::malloc(out.size() + 1);
    std::memcpy(p, out.c_str(), out.size() + 1);
    return p;
}

extern "C" char* ptl_bound_glsl(const char* glsl_body, const char* out_functions, int* bounded) {
    if (!glsl_body) return nullptr;
    std::set<std::string> with_out;
    std::string cur;
    for (const char* c = out_functions ? out_functions : ""; ; ++c) {
        if (*c == ',' || *c == '\0') {
            if (!cur.empty()) with_out.insert(cur);
            cur.clear();
            if (*c == '\0') break;
        } else {
            cur += *c;
        }
    }
    try {
        std::string out = bound_nearer_blocks(glsl_body, with_out, bounded);
        char* p = (char*)std::malloc(out.size() + 1);
        if (!p) {
            set_last_error("ptl_bound_glsl: out of memory");
            return nullptr;
        }
        std::memcpy(p, out.c_str(), out.size() + 1);
        return p;
    } catch (const std::exception& e) {  // malformed input (the tokenizer throws): an error, never an exception across the C boundary
        set_last_error(std::string("ptl_bound_glsl: ") + e.what());
        return nullptr;
    }
}

extern "C" char* ptl_hoist_glsl(const char* glsl, const char* uniforms, const char* out_functions, int body_only, const char* params, char** prologue) {
    if (!glsl) return nullptr;
    auto split = [](const char* text) {
        std::vector<std::string> parts;
        std::string cur;
        for (const char* c = text ? text : ""; *c; ++c) {
            if (*c == ';') {
                if (!cur.empty()) parts.push_back(cur);
                cur.clear();
            } else {
                cur += *c;
            }
        }
        if (!cur.empty()) parts.push_back(cur);
        return parts;
    };
    HoistParams hp;
    for (const std::string& u : split(uniforms)) {
        size_t sp = u.find(' ');
        if (sp != std::string::npos) hp.uniforms[u.substr(sp + 1)] = u.substr(0, sp);
    }
    for (const std::string& f : split(out_functions)) {  // "name" may write through an argument; "=name" is merely defined by the scene
        if (f[0] == '=') hp.scene_functions.insert(f.substr(1));
        else hp.functions_with_out_params.insert(f);
    }
    hp.body_only = body_only != 0;
    for (const std::string& name : split(params)) {  // "@r": a ray parameter whose origin is the camera's (first-trip variant)
        if (name[0] == '@') {
            hp.origin_uniform_rays.push_back(name.substr(1));
            hp.body_params.push_back(name.substr(1));
        } else {
            hp.body_params.push_back(name);
        }
    }
    hp.origin_expr = "PTL_DV_OUT.ptl_dv_origin";
    int counter = 0;
    HoistResult r = hoist_uniform_work(glsl, hp, counter);
    if (prologue) {
        std::string text;
        for (auto& m : r.members) text += "// member: " + m.type + " " + m.name + (m.length ? "[" + std::to_string(m.length) + "]" : "") + "\n";
        text += r.prologue;
        *prologue = strdup(text.c_str());
    }
    return strdup(r.glsl.c_str());
}

extern "C" int ptl_scene_to_ron(ptl_scene* s, char** text) {
    if (!s || !text) return PTL_ERR_INVALID;
    return guarded([&] {
        *text = strdup(s->scene->to_ron().c_str());
        return PTL_OK;
    });
}

extern "C" char* ptl_ron_format(const char* text) {
    if (!text) return nullptr;
    try {
        std::string out = ron::to_string(ron::parse(text));
        return strdup(out.c_str());
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return nullptr;
    }
}

extern "C" int ptl_formula_eval(const char* text, const char* const* names, const double* values, int n, double time, double* out) {
    if (!text || !out) return PTL_ERR_INVALID;
    std::string err;
    auto f = Formula::compile(text, &err);
    if (!f) {
        set_last_error(err);
        return 1;
    }
    FormulaNamespace ns = [&](const std::string& name, const std::vector<double>& args) -> std::optional<double> {
        bool known = false;
        auto r = formula_custom_function(name, args, &known);
        if (known) return r;
        if (name == "time" || name == "total_time") return time;
        for (int k = 0; k < n; ++k)
            if (name == names[k]) return values[k];
        return std::nullopt;
    };
    auto v = f->eval(ns);
    if (!v) return 1;
    *out = *v;
    return PTL_OK;
}
