// scene.h -- in-memory scene model and its constant evaluator.
//
// Host-side mirror of the reference's L4 "scene model + evaluator" for exactly the part the
// render path reads: src/gui/scene_serialized.rs:610-646 (on-disk schema),
// :1102-1477 (deserialize_scene_new_format), src/gui/uniform.rs:268-278,1009-1140
// (AnyUniform::get), src/gui/matrix.rs:16-65,510-631 (Matrix::get).  The reference keeps
// elements in Storage2<T> graphs keyed by UniqueId (src/gui/storage2.rs); here elements are
// plain vectors in declaration order and references are indices (-1 = None).  Inline
// (unnamed) matrices get the generated name `id<N>`, like src/gui/object.rs:188-193.
#pragma once
#include <array>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "dmath.h"
#include "formula.h"
#include "ron.h"

namespace ptl {

struct SceneError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// --- uniforms (src/gui/uniform.rs:268-278) ----------------------------------------------
struct Uniform {
    enum Kind { Bool, Int, Float, Angle, Progress, Formula, FormulaInt, Trefoil } kind = Float;
    bool b = false;
    int i = 0;
    double f = 0.0;
    std::string formula;
    int trefoil[18][3] = {};  // TrefoilSpecial: (enabled, value, color) x 18 (src/gui/uniform.rs:19-20)
};
struct UniformValue {
    enum Kind { Bool, Int, Float } kind = Float;
    bool b = false;
    int i = 0;
    double f = 0.0;
    double as_f64() const { return kind == Bool ? (b ? 1.0 : 0.0) : (kind == Int ? (double)i : f); }
};
struct UniformEntry {
    std::string name;  // empty for inline uniforms
    Uniform value;
};

// ParametrizeOrNot (src/gui/uniform.rs:330-334)
struct Param {
    bool is_uniform = false;
    double value = 0.0;
    int uniform = -1;
};

// --- matrices (src/gui/matrix.rs:16-65) ---------------------------------------------------
struct Matrix {
    enum Kind { Mul, Teleport, Simple, Parametrized, Exact, ExactFull, If, Sqrt, Lerp, Camera, Inv } kind = Simple;
    int a = -1, b = -1, c = -1;  // Mul{to=a, what=b}; Teleport{first=a, second=b, what=c}; If{then=a, otherwise=b};
                                 // Sqrt/Inv{a}; Lerp{first=a, second=b}
    DVec3 offset, rotate;
    double scale = 1.0;
    bool mirror[3] = {false, false, false};
    Param p[16];  // Parametrized: offset xyz [0..2], rotate xyz [3..5], mirror xyz [6..8], scale [9]
                  // Exact: i,j,k,pos xyz [0..11]; ExactFull: c0..c3 xyzw [0..15]
    Param cond;   // If.condition / Lerp.t
};
struct MatrixEntry {
    std::string name;  // generated `id<N>` for inline matrices
    bool named = false;
    Matrix value;
};

// --- objects / materials (src/gui/object.rs:32-63, src/gui/material.rs:14-38) -------------
enum class Subspace { Normal, Subspace, Both };
struct Object {
    enum Kind { DebugMatrix, Flat, Complex } kind = Flat;
    bool portal = false;
    int m0 = -1, m1 = -1;  // Simple(m0) / Portal(m0, m1) / DebugMatrix(m0)
    std::string code;      // is_inside (Flat) or intersect (Complex) snippet
    Subspace in_subspace = Subspace::Normal;
    std::string name;
};
struct Material {
    enum Kind { Simple, Reflect, Refract, Complex } kind = Simple;
    double color[3] = {0.5, 0.2, 0.2};  // Simple.color / add_to_color
    double normal_coef = 0.5, grid_scale = 4.0, grid_coef = 0.3, refractive_index = 1.0;
    bool grid = true, grid2 = false, grid3 = false;
    std::string code;
    std::string name;
};
struct NamedCode {
    std::string name, code;
};
struct Texture {
    std::string name, path;
};

// Named cameras (src/gui/camera.rs:46-66) and stages (src/gui/animation.rs:29-60,171-183,217-237)
struct SceneCamera {
    std::string name;  // empty for a camera defined inline in a stage
    bool look_at_matrix = false;  // CamLookAt::MatrixCenter(id) vs Coordinate(pos)
    int matrix = -1;
    DVec3 coordinate;
    double alpha = 0.0, beta = 0.0, r = 3.5;
    bool in_subspace = false, free_movement = false;
    DMat4 teleport = DMat4::identity();  // Cam::matrix
};
struct StageChange {
    enum Kind { ProvidedToUser, FromDev, Changed, ChangedAndToUser } kind = FromDev;
    int ref = -1;  // Changed*/: index of the replacing uniform / matrix (-1 = Changed(None))
};
struct AnimationStage {
    std::string name;
    std::vector<std::pair<std::string, StageChange>> uniforms, matrices;  // by element name
    bool has_set_cam = false;  // set_cam: Some(..)
    int set_cam = -1;          // camera index, -1 = original camera
};

// Easing (src/gui/easing.rs:6-101)
enum class Easing { Linear, In, Out, InOut, InOutFast, ElasticOut };
double ease(Easing e, double t);

// CurrentStage (src/gui/scene.rs:60-66); `index` into stages / animations
struct StageRef {
    enum Kind { Dev, Animation, RealAnimation } kind = Dev;
    int index = -1;
    bool operator==(const StageRef& o) const { return kind == o.kind && (kind == Dev || index == o.index); }
};

// RealAnimation (src/gui/animation.rs:1014-1043): one clip of the video pipeline
struct RealAnimation {
    std::string name;
    double duration = 0.0;
    StageRef base;                                                 // animation_stage
    std::vector<std::pair<std::string, int>> uniforms, matrices;  // the Changed(Some(ref)) entries; CopyPrev / Changed(None) leave the base stage's value
    bool use_prev_cam = false, use_start_cam_as_end = false;
    int cam_start = -1, cam_end = -1;  // camera indices
    std::optional<bool> use_any_cam_as_start, use_any_cam_as_end;
    int cam_any_start = -1, cam_any_end = -1;  // animation indices
    Easing cam_easing = Easing::Linear;
    bool has_easing_uniform = false;  // cam_easing_uniform: Some(Some(id))
    int easing_uniform = -1;
};

// CalculatedCam (src/gui/camera.rs:22-32)
struct CalculatedCam {
    DVec3 look_at;
    double alpha = 0.0, beta = 0.0, r = 3.5;
    bool free_movement = false, in_subspace = false;
    DMat4 matrix = DMat4::identity();
    bool override_matrix = true;
};

// Scene `cam` block (src/gui/scene.rs:33-52)
struct CamSettings {
    DVec3 look_at;
    double alpha = 0.0, beta = 0.0, r = 3.5, offset_after_material = 0.000025;
};

class Scene {
public:
    static std::shared_ptr<Scene> from_ron_text(const std::string& text);
    static std::shared_ptr<Scene> from_file(const std::string& path);

    CamSettings cam;
    std::vector<UniformEntry> uniforms;
    std::vector<MatrixEntry> matrices;
    std::vector<Object> objects;
    std::vector<Material> materials;
    std::vector<NamedCode> intersection_materials;
    std::vector<NamedCode> library;
    std::vector<Texture> textures;
    std::vector<std::string> videos;  // names: a video is one more sampler (src/gui/scene.rs:405-409)
    // Video (src/gui/video.rs, VideoSer scene_serialized.rs:51-55): frames are the PNG files of video_png/<stem of path>/, the one
    // shown is picked by a uniform in [0, 1] (VideoRuntime, src/main.rs:771-925)
    struct Video {
        std::string name, path;
        int uniform = -1;
    };
    std::vector<Video> video_sources;
    std::optional<std::string> skybox;
    bool use_time = false;
    std::vector<SceneCamera> cameras;
    std::vector<AnimationStage> stages;
    std::vector<RealAnimation> animations;
    StageRef current_stage;
    int current_cam = -1;        // egui memory "CurrentCam": camera the current stage selects, -1 = original camera
    bool run_animations = false; // src/gui/scene.rs:115
    double prev_t_raw = 0.0;
    std::vector<std::pair<std::string, Uniform>> dev_uniforms;  // dev_stage: the values FromDev restores
    std::vector<std::pair<std::string, Matrix>> dev_matrices;
    // set_id aliases installed by a stage: element i evaluates as element alias[i] (-1: itself)
    std::vector<int> uniform_alias, matrix_alias;

    // bumped by every mutation that can change an evaluated value (renderers cache on it)
    unsigned long long version = 1;
    // formula time inputs (FormulasCache, src/gui/uniform.rs:625-697)
    double time = 0.0, total_time = 0.0;
    DMat4 camera_matrix = DMat4::identity();

    int find_uniform(const std::string& name) const;
    int find_matrix(const std::string& name) const;

    // The document the scene was read from, kept whole (descriptions, GUI-only blocks, everything this model does not
    // interpret) and edited in step with the model, so that to_ron() writes a file the reference loads back unchanged:
    // serialize_scene_new_format + ron::ser::to_string_pretty (src/gui/scene_serialized.rs:22-24,654-1100).
    ron::Value doc;
    std::string to_ron() const;
    void apply_stage_to_doc(StageRef stage);

    // AnyUniform::get / Matrix::get; nullopt = "can't be getted" in the reference
    std::optional<UniformValue> eval_uniform(int index) const;
    std::optional<DMat4> eval_matrix(int index) const;
    std::optional<double> eval_param(const Param& p) const;
    // Did any evaluation since the last call read a per-frame input (`time`, `total_time`, Matrix::Camera)?  Clears the mark.
    // (How the kernel specialiser tells clip-constant values from animated ones.)
    bool take_frame_input_mark() const {
        bool t = frame_input_read_;
        frame_input_read_ = false;
        return t;
    }
    // the element a uniform currently evaluates as (stage / clip replacements followed); nullptr if out of range
    const Uniform* resolved_uniform(int index) const;

    // overrides (what a stage / animation / user slider does): set the stored value
    bool set_uniform_value(const std::string& name, double v);
    // TrefoilSpecial::decode / encode (src/gui/uniform.rs:23-98): "1a 2a G,1b 3b B" = part 1a teleports to 2a with colour G ...
    // set_trefoil returns false for an unknown uniform, a uniform of another kind, or text that does not decode.
    static bool trefoil_decode(const std::string& text, int out[18][3]);
    static std::string trefoil_encode(const int parts[18][3]);
    bool set_trefoil(const std::string& name, const std::string& text);
    std::optional<std::string> get_trefoil(const std::string& name) const;

    // Scene::init_stage_by_name (src/gui/scene.rs:1237-1250): apply an animation stage's overrides.
    // Returns false if there is no such stage; *camera = index into `cameras` the stage selects, or -1.
    bool init_stage_by_name(const std::string& name, int* camera);
    int find_camera(const std::string& name) const;
    // Scene::init_stage (src/gui/scene.rs:1180-1236), all three kinds; sets current_stage and current_cam
    void init_stage(StageRef stage, int depth = 0);
    // Scene::init_animation_by_name / _by_position (src/gui/scene.rs:1254-1277)
    bool init_animation_by_name(const std::string& name);
    int find_animation(const std::string& name) const;
    // Scene::get_start_cam / get_end_cam (src/gui/scene.rs:1303-1344): camera index or -1
    int animation_start_cam(int animation, int depth = 0) const;
    int animation_end_cam(int animation, int depth = 0) const;
    double total_animation_duration() const;
    // Scene::update (src/gui/scene.rs:1353-1493): maps wall-clock seconds to the formulas' `time` /
    // `total_time`, and returns the interpolated "OverrideCam" of a real animation (if it has both cameras)
    std::optional<CalculatedCam> update(double seconds);
    // Cam::get (src/gui/camera.rs:125-140)
    std::optional<CalculatedCam> calculated_cam(const SceneCamera& c) const;
    // Cam::get_pos (src/gui/camera.rs:96-108)
    std::optional<DVec3> camera_look_at(const SceneCamera& c) const;

private:
    mutable std::map<std::string, std::shared_ptr<Formula>> formula_cache_;
    mutable std::vector<char> uniform_busy_, matrix_busy_;  // cycle guards (Storage2::get)
    mutable bool frame_input_read_ = false;
    std::optional<double> eval_formula(const std::string& text) const;
};

}  // namespace ptl
