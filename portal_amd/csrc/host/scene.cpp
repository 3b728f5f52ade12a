// scene.cpp -- RON document -> Scene, and evaluation of uniforms / matrices (see scene.h).
#include "scene.h"

#include <cstring>
#include <fstream>
#include <sstream>

namespace ptl {
namespace {

using ron::Value;

const Value& storage_list(const Value& doc, const char* field, bool required) {
    static const Value empty_list = [] {
        Value v;
        v.kind = Value::List;
        return v;
    }();
    const Value* f = doc.find(field);
    if (!f) {
        if (required) throw SceneError(std::string("scene: missing field `") + field + "`");
        return empty_list;
    }
    const Value& inner = f->unwrap_newtypes();  // SerStorage<T>(Vec<Named<T>>)
    if (inner.kind != Value::List) throw SceneError(std::string("scene: field `") + field + "` is not a list");
    return inner;
}

const std::string& as_string(const Value& v, const char* what) {
    const Value& s = v.unwrap_newtypes();
    if (s.kind != Value::String) throw SceneError(std::string("scene: expected string for ") + what);
    return s.s;
}
double as_f64(const Value& v, const char* what) {
    const Value& n = v.unwrap_newtypes();
    if (!n.is_number()) throw SceneError(std::string("scene: expected number for ") + what);
    return n.number();
}
bool as_bool(const Value& v, const char* what) {
    const Value& n = v.unwrap_newtypes();
    if (n.kind != Value::Bool) throw SceneError(std::string("scene: expected bool for ") + what);
    return n.b;
}
DVec3 as_dvec3(const Value& v, const char* what) {
    if (v.kind == Value::Tuple && v.items.size() == 3) return {as_f64(v.items[0], what), as_f64(v.items[1], what), as_f64(v.items[2], what)};
    if (v.kind == Value::Struct) return {as_f64(v.at("x"), what), as_f64(v.at("y"), what), as_f64(v.at("z"), what)};
    throw SceneError(std::string("scene: expected 3-vector for ") + what);
}

struct Loader {
    Scene& scene;
    std::map<std::string, int> matrix_by_name;

    Uniform parse_uniform(const Value& v) {
        Uniform u;
        if (v.kind != Value::Tuple && v.kind != Value::Struct) throw SceneError("scene: bad uniform value");
        const std::string& tag = v.s;
        if (tag == "Bool") {
            u.kind = Uniform::Bool;
            u.b = as_bool(v.items.at(0), "Bool uniform");
        } else if (tag == "Int") {
            u.kind = Uniform::Int;
            u.i = (int)as_f64(v.items.at(0).unwrap_newtypes().at("value"), "Int uniform");
        } else if (tag == "Float") {
            u.kind = Uniform::Float;
            u.f = as_f64(v.items.at(0).unwrap_newtypes().at("value"), "Float uniform");
        } else if (tag == "Angle") {
            u.kind = Uniform::Angle;
            u.f = as_f64(v.items.at(0), "Angle uniform");
        } else if (tag == "Progress") {
            u.kind = Uniform::Progress;
            u.f = as_f64(v.items.at(0), "Progress uniform");
        } else if (tag == "Formula" || tag == "FormulaInt") {
            u.kind = tag == "Formula" ? Uniform::Formula : Uniform::FormulaInt;
            u.formula = as_string(v.items.at(0), "Formula uniform");
        } else if (tag == "TrefoilSpecial") {
            u.kind = Uniform::Trefoil;
            const Value& arr = v.items.at(0).unwrap_newtypes();  // TrefoilSpecial(( ((b, v, c), ... x18) ))
            if (arr.kind != Value::Tuple || arr.items.size() != 18) throw SceneError("scene: TrefoilSpecial needs 18 entries");
            for (int k = 0; k < 18; ++k) {
                const Value& e = arr.items[k];
                if (e.items.size() != 3) throw SceneError("scene: bad TrefoilSpecial entry");
                u.trefoil[k][0] = as_bool(e.items[0], "trefoil enabled") ? 1 : 0;
                u.trefoil[k][1] = (int)as_f64(e.items[1], "trefoil value");
                u.trefoil[k][2] = (int)as_f64(e.items[2], "trefoil color");
            }
        } else {
            throw SceneError("scene: unknown uniform kind `" + tag + "`");
        }
        return u;
    }
    // Option<UniformRef> -> index
    int uniform_ref(const Value& opt) {
        const Value* r = opt.some();
        if (!r) return -1;
        if (r->is_named("Named")) return scene.find_uniform(as_string(r->items.at(0), "UniformRef"));
        if (r->is_named("Inline")) {
            scene.uniforms.push_back(UniformEntry{"", parse_uniform(r->items.at(0))});
            return (int)scene.uniforms.size() - 1;
        }
        throw SceneError("scene: bad UniformRef");
    }
    Param parse_param(const Value& v) {
        Param p;
        if (v.is_named("Value")) {
            p.value = as_f64(v.items.at(0), "ParametrizeOrNot::Value");
        } else if (v.is_named("Uniform")) {
            p.is_uniform = true;
            p.uniform = uniform_ref(v.items.at(0));
        } else {
            throw SceneError("scene: bad ParametrizeOrNot");
        }
        return p;
    }
    void parse_tvec(const Value& v, Param* out, int n) {
        static const char* names[4] = {"x", "y", "z", "w"};
        for (int k = 0; k < n; ++k) out[k] = parse_param(v.at(names[k]));
    }
    // Option<MatrixRef> -> index; inline matrices are appended depth-first
    int matrix_ref(const Value& opt) {
        const Value* r = opt.some();
        if (!r) return -1;
        if (r->is_named("Named")) {
            auto it = matrix_by_name.find(as_string(r->items.at(0), "MatrixRef"));
            return it == matrix_by_name.end() ? -1 : it->second;
        }
        if (r->is_named("Inline")) {
            int idx = (int)scene.matrices.size();
            scene.matrices.push_back(MatrixEntry{"id" + std::to_string(idx), false, Matrix{}});
            Matrix m = parse_matrix(r->items.at(0));
            scene.matrices[idx].value = m;
            return idx;
        }
        throw SceneError("scene: bad MatrixRef");
    }
    Matrix parse_matrix(const Value& v) {
        Matrix m;
        const std::string& tag = v.s;
        if (tag == "Mul") {
            m.kind = Matrix::Mul;
            m.a = matrix_ref(v.at("to"));
            m.b = matrix_ref(v.at("what"));
        } else if (tag == "Teleport") {
            m.kind = Matrix::Teleport;
            m.a = matrix_ref(v.at("first_portal"));
            m.b = matrix_ref(v.at("second_portal"));
            m.c = matrix_ref(v.at("what"));
        } else if (tag == "Simple") {
            m.kind = Matrix::Simple;
            m.offset = as_dvec3(v.at("offset"), "Simple.offset");
            m.scale = as_f64(v.at("scale"), "Simple.scale");
            m.rotate = as_dvec3(v.at("rotate"), "Simple.rotate");
            const Value& mir = v.at("mirror");
            for (int k = 0; k < 3; ++k) m.mirror[k] = as_bool(mir.items.at(k), "Simple.mirror");
        } else if (tag == "Parametrized") {
            m.kind = Matrix::Parametrized;
            parse_tvec(v.at("offset"), m.p + 0, 3);
            parse_tvec(v.at("rotate"), m.p + 3, 3);
            parse_tvec(v.at("mirror"), m.p + 6, 3);
            m.p[9] = parse_param(v.at("scale"));
        } else if (tag == "Exact") {
            m.kind = Matrix::Exact;
            parse_tvec(v.at("i"), m.p + 0, 3);
            parse_tvec(v.at("j"), m.p + 3, 3);
            parse_tvec(v.at("k"), m.p + 6, 3);
            parse_tvec(v.at("pos"), m.p + 9, 3);
        } else if (tag == "ExactFull") {
            m.kind = Matrix::ExactFull;
            parse_tvec(v.at("c0"), m.p + 0, 4);
            parse_tvec(v.at("c1"), m.p + 4, 4);
            parse_tvec(v.at("c2"), m.p + 8, 4);
            parse_tvec(v.at("c3"), m.p + 12, 4);
        } else if (tag == "If") {
            m.kind = Matrix::If;
            m.cond = parse_param(v.at("condition"));
            m.a = matrix_ref(v.at("then"));
            m.b = matrix_ref(v.at("otherwise"));
        } else if (tag == "Sqrt") {
            m.kind = Matrix::Sqrt;
            m.a = matrix_ref(v.items.at(0));
        } else if (tag == "Inv") {
            m.kind = Matrix::Inv;
            m.a = matrix_ref(v.items.at(0));
        } else if (tag == "Lerp") {
            m.kind = Matrix::Lerp;
            m.cond = parse_param(v.at("t"));
            m.a = matrix_ref(v.at("first"));
            m.b = matrix_ref(v.at("second"));
        } else if (tag == "Camera") {
            m.kind = Matrix::Camera;
        } else {
            throw SceneError("scene: unknown matrix kind `" + tag + "`");
        }
        return m;
    }
    Subspace parse_subspace(const Value* v) {
        if (!v) return Subspace::Normal;
        if (v->is_named("Subspace")) return Subspace::Subspace;
        if (v->is_named("Both")) return Subspace::Both;
        return Subspace::Normal;
    }
    void parse_kind(const Value& kind, Object& o) {
        if (kind.is_named("Simple")) {
            o.portal = false;
            o.m0 = matrix_ref(kind.items.at(0));
        } else if (kind.is_named("Portal")) {
            o.portal = true;
            o.m0 = matrix_ref(kind.items.at(0));
            o.m1 = matrix_ref(kind.items.at(1));
        } else {
            throw SceneError("scene: bad ObjectType");
        }
    }

    void load(const Value& doc) {
        if (doc.kind != Value::Struct) throw SceneError("scene: top level is not a struct");
        // cam
        const Value& cam = doc.at("cam");
        scene.cam.look_at = as_dvec3(cam.at("look_at"), "cam.look_at");
        scene.cam.alpha = as_f64(cam.at("alpha"), "cam.alpha");
        scene.cam.beta = as_f64(cam.at("beta"), "cam.beta");
        scene.cam.r = as_f64(cam.at("r"), "cam.r");
        scene.cam.offset_after_material = as_f64(cam.at("offset_after_material"), "cam.offset_after_material");
        if (const Value* v = doc.find("use_time")) scene.use_time = as_bool(*v, "use_time");
        if (const Value* v = doc.find("skybox"))
            if (const Value* s = v->some()) scene.skybox = as_string(*s, "skybox");

        // order follows deserialize_scene_new_format (scene_serialized.rs:1102-1230)
        for (const Value& t : storage_list(doc, "textures", true).items)
            scene.textures.push_back(Texture{as_string(t.at("name"), "texture name"), as_string(t.at("data"), "texture path")});
        for (const Value& v : storage_list(doc, "videos", false).items) scene.videos.push_back(as_string(v.at("name"), "video name"));
        for (const Value& u : storage_list(doc, "uniforms", true).items)
            scene.uniforms.push_back(UniformEntry{as_string(u.at("name"), "uniform name"), parse_uniform(u.at("data"))});
        for (const Value& v : storage_list(doc, "videos", false).items) {  // second pass: the frame-position uniform can be named now
            Scene::Video video;
            video.name = as_string(v.at("name"), "video name");
            const Value& d = v.at("data");
            if (const Value* path = d.find("path")) video.path = as_string(*path, "video path");
            if (const Value* u = d.find("uniform")) video.uniform = uniform_ref(*u);
            scene.video_sources.push_back(std::move(video));
        }

        const Value& mats = storage_list(doc, "matrices", true);
        for (const Value& m : mats.items) {
            std::string name = as_string(m.at("name"), "matrix name");
            matrix_by_name[name] = (int)scene.matrices.size();
            scene.matrices.push_back(MatrixEntry{name, true, Matrix{}});
        }
        for (const Value& m : mats.items) {
            int idx = matrix_by_name[as_string(m.at("name"), "matrix name")];
            Matrix value = parse_matrix(m.at("data"));
            scene.matrices[idx].value = value;
        }

        for (const Value& o : storage_list(doc, "objects", true).items) {
            Object obj;
            obj.name = as_string(o.at("name"), "object name");
            const Value& d = o.at("data");
            if (d.is_named("DebugMatrix")) {
                obj.kind = Object::DebugMatrix;
                obj.m0 = matrix_ref(d.items.at(0));
            } else if (d.is_named("Flat")) {
                obj.kind = Object::Flat;
                parse_kind(d.at("kind"), obj);
                obj.code = as_string(d.at("is_inside"), "is_inside code");
                obj.in_subspace = parse_subspace(d.find("in_subspace"));
            } else if (d.is_named("Complex")) {
                obj.kind = Object::Complex;
                parse_kind(d.at("kind"), obj);
                obj.code = as_string(d.at("intersect"), "intersect code");
                obj.in_subspace = parse_subspace(d.find("in_subspace"));
            } else {
                throw SceneError("scene: unknown object kind `" + d.s + "`");
            }
            scene.objects.push_back(std::move(obj));
        }

        for (const Value& m : storage_list(doc, "materials", true).items) {
            Material mat;
            mat.name = as_string(m.at("name"), "material name");
            const Value& d = m.at("data");
            auto rgb = [&](const Value& c) {
                for (int k = 0; k < 3; ++k) mat.color[k] = as_f64(c.items.at(k), "material color");
            };
            if (d.is_named("Simple")) {
                mat.kind = Material::Simple;
                rgb(d.at("color"));
                mat.normal_coef = as_f64(d.at("normal_coef"), "normal_coef");
                mat.grid = as_bool(d.at("grid"), "grid");
                mat.grid_scale = as_f64(d.at("grid_scale"), "grid_scale");
                mat.grid_coef = as_f64(d.at("grid_coef"), "grid_coef");
                mat.grid2 = d.find("grid2") ? as_bool(d.at("grid2"), "grid2") : false;
                mat.grid3 = d.find("grid3") ? as_bool(d.at("grid3"), "grid3") : false;
            } else if (d.is_named("Reflect")) {
                mat.kind = Material::Reflect;
                rgb(d.at("add_to_color"));
            } else if (d.is_named("Refract")) {
                mat.kind = Material::Refract;
                rgb(d.at("add_to_color"));
                mat.refractive_index = as_f64(d.at("refractive_index"), "refractive_index");
            } else if (d.is_named("Complex")) {
                mat.kind = Material::Complex;
                mat.code = as_string(d.at("code"), "material code");
            } else {
                throw SceneError("scene: unknown material kind `" + d.s + "`");
            }
            scene.materials.push_back(std::move(mat));
        }
        for (const Value& c : storage_list(doc, "cameras", false).items) {
            SceneCamera cam = parse_camera(c.at("data"));
            cam.name = as_string(c.at("name"), "camera name");
            scene.cameras.push_back(cam);
        }
        for (const Value& m : storage_list(doc, "intersection_materials", false).items)
            scene.intersection_materials.push_back(NamedCode{as_string(m.at("name"), "name"), as_string(m.at("data"), "intersection material code")});
        for (const Value& m : storage_list(doc, "library", true).items)
            scene.library.push_back(NamedCode{as_string(m.at("name"), "name"), as_string(m.at("data"), "library code")});

        // dev_stage + animation stages (scene_serialized.rs:1232-1330)
        if (const Value* dev = doc.find("dev_stage")) {
            if (const Value* u = dev->find("uniforms"))
                for (auto& kv : u->entries) scene.dev_uniforms.emplace_back(as_string(kv.first, "dev_stage key"), parse_uniform(kv.second));
            if (const Value* mm = dev->find("matrices"))
                for (auto& kv : mm->entries) scene.dev_matrices.emplace_back(as_string(kv.first, "dev_stage key"), parse_matrix(kv.second));
        }
        for (const Value& st : storage_list(doc, "animation_stages", false).items) {
            AnimationStage stage;
            stage.name = as_string(st.at("name"), "stage name");
            const Value& d = st.at("data");
            auto change = [&](const Value& v, bool is_matrix) {
                StageChange c;
                if (v.is_named("ProvidedToUser")) c.kind = StageChange::ProvidedToUser;
                else if (v.is_named("FromDev")) c.kind = StageChange::FromDev;
                else if (v.is_named("Changed") || v.is_named("ChangedAndToUser")) {
                    c.kind = v.is_named("Changed") ? StageChange::Changed : StageChange::ChangedAndToUser;
                    c.ref = is_matrix ? matrix_ref(v.items.at(0)) : uniform_ref(v.items.at(0));
                } else throw SceneError("scene: bad stage entry `" + v.s + "`");
                return c;
            };
            if (const Value* u = d.find("uniforms"))
                for (auto& kv : u->entries) stage.uniforms.emplace_back(as_string(kv.first, "stage key"), change(kv.second, false));
            if (const Value* mm = d.find("matrices"))
                for (auto& kv : mm->entries) stage.matrices.emplace_back(as_string(kv.first, "stage key"), change(kv.second, true));
            if (const Value* sc = d.find("set_cam")) {
                if (const Value* outer = sc->some()) {  // Some(..)
                    stage.has_set_cam = true;
                    stage.set_cam = cam_ref(*outer);  // Some(Some(CamRef))
                }
            }
            scene.stages.push_back(std::move(stage));
        }

        // real animations (scene_serialized.rs:1370-1473): names first, they may refer to each other
        const auto& anims = storage_list(doc, "animations", false).items;
        for (const Value& a : anims) {
            RealAnimation ra;
            ra.name = as_string(a.at("name"), "animation name");
            scene.animations.push_back(std::move(ra));
        }
        for (size_t k = 0; k < anims.size(); ++k) {
            const Value& d = anims[k].at("data");
            RealAnimation& ra = scene.animations[k];
            if (const Value* v = d.find("duration")) ra.duration = as_f64(*v, "animation duration");
            ra.base = stage_ref(d.at("animation_stage"));
            auto changes = [&](const Value* map, bool is_matrix, std::vector<std::pair<std::string, int>>& out) {
                if (!map) return;
                for (auto& kv : map->unwrap_newtypes().entries) {  // RealStageChangingSer is a newtype: ({..})
                    const Value& v = kv.second;
                    if (v.is_named("CopyPrev")) continue;
                    if (!v.is_named("Changed")) throw SceneError("scene: bad real-animation entry `" + v.s + "`");
                    int ref = is_matrix ? matrix_ref(v.items.at(0)) : uniform_ref(v.items.at(0));
                    if (ref >= 0) out.emplace_back(as_string(kv.first, "animation key"), ref);
                }
            };
            changes(d.find("uniforms"), false, ra.uniforms);
            changes(d.find("matrices"), true, ra.matrices);
            if (const Value* v = d.find("use_prev_cam")) ra.use_prev_cam = as_bool(*v, "use_prev_cam");
            if (const Value* v = d.find("use_start_cam_as_end")) ra.use_start_cam_as_end = as_bool(*v, "use_start_cam_as_end");
            if (const Value* v = d.find("cam_start")) ra.cam_start = cam_ref(*v);
            if (const Value* v = d.find("cam_end")) ra.cam_end = cam_ref(*v);
            auto opt_bool = [&](const char* key) -> std::optional<bool> {
                const Value* v = d.find(key);
                const Value* inner = v ? v->some() : nullptr;
                if (!inner) return std::nullopt;
                return as_bool(*inner, key);
            };
            ra.use_any_cam_as_start = opt_bool("use_any_cam_as_start");
            ra.use_any_cam_as_end = opt_bool("use_any_cam_as_end");
            auto anim_ref = [&](const char* key) {
                const Value* v = d.find(key);
                const Value* inner = v ? v->some() : nullptr;
                return inner ? scene.find_animation(as_string(*inner, key)) : -1;
            };
            ra.cam_any_start = anim_ref("cam_any_start");
            ra.cam_any_end = anim_ref("cam_any_end");
            if (const Value* v = d.find("cam_easing")) {
                static const char* names[6] = {"Linear", "In", "Out", "InOut", "InOutFast", "ElasticOut"};
                bool ok = false;
                for (int e = 0; e < 6; ++e)
                    if (v->is_named(names[e])) {
                        ra.cam_easing = (Easing)e;
                        ok = true;
                    }
                if (!ok) throw SceneError("scene: unknown easing `" + v->s + "`");
            }
            if (const Value* v = d.find("cam_easing_uniform")) {
                int ref = uniform_ref(*v);
                ra.has_easing_uniform = ref >= 0;
                ra.easing_uniform = ref;
            }
        }
        if (const Value* cs = doc.find("current_stage")) scene.current_stage = stage_ref(*cs);
    }

    // CurrentStageSer (scene_serialized.rs:210-229): unknown names fall back to Dev
    StageRef stage_ref(const Value& v) {
        StageRef r;
        if (v.is_named("Animation")) {
            std::string name = as_string(v.items.at(0), "stage name");
            for (size_t k = 0; k < scene.stages.size(); ++k)
                if (scene.stages[k].name == name && r.index < 0) r = StageRef{StageRef::Animation, (int)k};
        } else if (v.is_named("RealAnimation")) {
            int idx = scene.find_animation(as_string(v.items.at(0), "animation name"));
            if (idx >= 0) r = StageRef{StageRef::RealAnimation, idx};
        }
        return r;
    }

    // Option<CamRef> -> camera index (inline cameras are appended)
    int cam_ref(const Value& opt) {
        const Value* inner = opt.some();
        if (!inner) return -1;
        if (inner->is_named("Named")) return scene.find_camera(as_string(inner->items.at(0), "CamRef"));
        if (inner->is_named("Inline")) {
            scene.cameras.push_back(parse_camera(inner->items.at(0)));
            return (int)scene.cameras.size() - 1;
        }
        throw SceneError("scene: bad CamRef");
    }

    SceneCamera parse_camera(const Value& d) {
        SceneCamera cam;
        const Value& la = d.at("look_at");
        if (la.is_named("MatrixCenter")) {
            cam.look_at_matrix = true;
            cam.matrix = matrix_ref(la.items.at(0));
        } else {
            cam.coordinate = as_dvec3(la.items.at(0), "camera look_at");
        }
        cam.alpha = as_f64(d.at("alpha"), "camera alpha");
        cam.beta = as_f64(d.at("beta"), "camera beta");
        cam.r = as_f64(d.at("r"), "camera r");
        if (const Value* v = d.find("in_subspace")) cam.in_subspace = as_bool(*v, "camera in_subspace");
        if (const Value* v = d.find("free_movement")) cam.free_movement = as_bool(*v, "camera free_movement");
        if (const Value* v = d.find("matrix")) {
            if (v->kind == Value::Tuple && v->items.size() == 16) {
                double e[16];
                for (int k = 0; k < 16; ++k) e[k] = as_f64(v->items[k], "camera matrix");
                cam.teleport = DMat4::from_cols({e[0], e[1], e[2], e[3]}, {e[4], e[5], e[6], e[7]}, {e[8], e[9], e[10], e[11]}, {e[12], e[13], e[14], e[15]});
            }
        }
        return cam;
    }
};

}  // namespace

std::shared_ptr<Scene> Scene::from_ron_text(const std::string& text) {
    ron::Value doc = ron::parse(text);
    auto scene = std::make_shared<Scene>();
    Loader loader{*scene, {}};
    loader.load(doc);
    scene->doc = std::move(doc);
    return scene;
}

std::shared_ptr<Scene> Scene::from_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError("cannot open scene file `" + path + "`");
    std::stringstream ss;
    ss << f.rdbuf();
    return from_ron_text(ss.str());
}

int Scene::find_uniform(const std::string& name) const {
    for (size_t k = 0; k < uniforms.size(); ++k)
        if (!uniforms[k].name.empty() && uniforms[k].name == name) return (int)k;
    return -1;
}
int Scene::find_matrix(const std::string& name) const {
    for (size_t k = 0; k < matrices.size(); ++k)
        if (matrices[k].name == name) return (int)k;
    return -1;
}

namespace {

ron::Value ron_float(double f) {
    ron::Value v;
    v.kind = ron::Value::Float;
    v.f = f;
    return v;
}
ron::Value ron_unit(const char* name) {
    ron::Value v;
    v.kind = ron::Value::Unit;
    v.s = name;
    return v;
}
ron::Value ron_newtype(const char* name, ron::Value inner) {
    ron::Value v;
    v.kind = ron::Value::Tuple;
    v.s = name;
    v.items.push_back(std::move(inner));
    return v;
}

// the `data:` of uniform `name` in the document follows the model (AnyUniform, scene_serialized.rs:36-49)
void write_uniform_to_doc(ron::Value& doc, const std::string& name, const Uniform& u) {
    for (auto& field : doc.fields) {
        if (field.first != "uniforms") continue;
        ron::Value* list = &field.second;
        while (list->kind == ron::Value::Tuple && list->items.size() == 1) list = &list->items[0];
        for (ron::Value& item : list->items) {
            const ron::Value* n = item.find("name");
            if (!n || n->s != name) continue;
            for (auto& f : item.fields) {
                if (f.first != "data") continue;
                ron::Value& data = f.second;
                auto clamped = [&](const char* tag, bool is_int) {  // Float((min: .., max: .., value: x)): keep the limits
                    if (data.is_named(tag) && data.items.size() == 1) {
                        ron::Value* inner = &data.items[0];
                        while (inner->kind == ron::Value::Tuple && inner->items.size() == 1) inner = &inner->items[0];
                        for (auto& g : inner->fields)
                            if (g.first == "value") {
                                if (is_int) {
                                    g.second.kind = ron::Value::Int;
                                    g.second.i = u.i;
                                } else {
                                    g.second = ron_float(u.f);
                                }
                                return;
                            }
                    }
                    ron::Value st;
                    st.kind = ron::Value::Struct;
                    st.fields.emplace_back("min", ron_unit("None"));
                    st.fields.emplace_back("max", ron_unit("None"));
                    ron::Value val = ron_float(u.f);
                    if (is_int) {
                        val.kind = ron::Value::Int;
                        val.i = u.i;
                    }
                    st.fields.emplace_back("value", val);
                    data = ron_newtype(tag, ron_newtype("", std::move(st)));
                };
                switch (u.kind) {
                    case Uniform::Bool: {
                        ron::Value b;
                        b.kind = ron::Value::Bool;
                        b.b = u.b;
                        data = ron_newtype("Bool", b);
                        break;
                    }
                    case Uniform::Int: clamped("Int", true); break;
                    case Uniform::Float: clamped("Float", false); break;
                    case Uniform::Angle: data = ron_newtype("Angle", ron_float(u.f)); break;
                    case Uniform::Progress: data = ron_newtype("Progress", ron_float(u.f)); break;
                    default: break;
                }
            }
        }
    }
}

ron::Value* doc_field(ron::Value& doc, const char* name) {
    for (auto& f : doc.fields)
        if (f.first == name) return &f.second;
    return nullptr;
}
ron::Value* unwrap(ron::Value* v) {
    while (v && v->kind == ron::Value::Tuple && v->s.empty() && v->items.size() == 1) v = &v->items[0];
    return v;
}
// `data:` of the element called `name` in the storage list `list_name` ("uniforms" / "matrices")
ron::Value* element_data(ron::Value& doc, const char* list_name, const std::string& name) {
    ron::Value* list = unwrap(doc_field(doc, list_name));
    if (!list) return nullptr;
    for (ron::Value& item : list->items) {
        const ron::Value* n = item.find("name");
        if (!n || n->s != name) continue;
        for (auto& f : item.fields)
            if (f.first == "data") return &f.second;
    }
    return nullptr;
}
// Storage2::set_id / set on the document: what a stage entry puts into the element it names.
//   Changed(Some(Inline(X))) / ChangedAndToUser(..) -> X;  ..(Some(Named(n))) -> a copy of n's data;  FromDev / ProvidedToUser -> dev value
void apply_entries_to_doc(ron::Value& doc, const ron::Value* entries, const char* list_name, const ron::Value* dev_entries) {
    if (!entries) return;
    const ron::Value* map = entries;
    while (map->kind == ron::Value::Tuple && map->s.empty() && map->items.size() == 1) map = &map->items[0];
    for (auto& kv : map->entries) {
        const std::string& name = kv.first.s;
        const ron::Value& how = kv.second;
        ron::Value replacement;
        bool have = false;
        if ((how.is_named("Changed") || how.is_named("ChangedAndToUser")) && how.items.size() == 1) {
            const ron::Value* ref = how.items[0].some();
            if (ref && ref->is_named("Inline") && ref->items.size() == 1) {
                replacement = ref->items[0];
                have = true;
            } else if (ref && ref->is_named("Named") && ref->items.size() == 1) {
                if (const ron::Value* src = element_data(doc, list_name, ref->items[0].s)) {
                    replacement = *src;
                    have = true;
                }
            }
        } else if ((how.is_named("FromDev") || how.is_named("ProvidedToUser")) && dev_entries) {
            const ron::Value* dev = dev_entries;
            while (dev->kind == ron::Value::Tuple && dev->s.empty() && dev->items.size() == 1) dev = &dev->items[0];
            for (auto& d : dev->entries)
                if (d.first.s == name) {
                    replacement = d.second;
                    have = true;
                }
        }
        if (!have) continue;
        if (ron::Value* data = element_data(doc, list_name, name)) *data = std::move(replacement);
    }
}
// the `data:` block of the stage / clip called `name` in the storage list `list_name`
const ron::Value* storage_item_data(const ron::Value& doc, const char* list_name, const std::string& name) {
    const ron::Value* list = doc.find(list_name);
    while (list && list->kind == ron::Value::Tuple && list->s.empty() && list->items.size() == 1) list = &list->items[0];
    if (!list) return nullptr;
    for (const ron::Value& item : list->items) {
        const ron::Value* n = item.find("name");
        if (n && n->s == name) return item.find("data");
    }
    return nullptr;
}

}  // namespace

// The document side of Scene::init_stage: the reference's set_id / set copy values INTO the stored elements, so a scene saved
// after a stage was applied holds the stage's values.  Same here, on the RON tree (subtrees are copied as they are).
void Scene::apply_stage_to_doc(StageRef stage) {
    const ron::Value* dev = doc.find("dev_stage");
    const ron::Value* dev_u = dev ? dev->find("uniforms") : nullptr;
    const ron::Value* dev_m = dev ? dev->find("matrices") : nullptr;
    if (stage.kind == StageRef::Dev) {
        auto restore = [&](const ron::Value* entries, const char* list_name) {
            if (!entries) return;
            const ron::Value* map = entries;
            while (map->kind == ron::Value::Tuple && map->s.empty() && map->items.size() == 1) map = &map->items[0];
            for (auto& d : map->entries)
                if (ron::Value* data = element_data(doc, list_name, d.first.s)) *data = d.second;
        };
        ron::Value du = dev_u ? *dev_u : ron::Value{}, dm = dev_m ? *dev_m : ron::Value{};  // copies: `doc` is edited below
        restore(dev_u ? &du : nullptr, "uniforms");
        restore(dev_m ? &dm : nullptr, "matrices");
        return;
    }
    const char* list_name = stage.kind == StageRef::Animation ? "animation_stages" : "animations";
    const std::string& name = stage.kind == StageRef::Animation ? stages.at(stage.index).name : animations.at(stage.index).name;
    const ron::Value* data = storage_item_data(doc, list_name, name);
    if (!data) return;
    ron::Value entries_u = data->find("uniforms") ? *data->find("uniforms") : ron::Value{};
    ron::Value entries_m = data->find("matrices") ? *data->find("matrices") : ron::Value{};
    ron::Value du = dev_u ? *dev_u : ron::Value{}, dm = dev_m ? *dev_m : ron::Value{};
    apply_entries_to_doc(doc, &entries_u, "uniforms", dev_u ? &du : nullptr);
    apply_entries_to_doc(doc, &entries_m, "matrices", dev_m ? &dm : nullptr);
}

std::string Scene::to_ron() const {
    ron::Value out = doc;
    // current_stage (CurrentStageSer, scene_serialized.rs:585-590)
    ron::Value stage = ron_unit("Dev");
    auto named = [](const char* tag, const std::string& name) {
        ron::Value s;
        s.kind = ron::Value::String;
        s.s = name;
        return ron_newtype(tag, s);
    };
    if (current_stage.kind == StageRef::Animation && current_stage.index >= 0) stage = named("Animation", stages[current_stage.index].name);
    if (current_stage.kind == StageRef::RealAnimation && current_stage.index >= 0) stage = named("RealAnimation", animations[current_stage.index].name);
    bool had = false;
    for (auto& f : out.fields)
        if (f.first == "current_stage") {
            f.second = stage;
            had = true;
        }
    if (!had && current_stage.kind != StageRef::Dev) out.fields.emplace_back("current_stage", stage);
    // cam block (CamSettings): the camera a renderer last handed back, see Scene::cam
    for (auto& f : out.fields) {
        if (f.first != "cam") continue;
        for (auto& g : f.second.fields) {
            if (g.first == "alpha") g.second = ron_float(cam.alpha);
            else if (g.first == "beta") g.second = ron_float(cam.beta);
            else if (g.first == "r") g.second = ron_float(cam.r);
            else if (g.first == "offset_after_material") g.second = ron_float(cam.offset_after_material);
            else if (g.first == "look_at" && g.second.kind == ron::Value::Tuple && g.second.items.size() == 3) {
                g.second.items[0] = ron_float(cam.look_at.x);
                g.second.items[1] = ron_float(cam.look_at.y);
                g.second.items[2] = ron_float(cam.look_at.z);
            }
        }
    }
    return ron::to_string(out);
}

bool Scene::set_uniform_value(const std::string& name, double v) {
    int idx = find_uniform(name);
    if (idx < 0) return false;
    if (idx < (int)uniform_alias.size() && uniform_alias[idx] >= 0) {  // Storage2::set_id copied the value: edits apply to the copy
        uniforms[idx].value = uniforms[uniform_alias[idx]].value;
        uniform_alias[idx] = -1;
    }
    Uniform& u = uniforms[idx].value;
    ++version;
    switch (u.kind) {
        case Uniform::Bool: u.b = v > 0.5; break;
        case Uniform::Int: u.i = (int)v; break;
        case Uniform::Float: case Uniform::Angle: case Uniform::Progress: u.f = v; break;
        default: u.kind = Uniform::Float; u.f = v; break;  // a stage replaces a formula by a value
    }
    write_uniform_to_doc(doc, name, u);
    return true;
}

bool Scene::trefoil_decode(const std::string& text, int out[18][3]) {
    int parts[18][3] = {};
    auto name_to_index = [](const std::string& n) -> int {  // "1a" .. "6c"
        if (n.size() != 2 || n[1] < 'a' || n[1] > 'c' || n[0] < '1' || n[0] > '6') return -1;
        return (n[1] - 'a') + (n[0] - '1') * 3;
    };
    static const char* kColors[6] = {"R", "G", "B", "Y", "V", "S"};
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t comma = text.find(',', pos);
        if (comma == std::string::npos) comma = text.size();
        std::string item = text.substr(pos, comma - pos);
        pos = comma + 1;
        if (item.empty()) continue;
        std::vector<std::string> words;  // split(" "): exactly three words, no tolerance for double spaces (like the reference)
        size_t w = 0;
        for (;;) {
            size_t sp = item.find(' ', w);
            words.push_back(item.substr(w, sp == std::string::npos ? std::string::npos : sp - w));
            if (sp == std::string::npos) break;
            w = sp + 1;
        }
        if (words.size() != 3) return false;
        int index = name_to_index(words[0]), to = name_to_index(words[1]), color = -1;
        for (int c = 0; c < 6; ++c)
            if (words[2] == kColors[c]) color = c;
        if (index < 0 || to < 0 || color < 0) return false;
        parts[index][0] = 1;
        parts[index][1] = to;
        parts[index][2] = color;
    }
    std::memcpy(out, parts, sizeof parts);
    return true;
}

std::string Scene::trefoil_encode(const int parts[18][3]) {
    static const char* kColors[6] = {"R", "G", "B", "Y", "V", "S"};
    auto index_to_name = [](int i) { return std::string(1, (char)('1' + i / 3)) + (char)('a' + i % 3); };
    std::string out;
    for (int i = 0; i < 18; ++i) {
        if (!parts[i][0]) continue;
        if (!out.empty()) out += ',';
        out += index_to_name(i) + " " + index_to_name(parts[i][1]) + " " + kColors[parts[i][2] % 6];
    }
    return out;
}

bool Scene::set_trefoil(const std::string& name, const std::string& text) {
    int idx = find_uniform(name);
    if (idx < 0) return false;
    int parts[18][3];
    if (!trefoil_decode(text, parts)) return false;
    if (idx < (int)uniform_alias.size() && uniform_alias[idx] >= 0) {
        uniforms[idx].value = uniforms[uniform_alias[idx]].value;
        uniform_alias[idx] = -1;
    }
    Uniform& u = uniforms[idx].value;
    if (u.kind != Uniform::Trefoil) return false;
    std::memcpy(u.trefoil, parts, sizeof parts);
    ++version;
    // the document: TrefoilSpecial(( ((enabled, to, colour), ... x18) ))
    for (auto& field : doc.fields) {
        if (field.first != "uniforms") continue;
        ron::Value* list = &field.second;
        while (list->kind == ron::Value::Tuple && list->items.size() == 1 && list->s.empty()) list = &list->items[0];
        for (ron::Value& item : list->items) {
            const ron::Value* n = item.find("name");
            if (!n || n->s != name) continue;
            for (auto& f : item.fields) {
                if (f.first != "data") continue;
                ron::Value tuple;
                tuple.kind = ron::Value::Tuple;
                for (int i = 0; i < 18; ++i) {
                    ron::Value e, b, to, col;
                    e.kind = ron::Value::Tuple;
                    b.kind = ron::Value::Bool;
                    b.b = parts[i][0] != 0;
                    to.kind = col.kind = ron::Value::Int;
                    to.i = parts[i][1];
                    col.i = parts[i][2];
                    e.items = {b, to, col};
                    tuple.items.push_back(e);
                }
                ron::Value inner;
                inner.kind = ron::Value::Tuple;
                inner.items.push_back(tuple);
                ron::Value outer;
                outer.kind = ron::Value::Tuple;
                outer.s = "TrefoilSpecial";
                outer.items.push_back(inner);
                f.second = outer;
            }
        }
    }
    return true;
}

std::optional<std::string> Scene::get_trefoil(const std::string& name) const {
    const Uniform* u = resolved_uniform(find_uniform(name));
    if (!u || u->kind != Uniform::Trefoil) return std::nullopt;
    return trefoil_encode(u->trefoil);
}

int Scene::find_camera(const std::string& name) const {
    for (size_t k = 0; k < cameras.size(); ++k)
        if (!cameras[k].name.empty() && cameras[k].name == name) return (int)k;
    return -1;
}

std::optional<DVec3> Scene::camera_look_at(const SceneCamera& c) const {
    if (!c.look_at_matrix) return c.coordinate;
    auto m = eval_matrix(c.matrix);
    if (!m) return std::nullopt;
    double inv_w = 1.0 / m->c[3].w;  // project_point3(ZERO): w_axis.xyz / w_axis.w
    return DVec3(m->c[3].x * inv_w, m->c[3].y * inv_w, m->c[3].z * inv_w) + DVec3(0.001, 0.001, 0.001);
}

bool Scene::init_stage_by_name(const std::string& name, int* camera) {
    int found = -1;
    for (size_t k = 0; k < stages.size(); ++k)
        if (stages[k].name == name) found = (int)k;
    if (found < 0) return false;
    init_stage(StageRef{StageRef::Animation, found});
    if (camera) *camera = current_cam;
    return true;
}

int Scene::find_animation(const std::string& name) const {
    for (size_t k = 0; k < animations.size(); ++k)
        if (animations[k].name == name) return (int)k;
    return -1;
}

bool Scene::init_animation_by_name(const std::string& name) {
    int idx = find_animation(name);
    if (idx < 0) return false;
    init_stage(StageRef{StageRef::RealAnimation, idx});
    return true;
}

void Scene::init_stage(StageRef stage, int depth) {
    ++version;
    if (uniform_alias.size() != uniforms.size()) uniform_alias.assign(uniforms.size(), -1);
    if (matrix_alias.size() != matrices.size()) matrix_alias.assign(matrices.size(), -1);
    auto restore_dev_uniform = [&](const std::string& name, int idx) {
        for (auto& dv : dev_uniforms)
            if (dv.first == name) {
                uniforms[idx].value = dv.second;
                uniform_alias[idx] = -1;
            }
    };
    auto restore_dev_matrix = [&](const std::string& name, int idx) {
        for (auto& dv : dev_matrices)
            if (dv.first == name) {
                matrices[idx].value = dv.second;
                matrix_alias[idx] = -1;
            }
    };
    switch (stage.kind) {
        case StageRef::Animation: {
            const AnimationStage& st = stages.at(stage.index);
            // StageChanging::init_stage (animation.rs:171-183): Changed* -> set_id, FromDev / ProvidedToUser -> dev value
            for (auto& kv : st.uniforms) {
                int idx = find_uniform(kv.first);
                if (idx < 0) continue;
                const StageChange& c = kv.second;
                if ((c.kind == StageChange::Changed || c.kind == StageChange::ChangedAndToUser) && c.ref >= 0) {
                    uniform_alias[idx] = c.ref != idx ? c.ref : -1;
                } else if (c.kind == StageChange::FromDev || c.kind == StageChange::ProvidedToUser) {
                    restore_dev_uniform(kv.first, idx);
                }
            }
            for (auto& kv : st.matrices) {
                int idx = find_matrix(kv.first);
                if (idx < 0 || !matrices[idx].named) continue;
                const StageChange& c = kv.second;
                if ((c.kind == StageChange::Changed || c.kind == StageChange::ChangedAndToUser) && c.ref >= 0) {
                    matrix_alias[idx] = c.ref != idx ? c.ref : -1;
                } else if (c.kind == StageChange::FromDev || c.kind == StageChange::ProvidedToUser) {
                    restore_dev_matrix(kv.first, idx);
                }
            }
            current_cam = st.has_set_cam ? st.set_cam : -1;
            break;
        }
        case StageRef::Dev: {  // DevStageChanging::init_stage (animation.rs:222-228)
            for (auto& dv : dev_uniforms) {
                int idx = find_uniform(dv.first);
                if (idx >= 0) restore_dev_uniform(dv.first, idx);
            }
            for (auto& dv : dev_matrices) {
                int idx = find_matrix(dv.first);
                if (idx >= 0 && matrices[idx].named) restore_dev_matrix(dv.first, idx);
            }
            current_cam = -1;
            break;
        }
        case StageRef::RealAnimation: {  // scene.rs:1208-1231
            const RealAnimation& a = animations.at(stage.index);
            if (!(a.base == stage) && depth < 64) init_stage(a.base, depth + 1);  // else: "Initialization recursion!"
            // RealAnimationStageChanging::init_stage (animation.rs:985-991): only Changed(Some(..)) entries act
            for (auto& kv : a.uniforms) {
                int idx = find_uniform(kv.first);
                if (idx >= 0) uniform_alias[idx] = kv.second != idx ? kv.second : -1;
            }
            for (auto& kv : a.matrices) {
                int idx = find_matrix(kv.first);
                if (idx >= 0 && matrices[idx].named) matrix_alias[idx] = kv.second != idx ? kv.second : -1;
            }
            int cam = animation_start_cam(stage.index);
            if (cam >= 0) current_cam = cam;
            break;
        }
    }
    apply_stage_to_doc(stage);  // (a clip's base stage has been applied by the recursive call above)
    current_stage = stage;
}

int Scene::animation_start_cam(int id, int depth) const {
    if (id < 0 || id >= (int)animations.size() || depth > 256) return -1;
    const RealAnimation& a = animations[id];
    if (a.use_prev_cam) return id > 0 ? animation_end_cam(id - 1, depth + 1) : -1;
    if (a.use_any_cam_as_start) {
        if (a.cam_any_start < 0) return -1;
        return *a.use_any_cam_as_start ? animation_end_cam(a.cam_any_start, depth + 1) : animation_start_cam(a.cam_any_start, depth + 1);
    }
    return a.cam_start;
}

int Scene::animation_end_cam(int id, int depth) const {
    if (id < 0 || id >= (int)animations.size() || depth > 256) return -1;
    const RealAnimation& a = animations[id];
    if (a.use_start_cam_as_end) return animation_start_cam(id, depth + 1);
    if (a.use_any_cam_as_end) {
        if (a.cam_any_end < 0) return -1;
        return *a.use_any_cam_as_end ? animation_end_cam(a.cam_any_end, depth + 1) : animation_start_cam(a.cam_any_end, depth + 1);
    }
    return a.cam_end;
}

double Scene::total_animation_duration() const {
    double total = 0.0;
    for (auto& a : animations) total += a.duration;
    return total;
}

std::optional<CalculatedCam> Scene::calculated_cam(const SceneCamera& c) const {
    auto look = camera_look_at(c);
    if (!look) return std::nullopt;
    CalculatedCam out;
    out.look_at = *look;
    out.alpha = c.alpha;
    out.beta = c.beta;
    out.r = c.r;
    out.in_subspace = c.in_subspace;
    out.free_movement = c.free_movement;
    out.matrix = c.teleport;
    out.override_matrix = true;
    return out;
}

double ease(Easing e, double t) {
    constexpr double kPi = 3.14159265358979323846264338327950288;
    auto in = [&](double x) { return 1.0 - std::cos(x * kPi * 0.5); };
    auto in_out = [&](double x) { return (1.0 - std::cos(x * kPi)) * 0.5; };
    switch (e) {
        case Easing::Linear: return t;
        case Easing::In: return in(t);
        case Easing::Out: return 1.0 - in(1.0 - t);
        case Easing::InOut: return in_out(t);
        case Easing::InOutFast: return in_out(in_out(t));
        case Easing::ElasticOut: {
            double c4 = (2.0 * kPi) / 3.0;
            if (t == 0.0) return 0.0;
            if (t == 1.0) return 1.0;
            return std::pow(2.0, -10.0 * t) * std::sin((t * 10.0 - 0.75) * c4) + 1.0;
        }
    }
    return t;
}

std::optional<CalculatedCam> Scene::update(double seconds) {
    double t = seconds, total = 0.0;
    if (run_animations) {  // scene.rs:1359-1386
        double total_duration = total_animation_duration();
        if (total_duration > 0.0) {
            t = std::fmod(t, total_duration);
            total = t;
        } else {
            t = 0.0;
        }
        for (size_t id = 0; id < animations.size(); ++id) {
            double duration = animations[id].duration;
            if (t < duration) {
                StageRef want{StageRef::RealAnimation, (int)id};
                if (!(current_stage == want)) init_stage(want);
                t /= duration;
                break;
            }
            t -= duration;
        }
    } else if (current_stage.kind == StageRef::RealAnimation) {  // scene.rs:1387-1426 (no manual-time slider offline)
        double duration = animations[current_stage.index].duration;
        if (duration > 0.0) {
            double local_seconds = std::fmod(t, duration);
            t = local_seconds / duration;
            double prefix = 0.0;
            for (int k = 0; k < current_stage.index; ++k) prefix += animations[k].duration;
            total = prefix + local_seconds;
        } else {
            t = 0.0;
        }
    } else {
        total = t;
    }
    if (t != time || total != total_time) ++version;
    time = t;
    total_time = total;

    if (current_stage.kind != StageRef::RealAnimation) return std::nullopt;
    int id = current_stage.index;
    const RealAnimation& a = animations[id];
    int c1 = animation_start_cam(id), c2 = animation_end_cam(id);
    if (c1 < 0 || c2 < 0) return std::nullopt;
    auto cam1 = calculated_cam(cameras.at(c1));
    auto cam2 = calculated_cam(cameras.at(c2));
    if (!cam1 || !cam2) throw SceneError("scene: animation camera can't be evaluated");  // .unwrap() in the reference
    double t_raw = std::fmod(time, 1.0);
    double te = ease(a.cam_easing, t_raw);
    if (a.has_easing_uniform) {  // scene.rs:1453-1469
        if (auto v = eval_uniform(a.easing_uniform)) {
            double x = v->as_f64();
            if (!std::isfinite(x)) x = 0.0;
            te = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
        }
    }
    CalculatedCam cam;
    cam.look_at = cam1->look_at + (cam2->look_at - cam1->look_at) * te;  // glam 0.13 DVec3::lerp
    auto lerp = [&](double lo, double hi) { return (1.0 - te) * lo + te * hi; };  // emath 0.31 lerp
    cam.alpha = lerp(cam1->alpha, cam2->alpha);
    cam.beta = lerp(cam1->beta, cam2->beta);
    cam.r = lerp(cam1->r, cam2->r);
    cam.in_subspace = cam1->in_subspace;
    cam.free_movement = cam1->free_movement;
    cam.matrix = cam1->matrix;
    cam.override_matrix = t_raw < prev_t_raw || t_raw == 0.0;
    prev_t_raw = t_raw;
    return cam;
}

const Uniform* Scene::resolved_uniform(int index) const {
    for (int hops = 0; index >= 0 && index < (int)uniform_alias.size() && uniform_alias[index] >= 0 && hops < 64; ++hops) index = uniform_alias[index];
    if (index < 0 || index >= (int)uniforms.size()) return nullptr;
    return &uniforms[index].value;
}

std::optional<double> Scene::eval_formula(const std::string& text) const {
    auto it = formula_cache_.find(text);
    if (it == formula_cache_.end()) it = formula_cache_.emplace(text, Formula::compile(text)).first;
    if (!it->second) return std::nullopt;
    FormulaNamespace ns = [this](const std::string& name, const std::vector<double>& args) -> std::optional<double> {
        bool known = false;
        auto r = formula_custom_function(name, args, &known);
        if (known) return r;
        if (name == "time") {
            frame_input_read_ = true;
            return time;
        }
        if (name == "total_time") {
            frame_input_read_ = true;
            return total_time;
        }
        int idx = find_uniform(name);  // free variable = another named uniform
        if (idx < 0) return std::nullopt;
        auto v = eval_uniform(idx);
        if (!v) return std::nullopt;
        return v->as_f64();
    };
    return it->second->eval(ns);
}

std::optional<UniformValue> Scene::eval_uniform(int index) const {
    for (int hops = 0; index >= 0 && index < (int)uniform_alias.size() && uniform_alias[index] >= 0 && hops < 64; ++hops) index = uniform_alias[index];
    if (index < 0 || index >= (int)uniforms.size()) return std::nullopt;
    if (uniform_busy_.size() < uniforms.size()) uniform_busy_.resize(uniforms.size(), 0);
    if (uniform_busy_[index]) return std::nullopt;  // recursion
    const Uniform& u = uniforms[index].value;
    UniformValue out;
    switch (u.kind) {
        case Uniform::Bool: out.kind = UniformValue::Bool; out.b = u.b; return out;
        case Uniform::Int: out.kind = UniformValue::Int; out.i = u.i; return out;
        case Uniform::Float: case Uniform::Angle: case Uniform::Progress: out.kind = UniformValue::Float; out.f = u.f; return out;
        case Uniform::Formula: case Uniform::FormulaInt: {
            uniform_busy_[index] = 1;
            auto v = eval_formula(u.formula);
            uniform_busy_[index] = 0;
            if (!v) return std::nullopt;
            if (u.kind == Uniform::Formula) {
                out.kind = UniformValue::Float;
                out.f = *v;
            } else {
                out.kind = UniformValue::Int;
                double d = *v;  // Rust `as i32`: saturating, NaN -> 0
                out.i = std::isnan(d) ? 0 : (d >= 2147483647.0 ? 2147483647 : (d <= -2147483648.0 ? (-2147483647 - 1) : (int)d));
            }
            return out;
        }
        case Uniform::Trefoil: return std::nullopt;
    }
    return std::nullopt;
}

std::optional<double> Scene::eval_param(const Param& p) const {
    if (!p.is_uniform) return p.value;
    auto v = eval_uniform(p.uniform);
    if (!v) return std::nullopt;
    return v->as_f64();
}

std::optional<DMat4> Scene::eval_matrix(int index) const {
    for (int hops = 0; index >= 0 && index < (int)matrix_alias.size() && matrix_alias[index] >= 0 && hops < 64; ++hops) index = matrix_alias[index];
    if (index < 0 || index >= (int)matrices.size()) return std::nullopt;
    if (matrix_busy_.size() < matrices.size()) matrix_busy_.resize(matrices.size(), 0);
    if (matrix_busy_[index]) return std::nullopt;
    struct Guard {
        char& flag;
        explicit Guard(char& f) : flag(f) { flag = 1; }
        ~Guard() { flag = 0; }
    } guard(matrix_busy_[index]);

    const Matrix& m = matrices[index].value;
    auto srt = [](const DVec3& scale, const DVec3& rot, const DVec3& offset) {
        return DMat4::from_scale_rotation_translation(scale, DQuat::rotation_x(rot.x) * DQuat::rotation_y(rot.y) * DQuat::rotation_z(rot.z), offset);
    };
    switch (m.kind) {
        case Matrix::Mul: {
            auto to = eval_matrix(m.a);
            if (!to) return std::nullopt;
            auto what = eval_matrix(m.b);
            if (!what) return std::nullopt;
            return *what * *to;
        }
        case Matrix::Teleport: {
            auto first = eval_matrix(m.a);
            if (!first) return std::nullopt;
            auto second = eval_matrix(m.b);
            if (!second) return std::nullopt;
            auto what = eval_matrix(m.c);
            if (!what) return std::nullopt;
            return *second * first->inverse() * *what;
        }
        case Matrix::Simple:
            return srt(DVec3(m.scale * (m.mirror[0] ? -1.0 : 1.0), m.scale * (m.mirror[1] ? -1.0 : 1.0), m.scale * (m.mirror[2] ? -1.0 : 1.0)), m.rotate, m.offset);
        case Matrix::Parametrized: {
            double v[10];
            // evaluation order of the reference: scale, mirror xyz, rotate xyz, offset xyz
            auto s = eval_param(m.p[9]);
            if (!s) return std::nullopt;
            v[9] = *s;
            for (int k : {6, 7, 8, 3, 4, 5, 0, 1, 2}) {
                auto x = eval_param(m.p[k]);
                if (!x) return std::nullopt;
                v[k] = *x;
            }
            return srt(DVec3(v[9] * (1.0 - 2.0 * v[6]), v[9] * (1.0 - 2.0 * v[7]), v[9] * (1.0 - 2.0 * v[8])), DVec3(v[3], v[4], v[5]), DVec3(v[0], v[1], v[2]));
        }
        case Matrix::Exact: {
            double v[12];
            for (int k = 0; k < 12; ++k) {
                auto x = eval_param(m.p[k]);
                if (!x) return std::nullopt;
                v[k] = *x;
            }
            return DMat4::from_cols({v[0], v[1], v[2], 0.0}, {v[3], v[4], v[5], 0.0}, {v[6], v[7], v[8], 0.0}, {v[9], v[10], v[11], 1.0});
        }
        case Matrix::ExactFull: {
            double v[16];
            for (int k = 0; k < 16; ++k) {
                auto x = eval_param(m.p[k]);
                if (!x) return std::nullopt;
                v[k] = *x;
            }
            return DMat4::from_cols({v[0], v[1], v[2], v[3]}, {v[4], v[5], v[6], v[7]}, {v[8], v[9], v[10], v[11]}, {v[12], v[13], v[14], v[15]});
        }
        case Matrix::If: {
            auto c = eval_param(m.cond);
            if (!c) return std::nullopt;
            return eval_matrix(*c > 0.5 ? m.a : m.b);
        }
        case Matrix::Inv: {
            auto a = eval_matrix(m.a);
            if (!a) return std::nullopt;
            return a->inverse();
        }
        case Matrix::Camera: frame_input_read_ = true; return camera_matrix;
        case Matrix::Lerp: {  // src/gui/matrix.rs:614-627: decompose both into TRS, blend each part, recompose
            auto t = eval_param(m.cond);
            if (!t) return std::nullopt;
            auto first = eval_matrix(m.a);
            if (!first) return std::nullopt;
            auto second = eval_matrix(m.b);
            if (!second) return std::nullopt;
            DVec3 fs, ft, ss, st;
            DQuat fr, sr;
            first->to_scale_rotation_translation(&fs, &fr, &ft);
            second->to_scale_rotation_translation(&ss, &sr, &st);
            return DMat4::from_scale_rotation_translation(fs.lerp(ss, *t), fr.lerp(sr, *t), ft.lerp(st, *t));
        }
        case Matrix::Sqrt: {
            // src/gui/matrix.rs:606-612,909-988 gets M with M*M ~ A from a BFGS minimiser (argmin, <= 60 iterations, accepted when
            // the squared residual is < 1e-4), i.e. a loose approximation whose digits belong to that solver.  Here: the exact
            // principal square root (Denman-Beavers), which is what that minimisation converges towards from its start M = A.
            auto a = eval_matrix(m.a);
            if (!a) return std::nullopt;
            bool ok = false;
            DMat4 root = a->sqrt_principal(&ok);
            if (!ok) return std::nullopt;  // "Can't calculate sqrt!"
            return root;
        }
    }
    return std::nullopt;
}

}  // namespace ptl
