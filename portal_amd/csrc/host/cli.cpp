// cli.cpp -- `portal-amd render-frame` and `portal-amd render`, the offline counterparts of the reference CLI
// (`portal render-frame <scene> ...` src/main.rs:2726-2755,2876-2946 and `portal render <scenes> [animations] ...`
// src/main.rs:2757-2874 + render_animation src/main.rs:1758-1873), writing PNG files instead of drawing to a
// window.  Everything goes through the C ABI (include/portal_amd.h); this file holds no rendering logic.
//
// Video pipeline (render): for every frame i of a clip, `motion_blur_frames` sub-frames are traced straight into
// device buffers (aa_start = j, time = i/count + j/blur/count*exposure), averaged on the GPU (ptl_average_images),
// downloaded once, and PNG-encoded on a pool of host threads while the GPU already traces the next frame.
#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/portal_amd.h"

namespace {

void usage() {
    std::fprintf(stderr,
                 "usage: portal-amd render-frame <scene.ron> [--stage NAME | --animation NAME] [--camera NAME] [--time T] [--output out.png]\n"
                 "                  [--width W] [--height H] [--aa-count N] [--render-depth D] [--device I] [--asset-root DIR] [--panini D --fov DEG]\n"
                 "                  [--gpus N | --devices a,b,..] [--transport stores|copy|rccl] [--multi-process]   one frame across the GPUs of a node\n"
                 "                  [--specialize 0] do NOT bake the scene state into the kernel   [--fast] tolerance mode   [--exact-cr] numerics contract 1   [--opt3] JIT at -O3 like the library default (render-frame: -O1)   [--timing] where the wall time went\n"
                 "       portal-amd precompile <scene.ron> [--stage NAME] [--specialize 0]      fill the code-object cache (no GPU needed)\n"
                 "       portal-amd render <scene[,scene..]> [clip[,clip..]] [--width 3840] [--height 2160] [--fps 60] [--motion-blur-frames 1]\n"
                 "                  [--stereoimage] [--batch-subframes 0|1] [--no-skip-existing] [--filter-starts-with P] [--aa-count 4] [--render-depth 150]\n"
                 "                  [--scenes-dir DIR] [--out-dir DIR] [--device I] [--shard K/N] [--max-frames N] [--asset-root DIR]\n"
                 "                  [--specialize 0|1]   bake what is constant within a clip into the kernel (default: when it pays)\n"
                 "                  [--fast]             tolerance mode for the whole clip (hardware rcp / sqrt, FMA contraction)\n"
                 "                  [--timing]           wait for every kernel and report GPU time (no host/GPU overlap)\n"
                 "       portal-amd emit-source <scene.ron> [--stage NAME]     print the generated HIP kernel source\n"
                 "       portal-amd check <scene.ron> [--stage NAME]           compile for gfx950 (no GPU needed); errors by scene element\n"
                 "       portal-amd write <scene.ron> [--stage NAME] [--set UNIFORM=VALUE ...] --output out.ron     the reference's RON writer\n"
                 "       portal-amd version\n");
}

std::vector<std::string> split_list(const std::string& s) {  // "a, b,,c" -> {a, b, c}
    std::vector<std::string> out;
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t comma = s.find(',', pos);
        if (comma == std::string::npos) comma = s.size();
        std::string item = s.substr(pos, comma - pos);
        size_t b = item.find_first_not_of(" \t"), e = item.find_last_not_of(" \t");
        if (b != std::string::npos) out.push_back(item.substr(b, e - b + 1));
        pos = comma + 1;
    }
    return out;
}

bool exists(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0;
}

void make_dirs(const std::string& path) {  // mkdir -p
    for (size_t p = 1; p <= path.size(); ++p)
        if (p == path.size() || path[p] == '/') ::mkdir(path.substr(0, p).c_str(), 0777);
}

std::string dir_of(const std::string& path) {
    size_t p = path.rfind('/');
    return p == std::string::npos ? "" : path.substr(0, p);
}

// The reference compiles its scene list in (src/gui/scenes.rs); here a scene is a path, or a bare name under --scenes-dir.
std::string scene_file(const std::string& arg, const std::string& scenes_dir) {
    bool looks_like_a_path = arg.find('/') != std::string::npos || (arg.size() > 4 && arg.compare(arg.size() - 4, 4, ".ron") == 0);
    if (exists(arg) || looks_like_a_path) return arg;
    return scenes_dir + "/" + arg + ".ron";
}

std::string scene_link(const std::string& path) {  // "dir/name.ron" -> "name"
    size_t slash = path.rfind('/');
    std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
    return base.size() > 4 && base.substr(base.size() - 4) == ".ron" ? base.substr(0, base.size() - 4) : base;
}

// Host threads that PNG-encode finished frames while the GPU traces the next ones.
class EncoderPool {
public:
    explicit EncoderPool(int threads, size_t max_pending) : max_pending_(max_pending) {
        for (int k = 0; k < threads; ++k) workers_.emplace_back([this] { run(); });
    }
    ~EncoderPool() { finish(); }
    void submit(std::function<void()> job) {
        std::unique_lock<std::mutex> lock(mu_);
        space_.wait(lock, [&] { return jobs_.size() < max_pending_; });
        jobs_.push_back(std::move(job));
        work_.notify_one();
    }
    void finish() {
        {
            std::unique_lock<std::mutex> lock(mu_);
            done_ = true;
        }
        work_.notify_all();
        for (auto& t : workers_)
            if (t.joinable()) t.join();
    }

private:
    void run() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lock(mu_);
                work_.wait(lock, [&] { return done_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                job = std::move(jobs_.front());
                jobs_.pop_front();
                space_.notify_one();
            }
            job();
        }
    }
    std::mutex mu_;
    std::condition_variable work_, space_;
    std::deque<std::function<void()>> jobs_;
    std::vector<std::thread> workers_;
    size_t max_pending_;
    bool done_ = false;
};

// Page-locked frame buffers recycled between the download and the encoder threads.
class PinnedFrames {
public:
    PinnedFrames(size_t bytes, int count) : bytes_(bytes) {
        for (int k = 0; k < count; ++k) {
            void* p = nullptr;
            if (ptl_host_alloc(bytes, &p) != PTL_OK) break;
            all_.push_back(p);
            free_.push_back(p);
        }
    }
    ~PinnedFrames() {
        for (void* p : all_) ptl_host_free(p);
    }
    bool ok() const { return !all_.empty(); }
    uint8_t* take() {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] { return !free_.empty(); });
        void* p = free_.back();
        free_.pop_back();
        return static_cast<uint8_t*>(p);
    }
    void give(uint8_t* p) {
        std::unique_lock<std::mutex> lock(mu_);
        free_.push_back(p);
        cv_.notify_one();
    }

private:
    size_t bytes_;
    std::vector<void*> all_, free_;
    std::mutex mu_;
    std::condition_variable cv_;
};

// No occupancy hint: with the basic VGPR allocator (kernel.cpp) a 4-waves bound makes the un-specialised portal_in_portal kernel
// spill (128 VGPRs + 240 B scratch: 2.19 ms against 1.52 ms at 4K, profiles/r02/variants1_prologue_waveloop_fast.jsonl); the
// other scenes do not care.  (Round 1, greedy allocator: the hint was a 18 % gain on that kernel.)
constexpr unsigned kRenderFlags = 0u;
// `render` (clips): the kernel a clip runs on when it gets no clip-constant build of its own still has the zero patterns of the scene's
// matrices and the mode switches compiled in (PTL_FLAG_SPECIALIZE_PATTERNS, bit 20: no value baked, so nothing moves under it but a
// pattern -- one rebuild per stage at most): 0.58 against 0.83 ms on the headline frame (profiles/r04/ab_bounded_snippets.jsonl `patterns`)
constexpr unsigned kClipFlags = kRenderFlags | (1u << 20);
// ... and, for a clip with motion blur, the slices entry (bit 22): the blur sub-frames of an output frame differ in their uniforms only and are
// traced by ONE launch (grid.z = sub-frame, a uniform block per slice), so the ramp and tail of a small frame overlap with its neighbours' instead
// of adding up: 1080p monoportal 0.0526 -> 0.0415 ms per sub-frame, 720p 0.0319 -> 0.0213, 4K aa 4 0.885 -> 0.861 (profiles/r04/concurrent_draws.jsonl)
constexpr unsigned kSlicesFlag = 1u << 22;
inline bool batch_subframes(int blur, int batch_option) { return batch_option != 0 && blur >= 2 && blur <= 16; }
// frames of a clip are intermediates (ffmpeg reads them, then anim/ is removed): fast deflate, 2.3x the encode rate of level 6
constexpr int kFrameDeflateLevel = 3;

struct Options {
    std::string scene, clips, output = "frame.png", asset_root = ".", stage, animation, camera, scenes_dir = "scenes", out_dir = ".", starts_with;
    bool have_camera = false, stereo = false, skip_existing = true;
    int batch = -1;       // render --batch-subframes 0|1: one launch for a frame's blur sub-frames (default: on where 2 <= blur <= 16)
    int concurrent = -1;  // render --concurrent-draws K: kernel instances in flight for a frame's blur sub-frames (default 1: measured, no gain)
    std::vector<std::pair<std::string, double>> sets;  // --set name=value
    bool timing = false;  // --timing: wait for every kernel and report GPU milliseconds (serialises host and GPU)
    int specialize = -1;  // -1 auto: clip-constant specialisation when the clip has enough sub-frames to repay the extra JIT
    int width = 1920, height = 1080, aa = 1, depth = 100, device = 0, fps = 60, blur = 1, shard = 0, shards = 1, max_frames = -1;
    double time = 0.0, panini = -1.0, fov = 90.0;
    // render-frame across GPUs: --gpus N (devices 0..N-1) or --devices a,b,.. ; --transport stores|copy|rccl ; --multi-process
    int gpus = 1, rank = 0, world = 1;
    std::string devices, transport = "stores", ipc_handle;
    bool multi_process = false, fast = false, exact_cr = false, opt3 = false;
    std::vector<std::string> argv;  // the command line as given (handed on to shard processes)
};

// SceneRenderer::update_inner_variables (src/main.rs:1688-1756): per-clip settings the reference hard-codes for its
// published videos.  Data, not logic: clip name -> what changes.
struct ClipOverride {
    const char* clip;
    int subspace_degree;  // 0 = leave
    int render_depth;     // 0 = leave
    int fps;              // 0 = leave
};
const ClipOverride kClipOverrides[] = {
    {"v2.face.2", 500, 0, 0},     {"v2.face.3", 500, 0, 0},     {"v2.face.4", 500, 0, 0},      {"v2.face.5", 500, 0, 0},
    {"v2.inside.1", 500, 0, 0},   {"v2.inside.3", 500, 0, 0},   {"v2.intro.1", 500, 0, 0},     {"v2.normal.2", 500, 0, 0},
    {"v2.normal.3", 500, 0, 0},   {"v2.rod.2", 500, 0, 0},      {"v2.rod.3", 500, 0, 0},       {"v2.spiral.3", 500, 0, 0},
    {"v2.spiral.4", 1000, 0, 0},  {"v2.spiral.5", 1000, 0, 0},  {"v2.spiral.6", 1000, 0, 0},   {"v2.spiral.7", 500, 0, 0},
    {"v2.spiral.9", 500, 0, 0},   {"v2.spaaaace.0", 500, 0, 0}, {"v4.golden.0", 500, 0, 0},    {"v4.golden.1", 500, 0, 0},
    {"v4.golden.2", 500, 0, 0},   {"v4.thumbnail.2", 500, 0, 0}, {"v2.rotated.0", 0, 100, 0},  {"v2.spiral.0", 0, 100, 0},
    {"v2.screenshot.5", 0, 100, 0}, {"v2.screenshot.6", 0, 100, 0}, {"v2.screenshot.3", 0, 0, 600},
};

void apply_clip_overrides(ptl_scene* scene, ptl_renderer* r, const std::string& clip, int* fps) {
    for (const ClipOverride& o : kClipOverrides) {
        if (clip != o.clip) continue;
        if (o.subspace_degree && scene) ptl_scene_set_uniform(scene, "subspace_degree", o.subspace_degree);  // no such uniform: nothing happens
        if (o.render_depth && r) ptl_renderer_set_option(r, "render_depth", o.render_depth);
        if (o.fps && fps) *fps = o.fps;
    }
}

int fail(const char* what) {
    std::fprintf(stderr, "%s: %s\n", what, ptl_last_error());
    return 1;
}

// Everything of `render-frame` between creating a renderer and drawing (src/main.rs:2893-2928): stage / clip, camera, options,
// one SceneRenderer::update at --time.  `scene_state` = false for ranks 1.. of a multi-GPU frame, which share rank 0's scene.
int setup_renderer(const Options& o, ptl_scene* scene, ptl_renderer* r, bool scene_state) {
    ptl_renderer_set_option(r, "aa_count", o.aa);
    ptl_renderer_set_option(r, "render_depth", o.depth);
    char stage_cam[256] = "";
    if (scene_state && !o.stage.empty() && ptl_scene_init_stage(scene, o.stage.c_str(), stage_cam, sizeof stage_cam) != PTL_OK) {  // src/main.rs:2900-2904
        std::fprintf(stderr, "Scene `%s` has no stage named `%s`\n", o.scene.c_str(), o.stage.c_str());
        return 1;
    }
    if (!o.animation.empty()) {  // src/main.rs:2905-2917
        if (scene_state && ptl_scene_init_animation(scene, o.animation.c_str()) != PTL_OK) {
            std::fprintf(stderr, "Scene `%s` has no animation named `%s`\n", o.scene.c_str(), o.animation.c_str());
            return 1;
        }
        apply_clip_overrides(scene_state ? scene : nullptr, r, o.animation, nullptr);
    }
    if (o.have_camera && ptl_renderer_use_camera(r, o.camera.c_str()) != PTL_OK) {  // --camera wins (src/main.rs:2918-2926)
        std::fprintf(stderr, "Scene `%s` has no camera named `%s`\n", o.scene.c_str(), o.camera.c_str());
        return 1;
    }
    if (o.panini >= 0.0) {
        ptl_renderer_set_option(r, "use_panini_projection", 1);
        ptl_renderer_set_option(r, "panini_param", o.panini);
    }
    ptl_renderer_set_option(r, "view_angle", o.fov / 180.0 * 3.14159265358979323846);
    if (ptl_renderer_update(r, o.time, nullptr, nullptr) != PTL_OK) return fail("update");  // src/main.rs:2928
    return 0;
}

unsigned frame_flags(const Options& o) {
    unsigned f = kRenderFlags;
    // One frame of one scene state: baking the state in is the cheaper build (0.74 s against 1.08 s of hiprtc for the headline scene:
    // the folded source is smaller) AND the faster kernel (0.53 against 1.31 ms), so it is the default; --specialize 0 keeps every
    // scene uniform a run-time value (profiles/r02/render_frame_e2e.log).
    if (o.specialize != 0) f |= 1u | 4u;
    if (o.fast) f |= 64u;                // --fast: tolerance mode (PTL_FLAG_FAST_MATH)
    if (o.exact_cr) f |= 16384u;         // --exact-cr: numerics contract 1 (PTL_FLAG_EXACT_CR)
    // ONE frame: the wall time is the JIT's, not the kernel's (profiles/r03/render_frame_e2e.log: 2.4 s of -O3 hiprtc for a 0.33 ms kernel, 1.2 s
    // of -O1 for a 0.36 ms one) -- unless the caller wants the shipped optimisation level (--opt3), e.g. to fill the cache for a bench
    if (!o.opt3) f |= 262144u;           // PTL_FLAG_QUICK_JIT
    return f;
}

double seconds_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

std::string hex_of(const unsigned char* bytes, size_t n) {
    static const char* digits = "0123456789abcdef";
    std::string out;
    for (size_t k = 0; k < n; ++k) {
        out += digits[bytes[k] >> 4];
        out += digits[bytes[k] & 15];
    }
    return out;
}
bool unhex(const std::string& hex, unsigned char* out, size_t n) {
    if (hex.size() != 2 * n) return false;
    auto val = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1; };
    for (size_t k = 0; k < n; ++k) {
        int hi = val(hex[2 * k]), lo = val(hex[2 * k + 1]);
        if (hi < 0 || lo < 0) return false;
        out[k] = (unsigned char)(hi * 16 + lo);
    }
    return true;
}

std::vector<int> frame_devices(const Options& o) {
    std::vector<int> devices;
    for (auto& d : split_list(o.devices)) devices.push_back(std::atoi(d.c_str()));
    if (devices.empty())
        for (int k = 0; k < std::max(1, o.gpus); ++k) devices.push_back(o.gpus > 1 ? k : o.device);
    return devices;
}

// `render-frame --gpus N --multi-process`: rank 0 (this process) owns the frame in its GPU's memory and exports it (HIP IPC);
// ranks 1.. are child processes (`portal-amd render-shard`, the same command line + rank / device / handle) that map it and
// let their kernels store their row blocks straight into it over xGMI.  Completion = the children's exit.
int render_frame_processes(const Options& o, const std::vector<int>& devices) {
    const int n = (int)devices.size();
    auto t0 = std::chrono::steady_clock::now();
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(o.scene.c_str(), &scene) != PTL_OK) {
        std::fprintf(stderr, "Failed to parse scene `%s`: %s\n", o.scene.c_str(), ptl_last_error());
        return 1;
    }
    size_t bytes = (size_t)o.width * o.height * 4;
    void* frame = nullptr;
    unsigned char handle[PTL_IPC_HANDLE_BYTES];
    if (ptl_device_alloc(devices[0], bytes, &frame) != PTL_OK || ptl_ipc_export(frame, handle) != PTL_OK) return fail("frame buffer / ipc export");
    std::vector<pid_t> children;
    // every failure path below reaps the shard processes already forked: they render into `frame`, which dies with this process
    auto reap = [&](int rc) {
        for (pid_t pid : children) {
            int status = 0;
            ::waitpid(pid, &status, 0);
        }
        children.clear();
        return rc;
    };
    for (int k = 1; k < n; ++k) {
        std::vector<std::string> args = o.argv;
        args[1] = "render-shard";
        for (auto extra : {std::string("--rank"), std::to_string(k), std::string("--world"), std::to_string(n), std::string("--device"),
                           std::to_string(devices[k]), std::string("--ipc-handle"), hex_of(handle, sizeof handle)})
            args.push_back(extra);
        pid_t pid = ::fork();
        if (pid == 0) {
            std::vector<char*> argv;
            for (auto& a : args) argv.push_back(const_cast<char*>(a.c_str()));
            argv.push_back(nullptr);
            ::execv("/proc/self/exe", argv.data());
            std::perror("execv");
            ::_exit(127);
        }
        if (pid < 0) return reap(fail("fork"));
        children.push_back(pid);
    }
    std::vector<char> log(1 << 16);
    ptl_renderer* r = nullptr;
    if (ptl_renderer_create(scene, devices[0], o.asset_root.c_str(), frame_flags(o), &r, log.data(), log.size()) != PTL_OK) {
        std::fprintf(stderr, "renderer: %s\n%s\n", ptl_last_error(), log.data());
        return reap(1);
    }
    if (int rc = setup_renderer(o, scene, r, true)) return reap(rc);
    ptl_frame f{o.width, o.height, 0, n, 1};
    float ms = 0.0f;
    if (ptl_renderer_draw(r, &f, frame, nullptr, nullptr, nullptr, &ms) != PTL_OK) return reap(fail("render"));
    int failed = 0;
    for (pid_t pid : children) {
        int status = 0;
        if (::waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) ++failed;
    }
    if (failed) {
        std::fprintf(stderr, "%d of %d shard processes failed\n", failed, n - 1);
        return 1;
    }
    std::vector<uint8_t> img(bytes);
    if (ptl_device_download(img.data(), frame, bytes, nullptr) != PTL_OK) return fail("download");
    if (!dir_of(o.output).empty()) make_dirs(dir_of(o.output));
    if (ptl_png_write(o.output.c_str(), img.data(), o.width, o.height) != PTL_OK) return fail("png");
    std::printf("Rendered `%s` to `%s` (%dx%d, aa %d, depth %d) with %d processes, one per GPU, storing into rank 0's frame (HIP IPC): rank 0 kernel %.3f ms; total %.2f s\n",
                o.scene.c_str(), o.output.c_str(), o.width, o.height, o.aa, o.depth, n, ms, seconds_since(t0));
    ptl_renderer_destroy(r);
    ptl_device_free(frame);
    ptl_scene_free(scene);
    return 0;
}

// One of those child processes.
int render_shard(const Options& o) {
    unsigned char handle[PTL_IPC_HANDLE_BYTES];
    if (o.world < 2 || o.rank < 1 || o.rank >= o.world || !unhex(o.ipc_handle, handle, sizeof handle)) {
        std::fprintf(stderr, "render-shard is started by `render-frame --gpus N --multi-process`\n");
        return 2;
    }
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(o.scene.c_str(), &scene) != PTL_OK) return fail("scene");
    std::vector<char> log(1 << 16);
    ptl_renderer* r = nullptr;
    if (ptl_renderer_create(scene, o.device, o.asset_root.c_str(), frame_flags(o), &r, log.data(), log.size()) != PTL_OK) {
        std::fprintf(stderr, "renderer: %s\n%s\n", ptl_last_error(), log.data());
        return 1;
    }
    if (int rc = setup_renderer(o, scene, r, true)) return rc;
    void* frame = nullptr;
    if (ptl_ipc_open(o.device, handle, &frame) != PTL_OK) return fail("ipc open");
    ptl_frame f{o.width, o.height, o.rank, o.world, 1};
    float ms = 0.0f;
    if (ptl_renderer_draw(r, &f, frame, nullptr, nullptr, nullptr, &ms) != PTL_OK) return fail("render");  // timed: returns when the kernel (and its stores) have completed
    ptl_ipc_close(frame);
    ptl_renderer_destroy(r);
    ptl_scene_free(scene);
    return 0;
}

int render_frame(const Options& o) {
    std::vector<int> devices = frame_devices(o);
    if (devices.size() > 1 && o.multi_process) return render_frame_processes(o, devices);
    auto t0 = std::chrono::steady_clock::now();
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(o.scene.c_str(), &scene) != PTL_OK) {
        std::fprintf(stderr, "Failed to parse scene `%s`: %s\n", o.scene.c_str(), ptl_last_error());
        return 1;
    }
    double t_load = seconds_since(t0);
    std::vector<char> log(1 << 16);
    std::vector<uint8_t> img((size_t)o.width * o.height * 4);
    if (devices.size() > 1 || o.transport == "rccl") {  // one process, one renderer per GPU (include/portal_amd.h layer 3); `--transport rccl` also with one
        ptl_frame_group* g = nullptr;
        int transport = o.transport == "copy" ? PTL_GROUP_COPY_GATHER : o.transport == "rccl" ? PTL_GROUP_RCCL_GATHER : PTL_GROUP_PEER_STORES;
        if (ptl_frame_group_create(scene, devices.data(), (int)devices.size(), o.asset_root.c_str(), frame_flags(o), transport, &g, log.data(), log.size()) != PTL_OK) {
            std::fprintf(stderr, "frame group: %s\n%s\n", ptl_last_error(), log.data());
            return 1;
        }
        double t_build = seconds_since(t0);
        for (int k = 0; k < ptl_frame_group_size(g); ++k)
            if (int rc = setup_renderer(o, scene, ptl_frame_group_renderer(g, k), k == 0)) return rc;
        std::vector<float> ms(devices.size(), 0.0f);
        auto t_draw0 = std::chrono::steady_clock::now();
        if (ptl_frame_group_draw(g, o.width, o.height, nullptr, ms.data()) != PTL_OK) return fail("render");
        double draw_ms = seconds_since(t_draw0) * 1e3;
        if (ptl_frame_group_download(g, img.data()) != PTL_OK) return fail("download");
        if (!dir_of(o.output).empty()) make_dirs(dir_of(o.output));
        if (ptl_png_write(o.output.c_str(), img.data(), o.width, o.height) != PTL_OK) return fail("png");
        std::string per_rank;
        for (float m : ms) per_rank += (per_rank.empty() ? "" : " ") + std::to_string(m).substr(0, 6);
        std::printf("Rendered `%s` to `%s` (%dx%d, aa %d, depth %d) on %zu GPUs (%s): kernel ms per rank [%s], frame %.3f ms wall; build %.2f s, total %.2f s\n",
                    o.scene.c_str(), o.output.c_str(), o.width, o.height, o.aa, o.depth, devices.size(),
                    transport == PTL_GROUP_COPY_GATHER ? "packed shards + one strided peer copy each" : transport == PTL_GROUP_RCCL_GATHER ? "packed shards + one RCCL gather" : "kernels store into GPU 0's frame", per_rank.c_str(), draw_ms,
                    t_build - t_load, seconds_since(t0));
        ptl_frame_group_destroy(g);
        ptl_scene_free(scene);
        return 0;
    }
    ptl_renderer* r = nullptr;
    if (ptl_renderer_create(scene, devices[0], o.asset_root.c_str(), frame_flags(o), &r, log.data(), log.size()) != PTL_OK) {
        std::fprintf(stderr, "renderer: %s\n%s\n", ptl_last_error(), log.data());
        return 1;
    }
    double t_build = seconds_since(t0);
    if (int rc = setup_renderer(o, scene, r, true)) return rc;
    ptl_frame frame{o.width, o.height, 0, 1, 0};
    float ms = 0.0f;
    if (ptl_renderer_draw_to_host(r, &frame, img.data(), nullptr, nullptr, &ms) != PTL_OK) return fail("render");
    double t_draw = seconds_since(t0);
    if (!dir_of(o.output).empty()) make_dirs(dir_of(o.output));
    if (ptl_png_write(o.output.c_str(), img.data(), o.width, o.height) != PTL_OK) return fail("png");
    double total = seconds_since(t0);
    std::printf("Rendered `%s` to `%s` (%dx%d, aa %d, depth %d): kernel %.3f ms, %.1f Mray/s; total %.2f s\n", o.scene.c_str(), o.output.c_str(),
                o.width, o.height, o.aa, o.depth, ms, (double)o.width * o.height * o.aa / (ms * 1e3), total);
    if (o.timing)  // where the wall time went: the JIT (or the code-object cache) dominates a single frame
        std::printf("timing: scene load %.3f s, generate + compile/load kernel %.3f s, update + draw + download %.3f s, png %.3f s\n", t_load,
                    t_build - t_load, t_draw - t_build, total - t_draw);
    ptl_renderer_destroy(r);
    ptl_scene_free(scene);
    return 0;
}

// `portal-amd precompile <scene>`: fill the code-object cache for a scene without a GPU (hiprtc only): the dynamic-uniform kernel
// `render-frame` / `render` start with and, with --specialize 1, the kernel with the scene's current state baked in.  A later
// run on a GPU box with the same toolchain finds them by source + option + toolchain hash and only loads them.
int precompile(const Options& o) {
    auto t0 = std::chrono::steady_clock::now();
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(o.scene.c_str(), &scene) != PTL_OK) {
        std::fprintf(stderr, "Failed to parse scene `%s`: %s\n", o.scene.c_str(), ptl_last_error());
        return 1;
    }
    char stage_cam[256] = "";
    if (!o.stage.empty() && ptl_scene_init_stage(scene, o.stage.c_str(), stage_cam, sizeof stage_cam) != PTL_OK) return fail("stage");
    std::vector<char> log(1 << 16);
    std::vector<unsigned> variants = {frame_flags(o)};
    if (o.specialize != 0) variants.push_back(kClipFlags | (batch_subframes(o.blur, o.batch) ? kSlicesFlag : 0u) | (o.fast ? 64u : 0u) | (o.exact_cr ? 16384u : 0u) | (o.opt3 ? 0u : 262144u));  // + the dynamic-uniform kernel `render` starts clips with
    for (unsigned flags : variants) {
        auto t1 = std::chrono::steady_clock::now();
        ptl_renderer* r = nullptr;
        if (ptl_renderer_create(scene, -1, o.asset_root.c_str(), flags, &r, log.data(), log.size()) != PTL_OK) {
            std::fprintf(stderr, "compile: %s\n%s\n", ptl_last_error(), log.data());
            return 1;
        }
        const void* code = nullptr;
        size_t size = 0;
        ptl_kernel_code_object(ptl_renderer_kernel(r), &code, &size);
        std::printf("flags 0x%x: %zu B code object in %.2f s\n", flags, size, seconds_since(t1));
        ptl_renderer_destroy(r);
    }
    std::printf("precompiled `%s` in %.2f s\n", o.scene.c_str(), seconds_since(t0));
    ptl_scene_free(scene);
    return 0;
}

// What the clip loop keeps in flight: the download of frame i runs on its own stream into page-locked memory while frame i+1
// is being traced; `kRing` result buffers so a frame is not overwritten before its copy has left.
struct FramePipeline {
    static constexpr int kRing = 3;
    int device = 0;
    void* copy_stream = nullptr;
    void* results[kRing] = {nullptr, nullptr, nullptr};   // device RGBA8 frames ready for download
    void* copied[kRing] = {nullptr, nullptr, nullptr};    // event: the copy out of results[k] has finished
    bool copy_pending[kRing] = {false, false, false};
    void* produced = nullptr;                             // event: results[k] is complete on the tracing stream

    bool create(int dev, size_t bytes) {
        device = dev;
        if (ptl_stream_create(dev, &copy_stream) != PTL_OK || ptl_event_create(dev, &produced) != PTL_OK) return false;
        for (int k = 0; k < kRing; ++k)
            if (ptl_device_alloc(dev, bytes, &results[k]) != PTL_OK || ptl_event_create(dev, &copied[k]) != PTL_OK) return false;
        return true;
    }
    ~FramePipeline() {
        if (copy_stream) ptl_stream_destroy(copy_stream);
        if (produced) ptl_event_destroy(produced);
        for (int k = 0; k < kRing; ++k) {
            if (copied[k]) ptl_event_destroy(copied[k]);
            if (results[k]) ptl_device_free(results[k]);
        }
    }
};

// render_animation (src/main.rs:1758-1873)
int render_clip(const Options& o, ptl_scene* scene, ptl_renderer* r, const std::string& scene_name, const std::string& clip, double duration, int fps,
                int width, int height, std::vector<void*>& subframes, FramePipeline& pipe, EncoderPool& pool, PinnedFrames& pinned) {
    auto started = std::chrono::steady_clock::now();
    int rejits_before = ptl_renderer_rejit_count(r);
    std::string video_base = o.out_dir + "/video/" + scene_name + "/" + clip;
    if (o.skip_existing && exists(video_base + ".mov")) {
        std::printf("Skip `%s/%s`, because it's already exists\n", scene_name.c_str(), clip.c_str());
        return 0;
    }
    make_dirs(dir_of(video_base));
    std::string anim_dir = o.out_dir + "/anim";
    make_dirs(anim_dir);
    int count = std::max(1, (int)((float)duration * (float)fps));  // ((duration_seconds * fps as f32) as usize).max(1)
    const double exposure = 0.5;
    size_t frame_bytes = (size_t)width * height * 4;
    double gpu_ms = 0.0;
    long traced = 0, drawn_frames = 0;
    ptl_frame frame{width, height, 0, 1, 0};
    int last = o.max_frames >= 0 ? std::min(count, o.max_frames) : count;
    for (int i = 0; i < last; ++i) {
        std::string name = anim_dir + "/frame_" + std::to_string(i) + ".png";
        if (i % o.shards != o.shard || exists(name)) {
            // Not ours (shard K of N takes every N-th frame) or already on disk.  The camera is stateful -- where it is relative
            // to the portals depends on the path it took (teleport_camera) -- so the host step still runs for every sub-frame:
            // a shard, or a resumed run, then sees exactly the cameras of an uninterrupted run.  (The reference skips the
            // update as well, src/main.rs:1789-1792, and so renders a resumed clip from a different camera history.)
            for (int j = 0; j < o.blur; ++j) {
                double t = ((double)i / count) + (double)j / o.blur / count * exposure;
                if (ptl_renderer_update(r, t * (double)(float)duration, nullptr, nullptr) != PTL_OK) return fail("update");
            }
            continue;
        }
        int slot = (int)(drawn_frames++ % FramePipeline::kRing);
        if (pipe.copy_pending[slot] && ptl_stream_wait_event(nullptr, pipe.copied[slot]) != PTL_OK) return fail("wait");  // GPU-side: slot is free
        const bool batched = batch_subframes(o.blur, o.batch);
        for (int j = 0; j < o.blur; ++j) {
            double t = ((double)i / count) + (double)j / o.blur / count * exposure;
            ptl_renderer_set_option(r, "aa_start", j);
            if (ptl_renderer_update(r, t * (double)(float)duration, nullptr, nullptr) != PTL_OK) return fail("update");
            void* target = o.blur > 1 ? subframes[j] : pipe.results[slot];  // one image: average_images hands it back untouched
            float ms = 0.0f;
            if (batched) {
                // everything a draw does short of launching; the launch follows behind the last sub-frame, once for all of them
                if (ptl_renderer_stage_slice(r, &frame, j) != PTL_OK) return fail("stage");
                if (j == o.blur - 1 && ptl_renderer_draw_slices(r, &frame, o.blur, subframes[0], nullptr, (unsigned long long)width * height, nullptr, o.timing ? &ms : nullptr) != PTL_OK)
                    return fail("render");
            } else if (ptl_renderer_draw(r, &frame, target, nullptr, nullptr, nullptr, o.timing ? &ms : nullptr) != PTL_OK) {
                // (without --timing the launch is not waited for: the host evaluates the next sub-frame's uniforms while this one traces)
                return fail("render");
            }
            gpu_ms += ms;
            ++traced;
            bool first = i == 0 && j == 0, final_one = i == count - 1 && j == o.blur - 1;
            if (batched) first = i == 0 && j == o.blur - 1;  // (the stills are read behind the launch: sub-frame 0 of the first frame, the last of the last)
            if (first || final_one) {  // the clip's .start.png / .end.png stills: same pool, same pinned buffers
                if (ptl_renderer_join(r, nullptr) != PTL_OK) return fail("join");  // (the download below is on the default stream)
                for (int which = 0; which < 2; ++which) {
                    if (!(which == 0 ? first : final_one)) continue;
                    uint8_t* still = pinned.take();
                    if (ptl_device_download(still, (batched && which == 0) ? subframes[0] : target, frame_bytes, nullptr) != PTL_OK) return fail("download");
                    std::string still_name = video_base + (which == 0 ? ".start.png" : ".end.png");
                    pool.submit([still, still_name, width, height, &pinned] {
                        if (ptl_png_write(still_name.c_str(), still, width, height) != PTL_OK) std::fprintf(stderr, "\n%s\n", ptl_last_error());
                        pinned.give(still);
                    });
                }
            }
        }
        if (ptl_renderer_join(r, nullptr) != PTL_OK) return fail("join");  // the default stream goes on behind every sub-frame of this frame
        if (o.blur > 1) {
            float ms = 0.0f;
            if (ptl_average_images(o.device, subframes.data(), o.blur, pipe.results[slot], width, height, nullptr, o.timing ? &ms : nullptr) != PTL_OK)
                return fail("average_images");
            gpu_ms += ms;
        }
        // hand the finished frame to the copy stream and go on tracing; the encoder job waits for its own event
        uint8_t* pixels = pinned.take();  // blocks while every buffer is still being encoded
        void* arrived = nullptr;
        if (ptl_event_record(pipe.produced, nullptr) != PTL_OK || ptl_stream_wait_event(pipe.copy_stream, pipe.produced) != PTL_OK ||
            ptl_device_download_async(pixels, pipe.results[slot], frame_bytes, pipe.copy_stream) != PTL_OK ||
            ptl_event_record(pipe.copied[slot], pipe.copy_stream) != PTL_OK || ptl_event_create(pipe.device, &arrived) != PTL_OK ||
            ptl_event_record(arrived, pipe.copy_stream) != PTL_OK)
            return fail("download");
        pipe.copy_pending[slot] = true;
        pool.submit([pixels, name, width, height, arrived, &pinned] {
            if (ptl_event_synchronize(arrived) != PTL_OK || ptl_png_write_level(name.c_str(), pixels, width, height, kFrameDeflateLevel) != PTL_OK)
                std::fprintf(stderr, "\n%s\n", ptl_last_error());
            ptl_event_destroy(arrived);
            pinned.give(pixels);
        });
        std::printf("\r%d/%d done      ", i, count);
        std::fflush(stdout);
    }
    std::printf("\n");
    double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count();
    if (o.timing)
        std::printf("Traced `%s/%s`: %ld sub-frames %dx%d, GPU %.1f ms (%.3f ms each), submitted after %.2f s, kernel rebuilt %d times\n", scene_name.c_str(),
                    clip.c_str(), traced, width, height, gpu_ms, traced ? gpu_ms / traced : 0.0, wall, ptl_renderer_rejit_count(r) - rejits_before);
    else
        std::printf("Traced `%s/%s`: %ld sub-frames %dx%d submitted after %.2f s, kernel rebuilt %d times\n", scene_name.c_str(), clip.c_str(), traced, width,
                    height, wall, ptl_renderer_rejit_count(r) - rejits_before);
    (void)scene;
    return 0;
}

// Run a program without a shell (arguments are passed as they are: no quoting rules to get wrong).  -1 = could not start.
int run_program(const std::vector<std::string>& argv, bool quiet) {
    pid_t pid = fork();
    if (pid < 0) return -1;
    if (pid == 0) {
        if (quiet) {
            int null_fd = open("/dev/null", O_WRONLY);
            if (null_fd >= 0) {
                dup2(null_fd, 1);
                dup2(null_fd, 2);
            }
        }
        std::vector<char*> args;
        for (const std::string& a : argv) args.push_back(const_cast<char*>(a.c_str()));
        args.push_back(nullptr);
        execvp(args[0], args.data());
        _exit(127);
    }
    int status = 0;
    if (waitpid(pid, &status, 0) < 0) return -1;
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

void remove_tree(const std::string& path) {  // rm -rf of a directory we created ourselves (frames only, one level)
    if (DIR* d = opendir(path.c_str())) {
        while (dirent* e = readdir(d)) {
            std::string name = e->d_name;
            if (name != "." && name != "..") ::unlink((path + "/" + name).c_str());
        }
        closedir(d);
    }
    ::rmdir(path.c_str());
}

// the reference's ffmpeg hand-off (src/main.rs:1829-1869), same arguments; frames are kept when there is no ffmpeg
int encode_video(const Options& o, const std::string& scene_name, const std::string& clip, int fps) {
    std::string anim = o.out_dir + "/anim", video = o.out_dir + "/video/" + scene_name + "/" + clip + ".mov";
    if (run_program({"ffmpeg", "-version"}, true) != 0) {
        // no encoder on this machine: park the clip's frames next to where the video would be, so the next clip starts
        // from an empty anim/ (the reference removes anim/ after ffmpeg; frame_%d.png of another clip would be "existing")
        std::string frames = o.out_dir + "/video/" + scene_name + "/" + clip + ".frames";
        remove_tree(frames);
        if (::rename(anim.c_str(), frames.c_str()) != 0) std::fprintf(stderr, "could not move anim/ to %s\n", frames.c_str());
        std::printf("ffmpeg not found: frames kept in `%s` (ffmpeg -framerate %d -i frame_%%d.png ... ../%s.mov)\n", frames.c_str(), fps, clip.c_str());
        return 0;
    }
    std::printf("Start ffmpeg to render video\n");
    auto started = std::chrono::steady_clock::now();
    int status = run_program(
        {"ffmpeg", "-framerate", std::to_string(fps), "-i", anim + "/frame_%d.png", "-vf",
         "zscale=primariesin=bt709:transferin=iec61966-2-1:matrixin=bt709:rangein=full:primaries=bt709:transfer=iec61966-2-1:matrix=bt709:range=full,"
         "format=yuv420p10le",
         "-c:v", "libx265", "-pix_fmt", "yuv420p10le", "-crf", "15", "-preset", "slow", "-x265-params",
         "colorprim=bt709:transfer=iec61966-2-1:colormatrix=bt709:range=full", "-colorspace", "bt709", "-color_primaries", "bt709", "-color_trc",
         "iec61966-2-1", "-color_range", "pc", "-movflags", "+write_colr+faststart", "-tag:v", "hvc1", "-y", video},
        true);
    std::printf("ffmpeg status: %d\nffmpeg time: %.2f s\n", status, std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count());
    remove_tree(anim);  // like the reference, whatever ffmpeg said (src/main.rs:1860)
    return 0;
}

// Warm the code-object cache for the NEXT clip's specialised kernel while the current clip renders: a private copy of the scene
// is taken through the same history (every clip initialised so far, with its overrides), then compiled for gfx950 without a
// device.  When the main thread gets to that clip it generates the same source and finds the binary on disk; if the histories
// ever disagree it just compiles as before.
void prefetch_clip_kernel(std::string path, std::vector<std::string> history, std::string asset_root, unsigned extra_flags, bool stereo) {  // extra_flags: --fast / --exact-cr / slices
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(path.c_str(), &scene) != PTL_OK) return;
    for (const std::string& clip : history) {
        if (ptl_scene_init_animation(scene, clip.c_str()) != PTL_OK) {
            ptl_scene_free(scene);
            return;
        }
        apply_clip_overrides(scene, nullptr, clip, nullptr);
    }
    // (the mode switches are compiled into a specialised kernel: the compile-only renderer must have the ones the clip is drawn with)
    const char* names[] = {"draw_side_by_side"};
    const double values[] = {stereo ? 1.0 : 0.0};
    ptl_renderer* r = nullptr;
    if (ptl_renderer_create_with_options(scene, -1, asset_root.c_str(), kClipFlags | 8u | extra_flags, names, values, 1, &r, nullptr, 0) == PTL_OK) {
        ptl_renderer_prebuild_teleport(r);  // the camera of a clip moves: its teleport queries need the other half of the build as well
        ptl_renderer_destroy(r);
    }
    ptl_scene_free(scene);
}

int render(const Options& o) {
    int width = o.stereo ? o.width * 2 : o.width;  // src/main.rs:2822-2829
    auto total_start = std::chrono::steady_clock::now();
    for (const std::string& scene_arg : split_list(o.scene)) {
        std::string path = scene_file(scene_arg, o.scenes_dir), scene_name = scene_link(path);
        std::printf("Rendering scene %s\n", scene_name.c_str());
        ptl_scene* scene = nullptr;
        if (ptl_scene_load_file(path.c_str(), &scene) != PTL_OK) {
            std::fprintf(stderr, "Failed to parse scene `%s`: %s\n", scene_name.c_str(), ptl_last_error());
            return 1;
        }
        // which clips: the named ones (render_named_animations) or all, optionally filtered (render_all_animations)
        std::vector<std::pair<std::string, double>> clips;
        char name[256];
        double duration = 0.0;
        for (int k = 0; ptl_scene_animation(scene, k, name, sizeof name, &duration) == PTL_OK; ++k) clips.emplace_back(name, duration);
        std::vector<std::pair<std::string, double>> todo;
        if (!o.clips.empty()) {
            for (const std::string& want : split_list(o.clips)) {
                auto it = std::find_if(clips.begin(), clips.end(), [&](auto& c) { return c.first == want; });
                if (it == clips.end()) {
                    std::fprintf(stderr, "Scene `%s` has no animation named `%s`\n", scene_name.c_str(), want.c_str());
                    return 1;
                }
                todo.push_back(*it);
            }
        } else {
            for (auto& c : clips)
                if (o.starts_with.empty() || c.first.compare(0, o.starts_with.size(), o.starts_with) == 0) todo.push_back(c);
        }
        int threads = (int)std::min(64u, std::max(2u, std::thread::hardware_concurrency() * 3 / 4));
        // Specialised kernels of the clips to come are compiled ahead by a few background threads (in clip order), so a run of
        // many short clips is not a run of JIT waits; the main thread only waits if it reaches a clip before its binary is ready.
        struct Prefetcher {
            std::vector<std::thread> workers;
            std::mutex mu;
            std::condition_variable cv;
            std::vector<char> done;
            size_t next = 0;
            bool stop = false;
            ~Prefetcher() {
                {
                    std::unique_lock<std::mutex> lock(mu);
                    stop = true;
                }
                for (auto& t : workers)
                    if (t.joinable()) t.join();
            }
        } pf;
        pf.done.assign(todo.size(), 0);
        // which clips get a specialised kernel: it repays its extra JIT (~1 s) only on a clip with enough work
        std::vector<char> specialise(todo.size(), 0);
        for (size_t k = 0; k < todo.size(); ++k) {
            int fps = o.fps;
            apply_clip_overrides(nullptr, nullptr, todo[k].first, &fps);
            int count = std::max(1, (int)((float)todo[k].second * (float)fps));
            double samples = (double)width * o.height * o.aa * count * o.blur;
            specialise[k] = o.specialize >= 0 ? o.specialize != 0 : samples >= 1e10;
        }
        std::vector<char> log(1 << 16);
        ptl_renderer* r = nullptr;
        // The un-baked kernel (every scene uniform a run-time value) is wanted NOW when a clip starts on it: the quick build (bit 18; the
        // library ignores it for the clip-constant kernels "specialize_static" asks for, which stay at -O3).  When the first clip gets a
        // clip-constant kernel anyway, the renderer is created on that one directly (the scene taken into the clip first, as the clip loop
        // and prefetch_clip_kernel do) instead of building an un-baked kernel nothing would run on.  profiles/r03/video_*.log
        bool start_baked = !todo.empty() && specialise[0];
        if (start_baked) {
            if (ptl_scene_init_animation(scene, todo[0].first.c_str()) != PTL_OK) return fail("init_animation");
            apply_clip_overrides(scene, nullptr, todo[0].first, nullptr);
        }
        unsigned start_flags = kClipFlags | (batch_subframes(o.blur, o.batch) ? kSlicesFlag : 0u) | (o.fast ? 64u : 0u) | (o.exact_cr ? 16384u : 0u) | (o.opt3 ? 0u : 262144u) | (start_baked ? 8u : 0u);
        const char* create_names[] = {"aa_count", "render_depth", "draw_side_by_side"};  // before the first build: a baked kernel has its mode switches compiled in
        const double create_values[] = {(double)o.aa, (double)o.depth, o.stereo ? 1.0 : 0.0};
        if (ptl_renderer_create_with_options(scene, o.device, o.asset_root.c_str(), start_flags, create_names, create_values, 3, &r, log.data(), log.size()) !=
            PTL_OK) {  // --fast: tolerance mode for the whole clip
            std::fprintf(stderr, "renderer: %s\n%s\n", ptl_last_error(), log.data());
            return 1;
        }
        ptl_renderer_set_option(r, "aa_count", o.aa);
        ptl_renderer_set_option(r, "render_depth", o.depth);
        ptl_renderer_set_option(r, "draw_side_by_side", o.stereo ? 1 : 0);
        // The blur sub-frames of one output frame differ in their uniforms only; with several kernel instances in flight (each has a uniform
        // block of its own, "concurrent_draws") consecutive sub-frames need not wait for the one block of a module (the reference re-draws with
        // `_aa_start` windows one after the other, src/main.rs:1798).  Measured (tools/concurrent_draws.py, profiles/r04/concurrent_draws.jsonl):
        // identical frames and NO gain -- 1080p monoportal 0.0519 ms per sub-frame with one instance, 0.0522 with two, 0.0559 with four; 720p 0.031
        // -> 0.039; 4K aa 4 0.884 -> 0.885 / 0.908: the cross-stream event waits cost what the overlapped tails save.  So: off unless asked for.
        const int lanes = o.concurrent >= 1 ? std::min(8, o.concurrent) : 1;
        if (ptl_renderer_set_option(r, "concurrent_draws", lanes) != PTL_OK) return fail("concurrent_draws");
        std::vector<void*> subframes(std::max(1, o.blur), nullptr);
        size_t bytes = (size_t)width * o.height * 4;
        // ONE allocation, sub-frame j at j * bytes: what the one-launch form writes (slice z behind slice z - 1)
        void* subframe_block = nullptr;
        if (ptl_device_alloc(o.device, bytes * subframes.size(), &subframe_block) != PTL_OK) return fail("alloc");
        for (size_t j = 0; j < subframes.size(); ++j) subframes[j] = static_cast<char*>(subframe_block) + j * bytes;
        FramePipeline pipe;
        if (!pipe.create(o.device, bytes)) return fail("pipeline");

        if (o.specialize != 0 && todo.size() > 1) {
            int n_workers = (int)std::min<size_t>({(size_t)6, todo.size() - 1, (size_t)std::max(1u, std::thread::hardware_concurrency() / 4)});
            pf.next = 1;  // the first clip is compiled by the main thread right away
            for (int wk = 0; wk < n_workers; ++wk)
                pf.workers.emplace_back([&pf, &todo, &specialise, path, asset_root = o.asset_root, extra_flags = (o.fast ? 64u : 0u) | (o.exact_cr ? 16384u : 0u) | (batch_subframes(o.blur, o.batch) ? kSlicesFlag : 0u), stereo = o.stereo] {
                    for (;;) {
                        size_t k;
                        {
                            std::unique_lock<std::mutex> lock(pf.mu);
                            if (pf.stop || pf.next >= todo.size()) return;
                            k = pf.next++;
                        }
                        if (specialise[k]) {
                            std::vector<std::string> history;
                            for (size_t c = 0; c <= k; ++c) history.push_back(todo[c].first);
                            prefetch_clip_kernel(path, history, asset_root, extra_flags, stereo);
                        }
                        {
                            std::unique_lock<std::mutex> lock(pf.mu);
                            pf.done[k] = 1;
                        }
                        pf.cv.notify_all();
                    }
                });
        }
        for (size_t k = 0; k < todo.size(); ++k) {
            const std::string& clip = todo[k].first;
            if (!pf.workers.empty() && k >= 1) {  // wait for this clip's binary only if a worker has already picked it up
                std::unique_lock<std::mutex> lock(pf.mu);
                pf.cv.wait(lock, [&] { return pf.done[k] || pf.next <= k; });
                if (!pf.done[k] && pf.next <= k) pf.next = k + 1;  // nobody started it: the main thread compiles it itself below
            }
            if (ptl_scene_init_animation(scene, clip.c_str()) != PTL_OK) return fail("init_animation");
            if (ptl_renderer_update(r, 0.0, nullptr, nullptr) != PTL_OK) return fail("update");
            int fps = o.fps;
            ptl_renderer_set_option(r, "render_depth", o.depth);
            apply_clip_overrides(scene, r, clip, &fps);
            if (ptl_renderer_set_option(r, "specialize_static", specialise[k] ? 1 : 0) != PTL_OK) return fail("specialize");
            std::printf("Rendering animation %s, %zu/%zu\n", clip.c_str(), k + 1, todo.size());
            {
                auto clip_start = std::chrono::steady_clock::now();
                // frames in flight between download and encode: one per encoder thread, but no more than ~2 GB of page-locked memory
                int in_flight = (int)std::max<size_t>(4, std::min<size_t>((size_t)threads + 2, ((size_t)2 << 30) / bytes));
                PinnedFrames pinned(bytes, in_flight);
                if (!pinned.ok()) return fail("pinned host memory");
                EncoderPool pool(threads, (size_t)threads * 2);
                int rc = render_clip(o, scene, r, scene_name, clip, todo[k].second, fps, width, o.height, subframes, pipe, pool, pinned);
                pool.finish();  // joins the encoders: every frame file is on disk (and every pinned buffer is back)
                if (rc != 0) return rc;
                std::printf("Clip `%s` on disk after %.2f s\n", clip.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - clip_start).count());
            }
            if (o.shards == 1 && o.max_frames < 0) encode_video(o, scene_name, clip, fps);
        }
        ptl_device_free(subframe_block);
        ptl_renderer_destroy(r);
        ptl_scene_free(scene);
    }
    std::printf("Total render time: %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - total_start).count());
    return 0;
}

// `check`: what the reference GUI shows when a scene does not compile (shader_error_parser + LineNumbersByKey,
// src/gui/scene.rs:1144-1171): every compiler diagnostic is attributed to the scene element whose snippet produced the
// line, with the line number inside that snippet.  Needs no GPU: hiprtc compiles for gfx950 anywhere.
int check(const Options& o) {
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(o.scene.c_str(), &scene) != PTL_OK) {
        std::printf("%s: cannot load: %s\n", o.scene.c_str(), ptl_last_error());
        return 1;
    }
    char stage_cam[256] = "";
    if (!o.stage.empty() && ptl_scene_init_stage(scene, o.stage.c_str(), stage_cam, sizeof stage_cam) != PTL_OK) {
        std::printf("Scene `%s` has no stage named `%s`\n", o.scene.c_str(), o.stage.c_str());
        return 1;
    }
    std::vector<char> log(1 << 18);
    ptl_renderer* r = nullptr;
    int rc = ptl_renderer_create(scene, -1, o.asset_root.c_str(), 0, &r, log.data(), log.size());
    if (rc == PTL_OK) {
        const ptl_uniform_desc* descs = nullptr;
        int n = 0;
        size_t block = 0;
        ptl_scene_uniform_layout(scene, &descs, &n, &block);
        std::printf("%s: ok (%d uniforms, %zu-byte block)\n", o.scene.c_str(), n, block);
        ptl_renderer_destroy(r);
        r = nullptr;
        // with a GPU: the build `render-frame` draws with (everything baked), and -- where it has affine rays -- the checking build of the same state
        // at 64 x 36: does any ray reach a product with a w the kernel assumes otherwise? (ptl_renderer_check_affine)
        if (ptl_device_count() > 0 && ptl_renderer_create(scene, o.device, o.asset_root.c_str(), 5u | 262144u, &r, log.data(), log.size()) == PTL_OK) {
            if (ptl_renderer_affine_rays(r) == 1) {
                unsigned long long bad = 0;
                if (ptl_renderer_check_affine(r, 64, 36, &bad) == PTL_OK)
                    std::printf("  affine rays: %s\n", bad == 0 ? "hold on every ray of a 64x36 frame (checking build)" : "BROKEN by this scene's snippets -- switched off (please report: the snippet scan let it through)");
                else
                    std::printf("  affine rays: not checked (%s)\n", ptl_last_error());
                if (bad != 0) {
                    ptl_renderer_destroy(r);
                    ptl_scene_free(scene);
                    return 3;
                }
            } else {
                std::printf("  affine rays: off for this scene (general products)\n");
            }
            ptl_renderer_destroy(r);
        }
        ptl_scene_free(scene);
        return 0;
    }
    std::string why = ptl_last_error();
    std::printf("%s: does not compile (%s)\n", o.scene.c_str(), why.substr(0, why.find(':')).c_str());
    int errors = 0;
    std::string text = log.data();
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        int src_line = 0, col = 0;
        size_t tag = line.find("portal_scene.hip:");
        if (tag == std::string::npos || std::sscanf(line.c_str() + tag, "portal_scene.hip:%d:%d:", &src_line, &col) != 2) continue;
        size_t msg = line.find(": ", tag + 17);
        std::string message = msg == std::string::npos ? line : line.substr(msg + 2);
        bool is_error = message.rfind("error", 0) == 0 || message.rfind("fatal error", 0) == 0;
        if (!is_error && message.rfind("warning", 0) != 0) continue;  // notes follow their error
        char kind[64] = "", name[256] = "";
        int local = 0;
        if (ptl_scene_source_line_owner(scene, src_line, kind, sizeof kind, name, sizeof name, &local) == PTL_OK)
            std::printf("  %s `%s`, line %d: %s\n", kind, name, local, message.c_str());
        else
            std::printf("  generated code, line %d: %s\n", src_line, message.c_str());
        errors += is_error;
    }
    if (errors == 0) std::printf("%s\n", log.data());
    ptl_scene_free(scene);
    return 1;
}

// `write`: load, apply --stage / --set, write the scene back in the reference's own .ron layout (an untouched scene comes
// back byte for byte; serialize_scene_new_format + ron pretty printer, src/gui/scene_serialized.rs:22-24,654-1100).
int write_scene(const Options& o) {
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(o.scene.c_str(), &scene) != PTL_OK) {
        std::fprintf(stderr, "Failed to parse scene `%s`: %s\n", o.scene.c_str(), ptl_last_error());
        return 1;
    }
    char stage_cam[256] = "";
    if (!o.stage.empty() && ptl_scene_init_stage(scene, o.stage.c_str(), stage_cam, sizeof stage_cam) != PTL_OK) {
        std::fprintf(stderr, "Scene `%s` has no stage named `%s`\n", o.scene.c_str(), o.stage.c_str());
        return 1;
    }
    for (auto& kv : o.sets)
        if (ptl_scene_set_uniform(scene, kv.first.c_str(), kv.second) != PTL_OK) {
            std::fprintf(stderr, "Scene `%s` has no uniform named `%s`\n", o.scene.c_str(), kv.first.c_str());
            return 1;
        }
    char* text = nullptr;
    if (ptl_scene_to_ron(scene, &text) != PTL_OK) return fail("write");
    std::FILE* f = o.output == "frame.png" ? stdout : std::fopen(o.output.c_str(), "wb");
    if (!f) {
        std::fprintf(stderr, "cannot open `%s`\n", o.output.c_str());
        return 1;
    }
    std::fwrite(text, 1, std::strlen(text), f);
    if (f != stdout) std::fclose(f);
    ptl_free(text);
    ptl_scene_free(scene);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        usage();
        return 2;
    }
    std::string cmd = argv[1];
    if (cmd == "version") {
        std::printf("%s\ndevices: %d\n", ptl_version(), ptl_device_count());
        return 0;
    }
    if (argc < 3 || (cmd != "render-frame" && cmd != "render" && cmd != "emit-source" && cmd != "check" && cmd != "write" && cmd != "precompile" &&
                     cmd != "render-shard")) {
        usage();
        return 2;
    }
    Options o;
    o.scene = argv[2];
    o.argv.assign(argv, argv + argc);
    if (cmd == "render") {  // CLI defaults of RenderCliOptions (src/main.rs:2757-2805)
        o.width = 3840;
        o.height = 2160;
        o.aa = 4;
        o.depth = 150;
    }
    for (int i = 3; i < argc; ++i) {
        std::string a = argv[i];
        std::replace(a.begin(), a.end(), '_', '-');  // the reference accepts --aa_count etc. as aliases
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) {
                usage();
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--width") o.width = std::atoi(next());
        else if (a == "--height") o.height = std::atoi(next());
        else if (a == "--aa-count") o.aa = std::atoi(next());
        else if (a == "--render-depth") o.depth = std::atoi(next());
        else if (a == "--time") o.time = std::atof(next());
        else if (a == "--output") o.output = next();
        else if (a == "--device") o.device = std::atoi(next());
        else if (a == "--asset-root") o.asset_root = next();
        else if (a == "--stage") o.stage = next();
        else if (a == "--animation") o.animation = next();
        else if (a == "--camera") { o.camera = next(); o.have_camera = true; }
        else if (a == "--panini") o.panini = std::atof(next());
        else if (a == "--fov") o.fov = std::atof(next());
        else if (a == "--fps") o.fps = std::atoi(next());
        else if (a == "--motion-blur-frames") o.blur = std::atoi(next());
        else if (a == "--stereoimage" || a == "--stereo-image") o.stereo = true;
        else if (a == "--concurrent-draws") o.concurrent = std::atoi(next());
        else if (a == "--batch-subframes") o.batch = std::atoi(next());
        else if (a == "--no-skip-existing") o.skip_existing = false;
        else if (a == "--filter-starts-with" || a == "--starts-with") o.starts_with = next();
        else if (a == "--scenes-dir") o.scenes_dir = next();
        else if (a == "--out-dir") o.out_dir = next();
        else if (a == "--max-frames") o.max_frames = std::atoi(next());
        else if (a == "--specialize") o.specialize = std::atoi(next());
        else if (a == "--timing") o.timing = true;
        else if (a == "--gpus") o.gpus = std::atoi(next());
        else if (a == "--devices") o.devices = next();
        else if (a == "--transport") o.transport = next();
        else if (a == "--multi-process") o.multi_process = true;
        else if (a == "--fast") o.fast = true;
        else if (a == "--exact-cr") o.exact_cr = true;
        else if (a == "--opt3") o.opt3 = true;
        else if (a == "--rank") o.rank = std::atoi(next());
        else if (a == "--world") o.world = std::atoi(next());
        else if (a == "--ipc-handle") o.ipc_handle = next();
        else if (a == "--set") {
            std::string kv = next();
            size_t eq = kv.find('=');
            if (eq == std::string::npos) {
                std::fprintf(stderr, "--set name=value\n");
                return 2;
            }
            o.sets.emplace_back(kv.substr(0, eq), std::atof(kv.c_str() + eq + 1));
        }
        else if (a == "--shard") {
            if (std::sscanf(next(), "%d/%d", &o.shard, &o.shards) != 2 || o.shards < 1 || o.shard < 0 || o.shard >= o.shards) {
                std::fprintf(stderr, "--shard K/N with 0 <= K < N\n");
                return 2;
            }
        } else if (cmd == "render" && a.rfind("--", 0) != 0 && o.clips.empty()) o.clips = argv[i];
        else {
            std::fprintf(stderr, "unknown option %s\n", argv[i]);
            return 2;
        }
    }
    if (!o.stage.empty() && !o.animation.empty()) {
        std::fprintf(stderr, "--stage and --animation exclude each other\n");
        return 2;
    }
    if (o.blur < 1 || o.blur > 256) {
        std::fprintf(stderr, "--motion-blur-frames must be 1..256\n");
        return 2;
    }
    if (cmd == "render") return render(o);
    if (o.transport != "stores" && o.transport != "copy" && o.transport != "rccl") {
        std::fprintf(stderr, "--transport stores|copy|rccl\n");
        return 2;
    }
    if (cmd == "render-frame") return render_frame(o);
    if (cmd == "render-shard") return render_shard(o);
    if (cmd == "precompile") return precompile(o);
    if (cmd == "check") return check(o);
    if (cmd == "write") return write_scene(o);
    // emit-source
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(o.scene.c_str(), &scene) != PTL_OK) {
        std::fprintf(stderr, "Failed to parse scene `%s`: %s\n", o.scene.c_str(), ptl_last_error());
        return 1;
    }
    char stage_cam[256] = "";
    if (!o.stage.empty() && ptl_scene_init_stage(scene, o.stage.c_str(), stage_cam, sizeof stage_cam) != PTL_OK) {
        std::fprintf(stderr, "Scene `%s` has no stage named `%s`\n", o.scene.c_str(), o.stage.c_str());
        return 1;
    }
    char* src = nullptr;
    if (ptl_scene_generate_source(scene, 0, &src) != PTL_OK) return fail("generate");
    std::fputs(src, stdout);
    ptl_free(src);
    ptl_scene_free(scene);
    return 0;
}
