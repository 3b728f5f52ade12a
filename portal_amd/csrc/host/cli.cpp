// cli.cpp -- `portal-amd render-frame`, the offline counterpart of the reference CLI
// (`portal render-frame <scene> --width --height --aa-count --render-depth --output`,
// src/main.rs:2726-2755,2876-2946), writing a PNG instead of drawing to a window.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/portal_amd.h"

static void usage() {
    std::fprintf(stderr,
                 "usage: portal-amd render-frame <scene.ron> [--width W] [--height H] [--aa-count N] [--render-depth D]\n"
                 "                  [--stage NAME] [--camera NAME] [--time T] [--output out.png] [--device I] [--asset-root DIR] [--panini D --fov DEG]\n"
                 "       portal-amd emit-source <scene.ron>       print the generated HIP kernel source\n"
                 "       portal-amd version\n");
}

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
    if (argc < 3) {
        usage();
        return 2;
    }
    std::string scene_path = argv[2];
    int width = 1920, height = 1080, aa = 1, depth = 100, device = 0;  // CLI defaults: src/main.rs:2744-2754
    double time = 0.0, panini = -1.0, fov = 90.0;
    std::string output = "frame.png", asset_root = ".", stage, camera;
    bool have_camera = false;
    for (int i = 3; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) {
                usage();
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--width") width = std::atoi(next());
        else if (a == "--height") height = std::atoi(next());
        else if (a == "--aa-count") aa = std::atoi(next());
        else if (a == "--render-depth") depth = std::atoi(next());
        else if (a == "--time") time = std::atof(next());
        else if (a == "--output") output = next();
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--asset-root") asset_root = next();
        else if (a == "--stage") stage = next();
        else if (a == "--camera") { camera = next(); have_camera = true; }
        else if (a == "--panini") panini = std::atof(next());
        else if (a == "--fov") fov = std::atof(next());
        else {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        }
    }
    ptl_scene* scene = nullptr;
    if (ptl_scene_load_file(scene_path.c_str(), &scene) != PTL_OK) {
        std::fprintf(stderr, "Failed to parse scene `%s`: %s\n", scene_path.c_str(), ptl_last_error());
        return 1;
    }
    ptl_scene_set_time(scene, time, time);
    char stage_cam[256] = "";
    if (!stage.empty() && ptl_scene_init_stage(scene, stage.c_str(), stage_cam, sizeof stage_cam) != PTL_OK) {  // src/main.rs:2900-2904
        std::fprintf(stderr, "Scene `%s` has no stage named `%s`\n", scene_path.c_str(), stage.c_str());
        return 1;
    }
    if (cmd == "emit-source") {
        char* src = nullptr;
        if (ptl_scene_generate_source(scene, 0, &src) != PTL_OK) {
            std::fprintf(stderr, "%s\n", ptl_last_error());
            return 1;
        }
        std::fputs(src, stdout);
        ptl_free(src);
        return 0;
    }
    if (cmd != "render-frame") {
        usage();
        return 2;
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<char> log(1 << 16);
    ptl_renderer* r = nullptr;
    int rc = ptl_renderer_create(scene, device, asset_root.c_str(), 0, &r, log.data(), log.size());
    if (rc != PTL_OK) {
        std::fprintf(stderr, "renderer: %s\n%s\n", ptl_last_error(), log.data());
        return 1;
    }
    if (have_camera || stage_cam[0]) {  // --camera wins over the stage's camera (src/main.rs:2918-2926)
        const char* which = have_camera ? camera.c_str() : stage_cam;
        if (ptl_renderer_use_camera(r, which) != PTL_OK) {
            std::fprintf(stderr, "Scene `%s` has no camera named `%s`\n", scene_path.c_str(), which);
            return 1;
        }
    }
    ptl_renderer_set_option(r, "aa_count", aa);
    ptl_renderer_set_option(r, "render_depth", depth);
    if (panini >= 0.0) {
        ptl_renderer_set_option(r, "use_panini_projection", 1);
        ptl_renderer_set_option(r, "panini_param", panini);
    }
    ptl_renderer_set_option(r, "view_angle", fov / 180.0 * 3.14159265358979323846);
    ptl_frame frame{width, height, 0, 1};
    std::vector<uint8_t> img((size_t)width * height * 4);
    float ms = 0.0f;
    rc = ptl_renderer_draw_to_host(r, &frame, img.data(), nullptr, nullptr, &ms);
    if (rc != PTL_OK) {
        std::fprintf(stderr, "render: %s\n", ptl_last_error());
        return 1;
    }
    if (ptl_png_write(output.c_str(), img.data(), width, height) != PTL_OK) {
        std::fprintf(stderr, "%s\n", ptl_last_error());
        return 1;
    }
    double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Rendered `%s` to `%s` (%dx%d, aa %d, depth %d): kernel %.3f ms, %.1f Mray/s; total %.2f s\n", scene_path.c_str(),
                output.c_str(), width, height, aa, depth, ms, (double)width * height * aa / (ms * 1e3), total);
    ptl_renderer_destroy(r);
    ptl_scene_free(scene);
    return 0;
}
