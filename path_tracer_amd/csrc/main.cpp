// main.cpp -- the CS488 command-line host on top of the C ABI (include/pt_amd.h).
//
// Keeps the reference program's surface (src/main.cpp:6-37 and CS488Window::start,
// src/cs488.h:2271-2325): `CS488 <obj> [hdr]`, the same load / build messages on stdout, the same
// fallbacks to a single triangle, and per frame one line with the total samples per pixel
// followed by one line with the frame's seconds.  The GLFW window is replaced by a headless loop:
// the program renders --frames progressive frames and optionally writes the frame buffer.
//
// Compile-time constants of the reference that are options here (defaults unchanged):
//   --width 512 --height 384      globalWidth / globalHeight      (src/cs488.h:48-49)
//   --spp 10                      SAMPLES_PER_PIXEL               (src/cs488.h:101)
//   --accel kd|bvh                #define KD_TREE, on by default  (src/cs488.h:1197)
//   --no-progressive              #undef PROGRESSIVE_PATHTRACING  (src/cs488.h:99,1862-1872)
//   --max-level 15                maxLevel                        (src/cs488.h:60)
//   --eye/--lookat/--up x y z, --fov deg   camera globals         (src/cs488.h:75,83-85)
// Headless additions: --frames N (default 10), --out file.png|.pfm (the `I` key's
// FrameBuffer.save, src/cs488.h:249-253), --seed N, --device N, --stats (ray statistics and
// Mrays/s on stderr after the last frame).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"

namespace {

struct Options {
    std::vector<std::string> positional;
    int width = 512, height = 384, spp = 10, frames = 10, accel = PT_ACCEL_KD, device = 0, max_level = 15;
    bool progressive = true, stats = false;
    bool have_seed = false;
    uint64_t seed = 0;
    float eye[3] = {0.0f, 0.0f, 1.5f}, lookat[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    float fov = 45.0f;
    std::string out;
};

void usage() {
    fprintf(stderr,
            "usage: CS488 <scene.obj> [probe.hdr] [--accel kd|bvh] [--spp N] [--width W --height H] [--frames N]\n"
            "             [--out frame.png|frame.pfm] [--no-progressive] [--max-level N] [--eye x y z] [--lookat x y z]\n"
            "             [--up x y z] [--fov deg] [--seed N] [--device N] [--stats] [--headless]\n");
}

bool parse(int argc, const char* argv[], Options& o) {
    auto need = [&](int& i, int n) -> bool { if (i + n >= argc) { usage(); return false; } return true; };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--", 0) != 0) { o.positional.push_back(a); continue; }
        if (a == "--headless") continue;  // the only mode there is
        else if (a == "--stats") o.stats = true;
        else if (a == "--no-progressive") o.progressive = false;
        else if (a == "--accel") { if (!need(i, 1)) return false; std::string v = argv[++i]; if (v == "kd") o.accel = PT_ACCEL_KD; else if (v == "bvh") o.accel = PT_ACCEL_BVH; else { usage(); return false; } }
        else if (a == "--spp") { if (!need(i, 1)) return false; o.spp = atoi(argv[++i]); }
        else if (a == "--width") { if (!need(i, 1)) return false; o.width = atoi(argv[++i]); }
        else if (a == "--height") { if (!need(i, 1)) return false; o.height = atoi(argv[++i]); }
        else if (a == "--frames") { if (!need(i, 1)) return false; o.frames = atoi(argv[++i]); }
        else if (a == "--max-level") { if (!need(i, 1)) return false; o.max_level = atoi(argv[++i]); }
        else if (a == "--device") { if (!need(i, 1)) return false; o.device = atoi(argv[++i]); }
        else if (a == "--seed") { if (!need(i, 1)) return false; o.seed = strtoull(argv[++i], nullptr, 0); o.have_seed = true; }
        else if (a == "--fov") { if (!need(i, 1)) return false; o.fov = (float)atof(argv[++i]); }
        else if (a == "--out") { if (!need(i, 1)) return false; o.out = argv[++i]; }
        else if (a == "--eye" || a == "--lookat" || a == "--up") {
            if (!need(i, 3)) return false;
            float* dst = a == "--eye" ? o.eye : (a == "--lookat" ? o.lookat : o.up);
            for (int k = 0; k < 3; k++) dst[k] = (float)atof(argv[++i]);
        } else { usage(); return false; }
    }
    return o.width > 0 && o.height > 0 && o.spp > 0 && o.frames >= 0;
}

int die(PtContext* ctx, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, ctx ? pt_last_error(ctx) : "");
    if (ctx) pt_destroy(ctx);
    return 1;
}

}  // namespace

int main(int argc, const char* argv[]) {
    Options opt;
    if (!parse(argc, argv, opt)) return 2;

    PtContext* ctx = nullptr;
    int rc = pt_create(opt.device, &ctx);
    if (rc != PT_OK) {
        fprintf(stderr, "No usable GPU (device %d): this program renders on an AMD GPU only.\n", opt.device);
        return 1;
    }

    // setupScene, src/main.cpp:6-27
    if (!opt.positional.empty()) {
        const char* obj = opt.positional[0].c_str();
        printf("Loading \"%s\"...\n", obj);  // src/cs488.h:712
        rc = pt_load_obj(ctx, obj);
        if (rc == PT_OK) {
            printf("Loaded \"%s\" with %d triangles.\n", obj, pt_num_triangles(ctx));  // src/cs488.h:779
        } else if (rc == PT_ERR_IO || rc == PT_ERR_INVALID_OBJ) {
            if (rc == PT_ERR_IO) printf("Cannot open \"%s\" for reading\n", obj);  // src/cs488.h:961
            printf("Invalid .obj file.\n");
            printf("Making a single triangle instead.\n");
            pt_single_triangle(ctx);
        } else {
            return die(ctx, "pt_load_obj", rc);
        }
    } else {
        printf("Specify .obj file in the command line arguments. Example: CS488.exe cornellbox.obj\n");
        printf("Making a single triangle instead.\n");
        pt_single_triangle(ctx);
    }
    if (opt.positional.size() > 1) {
        const char* hdr = opt.positional[1].c_str();
        rc = pt_load_env(ctx, hdr);
        if (rc == PT_OK) printf("Loaded \"%s\".\n", hdr);                     // src/cs488.h:217
        else std::cerr << "Unable to load: " << hdr << std::endl;             // src/cs488.h:204
    }

    // CS488Window::start -> Scene::preCalc, src/cs488.h:2272, 1733-1747
    printf(opt.accel == PT_ACCEL_KD ? "Building KdTree...\n" : "Building BVH...\n");  // src/cs488.h:1389,1662
    fflush(stdout);
    if (opt.have_seed) pt_set_option(ctx, PT_OPT_SEED, opt.seed);
    pt_set_option(ctx, PT_OPT_MAX_LEVEL, (uint64_t)opt.max_level);
    if (opt.stats) pt_set_option(ctx, PT_OPT_COUNT_TESTS, 1);
    rc = pt_build(ctx, opt.accel);
    if (rc != PT_OK) return die(ctx, "pt_build", rc);
    printf("Done.\n");
    rc = pt_set_camera(ctx, opt.eye, opt.lookat, opt.up, opt.fov, 0.032f);
    if (rc != PT_OK) return die(ctx, "pt_set_camera", rc);
    rc = pt_set_resolution(ctx, opt.width, opt.height);
    if (rc != PT_OK) return die(ctx, "pt_set_resolution", rc);

    // the frame loop of src/cs488.h:2278-2324 without window events
    double total_seconds = 0.0;
    for (int frame = 0; frame < opt.frames; frame++) {
        auto t0 = std::chrono::high_resolution_clock::now();
        if (!opt.progressive) {
            rc = pt_reset(ctx);
            if (rc != PT_OK) return die(ctx, "pt_reset", rc);
        }
        rc = pt_render(ctx, opt.spp);
        if (rc == PT_OK) rc = pt_synchronize(ctx);
        if (rc != PT_OK) return die(ctx, "pt_render", rc);
        std::cout << pt_total_samples(ctx) << std::endl;  // src/cs488.h:1896
        auto t1 = std::chrono::high_resolution_clock::now();
        const double seconds = std::chrono::duration_cast<std::chrono::duration<double>>(t1 - t0).count();
        total_seconds += seconds;
        std::cout << seconds << std::endl;  // src/cs488.h:2322
    }

    if (!opt.out.empty()) {
        const bool pfm = opt.out.size() > 4 && opt.out.compare(opt.out.size() - 4, 4, ".pfm") == 0;
        rc = pfm ? pt_save_pfm(ctx, 1, opt.out.c_str()) : pt_save_png(ctx, 1, opt.out.c_str());
        if (rc != PT_OK) return die(ctx, "saving the frame", rc);
        printf("Saved \"%s\".\n", opt.out.c_str());  // src/cs488.h:231
    }
    if (opt.stats) {
        uint64_t cnt[PT_CNT_COUNT];
        if (pt_get_counters(ctx, cnt) == PT_OK && total_seconds > 0) {
            const double rays = (double)cnt[PT_CNT_RAYS];
            fprintf(stderr, "rays %.0f  samples %.0f  rays/sample %.3f  box tests/ray %.2f  tri tests/ray %.2f  dropped NaN samples %llu\n",
                    rays, (double)cnt[PT_CNT_SAMPLES], rays / (double)cnt[PT_CNT_SAMPLES], cnt[PT_CNT_BOX_TESTS] / rays,
                    cnt[PT_CNT_TRI_TESTS] / rays, (unsigned long long)cnt[PT_CNT_NAN_SAMPLES]);
            fprintf(stderr, "%.2f Mrays/s  %.2f spp/s over %d frames (%.3f s)\n", rays / total_seconds * 1e-6,
                    (double)opt.spp * opt.frames / total_seconds, opt.frames, total_seconds);
        }
    }
    pt_destroy(ctx);
    return 0;
}
