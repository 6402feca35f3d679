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
// Mrays/s on stderr after the last frame; --count-tests adds slab and triangle tests per ray, counted by slower kernels), --sequence PREFIX (every frame to PREFIX%04d.png),
// --obj-polygons (faces with more than three vertices as triangle fans, negative indices; the
// reference reads three vertices per face, src/cs488.h:1059), --gpus N (the frame's samples are split over
// GPUs device .. device+N-1 of this node and summed onto the first with one RCCL reduce per frame, in place of the
// reference's column-strip threads, src/cs488.h:1885-1895; the output lines stay the same), --drain-frames (wait for
// every frame before the next is queued, as the reference's join at src/cs488.h:1893-1895 does; without it a frame
// whose pixels nobody reads that frame -- no recording, no --sequence -- only waits until the frame a few before it
// has ended, so consecutive frames overlap on the GPU.  A frame's two lines are still printed only once the frame HAS
// ended, as the reference prints them after its join (src/cs488.h:1896, 2322): they wait in a short list until the
// frame's end event has passed, and the seconds are then the time since the previous frame's lines),
// --reduce-every K (with --gpus: K frames go into one per-frame buffer before it is reduced, for frames of a few ms).
//
// Window input is replaced by a script (--script FILE), one event per line, applied where the
// reference polls events, i.e. before the frame with that index is rendered (src/cs488.h:2283):
//   <frame> key W|S|A|D|Q|Z      move eye and lookat by SCLFACT along view / right / up (src/cs488.h:270-298)
//   <frame> key I                save the frame buffer as output<n>.png, n = int(1000 rand()) (:249-253)
//   <frame> key F                start / stop recording (:255-268) into the animated output<n>.gif, one
//                                frame per rendered frame (:2297-2310)
//   <frame> key ESCAPE           leave the loop (:245-247)
//   <frame> drag <dx> <dy>       cursor movement in pixels with the left button held (:317-344)
//   <frame> save <file>          write the frame buffer (.png or .pfm)
// As in the reference, accumulation restarts only when the view direction or the right vector
// changed (src/cs488.h:2266-2293): translations keep accumulating.
#include <chrono>
#include <cmath>
#include <deque>
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
    int gpus = 0;  // 0: one context, no group; N >= 1: a group of N contexts (N = 1 still goes through the reduce)
    int reduce_every = 1;
    bool progressive = true, stats = false, count_tests = false, drain_frames = false, trace_frames = false;
    bool have_seed = false;
    uint64_t seed = 0;
    float eye[3] = {0.0f, 0.0f, 1.5f}, lookat[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    float fov = 45.0f;
    std::string out, script, sequence;
    bool obj_extended = false;
};

struct Event {
    int frame;
    std::string kind, arg;  // key <name> | drag | save <file>
    float dx = 0.0f, dy = 0.0f;
};

// float3 arithmetic of the input callbacks (src/cs488.h:270-341), component-wise like linalg.h
struct F3 { float x, y, z; };
F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
float dot(F3 a, F3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
F3 normalize(F3 a) { const float l = sqrtf(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }
bool differs(F3 a, F3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }

// rotateVector of cursorPosFunc (src/cs488.h:324-335)
F3 rotate_about(float theta, F3 v, F3 w) {
    const float c = cosf(theta), s = sinf(theta);
    const F3 v0 = dot(v, w) * w;
    const F3 v1 = v - v0;
    const F3 v2 = cross(w, v1);
    return v0 + c * v1 + s * v2;
}

// PCG32::rand (src/cs488.h:130-143): names of the files the I and F keys write, one draw per frame (:2317)
struct Pcg32 {
    uint64_t state = 0xcafef00dd15ea5e5ull;
    float rand() {
        uint64_t x = state;
        const unsigned count = (unsigned)(x >> 61);
        state = x * 6364136223846793005ull;
        x ^= x >> 22;
        return (float)((double)(uint32_t)(x >> (22 + count)) / 4294967296.0);
    }
};

bool read_script(const std::string& path, std::vector<Event>& out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char line[512];
    while (fgets(line, sizeof(line), f)) {
        char kind[32] = "", arg[256] = "";
        Event e;
        if (line[0] == '#' || sscanf(line, "%d %31s", &e.frame, kind) != 2) continue;
        e.kind = kind;
        if (e.kind == "drag") {
            if (sscanf(line, "%*d %*s %f %f", &e.dx, &e.dy) != 2) { fclose(f); return false; }
        } else if (e.kind == "key" || e.kind == "save") {
            if (sscanf(line, "%*d %*s %255s", arg) != 1) { fclose(f); return false; }
            e.arg = arg;
        } else { fclose(f); return false; }
        out.push_back(e);
    }
    fclose(f);
    return true;
}

void usage() {
    fprintf(stderr,
            "usage: CS488 <scene.obj> [probe.hdr] [--accel kd|bvh] [--spp N] [--width W --height H] [--frames N]\n"
            "             [--out frame.png|frame.pfm] [--no-progressive] [--max-level N] [--eye x y z] [--lookat x y z]\n"
            "             [--up x y z] [--fov deg] [--seed N] [--device N] [--gpus N] [--reduce-every K] [--stats] [--count-tests] [--headless] [--drain-frames]\n"
            "             [--script events.txt] [--sequence PREFIX] [--obj-polygons]\n");
}

bool parse(int argc, const char* argv[], Options& o) {
    auto need = [&](int& i, int n) -> bool { if (i + n >= argc) { usage(); return false; } return true; };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--", 0) != 0) { o.positional.push_back(a); continue; }
        if (a == "--headless") continue;  // the only mode there is
        else if (a == "--stats") o.stats = true;
        else if (a == "--count-tests") { o.stats = true; o.count_tests = true; }
        else if (a == "--drain-frames") o.drain_frames = true;
        else if (a == "--trace-frames") o.trace_frames = true;  // stderr: per printed pair, frames queued and frames unfinished then
        else if (a == "--no-progressive") o.progressive = false;
        else if (a == "--accel") { if (!need(i, 1)) return false; std::string v = argv[++i]; if (v == "kd") o.accel = PT_ACCEL_KD; else if (v == "bvh") o.accel = PT_ACCEL_BVH; else { usage(); return false; } }
        else if (a == "--spp") { if (!need(i, 1)) return false; o.spp = atoi(argv[++i]); }
        else if (a == "--width") { if (!need(i, 1)) return false; o.width = atoi(argv[++i]); }
        else if (a == "--height") { if (!need(i, 1)) return false; o.height = atoi(argv[++i]); }
        else if (a == "--frames") { if (!need(i, 1)) return false; o.frames = atoi(argv[++i]); }
        else if (a == "--max-level") { if (!need(i, 1)) return false; o.max_level = atoi(argv[++i]); }
        else if (a == "--device") { if (!need(i, 1)) return false; o.device = atoi(argv[++i]); }
        else if (a == "--gpus") { if (!need(i, 1)) return false; o.gpus = atoi(argv[++i]); if (o.gpus < 1 || o.gpus > 64) { usage(); return false; } }
        else if (a == "--reduce-every") { if (!need(i, 1)) return false; o.reduce_every = atoi(argv[++i]); if (o.reduce_every < 1 || o.reduce_every > 1024) { usage(); return false; } }
        else if (a == "--seed") { if (!need(i, 1)) return false; o.seed = strtoull(argv[++i], nullptr, 0); o.have_seed = true; }
        else if (a == "--fov") { if (!need(i, 1)) return false; o.fov = (float)atof(argv[++i]); }
        else if (a == "--out") { if (!need(i, 1)) return false; o.out = argv[++i]; }
        else if (a == "--obj-polygons") o.obj_extended = true;
        else if (a == "--script") { if (!need(i, 1)) return false; o.script = argv[++i]; }
        else if (a == "--sequence") { if (!need(i, 1)) return false; o.sequence = argv[++i]; }
        else if (a == "--eye" || a == "--lookat" || a == "--up") {
            if (!need(i, 3)) return false;
            float* dst = a == "--eye" ? o.eye : (a == "--lookat" ? o.lookat : o.up);
            for (int k = 0; k < 3; k++) dst[k] = (float)atof(argv[++i]);
        } else { usage(); return false; }
    }
    return o.width > 0 && o.height > 0 && o.spp > 0 && o.frames >= 0;
}

int save_frame(PtContext* ctx, const std::string& path) {
    const bool pfm = path.size() > 4 && path.compare(path.size() - 4, 4, ".pfm") == 0;
    const int rc = pfm ? pt_save_pfm(ctx, 1, path.c_str()) : pt_save_png(ctx, 1, path.c_str());
    if (rc == PT_OK) printf("Saved \"%s\".\n", path.c_str());  // src/cs488.h:231
    return rc;
}

// Everything that has to be the same on every GPU of a group is done through these: the call on every context, the
// first failure returned.
struct Contexts {
    std::vector<PtContext*> all;
    PtGroup* group = nullptr;
    PtContext* root() const { return all.empty() ? nullptr : all[0]; }
    template <typename F> int each(F f) {
        for (PtContext* c : all) { const int rc = f(c); if (rc != PT_OK) return rc; }
        return PT_OK;
    }
    void destroy() {
        if (group) pt_group_destroy(group);
        group = nullptr;
        for (PtContext* c : all) pt_destroy(c);
        all.clear();
    }
};

int die(Contexts& cs, const char* what, int rc) {
    const char* why = cs.group && pt_group_last_error(cs.group)[0] ? pt_group_last_error(cs.group) : (cs.root() ? pt_last_error(cs.root()) : "");
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, why);
    cs.destroy();
    return 1;
}

}  // namespace

int main(int argc, const char* argv[]) {
    Options opt;
    if (!parse(argc, argv, opt)) return 2;

    Contexts cs;
    int rc = PT_OK;
    for (int g = 0; g < (opt.gpus > 0 ? opt.gpus : 1); g++) {
        PtContext* c = nullptr;
        rc = pt_create(opt.device + g, &c);
        if (rc != PT_OK) {
            fprintf(stderr, "No usable GPU (device %d): this program renders on an AMD GPU only.\n", opt.device + g);
            cs.destroy();
            return 1;
        }
        cs.all.push_back(c);
    }
    PtContext* ctx = cs.root();  // the context whose frame buffer is shown, saved and recorded

    // setupScene, src/main.cpp:6-27
    if (!opt.positional.empty()) {
        const char* obj = opt.positional[0].c_str();
        if (opt.obj_extended) cs.each([&](PtContext* c) { return pt_set_option(c, PT_OPT_OBJ_EXTENDED, 1); });
        printf("Loading \"%s\"...\n", obj);  // src/cs488.h:712
        rc = cs.each([&](PtContext* c) { return pt_load_obj(c, obj); });
        if (rc == PT_OK) {
            printf("Loaded \"%s\" with %d triangles.\n", obj, pt_num_triangles(ctx));  // src/cs488.h:779
        } else if (rc == PT_ERR_IO || rc == PT_ERR_INVALID_OBJ) {
            if (rc == PT_ERR_IO) printf("Cannot open \"%s\" for reading\n", obj);  // src/cs488.h:961
            printf("Invalid .obj file.\n");
            printf("Making a single triangle instead.\n");
            cs.each([&](PtContext* c) { return pt_single_triangle(c); });
        } else {
            return die(cs, "pt_load_obj", rc);
        }
    } else {
        printf("Specify .obj file in the command line arguments. Example: CS488.exe cornellbox.obj\n");
        printf("Making a single triangle instead.\n");
        cs.each([&](PtContext* c) { return pt_single_triangle(c); });
    }
    if (opt.positional.size() > 1) {
        const char* hdr = opt.positional[1].c_str();
        rc = cs.each([&](PtContext* c) { return pt_load_env(c, hdr); });
        if (rc == PT_OK) printf("Loaded \"%s\".\n", hdr);                     // src/cs488.h:217
        else std::cerr << "Unable to load: " << hdr << std::endl;             // src/cs488.h:204
    }

    // CS488Window::start -> Scene::preCalc, src/cs488.h:2272, 1733-1747
    printf(opt.accel == PT_ACCEL_KD ? "Building KdTree...\n" : "Building BVH...\n");  // src/cs488.h:1389,1662
    fflush(stdout);
    rc = cs.each([&](PtContext* c) {
        if (opt.have_seed) pt_set_option(c, PT_OPT_SEED, opt.seed);
        pt_set_option(c, PT_OPT_MAX_LEVEL, (uint64_t)opt.max_level);
        if (opt.count_tests) pt_set_option(c, PT_OPT_COUNT_TESTS, 1);  // the counting kernels: one launch per level, slower
        return pt_build(c, opt.accel);
    });
    if (rc != PT_OK) return die(cs, "pt_build", rc);
    printf("Done.\n");
    rc = cs.each([&](PtContext* c) { return pt_set_resolution(c, opt.width, opt.height); });
    if (rc != PT_OK) return die(cs, "pt_set_resolution", rc);
    if (opt.gpus > 0) {
        rc = pt_group_create((int)cs.all.size(), cs.all.data(), &cs.group);
        if (rc != PT_OK) return die(cs, "pt_group_create", rc);
        if (opt.reduce_every > 1 && (rc = pt_group_set_cadence(cs.group, opt.reduce_every)) != PT_OK) return die(cs, "pt_group_set_cadence", rc);
    }

    std::vector<Event> events;
    if (!opt.script.empty() && !read_script(opt.script, events)) {
        fprintf(stderr, "cannot read the event script \"%s\"\n", opt.script.c_str());
        cs.destroy();
        return 2;
    }

    // the frame loop of src/cs488.h:2278-2324; scripted events stand in for glfwPollEvents
    F3 eye = {opt.eye[0], opt.eye[1], opt.eye[2]}, lookat = {opt.lookat[0], opt.lookat[1], opt.lookat[2]};
    const F3 up = normalize(F3{opt.up[0], opt.up[1], opt.up[2]});  // src/cs488.h:85
    F3 view = normalize(lookat - eye), right = normalize(cross(view, up));
    F3 prev_view = view, prev_right = right;
    Pcg32 names;
    bool recording = false, quit = false;
    double total_seconds = 0.0;
    int frames_done = 0;
    uint64_t in_flight = 2;  // frames the context keeps in flight: as many may be queued ahead of the one waited for
    if (!cs.group) (void)pt_get_option(ctx, PT_OPT_STREAMS, &in_flight);
    // Frames that have been queued and whose two lines are still to be printed, oldest first: the total sample count the
    // frame reaches.  A frame's lines go out when the frame has ended on the GPU (src/cs488.h:1893-1896 prints after the
    // join), never before; `last_print` is when the previous frame's lines went out.
    std::deque<int> unprinted;
    int frames_queued = 0, pairs_printed = 0;
    auto last_print = std::chrono::high_resolution_clock::now();
    auto print_finished = [&](bool all) {
        const int unfinished = all ? 0 : (cs.group ? pt_group_frames_in_flight(cs.group) : pt_frames_in_flight(ctx));
        while ((int)unprinted.size() > (unfinished > 0 ? unfinished : 0)) {
            pairs_printed++;
            if (opt.trace_frames) fprintf(stderr, "frame-lines %d queued %d unfinished %d\n", pairs_printed, frames_queued, unfinished > 0 ? unfinished : 0);
            std::cout << unprinted.front() << std::endl;  // src/cs488.h:1896
            unprinted.pop_front();
            const auto now = std::chrono::high_resolution_clock::now();
            const double seconds = std::chrono::duration_cast<std::chrono::duration<double>>(now - last_print).count();
            last_print = now;
            total_seconds += seconds;
            std::cout << seconds << std::endl;  // src/cs488.h:2322
        }
    };
    for (int frame = 0; frame < opt.frames && !quit; frame++) {
        if (unprinted.empty()) last_print = std::chrono::high_resolution_clock::now();  // src/cs488.h:2280
        for (const Event& e : events) {
            if (e.frame != frame) continue;
            if (e.kind == "save" || (e.kind == "key" && (e.arg == "I" || e.arg == "F"))) {
                // these read pixels and print: the frames before them end first, and their lines go out first
                if ((rc = cs.group ? pt_group_synchronize(cs.group) : pt_synchronize(ctx)) != PT_OK) return die(cs, "pt_synchronize", rc);
                print_finished(true);
            }
            if (e.kind == "key") {
                const std::string& k = e.arg;
                if (k == "W") { eye = eye + 0.1f * view; lookat = lookat + 0.1f * view; }        // SCLFACT, src/cs488.h:70
                else if (k == "S") { eye = eye - 0.1f * view; lookat = lookat - 0.1f * view; }
                else if (k == "Q") { eye = eye + 0.1f * up; lookat = lookat + 0.1f * up; }
                else if (k == "Z") { eye = eye - 0.1f * up; lookat = lookat - 0.1f * up; }
                else if (k == "A") { eye = eye - 0.1f * right; lookat = lookat - 0.1f * right; }
                else if (k == "D") { eye = eye + 0.1f * right; lookat = lookat + 0.1f * right; }
                else if (k == "ESCAPE") quit = true;
                else if (k == "I") {
                    char name[64];
                    snprintf(name, sizeof(name), "output%d.png", int(1000.0 * names.rand()));
                    if ((rc = pt_save_png(ctx, 1, name)) != PT_OK) return die(cs, "saving the frame", rc);
                } else if (k == "F") {
                    if (!recording) {
                        char name[64];
                        snprintf(name, sizeof(name), "output%d.gif", int(1000.0 * names.rand()));  // src/cs488.h:258
                        printf("Saving \"%s\"...\n", name);
                        if ((rc = pt_record_begin(ctx, name, 1)) != PT_OK) return die(cs, "opening the recording", rc);
                        recording = true;
                        printf("(Recording started)\n");
                    } else {
                        pt_record_end(ctx);
                        recording = false;
                        printf("(Recording done)\n");
                    }
                }
            } else if (e.kind == "drag") {  // cursorPosFunc with the left button held, src/cs488.h:318-344
                const float deg_to_rad = 3.14159265358979323846f / 180.0f;
                const float xfact = -0.2f * e.dy, yfact = -0.2f * e.dx;  // ANGFACT, src/cs488.h:69
                F3 v = view;
                v = rotate_about(xfact * deg_to_rad, v, right);
                v = rotate_about(yfact * deg_to_rad, v, up);
                view = v;
                lookat = eye + view;
                right = cross(view, up);
            } else if (e.kind == "save") {
                if ((rc = save_frame(ctx, e.arg)) != PT_OK) return die(cs, "saving the frame", rc);
            }
        }
        if (quit) break;
        view = normalize(lookat - eye);           // src/cs488.h:2284-2285
        right = normalize(cross(view, up));
        const float e3[3] = {eye.x, eye.y, eye.z}, l3[3] = {lookat.x, lookat.y, lookat.z}, u3[3] = {up.x, up.y, up.z};
        rc = cs.each([&](PtContext* c) { return pt_set_camera(c, e3, l3, u3, opt.fov, 0.032f); });
        if (rc != PT_OK) return die(cs, "pt_set_camera", rc);
        if (differs(prev_view, view) || differs(prev_right, right) || !opt.progressive) {  // src/cs488.h:2287-2293
            rc = cs.group ? pt_group_reset(cs.group) : pt_reset(ctx);
            if (rc != PT_OK) return die(cs, "pt_reset", rc);
            prev_view = view;
            prev_right = right;
        }
        // Scene::Pathtrace, src/cs488.h:1878-1897: on one GPU, or split by samples over the group with one reduce
        rc = cs.group ? pt_group_render(cs.group, opt.spp) : pt_render(ctx, opt.spp);
        const bool pixels_read = recording || !opt.sequence.empty();
        const int keep = 2 * (int)in_flight + 2;  // well behind the frames that run: the wait then rarely sleeps
        if (rc == PT_OK) {
            if (opt.drain_frames || pixels_read) rc = cs.group ? pt_group_synchronize(cs.group) : pt_synchronize(ctx);
            else rc = cs.group ? pt_group_wait_frames(cs.group, keep) : pt_wait_frames(ctx, keep);
        }
        if (rc != PT_OK) return die(cs, "pt_render", rc);
        unprinted.push_back(pt_total_samples(ctx));
        frames_queued++;
        print_finished(opt.drain_frames || pixels_read);
        if (recording || !opt.sequence.empty()) {
            char name[1200];
            if (recording && (rc = pt_record_frame(ctx)) != PT_OK) return die(cs, "saving a recorded frame", rc);
            if (!opt.sequence.empty()) {
                snprintf(name, sizeof(name), "%s%04d.png", opt.sequence.c_str(), frame);
                if ((rc = pt_save_png(ctx, 1, name)) != PT_OK) return die(cs, "saving a sequence frame", rc);
            }
        }
        names.rand();  // src/cs488.h:2317
        frames_done++;
    }

    // what is still in flight belongs to the run's time, and its lines are still to come
    if ((rc = pt_synchronize(ctx)) != PT_OK) return die(cs, "pt_synchronize", rc);
    print_finished(true);
    if (!opt.out.empty() && (rc = save_frame(ctx, opt.out)) != PT_OK) return die(cs, "saving the frame", rc);
    if (opt.stats) {
        uint64_t cnt[PT_CNT_COUNT] = {};
        bool have = true;
        for (PtContext* c : cs.all) {  // a group's members each count the rays of their own share of the samples
            uint64_t one[PT_CNT_COUNT];
            have = have && pt_get_counters(c, one) == PT_OK;
            for (int k = 0; k < PT_CNT_COUNT; k++) cnt[k] += one[k];
        }
        if (have && total_seconds > 0) {
            const double rays = (double)cnt[PT_CNT_RAYS];
            fprintf(stderr, "rays %.0f  samples %.0f  rays/sample %.3f  dropped NaN samples %llu\n", rays, (double)cnt[PT_CNT_SAMPLES],
                    rays / (double)cnt[PT_CNT_SAMPLES], (unsigned long long)cnt[PT_CNT_NAN_SAMPLES]);
            if (opt.count_tests) fprintf(stderr, "box tests/ray %.2f  tri tests/ray %.2f\n", cnt[PT_CNT_BOX_TESTS] / rays, cnt[PT_CNT_TRI_TESTS] / rays);
            fprintf(stderr, "%.2f Mrays/s  %.2f spp/s over %d frames (%.3f s)\n", rays / total_seconds * 1e-6,
                    (double)opt.spp * frames_done / total_seconds, frames_done, total_seconds);
        }
    }
    cs.destroy();
    return 0;
}
