// pt_api.cpp -- implementation of the C ABI declared in include/pt_amd.h: host-side scene state,
// tree flattening, device memory, and the per-frame schedule of the wavefront kernels.
//
// There is no CPU rendering path in this library: every entry that produces pixels or hits
// launches HIP kernels, and pt_create fails when no GPU can be opened.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt_amd.h"
#include "device_types.h"
#include "scene.h"

using namespace pt;

namespace {

struct TimedSpan {
    hipEvent_t start, stop;
    int timer;
};

template <typename T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    hipError_t ensure(size_t n) {
        if (n <= count && ptr) return hipSuccess;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; count = 0; }
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&ptr, n * sizeof(T));
        if (e == hipSuccess) count = n;
        return e;
    }
    hipError_t upload(const T* src, size_t n, hipStream_t s) {
        hipError_t e = ensure(n > 0 ? n : 1);
        if (e != hipSuccess || n == 0) return e;
        return hipMemcpyAsync(ptr, src, n * sizeof(T), hipMemcpyHostToDevice, s);
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; count = 0;
    }
};

}  // namespace

struct PtGroup;

// What is needed to render a batch again should its overflow flag turn out set.
struct IssuedBatch {
    CameraGpu cam;
    int pixel0, n_pixels, sample0, chunk;
    int flag;  // index into the flag ring
};
enum { FLAG_RING = 1024, MAX_LANES = 4 };

// Batches in flight: consecutive batches go to the lanes in turn, each lane with its own stream and
// its own queues, so that the sparsely occupied late levels of one batch (and every kernel's
// ramp-down) run beside the dense early levels of the next.  Only the kernels that touch the image
// (accumulate, tally; resolve and clear anyway) run on the context's stream, in program order,
// which keeps the image's summation order and everything a caller orders on that stream as with
// one lane.
struct Lane {
    hipStream_t stream = nullptr;         // non-blocking; with one lane the context's stream is used instead
    DeviceBuffer<float4> q_ray_o[2], q_ray_d[2], q_beta[2], d_hit, d_sample_rad;
    DeviceBuffer<float4> d_eye;           // origin record shared by the camera rays of the lane's batch
    DeviceBuffer<uint64_t> d_keys;        // sampler keys of the lane's batch
    DeviceBuffer<int> d_qcount;
    DeviceBuffer<int> d_cursors;          // pool cursors of the batch's trace launches: TRACE_POOL_CURSORS per level, cleared per batch
    DeviceBuffer<unsigned long long> d_block_stats;
    size_t batch_alloc = 0;
    hipEvent_t done[2] = {nullptr, nullptr};      // the lane's kernels of a batch have been queued up to here
    hipEvent_t consumed[2] = {nullptr, nullptr};  // the context's stream has read the batch's results
    uint64_t uses = 0;
    void release() {
        for (int k = 0; k < 2; k++) { q_ray_o[k].release(); q_ray_d[k].release(); q_beta[k].release(); }
        d_hit.release(); d_sample_rad.release(); d_eye.release(); d_keys.release(); d_qcount.release(); d_cursors.release(); d_block_stats.release();
        batch_alloc = 0;
    }
};

struct PtContext {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int blocks = 2048;       // most workgroups a batch is ever split over (sizes the per-workgroup buffers)
    int cus = 256;
    int resident_per_cu = 0; // workgroups of the trace kernel a CU holds for the current scene and settings (0 = not asked yet)
    int blocks_per_cu = 0;   // workgroups per CU a batch is split over; 0 = what the trace kernel keeps resident
    std::string error;

    // host scene
    HostMesh mesh;
    HostImage env;
    HostTree tree;
    bool built = false;
    int accel = PT_ACCEL_BVH;

    // camera and film (defaults of src/cs488.h:48-49, 74-85)
    float eye[3] = {0.0f, 0.0f, 1.5f}, lookat[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    float fov = 45.0f, film = 0.032f;
    int width = 0, height = 0;
    int total_samples = 0;

    // options
    uint64_t seed = 0xcafef00dd15ea5e5ull;
    int max_level = 15;
    int count_tests = 0, profile = 0, traversal = 0x01;  // reciprocal-form slab test; leave rule and refill threshold chosen per scene
    size_t max_batch = (size_t)1 << 24;
    int queue_factor = 4;
    int lds_nodes_opt = -1;  // tree nodes staged in LDS; -1 = chosen per scene
    size_t kd_inline_refs = (size_t)1 << 24;  // kd-trees with more leaf references (1 GiB of records) keep one record per triangle behind index leaves
    int obj_extended = 0;   // pt_load_obj also takes polygons and negative indices
    int build_threads = 0;  // host threads of the tree builders, 0 = all
    int tail_level = 255;  // levels from here on run in the fused kernel; 255 = per scene (6 or 8), 0 = never
    int tail_level_used = 0;  // where the fused kernel took over in the batch queued last (0 = it did not run)

    // device scene
    DeviceBuffer<float4> d_nodes, d_wide, d_isect, d_tri_pos, d_tri_nrm, d_mats;
    DeviceBuffer<float2> d_tri_uv;
    DeviceBuffer<int> d_refs;
    DeviceBuffer<uint8_t> d_texels;
    DeviceBuffer<float> d_env;
    SceneGpu scene = {};

    // film and queues
    DeviceBuffer<float> d_accum, d_frame;
    float* ext_accum = nullptr;
    PtGroup* group = nullptr;             // set while the context is a member of a multi-GPU group
    Lane lanes[MAX_LANES];
    int n_lanes = 3;                      // batches in flight (PT_OPT_STREAMS)
    bool lanes_auto = true;               // not set by the caller: chosen per scene and frame size (choose_lanes); two once a collective library shares the GPU's queues
    size_t lanes_for_frame = 0;           // samples per frame the choice was made for (0 = not made)
    size_t frame_samples = 0;             // samples of the frame being queued (this context's share)
    // Rays per bounce level of an earlier batch, read back without a wait (run_batch): where the fused tail takes over
    // follows from them once they are there.
    enum { LEVEL_WORDS = 80 };
    DeviceBuffer<unsigned long long> d_level_rays;
    unsigned long long* h_level_rays = nullptr;   // pinned
    hipEvent_t level_copied = nullptr;
    bool level_copy_pending = false;
    int level_copy_slots = 0, level_copy_levels = 0;   // batch the pending copy describes
    std::vector<unsigned long long> level_rays;      // the last profile that arrived ...
    int level_rays_slots = 0;                        // ... and the batch size it was taken at (0 = none)
    bool lanes_checked = false;           // the lanes' streams have been probed for running side by side
    int lane_stream_swaps = 0;            // streams replaced by that probe
    DeviceBuffer<unsigned long long> d_totals;
    DeviceBuffer<unsigned long long> d_block_stats;  // of pt_trace_batch (the lanes have their own)
    DeviceBuffer<int> d_cursors;          // pool cursors of pt_trace_batch's launch, cleared before each
    DeviceBuffer<int> d_batch_flags;      // FLAG_RING overflow flags, one per batch in flight
    std::vector<IssuedBatch> issued;      // batches whose flags have not been looked at yet
    std::deque<hipEvent_t> frame_ends;    // one event per frame queued by pt_render / pt_render_split, oldest first (pt_wait_frames)
    uint64_t batch_seq = 0, repaired_batches = 0;
    size_t batch_hint = 0;                // samples of the largest batch of the frame being rendered
    uint64_t batch_rays = 0;  // rays traced through pt_trace_batch
    // staging of pt_trace_batch, kept between calls and grown on demand
    DeviceBuffer<float> tb_rays, tb_t, tb_bary;
    DeviceBuffer<int> tb_tri;
    DeviceBuffer<float4> tb_o, tb_d, tb_h;

    float4* static_scratch = nullptr;  // this device's copy of the file-scope staging array of pt_trace_batch_static

    GifWriter recording;

    // measurement
    uint64_t samples_started = 0;
    std::vector<TimedSpan> spans;
    std::deque<hipEvent_t> event_pool;
    double timer_ms[PT_TIMER_COUNT] = {};
    uint64_t timer_launches[PT_TIMER_COUNT] = {};

    float* accum_ptr() { return ext_accum ? ext_accum : d_accum.ptr; }
};

namespace {

void choose_lds_nodes(PtContext* c);

int fail(PtContext* c, int code, const std::string& msg) {
    if (c) c->error = msg;
    return code;
}

#define PT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail(ctx, PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));     \
    } while (0)

// No exception crosses the C boundary: the bodies of the entry points that allocate host memory or
// start threads run inside PT_TRY / PT_CATCH.
#define PT_TRY try {
#define PT_CATCH(ctx)                                                                             \
    }                                                                                             \
    catch (const std::bad_alloc&) { return fail_noexcept(ctx, PT_ERR_NO_MEMORY, "out of host memory"); }     \
    catch (const std::exception& e__) { return fail_noexcept(ctx, PT_ERR_NO_MEMORY, e__.what()); }            \
    catch (...) { return fail_noexcept(ctx, PT_ERR_NO_MEMORY, "unexpected exception"); }

int fail_noexcept(PtContext* c, int code, const char* msg) noexcept {
    try { if (c) c->error = msg; } catch (...) {}
    return code;
}

LaunchCfg cfg_of(PtContext* c) {
    LaunchCfg k;
    k.stream = c->stream;
    k.blocks = c->blocks;
    return k;
}

hipEvent_t take_event(PtContext* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.front();
        c->event_pool.pop_front();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;  // the span is then left out (ScopedTimer)
    return e;
}

// Marks the end of a frame on the context's stream for pt_wait_frames.  At most 64 marks are kept: a caller
// that never waits loses the oldest ones to a wait here, which is long over by then.
int mark_frame_end(PtContext* c) {
    while (c->frame_ends.size() >= 64) {
        (void)hipEventSynchronize(c->frame_ends.front());
        c->event_pool.push_back(c->frame_ends.front());
        c->frame_ends.pop_front();
    }
    hipEvent_t e = take_event(c);
    if (!e) return fail(c, PT_ERR_HIP, "no event for the end of the frame");
    PT_HIP(c, hipEventRecord(e, c->stream));
    c->frame_ends.push_back(e);
    return PT_OK;
}

struct ScopedTimer {
    PtContext* c;
    TimedSpan span;
    bool on;
    // profile 1: every stage; profile 2: the closest-hit kernel only (fewer events in the stream)
    hipStream_t stream;
    ScopedTimer(PtContext* ctx, int timer, hipStream_t on_stream = nullptr)
        : c(ctx), on(ctx->profile == 1 || (ctx->profile == 2 && timer == PT_TIMER_TRACE)), stream(on_stream ? on_stream : ctx->stream) {
        if (!on) return;
        span.timer = timer;
        span.start = take_event(c);
        span.stop = take_event(c);
        if (!span.start || !span.stop) {  // no event to be had: this launch goes untimed
            if (span.start) c->event_pool.push_back(span.start);
            if (span.stop) c->event_pool.push_back(span.stop);
            on = false;
            return;
        }
        (void)hipEventRecord(span.start, stream);
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(span.stop, stream);
        c->spans.push_back(span);
    }
};

// folds finished event pairs into the timers; the stream must be idle
void drain_spans(PtContext* c) {
    for (TimedSpan& s : c->spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess) {
            c->timer_ms[s.timer] += ms;
            c->timer_launches[s.timer] += 1;
        }
        c->event_pool.push_back(s.start);
        c->event_pool.push_back(s.stop);
    }
    c->spans.clear();
}

void to_public(const HostMaterial& m, PtMaterial* o) {
    memset(o, 0, sizeof(*o));
    snprintf(o->name, sizeof(o->name), "%s", m.name.c_str());
    o->type = m.type; o->eta = m.eta;
    o->Ka[0] = m.Ka.x; o->Ka[1] = m.Ka.y; o->Ka[2] = m.Ka.z;
    o->Kd[0] = m.Kd.x; o->Kd[1] = m.Kd.y; o->Kd[2] = m.Kd.z;
    o->Ks[0] = m.Ks.x; o->Ks[1] = m.Ks.y; o->Ks[2] = m.Ks.z;
    o->Ke[0] = m.Ke.x; o->Ke[1] = m.Ke.y; o->Ke[2] = m.Ke.z;
    o->Ns = m.Ns; o->opacity = m.opacity; o->roughness = m.roughness;
    o->textured = m.textured ? 1 : 0;
    o->tex_width = m.tex_w; o->tex_height = m.tex_h;
}

void from_public(const PtMaterial& o, HostMaterial& m) {
    m = HostMaterial();
    m.name = std::string(o.name, strnlen(o.name, sizeof(o.name)));
    m.type = o.type; m.eta = o.eta;
    m.Ka = v3(o.Ka[0], o.Ka[1], o.Ka[2]); m.Kd = v3(o.Kd[0], o.Kd[1], o.Kd[2]);
    m.Ks = v3(o.Ks[0], o.Ks[1], o.Ks[2]); m.Ke = v3(o.Ke[0], o.Ke[1], o.Ke[2]);
    m.Ns = o.Ns; m.opacity = o.opacity; m.roughness = o.roughness;
    m.textured = o.textured != 0;
}

// Host tree -> device arrays (layout: device_types.h).
int flatten_and_upload(PtContext* c) {
    const HostTree& t = c->tree;
    const int n = (int)t.nodes.size();
    if (n == 0) return fail(c, PT_ERR_NOT_READY, "empty tree");
    if (t.max_depth > TRACE_STACK) return fail(c, PT_ERR_TREE_DEPTH, "tree depth " + std::to_string(t.max_depth) + " exceeds the traversal stack");

    std::vector<int> gpu_index(n, -1), order;
    if (t.nodes[0].left >= 0) {
        order.push_back(0);
        gpu_index[0] = 0;
        for (size_t head = 0; head < order.size(); head++) {
            const HostTree::Node& nd = t.nodes[order[head]];
            const int kids[2] = {nd.left, nd.right};
            for (int k : kids) {
                if (t.nodes[k].left >= 0) { gpu_index[k] = (int)order.size(); order.push_back(k); }
            }
        }
    }
    // intersection records (layout: device_types.h); the arithmetic is the triangle test's own
    const HostMesh& hm = c->mesh;
    auto record = [&](int tri, int count, std::vector<float4>& out) {
        const vec3 p0 = hm.p(tri, 0);
        const vec3 a = p0 - hm.p(tri, 1);
        const vec3 b = p0 - hm.p(tri, 2);
        const vec3 ab = cross(a, b);
        float fid, fcount;
        memcpy(&fid, &tri, 4); memcpy(&fcount, &count, 4);
        out.push_back(make_float4(p0.x, p0.y, p0.z, fid));
        out.push_back(make_float4(a.x, a.y, a.z, fcount));
        out.push_back(make_float4(b.x, b.y, b.z, 0.0f));
        out.push_back(make_float4(ab.x, ab.y, ab.z, 0.0f));
    };
    // kd-tree leaves repeat triangles; their records are stored per leaf (64 bytes a reference: a leaf
    // is one contiguous run, which measured 23 % faster than index leaves on a 1.4 M-reference tree
    // although it is seven times the bytes) unless that takes more than 1 GiB, and once per triangle
    // behind `refs` then
    // -- and also for a mesh small enough that the whole tree is staged in LDS, where a leaf
    // reference then costs 4 bytes instead of 64
    const bool kd = t.kind == 1 && (t.refs.size() > c->kd_inline_refs || hm.num_tris() <= 256);
    std::vector<int> refs;
    std::vector<float4> isect;
    if (kd) {
        for (int i = 0; i < hm.num_tris(); i++) record(i, 0, isect);
    }
    auto leaf_reference = [&](int node) -> int {
        const HostTree::Node& nd = t.nodes[node];
        // a kd-tree has leaves without triangles (src/cs488.h:1306-1311 stops on cost, not on emptiness);
        // visiting one changes nothing, so the walk is told not to enter it
        if (nd.count == 0 && node != 0) return EMPTY_LEAF;
        if (kd) {
            const int off = (int)refs.size();
            refs.push_back(nd.count);
            for (int k = 0; k < nd.count; k++) refs.push_back(t.refs[nd.first + k]);
            return ~off;
        }
        const int off = (int)(isect.size() / 4);
        if (nd.count == 0) record(0, 0, isect);  // an empty leaf still needs a record to hold its count
        for (int k = 0; k < nd.count; k++) record(t.refs[nd.first + k], k == 0 ? nd.count : 0, isect);
        return ~off;
    };
    std::vector<float4> nodes(order.size() * 4);
    for (size_t g = 0; g < order.size(); g++) {
        const HostTree::Node& nd = t.nodes[order[g]];
        const HostTree::Node& L = t.nodes[nd.left];
        const HostTree::Node& R = t.nodes[nd.right];
        const int lref = L.left >= 0 ? gpu_index[nd.left] : leaf_reference(nd.left);
        const int rref = R.left >= 0 ? gpu_index[nd.right] : leaf_reference(nd.right);
        nodes[4 * g + 0] = make_float4(L.box[0], L.box[1], L.box[2], L.box[3]);
        nodes[4 * g + 1] = make_float4(L.box[4], L.box[5], R.box[0], R.box[1]);
        nodes[4 * g + 2] = make_float4(R.box[2], R.box[3], R.box[4], R.box[5]);
        float4 q3;
        memcpy(&q3.x, &lref, 4); memcpy(&q3.y, &rref, 4);
        q3.z = 0.0f; q3.w = 0.0f;
        nodes[4 * g + 3] = q3;
    }
    int root_ref = 0;
    if (t.nodes[0].left < 0) root_ref = leaf_reference(0);

    // triangles and materials
    const HostMesh& m = c->mesh;
    const int nt = m.num_tris();
    std::vector<float4> tp(3 * (size_t)nt), tn(3 * (size_t)nt);
    bool any_tex = false;
    for (const HostMaterial& hm : m.materials) any_tex = any_tex || (hm.textured && !hm.tex.empty());
    std::vector<float2> tuv(any_tex ? 3 * (size_t)nt : 0);
    for (int i = 0; i < nt; i++) {
        for (int k = 0; k < 3; k++) {
            float w = 0.0f;
            if (k == 0) { int id = m.mat[i]; memcpy(&w, &id, 4); }
            tp[3 * i + k] = make_float4(m.pos[9 * i + 3 * k], m.pos[9 * i + 3 * k + 1], m.pos[9 * i + 3 * k + 2], w);
            tn[3 * i + k] = make_float4(m.nrm[9 * i + 3 * k], m.nrm[9 * i + 3 * k + 1], m.nrm[9 * i + 3 * k + 2], 0.0f);
            if (any_tex) tuv[3 * i + k] = make_float2(m.uv[6 * i + 2 * k], m.uv[6 * i + 2 * k + 1]);
        }
    }
    std::vector<float4> mats(MATERIAL_STRIDE * m.materials.size());
    std::vector<uint8_t> texels;
    for (size_t i = 0; i < m.materials.size(); i++) {
        const HostMaterial& hm = m.materials[i];
        int type = hm.type, tw = 0, th = 0, off = 0;
        if (hm.textured && !hm.tex.empty()) {  // a texture that failed to decode renders untextured
            tw = hm.tex_w; th = hm.tex_h; off = (int)texels.size();
            texels.insert(texels.end(), hm.tex.begin(), hm.tex.end());
        }
        float ft, fw, fh, fo;
        memcpy(&ft, &type, 4); memcpy(&fw, &tw, 4); memcpy(&fh, &th, 4); memcpy(&fo, &off, 4);
        float4* rec = &mats[MATERIAL_STRIDE * i];
        rec[0] = make_float4(hm.Kd.x, hm.Kd.y, hm.Kd.z, ft);
        rec[1] = make_float4(hm.Ks.x, hm.Ks.y, hm.Ks.z, hm.eta);
        rec[2] = make_float4(hm.Ke.x, hm.Ke.y, hm.Ke.z, hm.opacity);
        rec[3] = make_float4(hm.roughness, fw, fh, fo);
        const vec3 lambert = hm.Kd / PT_PI;  // Material::BRDF (src/cs488.h:417-428)
        rec[4] = make_float4(lambert.x, lambert.y, lambert.z, 0.0f);
    }
    for (int id : m.mat) {
        if (id < 0 || id >= (int)m.materials.size()) return fail(c, PT_ERR_INVALID_ARG, "triangle with a material index outside the material table");
    }

    PT_HIP(c, c->d_nodes.upload(nodes.data(), nodes.size(), c->stream));
    PT_HIP(c, c->d_refs.upload(refs.data(), refs.size(), c->stream));
    PT_HIP(c, c->d_isect.upload(isect.data(), isect.size(), c->stream));
    PT_HIP(c, c->d_tri_pos.upload(tp.data(), tp.size(), c->stream));
    PT_HIP(c, c->d_tri_nrm.upload(tn.data(), tn.size(), c->stream));
    PT_HIP(c, c->d_tri_uv.upload(tuv.data(), tuv.size(), c->stream));
    PT_HIP(c, c->d_mats.upload(mats.data(), mats.size(), c->stream));
    PT_HIP(c, c->d_texels.upload(texels.data(), texels.size(), c->stream));
    PT_HIP(c, c->d_env.upload(c->env.rgb.data(), c->env.rgb.size(), c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));  // the staging vectors die with this scope

    SceneGpu& s = c->scene;
    s.wide = nullptr;  // built on demand (ensure_wide_nodes)
    s.nodes = c->d_nodes.ptr; s.probe_alias = c->d_nodes.ptr; s.refs = c->d_refs.ptr; s.isect = c->d_isect.ptr;
    s.tri_pos = c->d_tri_pos.ptr; s.tri_nrm = c->d_tri_nrm.ptr; s.tri_uv = c->d_tri_uv.ptr;
    s.mats = c->d_mats.ptr; s.texels = c->d_texels.ptr; s.env = c->d_env.ptr;
    s.env_w = c->env.w; s.env_h = c->env.h;
    for (int k = 0; k < 3; k++) { s.root_lo[k] = t.nodes[0].box[k]; s.root_hi[k] = t.nodes[0].box[3 + k]; }
    s.root_ref = root_ref;
    s.n_nodes = (int)order.size();
    s.n_tris = nt;
    s.n_mats = (int)m.materials.size();
    s.leaf_inline = kd ? 0 : 1;
    s.n_records = (int)(isect.size() / 4);
    s.n_refs = (int)refs.size();
    s.shared_origin = 0;
    bool metal = false, glass = false, other = false, plain = c->env.h == 0;
    for (const HostMaterial& hm2 : m.materials) {
        if (hm2.type == MAT_METAL) metal = true;
        else if (hm2.type == MAT_GLASS) glass = true;
        else if (hm2.type != MAT_LAMBERTIAN) other = true;
        if (hm2.textured || !(hm2.opacity >= 1.0f - PT_EPSILON)) plain = false;
    }
    // SHADE_ALL 0 / SHADE_LAMBERTIAN 1 / SHADE_PLAIN 2 / SHADE_GLASS 3 / SHADE_METAL 4 of shade_device.h
    s.fixed_radiance = 0;
    for (const HostMaterial& hm2 : m.materials)
        if (!(hm2.opacity >= 1.0f - PT_EPSILON)) s.fixed_radiance = 1;  // the alpha fork of src/cs488.h:2038
    s.shading_kind = (other || (metal && glass)) ? 0 : (glass ? 3 : (metal ? 4 : (plain ? 2 : 1)));
    choose_lds_nodes(c);
    c->lanes_for_frame = 0;  // batches in flight are chosen again at the next frame (choose_lanes)
    c->level_rays_slots = 0;  // and the rays per level measured on the scene before say nothing about this one
    s.planes_ok = 1;
    for (const HostTree::Node& nd : t.nodes)
        for (int k = 0; k < 6; k++)
            if (nd.box[k] != 0.0f && !(std::fabs(nd.box[k]) >= 1e-17f)) s.planes_ok = 0;
    return PT_OK;
}

// The 192-byte node records of the wide BVH walk (PT_OPT_TRAVERSAL bit 15; trace_region's WIDE form): per interior
// node its 64-byte record followed by those of its left and right child (zeros where a child is a leaf).  Built from
// the uploaded 64-byte nodes the first time the setting asks for them: three times the node bytes, which a scene
// that never uses the setting does not pay.
int ensure_wide_nodes(PtContext* c) {
    SceneGpu& s = c->scene;
    if (s.wide || !c->built || c->accel != PT_ACCEL_BVH || s.n_nodes <= 256) return PT_OK;
    const size_t n = (size_t)s.n_nodes;
    std::vector<float4> nodes(4 * n), wide(12 * n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, hipMemcpy(nodes.data(), c->d_nodes.ptr, nodes.size() * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t g = 0; g < n; g++) {
        int kid[2];
        memcpy(&kid[0], &nodes[4 * g + 3].x, 4); memcpy(&kid[1], &nodes[4 * g + 3].y, 4);
        for (int k = 0; k < 4; k++) wide[12 * g + k] = nodes[4 * g + k];
        for (int side = 0; side < 2; side++)
            if (kid[side] >= 0)
                for (int k = 0; k < 4; k++) wide[12 * g + 4 + 4 * side + k] = nodes[4 * (size_t)kid[side] + k];
    }
    PT_HIP(c, c->d_wide.upload(wide.data(), wide.size(), c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    s.wide = c->d_wide.ptr;
    c->resident_per_cu = 0;
    return PT_OK;
}

void camera_of(PtContext* c, CameraGpu& cam) {
    // Scene::generateRay (src/cs488.h:1825-1836) with the view direction of :2284 and the film
    // distance of :79
    const vec3 eye = v3(c->eye[0], c->eye[1], c->eye[2]);
    const vec3 view = normalize(v3(c->lookat[0], c->lookat[1], c->lookat[2]) - eye);
    const vec3 upn = normalize(v3(c->up[0], c->up[1], c->up[2]));
    const vec3 w = normalize(-view);
    const vec3 u = normalize(cross(upn, w));
    const vec3 v = cross(w, u);
    const float aspect = float(c->width / float(c->height));
    const float deg_to_rad = PT_PI / 180.0f;
    cam.eye[0] = eye.x; cam.eye[1] = eye.y; cam.eye[2] = eye.z;
    cam.u[0] = u.x; cam.u[1] = u.y; cam.u[2] = u.z;
    cam.v[0] = v.x; cam.v[1] = v.y; cam.v[2] = v.z;
    cam.w[0] = w.x; cam.w[1] = w.y; cam.w[2] = w.z;
    cam.aspect_film = aspect * c->film;
    cam.film = c->film;
    cam.dist = c->film / (2.0f * tanf(c->fov * deg_to_rad * 0.5f));
    cam.width = c->width; cam.height = c->height;
}

// How many of the top tree nodes the trace kernels stage in LDS.  Unless set explicitly: the
// largest of 256 / 128 / 64 / 0 with which a CU still holds five workgroups of the trace kernel
// (a kd-tree's LDS stack is wider, so it gets fewer nodes); staging more trades resident waves,
// which hide the latency of the node fetches that stay in memory, for fewer such fetches.
void choose_lds_nodes(PtContext* c) {
    SceneGpu& s = c->scene;
    c->resident_per_cu = 0;
    if (c->lds_nodes_opt >= 0) {
        s.lds_nodes = c->lds_nodes_opt < s.n_nodes ? c->lds_nodes_opt : s.n_nodes;
        return;
    }
    const int candidates[4] = {256, 128, 64, 0};
    int best = 0, best_resident = -1;
    for (int cand : candidates) {
        s.lds_nodes = cand < s.n_nodes ? cand : s.n_nodes;
        const int resident = trace_resident_blocks_per_cu(s, c->accel == PT_ACCEL_KD, c->traversal);
        if (resident >= 5) return;
        if (resident > best_resident) { best_resident = resident; best = s.lds_nodes; }
    }
    s.lds_nodes = best;
}

// The lane's stream, events and per-workgroup statistics (first use) and queues for `batch` samples.
int ensure_lane(PtContext* c, Lane& ln, size_t batch) {
    if (c->n_lanes > 1 && !ln.stream) {
        // non-blocking: the context's stream may be the null stream, which would otherwise order itself
        // against every launch of the lane
        // (a process's streams share a few hardware queues, GPU_MAX_HW_QUEUES; two busy streams that land
        // on one queue run in order, which costs the overlap, never the result)
        PT_HIP(c, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            PT_HIP(c, hipEventCreateWithFlags(&ln.done[k], hipEventDisableTiming));
            PT_HIP(c, hipEventCreateWithFlags(&ln.consumed[k], hipEventDisableTiming));
        }
    }
    if (!ln.d_block_stats.ptr) {
        PT_HIP(c, ln.d_block_stats.ensure((size_t)c->blocks * DC_BLOCK_COUNT));
        // on the stream the lane's kernels run on: a lane's stream does not order itself against the null stream
        PT_HIP(c, hipMemsetAsync(ln.d_block_stats.ptr, 0, (size_t)c->blocks * DC_BLOCK_COUNT * sizeof(unsigned long long),
                                 c->n_lanes > 1 ? ln.stream : c->stream));
    }
    if (batch <= ln.batch_alloc) return PT_OK;
    const size_t cap = batch * (size_t)c->queue_factor;
    for (int k = 0; k < 2; k++) {
        PT_HIP(c, ln.q_ray_o[k].ensure(cap));
        PT_HIP(c, ln.q_ray_d[k].ensure(cap));
        PT_HIP(c, ln.q_beta[k].ensure(cap));
    }
    PT_HIP(c, ln.d_hit.ensure(cap));
    PT_HIP(c, ln.d_sample_rad.ensure(2 * batch));  // two records per sample where radiance is summed in fixed point
    ln.batch_alloc = batch;
    return PT_OK;
}

// per-workgroup statistics summed on the host; the stream must be idle
int read_block_stats(PtContext* c, unsigned long long out[DC_BLOCK_COUNT]) {
    std::vector<unsigned long long> h((size_t)c->blocks * DC_BLOCK_COUNT);
    for (int k = 0; k < DC_BLOCK_COUNT; k++) out[k] = 0;
    for (int l = -1; l < MAX_LANES; l++) {
        const unsigned long long* src = l < 0 ? c->d_block_stats.ptr : c->lanes[l].d_block_stats.ptr;
        if (!src) continue;
        PT_HIP(c, hipMemcpy(h.data(), src, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int b = 0; b < c->blocks; b++)
            for (int k = 0; k < DC_BLOCK_COUNT; k++) out[k] += h[(size_t)b * DC_BLOCK_COUNT + k];
    }
    return PT_OK;
}

int clear_block_stats(PtContext* c) {
    for (int l = -1; l < MAX_LANES; l++) {
        unsigned long long* dst = l < 0 ? c->d_block_stats.ptr : c->lanes[l].d_block_stats.ptr;
        if (dst) PT_HIP(c, hipMemset(dst, 0, (size_t)c->blocks * DC_BLOCK_COUNT * sizeof(unsigned long long)));
    }
    PT_HIP(c, hipStreamSynchronize(nullptr));  // the lanes' streams do not wait for the null stream by themselves
    return PT_OK;
}

// Do two streams run side by side?  A process's streams are mapped onto a few hardware queues
// (GPU_MAX_HW_QUEUES), and two that share one run in order.  Two 150-us single-wave kernels take
// 150 us when they overlap and 300 when they do not.
bool streams_overlap(PtContext* c, hipStream_t a, hipStream_t b) {
    LaunchCfg ka = cfg_of(c), kb = cfg_of(c);
    ka.stream = a; kb.stream = b;
    const int us = 150;
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    const auto t0 = std::chrono::steady_clock::now();
    (void)launch_spin(ka, us);
    (void)launch_spin(kb, us);
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    const double elapsed = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return elapsed < 1.6 * us;
}

// The overlap of consecutive batches needs the lanes' streams and the context's stream on hardware
// queues of their own.  Which queue a new stream gets is the runtime's business, so the lanes are
// probed once (and again after pt_set_stream) and a lane whose stream runs in order with the
// context's stream or with an earlier lane gets a new stream -- created while the old one still
// exists, so that it lands elsewhere -- a few times at most.  Only speed depends on this.
int check_lanes(PtContext* c) {
    c->lanes_checked = true;
    if (c->n_lanes < 2) return PT_OK;
    for (int l = 0; l < c->n_lanes; l++) {
        Lane& ln = c->lanes[l];
        if (!ln.stream) continue;
        for (int attempt = 0; attempt < 6; attempt++) {
            bool clash = !streams_overlap(c, ln.stream, c->stream);
            for (int m = 0; m < l && !clash; m++)
                if (c->lanes[m].stream) clash = !streams_overlap(c, ln.stream, c->lanes[m].stream);
            if (!clash) break;
            hipStream_t fresh = nullptr;
            PT_HIP(c, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
            (void)hipStreamSynchronize(ln.stream);
            (void)hipStreamDestroy(ln.stream);
            ln.stream = fresh;
            c->lane_stream_swaps++;
        }
    }
    return PT_OK;
}

// How a batch is cut into workgroup regions: about 2048 camera samples each, at most the resident grid.
void shape_batch(PtContext* c, const Lane& ln, bool own_stream, BatchGpu& b) {
    int nb = (b.n_slots + 2047) / 2048;
    if (c->resident_per_cu <= 0) c->resident_per_cu = trace_resident_blocks_per_cu(c->scene, c->accel == PT_ACCEL_KD, c->traversal);
    // with another batch in flight beside this one, five regions per CU leave its kernels room (measured: 5 or 6
    // instead of 8 gains 3-5 % on the cornellbox and the teapot, nothing elsewhere); with two others, four do
    // (a tree walked in LDS +0.5-1.5 %; the teapot's kd-tree +6 %, the glass teapot's +8 %, its BVH and the bunny's
    // +1-2 %, a 210 k-triangle courtyard -0.5 %)
    const int beside = !own_stream ? c->resident_per_cu : (c->n_lanes >= 3 ? 4 : 5);
    const int per_cu = c->blocks_per_cu > 0 ? c->blocks_per_cu : (beside < c->resident_per_cu ? beside : c->resident_per_cu);
    const int most = c->cus * per_cu;
    nb = nb < 1 ? 1 : (nb > most ? most : nb);
    b.blocks = nb;
    b.slots_block = (((b.n_slots + 63) / 64 + nb - 1) / nb) * 64;  // runs of 64 samples dealt in rounds
    b.deal_shift = nb > 2 ? ((int)(nb * 0.6180339887) | 1) : 0;
    b.cap_block = (int)((ln.batch_alloc * (size_t)c->queue_factor) / (size_t)nb);
}

// One wavefront batch: camera rays, then trace + shade per bounce level on the batch's lane, then
// accumulate on the context's stream.
int run_batch(PtContext* c, const CameraGpu& cam, int pixel0, int n_pixels, int sample0, int chunk) {
    BatchGpu b;
    b.pixel0 = pixel0; b.n_pixels = n_pixels; b.chunk = chunk; b.sample0 = sample0;
    b.seed = c->seed;
    b.fixed_radiance = c->scene.fixed_radiance;
    b.n_slots = n_pixels * chunk;
    Lane& ln = c->lanes[c->batch_seq % (uint64_t)c->n_lanes];
    const bool own_stream = c->n_lanes > 1;
    {
        const size_t need = (size_t)b.n_slots > c->batch_hint ? (size_t)b.n_slots : c->batch_hint;
        int rc = ensure_lane(c, ln, need);
        if (rc != PT_OK) return rc;
        if (own_stream && !c->lanes_checked) {  // once: every lane's stream, then the probe for hardware queues of their own
            for (int l = 0; l < c->n_lanes && rc == PT_OK; l++) rc = ensure_lane(c, c->lanes[l], need);
            if (rc == PT_OK) rc = check_lanes(c);
            if (rc != PT_OK) return rc;
        }
    }
    shape_batch(c, ln, own_stream, b);
    const int nb = b.blocks;
    IssuedBatch rec;
    rec.cam = cam; rec.pixel0 = pixel0; rec.n_pixels = n_pixels; rec.sample0 = sample0; rec.chunk = chunk;
    rec.flag = (int)(c->batch_seq++ % FLAG_RING);
    b.overflow = c->d_batch_flags.ptr + rec.flag;
    PT_HIP(c, ln.d_keys.ensure((size_t)(chunk > 1024 ? chunk : 1024)));
    PT_HIP(c, ln.d_eye.ensure(1));
    b.keys = ln.d_keys.ptr;
    const LaunchCfg image_cfg = cfg_of(c);  // kernels that touch the image
    LaunchCfg cfg = image_cfg;              // kernels of the batch proper
    if (own_stream) cfg.stream = ln.stream;
    const int turn = (int)(ln.uses++ & 1);
    // the lane's previous batch must have been accumulated before its buffers are written again
    if (own_stream && ln.uses > 1) PT_HIP(c, hipStreamWaitEvent(ln.stream, ln.consumed[turn ^ 1], 0));
    const int levels = c->max_level + 2;  // levels 0 .. maxLevel+1; the last one only resolves misses
    // every stage writes every workgroup's count of the next level, so the counts need no clearing
    PT_HIP(c, ln.d_qcount.ensure(((size_t)levels + 1) * (size_t)c->blocks));
    PT_HIP(c, ln.d_cursors.ensure((size_t)levels * TRACE_POOL_INTS));
    QueueSet q[2];
    for (int k = 0; k < 2; k++) { q[k].ray_o = ln.q_ray_o[k].ptr; q[k].ray_d = ln.q_ray_d[k].ptr; q[k].beta = ln.q_beta[k].ptr; }
    // camera rays are stored as directions only: one shared origin record, no throughputs
    QueueSet camera_rays = q[0];
    camera_rays.ray_o = ln.d_eye.ptr;
    camera_rays.beta = nullptr;
    const SceneGpu& scene = c->scene;
    SceneGpu scene0 = c->scene;
    scene0.shared_origin = 1;
    // One batch at a time: the trace launches draw their rays from shared pools, so that every wave works until a
    // launch's rays are used up.  With another batch in flight private regions are faster (its kernels fill what the
    // workgroups that finish early leave; measured 15.5 against 14.9 Grays/s on the cornellbox, 12.5 against 11.1 alone).
    const int trace_mode = c->traversal | (own_stream ? 8 : 0);
    if ((trace_mode & 8) == 0)
        PT_HIP(c, hipMemsetAsync(ln.d_cursors.ptr, 0, (size_t)levels * TRACE_POOL_INTS * sizeof(int), cfg.stream));
    {
        ScopedTimer tm(c, PT_TIMER_RAYGEN, cfg.stream);
        PT_HIP(c, launch_raygen(cfg, cam, b, camera_rays, ln.d_sample_rad.ptr, ln.d_qcount.ptr));
    }
    // the deep, sparsely populated levels run in one fused launch (not while tests are counted or
    // the static refill is being measured: those are properties of the per-level kernels)
    // Russian roulette starts at level 5 (src/cs488.h:2153); a BVH walked in LDS is handed over one level later
    // (earlier is as fast, but keeps the per-level kernels measurable), a kd-tree walked in LDS after the camera
    // rays when the shading code is the light one (measured 9 % faster than at level 6; with the heavier shading
    // instantiations the fused kernel's registers cost the walk more than that), any other scene when its levels have thinned out further (the
    // fused kernel keeps fewer waves resident, which shows where node fetches come from memory)
    const bool in_lds = c->scene.lds_nodes >= c->scene.n_nodes;
    const bool light_shading = c->scene.shading_kind == 2;  // SHADE_PLAIN: the fused kernel keeps the walk's occupancy
    // (with a second batch in flight the launch gaps that the early hand-over saved are filled anyway: 6 is 3 % faster then)
    // the batches of a small frame (the reference's own 512 x 384 x 10) are handed over after the first bounce: their per-level launches are too
    // short to fill the chip (14.0 -> 15.0 Grays/s with a BVH, 9.4 -> 9.9 with a kd-tree; 3 and later are slower again)
    const bool small_batch = c->frame_samples <= ((size_t)4 << 20) && c->frame_samples > 0;
    // (a tree in memory: 4, and 2 for a small batch as well -- teapot 512 x 384: 5.8 -> 7.5 Grays/s, its kd-tree 2.8 -> 3.8; at 1024 x 1024
    // 11.0 -> 11.7; the 210 k-triangle courtyard +1.4 %; the deep levels of an open scene are nearly empty launches otherwise)
    // (a small batch of a kd-tree in LDS, or of any tree in LDS under the heavier shading kinds, already after the camera rays: at 512 x 384 x 10
    // cornellbox kd-tree 8.87 -> 9.16 Grays/s, cornellbox-glass 11.99 -> 12.48, its kd-tree 7.85 -> 8.20; the cornellbox BVH loses 1.3 % there)
    const bool early = in_lds && (c->accel == PT_ACCEL_KD || !light_shading);
    int tail = c->tail_level == 255 ? (in_lds ? (c->accel == PT_ACCEL_KD && light_shading && !own_stream ? 1 : (small_batch ? (early ? 1 : 2) : 6)) : (small_batch ? 2 : 4)) : c->tail_level;
    // ... and, once the rays per level of an earlier batch of this size have come back (no wait: the copy is looked at when
    // its event has passed), from those: the fused launch takes over at the first level that holds fewer than 2.4 M rays
    // (level 1: fewer than 0.6 M), at level 6 at the latest (4 for a tree in memory).  Measured optima this reproduces: cornellbox 1024 x 1024 6
    // (level 5 holds 2.45 M), at 512 x 384 2; teapot 2, at 512 x 384 1 (+6 % over 2); bunny 2 (+4 % over 4); the courtyard 4.
    if (c->level_copy_pending && hipEventQuery(c->level_copied) == hipSuccess) {
        c->level_copy_pending = false;
        c->level_rays.assign(c->h_level_rays, c->h_level_rays + c->level_copy_levels);
        c->level_rays_slots = c->level_copy_slots;
    }
    if (c->tail_level == 255 && c->level_rays_slots == b.n_slots && !(in_lds && c->accel == PT_ACCEL_KD && light_shading && !own_stream)) {
        tail = in_lds ? 6 : 4;  // the latest: where a tree of that kind is handed over with every level well filled
        for (int level = 1; level < tail && level < (int)c->level_rays.size(); level++)
            if (c->level_rays[level] < (level == 1 && !early ? 600000ull : 2400000ull)) { tail = level; break; }
    }
    const bool fuse = tail > 0 && tail < levels && !c->count_tests && (c->traversal & 4) == 0;
    const int split = fuse ? tail : levels;
    c->tail_level_used = fuse ? tail : 0;
    for (int level = 0; level < split; level++) {
        const QueueSet& in = level == 0 ? camera_rays : q[level & 1];
        const QueueSet& out = q[(level + 1) & 1];
        {
            ScopedTimer tm(c, PT_TIMER_TRACE, cfg.stream);
            PT_HIP(c, launch_trace(cfg, level == 0 ? scene0 : scene, c->accel == PT_ACCEL_KD, c->count_tests, trace_mode, in.ray_o, in.ray_d,
                                   ln.d_hit.ptr, ln.d_qcount.ptr + (size_t)level * nb, 0, nb, b.cap_block,
                                   ln.d_cursors.ptr + (size_t)level * TRACE_POOL_INTS, ln.d_block_stats.ptr));
        }
        {
            ScopedTimer tm(c, PT_TIMER_SHADE, cfg.stream);
            PT_HIP(c, launch_shade(cfg, scene, b, level, c->max_level, in, ln.d_hit.ptr, out, ln.d_sample_rad.ptr,
                                   ln.d_qcount.ptr + (size_t)level * nb, ln.d_qcount.ptr + (size_t)(level + 1) * nb));
        }
    }
    if (fuse) {
        ScopedTimer tm(c, PT_TIMER_TAIL, cfg.stream);
        PT_HIP(c, launch_tail(cfg, scene, b, c->accel == PT_ACCEL_KD, c->traversal, split, levels, c->max_level, q[0], q[1],
                              ln.d_hit.ptr, ln.d_sample_rad.ptr, ln.d_qcount.ptr, ln.d_block_stats.ptr));
    }
    if (own_stream) {
        PT_HIP(c, hipEventRecord(ln.done[turn], ln.stream));
        PT_HIP(c, hipStreamWaitEvent(c->stream, ln.done[turn], 0));
    }
    {
        ScopedTimer tm(c, PT_TIMER_ACCUM);
        PT_HIP(c, launch_accumulate(image_cfg, b, ln.d_sample_rad.ptr, c->accum_ptr(), c->d_totals.ptr));
        const bool profile_levels = !c->level_copy_pending && levels <= PtContext::LEVEL_WORDS;
        PT_HIP(c, launch_tally(image_cfg, ln.d_qcount.ptr, levels, nb, split, b.overflow, c->d_totals.ptr, profile_levels ? c->d_level_rays.ptr : nullptr));
        if (profile_levels) {  // one copy in flight at a time; looked at when a later batch is queued
            PT_HIP(c, hipMemcpyAsync(c->h_level_rays, c->d_level_rays.ptr, (size_t)levels * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            PT_HIP(c, hipEventRecord(c->level_copied, c->stream));
            c->level_copy_pending = true;
            c->level_copy_slots = b.n_slots;
            c->level_copy_levels = levels;
        }
    }
    if (own_stream) PT_HIP(c, hipEventRecord(ln.consumed[turn], c->stream));
    c->samples_started += (uint64_t)b.n_slots;
    c->issued.push_back(rec);
    return PT_OK;
}

// Waits for the stream and renders every batch whose queues overflowed again, in halves (half the
// samples per pixel, or half the pixels once a batch holds one sample per pixel): the queue memory
// stays the same, so a workgroup's region doubles relative to its samples with every halving.  A
// flagged batch added nothing to the image (accumulate_kernel), so the pieces complete it exactly.
int settle(PtContext* c) {
    PT_HIP(c, hipSetDevice(c->device));  // the caller's thread may have another device current
    PT_HIP(c, hipStreamSynchronize(c->stream));
    for (hipEvent_t e : c->frame_ends) c->event_pool.push_back(e);  // every frame queued so far has ended
    c->frame_ends.clear();
    drain_spans(c);
    bool repaired = false;
    std::vector<IssuedBatch> todo;
    while (!c->issued.empty() || !todo.empty()) {
        if (!c->issued.empty()) {  // the stream is idle here
            int flags[FLAG_RING];
            PT_HIP(c, hipMemcpy(flags, c->d_batch_flags.ptr, sizeof(flags), hipMemcpyDeviceToHost));
            for (const IssuedBatch& r : c->issued)
                if (flags[r.flag]) todo.push_back(r);
            c->issued.clear();
        }
        while (!todo.empty() && c->issued.size() + 2 <= FLAG_RING) {
            const IssuedBatch r = todo.back();
            todo.pop_back();
            if ((size_t)r.n_pixels * r.chunk <= 64)
                return fail(c, PT_ERR_QUEUE_OVERFLOW, "ray queue overflow on a batch of at most 64 samples; raise PT_OPT_QUEUE_FACTOR");
            c->repaired_batches++;
            repaired = true;
            c->samples_started -= (uint64_t)r.n_pixels * r.chunk;  // counted again by the pieces
            int rc;
            if (r.chunk > 1) {
                const int h = r.chunk / 2;
                rc = run_batch(c, r.cam, r.pixel0, r.n_pixels, r.sample0, h);
                if (rc == PT_OK) rc = run_batch(c, r.cam, r.pixel0, r.n_pixels, r.sample0 + h, r.chunk - h);
            } else {
                const int h = r.n_pixels / 2;
                rc = run_batch(c, r.cam, r.pixel0, h, r.sample0, 1);
                if (rc == PT_OK) rc = run_batch(c, r.cam, r.pixel0 + h, r.n_pixels - h, r.sample0, 1);
            }
            if (rc != PT_OK) return rc;
        }
        if (!c->issued.empty()) {
            PT_HIP(c, hipStreamSynchronize(c->stream));
            drain_spans(c);
        }
    }
    if (repaired && c->width > 0 && c->total_samples > 0) {
        PT_HIP(c, launch_resolve(cfg_of(c), c->accum_ptr(), c->d_frame.ptr, c->width * c->height, c->total_samples));
        PT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return PT_OK;
}

// The wait of the exported calls.  A member of a group is not settled by itself: what its overflowed batches owe
// has to go through the group's reduce (group_settle).
int group_settle(PtGroup* g);
int group_error(PtGroup* g, std::string& msg);
int wait_for(PtContext* c) {
    if (!c->group) return settle(c);
    const int rc = group_settle(c->group);
    (void)hipSetDevice(c->device);  // the group's wait went round the members' devices
    if (rc != PT_OK) { std::string msg; group_error(c->group, msg); return fail(c, rc, msg); }
    return PT_OK;
}

// The context's stream may be the null stream, which belongs to whichever device is current on the calling thread:
// every exported entry that waits for or launches on the stream makes the context's device current first (a process
// with one context per GPU would otherwise wait on, or launch to, another context's GPU).
int use_device(PtContext* c) {
    if (c->device != PT_DEVICE_NONE) PT_HIP(c, hipSetDevice(c->device));
    return PT_OK;
}

// Batches in flight, unless the caller chose: three -- except for frames of more than 4 Mi samples of a scene whose
// tree is walked in LDS and whose shading is the plain kind (diffuse surfaces and lights only: the Cornell box), where
// the third brings a frame 0-2 % (its kd-tree -0.6 %) and costs a lane's queues.  Third lane against two at
// 1024 x 1024 x 10: cornellbox-glass +6 %, cornellbox-metal +4 % (kd-tree +8 %), cornellbox-alpha and
// cornellball-glass +-0.5 %; at the reference's 512 x 384 x 10: cornellbox +15 %; teapot +14 %, a 210 k-triangle
// courtyard +4 %.  (A context whose frames go into a collective keeps two, see ensure_lanes' callers.)
int choose_lanes(PtContext* c, size_t frame_samples) {
    if (!c->lanes_auto || c->ext_accum || c->group || c->lanes_for_frame == frame_samples) return PT_OK;
    c->lanes_for_frame = frame_samples;
    const SceneGpu& s = c->scene;
    const bool plain = s.shading_kind == 2;  // SHADE_PLAIN
    const int lanes = (s.lds_nodes >= s.n_nodes && plain && !s.fixed_radiance && frame_samples > ((size_t)4 << 20)) ? 2 : 3;
    if (lanes != c->n_lanes) {
        if (c->device != PT_DEVICE_NONE) PT_HIP(c, hipStreamSynchronize(c->stream));  // every lane is idle after this
        c->n_lanes = lanes;
        c->lanes_checked = false;
    }
    return PT_OK;
}

int render_range(PtContext* c, int spp_frame, int first, int count) {
    if (!c->built) return fail(c, PT_ERR_NOT_READY, "pt_build has not been called");
    if (c->width <= 0 || c->height <= 0) return fail(c, PT_ERR_NOT_READY, "pt_set_resolution has not been called");
    if (spp_frame <= 0 || first < 0 || count < 0 || first + count > spp_frame) return fail(c, PT_ERR_INVALID_ARG, "bad sample range");
    PT_HIP(c, hipSetDevice(c->device));
    c->frame_samples = (size_t)c->width * c->height * (size_t)count;
    {
        const int rc = choose_lanes(c, c->frame_samples);
        if (rc != PT_OK) return rc;
    }
    const int base = c->total_samples;  // global index of this frame's first sample
    c->total_samples += spp_frame;      // src/cs488.h:1882
    const size_t n_pixels = (size_t)c->width * c->height;
    if (count > 0) {
        CameraGpu cam;
        camera_of(c, cam);
        // batches: whole image x sample chunk when a chunk of at least one sample fits, else pixel ranges x 1
        size_t chunk = c->max_batch / n_pixels;
        size_t pix_per_batch = n_pixels;
        if (chunk == 0) { chunk = 1; pix_per_batch = c->max_batch; }
        if (chunk > (size_t)count) chunk = (size_t)count;
        int rc = PT_OK;
        c->batch_hint = pix_per_batch * chunk;  // every lane's queues are sized for the frame's largest batch
        for (size_t p0 = 0; p0 < n_pixels; p0 += pix_per_batch) {
            const size_t np = p0 + pix_per_batch <= n_pixels ? pix_per_batch : n_pixels - p0;
            for (int s = 0; s < count; s += (int)chunk) {
                const int ch = s + (int)chunk <= count ? (int)chunk : count - s;
                if (c->issued.size() + 1 > FLAG_RING) {  // every flag of the ring is waiting to be read
                    rc = settle(c);
                    if (rc != PT_OK) return rc;
                }
                rc = run_batch(c, cam, (int)p0, (int)np, base + first + s, ch);
                if (rc != PT_OK) return rc;
            }
        }
    }
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_create(int device, PtContext** out) {
    if (!out) return PT_ERR_INVALID_ARG;
    *out = nullptr;
    if (device == PT_DEVICE_NONE) {  // scene and tree construction only; nothing here can render
        PtContext* h = new (std::nothrow) PtContext();
        if (!h) return PT_ERR_NO_MEMORY;
        h->device = PT_DEVICE_NONE;
        *out = h;
        return PT_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PT_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return PT_ERR_HIP;
    PtContext* c = new (std::nothrow) PtContext();
    if (!c) return PT_ERR_NO_MEMORY;
    c->device = device;
    // The context's stream is the device's null stream until pt_set_stream names another: a process has four
    // hardware queues per GPU and the null stream holds one of them anyway, so a stream of the context's own would
    // leave only two for the lanes (measured at 512 x 384: 0.72 ms per frame with an own stream and three lanes
    // on the four queues that five streams share, 0.54 with the null stream and three lanes).
    c->stream = nullptr;
    c->own_stream = false;
    hipDeviceProp_t prop;
    // eight 256-lane workgroups per CU are resident at once with the kernels' register and LDS use
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) { c->cus = prop.multiProcessorCount; c->blocks = prop.multiProcessorCount * 8; }
    if (c->blocks > TRACE_POOL_CURSORS * 64) c->blocks = TRACE_POOL_CURSORS * 64;  // the most regions one trace launch serves
    if (c->d_totals.ensure(DC_COUNT) != hipSuccess || hipMemset(c->d_totals.ptr, 0, DC_COUNT * sizeof(unsigned long long)) != hipSuccess ||
        c->d_batch_flags.ensure(FLAG_RING) != hipSuccess || hipMemset(c->d_batch_flags.ptr, 0, FLAG_RING * sizeof(int)) != hipSuccess ||
        c->d_cursors.ensure(TRACE_POOL_INTS) != hipSuccess ||
        c->d_level_rays.ensure(PtContext::LEVEL_WORDS) != hipSuccess ||
        hipHostMalloc((void**)&c->h_level_rays, PtContext::LEVEL_WORDS * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->d_level_rays.ptr, 0xff, PtContext::LEVEL_WORDS * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreateWithFlags(&c->level_copied, hipEventDisableTiming) != hipSuccess ||
        c->d_block_stats.ensure((size_t)c->blocks * DC_BLOCK_COUNT) != hipSuccess ||
        hipMemset(c->d_block_stats.ptr, 0, (size_t)c->blocks * DC_BLOCK_COUNT * sizeof(unsigned long long)) != hipSuccess) {
        pt_destroy(c);
        return PT_ERR_HIP;
    }
    (void)hipStreamSynchronize(nullptr);  // the clears above, before any stream of the context is used
    if (trace_static_scratch(device, &c->static_scratch) != hipSuccess) {  // the device is current here
        pt_destroy(c);
        return PT_ERR_HIP;
    }
    *out = c;
    return PT_OK;
}

void pt_destroy(PtContext* c) {
    if (!c) return;
    if (c->group) return;  // destroy the group first: it holds streams and buffers of this context
    if (c->device == PT_DEVICE_NONE) { delete c; return; }
    (void)hipSetDevice(c->device);
    if (c->device != PT_DEVICE_NONE) (void)hipStreamSynchronize(c->stream);
    drain_spans(c);
    for (hipEvent_t e : c->frame_ends) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->d_nodes.release(); c->d_wide.release(); c->d_isect.release(); c->d_tri_pos.release(); c->d_tri_nrm.release(); c->d_mats.release(); c->d_tri_uv.release();
    c->d_refs.release(); c->d_texels.release(); c->d_env.release(); c->d_accum.release(); c->d_frame.release();
    for (Lane& ln : c->lanes) {
        if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); }
        for (int k = 0; k < 2; k++) {
            if (ln.done[k]) (void)hipEventDestroy(ln.done[k]);
            if (ln.consumed[k]) (void)hipEventDestroy(ln.consumed[k]);
        }
        ln.release();
    }
    c->d_totals.release(); c->d_block_stats.release(); c->d_cursors.release(); c->d_batch_flags.release(); c->d_level_rays.release();
    if (c->h_level_rays) (void)hipHostFree(c->h_level_rays);
    if (c->level_copied) (void)hipEventDestroy(c->level_copied);
    c->tb_rays.release(); c->tb_t.release(); c->tb_bary.release(); c->tb_tri.release(); c->tb_o.release(); c->tb_d.release(); c->tb_h.release();
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* pt_last_error(PtContext* c) { return c ? c->error.c_str() : "null context"; }

int pt_set_stream(PtContext* c, void* hip_stream) {
    if (c && c->group) return fail(c, PT_ERR_NOT_READY, "the context is a member of a group: use the pt_group_* calls (or destroy the group first)");
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    drain_spans(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    c->lanes_checked = false;  // the lanes have to stand clear of this stream's hardware queue too
    return PT_OK;
}

int pt_set_option(PtContext* c, int option, uint64_t value) {
    PT_TRY
    if (!c) return PT_ERR_INVALID_ARG;
    {
        const int rc = use_device(c);
        if (rc != PT_OK) return rc;
    }
    if (!c->issued.empty()) {  // batches in flight are completed under the settings they started with
        int rc = wait_for(c);
        if (rc != PT_OK) return rc;
    }
    switch (option) {
        case PT_OPT_SEED: c->seed = value; break;
        case PT_OPT_MAX_LEVEL:
            if (value > 62) return fail(c, PT_ERR_INVALID_ARG, "max level above 62");
            c->max_level = (int)value; break;
        case PT_OPT_COUNT_TESTS: c->count_tests = value != 0; break;
        case PT_OPT_PROFILE:
            if (value > 2) return fail(c, PT_ERR_INVALID_ARG, "profile level above 2");
            c->profile = (int)value; break;
        case PT_OPT_MAX_BATCH:
            if (value == 0 || value > ((uint64_t)1 << 28)) return fail(c, PT_ERR_INVALID_ARG, "batch size out of range");
            c->max_batch = (size_t)value; c->batch_hint = 0; for (Lane& ln : c->lanes) ln.batch_alloc = 0; break;
        case PT_OPT_QUEUE_FACTOR:
            if (value < 2 || value > 64) return fail(c, PT_ERR_INVALID_ARG, "queue factor out of range");
            c->queue_factor = (int)value; c->batch_hint = 0; for (Lane& ln : c->lanes) ln.batch_alloc = 0; break;
        case PT_OPT_LDS_NODES:
            if (value > 2048 && value != 65535) return fail(c, PT_ERR_INVALID_ARG, "at most 2048 nodes fit the LDS stage");
            {
                const int before = c->lds_nodes_opt;
                c->lds_nodes_opt = value == 65535 ? -1 : (int)value;
                if (c->built && c->device != PT_DEVICE_NONE) {
                    choose_lds_nodes(c);
                    // a stage that leaves no room for a single workgroup (the staged nodes are 80 bytes apart, beside
                    // the traversal stacks) would make every trace launch fail: refused here instead
                    if (!trace_kernel_fits(c->scene, c->accel == PT_ACCEL_KD, c->traversal)) {
                        c->lds_nodes_opt = before;
                        choose_lds_nodes(c);
                        return fail(c, PT_ERR_INVALID_ARG, "that many nodes do not fit the LDS a workgroup of the trace kernel may use");
                    }
                }
            }
            break;
        case PT_OPT_TRAVERSAL:
            // a wave has 64 lanes: a refill threshold above that could never be met and the walk would not end
            if (value > 0xffff || ((value >> 8) & 127) > 64) return fail(c, PT_ERR_INVALID_ARG, "traversal bits 8..14 (refill threshold) above 64, or unknown bits");
            c->traversal = (int)value; c->resident_per_cu = 0;
            if ((value & 0x8000) != 0 && c->built && c->device != PT_DEVICE_NONE) {
                const int rc = ensure_wide_nodes(c);
                if (rc != PT_OK) return rc;
            }
            break;
        case PT_OPT_OBJ_EXTENDED: c->obj_extended = value != 0; break;
        case PT_OPT_BLOCKS_PER_CU:
            if (value > 8) return fail(c, PT_ERR_INVALID_ARG, "at most 8 workgroups per CU");
            c->blocks_per_cu = (int)value; break;
        case PT_OPT_BUILD_THREADS:
            if (value > 256) return fail(c, PT_ERR_INVALID_ARG, "more than 256 builder threads");
            c->build_threads = (int)value; break;
        case PT_OPT_STREAMS:
            if (value < 1 || value > MAX_LANES) return fail(c, PT_ERR_INVALID_ARG, "between 1 and 4 batches in flight");
            if (c->device != PT_DEVICE_NONE) PT_HIP(c, hipStreamSynchronize(c->stream));  // every lane is idle after this
            c->n_lanes = (int)value; c->lanes_auto = false; c->lanes_checked = false; break;
        case PT_OPT_TAIL_LEVEL:
            if (value > 64 && value != 255) return fail(c, PT_ERR_INVALID_ARG, "tail level above 64");
            c->tail_level = (int)value; break;
        default: return fail(c, PT_ERR_INVALID_ARG, "unknown option");
    }
    return PT_OK;
    PT_CATCH(c)
}

int pt_get_option(PtContext* c, int option, uint64_t* value) {
    if (!c || !value) return PT_ERR_INVALID_ARG;
    switch (option) {
        case PT_OPT_SEED: *value = c->seed; break;
        case PT_OPT_MAX_LEVEL: *value = (uint64_t)c->max_level; break;
        case PT_OPT_COUNT_TESTS: *value = (uint64_t)c->count_tests; break;
        case PT_OPT_PROFILE: *value = (uint64_t)c->profile; break;
        case PT_OPT_MAX_BATCH: *value = (uint64_t)c->max_batch; break;
        case PT_OPT_QUEUE_FACTOR: *value = (uint64_t)c->queue_factor; break;
        case PT_OPT_LDS_NODES: *value = c->built ? (uint64_t)c->scene.lds_nodes : (c->lds_nodes_opt < 0 ? 65535u : (uint64_t)c->lds_nodes_opt); break;
        case PT_OPT_TRAVERSAL: *value = (uint64_t)c->traversal; break;
        case PT_OPT_TAIL_LEVEL: *value = (uint64_t)c->tail_level; break;
        case PT_OPT_STREAMS: *value = (uint64_t)c->n_lanes; break;
        case PT_OPT_BUILD_THREADS: *value = (uint64_t)c->build_threads; break;
        case PT_OPT_OBJ_EXTENDED: *value = (uint64_t)c->obj_extended; break;
        case PT_OPT_BLOCKS_PER_CU: *value = (uint64_t)c->blocks_per_cu; break;
        case PT_OPT_TAIL_LEVEL_USED: *value = (uint64_t)c->tail_level_used; break;
        case PT_OPT_TREE_IN_LDS: *value = c->built && c->device != PT_DEVICE_NONE ? (uint64_t)trace_scene_in_lds(c->scene) : 0; break;
        default: return fail(c, PT_ERR_INVALID_ARG, "unknown option");
    }
    return PT_OK;
}

// ---- scene ----
int pt_load_obj(PtContext* c, const char* path) {
    PT_TRY
    if (!c || !path) return PT_ERR_INVALID_ARG;
    c->built = false;
    std::string err;
    if (!load_obj(path, c->mesh, err, c->obj_extended != 0)) {
        FILE* probe = fopen(path, "r");
        if (!probe) return fail(c, PT_ERR_IO, err);
        fclose(probe);
        return fail(c, PT_ERR_INVALID_OBJ, err);
    }
    return PT_OK;
    PT_CATCH(c)
}

int pt_single_triangle(PtContext* c) {
    PT_TRY
    if (!c) return PT_ERR_INVALID_ARG;
    c->built = false;
    make_single_triangle(c->mesh);
    return PT_OK;
    PT_CATCH(c)
}

int pt_load_env(PtContext* c, const char* path) {
    PT_TRY
    if (!c || !path) return PT_ERR_INVALID_ARG;
    std::string err;
    HostImage img;
    if (!load_hdr(path, img, err)) return fail(c, PT_ERR_IO, err);
    c->env = img;
    c->built = false;
    return PT_OK;
    PT_CATCH(c)
}

int pt_set_mesh(PtContext* c, int n, const float* pos9, const float* nrm9, const float* uv6, const int32_t* matid) {
    PT_TRY
    if (!c || n <= 0 || !pos9 || !nrm9) return fail(c, PT_ERR_INVALID_ARG, "pt_set_mesh: bad arguments");
    c->built = false;
    c->mesh.pos.assign(pos9, pos9 + 9 * (size_t)n);
    c->mesh.nrm.assign(nrm9, nrm9 + 9 * (size_t)n);
    if (uv6) c->mesh.uv.assign(uv6, uv6 + 6 * (size_t)n); else c->mesh.uv.assign(6 * (size_t)n, 0.0f);
    if (matid) c->mesh.mat.assign(matid, matid + n); else c->mesh.mat.assign(n, 0);
    if (c->mesh.materials.empty()) c->mesh.materials.resize(1);
    return PT_OK;
    PT_CATCH(c)
}

int pt_set_materials(PtContext* c, int n, const PtMaterial* mats) {
    PT_TRY
    if (!c || n <= 0 || !mats) return fail(c, PT_ERR_INVALID_ARG, "pt_set_materials: bad arguments");
    c->built = false;
    c->mesh.materials.resize(n);
    for (int i = 0; i < n; i++) from_public(mats[i], c->mesh.materials[i]);
    return PT_OK;
    PT_CATCH(c)
}

int pt_set_texture(PtContext* c, int material, const uint8_t* rgb, int w, int h) {
    PT_TRY
    if (!c || !rgb || w <= 0 || h <= 0 || material < 0 || material >= (int)c->mesh.materials.size())
        return fail(c, PT_ERR_INVALID_ARG, "pt_set_texture: bad arguments");
    HostMaterial& m = c->mesh.materials[material];
    m.tex.assign(rgb, rgb + (size_t)w * h * 3);
    m.tex_w = w; m.tex_h = h;
    m.textured = true;
    c->built = false;
    return PT_OK;
    PT_CATCH(c)
}

int pt_set_env(PtContext* c, const float* rgb, int w, int h) {
    PT_TRY
    if (!c) return PT_ERR_INVALID_ARG;
    c->built = false;
    if (!rgb || w <= 0 || h <= 0) { c->env = HostImage(); return PT_OK; }
    c->env.w = w; c->env.h = h;
    c->env.rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++) memcpy(&c->env.rgb[(size_t)(h - 1 - y) * w * 3], rgb + (size_t)y * w * 3, sizeof(float) * 3 * (size_t)w);
    return PT_OK;
    PT_CATCH(c)
}

int pt_num_triangles(PtContext* c) { return c ? c->mesh.num_tris() : PT_ERR_INVALID_ARG; }
int pt_num_materials(PtContext* c) { return c ? (int)c->mesh.materials.size() : PT_ERR_INVALID_ARG; }

int pt_get_mesh(PtContext* c, float* pos9, float* nrm9, float* uv6, int32_t* matid) {
    if (!c) return PT_ERR_INVALID_ARG;
    const size_t n = (size_t)c->mesh.num_tris();
    if (pos9) memcpy(pos9, c->mesh.pos.data(), sizeof(float) * 9 * n);
    if (nrm9) memcpy(nrm9, c->mesh.nrm.data(), sizeof(float) * 9 * n);
    if (uv6) memcpy(uv6, c->mesh.uv.data(), sizeof(float) * 6 * n);
    if (matid) memcpy(matid, c->mesh.mat.data(), sizeof(int32_t) * n);
    return PT_OK;
}

int pt_get_material(PtContext* c, int index, PtMaterial* out) {
    PT_TRY
    if (!c || !out || index < 0 || index >= (int)c->mesh.materials.size()) return fail(c, PT_ERR_INVALID_ARG, "pt_get_material: bad index");
    to_public(c->mesh.materials[index], out);
    return PT_OK;
    PT_CATCH(c)
}

int pt_get_texture(PtContext* c, int material, uint8_t* rgb) {
    if (!c || !rgb || material < 0 || material >= (int)c->mesh.materials.size()) return fail(c, PT_ERR_INVALID_ARG, "pt_get_texture: bad index");
    const HostMaterial& m = c->mesh.materials[material];
    if (m.tex.empty()) return fail(c, PT_ERR_NOT_READY, "material has no decoded texture");
    memcpy(rgb, m.tex.data(), m.tex.size());
    return PT_OK;
}

int pt_get_env(PtContext* c, float* rgb, int* w, int* h) {
    if (!c) return PT_ERR_INVALID_ARG;
    if (w) *w = c->env.w;
    if (h) *h = c->env.h;
    if (rgb && !c->env.rgb.empty()) memcpy(rgb, c->env.rgb.data(), sizeof(float) * c->env.rgb.size());
    return PT_OK;
}

// ---- accelerator ----
int pt_build(PtContext* c, int accel) {
    PT_TRY
    if (c && c->group) return fail(c, PT_ERR_NOT_READY, "the context is a member of a group: use the pt_group_* calls (or destroy the group first)");
    if (!c) return PT_ERR_INVALID_ARG;
    if (accel != PT_ACCEL_BVH && accel != PT_ACCEL_KD) return fail(c, PT_ERR_INVALID_ARG, "unknown accelerator kind");
    if (c->mesh.num_tris() == 0) return fail(c, PT_ERR_NOT_READY, "no mesh loaded");
    if (!c->issued.empty()) {
        int rc = settle(c);
        if (rc != PT_OK) return rc;
    }
    c->accel = accel;
    if (accel == PT_ACCEL_KD) build_kd(c->mesh, c->tree, c->build_threads); else build_bvh(c->mesh, c->tree, c->build_threads);
    if (c->device == PT_DEVICE_NONE) return PT_OK;  // tree available through pt_get_tree; not renderable
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipStreamSynchronize(c->stream));  // no lane reads the old scene any more
    int rc = flatten_and_upload(c);
    if (rc != PT_OK) return rc;
    c->built = true;
    if ((c->traversal & 0x8000) != 0) return ensure_wide_nodes(c);
    return PT_OK;
    PT_CATCH(c)
}

int pt_get_tree(PtContext* c, int* n_nodes, int* n_refs, int* max_depth, int32_t* left, int32_t* right, int32_t* first,
                int32_t* count, float* box6, int32_t* refs) {
    if (!c) return PT_ERR_INVALID_ARG;
    const HostTree& t = c->tree;
    if (n_nodes) *n_nodes = (int)t.nodes.size();
    if (n_refs) *n_refs = (int)t.refs.size();
    if (max_depth) *max_depth = t.max_depth;
    for (size_t i = 0; i < t.nodes.size(); i++) {
        if (left) left[i] = t.nodes[i].left;
        if (right) right[i] = t.nodes[i].right;
        if (first) first[i] = t.nodes[i].first;
        if (count) count[i] = t.nodes[i].count;
        if (box6) memcpy(box6 + 6 * i, t.nodes[i].box, sizeof(float) * 6);
    }
    if (refs && !t.refs.empty()) memcpy(refs, t.refs.data(), sizeof(int32_t) * t.refs.size());
    return PT_OK;
}

// ---- camera and film ----
int pt_set_camera(PtContext* c, const float eye[3], const float lookat[3], const float up[3], float fov_deg, float film_size) {
    if (!c || !eye || !lookat || !up) return PT_ERR_INVALID_ARG;
    for (int k = 0; k < 3; k++) { c->eye[k] = eye[k]; c->lookat[k] = lookat[k]; c->up[k] = up[k]; }
    c->fov = fov_deg; c->film = film_size;
    return PT_OK;
}

int pt_set_resolution(PtContext* c, int width, int height) {
    if (c && c->group) return fail(c, PT_ERR_NOT_READY, "the context is a member of a group: use the pt_group_* calls (or destroy the group first)");
    if (!c || width <= 0 || height <= 0) return fail(c, PT_ERR_INVALID_ARG, "bad resolution");
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    PT_HIP(c, hipSetDevice(c->device));
    c->width = width; c->height = height;
    const size_t n = (size_t)width * height * 3;
    PT_HIP(c, c->d_accum.ensure(n));
    PT_HIP(c, c->d_frame.ensure(n));
    return pt_reset(c);
}

int pt_reset(PtContext* c) {
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    if (c->width <= 0) return fail(c, PT_ERR_NOT_READY, "pt_set_resolution has not been called");
    const size_t n = (size_t)c->width * c->height * 3;
    PT_HIP(c, hipSetDevice(c->device));
    c->issued.clear();  // whatever overflowed belongs to the image that is being thrown away
    PT_HIP(c, launch_clear(cfg_of(c), c->accum_ptr(), n));
    PT_HIP(c, launch_clear(cfg_of(c), c->d_frame.ptr, n));
    c->total_samples = 0;
    return PT_OK;
}

// ---- hot path ----
int pt_render(PtContext* c, int spp) {
    PT_TRY
    if (c && c->group) return fail(c, PT_ERR_NOT_READY, "the context is a member of a group: use the pt_group_* calls (or destroy the group first)");
    if (!c) return PT_ERR_INVALID_ARG;
    int rc = render_range(c, spp, 0, spp);
    if (rc != PT_OK) return rc;
    rc = pt_resolve(c);
    if (rc != PT_OK) return rc;
    return mark_frame_end(c);
    PT_CATCH(c)
}

int pt_render_split(PtContext* c, int spp_frame, int first, int count) {
    PT_TRY
    if (c && c->group) return fail(c, PT_ERR_NOT_READY, "the context is a member of a group: use the pt_group_* calls (or destroy the group first)");
    if (!c) return PT_ERR_INVALID_ARG;
    const int rc = render_range(c, spp_frame, first, count);
    if (rc != PT_OK) return rc;
    return mark_frame_end(c);
    PT_CATCH(c)
}

int pt_resolve(PtContext* c) {
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    if (c->width <= 0) return fail(c, PT_ERR_NOT_READY, "pt_set_resolution has not been called");
    PT_HIP(c, hipSetDevice(c->device));
    ScopedTimer tm(c, PT_TIMER_ACCUM);
    PT_HIP(c, launch_resolve(cfg_of(c), c->accum_ptr(), c->d_frame.ptr, c->width * c->height, c->total_samples));
    return PT_OK;
}

int pt_total_samples(PtContext* c) { return c ? c->total_samples : PT_ERR_INVALID_ARG; }

int pt_synchronize(PtContext* c) {
    PT_TRY
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    return wait_for(c);
    PT_CATCH(c)
}

// Returns when at most `keep` of the frames queued so far are unfinished.  Unlike pt_synchronize it leaves the
// rest of the queue running and does not look at overflowed batches.
int pt_wait_frames(PtContext* c, int keep) {
    PT_TRY
    if (!c || keep < 0) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    PT_HIP(c, hipSetDevice(c->device));
    while ((int)c->frame_ends.size() > keep) {
        PT_HIP(c, hipEventSynchronize(c->frame_ends.front()));
        c->event_pool.push_back(c->frame_ends.front());
        c->frame_ends.pop_front();
    }
    return PT_OK;
    PT_CATCH(c)
}

// Frames queued by pt_render / pt_render_split whose end the GPU has not reached yet; never waits.
int pt_frames_in_flight(PtContext* c) {
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return 0;
    while (!c->frame_ends.empty() && hipEventQuery(c->frame_ends.front()) == hipSuccess) {
        c->event_pool.push_back(c->frame_ends.front());
        c->frame_ends.pop_front();
    }
    return (int)c->frame_ends.size();
}

int pt_download(PtContext* c, int which, float* dst) {
    PT_TRY
    if (!c || !dst) return PT_ERR_INVALID_ARG;
    if (c->width <= 0) return fail(c, PT_ERR_NOT_READY, "pt_set_resolution has not been called");
    int rc = pt_synchronize(c);
    if (rc != PT_OK) return rc;
    const size_t n = (size_t)c->width * c->height * 3;
    // (a member of a group renders every frame into the group's per-frame buffer; its accumulation is the internal one)
    const float* accum = c->group ? c->d_accum.ptr : c->accum_ptr();
    PT_HIP(c, hipMemcpy(dst, which == 0 ? accum : c->d_frame.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
    PT_CATCH(c)
}

void* pt_accum_device_ptr(PtContext* c) { return c ? (void*)c->accum_ptr() : nullptr; }

// A process has four hardware queues per GPU (GPU_MAX_HW_QUEUES): the context's stream and three lanes fill
// them.  Once the context's frames go into a collective, the collective library's own stream needs one, so a
// context whose number of lanes the caller has not chosen keeps two from then on.
static int leave_a_queue_for_the_collective(PtContext* c) {
    if (!c->lanes_auto || c->n_lanes <= 2) return PT_OK;
    if (c->device != PT_DEVICE_NONE) {
        PT_HIP(c, hipSetDevice(c->device));
        PT_HIP(c, hipStreamSynchronize(c->stream));  // every lane is idle after this
    }
    c->n_lanes = 2;
    c->lanes_checked = false;
    return PT_OK;
}

int pt_use_external_accum(PtContext* c, void* device_ptr) {
    if (c && c->group) return fail(c, PT_ERR_NOT_READY, "the context is a member of a group: use the pt_group_* calls (or destroy the group first)");
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    if (device_ptr) {
        const int rc = leave_a_queue_for_the_collective(c);  // waits once, the first time
        if (rc != PT_OK) return rc;
    }
    // no wait: everything that touches the buffers is ordered on the context's stream
    c->ext_accum = (float*)device_ptr;
    return PT_OK;
}

static int save_image(PtContext* c, int which, const char* path, bool png) {
    PT_TRY
    if (!c || !path) return PT_ERR_INVALID_ARG;
    std::vector<float> img((size_t)c->width * c->height * 3);
    int rc = pt_download(c, which, img.data());
    if (rc != PT_OK) return rc;
    std::string err;
    if (png) {
        std::vector<uint8_t> q;
        quantise_image(img.data(), c->width, c->height, q);
        if (!save_png_rgb8(path, q.data(), c->width, c->height, err)) return fail(c, PT_ERR_IO, err);
    } else if (!save_pfm(path, img.data(), c->width, c->height, err)) {
        return fail(c, PT_ERR_IO, err);
    }
    return PT_OK;
    PT_CATCH(c)
}

int pt_save_png(PtContext* c, int which, const char* path) { return save_image(c, which, path, true); }
int pt_save_pfm(PtContext* c, int which, const char* path) { return save_image(c, which, path, false); }

// ---- recording ----
int pt_record_begin(PtContext* c, const char* path, int delay_cs) {
    PT_TRY
    if (!c || !path || delay_cs < 0 || delay_cs > 65535) return PT_ERR_INVALID_ARG;
    if (c->width <= 0) return fail(c, PT_ERR_NOT_READY, "pt_set_resolution has not been called");
    std::string err;
    if (!c->recording.begin(path, c->width, c->height, delay_cs, err)) return fail(c, PT_ERR_IO, err);
    return PT_OK;
    PT_CATCH(c)
}

int pt_record_frame(PtContext* c) {
    PT_TRY
    if (!c) return PT_ERR_INVALID_ARG;
    if (!c->recording.open()) return fail(c, PT_ERR_NOT_READY, "pt_record_begin has not been called");
    if (c->recording.width() != c->width || c->recording.height() != c->height)
        return fail(c, PT_ERR_INVALID_ARG, "the resolution changed since pt_record_begin");
    std::vector<float> img((size_t)c->width * c->height * 3);
    int rc = pt_download(c, 1, img.data());
    if (rc != PT_OK) return rc;
    std::vector<uint8_t> q;
    quantise_image(img.data(), c->width, c->height, q);
    std::string err;
    if (!c->recording.frame(q.data(), err)) return fail(c, PT_ERR_IO, err);
    return PT_OK;
    PT_CATCH(c)
}

int pt_record_end(PtContext* c) {
    if (!c) return PT_ERR_INVALID_ARG;
    c->recording.end();
    return PT_OK;
}

int pt_encode_gif(const char* path, const uint8_t* rgb8, int n_frames, int width, int height, int delay_cs) {
    PT_TRY
    if (!path || !rgb8 || n_frames < 0 || delay_cs < 0 || delay_cs > 65535) return PT_ERR_INVALID_ARG;
    GifWriter g;
    std::string err;
    if (!g.begin(path, width, height, delay_cs, err)) return PT_ERR_IO;
    for (int f = 0; f < n_frames; f++)
        if (!g.frame(rgb8 + (size_t)f * width * height * 3, err)) return PT_ERR_IO;
    g.end();
    return PT_OK;
    PT_CATCH((PtContext*)nullptr)
}

// ---- parity entry ----
int pt_trace_batch(PtContext* c, int n, const float* rays6, int32_t* tri, float* t, float* bary2) {
    PT_TRY
    if (!c || n < 0 || (n > 0 && (!rays6 || !tri || !t))) return fail(c, PT_ERR_INVALID_ARG, "pt_trace_batch: bad arguments");
    if (!c->built) return fail(c, PT_ERR_NOT_READY, "pt_build has not been called");
    if (n == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    DeviceBuffer<float>&d_rays = c->tb_rays, &d_t = c->tb_t, &d_bary = c->tb_bary;
    DeviceBuffer<int>& d_tri = c->tb_tri;
    DeviceBuffer<float4>&d_o = c->tb_o, &d_d = c->tb_d, &d_h = c->tb_h;
    auto body = [&]() -> int {
    PT_HIP(c, d_rays.upload(rays6, 6 * (size_t)n, c->stream));
    PT_HIP(c, d_o.ensure(n)); PT_HIP(c, d_d.ensure(n)); PT_HIP(c, d_h.ensure(n));
    PT_HIP(c, d_tri.ensure(n)); PT_HIP(c, d_t.ensure(n)); PT_HIP(c, d_bary.ensure(2 * (size_t)n));
    const LaunchCfg cfg = cfg_of(c);
    PT_HIP(c, launch_pack_rays(cfg, d_rays.ptr, n, d_o.ptr, d_d.ptr));
    int nb = (n + 255) / 256;
    nb = nb > c->blocks ? c->blocks : nb;
    const int cap_block = (n + nb - 1) / nb;
    PT_HIP(c, hipMemsetAsync(c->d_cursors.ptr, 0, TRACE_POOL_INTS * sizeof(int), c->stream));
    {
        ScopedTimer tm(c, PT_TIMER_TRACE);
        PT_HIP(c, launch_trace(cfg, c->scene, c->accel == PT_ACCEL_KD, c->count_tests, c->traversal, d_o.ptr, d_d.ptr, d_h.ptr,
                               nullptr, n, nb, cap_block, c->d_cursors.ptr, c->d_block_stats.ptr));
    }
    c->batch_rays += (uint64_t)n;
    PT_HIP(c, launch_unpack_hits(cfg, d_h.ptr, n, d_tri.ptr, d_t.ptr, d_bary.ptr));
    PT_HIP(c, hipMemcpyAsync(tri, d_tri.ptr, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipMemcpyAsync(t, d_t.ptr, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (bary2) PT_HIP(c, hipMemcpyAsync(bary2, d_bary.ptr, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    return PT_OK;
    };
    const int rc = body();
    // nothing of the call may still be writing into the caller's arrays when it returns, whatever the outcome
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PT_OK) return fail(c, PT_ERR_HIP, "pt_trace_batch: stream synchronisation failed");
    return rc;
    PT_CATCH(c)
}

// The same closest-hit query on device-resident arrays, without any allocation: the packed rays and
// the hit records are staged in this GPU's copy of a file-scope device array.
int pt_trace_batch_static(PtContext* c, int n, const float* d_rays6, int32_t* d_tri, float* d_t, float* d_bary2) {
    if (!c || n < 0 || (n > 0 && (!d_rays6 || !d_tri || !d_t))) return fail_noexcept(c, PT_ERR_INVALID_ARG, "pt_trace_batch_static: bad arguments");
    if (c->device == PT_DEVICE_NONE || !c->built) return fail_noexcept(c, PT_ERR_NOT_READY, "pt_build has not been called");
    if (n > TRACE_STATIC_MAX_RAYS) return fail_noexcept(c, PT_ERR_INVALID_ARG, "pt_trace_batch_static: more than 131072 rays in one call (the static staging array holds no more)");
    if (n == 0) return PT_OK;
    if (!c->static_scratch) return fail_noexcept(c, PT_ERR_HIP, "pt_trace_batch_static: the staging array's device address is unknown");
    if (hipSetDevice(c->device) != hipSuccess) return fail_noexcept(c, PT_ERR_HIP, "pt_trace_batch_static: hipSetDevice failed");
    float4* const ray_o = c->static_scratch;
    float4* const ray_d = ray_o + TRACE_STATIC_MAX_RAYS;
    float4* const hits = ray_d + TRACE_STATIC_MAX_RAYS;
    const LaunchCfg cfg = cfg_of(c);
    // every entry a kernel of this call reads is written by the call first (pack -> trace -> unpack), so what
    // the previous call left in the array needs no clearing
    if (launch_pack_rays(cfg, d_rays6, n, ray_o, ray_d) != hipSuccess) return fail_noexcept(c, PT_ERR_HIP, "pt_trace_batch_static: launch failed");
    int nb = (n + 255) / 256;
    nb = nb > c->blocks ? c->blocks : nb;
    const int cap_block = (n + nb - 1) / nb;
    if (hipMemsetAsync(c->d_cursors.ptr, 0, TRACE_POOL_INTS * sizeof(int), c->stream) != hipSuccess ||
        launch_trace(cfg, c->scene, c->accel == PT_ACCEL_KD, c->count_tests, c->traversal, ray_o, ray_d, hits, nullptr, n, nb, cap_block,
                     c->d_cursors.ptr, c->d_block_stats.ptr) != hipSuccess)
        return fail_noexcept(c, PT_ERR_HIP, "pt_trace_batch_static: launch failed");
    c->batch_rays += (uint64_t)n;
    if (launch_unpack_hits(cfg, hits, n, d_tri, d_t, d_bary2) != hipSuccess) return fail_noexcept(c, PT_ERR_HIP, "pt_trace_batch_static: launch failed");
    return PT_OK;
}

int pt_camera_rays(PtContext* c, float* rays6) {
    if (!c || !rays6) return PT_ERR_INVALID_ARG;
    if (c->width <= 0) return fail(c, PT_ERR_NOT_READY, "pt_set_resolution has not been called");
    CameraGpu cam;
    camera_of(c, cam);
    const vec3 eye = v3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const vec3 u = v3(cam.u[0], cam.u[1], cam.u[2]), v = v3(cam.v[0], cam.v[1], cam.v[2]), w = v3(cam.w[0], cam.w[1], cam.w[2]);
    size_t k = 0;
    for (int y = 0; y < cam.height; y++) {
        for (int x = 0; x < cam.width; x++) {
            const float U = (x + 0.5f) / float(cam.width) - 0.5f;   // src/cs488.h:1788-1789
            const float V = (y + 0.5f) / float(cam.height) - 0.5f;
            const vec3 pix = eye + float(cam.aspect_film * U) * u + float(cam.film * V) * v - cam.dist * w;
            const vec3 d = normalize(pix - eye);
            rays6[k++] = eye.x; rays6[k++] = eye.y; rays6[k++] = eye.z;
            rays6[k++] = d.x; rays6[k++] = d.y; rays6[k++] = d.z;
        }
    }
    return PT_OK;
}

// The camera rays the ray-generation kernel draws for one global sample index, one per pixel, in pixel order.
int pt_camera_samples(PtContext* c, int sample, float* rays6) {
    PT_TRY
    if (!c || !rays6 || sample < 0) return fail(c, PT_ERR_INVALID_ARG, "pt_camera_samples: bad arguments");
    if (c->device == PT_DEVICE_NONE || !c->built) return fail(c, PT_ERR_NOT_READY, "pt_build has not been called");
    if (c->width <= 0) return fail(c, PT_ERR_NOT_READY, "pt_set_resolution has not been called");
    const size_t n_pixels = (size_t)c->width * c->height;
    if (n_pixels > c->max_batch) return fail(c, PT_ERR_INVALID_ARG, "pt_camera_samples: more pixels than PT_OPT_MAX_BATCH");
    int rc = wait_for(c);  // the lanes are idle after this
    if (rc != PT_OK) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    Lane& ln = c->lanes[0];
    rc = ensure_lane(c, ln, n_pixels);
    if (rc != PT_OK) return rc;
    CameraGpu cam;
    camera_of(c, cam);
    BatchGpu b;
    b.pixel0 = 0; b.n_pixels = (int)n_pixels; b.chunk = 1; b.sample0 = sample; b.seed = c->seed; b.n_slots = (int)n_pixels;
    b.fixed_radiance = c->scene.fixed_radiance;
    shape_batch(c, ln, c->n_lanes > 1, b);
    PT_HIP(c, ln.d_keys.ensure(1024));
    PT_HIP(c, ln.d_eye.ensure(1));
    PT_HIP(c, ln.d_qcount.ensure(((size_t)c->max_level + 3) * (size_t)c->blocks));
    b.keys = ln.d_keys.ptr;
    b.overflow = c->d_batch_flags.ptr + (int)(c->batch_seq++ % FLAG_RING);
    QueueSet out;
    out.ray_o = ln.d_eye.ptr; out.ray_d = ln.q_ray_d[0].ptr; out.beta = nullptr;
    const LaunchCfg cfg = cfg_of(c);
    PT_HIP(c, c->tb_rays.ensure(6 * n_pixels));
    PT_HIP(c, launch_raygen(cfg, cam, b, out, ln.d_sample_rad.ptr, ln.d_qcount.ptr));
    PT_HIP(c, launch_gather_camera_rays(cfg, cam, b, out.ray_d, ln.d_qcount.ptr, c->tb_rays.ptr));
    PT_HIP(c, hipMemcpyAsync(rays6, c->tb_rays.ptr, sizeof(float) * 6 * n_pixels, hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
    PT_CATCH(c)
}

// ---- measurement ----
int pt_get_counters(PtContext* c, uint64_t out[PT_CNT_COUNT]) {
    PT_TRY
    if (!c || !out) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    int rc = pt_synchronize(c);
    if (rc != PT_OK) return rc;
    unsigned long long d[DC_COUNT], st[DC_BLOCK_COUNT];
    PT_HIP(c, hipMemcpy(d, c->d_totals.ptr, sizeof(d), hipMemcpyDeviceToHost));
    rc = read_block_stats(c, st);
    if (rc != PT_OK) return rc;
    memset(out, 0, sizeof(uint64_t) * PT_CNT_COUNT);
    out[PT_CNT_RAYS] = d[DC_RAYS] + c->batch_rays;
    out[PT_CNT_BOX_TESTS] = st[DB_BOX];
    out[PT_CNT_TRI_TESTS] = st[DB_TRI];
    out[PT_CNT_SAMPLES] = c->samples_started;
    out[PT_CNT_NAN_SAMPLES] = d[DC_NAN];
    out[PT_CNT_HITS] = st[DB_HITS];
    out[PT_CNT_STREAM_SWAPS] = (uint64_t)c->lane_stream_swaps;
    out[PT_CNT_WAVE_STEPS] = st[DB_WAVE_STEPS];
    out[PT_CNT_WAVE_ROUNDS] = st[DB_WAVE_ROUNDS];
    out[PT_CNT_MAX_QUEUE] = d[DC_MAXQ];
    out[PT_CNT_REPAIRED] = c->repaired_batches;
    out[PT_CNT_TAIL_RAYS] = d[DC_TAIL_RAYS];
    return PT_OK;
    PT_CATCH(c)
}

int pt_reset_counters(PtContext* c) {
    PT_TRY
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    int rc0 = wait_for(c);
    if (rc0 != PT_OK) return rc0;
    PT_HIP(c, hipMemset(c->d_totals.ptr, 0, DC_COUNT * sizeof(unsigned long long)));
    int rc = clear_block_stats(c);
    if (rc != PT_OK) return rc;
    c->samples_started = 0;
    c->repaired_batches = 0;
    c->batch_rays = 0;
    for (int i = 0; i < PT_TIMER_COUNT; i++) { c->timer_ms[i] = 0; c->timer_launches[i] = 0; }
    return PT_OK;
    PT_CATCH(c)
}

int pt_get_timers(PtContext* c, double ms[PT_TIMER_COUNT], uint64_t launches[PT_TIMER_COUNT]) {
    if (!c) return PT_ERR_INVALID_ARG;
    if (c->device == PT_DEVICE_NONE) return fail(c, PT_ERR_NOT_READY, "context was created without a GPU");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    drain_spans(c);
    for (int i = 0; i < PT_TIMER_COUNT; i++) {
        if (ms) ms[i] = c->timer_ms[i];
        if (launches) launches[i] = c->timer_launches[i];
    }
    return PT_OK;
}


// ---- several GPUs in one process: sample split and one RCCL reduce per frame ----------------------
}  // extern "C"

// A host thread that queues one member's launches.  The reference gives every strip of a frame a thread of its own
// (src/cs488.h:1885-1890); here every GPU of a group gets one, because a frame is a few dozen launches per member and
// a short frame (0.4 ms at the reference's own resolution) cannot be queued for eight GPUs by one thread in its time.
// The thread lives as long as the group, keeps its member's device current, and runs one job at a time.
struct GroupWorker {
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, quit = false;
    int rc = PT_OK;
    void run(int device) {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(m);
        while (true) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> j = std::move(job);
            lk.unlock();
            int r;
            try { r = j(); } catch (...) { r = PT_ERR_NO_MEMORY; }
            lk.lock();
            rc = r;
            has_job = false;
            cv.notify_all();
        }
    }
    void post(std::function<int()> j) {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(j);
        has_job = true;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !has_job; });
        return rc;
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); }
        if (thread.joinable()) thread.join();
    }
};

struct PtGroup {
    std::vector<PtContext*> members;      // members[0] is the root
    std::vector<ncclComm_t> comms;        // one communicator handle per member (ncclCommInitAll)
    // Per member, on its GPU: two per-frame buffers used in turn.  The member's batches accumulate into the current one
    // on the member's stream; the reduce of a filled buffer runs on the member's exchange stream, so the accumulate
    // kernels of the next frame (other buffer) do not queue behind it.
    std::vector<float*> delta[2];
    std::vector<hipStream_t> xstream;     // per member: the stream its share of the reduce runs on
    std::vector<hipEvent_t> filled[2];    // per member and buffer: the buffer's last accumulate has been queued (member's stream)
    std::vector<hipEvent_t> reduced[2];   // per member and buffer: the reduce that read it has been queued (exchange stream)
    std::vector<GroupWorker*> workers;    // one host thread per member (none for a group of one)
    int cur = 0;                          // buffer the current frames go into
    int cadence = 1;                      // frames added into one buffer before it is reduced (pt_group_set_cadence)
    int in_buffer = 0;                    // frames in the current buffer that have not been reduced yet
    size_t n_floats = 0;
    uint64_t frames = 0, reduces = 0;
    std::string error;
};

namespace {

int gfail(PtGroup* g, int code, const std::string& msg) {
    if (g) g->error = msg;
    return code;
}

#define PT_GHIP(g, call)                                                                          \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) return gfail(g, PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define PT_GNCCL(g, call)                                                                         \
    do {                                                                                          \
        ncclResult_t r__ = (call);                                                                \
        if (r__ != ncclSuccess) return gfail(g, PT_ERR_RCCL, std::string(#call) + ": " + ncclGetErrorString(r__)); \
    } while (0)

// RCCL announces itself (version, host, library path) on standard output when a communicator is made.  A host that
// speaks a line protocol on stdout -- the reference's per-frame sample count and seconds -- cannot have that, so while
// the communicators are set up file descriptor 1 points at descriptor 2.
struct StdoutToStderr {
    int saved;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
    }
};

// rank r's share of a frame's samples: contiguous ranges, the first spp % n ranks take one more
void sample_share(int spp, int n, int r, int& first, int& count) {
    const int base = spp / n, extra = spp % n;
    first = r * base + (r < extra ? r : extra);
    count = base + (r < extra ? 1 : 0);
}

// Runs job(i) for every member, each on its member's thread, and returns the first failure.
int on_every_member(PtGroup* g, const std::function<int(int)>& job) {
    const int n = (int)g->members.size();
    if (g->workers.empty()) {
        for (int i = 0; i < n; i++) {
            PT_GHIP(g, hipSetDevice(g->members[i]->device));
            const int rc = job(i);
            if (rc != PT_OK) return gfail(g, rc, g->members[i]->error);
        }
        return PT_OK;
    }
    for (int i = 0; i < n; i++) g->workers[i]->post([&job, i] { return job(i); });
    int first_rc = PT_OK;
    for (int i = 0; i < n; i++) {
        const int rc = g->workers[i]->wait();
        if (rc != PT_OK && first_rc == PT_OK) first_rc = gfail(g, rc, g->members[i]->error);
    }
    return first_rc;
}

// Points every member at the group's current buffer, cleared, once no reduce reads it any more.
int open_buffer(PtGroup* g) {
    const int b = g->cur;
    return on_every_member(g, [g, b](int i) -> int {
        PtContext* c = g->members[i];
        // the buffer's previous turn was reduced on the exchange stream: the clear waits for that on the device
        PT_HIP(c, hipStreamWaitEvent(c->stream, g->reduced[b][i], 0));
        PT_HIP(c, launch_clear(cfg_of(c), g->delta[b][i], g->n_floats));
        c->ext_accum = g->delta[b][i];
        return PT_OK;
    });
}

// One reduce of the current buffers onto the root, each rank's call on its exchange stream behind the event that
// marks the buffer filled; the root folds the sum into its accumulation and resolves its frame there too.  Afterwards
// the other buffer is the current one.
int exchange(PtGroup* g) {
    const int n = (int)g->members.size();
    const int b = g->cur;
    for (int i = 0; i < n; i++) {
        PtContext* c = g->members[i];
        PT_GHIP(g, hipSetDevice(c->device));
        PT_GHIP(g, hipEventRecord(g->filled[b][i], c->stream));
        PT_GHIP(g, hipStreamWaitEvent(g->xstream[i], g->filled[b][i], 0));
    }
    PT_GNCCL(g, ncclGroupStart());
    for (int i = 0; i < n; i++) {
        const ncclResult_t r = ncclReduce(g->delta[b][i], g->delta[b][i], g->n_floats, ncclFloat, ncclSum, 0, g->comms[i], g->xstream[i]);
        if (r != ncclSuccess) { (void)ncclGroupEnd(); return gfail(g, PT_ERR_RCCL, std::string("ncclReduce: ") + ncclGetErrorString(r)); }
    }
    PT_GNCCL(g, ncclGroupEnd());
    PtContext* root = g->members[0];
    PT_GHIP(g, hipSetDevice(root->device));
    LaunchCfg xcfg = cfg_of(root);
    xcfg.stream = g->xstream[0];
    PT_GHIP(g, launch_add(xcfg, root->d_accum.ptr, g->delta[b][0], g->n_floats));
    PT_GHIP(g, launch_resolve(xcfg, root->d_accum.ptr, root->d_frame.ptr, root->width * root->height, root->total_samples));
    for (int i = 0; i < n; i++) {
        PT_GHIP(g, hipSetDevice(g->members[i]->device));
        PT_GHIP(g, hipEventRecord(g->reduced[b][i], g->xstream[i]));
    }
    g->reduces++;
    g->in_buffer = 0;
    g->cur = b ^ 1;
    return PT_OK;
}

int group_error(PtGroup* g, std::string& msg) { msg = g->error; return PT_OK; }

int wait_for_exchanges(PtGroup* g) {
    for (size_t i = 0; i < g->members.size(); i++) {
        PT_GHIP(g, hipSetDevice(g->members[i]->device));
        PT_GHIP(g, hipStreamSynchronize(g->xstream[i]));
    }
    return PT_OK;
}

// Waits for every member.  A batch whose ray queues overflowed added nothing to its frame's buffer (any scene can have
// one when the queues are small against the batch: PT_OPT_QUEUE_FACTOR), and that buffer has been reduced since; the
// wait renders such batches again.  So the frames not reduced yet go through their reduce first, every member is then
// pointed at the other buffer, cleared, the repairs land there, and if there were any that buffer goes through one
// more reduce (of no new samples).  Frames are never held up for this; until the wait the root's image is short of
// those batches, nothing else.
int group_settle(PtGroup* g) {
    if (g->in_buffer > 0) {
        const int rc = exchange(g);
        if (rc != PT_OK) return rc;
    }
    {
        const int rc = open_buffer(g);
        if (rc != PT_OK) return rc;
    }
    uint64_t before = 0, after = 0;
    for (PtContext* c : g->members) before += c->repaired_batches;
    {
        const int rc = on_every_member(g, [g](int i) { return settle(g->members[i]); });
        if (rc != PT_OK) return rc;
    }
    for (PtContext* c : g->members) after += c->repaired_batches;
    if (after != before) {
        const int rc = exchange(g);
        if (rc != PT_OK) return rc;
        const int rc2 = open_buffer(g);
        if (rc2 != PT_OK) return rc2;
    }
    return wait_for_exchanges(g);
}

void release_group(PtGroup* g) {
    for (GroupWorker* w : g->workers) { w->stop(); delete w; }
    g->workers.clear();
    for (size_t i = 0; i < g->members.size(); i++) {
        PtContext* c = g->members[i];
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (i < g->xstream.size() && g->xstream[i]) (void)hipStreamSynchronize(g->xstream[i]);
        if (i < g->comms.size() && g->comms[i]) (void)ncclCommDestroy(g->comms[i]);
        for (int b = 0; b < 2; b++) {
            if (i < g->delta[b].size() && g->delta[b][i]) (void)hipFree(g->delta[b][i]);
            if (i < g->filled[b].size() && g->filled[b][i]) (void)hipEventDestroy(g->filled[b][i]);
            if (i < g->reduced[b].size() && g->reduced[b][i]) (void)hipEventDestroy(g->reduced[b][i]);
        }
        if (i < g->xstream.size() && g->xstream[i]) (void)hipStreamDestroy(g->xstream[i]);
        c->ext_accum = nullptr;
        c->group = nullptr;
    }
    delete g;
}

}  // namespace

extern "C" {

int pt_group_create(int n, PtContext* const* members, PtGroup** out) {
    if (!out) return PT_ERR_INVALID_ARG;
    *out = nullptr;
    if (n <= 0 || n > 64 || !members) return PT_ERR_INVALID_ARG;
    try {
        for (int i = 0; i < n; i++) {
            PtContext* c = members[i];
            if (!c) return PT_ERR_INVALID_ARG;
            if (c->device == PT_DEVICE_NONE || !c->built || c->width <= 0) return fail(c, PT_ERR_NOT_READY, "a group member needs a GPU, a built scene and a resolution");
            if (c->group) return fail(c, PT_ERR_INVALID_ARG, "the context is a member of a group already");
            if (c->width != members[0]->width || c->height != members[0]->height) return fail(c, PT_ERR_INVALID_ARG, "group members differ in resolution");
            if (c->mesh.num_tris() != members[0]->mesh.num_tris() || c->accel != members[0]->accel) return fail(c, PT_ERR_INVALID_ARG, "group members hold different scenes");
            for (int k = 0; k < i; k++)
                if (members[k]->device == c->device) return fail(c, PT_ERR_INVALID_ARG, "two group members on one GPU");
        }
        PtGroup* g = new PtGroup();
        g->n_floats = (size_t)members[0]->width * members[0]->height * 3;
        g->comms.assign((size_t)n, nullptr);
        g->xstream.assign((size_t)n, nullptr);
        for (int b = 0; b < 2; b++) {
            g->delta[b].assign((size_t)n, nullptr);
            g->filled[b].assign((size_t)n, nullptr);
            g->reduced[b].assign((size_t)n, nullptr);
        }
        std::vector<int> devices;
        for (int i = 0; i < n; i++) {
            g->members.push_back(members[i]);
            devices.push_back(members[i]->device);
            members[i]->group = g;
        }
        ncclResult_t r;
        {
            StdoutToStderr quiet;
            r = ncclCommInitAll(g->comms.data(), n, devices.data());
        }
        if (r != ncclSuccess) {
            fail(members[0], PT_ERR_RCCL, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
            for (ncclComm_t& cm : g->comms) cm = nullptr;
            release_group(g);
            return PT_ERR_RCCL;
        }
        for (int i = 0; i < n; i++) {
            PtContext* c = members[i];
            bool ok = hipSetDevice(c->device) == hipSuccess && hipStreamCreateWithFlags(&g->xstream[i], hipStreamNonBlocking) == hipSuccess;
            for (int b = 0; b < 2 && ok; b++)
                ok = hipMalloc((void**)&g->delta[b][i], g->n_floats * sizeof(float)) == hipSuccess &&
                     hipEventCreateWithFlags(&g->filled[b][i], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&g->reduced[b][i], hipEventDisableTiming) == hipSuccess &&
                     hipEventRecord(g->reduced[b][i], g->xstream[i]) == hipSuccess;
            if (!ok) {
                fail(members[0], PT_ERR_HIP, "pt_group_create: no stream, events or memory for a member's per-frame buffers");
                release_group(g);
                return PT_ERR_HIP;
            }
            c->ext_accum = g->delta[0][i];  // from here on the member's batches accumulate into a per-frame buffer
            // the member's stream, two batches in flight and the exchange stream: the four hardware queues of a process
            (void)leave_a_queue_for_the_collective(c);
        }
        if (n > 1) {
            for (int i = 0; i < n; i++) {
                GroupWorker* w = new GroupWorker();
                g->workers.push_back(w);
                w->thread = std::thread(&GroupWorker::run, w, members[i]->device);
            }
        }
        *out = g;
        int rc = pt_group_reset(g);
        if (rc != PT_OK) { *out = nullptr; fail(members[0], rc, g->error); release_group(g); }
        return rc;
    } catch (...) {
        return PT_ERR_NO_MEMORY;
    }
}

void pt_group_destroy(PtGroup* g) {
    if (g) release_group(g);
}

const char* pt_group_last_error(PtGroup* g) { return g ? g->error.c_str() : "null group"; }

int pt_group_reset(PtGroup* g) {
    if (!g) return PT_ERR_INVALID_ARG;
    try {
        // whatever is on its way to the root belongs to the image that is being thrown away, but it must have landed
        // before the root's accumulation is cleared
        int rc = wait_for_exchanges(g);
        if (rc != PT_OK) return rc;
        g->in_buffer = 0;
        rc = on_every_member(g, [g](int i) -> int {
            PtContext* c = g->members[i];
            const int r = pt_reset(c);  // current per-frame buffer, frame buffer, total samples
            if (r != PT_OK) return r;
            if (i == 0) PT_HIP(c, launch_clear(cfg_of(c), c->d_accum.ptr, g->n_floats));
            return PT_OK;
        });
        return rc;
    } catch (...) {
        return gfail(g, PT_ERR_NO_MEMORY, "pt_group_reset: exception");
    }
}

int pt_group_set_cadence(PtGroup* g, int frames_per_reduce) {
    if (!g || frames_per_reduce < 1 || frames_per_reduce > 1024) return gfail(g, PT_ERR_INVALID_ARG, "pt_group_set_cadence: between 1 and 1024 frames per reduce");
    try {
        if (g->in_buffer >= frames_per_reduce) {
            const int rc = exchange(g);
            if (rc != PT_OK) return rc;
            const int rc2 = open_buffer(g);
            if (rc2 != PT_OK) return rc2;
        }
        g->cadence = frames_per_reduce;
        return PT_OK;
    } catch (...) {
        return gfail(g, PT_ERR_NO_MEMORY, "pt_group_set_cadence: exception");
    }
}

int pt_group_render(PtGroup* g, int spp) {
    if (!g || spp <= 0) return gfail(g, PT_ERR_INVALID_ARG, "pt_group_render: bad arguments");
    try {
        const int n = (int)g->members.size();
        const int b = g->cur;
        const bool fresh = g->in_buffer == 0;  // first frame into this buffer: clear it once its last reduce is through
        // every member's launches are queued by the member's own thread
        int rc = on_every_member(g, [g, n, b, spp, fresh](int i) -> int {
            PtContext* c = g->members[i];
            if (fresh) {
                PT_HIP(c, hipStreamWaitEvent(c->stream, g->reduced[b][i], 0));
                PT_HIP(c, launch_clear(cfg_of(c), g->delta[b][i], g->n_floats));
            }
            c->ext_accum = g->delta[b][i];
            int first, count;
            sample_share(spp, n, i, first, count);
            return render_range(c, spp, first, count);
        });
        if (rc != PT_OK) return rc;
        g->in_buffer++;
        g->frames++;
        // one reduce per `cadence` frames onto the root; no member is waited for (group_settle: what an overflowed batch owes)
        const bool reduce_now = g->in_buffer >= g->cadence;
        if (reduce_now) {
            rc = exchange(g);
            if (rc != PT_OK) return rc;
        }
        for (int i = 0; i < n; i++) {
            // the frame ends on a member where its share of the exchange does (or, between reduces, its accumulate)
            PtContext* c = g->members[i];
            PT_GHIP(g, hipSetDevice(c->device));
            while (c->frame_ends.size() >= 64) {
                (void)hipEventSynchronize(c->frame_ends.front());
                c->event_pool.push_back(c->frame_ends.front());
                c->frame_ends.pop_front();
            }
            hipEvent_t e = take_event(c);
            if (!e) return gfail(g, PT_ERR_HIP, "no event for the end of the frame");
            PT_GHIP(g, hipEventRecord(e, reduce_now ? g->xstream[i] : c->stream));
            c->frame_ends.push_back(e);
        }
        return PT_OK;
    } catch (...) {
        return gfail(g, PT_ERR_NO_MEMORY, "pt_group_render: exception");
    }
}

int pt_group_synchronize(PtGroup* g) {
    if (!g) return PT_ERR_INVALID_ARG;
    try {
        return group_settle(g);
    } catch (...) {
        return gfail(g, PT_ERR_NO_MEMORY, "pt_group_synchronize: exception");
    }
}

int pt_group_wait_frames(PtGroup* g, int keep) {
    if (!g || keep < 0) return PT_ERR_INVALID_ARG;
    try {
        for (PtContext* c : g->members) {
            PT_GHIP(g, hipSetDevice(c->device));
            while ((int)c->frame_ends.size() > keep) {
                PT_GHIP(g, hipEventSynchronize(c->frame_ends.front()));
                c->event_pool.push_back(c->frame_ends.front());
                c->frame_ends.pop_front();
            }
        }
        return PT_OK;
    } catch (...) {
        return gfail(g, PT_ERR_NO_MEMORY, "pt_group_wait_frames: exception");
    }
}

int pt_group_frames_in_flight(PtGroup* g) {
    if (!g) return PT_ERR_INVALID_ARG;
    int most = 0;
    for (PtContext* c : g->members) {
        (void)hipSetDevice(c->device);
        const int n = pt_frames_in_flight(c);
        most = n > most ? n : most;
    }
    return most;
}

int pt_group_size(PtGroup* g) { return g ? (int)g->members.size() : PT_ERR_INVALID_ARG; }

int pt_group_stats(PtGroup* g, uint64_t* frames, uint64_t* reduces, int* host_threads) {
    if (!g) return PT_ERR_INVALID_ARG;
    if (frames) *frames = g->frames;
    if (reduces) *reduces = g->reduces;
    if (host_threads) *host_threads = (int)g->workers.size();
    return PT_OK;
}

}  // extern "C"
