/* pt_amd.h -- C ABI of the MI355X path-tracing library (libpt_amd.so).
 *
 * The reference has no plugin or FFI interface: its renderer is reached through member calls on
 * global objects.  Each entry point below replaces one of those calls (or one global the call
 * reads); the citation names the reference lines, relative to the reference root.
 *
 * Conventions: every function returns PT_OK (0) or a negative PtStatus; nothing throws and
 * nothing is printed.  pt_last_error() returns a description of the last failure on a context.
 * A context belongs to one host thread and one GPU.  Arrays passed in are copied; arrays passed
 * out are written into caller memory.  Images are W*H*3 floats, row 0 = BOTTOM row, pixel
 * (i, j) at index i + j*W (src/cs488.h:195-198).
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtContext PtContext;
typedef struct PtGroup PtGroup;

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = -1,
    PT_ERR_IO = -2,           /* file missing or unreadable */
    PT_ERR_INVALID_OBJ = -3,  /* .obj parsed to zero triangles (src/cs488.h:714) */
    PT_ERR_NOT_READY = -4,    /* call order: mesh, build, resolution, then render */
    PT_ERR_HIP = -5,          /* a HIP runtime call failed; see pt_last_error */
    PT_ERR_QUEUE_OVERFLOW = -6, /* forked paths outgrew the ray queues even for a batch of 64 samples */
    PT_ERR_TREE_DEPTH = -7,   /* accelerator deeper than the traversal stack */
    PT_ERR_UNSUPPORTED = -8,
    PT_ERR_NO_MEMORY = -9,    /* host allocation failed (or another C++ exception was caught at the boundary) */
    PT_ERR_RCCL = -10         /* an RCCL call failed; see pt_last_error */
} PtStatus;

/* Accelerator kinds: the reference selects one at compile time with `#define KD_TREE`
 * (src/cs488.h:1197, on by default). */
enum { PT_ACCEL_BVH = 0, PT_ACCEL_KD = 1 };

/* Material types (src/cs488.h:363-367). */
enum { PT_MAT_LAMBERTIAN = 0, PT_MAT_METAL = 1, PT_MAT_GLASS = 2 };

/* The render-time fields of the reference's Material (src/cs488.h:368-389).  `name` keeps the
 * trailing newline the reference's parser leaves on it (src/cs488.h:871). */
typedef struct PtMaterial {
    char name[96];
    int32_t type;
    float eta;
    float Ka[3], Kd[3], Ks[3], Ke[3];
    float Ns, opacity, roughness;
    int32_t textured;      /* map_Kd seen (src/cs488.h:889-893) */
    int32_t tex_width, tex_height;
} PtMaterial;

/* Options for pt_set_option / pt_get_option. */
enum {
    PT_OPT_SEED = 1,          /* u64 seed of the counter-based sampler (default 0xcafef00dd15ea5e5, src/cs488.h:131) */
    PT_OPT_MAX_LEVEL = 2,     /* maxLevel (src/cs488.h:60), default 15 */
    PT_OPT_COUNT_TESTS = 3,   /* 1: trace kernels count box and triangle tests (slower) */
    PT_OPT_PROFILE = 4,       /* 1: bracket every kernel with HIP events on the context's stream; 2: only the launches
                                 of the closest-hit kernel (the events cost about 4 % of a frame at level 1, 1 % at 2) */
    PT_OPT_MAX_BATCH = 5,     /* most camera samples in flight in one wavefront batch */
    PT_OPT_QUEUE_FACTOR = 6,  /* queue capacity = factor x batch (forked paths), default 4; a batch that
                               * outgrows it adds nothing and is rendered again in halves at the next
                               * pt_synchronize / pt_download / pt_get_counters (PT_CNT_REPAIRED) */
    PT_OPT_LDS_NODES = 7,     /* tree nodes staged in LDS by the trace kernels (0 = none; 65535, the default = the
                                 largest of 256/128/64/0 that leaves five workgroups resident per CU).  Reads back as
                                 the number in use once the tree is built */
    PT_OPT_TRAVERSAL = 8,     /* trace-kernel form; every setting gives bit-identical hits.  bit 0: slab and triangle
                                 quotients from an exact reciprocal plus one exact correction step instead of the
                                 division sequence (default on); bit 1 (measurement; with bit 0 set and bits 2 and 15 clear): every lane
                                 walks two rays at a time and requests both rays' nodes before it takes either step; every
                                 workgroup traces its own region.  Measured 7-19 % slower per launch than one ray a lane on every
                                 scene; off by default; bit 2: refill a wave only when all its lanes are idle
                                 (measurement); bit 3: every workgroup traces the rays of its own queue region, also where the default
                                 lets the workgroups draw chunks of 64 rays from shared pools (pt_trace_batch, and
                                 rendering with PT_OPT_STREAMS = 1); bits 4..7: k of the node phase's early-leave rule (leave when the lanes
                                 on interior nodes number less than 1/k of those waiting at a leaf; 0, the default:
                                 4 for a scene held entirely in LDS, else 1; 15 = never leave early); bits 8..14: idle
                                 lanes a wave waits for before it fetches new rays (0, the default: 32 for a scene held
                                 entirely in LDS, else 8); bit 15 (measurement; a BVH that is walked in memory): nodes are
                                 fetched as 192-byte records that carry both children's node records, so a step that descends
                                 takes the child's step without another fetch -- one dependent fetch per two levels of
                                 BVH::traverse (src/cs488.h:1671-1706).  Measured 20-47 % slower than the 64-byte nodes (the
                                 walk is bound by the bytes it gathers, not by the length of its fetch chain); off by default */
    PT_OPT_TAIL_LEVEL = 9,    /* bounce levels from this one on are finished by one fused launch in which every
                                 workgroup alternates trace and shade on its own queue region (default 255 = per scene: 6,
                                 the level after Russian roulette starts, src/cs488.h:2153, for a tree held in LDS (2 for
                                 frames of at most 4 Mi samples; 1 for such a kd-tree in a plain Lambertian scene rendered one
                                 batch at a time), else 4 (2 for such small frames), and earlier once an earlier batch has shown that a
                                 level holds few rays; 0 = one launch per level and stage).
                                 Same device code either way, so the image does not depend on it */
    PT_OPT_BUILD_THREADS = 10,/* host threads pt_build may use for the tree (0 = all, default; 1 = none extra).  Large
                                 subtrees near the root are built concurrently and spliced in at their sequential
                                 positions: the tree is the same for every value */
    PT_OPT_OBJ_EXTENDED = 11, /* 1: pt_load_obj also accepts what the reference's reader does not (src/cs488.h:1059 reads
                                 three vertices per face, :944-946 positive indices only): polygons as triangle fans,
                                 negative indices, face lines of any length.  Default 0 = the reference's behaviour */
    PT_OPT_BLOCKS_PER_CU = 12,/* workgroups (queue regions) per compute unit a batch is split over, 1..8; 0 = chosen per
                                 scene so that one launch of the trace kernel is resident at once */
    PT_OPT_TREE_IN_LDS = 14,  /* read only: 1 when the built scene's whole tree and its intersection records are staged in LDS by
                                 the trace kernels (small scenes: the walk then has no global-memory path, so instruction
                                 issue and not memory bandwidth bounds it) */
    PT_OPT_TAIL_LEVEL_USED = 15, /* read only: the level at which the fused launch took over in the batch queued last (0 = every
                                 level had launches of its own); a measurement that changes the schedule (PT_OPT_STREAMS = 1)
                                 sets PT_OPT_TAIL_LEVEL to this to keep timing the same launches */
    PT_OPT_STREAMS = 13       /* batches in flight, 1..4.  Unless set, chosen when a frame is queued: 3, or 2 for frames of
                                 more than 4 Mi samples of a scene whose tree is walked in LDS and has no translucent surface
                                 (a third batch gains nothing there); and 2 from the moment the context joins a group or is
                                 given an external per-frame buffer, which leaves a hardware queue to the collective library.  Consecutive batches (frames, when a frame is one
                                 batch) run on internal streams with queues of their own, so the thinly populated late
                                 levels of one overlap the early levels of the next; the kernels that touch the image
                                 (accumulate, resolve, clear) stay on the context's stream in program order, so the image
                                 and everything the caller orders on that stream are as with 1.  With PT_OPT_PROFILE the
                                 per-kernel times then include the time a kernel shares the chip with another batch */
};

/* Counter indices for pt_get_counters (all cumulative since pt_reset_counters). */
enum {
    PT_CNT_RAYS = 0,        /* closest-hit queries (Scene::intersect calls, src/cs488.h:1750) */
    PT_CNT_BOX_TESTS = 1,   /* AABB::intersect calls (src/cs488.h:530); needs PT_OPT_COUNT_TESTS */
    PT_CNT_TRI_TESTS = 2,   /* raytraceTriangle calls (src/cs488.h:646); needs PT_OPT_COUNT_TESTS */
    PT_CNT_SAMPLES = 3,     /* camera samples started */
    PT_CNT_NAN_SAMPLES = 4, /* samples dropped by the NaN check (src/cs488.h:1841-1858) */
    PT_CNT_HITS = 5,        /* closest-hit queries that found a triangle */
    PT_CNT_MAX_QUEUE = 6,   /* largest ray-queue population seen */
    PT_CNT_REPAIRED = 7,    /* batches rendered again in halves after their ray queues overflowed */
    PT_CNT_TAIL_RAYS = 8,   /* the part of PT_CNT_RAYS traced by the fused launch of the deep levels */
    PT_CNT_WAVE_STEPS = 9,  /* node steps issued by wavefronts (each serves up to 64 rays: PT_CNT_BOX_TESTS / 2 /
                               (64 x this) is the share of lanes at work in the node phase); needs PT_OPT_COUNT_TESTS */
    PT_CNT_WAVE_ROUNDS = 10,/* triangle rounds issued by wavefronts, likewise against PT_CNT_TRI_TESTS */
    PT_CNT_STREAM_SWAPS = 11,/* internal streams replaced because they ran in order with another stream of the context
                               (hardware queue shared; see PT_OPT_STREAMS); not reset by pt_reset_counters */
    PT_CNT_COUNT = 12
};

/* Timer indices for pt_get_timers (milliseconds / launches, cumulative; needs PT_OPT_PROFILE). */
enum {
    PT_TIMER_RAYGEN = 0, PT_TIMER_TRACE = 1, PT_TIMER_SHADE = 2, PT_TIMER_ACCUM = 3,
    PT_TIMER_TAIL = 4,      /* the fused launch of the deep levels (trace and shade together) */
    PT_TIMER_COUNT = 5
};

/* pt_create(PT_DEVICE_NONE, ...) makes a context that never touches a GPU: it can load scenes and
 * build accelerator trees (pt_load_obj ... pt_build, pt_get_tree, pt_get_mesh) and refuses, with
 * PT_ERR_NOT_READY, everything that renders or traces. */
#define PT_DEVICE_NONE (-1)

/* ---- lifetime ------------------------------------------------------------------------------ */
/* Replaces the static construction of globalScene (src/cs488.h:1900).  `device` is a HIP device
 * ordinal.  Fails with PT_ERR_HIP when no GPU is usable: there is no CPU fallback. */
int pt_create(int device, PtContext** out);
/* (A member of a group is left alone: destroy the group first.) */
void pt_destroy(PtContext* ctx);
const char* pt_last_error(PtContext* ctx);
/* The stream (hipStream_t as void*) on which the context orders its work: the kernels that touch the image run on
 * it, the rest of a batch on internal streams joined to it by events.  Until this is called it is the device's
 * null stream (which holds one of the process's four hardware queues anyway; a further stream would take one
 * from the batches in flight). */
int pt_set_stream(PtContext* ctx, void* hip_stream);
int pt_set_option(PtContext* ctx, int option, uint64_t value);
int pt_get_option(PtContext* ctx, int option, uint64_t* value);

/* ---- scene: TriangleMesh::load / createSingleTriangle / Image::load / Scene::addObject ------ */
/* src/cs488.h:703-788 with ParseOBJ :917-1194 and LoadMTL :854-915.  PT_ERR_INVALID_OBJ when the
 * file yields no triangle; the caller then falls back as src/main.cpp:9-13 does. */
int pt_load_obj(PtContext* ctx, const char* path);
/* src/cs488.h:814-835 */
int pt_single_triangle(PtContext* ctx);
/* src/cs488.h:200-218 (Radiance .hdr; rows are flipped so that row 0 is the bottom row) */
int pt_load_env(PtContext* ctx, const char* hdr_path);
/* The same state set from memory.  pos9/nrm9: 9 floats per triangle (3 vertices x xyz);
 * uv6: 6 floats per triangle or NULL; matid: one index per triangle or NULL (all 0). */
int pt_set_mesh(PtContext* ctx, int n_tris, const float* pos9, const float* nrm9, const float* uv6, const int32_t* matid);
int pt_set_materials(PtContext* ctx, int n, const PtMaterial* mats);
/* 8-bit RGB texels, first row first, as the reference's decoder returns them (src/cs488.h:841). */
int pt_set_texture(PtContext* ctx, int material, const uint8_t* rgb, int width, int height);
/* float RGB environment image, first row of the file first (flipped on store like Image::load). */
int pt_set_env(PtContext* ctx, const float* rgb, int width, int height);
int pt_num_triangles(PtContext* ctx);
int pt_num_materials(PtContext* ctx);
int pt_get_mesh(PtContext* ctx, float* pos9, float* nrm9, float* uv6, int32_t* matid);
int pt_get_material(PtContext* ctx, int index, PtMaterial* out);
/* decoded texture of a material (tex_width*tex_height*3 bytes) */
int pt_get_texture(PtContext* ctx, int material, uint8_t* rgb);
/* stored environment image (row 0 = bottom); width/height may be queried with rgb == NULL */
int pt_get_env(PtContext* ctx, float* rgb, int* width, int* height);

/* ---- accelerator: Scene::preCalc (src/cs488.h:1733-1747) ------------------------------------ */
/* Builds the SAH BVH (src/cs488.h:1490-1667) or the SAH kd-tree (src/cs488.h:1249-1320,
 * 1388-1400) on the CPU, flattens it and uploads mesh, materials, textures, environment and
 * tree to the GPU. */
int pt_build(PtContext* ctx, int accel);
/* Host tree in the reference's node order (BVH: pre-order array index; kd: node, child pair,
 * left subtree, right subtree).  Call with all pointers NULL to read the sizes. */
int pt_get_tree(PtContext* ctx, int* n_nodes, int* n_refs, int* max_depth,
                int32_t* left, int32_t* right, int32_t* first, int32_t* count, float* box6, int32_t* refs);

/* ---- camera and film: the globals of src/cs488.h:48-49,74-87 ------------------------------- */
int pt_set_camera(PtContext* ctx, const float eye[3], const float lookat[3], const float up[3], float fov_deg, float film_size);
/* (Re)allocates and clears accumulation and frame buffers, total sample count = 0. */
int pt_set_resolution(PtContext* ctx, int width, int height);
/* FrameBuffer.clear + AccumulationBuffer.clear + totalSamples = 0 (src/cs488.h:2288-2290) */
int pt_reset(PtContext* ctx);

/* ---- the hot path: Scene::Pathtrace (src/cs488.h:1878-1897) --------------------------------- */
/* One progressive frame: totalSamples += spp, spp new samples for every pixel are added to the
 * accumulation buffer and the frame buffer is set to accumulation / totalSamples.  Asynchronous
 * on the context's stream; pt_download / pt_synchronize wait for it. */
int pt_render(PtContext* ctx, int spp);
/* Sample-split form for several GPUs: of a frame that adds spp_frame samples per pixel, this
 * context renders only sample indices [first, first + count) into its own accumulation buffer;
 * totalSamples still advances by spp_frame.  The caller sums the accumulation buffers of all
 * contexts (pt_accum_device_ptr) and then calls pt_resolve. */
int pt_render_split(PtContext* ctx, int spp_frame, int first, int count);
/* frame = accumulation / totalSamples (src/cs488.h:1860) */
int pt_resolve(PtContext* ctx);
int pt_total_samples(PtContext* ctx);
/* Waits for the stream, then completes every batch whose ray queues overflowed (see
 * PT_OPT_QUEUE_FACTOR) into the accumulation buffer that is current at the time of the call and
 * resolves again.  A caller that hands external buffers to a collective and renders scenes whose
 * paths fork heavily calls this before the collective. */
int pt_synchronize(PtContext* ctx);
/* Returns when at most `keep` of the frames queued by pt_render / pt_render_split are still unfinished (0 = all have
 * ended), leaving the others running: a frame loop that shows no pixels paces itself with this instead of draining the
 * GPU every frame as Scene::Pathtrace's join does (src/cs488.h:1893-1895).  Unlike pt_synchronize it does not render
 * overflowed batches again; pt_synchronize / pt_download before the pixels are read stay as they are. */
int pt_wait_frames(PtContext* ctx, int keep);
/* How many of the frames queued by pt_render / pt_render_split have not ended on the GPU yet; looks at the frames' end
 * events and never waits.  A frame loop that overlaps frames prints a frame's lines (src/cs488.h:1896, 2322: after the
 * join) once this says the frame is through. */
int pt_frames_in_flight(PtContext* ctx);
/* which: 0 = accumulation buffer, 1 = frame buffer.  Synchronises. */
int pt_download(PtContext* ctx, int which, float* dst);
/* Device address of the accumulation buffer (W*H*3 floats), for a collective over it. */
void* pt_accum_device_ptr(PtContext* ctx);
/* Render into caller-owned device memory (W*H*3 floats) instead; NULL restores the internal one.
 * Takes effect for the launches issued after the call (no synchronisation), so a caller may
 * alternate between two buffers and reduce one while the next frame renders into the other.
 * A batch whose ray queues overflowed (PT_OPT_QUEUE_FACTOR) is rendered again by the next call that waits, into
 * the buffer that is current THEN: a caller that reduces buffers without waiting names a cleared buffer of its own
 * before it calls pt_synchronize and sends that through its collective as well. */
int pt_use_external_accum(PtContext* ctx, void* device_ptr);
/* Image::save (src/cs488.h:219-232): clamp to [0,1], truncate to 8 bits, PNG, top row first. */
int pt_save_png(PtContext* ctx, int which, const char* path);
/* Raw float dump of the same image as a little-endian PFM file (bottom row first). */
int pt_save_pfm(PtContext* ctx, int which, const char* path);
/* Recording (the window's `F` key, src/cs488.h:255-266: GifBegin / GifEnd; :2297-2310: one GifWriteFrame per
 * frame after Pathtrace, rows top-down, the same clamp-and-truncate as Image::save): pt_record_begin opens an
 * animated GIF of the context's resolution (delay in 1/100 s; the reference uses 1), pt_record_frame appends the
 * current frame buffer, pt_record_end closes the file (also done by pt_destroy). */
int pt_record_begin(PtContext* ctx, const char* path, int delay_cs);
int pt_record_frame(PtContext* ctx);
int pt_record_end(PtContext* ctx);
/* The same encoder on caller-supplied frames (n_frames x height x width x 3 bytes, rows top-down); needs no GPU. */
int pt_encode_gif(const char* path, const uint8_t* rgb8, int n_frames, int width, int height, int delay_cs);

/* ---- several GPUs: the thread pool of Scene::Pathtrace (src/cs488.h:1885-1895) ---------------- */
/* The reference splits a frame over numThreads threads by image columns and joins them (src/cs488.h:1885-1895).
 * Here a frame is split over the GPUs of one node by SAMPLES: a group is n contexts of this process, one per GPU,
 * holding the same scene, accelerator, camera, options and resolution (the caller sets each of them up alike);
 * member r renders sample indices sample_share(spp, n, r) of every pixel into a per-frame buffer on its GPU, and ONE
 * RCCL ncclReduce(float, sum) per frame over xGMI adds those buffers onto members[0], which folds the sum into its
 * accumulation buffer and resolves its frame buffer.  Each member's launches are queued by a host thread of its own; a
 * member has two per-frame buffers used in turn and its share of the reduce runs on a stream of its own, so the next
 * frame's accumulate kernels do not queue behind the reduce.  The sampler is keyed by (pixel, global sample index), so the
 * image equals the single-GPU one up to the order of the float additions.  Everything is queued on the members'
 * streams; pt_group_synchronize (or pt_download on members[0]) waits -- and completes the batches whose ray queues
 * overflowed (PT_OPT_QUEUE_FACTOR): they are rendered again into cleared per-frame buffers, which go through one more
 * reduce; until such a wait the root's image is short of them.  While a context is in a group it is driven
 * through the group calls only (plus pt_set_camera, pt_download / pt_save_* on members[0], counters and timers);
 * pt_group_destroy releases the communicators and returns the contexts to stand-alone use. */
int pt_group_create(int n, PtContext* const* members, PtGroup** out);
void pt_group_destroy(PtGroup* group);
const char* pt_group_last_error(PtGroup* group);
int pt_group_size(PtGroup* group);
/* One progressive frame over all members: totalSamples += spp on each of them (src/cs488.h:1882). */
int pt_group_render(PtGroup* group, int spp);
/* FrameBuffer.clear + AccumulationBuffer.clear + totalSamples = 0 on every member (src/cs488.h:2288-2290) */
int pt_group_reset(PtGroup* group);
int pt_group_synchronize(PtGroup* group);
/* pt_wait_frames for a group: returns once at most `keep` of the frames queued by pt_group_render are unfinished on
 * any member (a frame ends on a member with its share of the reduce, on members[0] with the resolved frame).  Renders
 * no overflowed batch again; pt_group_synchronize before the pixels are read stays as it is. */
int pt_group_wait_frames(PtGroup* group, int keep);
/* Frames added into one per-frame buffer before the buffer is reduced (default 1: the reference joins its strips every
 * frame, src/cs488.h:1893-1895).  A reduce moves W*H*12 bytes per member whatever the frame's sample count, so for frames
 * of a few milliseconds (12.6 MB against a 2-ms frame at 1024 x 1024 x 10 spp) it is paid once per k frames instead: the
 * members keep adding frames into their current buffer, every k-th pt_group_render (and every pt_group_synchronize, for
 * the frames since) reduces it.  The sum is the same; members[0]'s frame buffer then shows the state of the last reduce,
 * fewer than k frames behind, until pt_group_synchronize / pt_download. */
int pt_group_set_cadence(PtGroup* group, int frames_per_reduce);
/* pt_frames_in_flight for a group: the largest count over the members. */
int pt_group_frames_in_flight(PtGroup* group);
/* Frames queued and reduces issued since the group was made, and the host threads that queue the members' launches
 * (one per member, as the reference gives every strip a thread, src/cs488.h:1885-1890; 0 for a group of one). */
int pt_group_stats(PtGroup* group, uint64_t* frames, uint64_t* reduces, int* host_threads);

/* ---- parity entry: Scene::intersect on a batch of rays (src/cs488.h:1750-1777) ------------- */
/* rays6: n x (origin xyz, direction xyz) in host memory.  tri: index of the winning triangle in
 * mesh order or -1; t: hit distance or FLT_MAX; bary2 (optional): the beta, gamma of
 * src/cs488.h:661-662. */
int pt_trace_batch(PtContext* ctx, int n, const float* rays6, int32_t* tri, float* t, float* bary2);
/* pt_trace_batch for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  All four arrays are DEVICE memory of the
 * context's GPU (d_bary2 may be NULL); the call only queues kernels on the context's stream and returns, so the
 * results are ready once that stream has been waited for (pt_synchronize, or the caller's own stream given to
 * pt_set_stream).  Neither this call nor anything it calls allocates or frees: the packed rays and the hit records
 * are staged in a file-scope device array of 6 MiB that every GPU the library runs on holds until the process
 * ends.  Limits: n <= PT_TRACE_STATIC_MAX_RAYS, otherwise PT_ERR_INVALID_ARG.  All contexts on one GPU share
 * that GPU's array: two such calls that overlap on one GPU (different streams or threads) race. */
#define PT_TRACE_STATIC_MAX_RAYS (1 << 17)
int pt_trace_batch_static(PtContext* ctx, int n, const float* d_rays6, int32_t* d_tri, float* d_t, float* d_bary2);
/* The reference's camera rays for the current camera and resolution, pixel-centre form
 * (src/cs488.h:1780-1794), n = W*H rays in pixel order; for tests of the camera model. */
int pt_camera_rays(PtContext* ctx, float* rays6);

/* Scene::generateRay (src/cs488.h:1825-1839) as the ray-generation kernel evaluates it: for the global sample index
 * `sample` (0 = the first sample after pt_reset) the jittered camera ray of every pixel, n = W*H rays in pixel order,
 * read back from the ray queue the kernel wrote.  For parity tests of the kernel; renders nothing. */
int pt_camera_samples(PtContext* ctx, int sample, float* rays6);

/* ---- measurement ---------------------------------------------------------------------------- */
int pt_get_counters(PtContext* ctx, uint64_t out[PT_CNT_COUNT]);
int pt_reset_counters(PtContext* ctx);
/* ms[i], launches[i] for i < PT_TIMER_COUNT; synchronises. */
int pt_get_timers(PtContext* ctx, double ms[PT_TIMER_COUNT], uint64_t launches[PT_TIMER_COUNT]);

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_H */
