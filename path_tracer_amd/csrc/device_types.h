// device_types.h -- GPU-side data layout of the wavefront path tracer and the host launchers of
// its kernels.
//
// Tree node (64 bytes, four float4): an INTERIOR node of the host tree, holding the boxes of
// both children so that one node fetch serves the pair of slab tests the reference makes per
// visit (src/cs488.h:1684-1685, 1357-1358).
//     q0 = (Lmin.x, Lmin.y, Lmin.z, Lmax.x)   q1 = (Lmax.y, Lmax.z, Rmin.x, Rmin.y)
//     q2 = (Rmin.z, Rmax.x, Rmax.y, Rmax.z)   q3 = bit patterns of (left, right, 0, 0)
// A child reference >= 0 is the index of another interior node; a reference < 0 is a leaf (see
// the intersection records below).  Interior nodes are numbered breadth-first, so the first K
// nodes are the top of the tree (the part staged in LDS).
//
// Intersection records (`isect`, four float4 = 64 bytes each): everything the triangle test needs
// that does not depend on the ray, computed on the host with the reference's own operations
// (src/cs488.h:651-652, 656 with src/linalg.h:500), so the values are bit for bit what the test
// would compute itself:
//     r0 = (p0.xyz, triangle index)   r1 = (a = p0 - p1, n)   r2 = (b = p0 - p2, 0)   r3 = (cross(a, b), 0)
// leaf_inline = 1 (BVH, and kd-trees with few leaf references): records are stored in leaf order, a
// leaf reference is ~(index of its first record) and n in the FIRST record of a leaf is the leaf's
// triangle count.  leaf_inline = 0 (kd-trees, whose leaves repeat triangles many times): records
// are stored in mesh order and a leaf reference is ~offset into `refs`, where refs[off] = n,
// refs[off+1..off+n] = triangle indices in the reference's leaf order.
//
// Triangles: three float4 per triangle in `tri_pos` (vertex k in .xyz; the .w of vertex 0 holds
// the material index as an int bit pattern), three float4 per triangle in `tri_nrm`, three
// float2 per triangle in `tri_uv` (absent when no material is textured).
//
// Materials: four float4 each: (Kd, type) (Ks, eta) (Ke, opacity) (roughness, tex_w, tex_h,
// texel offset), the last three as int bit patterns.
//
// Ray queues (structure of arrays, two sets swapped every bounce).  Every workgroup owns a
// private region of `cap_block` consecutive entries in each array and a private entry count per
// bounce level (qcount[level * blocks + block]): a workgroup consumes what it produced one level
// earlier and appends survivors with LDS atomics only, so no global atomic is on the path.
//     ray_o[i]  = (origin.xyz, miss_scale)     miss_scale multiplies the environment radiance a
//                                              miss adds (the pdf factor of src/cs488.h:2035)
//     ray_d[i]  = (direction.xyz, bits)        bits = path id << 8 | random dimension
//     beta[i]   = (throughput.rgb, slot)       slot = index of the camera sample in the batch
//     hit[i]    = (triangle, t, beta, gamma)   written by the trace stage
#pragma once

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

namespace pt {

struct SceneGpu {
    const float4* nodes;
    const float4* wide;   // BVH kept in memory: per interior node 12 float4 = the node, its left child's node record, its right
                          // child's (zeros where a child is a leaf); nullptr when not built
    const float4* probe_alias;  // the same address as `nodes` (measurement builds with PT_PROBE_EXTRA_FETCH load through it)
    const int* refs;
    const float4* isect;
    const float4* tri_pos;
    const float4* tri_nrm;
    const float2* tri_uv;
    const float4* mats;
    const uint8_t* texels;
    const float* env;  // rgb triples, row 0 = bottom
    int env_w, env_h;
    float root_lo[3], root_hi[3];
    int root_ref;
    int n_nodes;
    int lds_nodes;  // nodes [0, lds_nodes) are staged in LDS by the trace kernels
    int n_tris;
    int n_mats;       // materials in `mats`
    int leaf_inline;  // layout of the leaves, see above
    int n_records;    // intersection records in `isect`
    int n_refs;       // entries of `refs` (leaf_inline = 0)
    int shared_origin; // 1: every ray of the launch starts at ray_o[0] (camera rays are stored as directions only)
    int shading_kind; // which instantiation of the shading code the scene needs (SHADE_ALL / _LAMBERTIAN / _PLAIN / _GLASS / _METAL)
    int fixed_radiance; // 1: paths fork (translucent surfaces), per-sample radiance is summed in 64-bit fixed point (two float4 per sample)
    int planes_ok;  // every box coordinate is 0 or at least 1e-17 in magnitude (reciprocal slab form allowed)
};

// float4 records per material: Kd|type, Ks|eta, Ke|opacity, roughness|texture w,h,offset, Kd/pi
enum { MATERIAL_STRIDE = 5 };

struct CameraGpu {
    float eye[3], u[3], v[3], w[3];
    float aspect_film;  // aspect * film size
    float film;
    float dist;         // distance to the film plane
    int width, height;
};

struct QueueSet {
    float4* ray_o;
    float4* ray_d;
    float4* beta;
};

struct BatchGpu {
    int pixel0;        // first pixel (index i + j*W) of the batch
    int n_pixels;
    int chunk;         // samples per pixel in this batch
    int sample0;       // global index of the batch's first sample of every pixel
    uint64_t seed;
    int n_slots;       // n_pixels * chunk
    int blocks;        // workgroups that share the batch (grid size of every stage)
    int cap_block;     // queue entries owned by one workgroup
    int slots_block;   // most camera samples one workgroup generates
    int deal_shift;    // rotation of the run -> workgroup assignment per dealing round
    int fixed_radiance;  // copy of SceneGpu::fixed_radiance for the kernels that see only the batch
    const uint64_t* keys;  // sampler key of each of the batch's `chunk` sample indices
    int* overflow;     // the batch's flag: cleared by raygen, set by a workgroup whose queue region ran
                       // full, read by accumulate (a flagged batch adds nothing and is rendered again
                       // in smaller pieces by the host)
};

enum { TRACE_STACK = 64 };

// Device counters (uint64 each).  `totals` holds one set (DC_COUNT words) written by the tally and
// accumulate kernels; `block_stats` holds DC_BLOCK_COUNT words per workgroup, each updated only by
// its own workgroup with plain loads and stores, and is summed on the host.
// Child reference of a leaf without triangles (kd-trees have them): never entered by the walk.
enum : int { EMPTY_LEAF = INT_MIN + 1 };

enum { DC_RAYS = 0, DC_NAN = 1, DC_TAIL_RAYS = 2, DC_MAXQ = 3, DC_COUNT = 4 };
enum { DB_HITS = 0, DB_BOX = 1, DB_TRI = 2, DB_OVERFLOW = 3, DB_WAVE_STEPS = 4, DB_WAVE_ROUNDS = 5, DC_BLOCK_COUNT = 6 };

struct LaunchCfg {
    hipStream_t stream;
    int blocks;   // grid size of the streaming kernels (accumulate, resolve, clear)
};

// ---- launchers (each returns the hipError_t of the launch) ----
hipError_t launch_raygen(const LaunchCfg& cfg, const CameraGpu& cam, const BatchGpu& batch, QueueSet out,
                         float4* sample_rad, int* qcount);
// the camera rays of a one-sample-per-pixel batch, from the queue raygen wrote, in pixel order (n_pixels x 6 floats)
hipError_t launch_gather_camera_rays(const LaunchCfg& cfg, const CameraGpu& cam, const BatchGpu& batch, const float4* ray_d,
                                     const int* qcount, float* rays6);
// `grid_blocks` queue regions of `cap_block` entries, traced by as many workgroups, which draw the rays from
// shared pools: `cursors` = TRACE_POOL_INTS ints that are zero when the launch starts (the launch leaves them
// non-zero; one cursor per pool, each in a 256-byte stretch of its own, because atomics on one cache line are
// served one after the other).  With n_rays == nullptr the input is one flat array of n_fixed rays cut into consecutive regions.
enum { TRACE_POOL_CURSORS = 32, TRACE_POOL_STRIDE = 64, TRACE_POOL_INTS = TRACE_POOL_CURSORS * TRACE_POOL_STRIDE };
hipError_t launch_trace(const LaunchCfg& cfg, const SceneGpu& scene, int kd, int count_tests, int fast_slab,
                        const float4* ray_o, const float4* ray_d, float4* hit, const int* n_rays, int n_fixed,
                        int grid_blocks, int cap_block, int* cursors, unsigned long long* block_stats);
hipError_t launch_shade(const LaunchCfg& cfg, const SceneGpu& scene, const BatchGpu& batch, int level, int max_level,
                        QueueSet in, const float4* hit, QueueSet out, float4* sample_rad, const int* qcount_in,
                        int* qcount_out);
int trace_resident_blocks_per_cu(const SceneGpu& scene, int kd, int mode);
// false when not even one workgroup of the trace kernel could be resident with the scene's LDS stage
bool trace_kernel_fits(const SceneGpu& scene, int kd, int mode);
// 1 when the trace kernels stage the scene's whole tree and all intersection records in LDS
int trace_scene_in_lds(const SceneGpu& scene);
hipError_t launch_tail(const LaunchCfg& cfg, const SceneGpu& scene, const BatchGpu& batch, int kd, int mode, int level0, int levels,
                       int max_level, QueueSet qa, QueueSet qb, float4* hit, float4* sample_rad, int* qcount,
                       unsigned long long* block_stats);
hipError_t launch_accumulate(const LaunchCfg& cfg, const BatchGpu& batch, const float4* sample_rad, float* accum,
                             unsigned long long* totals);
// sums the per-workgroup queue counts of every level of a finished batch into totals
// (nothing is added for a batch whose overflow flag is set: its pieces are tallied when they are rendered again)
// (level_rays, if not null: the batch's rays per level, one word each, overwritten)
hipError_t launch_tally(const LaunchCfg& cfg, const int* qcount, int levels, int blocks, int tail_level, const int* overflow,
                        unsigned long long* totals, unsigned long long* level_rays);
hipError_t launch_resolve(const LaunchCfg& cfg, const float* accum, float* frame, int n_pixels, int total_samples);
hipError_t launch_clear(const LaunchCfg& cfg, float* buf, size_t n);
hipError_t launch_add(const LaunchCfg& cfg, float* dst, const float* src, size_t n);  // dst += src
// a single wave that stays busy for about `microseconds` (to probe whether two streams overlap)
hipError_t launch_spin(const LaunchCfg& cfg, int microseconds);
// unpacks rays6 (n x 6 floats) into queue form for the batch-trace entry
hipError_t launch_pack_rays(const LaunchCfg& cfg, const float* rays6, int n, float4* ray_o, float4* ray_d);
// Most rays of one pt_trace_batch_static call: its staging lives in a file-scope device array of
// 3 float4 per ray (6 MiB per GPU).
enum { TRACE_STATIC_MAX_RAYS = 1 << 17 };
// device address of that array on `device` (the current device); cached per device
hipError_t trace_static_scratch(int device, float4** out);
hipError_t launch_unpack_hits(const LaunchCfg& cfg, const float4* hit, int n, int* tri, float* t, float* bary2);

}  // namespace pt
