// device_types.h -- GPU-side data layout of the wavefront path tracer and the host launchers of
// its kernels.
//
// Tree node (64 bytes, four float4): an INTERIOR node of the host tree, holding the boxes of
// both children so that one node fetch serves the pair of slab tests the reference makes per
// visit (src/cs488.h:1684-1685, 1357-1358).
//     q0 = (Lmin.x, Lmin.y, Lmin.z, Lmax.x)   q1 = (Lmax.y, Lmax.z, Rmin.x, Rmin.y)
//     q2 = (Rmin.z, Rmax.x, Rmax.y, Rmax.z)   q3 = bit patterns of (left, right, 0, 0)
// A child reference >= 0 is the index of another interior node; a reference < 0 is a leaf and
// ~reference is the offset of its list in `refs`: refs[off] = n, refs[off+1..off+n] = triangle
// indices in the reference's leaf order.  Interior nodes are numbered breadth-first, so the
// first K nodes are the top of the tree (the part staged in LDS).
//
// Triangles: three float4 per triangle in `tri_pos` (vertex k in .xyz; the .w of vertex 0 holds
// the material index as an int bit pattern), three float4 per triangle in `tri_nrm`, three
// float2 per triangle in `tri_uv` (absent when no material is textured).
//
// Materials: four float4 each: (Kd, type) (Ks, eta) (Ke, opacity) (roughness, tex_w, tex_h,
// texel offset), the last three as int bit patterns.
//
// Ray queues (structure of arrays, capacity `cap` paths, two sets swapped every bounce):
//     ray_o[i]  = (origin.xyz, miss_scale)     miss_scale multiplies the environment radiance a
//                                              miss adds (the pdf factor of src/cs488.h:2035)
//     ray_d[i]  = (direction.xyz, bits)        bits = path id << 8 | random dimension
//     beta[i]   = (throughput.rgb, slot)       slot = index of the camera sample in the batch
//     hit[i]    = (triangle, t, beta, gamma)   written by the trace stage
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

struct SceneGpu {
    const float4* nodes;
    const int* refs;
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
};

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
    int cap;           // queue capacity
};

enum { TRACE_STACK = 64 };

// counters block in device memory (uint64 each)
enum { DC_RAYS = 0, DC_BOX = 1, DC_TRI = 2, DC_NAN = 3, DC_HITS = 4, DC_OVERFLOW = 5, DC_MAXQ = 6, DC_COUNT = 8 };

struct LaunchCfg {
    hipStream_t stream;
    int blocks;   // grid size of the grid-stride kernels
};

// ---- launchers (each returns the hipError_t of the launch) ----
hipError_t launch_raygen(const LaunchCfg& cfg, const CameraGpu& cam, const BatchGpu& batch, QueueSet out,
                         float4* sample_rad, int* qcount);
hipError_t launch_trace(const LaunchCfg& cfg, const SceneGpu& scene, int kd, int count_tests, int fast_slab,
                        const float4* ray_o, const float4* ray_d, float4* hit, const int* n_rays, int n_fixed,
                        unsigned long long* counters);
hipError_t launch_shade(const LaunchCfg& cfg, const SceneGpu& scene, const BatchGpu& batch, int level, int max_level,
                        QueueSet in, const float4* hit, QueueSet out, float4* sample_rad, int* qcount_in,
                        int* qcount_out, unsigned long long* counters);
hipError_t launch_accumulate(const LaunchCfg& cfg, const BatchGpu& batch, const float4* sample_rad, float* accum,
                             unsigned long long* counters);
hipError_t launch_resolve(const LaunchCfg& cfg, const float* accum, float* frame, int n_pixels, int total_samples);
hipError_t launch_clear(const LaunchCfg& cfg, float* buf, size_t n);
// unpacks rays6 (n x 6 floats) into queue form for the batch-trace entry
hipError_t launch_pack_rays(const LaunchCfg& cfg, const float* rays6, int n, float4* ray_o, float4* ray_d);
hipError_t launch_unpack_hits(const LaunchCfg& cfg, const float4* hit, int n, int* tri, float* t, float* bary2);

}  // namespace pt
