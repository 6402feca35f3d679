// shade_kernels.hip -- camera-ray generation, the per-level shading kernel, and the accumulate /
// resolve stages of the wavefront path tracer (gfx950).  The shading itself is in shade_device.h.
#include "shade_device.h"

namespace pt {

namespace {

using namespace shade_dev;

// ---- camera rays: Scene::generateRay (src/cs488.h:1825-1839) ----
// Camera samples are dealt to the workgroups in runs of 64 consecutive samples (one wave's worth,
// so a wave's primary rays stay coherent).  Round l deals runs [l*blocks, (l+1)*blocks), and within
// a round workgroup b takes position (b + l*deal_shift) mod blocks: every workgroup gets one run
// per round, and the shift keeps a workgroup's runs from lining up in one image column (a plain
// round-robin does when blocks*64 samples are a whole number of rows).  Every workgroup thus
// covers the whole image evenly and all of them finish a bounce level at about the same time.
__global__ __launch_bounds__(256) void raygen_kernel(CameraGpu cam, BatchGpu b, QueueSet out, float4* __restrict__ sample_rad,
                                                     int* __restrict__ qcount) {
    const int blk = blockIdx.x;
    const int runs = (b.n_slots + 63) / 64;  // runs in the batch; only the last one may be short
    const int full_rounds = runs / b.blocks, rest = runs % b.blocks;
    const int my_runs = full_rounds + (dealt_position(blk, full_rounds, b) < rest ? 1 : 0);
    const size_t region = (size_t)blk * b.cap_block;
    const vec3 eye = v3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const vec3 cu = v3(cam.u[0], cam.u[1], cam.u[2]), cv = v3(cam.v[0], cam.v[1], cam.v[2]), cw = v3(cam.w[0], cam.w[1], cam.w[2]);
    int written = 0;
    if (my_runs > 0) {
        // the batch's short run, if there is one, is the last run of the last round, hence this
        // workgroup's last run
        const int l = my_runs - 1;
        const int last_run = l * b.blocks + dealt_position(blk, l, b);
        const int last_len = last_run == runs - 1 ? b.n_slots - last_run * 64 : 64;
        written = (my_runs - 1) * 64 + last_len;
    }
    if (threadIdx.x == 0) {
        qcount[blk] = written;
        if (blk == 0) {
            *b.overflow = 0;
            out.ray_o[0] = make_float4(eye.x, eye.y, eye.z, 1.0f);  // the one origin record of all camera rays
        }
    }
    for (int j = threadIdx.x; j < written; j += blockDim.x) {
        const int l = j >> 6;
        const int slot = (l * b.blocks + dealt_position(blk, l, b)) * 64 + (j & 63);
        int pixel, k;
        slot_to_sample(b, slot, pixel, k);
        const int px = pixel % cam.width, py = pixel / cam.width;
        Sampler rng = make_sampler(b.keys[k], pixel, 1u << 8);
        const float r1 = rng.next();
        const float r2 = rng.next();
        const float U = (px + (r1 * 2 - 1)) / float(cam.width) - 0.5f;
        const float V = (py + (r2 * 2 - 1)) / float(cam.height) - 0.5f;
        const vec3 pix = eye + float(cam.aspect_film * U) * cu + float(cam.film * V) * cv - cam.dist * cw;
        const vec3 d = normalize_fast(pix - eye);
        // a camera ray is its direction: origin (the eye), miss scale (1), throughput (1, 1, 1) and sample slot
        // (a function of the position in the region) are known to the stages that follow
        out.ray_d[region + j] = make_float4(d.x, d.y, d.z, __uint_as_float((1u << 8) | rng.dim));
        if (b.fixed_radiance) {
            sample_rad[2 * (size_t)slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            sample_rad[2 * (size_t)slot + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            sample_rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
}

// The camera rays raygen_kernel left in the queue, put back into pixel order (for parity tests of the kernel
// itself): walks the same dealing as raygen_kernel and writes (eye, direction) of every sample of a batch with one
// sample per pixel to rays6[6 * (pixel - pixel0)].
__global__ __launch_bounds__(256) void gather_camera_rays_kernel(CameraGpu cam, BatchGpu b, const float4* __restrict__ ray_d,
                                                                 const int* __restrict__ qcount, float* __restrict__ rays6) {
    const int blk = blockIdx.x;
    const int n = qcount[blk];
    const size_t region = (size_t)blk * b.cap_block;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const int l = j >> 6;
        const int slot = (l * b.blocks + dealt_position(blk, l, b)) * 64 + (j & 63);
        int pixel, k;
        slot_to_sample(b, slot, pixel, k);
        const float4 d = ray_d[region + j];
        float* out = rays6 + 6 * (size_t)(pixel - b.pixel0);
        out[0] = cam.eye[0]; out[1] = cam.eye[1]; out[2] = cam.eye[2];
        out[3] = d.x; out[4] = d.y; out[5] = d.z;
    }
}

// Sampler keys of the batch's sample indices (they depend on the seed and the global sample index
// only, so they are hashed once per batch instead of once per path and bounce).
__global__ void sample_keys_kernel(uint64_t seed, int sample0, int chunk, uint64_t* __restrict__ keys) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < chunk) keys[k] = sample_key(seed, sample0 + k);
}

// ---- one bounce of shading for every path in the queue ----
#ifndef PT_SHADE_MIN_BLOCKS
#define PT_SHADE_MIN_BLOCKS 1   // (a build that holds shade_kernel to fewer registers, for measurements: make EXTRA=-DPT_SHADE_MIN_BLOCKS=6)
#endif
template <int KIND, bool CAMERA, int GEO>
__global__ __launch_bounds__(256, PT_SHADE_MIN_BLOCKS) void shade_kernel(SceneGpu sc, BatchGpu b, int level, int max_level, QueueSet in,
                                                    const float4* __restrict__ hit, QueueSet out,
                                                    float4* __restrict__ sample_rad, const int* __restrict__ qcount_in,
                                                    int* __restrict__ qcount_out) {
    extern __shared__ float4 geo[];
    __shared__ int s_out;
    const int blk = blockIdx.x;
    int n = qcount_in[blk];
    n = n < b.cap_block ? n : b.cap_block;
    if (n <= 0) {
        if (threadIdx.x == 0) qcount_out[blk] = 0;
        return;
    }
    const ShadeLds lds = stage_geometry(sc, geo, GEO);  // shade_region starts with a barrier
    const int produced = shade_region<KIND, CAMERA, GEO>(sc, b, level, max_level, in, hit, out, sample_rad, n, &s_out, lds);
    if (threadIdx.x == 0) qcount_out[blk] = produced;
}

// ---- per pixel: add the batch's samples to the accumulation buffer (src/cs488.h:1854-1858) ----
__global__ __launch_bounds__(256) void accumulate_kernel(BatchGpu b, const float4* __restrict__ sample_rad,
                                                         float* __restrict__ accum, unsigned long long* __restrict__ totals) {
    if (*b.overflow) return;  // incomplete batch: nothing of it may reach the image
    const int stride = gridDim.x * blockDim.x;
    unsigned int dropped = 0;
    for (int pl = blockIdx.x * blockDim.x + threadIdx.x; pl < b.n_pixels; pl += stride) {
        float* dst = accum + 3 * (size_t)(b.pixel0 + pl);
        float ax = dst[0], ay = dst[1], az = dst[2];
        const float4* src = sample_rad + (size_t)pl * b.chunk * (b.fixed_radiance ? 2 : 1);
        // four loads in flight per lane (one 64-byte sector), summed in sample order
        for (int k0 = 0; k0 < b.chunk; k0 += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (k0 + u >= b.chunk) v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                else if (b.fixed_radiance) v[u] = radiance_of_fixed(src[2 * (k0 + u)], src[2 * (k0 + u) + 1]);
                else v[u] = src[k0 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (k0 + u >= b.chunk) break;
                const float4 s = v[u];
                if (isnan(s.x) || isnan(s.y) || isnan(s.z)) {
                    dropped++;
                } else {
                    ax += s.x; ay += s.y; az += s.z;
                }
            }
        }
        dst[0] = ax; dst[1] = ay; dst[2] = az;
    }
    if (dropped) atomicAdd(&totals[DC_NAN], (unsigned long long)dropped);
}

// Per level (one workgroup each): total queue population of the batch = closest-hit queries of
// that level; the sum over levels goes to DC_RAYS, the largest level to DC_MAXQ.
// A batch whose overflow flag is set added nothing to the image and is rendered again in pieces,
// which are tallied themselves; it is left out here as in accumulate_kernel.
__global__ __launch_bounds__(256) void tally_kernel(const int* __restrict__ qcount, int blocks, int tail_level,
                                                    const int* __restrict__ overflow, unsigned long long* __restrict__ totals,
                                                    unsigned long long* __restrict__ level_rays) {
    __shared__ unsigned long long s_sum[256];
    const bool flagged = *overflow != 0;
    const int level = blockIdx.x;
    unsigned long long part = 0;
    for (int k = threadIdx.x; k < blocks; k += blockDim.x) part += (unsigned long long)qcount[(size_t)level * blocks + k];
    s_sum[threadIdx.x] = part;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    // the batch's population per level, for the host's scheduling (also for a flagged batch: the host reads these
    // words whatever became of the batch, and a flagged batch's counts are as good a profile as any)
    if (threadIdx.x == 0 && level_rays) level_rays[level] = s_sum[0];
    if (threadIdx.x == 0 && s_sum[0] != 0 && !flagged) {
        atomicAdd(&totals[DC_RAYS], s_sum[0]);
        if (level >= tail_level) atomicAdd(&totals[DC_TAIL_RAYS], s_sum[0]);
        atomicMax(&totals[DC_MAXQ], s_sum[0]);
    }
}

// frame = accumulation / totalSamples (src/cs488.h:1860)
__global__ void resolve_kernel(const float* __restrict__ accum, float* __restrict__ frame, size_t n, int total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float div = (float)total;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) frame[i] = accum[i] / div;
}

// dst += src (the root of a multi-GPU group folds a frame's reduced samples into its accumulation buffer)
__global__ void add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

__global__ void clear_kernel(float* buf, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) buf[i] = 0.0f;
}

// One wave that keeps its CU busy for a given time (100 MHz wall clock); the host uses pairs of
// these to find out whether two streams run side by side.  Bounded by an iteration count as well.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 24) && wall_clock64() - t0 < ticks; i++) {}
}

}  // namespace

hipError_t launch_spin(const LaunchCfg& cfg, int microseconds) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cfg.stream, (long long)microseconds * 100);
    return hipGetLastError();
}

hipError_t launch_raygen(const LaunchCfg& cfg, const CameraGpu& cam, const BatchGpu& batch, QueueSet out,
                         float4* sample_rad, int* qcount) {
    hipLaunchKernelGGL(sample_keys_kernel, dim3((batch.chunk + 255) / 256), dim3(256), 0, cfg.stream, batch.seed, batch.sample0,
                       batch.chunk, const_cast<uint64_t*>(batch.keys));
    hipLaunchKernelGGL(raygen_kernel, dim3(batch.blocks), dim3(256), 0, cfg.stream, cam, batch, out, sample_rad, qcount);
    return hipGetLastError();
}

hipError_t launch_gather_camera_rays(const LaunchCfg& cfg, const CameraGpu& cam, const BatchGpu& batch, const float4* ray_d,
                                     const int* qcount, float* rays6) {
    hipLaunchKernelGGL(gather_camera_rays_kernel, dim3(batch.blocks), dim3(256), 0, cfg.stream, cam, batch, ray_d, qcount, rays6);
    return hipGetLastError();
}

hipError_t launch_shade(const LaunchCfg& cfg, const SceneGpu& scene, const BatchGpu& batch, int level, int max_level,
                        QueueSet in, const float4* hit, QueueSet out, float4* sample_rad, const int* qcount_in,
                        int* qcount_out) {
    const size_t geo_bytes = geometry_lds_bytes(scene);
    const int geo_kind = geometry_kind(scene);
#define PT_SHADE_GEO(KIND, CAMERA, GEO)                                                                                     \
    hipLaunchKernelGGL((shade_kernel<KIND, CAMERA, GEO>), dim3(batch.blocks), dim3(256), geo_bytes, cfg.stream, scene, batch, level, \
                       max_level, in, hit, out, sample_rad, qcount_in, qcount_out)
#define PT_SHADE(KIND, CAMERA)                                                                                              \
    do {                                                                                                                    \
        if (geo_kind == GEO_ALL) PT_SHADE_GEO(KIND, CAMERA, GEO_ALL);                                                       \
        else PT_SHADE_GEO(KIND, CAMERA, GEO_NONE);                                                                          \
    } while (0)
    const bool camera = in.beta == nullptr;  // camera rays: directions only
    if (scene.shading_kind == SHADE_PLAIN) { if (camera) PT_SHADE(SHADE_PLAIN, true); else PT_SHADE(SHADE_PLAIN, false); }
    else if (scene.shading_kind == SHADE_LAMBERTIAN) { if (camera) PT_SHADE(SHADE_LAMBERTIAN, true); else PT_SHADE(SHADE_LAMBERTIAN, false); }
    else if (scene.shading_kind == SHADE_GLASS) { if (camera) PT_SHADE(SHADE_GLASS, true); else PT_SHADE(SHADE_GLASS, false); }
    else if (scene.shading_kind == SHADE_METAL) { if (camera) PT_SHADE(SHADE_METAL, true); else PT_SHADE(SHADE_METAL, false); }
    else { if (camera) PT_SHADE(SHADE_ALL, true); else PT_SHADE(SHADE_ALL, false); }
#undef PT_SHADE
#undef PT_SHADE_GEO
    return hipGetLastError();
}

hipError_t launch_accumulate(const LaunchCfg& cfg, const BatchGpu& batch, const float4* sample_rad, float* accum,
                             unsigned long long* totals) {
    hipLaunchKernelGGL(accumulate_kernel, dim3(cfg.blocks), dim3(256), 0, cfg.stream, batch, sample_rad, accum, totals);
    return hipGetLastError();
}

hipError_t launch_tally(const LaunchCfg& cfg, const int* qcount, int levels, int blocks, int tail_level, const int* overflow,
                        unsigned long long* totals, unsigned long long* level_rays) {
    hipLaunchKernelGGL(tally_kernel, dim3(levels), dim3(256), 0, cfg.stream, qcount, blocks, tail_level, overflow, totals, level_rays);
    return hipGetLastError();
}

hipError_t launch_resolve(const LaunchCfg& cfg, const float* accum, float* frame, int n_pixels, int total_samples) {
    hipLaunchKernelGGL(resolve_kernel, dim3(cfg.blocks), dim3(256), 0, cfg.stream, accum, frame, (size_t)n_pixels * 3, total_samples);
    return hipGetLastError();
}

hipError_t launch_add(const LaunchCfg& cfg, float* dst, const float* src, size_t n) {
    hipLaunchKernelGGL(add_kernel, dim3(cfg.blocks), dim3(256), 0, cfg.stream, dst, src, n);
    return hipGetLastError();
}

hipError_t launch_clear(const LaunchCfg& cfg, float* buf, size_t n) {
    hipLaunchKernelGGL(clear_kernel, dim3(cfg.blocks), dim3(256), 0, cfg.stream, buf, n);
    return hipGetLastError();
}

}  // namespace pt
