// trace_kernels.hip -- the per-level closest-hit kernel, the ray-batch entry kernels and their
// launchers (gfx950).  The walk itself is in trace_device.h.
#include "trace_device.h"
#include "trace_device_two.h"

#include <mutex>

// Scratch of pt_trace_batch_static (the entry point that allocates nothing): room for the packed
// origins, the packed directions and the hit records of TRACE_STATIC_MAX_RAYS rays, 6 MiB per GPU,
// held until the process ends.  In host code the array's name is not a device address: the address
// on a device comes from hipGetSymbolAddress with that device current (trace_static_scratch below),
// and only that pointer is handed to kernels.
static __device__ __align__(256) float4 scratch[3 * pt::TRACE_STATIC_MAX_RAYS];

#ifdef PT_TIMELINE
__device__ unsigned long long g_timeline[4096 * 8];
extern "C" int pt_debug_timeline(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(g_timeline)); }
#endif

namespace pt {

namespace {

using namespace trace_dev;

enum { SCRATCH_DEVICES = 64 };
float4* g_scratch_ptr[SCRATCH_DEVICES] = {};  // per device: each GPU holds its own copy at its own address
std::mutex g_scratch_mutex;

// 256-lane workgroups that draw the launch's rays from POOL_GROUPS shared pools (RayPool in trace_device.h): the
// launch has `n_regions` queue regions of `cap_block` entries, whose ray counts are read from device memory
// (written by the previous stage, so the host never waits for a count) or follow from n_fixed.
template <bool KD, bool COUNT, int SLAB, bool DYNAMIC, bool ALL_LDS, bool POOL>
__global__ __launch_bounds__(256, 6) void trace_kernel(SceneGpu sc, const float4* __restrict__ ray_o,
                                                    const float4* __restrict__ ray_d, float4* __restrict__ hit,
                                                    const int* __restrict__ n_ptr, int n_fixed, int cap_block, int n_regions,
                                                    int* __restrict__ cursors, unsigned long long* __restrict__ block_stats,
                                                    int leave_k) {
    extern __shared__ float4 lds_nodes[];
    __shared__ __attribute__((aligned(16))) unsigned int s_stat[PT_STAT_WORDS];
    __shared__ __attribute__((aligned(16))) int s_stack[LdsStack<KD>::ROWS * 256];
    __shared__ __attribute__((aligned(16))) float s_stack_t[KD ? LdsStack<KD>::ROWS * 256 : 4];
    __shared__ int s_count[POOL ? POOL_REGIONS : 1];
    const int blk = blockIdx.x;
#ifdef PT_TIMELINE
    const unsigned long long tl0 = wall_clock64();
#endif
    RayPool pool;
    const int group = blk % POOL_GROUPS;
    const int per_group = (n_regions + POOL_GROUPS - 1) / POOL_GROUPS;
    pool.cursor = cursors + group * TRACE_POOL_STRIDE;
    pool.counts = n_ptr;
    pool.n_fixed = n_fixed;
    pool.cap_block = cap_block;
    pool.first = group * per_group;
    pool.regions = n_regions - pool.first < per_group ? n_regions - pool.first : per_group;
    int n = 0;  // POOL = false: the workgroup traces the rays of region blk itself
    if (!POOL) {
        if (n_ptr) {
            n = n_ptr[blk];
            n = n < cap_block ? n : cap_block;
        } else {
            const long long rest = (long long)n_fixed - (long long)blk * cap_block;
            n = rest <= 0 ? 0 : (rest < cap_block ? (int)rest : cap_block);
        }
    }
    if (POOL ? pool.regions <= 0 : n <= 0) return;
    stage_tree<ALL_LDS>(sc, lds_nodes);
    TraceLds<KD> lds;
    lds.nodes = lds_nodes; lds.stat = s_stat; lds.stack = s_stack; lds.stack_t = s_stack_t; lds.count = s_count;
    trace_region<KD, COUNT, SLAB, DYNAMIC, ALL_LDS, POOL>(sc, ray_o, ray_d, hit, n, (size_t)blk * cap_block, block_stats, leave_k, lds, pool);
#ifdef PT_TIMELINE
    if (threadIdx.x == 0 && blk < 4096) {
        unsigned long long* tl = g_timeline + 8 * (size_t)blk;
        tl[0] = tl0; tl[1] = wall_clock64();
        for (int k = 0; k < 4; k++) tl[2 + k] = s_stat[8 + k];
        tl[6] = 1;
        tl[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

// The same kernel for a BVH that is walked in memory with 192-byte node records (trace_region's WIDE form): 48
// instead of 16 registers of node data per lane, so it is held to five waves per SIMD, which is what the LDS of the
// stacks and the staged top of the tree admits anyway.
template <bool POOL>
__global__ __launch_bounds__(256, 5) void trace_kernel_wide(SceneGpu sc, const float4* __restrict__ ray_o,
                                                         const float4* __restrict__ ray_d, float4* __restrict__ hit,
                                                         const int* __restrict__ n_ptr, int n_fixed, int cap_block, int n_regions,
                                                         int* __restrict__ cursors, unsigned long long* __restrict__ block_stats,
                                                         int leave_k) {
    extern __shared__ float4 lds_nodes[];
    __shared__ __attribute__((aligned(16))) unsigned int s_stat[PT_STAT_WORDS];
    __shared__ __attribute__((aligned(16))) int s_stack[LdsStack<false>::ROWS * 256];
    __shared__ __attribute__((aligned(16))) float s_stack_t[4];
    __shared__ int s_count[POOL ? POOL_REGIONS : 1];
    const int blk = blockIdx.x;
    RayPool pool;
    const int group = blk % POOL_GROUPS;
    const int per_group = (n_regions + POOL_GROUPS - 1) / POOL_GROUPS;
    pool.cursor = cursors + group * TRACE_POOL_STRIDE;
    pool.counts = n_ptr;
    pool.n_fixed = n_fixed;
    pool.cap_block = cap_block;
    pool.first = group * per_group;
    pool.regions = n_regions - pool.first < per_group ? n_regions - pool.first : per_group;
    int n = 0;
    if (!POOL) {
        if (n_ptr) {
            n = n_ptr[blk];
            n = n < cap_block ? n : cap_block;
        } else {
            const long long rest = (long long)n_fixed - (long long)blk * cap_block;
            n = rest <= 0 ? 0 : (rest < cap_block ? (int)rest : cap_block);
        }
    }
    if (POOL ? pool.regions <= 0 : n <= 0) return;
    stage_tree<false>(sc, lds_nodes);
    TraceLds<false> lds;
    lds.nodes = lds_nodes; lds.stat = s_stat; lds.stack = s_stack; lds.stack_t = s_stack_t; lds.count = s_count;
    trace_region<false, false, SLAB_RECIP, true, false, POOL, true>(sc, ray_o, ray_d, hit, n, (size_t)blk * cap_block, block_stats, leave_k, lds, pool);
}

// Two rays per lane (trace_region_two): every workgroup traces its own region.  The stacks have fewer LDS rows per ray
// than trace_kernel's, so that a CU holds as many workgroups of 512 rays in flight as the registers admit.
template <bool KD, bool ALL_LDS>
__global__ __launch_bounds__(256, 4) void trace_kernel_two(SceneGpu sc, const float4* __restrict__ ray_o,
                                                        const float4* __restrict__ ray_d, float4* __restrict__ hit,
                                                        const int* __restrict__ n_ptr, int n_fixed, int cap_block,
                                                        unsigned long long* __restrict__ block_stats, int leave_k) {
    extern __shared__ float4 lds_nodes[];
    __shared__ __attribute__((aligned(16))) unsigned int s_stat[PT_STAT_WORDS];
    __shared__ __attribute__((aligned(16))) int s_stack[2 * LdsStackTwo<KD>::ROWS * 256];
    __shared__ __attribute__((aligned(16))) float s_stack_t[KD ? 2 * LdsStackTwo<KD>::ROWS * 256 : 4];
    const int blk = blockIdx.x;
    int n = 0;
    if (n_ptr) {
        n = n_ptr[blk];
        n = n < cap_block ? n : cap_block;
    } else {
        const long long rest = (long long)n_fixed - (long long)blk * cap_block;
        n = rest <= 0 ? 0 : (rest < cap_block ? (int)rest : cap_block);
    }
    if (n <= 0) return;
    if (ALL_LDS) stage_tree<true>(sc, lds_nodes);
    trace_region_two<KD, ALL_LDS>(sc, ray_o, ray_d, hit, n, (size_t)blk * cap_block, block_stats, leave_k, lds_nodes, s_stat, s_stack, s_stack_t);
}

// Does `mode` (PT_OPT_TRAVERSAL) ask for two rays per lane and can this launch use that form?
inline bool two_applies(const SceneGpu& scene, int mode, bool count) {
    // (its node loads address the node array through a 32-bit byte offset)
    return (mode & 2) != 0 && !count && (mode & 1) != 0 && (mode & 4) == 0 && (mode & 0x8000) == 0 && scene.n_nodes < (1 << 25);
}

// Does `mode` (PT_OPT_TRAVERSAL) ask for the 192-byte node records and can this launch use them?
inline bool wide_applies(const SceneGpu& scene, int mode, bool kd, bool count) {
    return (mode & 0x8000) != 0 && !kd && !count && (mode & 1) != 0 && (mode & 4) == 0 && scene.wide != nullptr && !fits_lds(scene);
}

__global__ void pack_rays_kernel(const float* __restrict__ rays6, int n, float4* ray_o, float4* ray_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float* r = rays6 + 6 * (size_t)i;
        ray_o[i] = make_float4(r[0], r[1], r[2], 1.0f);
        ray_d[i] = make_float4(r[3], r[4], r[5], 0.0f);
    }
}

__global__ void unpack_hits_kernel(const float4* __restrict__ hit, int n, int* tri, float* t, float* bary2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float4 h = hit[i];
        tri[i] = __float_as_int(h.x);
        t[i] = h.y;
        const bool found = __float_as_int(h.x) >= 0;  // a miss record holds the ray's miss scale in place of beta
        if (bary2) { bary2[2 * i] = found ? h.z : 0.0f; bary2[2 * i + 1] = found ? h.w : 0.0f; }
    }
}

template <bool KD, bool COUNT, int SLAB, bool DYNAMIC, bool POOL>
void launch_variant(const LaunchCfg& cfg, const SceneGpu& scene, const float4* ray_o, const float4* ray_d, float4* hit,
                    const int* n_rays, int n_fixed, int grid_blocks, int cap_block, int* cursors, unsigned long long* block_stats,
                    int leave_k) {
    const size_t lds = dynamic_lds_bytes(scene);
    if (fits_lds(scene))
        hipLaunchKernelGGL((trace_kernel<KD, COUNT, SLAB, DYNAMIC, true, POOL>), dim3(grid_blocks), dim3(256), lds, cfg.stream, scene,
                           ray_o, ray_d, hit, n_rays, n_fixed, cap_block, grid_blocks, cursors, block_stats, leave_k);
    else
        hipLaunchKernelGGL((trace_kernel<KD, COUNT, SLAB, DYNAMIC, false, POOL>), dim3(grid_blocks), dim3(256), lds, cfg.stream, scene,
                           ray_o, ray_d, hit, n_rays, n_fixed, cap_block, grid_blocks, cursors, block_stats, leave_k);
}

template <bool KD, bool COUNT>
void launch_by_mode(int mode, const LaunchCfg& cfg, const SceneGpu& scene, const float4* ray_o, const float4* ray_d,
                    float4* hit, const int* n_rays, int n_fixed, int grid_blocks, int cap_block, int* cursors,
                    unsigned long long* block_stats) {
    // mode: bit 0 = reciprocal-form slab test, bit 2 = refill only when the whole wave is idle
    // (measurement), bit 3 = every workgroup traces its own region instead of drawing from the shared pools, bits 4..7 = leave_k of the node phase, bits 8..14 = refill threshold (0 = chosen per scene)
    const bool recip = (mode & 1) != 0;
    const bool dynamic = (mode & 4) == 0;
    // the refill threshold rides above leave_k; unset, a scene walked entirely in LDS waits for half a wave
    // (short walks: fuller refills pay), any other for eight lanes (long walks: idle lanes cost more)
    const int asked = (mode >> 8) & 127;
    const int refill = asked ? (asked > 64 ? 64 : asked) : (fits_lds(scene) ? 32 : 8);  // a wave never has more than 64 idle lanes
    // node-phase leave rule: 0 = per scene (4 in LDS, else 1), 15 = stay until nobody walks
    const int k_bits = (mode >> 4) & 15;
    const int k = k_bits == 0 ? (fits_lds(scene) ? 4 : 1) : (k_bits == 15 ? 65 : k_bits);  // 65: with at least one lane walking the product exceeds any count of waiting lanes
    const int leave_k = k | (refill << 8);
#define PT_ARGS cfg, scene, ray_o, ray_d, hit, n_rays, n_fixed, grid_blocks, cap_block, cursors, block_stats, leave_k
    if (two_applies(scene, mode, COUNT)) {
        if (fits_lds(scene))
            hipLaunchKernelGGL((trace_kernel_two<KD, true>), dim3(grid_blocks), dim3(256), dynamic_lds_bytes(scene), cfg.stream, scene, ray_o,
                               ray_d, hit, n_rays, n_fixed, cap_block, block_stats, leave_k);
        else
            hipLaunchKernelGGL((trace_kernel_two<KD, false>), dim3(grid_blocks), dim3(256), 0, cfg.stream, scene, ray_o, ray_d,
                               hit, n_rays, n_fixed, cap_block, block_stats, leave_k);
    } else if (wide_applies(scene, mode, KD, COUNT)) {
        const size_t lds = dynamic_lds_bytes(scene);
        if ((mode & 8) != 0 || cap_block > POOL_MAX_REGION)
            hipLaunchKernelGGL((trace_kernel_wide<false>), dim3(grid_blocks), dim3(256), lds, cfg.stream, scene, ray_o, ray_d, hit, n_rays,
                               n_fixed, cap_block, grid_blocks, cursors, block_stats, leave_k);
        else
            hipLaunchKernelGGL((trace_kernel_wide<true>), dim3(grid_blocks), dim3(256), lds, cfg.stream, scene, ray_o, ray_d, hit, n_rays,
                               n_fixed, cap_block, grid_blocks, cursors, block_stats, leave_k);
    } else if ((mode & 8) != 0 || cap_block > POOL_MAX_REGION) {  // every workgroup traces its own region (always with the dynamic refill)
        if (recip) launch_variant<KD, COUNT, SLAB_RECIP, true, false>(PT_ARGS);
        else launch_variant<KD, COUNT, SLAB_DIVIDE, true, false>(PT_ARGS);
    } else if (dynamic) {
        if (recip) launch_variant<KD, COUNT, SLAB_RECIP, true, true>(PT_ARGS);
        else launch_variant<KD, COUNT, SLAB_DIVIDE, true, true>(PT_ARGS);
    } else {
        if (recip) launch_variant<KD, COUNT, SLAB_RECIP, false, true>(PT_ARGS);
        else launch_variant<KD, COUNT, SLAB_DIVIDE, false, true>(PT_ARGS);
    }
#undef PT_ARGS
}

}  // namespace

hipError_t launch_trace(const LaunchCfg& cfg, const SceneGpu& scene, int kd, int count_tests, int mode,
                        const float4* ray_o, const float4* ray_d, float4* hit, const int* n_rays, int n_fixed,
                        int grid_blocks, int cap_block, int* cursors, unsigned long long* block_stats) {
    static_assert((int)POOL_GROUPS == (int)TRACE_POOL_CURSORS, "one cursor per pool");
    if (grid_blocks > POOL_GROUPS * POOL_REGIONS || !cursors) return hipErrorInvalidValue;
    if (kd) {
        if (count_tests) launch_by_mode<true, true>(mode, cfg, scene, ray_o, ray_d, hit, n_rays, n_fixed, grid_blocks, cap_block, cursors, block_stats);
        else launch_by_mode<true, false>(mode, cfg, scene, ray_o, ray_d, hit, n_rays, n_fixed, grid_blocks, cap_block, cursors, block_stats);
    } else {
        if (count_tests) launch_by_mode<false, true>(mode, cfg, scene, ray_o, ray_d, hit, n_rays, n_fixed, grid_blocks, cap_block, cursors, block_stats);
        else launch_by_mode<false, false>(mode, cfg, scene, ray_o, ray_d, hit, n_rays, n_fixed, grid_blocks, cap_block, cursors, block_stats);
    }
    return hipGetLastError();
}

namespace {
template <bool KD, int SLAB>
int resident_of(const SceneGpu& scene) {
    int n = 0;
    const size_t lds = dynamic_lds_bytes(scene);
    const hipError_t e = fits_lds(scene)
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<KD, false, SLAB, true, true, true>, 256, lds)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<KD, false, SLAB, true, false, true>, 256, lds);
    return e == hipSuccess ? n : 0;
}
}  // namespace

// Workgroups of the trace kernel one CU holds at a time for this scene (registers and LDS).  A batch
// split over exactly that many regions per CU runs every trace launch as one resident round; with
// more regions than that the last, partly filled round costs as much as a full one.
int trace_resident_blocks_per_cu(const SceneGpu& scene, int kd, int mode) {
    const bool recip = (mode & 1) != 0;
    if (two_applies(scene, mode, false)) {
        int w = 0;
        const bool lds = fits_lds(scene);
        const size_t bytes = lds ? dynamic_lds_bytes(scene) : 0;
        const hipError_t e =
            kd ? (lds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, trace_kernel_two<true, true>, 256, bytes)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, trace_kernel_two<true, false>, 256, bytes))
               : (lds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, trace_kernel_two<false, true>, 256, bytes)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, trace_kernel_two<false, false>, 256, bytes));
        if (e == hipSuccess && w >= 1) return w > 8 ? 8 : w;
    }
    if (wide_applies(scene, mode, kd != 0, false)) {
        int w = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, trace_kernel_wide<true>, 256, dynamic_lds_bytes(scene)) == hipSuccess && w >= 1)
            return w > 8 ? 8 : w;
    }
    int n = kd ? (recip ? resident_of<true, SLAB_RECIP>(scene) : resident_of<true, SLAB_DIVIDE>(scene))
               : (recip ? resident_of<false, SLAB_RECIP>(scene) : resident_of<false, SLAB_DIVIDE>(scene));
    return n < 1 ? 8 : (n > 8 ? 8 : n);
}

bool trace_kernel_fits(const SceneGpu& scene, int kd, int mode) {
    // static LDS of the kernel (stacks, counters) + the dynamic stage must stay within what a workgroup can have
    // (160 KiB on gfx950); the occupancy query says 0 (or fails) beyond it
    const bool recip = (mode & 1) != 0;
    const int n = kd ? (recip ? resident_of<true, SLAB_RECIP>(scene) : resident_of<true, SLAB_DIVIDE>(scene))
                     : (recip ? resident_of<false, SLAB_RECIP>(scene) : resident_of<false, SLAB_DIVIDE>(scene));
    return n >= 1;
}

int trace_scene_in_lds(const SceneGpu& scene) { return fits_lds(scene) ? 1 : 0; }

// Device address of `scratch` on `device`, which must be the current device.  Looked up once per
// device (pt_create does it, so the entry point itself only reads the table).
hipError_t trace_static_scratch(int device, float4** out) {
    *out = nullptr;
    if (device < 0 || device >= SCRATCH_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<float4*>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

hipError_t launch_pack_rays(const LaunchCfg& cfg, const float* rays6, int n, float4* ray_o, float4* ray_d) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, rays6, n, ray_o, ray_d);
    return hipGetLastError();
}

hipError_t launch_unpack_hits(const LaunchCfg& cfg, const float4* hit, int n, int* tri, float* t, float* bary2) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, cfg.stream, hit, n, tri, t, bary2);
    return hipGetLastError();
}

}  // namespace pt
