// tail_kernels.hip -- one launch that finishes the deep bounce levels of a batch (gfx950).
//
// A workgroup owns its region of the ray queues: it consumes what it produced one level earlier,
// so nothing couples the workgroups between levels and a workgroup can run trace -> shade ->
// trace -> ... for all remaining levels by itself, with workgroup barriers only.  From the level
// at which Russian roulette has thinned the paths out (src/cs488.h:2153-2160), a launch per level
// and stage costs more in launch gaps and nearly empty grids than the work itself; this kernel
// replaces those launches.  It walks and shades with the same device code as the per-level
// kernels (trace_device.h, shade_device.h), so results are identical.
#include "shade_device.h"
#include "trace_device.h"

namespace pt {

namespace {

using namespace trace_dev;
using namespace shade_dev;

// Six workgroups per CU (80 vector registers): left to itself the compiler takes 83-113, four or five waves per SIMD.
// Against that, same call, alternating: small frames (512 x 384 x 10, where the fused launch takes over after the first
// levels) cornellbox +1 %, cornellbox-glass +4 % (kd-tree +2 %), teapot +1 % (kd-tree +5 %); at 1024 x 1024 bunny and
// teapot kd-trees +4 %, testObj +2 %, the courtyard's kd-tree +2 %, its BVH -0.6 %, cornellbox +-0.3 %.  Eight (64
// registers) spills in the heavier shading kinds: cornellbox-glass -6 %.  (make EXTRA=-DPT_TAIL_MIN_BLOCKS=n for another.)
#ifndef PT_TAIL_MIN_BLOCKS
#define PT_TAIL_MIN_BLOCKS 6
#endif
template <bool KD, int SLAB, bool ALL_LDS, int KIND, int GEO>
__global__ __launch_bounds__(256, PT_TAIL_MIN_BLOCKS) void tail_kernel(SceneGpu sc, BatchGpu b, int level0, int levels, int max_level, QueueSet qa,
                                                   QueueSet qb, float4* __restrict__ hit, float4* __restrict__ sample_rad,
                                                   int* __restrict__ qcount, unsigned long long* __restrict__ block_stats,
                                                   int leave_k) {
    extern __shared__ float4 lds_nodes[];
    __shared__ __attribute__((aligned(16))) unsigned int s_stat[PT_STAT_WORDS];
    __shared__ __attribute__((aligned(16))) int s_stack[LdsStack<KD>::ROWS * 256];
    __shared__ __attribute__((aligned(16))) float s_stack_t[KD ? LdsStack<KD>::ROWS * 256 : 4];
    __shared__ int s_out;
    const int blk = blockIdx.x;
    const int nb = b.blocks;
    int n = qcount[(size_t)level0 * nb + blk];
    n = n < b.cap_block ? n : b.cap_block;
    const size_t region = (size_t)blk * b.cap_block;
    // the shading data that is staged (shade_device.h: GEO) sits behind the staged part of the tree
    float4* const geo = lds_nodes + (dynamic_lds_bytes(sc) + 15) / 16;
    if (n > 0) stage_tree<ALL_LDS>(sc, lds_nodes);  // n is the same for every thread of the workgroup
    ShadeLds shade_lds = ShadeLds();
    if (n > 0) shade_lds = stage_geometry(sc, geo, GEO);
    TraceLds<KD> lds;
    lds.nodes = lds_nodes; lds.stat = s_stat; lds.stack = s_stack; lds.stack_t = s_stack_t; lds.count = nullptr;
    const RayPool no_pool = {};
    for (int level = level0; level < levels; level++) {
        if (n > 0) {
            const QueueSet& in = (level & 1) ? qb : qa;
            const QueueSet& out = (level & 1) ? qa : qb;
            trace_region<KD, false, SLAB, true, ALL_LDS, false>(sc, in.ray_o, in.ray_d, hit, n, region, block_stats, leave_k, lds, no_pool);
            n = shade_region<KIND, false, GEO>(sc, b, level, max_level, in, hit, out, sample_rad, n, &s_out, shade_lds);
            __syncthreads();  // s_out is rewritten by the next level's shading
        }
        if (threadIdx.x == 0) qcount[(size_t)(level + 1) * nb + blk] = n;  // the tally reads every level
    }
}

template <bool KD, int SLAB>
void launch_tail_variant(const LaunchCfg& cfg, const SceneGpu& scene, const BatchGpu& batch, int level0, int levels, int max_level,
                         QueueSet qa, QueueSet qb, float4* hit, float4* sample_rad, int* qcount, unsigned long long* block_stats,
                         int leave_k) {
    const size_t lds = ((dynamic_lds_bytes(scene) + 15) / 16) * 16 + geometry_lds_bytes(scene);
    const int geo_kind = geometry_kind(scene);
#define PT_TAIL(ALL, KIND)                                                                                             \
    do {                                                                                                               \
        if (geo_kind == GEO_ALL)                                                                                       \
            hipLaunchKernelGGL((tail_kernel<KD, SLAB, ALL, KIND, GEO_ALL>), dim3(batch.blocks), dim3(256), lds, cfg.stream, scene, batch, \
                               level0, levels, max_level, qa, qb, hit, sample_rad, qcount, block_stats, leave_k);      \
        else                                                                                                           \
            hipLaunchKernelGGL((tail_kernel<KD, SLAB, ALL, KIND, GEO_NONE>), dim3(batch.blocks), dim3(256), lds, cfg.stream, scene, batch, \
                               level0, levels, max_level, qa, qb, hit, sample_rad, qcount, block_stats, leave_k);      \
    } while (0)
#define PT_TAIL_KIND(ALL)                                                            \
    do {                                                                             \
        if (scene.shading_kind == SHADE_PLAIN) PT_TAIL(ALL, SHADE_PLAIN);            \
        else if (scene.shading_kind == SHADE_LAMBERTIAN) PT_TAIL(ALL, SHADE_LAMBERTIAN); \
        else if (scene.shading_kind == SHADE_GLASS) PT_TAIL(ALL, SHADE_GLASS);       \
        else if (scene.shading_kind == SHADE_METAL) PT_TAIL(ALL, SHADE_METAL);       \
        else PT_TAIL(ALL, SHADE_ALL);                                                \
    } while (0)
    if (fits_lds(scene)) PT_TAIL_KIND(true); else PT_TAIL_KIND(false);
#undef PT_TAIL_KIND
#undef PT_TAIL
}

}  // namespace

// Runs levels [level0, levels) of the batch; qa is the queue set that even levels read.  `mode`
// as in launch_trace (the tail always refills dynamically and never counts tests).
hipError_t launch_tail(const LaunchCfg& cfg, const SceneGpu& scene, const BatchGpu& batch, int kd, int mode, int level0, int levels,
                       int max_level, QueueSet qa, QueueSet qb, float4* hit, float4* sample_rad, int* qcount,
                       unsigned long long* block_stats) {
    const bool recip = (mode & 1) != 0;
    // the refill threshold rides above leave_k; unset, a scene walked entirely in LDS waits for half a wave
    // (short walks: fuller refills pay), any other for eight lanes (long walks: idle lanes cost more)
    const int asked = (mode >> 8) & 127;
    const int refill = asked ? (asked > 64 ? 64 : asked) : (fits_lds(scene) ? 32 : 8);  // a wave never has more than 64 idle lanes
    // node-phase leave rule: 0 = per scene (4 in LDS, else 1), 15 = stay until nobody walks
    const int k_bits = (mode >> 4) & 15;
    const int k = k_bits == 0 ? (fits_lds(scene) ? 4 : 1) : (k_bits == 15 ? 65 : k_bits);  // 65: with at least one lane walking the product exceeds any count of waiting lanes
    const int leave_k = k | (refill << 8);
#define PT_ARGS cfg, scene, batch, level0, levels, max_level, qa, qb, hit, sample_rad, qcount, block_stats, leave_k
    if (kd) {
        if (recip) launch_tail_variant<true, SLAB_RECIP>(PT_ARGS); else launch_tail_variant<true, SLAB_DIVIDE>(PT_ARGS);
    } else {
        if (recip) launch_tail_variant<false, SLAB_RECIP>(PT_ARGS); else launch_tail_variant<false, SLAB_DIVIDE>(PT_ARGS);
    }
#undef PT_ARGS
    return hipGetLastError();
}

}  // namespace pt
