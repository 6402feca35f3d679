// trace_kernels.hip -- closest-hit stage of the wavefront path tracer (gfx950).
//
// One lane walks one ray through the flattened BVH or kd-tree and keeps the nearest triangle.
// The walk reproduces the reference's recursion exactly, so that the winning triangle and its
// distance are the reference's own:
//   * slab test per child with true divisions, NaN-transparent comparisons and the entry
//     distance as result (AABB::intersect, src/cs488.h:530-567);
//   * a child is entered only if its entry distance is below the best hit so far; with both
//     children eligible the one with the smaller entry distance goes first and on a tie the
//     RIGHT one (src/cs488.h:1687-1697, 1360-1370);
//   * BVH: the deferred child is entered unconditionally (src/cs488.h:1692-1696);
//     kd-tree: it is entered only if its entry distance is still below the best hit
//     (src/cs488.h:1366,1369);
//   * a leaf tests its triangles in stored order and a hit replaces the best only when strictly
//     nearer (src/cs488.h:1677-1681, 1350-1354);
//   * the triangle test is the Cramer's-rule form with inclusive barycentric and distance
//     bounds (TriangleMesh::raytraceTriangle, src/cs488.h:646-681).
// This file is compiled with -ffp-contract=off: the x86 reference evaluates every product and
// sum separately, and a fused multiply-add here would change low-order bits of t.
//
// Launch shape: one 256-lane workgroup per queue region; the region's length is read from device
// memory (written by the previous stage), so the host never waits for a count.
// The first `lds_nodes` tree nodes (the top levels; nodes are numbered breadth-first) are copied
// into LDS by every workgroup that has work.
#include <float.h>

#include "device_types.h"
#include "pt_math.h"

namespace pt {

namespace {

__device__ __forceinline__ bool slab_test(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                          vec3 o, vec3 d, float& t_entry) {
    float tx1 = (lox - o.x) / d.x, ty1 = (loy - o.y) / d.y, tz1 = (loz - o.z) / d.z;
    float tx2 = (hix - o.x) / d.x, ty2 = (hiy - o.y) / d.y, tz2 = (hiz - o.z) / d.z;
    if (tx1 > tx2) { float s = tx1; tx1 = tx2; tx2 = s; }
    if (ty1 > ty2) { float s = ty1; ty1 = ty2; ty2 = s; }
    if (tz1 > tz2) { float s = tz1; tz1 = tz2; tz2 = s; }
    float t1 = tx1; if (t1 < ty1) t1 = ty1; if (t1 < tz1) t1 = tz1;
    float t2 = tx2; if (t2 > ty2) t2 = ty2; if (t2 > tz2) t2 = tz2;
    t_entry = t1;
    if (t1 > t2) return false;
    if ((t1 < 0.0f) && (t2 < 0.0f)) return false;
    return true;
}

struct Best {
    float t, beta, gamma;
    int tri;
};

__device__ __forceinline__ void triangle_test(const SceneGpu& sc, int tri, vec3 o, vec3 d, Best& best) {
    const float4 A = sc.tri_pos[3 * tri], B = sc.tri_pos[3 * tri + 1], C = sc.tri_pos[3 * tri + 2];
    const vec3 p0 = v3(A.x, A.y, A.z);
    const vec3 a = p0 - v3(B.x, B.y, B.z);
    const vec3 b = p0 - v3(C.x, C.y, C.z);
    const vec3 e = p0 - o;
    const vec3 ab = cross(a, b);
    const float D = dot(ab, d);
    if (D != 0) {
        const float Da = dot(cross(e, b), d);
        const float Db = dot(cross(a, e), d);
        const float Dc = dot(ab, e);
        const float beta = Da / D;
        const float gamma = Db / D;
        const float alpha = 1 - beta - gamma;
        const float t = Dc / D;
        if (0 <= alpha && alpha <= 1 && 0 <= beta && beta <= 1 && 0 <= gamma && gamma <= 1 && 0.0f <= t && t <= FLT_MAX) {
            if (t < best.t) { best.t = t; best.beta = beta; best.gamma = gamma; best.tri = tri; }
        }
    }
}

template <bool KD, bool COUNT>
__global__ __launch_bounds__(256) void trace_kernel(SceneGpu sc, const float4* __restrict__ ray_o,
                                                    const float4* __restrict__ ray_d, float4* __restrict__ hit,
                                                    const int* __restrict__ n_ptr, int n_fixed, int cap_block,
                                                    unsigned long long* __restrict__ block_stats) {
    extern __shared__ float4 lds_nodes[];
    __shared__ __attribute__((aligned(16))) unsigned int s_stat[4];  // 16 bytes: keeps the dynamic region 16-byte aligned
    const int blk = blockIdx.x;
    int n;
    if (n_ptr) {
        n = n_ptr[blk];
        n = n < cap_block ? n : cap_block;
    } else {
        const long long rest = (long long)n_fixed - (long long)blk * cap_block;
        n = rest <= 0 ? 0 : (rest < cap_block ? (int)rest : cap_block);
    }
    if (n <= 0) return;
    const size_t region = (size_t)blk * cap_block;

    const int staged = sc.lds_nodes;
    for (int k = threadIdx.x; k < staged * 4; k += blockDim.x) lds_nodes[k] = sc.nodes[k];
    if (threadIdx.x < 3) s_stat[threadIdx.x] = 0;
    __syncthreads();

    const vec3 rlo = v3(sc.root_lo[0], sc.root_lo[1], sc.root_lo[2]);
    const vec3 rhi = v3(sc.root_hi[0], sc.root_hi[1], sc.root_hi[2]);
    unsigned int n_box = 0, n_tri = 0, n_hit = 0;

    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const size_t i = region + j;
        const float4 ro = ray_o[i], rd = ray_d[i];
        const vec3 o = v3(ro.x, ro.y, ro.z), d = v3(rd.x, rd.y, rd.z);
        Best best;
        best.t = FLT_MAX; best.beta = 0.0f; best.gamma = 0.0f; best.tri = -1;

        int stack[TRACE_STACK];
        float stack_t[KD ? TRACE_STACK : 1];
        int sp = 0;
        float t_root;
        if (COUNT) n_box++;
        bool walking = slab_test(rlo.x, rlo.y, rlo.z, rhi.x, rhi.y, rhi.z, o, d, t_root);
        int cur = sc.root_ref;
        while (walking) {
            if (cur < 0) {
                const int off = ~cur;
                const int cnt = sc.refs[off];
                for (int k = 1; k <= cnt; k++) {
                    if (COUNT) n_tri++;
                    triangle_test(sc, sc.refs[off + k], o, d, best);
                }
                walking = false;
            } else {
                float4 q0, q1, q2, q3;
                if (cur < staged) {
                    q0 = lds_nodes[4 * cur]; q1 = lds_nodes[4 * cur + 1]; q2 = lds_nodes[4 * cur + 2]; q3 = lds_nodes[4 * cur + 3];
                } else {
                    const float4* np = sc.nodes + 4 * (size_t)cur;
                    q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
                }
                float tl, tr;
                if (COUNT) n_box += 2;
                bool hl = slab_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, d, tl);
                bool hr = slab_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, d, tr);
                hl = hl && (tl < best.t);
                hr = hr && (tr < best.t);
                const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
                if (hl && hr) {
                    const bool left_first = tl < tr;
                    stack[sp] = left_first ? right : left;
                    if (KD) stack_t[sp] = left_first ? tr : tl;
                    sp++;
                    cur = left_first ? left : right;
                    continue;
                } else if (hl) {
                    cur = left;
                    continue;
                } else if (hr) {
                    cur = right;
                    continue;
                }
                walking = false;
            }
            // take the next deferred child
            while (sp > 0) {
                sp--;
                if (KD && !(stack_t[sp] < best.t)) continue;
                cur = stack[sp];
                walking = true;
                break;
            }
        }
        if (best.tri >= 0) n_hit++;
        hit[i] = make_float4(__int_as_float(best.tri), best.t, best.beta, best.gamma);
    }

    // statistics: wave sums -> LDS -> this workgroup's own slots (no global atomics)
    unsigned int tot_hit = n_hit, tot_box = n_box, tot_tri = n_tri;
    for (int off = 32; off > 0; off >>= 1) {
        tot_hit += __shfl_down(tot_hit, off);
        if (COUNT) { tot_box += __shfl_down(tot_box, off); tot_tri += __shfl_down(tot_tri, off); }
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_stat[0], tot_hit);
        if (COUNT) { atomicAdd(&s_stat[1], tot_box); atomicAdd(&s_stat[2], tot_tri); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* st = block_stats + (size_t)blk * DC_BLOCK_COUNT;
        st[DB_HITS] += s_stat[0];
        if (COUNT) { st[DB_BOX] += s_stat[1]; st[DB_TRI] += s_stat[2]; }
    }
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
        if (bary2) { bary2[2 * i] = h.z; bary2[2 * i + 1] = h.w; }
    }
}

}  // namespace

hipError_t launch_trace(const LaunchCfg& cfg, const SceneGpu& scene, int kd, int count_tests, int fast_slab,
                        const float4* ray_o, const float4* ray_d, float4* hit, const int* n_rays, int n_fixed,
                        int grid_blocks, int cap_block, unsigned long long* block_stats) {
    (void)fast_slab;
    const size_t lds = (size_t)scene.lds_nodes * 64;
    dim3 grid(grid_blocks), block(256);
    if (kd) {
        if (count_tests) hipLaunchKernelGGL((trace_kernel<true, true>), grid, block, lds, cfg.stream, scene, ray_o, ray_d, hit, n_rays, n_fixed, cap_block, block_stats);
        else hipLaunchKernelGGL((trace_kernel<true, false>), grid, block, lds, cfg.stream, scene, ray_o, ray_d, hit, n_rays, n_fixed, cap_block, block_stats);
    } else {
        if (count_tests) hipLaunchKernelGGL((trace_kernel<false, true>), grid, block, lds, cfg.stream, scene, ray_o, ray_d, hit, n_rays, n_fixed, cap_block, block_stats);
        else hipLaunchKernelGGL((trace_kernel<false, false>), grid, block, lds, cfg.stream, scene, ray_o, ray_d, hit, n_rays, n_fixed, cap_block, block_stats);
    }
    return hipGetLastError();
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
