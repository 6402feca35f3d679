// trace_device_two.h -- measurement form of the closest-hit walk: every lane carries TWO rays (gfx950).
// Written for trees that are walked in memory; slower than one ray a lane on every scene measured, and kept behind a
// setting for that measurement.
//
// The walk of a large tree waits for its node fetches.  One way to have more of them in flight per SIMD is more
// resident waves; this form is the other one: a lane holds the state of two independent rays, and a step of the
// node phase first requests the node of each ray that stands on an interior node and only then takes the two steps,
// so a wave has twice the fetches in flight per register file and LDS stack it occupies.  Each ray's own sequence of
// decisions is trace_region's (trace_device.h; BVH::traverse, src/cs488.h:1671-1706, and the kd-tree's,
// src/cs488.h:1344-1375), so its hit record is bit-identical; only which lane and which of its two slots walks a
// ray differs.  Private queue regions, the dynamic refill and the reciprocal slab form only.  A scene held entirely in
// LDS has the same form with the nodes and records read from there (two dependent chains per wave instead of one).
#pragma once

#include "trace_device.h"

namespace pt {

namespace trace_dev {

#ifndef PT_TWO_STACK_ROWS_BVH
#define PT_TWO_STACK_ROWS_BVH 8
#endif
#ifndef PT_TWO_STACK_ROWS_KD
#define PT_TWO_STACK_ROWS_KD 6
#endif
// LDS rows of one ray's stack (entry e of slot r of lane t at [(r * ROWS + e) * 256 + t]); deeper entries in scratch
template <bool KD> struct LdsStackTwo { enum { ROWS = KD ? PT_TWO_STACK_ROWS_KD : PT_TWO_STACK_ROWS_BVH, ENTRIES = ROWS - 1 }; };

template <int R> using slot_c = std::integral_constant<int, R>;

// ALL_LDS: the whole tree and its leaf records are staged at lds_nodes (stage_tree<true>); otherwise nothing is.
template <bool KD, bool ALL_LDS>
__device__ __forceinline__ void trace_region_two(const SceneGpu& sc, const float4* __restrict__ ray_o,
                                                 const float4* __restrict__ ray_d, float4* __restrict__ hit, int n, size_t region,
                                                 unsigned long long* __restrict__ block_stats, int leave_k_and_refill,
                                                 const float4* lds_nodes, unsigned int* s_stat, int* s_stack, float* s_stack_t) {
    constexpr int ROWS = LdsStackTwo<KD>::ROWS;
    constexpr int LDS_STACK = LdsStackTwo<KD>::ENTRIES;
    constexpr int DEEP = TRACE_STACK - LDS_STACK;
    const int blk = blockIdx.x;
    const float4* const lds_records = lds_nodes + LDS_ITEM_F4 * sc.lds_nodes;
    const int* const lds_refs = reinterpret_cast<const int*>(lds_records + LDS_ITEM_F4 * sc.n_records);
    if (threadIdx.x < 6) s_stat[threadIdx.x] = 0;
    __syncthreads();

    const vec3 rlo = v3(sc.root_lo[0], sc.root_lo[1], sc.root_lo[2]);
    const vec3 rhi = v3(sc.root_hi[0], sc.root_hi[1], sc.root_hi[2]);
    unsigned int n_hit = 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    // per-lane state of the two rays in progress
    vec3 o[2], d[2], inv[2];
    Best best[2];
    int deep[2][DEEP];
    float deep_t[2][KD ? DEEP : 1];
    int sp[2], cur[2], leaf_k[2], leaf_n[2];
    size_t slot[2];
    float miss_scale[2];
    bool fast_ok[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        o[r] = v3(0.0f); d[r] = v3(0.0f); inv[r] = v3(0.0f);
        best[r].t = FLT_MAX; best[r].beta = 0.0f; best[r].gamma = 0.0f; best[r].tri = -1;
        sp[r] = 0; cur[r] = NO_RAY; leaf_k[r] = 0; leaf_n[r] = 0; slot[r] = 0; miss_scale[r] = 1.0f; fast_ok[r] = false;
    }
    bool drained = false;  // wave-uniform: the region has no unassigned ray left
    // the node array as a buffer: loads at offsets beyond its size return zeros and make no memory access
    const __amdgpu_buffer_rsrc_t node_buffer =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(sc.nodes), 0, (int)((unsigned int)sc.n_nodes << 6), 0x00020000);
    const int asked_min = (leave_k_and_refill >> 8) ? (leave_k_and_refill >> 8) : 16;
    const int refill_min = 2 * asked_min;  // the threshold counts idle lanes of a wave with one ray a lane: the same share here
    const int leave_k = leave_k_and_refill & 0xff;

    // trace_region's pop_or_finish for slot R
    auto pop_or_finish = [&](auto rc, bool want) {
        constexpr int R = decltype(rc)::value;
        int* const stack = s_stack + R * ROWS * 256;
        float* const stack_t = s_stack_t + (KD ? R * ROWS * 256 : 0);
        const int top = sp[R] - 1;
        const int row = min(max(top, 0), LDS_STACK - 1);
        int ref = stack[row * 256 + tid];
        float t_ref = 0.0f;
        if (KD) t_ref = stack_t[row * 256 + tid];
        if (__builtin_expect(lane_mask(want && top >= LDS_STACK) != 0ull, 0)) {
            if (want && top >= LDS_STACK) {
                ref = deep[R][top - LDS_STACK];
                if (KD) t_ref = deep_t[R][top - LDS_STACK];
            }
            wait_for_vector_memory();
        }
        bool popped = want && top >= 0;
        sp[R] = popped ? top : sp[R];
        if (KD) {  // a deferred child that the best hit has overtaken since is skipped (src/cs488.h:1366, 1369)
            bool again = popped && !(t_ref < best[R].t);
            while (lane_mask(again) != 0ull) {
                if (again) {
                    if (sp[R] > 0) {
                        sp[R]--;
                        if (sp[R] < LDS_STACK) { ref = stack[sp[R] * 256 + tid]; t_ref = stack_t[sp[R] * 256 + tid]; }
                        else { ref = deep[R][sp[R] - LDS_STACK]; t_ref = deep_t[R][sp[R] - LDS_STACK]; wait_for_vector_memory(); }
                        again = !(t_ref < best[R].t);
                    } else {
                        popped = false;
                        again = false;
                    }
                }
            }
        }
        const bool finished = want && !popped;
        cur[R] = popped ? ref : (finished ? (int)RAY_DONE : cur[R]);
    };
    auto write_finished = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if (cur[R] == RAY_DONE) {
            if (best[R].tri >= 0) n_hit++;
            hit[slot[R]] = make_float4(__int_as_float(best[R].tri), best[R].t, best[R].tri >= 0 ? best[R].beta : miss_scale[R], best[R].gamma);
            cur[R] = NO_RAY;
        }
    };
    // Hands the next unassigned queue entries of the region to the lanes of `want` (one LDS atomic).
    auto assign = [&](unsigned long long want, bool& mine, size_t& cand) {
        mine = false;
        cand = 0;
        const int asked = __popcll(want);
        if (asked == 0 || drained) return;  // wave-uniform
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&s_stat[3], (unsigned int)asked);
        base = __shfl(base, 0);
        drained = base + asked >= n;
        const int j = base + __popcll(want & ((1ull << lane) - 1ull));
        mine = ((want >> lane) & 1ull) != 0ull && j < n;
        cand = region + j;
    };
    auto start_ray = [&](auto rc, size_t at, const float4& ro, float dx, float dy, float dz) {
        constexpr int R = decltype(rc)::value;
        slot[R] = at;
        o[R] = v3(ro.x, ro.y, ro.z); d[R] = v3(dx, dy, dz);
        miss_scale[R] = ro.w;
        best[R].t = FLT_MAX; best[R].beta = 0.0f; best[R].gamma = 0.0f; best[R].tri = -1;
        sp[R] = 0;
        fast_ok[R] = sc.planes_ok != 0 && recip_ok(o[R], d[R], rlo, rhi);
        if (fast_ok[R]) inv[R] = v3(exact_reciprocal(d[R].x), exact_reciprocal(d[R].y), exact_reciprocal(d[R].z));
        float t_root;
        const bool in_root = fast_ok[R] ? slab_test_recip(rlo.x, rlo.y, rlo.z, rhi.x, rhi.y, rhi.z, o[R], d[R], inv[R], t_root)
                                        : slab_test(rlo.x, rlo.y, rlo.z, rhi.x, rhi.y, rhi.z, o[R], d[R], t_root);
        if (in_root) {  // src/cs488.h:1438, 1405
            cur[R] = sc.root_ref;
            leaf_k[R] = 0;
        } else {
            hit[slot[R]] = make_float4(__int_as_float(-1), FLT_MAX, ro.w, 0.0f);
        }
    };

    while (true) {
        // ---- idle slots take the next rays of the region, once enough of them are idle ----
        const unsigned long long idle0 = lane_mask(cur[0] <= RAY_DONE), idle1 = lane_mask(cur[1] <= RAY_DONE);
        const int n_idle = __popcll(idle0) + __popcll(idle1);
        if (n_idle == 128 && drained) break;
        if (!drained && n_idle >= refill_min) {
            bool mine0, mine1;
            size_t cand0, cand1;
            assign(idle0, mine0, cand0);
            assign(idle1, mine1, cand1);
            float4 ro0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rd0 = ro0, ro1 = ro0, rd1 = ro0;
            if (mine0) { ro0 = ray_o[sc.shared_origin ? 0 : cand0]; rd0 = ray_d[cand0]; }
            if (mine1) { ro1 = ray_o[sc.shared_origin ? 0 : cand1]; rd1 = ray_d[cand1]; }
            wait_for_vector_memory();  // here, so that no path into the walking loops carries a pending load
            write_finished(slot_c<0>());
            write_finished(slot_c<1>());
            if (mine0) start_ray(slot_c<0>(), cand0, ro0, rd0.x, rd0.y, rd0.z);
            if (mine1) start_ray(slot_c<1>(), cand1, ro1, rd1.x, rd1.y, rd1.z);
        }

        // ---- node phase: the slots on interior nodes walk until too few of them are left ----
        auto node_phase = [&](auto fast_form) {
            constexpr bool FAST = decltype(fast_form)::value;
            while (true) {
                const int walking0 = __popcll(lane_mask(cur[0] >= 0)), walking1 = __popcll(lane_mask(cur[1] >= 0));
                const int walking = walking0 + walking1;
                if (walking == 0) break;
                const int waiting = __popcll(lane_mask((unsigned int)cur[0] > (unsigned int)RAY_DONE)) +
                                    __popcll(lane_mask((unsigned int)cur[1] > (unsigned int)RAY_DONE));
                if (walking * leave_k < waiting) break;
                // Both slots' nodes are requested before either step is taken.  A tree in memory is read from memory
                // only (no top levels staged in LDS: a wave that reads some nodes from LDS and some from memory into the
                // same registers orders the two, and the second slot's request would wait for the first slot's data).  The loads are
                // buffer loads that every lane issues for both slots, without a branch (loads under a condition are
                // moved into the conditional block of the step that uses them, behind the other slot's, and the wait
                // in front of the first step then has to cover all eight); a slot that is not walking passes an
                // offset beyond the buffer, for which the hardware returns zeros without touching the cache.  The two
                // offsets are formed first and pinned, so that nothing between the eight loads needs a register that
                // a load in flight still owns.
                typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
                unsigned int a0 = cur[0] >= 0 ? (unsigned int)cur[0] << 6 : 0xfffffff0u;
                unsigned int a1 = cur[1] >= 0 ? (unsigned int)cur[1] << 6 : 0xfffffff0u;
                asm volatile("" : "+v"(a0), "+v"(a1));
                float4 q[2][4];
                auto fetch = [&](auto rc, unsigned int at) {
                    constexpr int R = decltype(rc)::value;
                    if (ALL_LDS) {  // an idle slot reads node 0 with every other idle lane: one broadcast
                        const float4* np = lds_item(lds_nodes, max(cur[R], 0));
                        const float2 kids = *reinterpret_cast<const float2*>(np + 3);
                        q[R][0] = np[0]; q[R][1] = np[1]; q[R][2] = np[2]; q[R][3] = make_float4(kids.x, kids.y, 0.0f, 0.0f);
                        return;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(node_buffer, at + 16u * k, 0, 0);
                        q[R][k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                    }
                };
                auto step = [&](auto rc) {
                    constexpr int R = decltype(rc)::value;
                    if (cur[R] >= 0) {
                        int* const stack = s_stack + R * ROWS * 256;
                        float* const stack_t = s_stack_t + (KD ? R * ROWS * 256 : 0);
                        NodeStep st;
                        if (FAST) node_step_recip<KD>(q[R][0], q[R][1], q[R][2], q[R][3], o[R], d[R], inv[R], best[R].t, st);
                        else node_step<false, KD>(q[R][0], q[R][1], q[R][2], q[R][3], o[R], d[R], inv[R], best[R].t, st);
                        const int row = min(sp[R], LDS_STACK);
                        stack[row * 256 + tid] = st.deferred;
                        if (KD) stack_t[row * 256 + tid] = st.deferred_t;
                        if (__builtin_expect(lane_mask(st.push && sp[R] >= LDS_STACK) != 0ull, 0)) {
                            if (st.push && sp[R] >= LDS_STACK) {
                                deep[R][sp[R] - LDS_STACK] = st.deferred;
                                if (KD) deep_t[R][sp[R] - LDS_STACK] = st.deferred_t;
                            }
                        }
                        sp[R] += st.push ? 1 : 0;
                        pop_or_finish(rc, !st.descend);
                        cur[R] = st.descend ? st.next : cur[R];
                        leaf_k[R] = 0;
                    }
                };
                // pins a slot's node in registers here, outside the condition of its step: loads that only a
                // conditional block uses are moved into that block, behind the other slot's
#define PT_PIN_NODE(R)                                                                                                       \
    asm volatile("" : "+v"(q[R][0].x), "+v"(q[R][0].y), "+v"(q[R][0].z), "+v"(q[R][0].w), "+v"(q[R][1].x), "+v"(q[R][1].y), \
                      "+v"(q[R][1].z), "+v"(q[R][1].w), "+v"(q[R][2].x), "+v"(q[R][2].y), "+v"(q[R][2].z), "+v"(q[R][2].w), \
                      "+v"(q[R][3].x), "+v"(q[R][3].y))
                // a slot in which no lane of the wave walks is left out (wave-uniform): its loads would cost the
                // vector-memory pipeline what any others cost
                if (walking0 != 0 && walking1 != 0) {
                    fetch(slot_c<0>(), a0);
                    fetch(slot_c<1>(), a1);
                    PT_PIN_NODE(0);
                    step(slot_c<0>());
                    PT_PIN_NODE(1);
                    step(slot_c<1>());
                } else if (walking0 != 0) {
                    fetch(slot_c<0>(), a0);
                    PT_PIN_NODE(0);
                    step(slot_c<0>());
                } else {
                    fetch(slot_c<1>(), a1);
                    PT_PIN_NODE(1);
                    step(slot_c<1>());
                }
#undef PT_PIN_NODE
            }
        };
        if (lane_mask((cur[0] > RAY_DONE && !fast_ok[0]) || (cur[1] > RAY_DONE && !fast_ok[1])) == 0ull) node_phase(std::true_type());
        else node_phase(std::false_type());

        // ---- leaf phase: every slot on a leaf tests up to LEAF_BURST of its triangles, in stored order
        // (src/cs488.h:1677-1681, 1350-1354), one slot after the other
        auto leaf_phase = [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            if ((unsigned int)cur[R] > (unsigned int)RAY_DONE) {
                const int off = ~cur[R];
                if (KD && !sc.leaf_inline) {
                    const int n_here = ALL_LDS ? lds_refs[off] : sc.refs[off];
                    leaf_n[R] = leaf_k[R] == 0 ? n_here : leaf_n[R];
#pragma unroll
                    for (int j = 0; j < LEAF_BURST; j++) {
                        const bool valid = leaf_k[R] + j < leaf_n[R];
                        if (j > 0 && lane_mask(valid) == 0ull) break;
                        const int k = valid ? leaf_k[R] + j : leaf_n[R] - 1;
                        const float4* rec = ALL_LDS ? lds_item(lds_records, lds_refs[off + 1 + k]) : sc.isect + 4 * (size_t)sc.refs[off + 1 + k];
                        triangle_round<true>(rec[0], rec[1], rec[2], rec[3], o[R], d[R], valid, best[R]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < LEAF_BURST; j++) {
                        const bool valid = j == 0 || leaf_k[R] + j < leaf_n[R];
                        if (j > 0 && lane_mask(valid) == 0ull) break;
                        const int k = valid ? leaf_k[R] + j : leaf_n[R] - 1;
                        const float4* rec = ALL_LDS ? lds_item(lds_records, off + k) : sc.isect + 4 * (size_t)(off + k);
                        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                        if (j == 0) leaf_n[R] = leaf_k[R] == 0 ? __float_as_int(r1.w) : leaf_n[R];  // a leaf's first record carries its size
                        triangle_round<true>(r0, r1, r2, r3, o[R], d[R], valid, best[R]);
                    }
                }
                leaf_k[R] += LEAF_BURST;
                pop_or_finish(rc, leaf_k[R] >= leaf_n[R]);
                leaf_k[R] = leaf_k[R] >= leaf_n[R] ? 0 : leaf_k[R];
            }
        };
        leaf_phase(slot_c<0>());
        leaf_phase(slot_c<1>());
    }

    write_finished(slot_c<0>());
    write_finished(slot_c<1>());
    unsigned int tot_hit = n_hit;
    for (int off = 32; off > 0; off >>= 1) tot_hit += __shfl_down(tot_hit, off);
    if (lane == 0) atomicAdd(&s_stat[0], tot_hit);
    __syncthreads();
    if (threadIdx.x == 0) block_stats[(size_t)blk * DC_BLOCK_COUNT + DB_HITS] += s_stat[0];
}

}  // namespace trace_dev

}  // namespace pt
