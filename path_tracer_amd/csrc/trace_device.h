// trace_device.h -- device code of the closest-hit stage (gfx950), shared by the per-level trace
// kernel and the fused kernel that finishes the deep bounce levels.
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
// memory (written by the previous stage), so the host never waits for a count.  A wave alternates
// two tight loops: the node phase, in which every lane standing on an interior node takes steps
// (two slab tests, push, descend or pop) while the other lanes wait, and the leaf phase, in which
// every lane standing on a leaf tests up to four of its triangles.  The node phase is left when
// none of its lanes remain or when they are outnumbered k to 1 by the lanes waiting at a leaf, so
// neither code path runs nearly empty because of one long walk or one large leaf.  Lanes whose ray
// is finished take the next rays of the region through an LDS counter (one atomic per refill) as
// soon as half of the wave is idle (tunable).  The first `lds_nodes` tree nodes (the top levels; nodes are numbered breadth-first) are
// copied into LDS by every workgroup that has work; the first LDS_STACK entries of every lane's
// traversal stack live in LDS as well.
#pragma once

#include <float.h>
#include <limits.h>

#include <type_traits>

#include "device_math.h"
#include "device_types.h"
#include "pt_math.h"

namespace pt {

namespace trace_dev {

// The wave's mask of a per-lane condition, read straight from the comparison's result (__ballot
// takes an int and costs a select and a second comparison per use).
__device__ __forceinline__ unsigned long long lane_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// s_waitcnt vmcnt(0): every vector-memory operation of the wave has completed.  The compiler places its own waits
// where a loaded register is first used, on every path that may reach the use; written where the load is, the wait
// stays off the paths that never loaded.
__device__ __forceinline__ void wait_for_vector_memory() { __builtin_amdgcn_s_waitcnt(0x0F70); }


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

// a.x + a.y + a.z, in that order
__device__ __forceinline__ float sum3(vec3 a) { return (a.x + a.y) + a.z; }

struct Best {
    float t, beta, gamma;
    int tri;
};

// TriangleMesh::raytraceTriangle (src/cs488.h:646-681) on an intersection record: the edge vectors
// a, b and cross(a, b) come precomputed (they do not depend on the ray), the rest is evaluated here
// in the reference's order.  The three quotients by the determinant are true divisions in the
// reference; FAST_DIV forms them from the exact reciprocal whenever the determinant and the three
// numerators are in a range in which no intermediate overflows or becomes subnormal (a zero
// numerator is left to the division so that the sign of a zero quotient is the reference's).
template <bool FAST_DIV>
__device__ __forceinline__ void triangle_test(const float4& r0, const float4& r1, const float4& r2, const float4& r3,
                                              vec3 o, vec3 d, Best& best) {
    const vec3 a = v3(r1.x, r1.y, r1.z);
    const vec3 b = v3(r2.x, r2.y, r2.z);
    const vec3 ab = v3(r3.x, r3.y, r3.z);
    const vec3 e = v3(r0.x, r0.y, r0.z) - o;
    const float D = dot(ab, d);
    if (D != 0) {
        const float Da = dot(cross(e, b), d);
        const float Db = dot(cross(a, e), d);
        const float Dc = dot(ab, e);
        float beta, gamma, t;
        const float lo = fminf(fminf(fabsf(Da), fabsf(Db)), fabsf(Dc));
        const float hi = fmaxf(fmaxf(fabsf(Da), fabsf(Db)), fabsf(Dc));
        if (FAST_DIV && lo >= 1e-22f && hi <= 1e15f && fabsf(D) >= 1e-15f && fabsf(D) <= 1e15f) {
            const float r = exact_reciprocal(D);
            beta = div_by_recip(Da, D, r);
            gamma = div_by_recip(Db, D, r);
            t = div_by_recip(Dc, D, r);
        } else {
            beta = Da / D;
            gamma = Db / D;
            t = Dc / D;
        }
        const float alpha = 1 - beta - gamma;
        if (0 <= alpha && alpha <= 1 && 0 <= beta && beta <= 1 && 0 <= gamma && gamma <= 1 && 0.0f <= t && t <= FLT_MAX) {
            if (t < best.t) { best.t = t; best.beta = beta; best.gamma = gamma; best.tri = __float_as_int(r0.w); }
        }
    }
}

// The same test for one round of the leaf phase, for all lanes of the wave at once: `valid` is false
// for a lane that has no triangle in this round (it computes on a repeated record and its result is
// dropped).  Which of the two quotient forms is used is decided per wave, not per lane (both give
// RN(numerator / D) for operands inside the guard, and the division is right for all), and the
// acceptance test of the guarded form is folded into a minimum and a maximum: everything in it is
// finite there, so 0 <= alpha, beta, gamma, t  <=>  min(alpha, beta, gamma, t) >= 0, likewise <= 1,
// and t <= FLT_MAX follows from t < best.t.  Few scalar instructions result, which is what the
// leaf phase is short of.
template <bool FAST_DIV>
__device__ __forceinline__ void triangle_round(const float4& r0, const float4& r1, const float4& r2, const float4& r3,
                                               vec3 o, vec3 d, bool valid, Best& best) {
    const vec3 a = v3(r1.x, r1.y, r1.z);
    const vec3 b = v3(r2.x, r2.y, r2.z);
    const vec3 ab = v3(r3.x, r3.y, r3.z);
    const vec3 e = v3(r0.x, r0.y, r0.z) - o;
    // The reference's dot product starts its sum at 0 (src/linalg.h:401, 501), which changes one thing only: a
    // sum of three products that are all -0 comes out as +0.  The sums are formed without that addition here; the
    // sign of a zero determinant is never looked at, a zero numerator always takes the division form below, and
    // there the three numerators get their + 0 back.
    const float D = sum3(ab * d);
    float Da = sum3(cross(e, b) * d);
    float Db = sum3(cross(a, e) * d);
    float Dc = sum3(ab * e);
    const bool live = valid && D != 0;
    bool guarded = false;
    if (FAST_DIV) {
        const float lo = fminf(fminf(fabsf(Da), fabsf(Db)), fabsf(Dc));
        const float hi = fmaxf(fmaxf(fabsf(Da), fabsf(Db)), fabsf(Dc));
        const bool in_guard = lo >= 1e-22f && hi <= 1e15f && fabsf(D) >= 1e-15f && fabsf(D) <= 1e15f;
        guarded = lane_mask(live && !in_guard) == 0ull;
    }
    float beta, gamma, t;
    bool accept;
    if (guarded) {
        const float r = exact_reciprocal(D);
        beta = div_by_recip(Da, D, r);
        gamma = div_by_recip(Db, D, r);
        t = div_by_recip(Dc, D, r);
        const float alpha = 1 - beta - gamma;
        const float least = fminf(fminf(fminf(alpha, beta), gamma), t);
        const float most = fmaxf(fmaxf(alpha, beta), gamma);
        accept = live && least >= 0.0f && most <= 1.0f && t < best.t;
    } else {
        Da = 0.0f + Da; Db = 0.0f + Db; Dc = 0.0f + Dc;
        beta = Da / D;
        gamma = Db / D;
        t = Dc / D;
        const float alpha = 1 - beta - gamma;
        accept = live && 0 <= alpha && alpha <= 1 && 0 <= beta && beta <= 1 && 0 <= gamma && gamma <= 1 && 0.0f <= t && t <= FLT_MAX && t < best.t;
    }
    best.t = accept ? t : best.t;
    best.beta = accept ? beta : best.beta;
    best.gamma = accept ? gamma : best.gamma;
    best.tri = accept ? __float_as_int(r0.w) : best.tri;
}

// ---- slab test without the division sequence ---------------------------------------------------
// SLAB_DIVIDE: the division form above, always.
// SLAB_RECIP:  every quotient (plane - origin) / direction is formed from the per-ray reciprocal
//   r = RN(1 / d) as  q0 = RN(a r);  e = RN(fma(-q0, d, a));  q = RN(fma(e, r, q0)).
//   The residual e is exact and the corrected q equals RN(a / d) for EVERY pair of binary32
//   significands -- checked exhaustively (2^46 pairs, tools/verify_recip_division.hip) -- hence for
//   all operands as long as nothing overflows or becomes subnormal.  Rays for which that cannot be
//   promised from their origin and direction alone (`recip_ok` below) use the division form, so both
//   settings give bit-identical entry distances and therefore the identical walk.
enum { SLAB_DIVIDE = 0, SLAB_RECIP = 1 };

// Only called for rays that passed recip_ok: all six quotients are finite, so the reference's
// swap / running-max / running-min sequence (src/cs488.h:541-560) reduces to plain min and max
// (the sign of a zero may differ, which no comparison below or in the walk can see).
__device__ __forceinline__ bool slab_test_recip(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                                vec3 o, vec3 d, vec3 r, float& t_entry) {
    const float tx1 = div_by_recip(lox - o.x, d.x, r.x), ty1 = div_by_recip(loy - o.y, d.y, r.y), tz1 = div_by_recip(loz - o.z, d.z, r.z);
    const float tx2 = div_by_recip(hix - o.x, d.x, r.x), ty2 = div_by_recip(hiy - o.y, d.y, r.y), tz2 = div_by_recip(hiz - o.z, d.z, r.z);
    const float t1 = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    const float t2 = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    t_entry = t1;
    return !(t1 > t2) && !((t1 < 0.0f) && (t2 < 0.0f));
}

// May the reciprocal form be used for this ray?  Needs, per axis: a finite direction component
// with 1e-10 <= |d| <= 1e10 (so r and every quotient stay normal), an origin coordinate that is 0
// or at least 1e-17 in magnitude (with the same property of the box planes, checked on the host,
// a non-zero plane - origin is then at least 1e-25 and its quotient and residual stay normal), and
// |plane - origin| < 1e15 for every plane inside the root box.
__device__ __forceinline__ bool recip_ok(vec3 o, vec3 d, vec3 rlo, vec3 rhi) {
    const float dmax = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    const float dmin = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    const float reach = fmaxf(fmaxf(fmaxf(fabsf(rlo.x - o.x), fabsf(rhi.x - o.x)), fmaxf(fabsf(rlo.y - o.y), fabsf(rhi.y - o.y))),
                              fmaxf(fabsf(rlo.z - o.z), fabsf(rhi.z - o.z)));
    const bool ox = o.x == 0.0f || fabsf(o.x) >= 1e-17f, oy = o.y == 0.0f || fabsf(o.y) >= 1e-17f, oz = o.z == 0.0f || fabsf(o.z) >= 1e-17f;
    return (dmax <= 1e10f) && (dmin >= 1e-10f) && (reach < 1e15f) && ox && oy && oz;  // NaNs fail the comparisons
}

// What one visit of an interior node decides.
struct NodeStep {
    int next;          // child to enter now (valid if descend)
    int deferred;      // child to push (valid if push)
    float deferred_t;  // its entry distance (kd-tree only)
    bool descend, push;
};

// The reference's decisions for one interior node (src/cs488.h:1684-1701, 1357-1375).
template <bool RECIP, bool KD>
__device__ __forceinline__ void node_step(const float4& q0, const float4& q1, const float4& q2, const float4& q3,
                                          vec3 o, vec3 d, vec3 r, float best_t, NodeStep& st) {
    float tl, tr;
    bool hl, hr;
    if (RECIP) {
        hl = slab_test_recip(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, d, r, tl);
        hr = slab_test_recip(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, d, r, tr);
    } else {
        hl = slab_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, d, tl);
        hr = slab_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, d, tr);
    }
    const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
    hl = hl && (tl < best_t);
    hr = hr && (tr < best_t);
    if (KD) {  // an empty leaf (kd-trees have them) holds nothing to test: entering it would change
        hl = hl && left != EMPTY_LEAF;   // neither the hit nor the order of the rest
        hr = hr && right != EMPTY_LEAF;
    }
    st.push = hl && hr;
    st.descend = hl || hr;
    const bool left_first = hl && (!hr || (tl < tr));  // both: nearer first, right on ties; one: that one
    st.next = left_first ? left : right;
    st.deferred = left_first ? right : left;
    st.deferred_t = left_first ? tr : tl;
}

// The same decisions for a ray that passed recip_ok, written so that the compiler needs no scalar
// mask arithmetic for them (a gfx950 SIMD issues one scalar instruction per four cycles against one
// vector instruction per two, and the short-circuit form of the conditions above costs 25 scalar
// instructions per step).  All six quotients per box are finite here, so
//   hit  <=>  t1 <= t2  and not (t1 < 0 and t2 < 0)  <=>  max(t1, 0) <= t2,
// and with e = t1 for a hit box, +inf otherwise (or for an empty kd leaf):
//   enter left / right  <=>  e_l < best / e_r < best;   left first  <=>  e_l < e_r
// (if only the left is entered, e_l < best <= e_r; if only the right, e_r < best <= e_l; with both,
// nearer first and the right one on ties); descend <=> min(e_l, e_r) < best, push <=> max(..) < best,
// and the pushed child's entry distance is that max.
template <bool KD>
__device__ __forceinline__ void node_step_recip(const float4& q0, const float4& q1, const float4& q2, const float4& q3,
                                                vec3 o, vec3 d, vec3 r, float best_t, NodeStep& st) {
    const float lx1 = div_by_recip(q0.x - o.x, d.x, r.x), ly1 = div_by_recip(q0.y - o.y, d.y, r.y), lz1 = div_by_recip(q0.z - o.z, d.z, r.z);
    const float lx2 = div_by_recip(q0.w - o.x, d.x, r.x), ly2 = div_by_recip(q1.x - o.y, d.y, r.y), lz2 = div_by_recip(q1.y - o.z, d.z, r.z);
    const float l1 = fmaxf(fmaxf(fminf(lx1, lx2), fminf(ly1, ly2)), fminf(lz1, lz2));
    const float l2 = fminf(fminf(fmaxf(lx1, lx2), fmaxf(ly1, ly2)), fmaxf(lz1, lz2));
    const float rx1 = div_by_recip(q1.z - o.x, d.x, r.x), ry1 = div_by_recip(q1.w - o.y, d.y, r.y), rz1 = div_by_recip(q2.x - o.z, d.z, r.z);
    const float rx2 = div_by_recip(q2.y - o.x, d.x, r.x), ry2 = div_by_recip(q2.z - o.y, d.y, r.y), rz2 = div_by_recip(q2.w - o.z, d.z, r.z);
    const float r1 = fmaxf(fmaxf(fminf(rx1, rx2), fminf(ry1, ry2)), fminf(rz1, rz2));
    const float r2 = fminf(fminf(fmaxf(rx1, rx2), fmaxf(ry1, ry2)), fmaxf(rz1, rz2));
    const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
    float el = fmaxf(l1, 0.0f) <= l2 ? l1 : INFINITY;
    float er = fmaxf(r1, 0.0f) <= r2 ? r1 : INFINITY;
    if (KD) {
        el = left == EMPTY_LEAF ? INFINITY : el;
        er = right == EMPTY_LEAF ? INFINITY : er;
    }
    const bool left_first = el < er;
    st.next = left_first ? left : right;
    st.deferred = left_first ? right : left;
    st.deferred_t = fmaxf(el, er);
    st.descend = fminf(el, er) < best_t;
    st.push = st.deferred_t < best_t;
}

// A lane's traversal stack: LdsStack::ENTRIES entries in LDS (entry e of lane t at [e * 256 + t]:
// consecutive lanes on consecutive banks), deeper ones in the lane's scratch array.  The LDS array
// has one more row that is never read: a step stores its deferred child unconditionally, into the
// row above the top (dead until a push makes it the top) or, from a lane whose LDS rows are full,
// into that spare row.
#ifndef PT_BVH_STACK_ROWS
#define PT_BVH_STACK_ROWS 16   // (a build with another value, for measurements: make EXTRA=-DPT_BVH_STACK_ROWS=8 OUT=...)
#endif
#ifndef PT_KD_STACK_ROWS
#define PT_KD_STACK_ROWS 8
#endif
enum { LDS_STACK_ROWS_BVH = PT_BVH_STACK_ROWS, LDS_STACK_ROWS_KD = PT_KD_STACK_ROWS };  // a kd-tree entry is two words (node, entry distance)
template <bool KD> struct LdsStack { enum { ROWS = KD ? LDS_STACK_ROWS_KD : LDS_STACK_ROWS_BVH, ENTRIES = ROWS - 1 }; };
// Triangles a lane tests per visit of the leaf phase.
enum { LEAF_BURST = 4 };

// A lane's `cur`: >= 0 interior node, < 0 leaf (~offset of its first record), NO_RAY = no ray, RAY_DONE = no ray
// and the record of the last one still to be stored (EMPTY_LEAF, INT_MIN + 1, lies between the two and is never
// a lane's position).  Leaf offsets stay far below 2^31 - 3, so every leaf reference is above RAY_DONE.
enum : int { NO_RAY = INT_MIN, RAY_DONE = INT_MIN + 2 };

// Nodes and records staged in LDS are 64 bytes each but stored 80 bytes apart (five float4): with a stride of 64
// bytes every node's quarter p would start at one of only four bank positions, so lanes of one ds_read_b128 that read
// different nodes would collide four to one and worse; 20 dwords apart, sixteen consecutive nodes start at sixteen
// different multiples of four banks.
enum { LDS_ITEM_F4 = 5 };
// Item `i` of an LDS array of such items.  The index is small, so the offset is a 24-bit multiply-add (full
// rate) where plain pointer arithmetic compiles to a 32-bit multiply (quarter rate on gfx950) in every step.
__device__ __forceinline__ const float4* lds_item(const float4* base, int i) {
    return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + __umul24((unsigned int)i, 16u * LDS_ITEM_F4));
}

// ALL_LDS: every interior node and every (leaf-ordered) intersection record is staged in LDS --
// small scenes -- so neither the walk nor the leaf phase has a global-memory path.
enum { LDS_RECORDS_MAX = 256, LDS_REFS_MAX = 2048 };

// May the ALL_LDS instantiation serve this scene?  (host side; decides the dynamic LDS size too)
__host__ __device__ inline bool fits_lds(const SceneGpu& sc) {
    return sc.lds_nodes >= sc.n_nodes && sc.n_records <= LDS_RECORDS_MAX && (sc.leaf_inline || sc.n_refs <= LDS_REFS_MAX);
}
__host__ __device__ inline size_t dynamic_lds_bytes(const SceneGpu& sc) {
    size_t b = (size_t)sc.lds_nodes * (16 * LDS_ITEM_F4);
    if (fits_lds(sc)) b += (size_t)sc.n_records * (16 * LDS_ITEM_F4) + (sc.leaf_inline ? 0 : (size_t)sc.n_refs * 4);
    return b;
}
#ifdef PT_TIMELINE
enum { PT_STAT_WORDS = 16 };  // + the time each wave ran out of rays
#else
enum { PT_STAT_WORDS = 8 };
#endif
// LDS a workgroup needs for the walk: the staged nodes (and records), four counters, the stacks.
template <bool KD>
struct TraceLds {
    float4* nodes;          // dynamic LDS: lds_nodes node records, then (ALL_LDS) the leaf records
    unsigned int* stat;     // 6: hits, slab tests, triangle tests, next ray of the region, wave node steps, wave leaf rounds
    int* stack;             // LdsStack<KD>::ROWS * 256
    float* stack_t;         // the same number of entry distances (kd-tree only)
    int* count;             // POOL_REGIONS words about the regions the workgroup draws from (pooled source only)
};

// Pooled ray source.  The regions of a launch are cut into POOL_GROUPS runs of consecutive regions; the workgroups
// that serve one run (workgroup b serves run b % POOL_GROUPS) share ONE cursor in global memory over the run's rays,
// handed out in chunks of POOL_CHUNK consecutive queue entries, one chunk per wave at a time.  Which wave traces a
// ray does not change its hit record, so the rays need not stay with the workgroup whose region holds them.  Why
// share: a SIMD issues from its oldest waves first, so workgroups with equal shares of the rays finish one after the
// other (measured on a 5.8 M-ray launch: between 195 and 345 us, the last-placed ones at 447 us) and the chip runs
// half empty for the second half of the launch; drawing from a pool every wave works until the run is empty (395 us).
// That pays when one launch has the chip to itself.  With a second batch in flight its kernels fill what the early
// finishers leave, and private regions are the faster form (the host chooses, see launch_trace's mode).
// The cursor must be zero when the launch starts.
enum { POOL_GROUPS = 32, POOL_REGIONS = 64, POOL_CHUNK = 64 };
// most queue entries per region the pooled source serves: 64 regions' chunk counts x 64 must fit an int
enum { POOL_MAX_REGION = 1 << 24 };
struct RayPool {
    int* cursor;         // next chunk of the run
    const int* counts;   // rays in each region of the launch (device memory), or nullptr: one flat array of n_fixed rays
    int n_fixed;
    int cap_block;       // queue entries per region
    int first;           // first region of the run
    int regions;         // regions in the run (<= POOL_REGIONS)
};

// Inclusive prefix sum over the wave.
__device__ __forceinline__ int wave_inclusive_sum(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(v, off);
        v += lane >= off ? up : 0;
    }
    return v;
}

// Copies the top of the tree (and, with ALL_LDS, the leaf records) into LDS; every thread of the
// workgroup calls it, a barrier must follow before the first walk.
template <bool ALL_LDS>
__device__ __forceinline__ void stage_tree(const SceneGpu& sc, float4* lds_nodes) {
    const int staged = sc.lds_nodes;
    for (int k = threadIdx.x; k < staged * 4; k += blockDim.x) lds_nodes[(k >> 2) * LDS_ITEM_F4 + (k & 3)] = sc.nodes[k];
    float4* lds_records = lds_nodes + LDS_ITEM_F4 * staged;
    if (ALL_LDS) {
        for (int k = threadIdx.x; k < sc.n_records * 4; k += blockDim.x) lds_records[(k >> 2) * LDS_ITEM_F4 + (k & 3)] = sc.isect[k];
        if (!sc.leaf_inline) {  // leaves as index lists behind the records
            int* lds_refs = reinterpret_cast<int*>(lds_records + LDS_ITEM_F4 * sc.n_records);
            for (int k = threadIdx.x; k < sc.n_refs; k += blockDim.x) lds_refs[k] = sc.refs[k];
        }
    }
}

// Closest hits for the n rays at [region, region + n) of the queue (POOL = false) or for the rays
// the workgroup's waves draw from a pool shared with other workgroups (POOL = true: n and region are
// unused); called by all 256 threads of the workgroup with the tree already staged.  Ends with the
// workgroup's hit records written and a barrier passed.
// WIDE (BVH in memory only): a node outside the LDS stage is fetched as a 192-byte record that carries the node
// and, behind it, the node records of its two children (SceneGpu::wide).  All three parts are requested at once;
// the lane takes the node's step and, if that step descends into an interior child, the child's step right away
// from the part already loaded -- one dependent fetch per two levels of BVH::traverse (src/cs488.h:1671-1706)
// instead of one per level.  The steps, their order and their stack traffic are the binary walk's own.
// (Measured slower than the 64-byte nodes on every scene, by 20-47 %: the walk is bound by the bytes it gathers
// through the vector cache, not by the length of its chain of dependent fetches.  Kept as a measurement setting.)
template <bool KD, bool COUNT, int SLAB, bool DYNAMIC, bool ALL_LDS, bool POOL, bool WIDE = false>
__device__ __forceinline__ void trace_region(const SceneGpu& sc, const float4* __restrict__ ray_o,
                                             const float4* __restrict__ ray_d, float4* __restrict__ hit, int n, size_t region,
                                             unsigned long long* __restrict__ block_stats, int leave_k_and_refill, const TraceLds<KD>& lds,
                                             const RayPool& pool) {
    constexpr int LDS_STACK = LdsStack<KD>::ENTRIES;
    float4* const lds_nodes = lds.nodes;
    unsigned int* const s_stat = lds.stat;
    int* const s_stack = lds.stack;
    float* const s_stack_t = lds.stack_t;
    const int blk = blockIdx.x;
    const int staged = sc.lds_nodes;
    float4* const lds_records = lds_nodes + LDS_ITEM_F4 * staged;
    const int* const lds_refs = reinterpret_cast<const int*>(lds_records + LDS_ITEM_F4 * sc.n_records);
    if (threadIdx.x < 6) s_stat[threadIdx.x] = 0;
    int* const s_count = lds.count;
    if (POOL) {
        if (threadIdx.x < POOL_REGIONS) {  // the first wave: chunks up to and including each region of the run
            int cnt = 0;
            if ((int)threadIdx.x < pool.regions) {
                const int r = pool.first + (int)threadIdx.x;
                if (pool.counts) cnt = pool.counts[r];
                else { const long long rest = (long long)pool.n_fixed - (long long)r * pool.cap_block; cnt = rest > INT_MAX ? INT_MAX : (int)rest; }
                cnt = cnt < 0 ? 0 : (cnt > pool.cap_block ? pool.cap_block : cnt);
            }
            const int incl = wave_inclusive_sum((cnt + POOL_CHUNK - 1) / POOL_CHUNK, (int)threadIdx.x);
            s_count[threadIdx.x] = incl * POOL_CHUNK + ((cnt - 1) & (POOL_CHUNK - 1));  // and what its last chunk holds, less one
        }
    }
    __syncthreads();

    const vec3 rlo = v3(sc.root_lo[0], sc.root_lo[1], sc.root_lo[2]);
    const vec3 rhi = v3(sc.root_hi[0], sc.root_hi[1], sc.root_hi[2]);
    unsigned int n_box = 0, n_tri = 0, n_hit = 0;
    unsigned int n_wave_steps = 0, n_wave_rounds = 0;  // COUNT: node steps and leaf rounds the wave issued (counted by one lane)
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    // per-lane state of the ray in progress
    vec3 o = v3(0.0f), d = v3(0.0f), inv = v3(0.0f);
    Best best;
    best.t = FLT_MAX; best.beta = 0.0f; best.gamma = 0.0f; best.tri = -1;
    int deep[TRACE_STACK - LDS_STACK];
    float deep_t[KD ? TRACE_STACK - LDS_STACK : 1];
    int sp = 0, cur = NO_RAY;
    int leaf_k = 0, leaf_n = 0;  // next triangle of the leaf the lane stands on, and the leaf's size
    size_t slot = 0;
#ifdef PT_KD_MAILBOX
    unsigned long long tested = 0ull;  // triangles the lane's ray has been tested against
#endif
    float miss_scale = 1.0f;  // the ray's factor on the environment radiance (ray_o.w), handed on in a miss record
    bool fast_ok = false;
    bool drained = false;  // wave-uniform: the region has no unassigned ray left
    const int refill_min = DYNAMIC ? ((leave_k_and_refill >> 8) ? (leave_k_and_refill >> 8) : 16) : 64;
    const int leave_k = leave_k_and_refill & 0xff;

    // pooled source: the wave's current chunk (wave-uniform) and the number of chunks in the run
    size_t chunk_pos = 0;
    int chunk_left = 0, n_chunks = 0;
    if (POOL) n_chunks = s_count[POOL_REGIONS - 1] / POOL_CHUNK;
    // Takes the wave's next chunk: chunks are numbered region by region, s_count[r] = 64 x (chunks up to and including
    // region r) + (rays in r's last chunk - 1).
    auto next_chunk_of_pool = [&]() {
        int c0 = 0;
        if (lane == 0) c0 = atomicAdd(pool.cursor, 1);
        const int c = __builtin_amdgcn_readfirstlane(c0);
        if (c >= n_chunks) { drained = true; return; }
        const int packed = s_count[lane];
        const int j = __popcll(lane_mask(packed / POOL_CHUNK <= c));  // regions that end before chunk c
        const int before = j > 0 ? __builtin_amdgcn_readlane(packed, j - 1) / POOL_CHUNK : 0;
        const int here = __builtin_amdgcn_readlane(packed, j);
        const int piece = c - before;
        const bool last = piece == here / POOL_CHUNK - before - 1;
        chunk_pos = (size_t)(pool.first + j) * (size_t)pool.cap_block + (size_t)(piece * POOL_CHUNK);
        chunk_left = last ? (here & (POOL_CHUNK - 1)) + 1 : POOL_CHUNK;
    };

    // For the lanes that `want` it: takes the next deferred child, or, when none is left, marks the
    // ray finished (its record is stored at the next refill).  Called by all lanes of a region of code, with few branches:
    // the LDS row is read unconditionally (clamped), deep entries and skipped kd entries are rare
    // and handled behind wave-wide tests.
    auto pop_or_finish = [&](bool want) {
        const int top = sp - 1;
        const int row = min(max(top, 0), LDS_STACK - 1);
        int ref = s_stack[row * 256 + tid];
        float t_ref = 0.0f;
        if (KD) t_ref = s_stack_t[row * 256 + tid];
        if (__builtin_expect(lane_mask(want && top >= LDS_STACK) != 0ull, 0)) {
            if (want && top >= LDS_STACK) {
                ref = deep[top - LDS_STACK];
                if (KD) t_ref = deep_t[top - LDS_STACK];
            }
            wait_for_vector_memory();  // keeps the wait for these rare loads out of the common path
        }
        bool popped = want && top >= 0;
        sp = popped ? top : sp;
        if (KD) {  // a deferred child that the best hit has overtaken since is skipped (src/cs488.h:1366, 1369)
            bool again = popped && !(t_ref < best.t);
            while (lane_mask(again) != 0ull) {
                if (again) {
                    if (sp > 0) {
                        sp--;
                        if (sp < LDS_STACK) { ref = s_stack[sp * 256 + tid]; t_ref = s_stack_t[sp * 256 + tid]; }
                        else { ref = deep[sp - LDS_STACK]; t_ref = deep_t[sp - LDS_STACK]; wait_for_vector_memory(); }
                        again = !(t_ref < best.t);
                    } else {
                        popped = false;
                        again = false;
                    }
                }
            }
        }
        const bool finished = want && !popped;
        cur = popped ? ref : (finished ? (int)RAY_DONE : cur);
    };
    // Stores the records of the lanes whose ray is finished.  A finished lane stays idle until the next refill, so
    // its result can wait in its registers until then: the two walking loops then hold no store to global memory,
    // and no wait for one either (a wait for outstanding vector-memory operations on gfx950 counts stores as well).
    auto write_finished = [&]() {
        if (cur == RAY_DONE) {
            if (best.tri >= 0) n_hit++;
            // a miss carries the ray's miss scale where a hit has beta, so the shading stage reads one
            // record per path instead of the ray's origin as well
            hit[slot] = make_float4(__int_as_float(best.tri), best.t, best.tri >= 0 ? best.beta : miss_scale, best.gamma);
            cur = NO_RAY;
        }
    };

    // Hands the next unassigned queue entries to the lanes of `want`: `mine` says whether the lane got one, `cand`
    // which.  Sets `drained` when nothing is left for the wave.
    auto assign = [&](unsigned long long want, bool& mine, size_t& cand) {
        mine = false;
        cand = 0;
        if (POOL) {
            while (want != 0ull) {
                if (chunk_left == 0) {
                    next_chunk_of_pool();
                    if (drained) break;
                }
                const int asked = __popcll(want);
                const int rank = __popcll(want & ((1ull << lane) - 1ull));
                const bool got = ((want >> lane) & 1ull) != 0ull && rank < chunk_left;
                cand = got ? chunk_pos + rank : cand;
                mine = mine || got;
                const int take = asked < chunk_left ? asked : chunk_left;
                chunk_pos += take;
                chunk_left -= take;
                want = lane_mask(((want >> lane) & 1ull) != 0ull && !got);
            }
        } else {
            const int asked = __popcll(want);
            int base = 0;
            if (lane == 0) base = (int)atomicAdd(&s_stat[3], (unsigned int)asked);
            base = __shfl(base, 0);
            drained = base + asked >= n;
            const int j = base + __popcll(want & ((1ull << lane) - 1ull));
            mine = ((want >> lane) & 1ull) != 0ull && j < n;
            cand = region + j;
        }
    };
    // Starts the ray (ro, rd) of queue entry `at` on this lane: reciprocals, root box.
    auto start_ray = [&](size_t at, const float4& ro, float dx, float dy, float dz) {
        slot = at;
        o = v3(ro.x, ro.y, ro.z); d = v3(dx, dy, dz);
        miss_scale = ro.w;
        best.t = FLT_MAX; best.beta = 0.0f; best.gamma = 0.0f; best.tri = -1;
        sp = 0;
#ifdef PT_KD_MAILBOX
        tested = 0ull;
#endif
        if (SLAB == SLAB_RECIP) {
            fast_ok = sc.planes_ok != 0 && recip_ok(o, d, rlo, rhi);
            if (fast_ok) inv = v3(exact_reciprocal(d.x), exact_reciprocal(d.y), exact_reciprocal(d.z));
        }
        float t_root;
        if (COUNT) n_box++;
        const bool in_root = (SLAB == SLAB_RECIP && fast_ok) ? slab_test_recip(rlo.x, rlo.y, rlo.z, rhi.x, rhi.y, rhi.z, o, d, inv, t_root)
                                                             : slab_test(rlo.x, rlo.y, rlo.z, rhi.x, rhi.y, rhi.z, o, d, t_root);
        if (in_root) {  // src/cs488.h:1438, 1405
            cur = sc.root_ref;
            leaf_k = 0;
        } else {
            hit[slot] = make_float4(__int_as_float(-1), FLT_MAX, ro.w, 0.0f);
        }
    };


    while (true) {
        // ---- idle lanes take the next rays of the region, once enough of them are idle ----
        const unsigned long long idle = lane_mask(cur <= RAY_DONE);
        const int n_idle = __popcll(idle);
        if (n_idle == 64 && drained) break;
        if (!drained && n_idle >= refill_min) {
            bool mine;   // the lane takes a new ray, the one at queue entry `cand`
            size_t cand;
            assign(idle, mine, cand);
            // The new rays are loaded and waited for first, the records of the finished ones stored after that
            // wait: nothing then ever waits for those stores (a wait for vector memory counts loads and stores alike).
            float4 ro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rd = ro;
            if (mine) {
                // camera rays share one origin record (the eye, miss scale 1) and are stored as directions only
                ro = ray_o[sc.shared_origin ? 0 : cand];
                rd = ray_d[cand];
            }
            wait_for_vector_memory();  // here, so that no path into the walking loops carries a pending load
            write_finished();
            if (mine) start_ray(cand, ro, rd.x, rd.y, rd.z);
        }

        // ---- node phase: lanes on interior nodes walk until too few of them are left ----
        // (leave when nobody walks, or when the walkers are fewer than 1/k of the lanes waiting at a leaf)
        // Written once and instantiated twice: which slab form the wave uses can only change at a refill, and a loop
        // that holds one form has neither the test nor the merges of the two forms' results in every step.
        auto node_phase = [&](auto fast_form) {
        constexpr bool FAST = decltype(fast_form)::value;
        while (true) {
            const int walking = __popcll(lane_mask(cur >= 0));
            if (walking == 0) break;
            if (walking * leave_k < __popcll(lane_mask((unsigned int)cur > (unsigned int)RAY_DONE))) break;
            if (COUNT) n_wave_steps += lane == 0 ? 1 : 0;
            if (!WIDE && cur >= 0) {
                float4 q0, q1, q2, q3;
                if (ALL_LDS || cur < staged) {
                    const float4* np = lds_item(lds_nodes, cur);
                    q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
#ifdef PT_PROBE_EXTRA_LDS
                    // measurement build: PT_PROBE_EXTRA_LDS more 4-byte LDS reads per step of a staged node
#pragma unroll
                    for (int k = 0; k < PT_PROBE_EXTRA_LDS; k++) {
                        const float x = reinterpret_cast<const volatile float*>(np)[k & 15];
                        asm volatile("" ::"v"(x));
                    }
#endif
                } else {
                    const float4* np = sc.nodes + 4 * (size_t)cur;
                    q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
#ifdef PT_PROBE_EXTRA_FETCH
                    // measurement build: PT_PROBE_EXTRA_FETCH more loads per step of a node in memory, from the node's
                    // own 64-byte line through a second pointer (so that they are not merged with the loads above);
                    // the values are only kept alive, the walk does not change.  PT_PROBE_FETCH_DWORD: 4 bytes a lane
                    // instead of 16; PT_PROBE_FETCH_ONE_LANE: only the first of the lanes that step loads.
                    const float4* alias = sc.probe_alias + 4 * (size_t)cur;
#pragma unroll
                    for (int k = 0; k < PT_PROBE_EXTRA_FETCH; k++) {
#ifdef PT_PROBE_FETCH_ONE_LANE
                        if (lane != __ffsll((long long)lane_mask(true)) - 1) continue;
#endif
#ifdef PT_PROBE_FETCH_DWORD
                        const float x1 = reinterpret_cast<const float*>(alias)[4 * (k & 3)];
                        asm volatile("" ::"v"(x1));
#else
                        const float4 x = alias[k & 3];
                        asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
#endif
                    }
#endif
                }
#ifdef PT_PROBE_EXTRA_SALU
                // measurement build: PT_PROBE_EXTRA_SALU more scalar instructions per node step
                {
                    unsigned int sx = 0;
#pragma unroll
                    for (int k = 0; k < PT_PROBE_EXTRA_SALU; k++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
                    asm volatile("" ::"s"(sx));
                }
#endif
#ifdef PT_PROBE_EXTRA_VALU
                // measurement build: PT_PROBE_EXTRA_VALU more vector instructions per node step (four independent
                // chains of fused multiply-adds on values the walk does not use), issued while the node's loads are
                // in flight
                {
                    float c0 = o.x, c1 = o.y, c2 = o.z, c3 = d.x;
#pragma unroll
                    for (int k = 0; k < PT_PROBE_EXTRA_VALU / 4; k++) {
                        c0 = __builtin_fmaf(c0, d.y, inv.x); c1 = __builtin_fmaf(c1, d.z, inv.y);
                        c2 = __builtin_fmaf(c2, d.x, inv.z); c3 = __builtin_fmaf(c3, d.y, inv.x);
                    }
                    asm volatile("" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3));
                }
#endif
                if (COUNT) n_box += 2;
                NodeStep st;
                if (FAST) node_step_recip<KD>(q0, q1, q2, q3, o, d, inv, best.t, st);
                else node_step<false, KD>(q0, q1, q2, q3, o, d, inv, best.t, st);
                // the deferred child goes to the row above the top whether or not it is pushed (that row
                // is dead until it is), or to the spare row once the lane's LDS rows are full
                const int row = min(sp, LDS_STACK);
                s_stack[row * 256 + tid] = st.deferred;
                if (KD) s_stack_t[row * 256 + tid] = st.deferred_t;
                if (__builtin_expect(lane_mask(st.push && sp >= LDS_STACK) != 0ull, 0)) {
                    if (st.push && sp >= LDS_STACK) {
                        deep[sp - LDS_STACK] = st.deferred;
                        if (KD) deep_t[sp - LDS_STACK] = st.deferred_t;
                    }
                }
                sp += st.push ? 1 : 0;
                pop_or_finish(!st.descend);
                cur = st.descend ? st.next : cur;
                leaf_k = 0;
            }
            if (WIDE && cur >= 0) {  // the same visit, and the child's from the record that came with it
                float4 q0, q1, q2, q3;
                float4 kid[8];       // WIDE: the node records of the two children, loaded with the node
                bool have_kids = false;
                if (ALL_LDS || cur < staged) {
                    const float4* np = lds_item(lds_nodes, cur);
                    q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
                } else if (WIDE) {
                    const float4* np = sc.wide + 12 * (size_t)cur;
                    q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
#pragma unroll
                    for (int k = 0; k < 8; k++) kid[k] = np[4 + k];
                    have_kids = true;
                } else {
                    const float4* np = sc.nodes + 4 * (size_t)cur;
                    q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
                }
                // One visit of an interior node: the reference's decisions, the deferred child to the stack, then
                // either the child to enter or the next deferred one.
                auto visit = [&](const float4& n0, const float4& n1, const float4& n2, const float4& n3) {
                    if (COUNT) n_box += 2;
                    NodeStep st;
                    if (FAST) node_step_recip<KD>(n0, n1, n2, n3, o, d, inv, best.t, st);
                    else node_step<false, KD>(n0, n1, n2, n3, o, d, inv, best.t, st);
                    // the deferred child goes to the row above the top whether or not it is pushed (that row
                    // is dead until it is), or to the spare row once the lane's LDS rows are full
                    const int row = min(sp, LDS_STACK);
                    s_stack[row * 256 + tid] = st.deferred;
                    if (KD) s_stack_t[row * 256 + tid] = st.deferred_t;
                    if (__builtin_expect(lane_mask(st.push && sp >= LDS_STACK) != 0ull, 0)) {
                        if (st.push && sp >= LDS_STACK) {
                            deep[sp - LDS_STACK] = st.deferred;
                            if (KD) deep_t[sp - LDS_STACK] = st.deferred_t;
                        }
                    }
                    sp += st.push ? 1 : 0;
                    pop_or_finish(!st.descend);
                    cur = st.descend ? st.next : cur;
                };
                visit(q0, q1, q2, q3);
                if (WIDE) {
                    // the child just entered is an interior node whose record came with its parent's: its visit now
                    // (a lane that popped instead stands on an older entry of its stack, never on a child of this node)
                    const int left_ref = __float_as_int(q3.x), right_ref = __float_as_int(q3.y);
                    const bool second = have_kids && cur >= 0 && (cur == left_ref || cur == right_ref);
                    if (lane_mask(second) != 0ull) {
                        if (second) {
                            const bool is_left = cur == left_ref;
                            const float4 c0 = is_left ? kid[0] : kid[4], c1 = is_left ? kid[1] : kid[5];
                            const float4 c2 = is_left ? kid[2] : kid[6], c3 = is_left ? kid[3] : kid[7];
                            visit(c0, c1, c2, c3);
                        }
                    }
                }
                leaf_k = 0;
            }
        }
        };
        // one form for the whole wave: the division form is right for every ray
        if (SLAB == SLAB_RECIP && lane_mask(cur > RAY_DONE && !fast_ok) == 0ull) node_phase(std::true_type());
        else node_phase(std::false_type());

        // ---- leaf phase: every lane on a leaf tests up to LEAF_BURST of its triangles, in stored
        // order (src/cs488.h:1677-1681, 1350-1354); a larger leaf is finished in the next round, so
        // one big leaf does not hold the other lanes.  A lane whose leaf ends earlier repeats its
        // last record with the result switched off, so the rounds need no per-lane control flow.
        if ((unsigned int)cur > (unsigned int)RAY_DONE) {
            const int off = ~cur;
            if (KD && !sc.leaf_inline) {
                const int n_here = ALL_LDS ? lds_refs[off] : sc.refs[off];
                leaf_n = leaf_k == 0 ? n_here : leaf_n;
#pragma unroll
                for (int j = 0; j < LEAF_BURST; j++) {
                    const bool valid = leaf_k + j < leaf_n;
                    if (j > 0 && lane_mask(valid) == 0ull) break;
                    const int k = valid ? leaf_k + j : leaf_n - 1;
                    if (COUNT) n_tri += valid ? 1 : 0;
                    if (COUNT) n_wave_rounds += lane == __ffsll((long long)lane_mask(true)) - 1 ? 1 : 0;
#ifdef PT_KD_MAILBOX
                    // measurement build: a ray remembers the triangles it has tested (a kd-tree repeats a triangle in
                    // every leaf it spans; the repeat cannot win, the test accepts only a strictly nearer hit) --
                    // one bit per triangle, scenes of at most 64 triangles held in LDS.  A round is skipped when no
                    // lane of the wave has an untested triangle in it.
                    if (ALL_LDS && sc.n_tris <= 64) {
                        const int id = lds_refs[off + 1 + k];
                        const bool test = valid && ((tested >> id) & 1ull) == 0ull;
                        tested |= test ? (1ull << id) : 0ull;
                        if (COUNT) n_tri -= valid && !test ? 1 : 0;
                        if (lane_mask(test) != 0ull) {
                            const float4* rec = lds_item(lds_records, id);
                            triangle_round<SLAB == SLAB_RECIP>(rec[0], rec[1], rec[2], rec[3], o, d, test, best);
                        }
                        continue;
                    }
#endif
                    const float4* rec = ALL_LDS ? lds_item(lds_records, lds_refs[off + 1 + k])
                                                : sc.isect + 4 * (size_t)sc.refs[off + 1 + k];
                    triangle_round<SLAB == SLAB_RECIP>(rec[0], rec[1], rec[2], rec[3], o, d, valid, best);
                }
            } else {
#pragma unroll
                for (int j = 0; j < LEAF_BURST; j++) {
                    const bool valid = j == 0 || leaf_k + j < leaf_n;
                    if (j > 0 && lane_mask(valid) == 0ull) break;
                    const int k = valid ? leaf_k + j : leaf_n - 1;
                    const float4* rec = ALL_LDS ? lds_item(lds_records, off + k) : sc.isect + 4 * (size_t)(off + k);
                    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                    if (j == 0) leaf_n = leaf_k == 0 ? __float_as_int(r1.w) : leaf_n;  // a leaf's first record carries its size
                    if (COUNT) n_tri += valid ? 1 : 0;
                    if (COUNT) n_wave_rounds += lane == __ffsll((long long)lane_mask(true)) - 1 ? 1 : 0;
                    triangle_round<SLAB == SLAB_RECIP>(r0, r1, r2, r3, o, d, valid, best);
                }
            }
            leaf_k += LEAF_BURST;
            pop_or_finish(leaf_k >= leaf_n);
            leaf_k = leaf_k >= leaf_n ? 0 : leaf_k;
        }
    }

    write_finished();
#ifdef PT_TIMELINE
    if (lane == 0) s_stat[8 + (tid >> 6)] = (unsigned int)wall_clock64();
#endif
    // statistics: wave sums -> LDS -> this workgroup's own slots (no global atomics)
    unsigned int tot_hit = n_hit, tot_box = n_box, tot_tri = n_tri, tot_steps = n_wave_steps, tot_rounds = n_wave_rounds;
    for (int off = 32; off > 0; off >>= 1) {
        tot_hit += __shfl_down(tot_hit, off);
        if (COUNT) {
            tot_box += __shfl_down(tot_box, off); tot_tri += __shfl_down(tot_tri, off);
            tot_steps += __shfl_down(tot_steps, off); tot_rounds += __shfl_down(tot_rounds, off);
        }
    }
    if (lane == 0) {
        atomicAdd(&s_stat[0], tot_hit);
        if (COUNT) {
            atomicAdd(&s_stat[1], tot_box); atomicAdd(&s_stat[2], tot_tri);
            atomicAdd(&s_stat[4], tot_steps); atomicAdd(&s_stat[5], tot_rounds);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* st = block_stats + (size_t)blk * DC_BLOCK_COUNT;
        st[DB_HITS] += s_stat[0];
        if (COUNT) {
            st[DB_BOX] += s_stat[1]; st[DB_TRI] += s_stat[2];
            st[DB_WAVE_STEPS] += s_stat[4]; st[DB_WAVE_ROUNDS] += s_stat[5];
        }
    }
}


}  // namespace trace_dev

}  // namespace pt
