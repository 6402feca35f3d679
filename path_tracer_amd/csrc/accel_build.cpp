// accel_build.cpp -- CPU builders of the two accelerators, producing the reference's trees
// node for node so that traversal visits triangles in the reference's order.
//
// BVH: BVH::build / splitBVH / sortAxis (src/cs488.h:1638-1667, 1490-1634, 1455-1486).
// kd-tree: KdTree::build / buildRec / findPlane (src/cs488.h:1388-1400, 1296-1320, 1249-1294).
// The kd builder evaluates the same candidate planes with the same cost arithmetic in the same
// order, but counts the triangles on each side by binary search in per-axis sorted lists
// (O(n log n) a node) where the reference recounts every triangle for every candidate (O(n^2)).
//
// Both builders are recursive and the two subtrees of a node do not depend on each other, so near
// the root (large subtrees, first few levels) the left subtree is built by another thread into a
// tree of its own and spliced in afterwards at the position the sequential recursion would have
// given it: node and reference numbering are exactly those of the single-threaded build.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <future>
#include <thread>

#include "scene.h"

namespace pt {

namespace {

// AABB with the reference's update and area arithmetic (src/cs488.h:491-527).
struct Bounds {
    vec3 lo = {FLT_MAX, FLT_MAX, FLT_MAX};
    vec3 hi = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    vec3 size = {0.0f, 0.0f, 0.0f};
    void grow(vec3 p) {
        if (lo.x > p.x) lo.x = p.x;
        if (lo.y > p.y) lo.y = p.y;
        if (lo.z > p.z) lo.z = p.z;
        if (hi.x < p.x) hi.x = p.x;
        if (hi.y < p.y) hi.y = p.y;
        if (hi.z < p.z) hi.z = p.z;
        size = hi - lo;
    }
    float area() const { return 2.0f * (size.x * size.y + size.y * size.z + size.z * size.x); }
    void store(float* box6) const {
        box6[0] = lo.x; box6[1] = lo.y; box6[2] = lo.z;
        box6[3] = hi.x; box6[4] = hi.y; box6[5] = hi.z;
    }
};

// Subtrees of at least this many triangles, at most this deep, get a thread of their own.
enum { PARALLEL_MIN_TRIS = 4096, PARALLEL_MAX_DEPTH = 4 };

bool worth_a_thread(int n, int depth, int threads) {
    if (threads == 0) threads = (int)std::thread::hardware_concurrency();
    return threads > 1 && n >= PARALLEL_MIN_TRIS && depth < PARALLEL_MAX_DEPTH;
}

// Appends subtree `sub` (node 0 = its root, children by index, leaves by first/count into its own
// refs) to `tree`, as the nodes [at, at + sub.nodes.size()).
void splice(HostTree& tree, const HostTree& sub, int at, int skip_root, int root_id) {
    // node k of `sub` becomes at + k - skip_root, except its root when that already exists as root_id
    const int ref_base = (int)tree.refs.size();
    auto renumber = [&](int k) { return (skip_root && k == 0) ? root_id : at + k - skip_root; };
    for (size_t k = 0; k < sub.nodes.size(); k++) {
        HostTree::Node nd = sub.nodes[k];
        if (nd.left >= 0) { nd.left = renumber(nd.left); nd.right = renumber(nd.right); }
        else nd.first += ref_base;
        if (skip_root && k == 0) tree.nodes[root_id] = nd; else tree.nodes.push_back(nd);
    }
    tree.refs.insert(tree.refs.end(), sub.refs.begin(), sub.refs.end());
}

struct BvhShared {
    const HostMesh& mesh;
    std::vector<float> centroid[3];
    int threads;
    BvhShared(const HostMesh& m, int threads_) : mesh(m), threads(threads_) {
        const int n = mesh.num_tris();
        for (int a = 0; a < 3; a++) centroid[a].resize(n);
        for (int i = 0; i < n; i++) {
            vec3 c = (mesh.p(i, 0) + mesh.p(i, 1) + mesh.p(i, 2)) * (1.0f / 3.0f);  // src/cs488.h:693
            centroid[0][i] = c.x; centroid[1][i] = c.y; centroid[2][i] = c.z;
        }
    }
};

struct BvhBuilder {
    const HostMesh& mesh;
    const std::vector<float>* centroid;
    HostTree& tree;
    const BvhShared& shared;
    int deepest = 0;

    BvhBuilder(const BvhShared& s, HostTree& t) : mesh(s.mesh), centroid(s.centroid), tree(t), shared(s) {}

    void grow_tri(Bounds& b, int t) const { b.grow(mesh.p(t, 0)); b.grow(mesh.p(t, 1)); b.grow(mesh.p(t, 2)); }

    // In-place Hoare-partition quicksort with the middle element as pivot.  The result for equal
    // keys depends on this exact exchange sequence, and leaf order decides ties between
    // triangles hit at the same distance, so no library sort can stand in for it.
    void order_by(int* ids, const std::vector<float>& key, int first, int last) const {
        while (true) {
            int i = first, j = last;
            const float pivot = key[ids[(first + last) / 2]];
            while (true) {
                while (key[ids[i]] < pivot) ++i;
                while (key[ids[j]] > pivot) --j;
                if (i >= j) break;
                std::swap(ids[i], ids[j]);
                ++i; --j;
            }
            // left part [first, i-1], right part [j+1, last]; recurse left, loop on right
            if (first < i - 1) order_by(ids, key, first, i - 1);
            if (j + 1 < last) { first = j + 1; } else { break; }
        }
    }

    int emit(std::vector<int>& ids, const Bounds& box, int depth) {
        const int n = (int)ids.size();
        if (depth > deepest) deepest = depth;
        const float parent_area = box.area();
        float best_cost = INFINITY;
        int best_axis = -1, best_cut = -1;
        std::vector<float> left_area(n), right_area(n);
        for (int axis = 0; axis < 3; axis++) {
            order_by(ids.data(), centroid[axis], 0, n - 1);
            Bounds l, r;
            for (int s = 0; s < n; s++) {
                grow_tri(l, ids[s]);
                if (s > 0) grow_tri(r, ids[n - s]);
                left_area[s] = l.area();
                right_area[n - s - 1] = r.area();
            }
            for (int i = 0; i < n; i++) {
                // counts i and n-i are the reference's own (src/cs488.h:1551)
                const float c = 2 * 1.0f + (left_area[i] * i + right_area[i] * (n - i)) * 1.0f / parent_area;
                if (c < best_cost) { best_cost = c; best_cut = i; best_axis = axis; }
            }
        }
        if (best_axis >= 0) order_by(ids.data(), centroid[best_axis], 0, n - 1);

        const int id = (int)tree.nodes.size();
        tree.nodes.push_back(HostTree::Node());
        box.store(tree.nodes[id].box);
        if (n <= 4 || best_cost > n) {  // src/cs488.h:1567
            tree.nodes[id].first = (int)tree.refs.size();
            tree.nodes[id].count = n;
            tree.refs.insert(tree.refs.end(), ids.begin(), ids.end());
            return id;
        }
        std::vector<int> lo_ids(ids.begin(), ids.begin() + best_cut + 1);
        std::vector<int> hi_ids(ids.begin() + best_cut + 1, ids.end());
        Bounds lb, rb;
        for (int t : lo_ids) grow_tri(lb, t);
        for (int t : hi_ids) grow_tri(rb, t);
        if (worth_a_thread(n, depth, shared.threads)) {
            // the right subtree in this thread, into a tree of its own as well: its place in the
            // numbering is known only when the left one is complete
            HostTree left_tree, right_tree;
            BvhBuilder lb_builder(shared, left_tree), rb_builder(shared, right_tree);
            auto left_done = std::async(std::launch::async, [&]() { lb_builder.emit(lo_ids, lb, depth + 1); });
            rb_builder.emit(hi_ids, rb, depth + 1);
            left_done.get();
            const int l = (int)tree.nodes.size();
            splice(tree, left_tree, l, 0, -1);
            const int r = (int)tree.nodes.size();
            splice(tree, right_tree, r, 0, -1);
            tree.nodes[id].left = l;
            tree.nodes[id].right = r;
            deepest = std::max(deepest, std::max(lb_builder.deepest, rb_builder.deepest));
            return id;
        }
        const int l = emit(lo_ids, lb, depth + 1);
        tree.nodes[id].left = l;
        const int r = emit(hi_ids, rb, depth + 1);
        tree.nodes[id].right = r;
        return id;
    }
};

struct KdShared {
    std::vector<float> tmin[3], tmax[3];
    int threads;
    KdShared(const HostMesh& mesh, int threads_) : threads(threads_) {
        const int n = mesh.num_tris();
        for (int a = 0; a < 3; a++) {
            tmin[a].resize(n); tmax[a].resize(n);
            for (int i = 0; i < n; i++) {
                const float c0 = comp(mesh.p(i, 0), a), c1 = comp(mesh.p(i, 1), a), c2 = comp(mesh.p(i, 2), a);
                tmin[a][i] = std::min(std::min(c0, c1), c2);
                tmax[a][i] = std::max(std::max(c0, c1), c2);
            }
        }
    }
};

struct KdBuilder {
    const KdShared& shared;
    const std::vector<float>* tmin;
    const std::vector<float>* tmax;
    HostTree& tree;
    int deepest = 0;

    KdBuilder(const KdShared& s, HostTree& t) : shared(s), tmin(s.tmin), tmax(s.tmax), tree(t) {}

    static Bounds from_box(const float* b) {
        Bounds r;
        r.lo = v3(b[0], b[1], b[2]); r.hi = v3(b[3], b[4], b[5]);
        r.size = r.hi - r.lo;
        return r;
    }
    // the two halves of a box cut at `value` (src/cs488.h:1239-1246)
    static void halves(const Bounds& b, int axis, float value, Bounds& l, Bounds& r) {
        vec3 lmax = b.hi, rmin = b.lo;
        if (axis == 0) { lmax.x = value; rmin.x = value; } else if (axis == 1) { lmax.y = value; rmin.y = value; } else { lmax.z = value; rmin.z = value; }
        l = Bounds(); r = Bounds();
        l.grow(b.lo); l.grow(lmax);
        r.grow(rmin); r.grow(b.hi);
    }
    static float split_cost(float parent, float a1, float a2, int n1, int n2) {  // src/cs488.h:1229-1233
        float cost = 2 * 0.1f + (a1 / parent) * n1 * 1.0f + (a2 / parent) * n2 * 1.0f;
        float lambda = (std::min(n1, n2) == 0 ? 0.8f : 1.0f);
        return lambda * cost;
    }

    void grow(int id, std::vector<int>& ids, int depth) {
        if (depth > deepest) deepest = depth;
        const int n = (int)ids.size();
        bool split = false;
        int axis_best = 0;
        float value_best = 0.0f;
        const Bounds box = from_box(tree.nodes[id].box);
        if (n > 1) {
            float cost_best = FLT_MAX;
            const float parent_area = box.area();
            std::vector<float> lows(n), highs(n);
            for (int axis = 0; axis < 3; axis++) {
                for (int k = 0; k < n; k++) { lows[k] = tmin[axis][ids[k]]; highs[k] = tmax[axis][ids[k]]; }
                std::sort(lows.begin(), lows.end());
                std::sort(highs.begin(), highs.end());
                const float bmin = comp(box.lo, axis), bmax = comp(box.hi, axis);
                for (int k = 0; k < n; k++) {
                    for (int side = 0; side < 2; side++) {
                        const float v = side == 0 ? tmin[axis][ids[k]] : tmax[axis][ids[k]];
                        if (v - PT_EPSILON < bmin || v + PT_EPSILON > bmax) continue;  // src/cs488.h:1273
                        // triangles with min <= v go left, triangles with max > v go right (src/cs488.h:1280-1281)
                        const int n1 = (int)(std::upper_bound(lows.begin(), lows.end(), v) - lows.begin());
                        const int n2 = n - (int)(std::upper_bound(highs.begin(), highs.end(), v) - highs.begin());
                        Bounds b1, b2;
                        halves(box, axis, v, b1, b2);
                        const float c = split_cost(parent_area, b1.area(), b2.area(), n1, n2);
                        if (c < cost_best) { cost_best = c; axis_best = axis; value_best = v; }
                    }
                }
            }
            split = !(cost_best >= (float)n * 1.0f);  // src/cs488.h:1306
        }
        if (!split) {
            tree.nodes[id].first = (int)tree.refs.size();
            tree.nodes[id].count = n;
            tree.refs.insert(tree.refs.end(), ids.begin(), ids.end());
            return;
        }
        const int l = (int)tree.nodes.size();
        tree.nodes.push_back(HostTree::Node());
        tree.nodes.push_back(HostTree::Node());
        tree.nodes[id].left = l;
        tree.nodes[id].right = l + 1;
        Bounds lb, rb;
        halves(box, axis_best, value_best, lb, rb);
        lb.store(tree.nodes[l].box);
        rb.store(tree.nodes[l + 1].box);
        std::vector<int> lids, rids;
        for (int t : ids) {
            if (tmin[axis_best][t] <= value_best) lids.push_back(t);
            if (tmax[axis_best][t] > value_best) rids.push_back(t);
        }
        std::vector<int>().swap(ids);  // the parent's list is not needed below this point
        if (worth_a_thread(n, depth, shared.threads)) {
            // each child grows in a tree of its own whose node 0 is a copy of the child; the
            // descendants are appended in the sequential order: all of the left child's, then the right's
            HostTree left_tree, right_tree;
            left_tree.nodes.push_back(tree.nodes[l]);
            right_tree.nodes.push_back(tree.nodes[l + 1]);
            KdBuilder lbuild(shared, left_tree), rbuild(shared, right_tree);
            auto left_done = std::async(std::launch::async, [&]() { lbuild.grow(0, lids, depth + 1); });
            rbuild.grow(0, rids, depth + 1);
            left_done.get();
            splice(tree, left_tree, (int)tree.nodes.size(), 1, l);
            splice(tree, right_tree, (int)tree.nodes.size(), 1, l + 1);
            deepest = std::max(deepest, std::max(lbuild.deepest, rbuild.deepest));
            return;
        }
        grow(l, lids, depth + 1);
        grow(l + 1, rids, depth + 1);
    }
};

}  // namespace

void build_bvh(const HostMesh& mesh, HostTree& tree, int threads) {
    tree = HostTree();
    tree.kind = 0;
    const int n = mesh.num_tris();
    if (n == 0) return;
    BvhShared shared(mesh, threads);
    BvhBuilder b(shared, tree);
    std::vector<int> ids(n);
    Bounds box;
    for (int i = 0; i < n; i++) { ids[i] = i; b.grow_tri(box, i); }
    tree.nodes.reserve(2 * (size_t)n);
    b.emit(ids, box, 0);
    tree.max_depth = b.deepest;
}

void build_kd(const HostMesh& mesh, HostTree& tree, int threads) {
    tree = HostTree();
    tree.kind = 1;
    const int n = mesh.num_tris();
    if (n == 0) return;
    KdShared shared(mesh, threads);
    KdBuilder b(shared, tree);
    std::vector<int> ids(n);
    Bounds box;
    for (int i = 0; i < n; i++) {
        ids[i] = i;
        box.grow(mesh.p(i, 0)); box.grow(mesh.p(i, 1)); box.grow(mesh.p(i, 2));
    }
    tree.nodes.push_back(HostTree::Node());
    box.store(tree.nodes[0].box);
    b.grow(0, ids, 0);
    tree.max_depth = b.deepest;
}

}  // namespace pt
