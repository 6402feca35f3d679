// pt_oracle.cpp -- CPU restatement of the CS488 path tracer's per-pixel sample loop.
//
// THIS FILE IS TEST INFRASTRUCTURE.  It is the checker that the HIP path is compared with;
// it is never linked into, loaded by or called from the product library.  Only tests/, the
// smoke check and the cpu_baseline leg of bench.py may load it.
//
// Parity status: the reference cannot be built in this environment (its only translation unit
// includes <GL/glew.h> and <GLFW/glfw3.h>, which are absent, and stand-ins are not allowed) and
// it ships no tests or golden vectors.  Bit-level parity of this restatement is therefore
// UNPINNED; it is pinned only statistically, against the converged screenshots the reference
// holds under project_demo/ (see tests/test_oracle_screenshots.py).
//
// Every function cites the reference lines it follows (paths relative to the reference root).
// Arithmetic follows the reference's evaluation order in IEEE binary32; build with
// -ffp-contract=off and without -ffast-math so that no FMA or reassociation appears.
//
// Deliberate differences from the reference (all outside the arithmetic of one sample):
//   * resolution, samples per pixel, thread count, camera and accelerator kind are run-time
//     parameters instead of compile-time constants (src/cs488.h:48-49,98,101,1197);
//   * HitInfo additionally records the triangle index and barycentrics of the winner (the
//     reference has no triangle id, src/cs488.h:464-476);
//   * with more than one thread every column strip owns a private PCG32 stream (the reference
//     shares one unsynchronised global state between its threads, src/cs488.h:131,1887, which is
//     a data race); with one thread the single global stream of the reference is reproduced;
//   * the environment lookup clamps its texel index (the reference indexes out of bounds when
//     d.x = d.y = 0, src/cs488.h:1721-1726);
//   * images and textures are handed in as decoded pixels by the caller; this file has no
//     PNG or Radiance decoder (the reference uses stb_image, src/cs488.h:202,841).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------------------------
// vector math, evaluation order of src/linalg.h
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
static inline V3 mk(float a, float b, float c) { V3 r = {a, b, c}; return r; }
static inline V3 mk(float a) { V3 r = {a, a, a}; return r; }
static inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
static inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
// src/linalg.h:367,401,501 -- dot = sum(a*b), sum folds from a zero scalar, left to right
static inline float dot(V3 a, V3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
// src/linalg.h:500
static inline V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// src/linalg.h:503-504
static inline float length(V3 a) { return sqrtf(dot(a, a)); }
static inline V3 normalize(V3 a) { return a / length(a); }

static const float kPi = 3.14159265358979f;   // src/cs488.h:53
static const float kDegToRad = kPi / 180.0f;  // src/cs488.h:54
static const float kEps = 5e-5f;              // src/cs488.h:59
static const float kEtaAir = 1.0f;            // src/cs488.h:61

// ---------------------------------------------------------------------------------------------
// PCG32 (pcg32_fast), src/cs488.h:130-143
// ---------------------------------------------------------------------------------------------
struct Rng {
    uint64_t state;
    uint32_t next_u32() {
        uint64_t x = state;
        const unsigned count = (unsigned)(x >> 61);
        state = x * 6364136223846793005ull;
        x ^= x >> 22;
        return (uint32_t)(x >> (22 + count));
    }
    float next() { return float(double(next_u32()) / 4294967296.0); }
};
static const uint64_t kDefaultSeed = 0xcafef00dd15ea5e5ull;  // src/cs488.h:131

struct Counters {
    uint64_t rays = 0, box_tests = 0, tri_tests = 0, nan_samples = 0;
    uint64_t shade_calls[20] = {0};
};
static thread_local Rng* tl_rng = nullptr;
static thread_local Counters* tl_cnt = nullptr;
static inline float rnd() { return tl_rng->next(); }

// ---------------------------------------------------------------------------------------------
// scene types, src/cs488.h:351-356 (Ray), :363-389 (Material), :464-476 (HitInfo),
// :481-568 (AABB), :574-581 (Triangle)
// ---------------------------------------------------------------------------------------------
struct Ray { V3 o, d; };

enum { LAMBERTIAN = 0, METAL = 1, GLASS = 2 };

struct Material {
    std::string name;
    int type = LAMBERTIAN;
    float eta = 1.0f;
    V3 Ka = mk(0.0f), Kd = mk(0.9f), Ks = mk(0.0f), Ke = mk(0.0f);
    float Ns = 0.0f, opacity = 1.0f, roughness = 0.6f;
    bool textured = false;
    std::string tex_file;
    std::vector<uint8_t> tex;  // 3 x u8, row 0 first as decoded
    int tw = 0, th = 0;

    // src/cs488.h:402-414
    V3 fetch(float u, float v) const {
        int x = int(u * tw) % tw;
        int y = int(v * th) % th;
        if (x < 0) x += tw;
        if (y < 0) y += th;
        const int pix = (x + y * tw) * 3;
        return mk((float)tex[pix], (float)tex[pix + 1], (float)tex[pix + 2]) / 255.0f;
    }
    // src/cs488.h:417-428: only the Lambertian case is non-zero
    V3 brdf() const { return type == LAMBERTIAN ? Kd / kPi : mk(0.0f); }
};

struct Box {
    V3 lo = mk(FLT_MAX), hi = mk(-FLT_MAX), size = mk(0.0f);
    void fit(V3 v) {  // src/cs488.h:513-523
        if (lo.x > v.x) lo.x = v.x;
        if (lo.y > v.y) lo.y = v.y;
        if (lo.z > v.z) lo.z = v.z;
        if (hi.x < v.x) hi.x = v.x;
        if (hi.y < v.y) hi.y = v.y;
        if (hi.z < v.z) hi.z = v.z;
        size = hi - lo;
    }
    float area() const {  // src/cs488.h:525-527
        return 2.0f * (size.x * size.y + size.y * size.z + size.z * size.x);
    }
    // src/cs488.h:530-567: slab test with true divisions; NaNs compare false and fall through
    bool hit(const Ray& r, float& t_entry) const {
        if (tl_cnt) tl_cnt->box_tests++;
        float tx1 = (lo.x - r.o.x) / r.d.x, ty1 = (lo.y - r.o.y) / r.d.y, tz1 = (lo.z - r.o.z) / r.d.z;
        float tx2 = (hi.x - r.o.x) / r.d.x, ty2 = (hi.y - r.o.y) / r.d.y, tz2 = (hi.z - r.o.z) / r.d.z;
        if (tx1 > tx2) { float s = tx1; tx1 = tx2; tx2 = s; }
        if (ty1 > ty2) { float s = ty1; ty1 = ty2; ty2 = s; }
        if (tz1 > tz2) { float s = tz1; tz1 = tz2; tz2 = s; }
        float t1 = tx1; if (t1 < ty1) t1 = ty1; if (t1 < tz1) t1 = tz1;
        float t2 = tx2; if (t2 > ty2) t2 = ty2; if (t2 > tz2) t2 = tz2;
        if (t1 > t2) return false;
        if ((t1 < 0.0) && (t2 < 0.0)) return false;
        t_entry = t1;
        return true;
    }
};

struct Tri {
    V3 p[3], n[3];
    float uv[3][2];
    int mat = 0;
    V3 center;
};

struct Hit {
    float t;
    V3 P, N;
    float u, v;
    const Material* mat;
    bool backface;
    int tri;            // not in the reference: index of the winning triangle
    float beta, gamma;  // not in the reference: its barycentrics
    V3 color() const {  // src/cs488.h:473-475
        return mat->textured ? mat->brdf() * mat->fetch(u, v) : mat->brdf();
    }
};

// src/cs488.h:622-624
static inline float det3(V3 a, V3 b, V3 c) { return dot(cross(a, b), c); }

struct Mesh {
    std::vector<Tri> tris;
    std::vector<Material> mats;
    std::string base_dir;

    // src/cs488.h:646-681
    bool hit_tri(Hit& out, const Ray& r, int idx, float tmin, float tmax) const {
        if (tl_cnt) tl_cnt->tri_tests++;
        const Tri& T = tris[idx];
        V3 a = T.p[0] - T.p[1];
        V3 b = T.p[0] - T.p[2];
        V3 c = r.d;
        V3 d = T.p[0] - r.o;
        float D = det3(a, b, c);
        if (D != 0) {
            float Da = det3(d, b, c);
            float Db = det3(a, d, c);
            float Dc = det3(a, b, d);
            float beta = Da / D;
            float gamma = Db / D;
            float alpha = 1 - beta - gamma;
            float t = Dc / D;
            if (0 <= alpha && alpha <= 1 && 0 <= beta && beta <= 1 && 0 <= gamma && gamma <= 1 &&
                tmin <= t && t <= tmax) {
                out.mat = &mats[T.mat];
                out.u = alpha * T.uv[0][0] + beta * T.uv[1][0] + gamma * T.uv[2][0];
                out.v = alpha * T.uv[0][1] + beta * T.uv[1][1] + gamma * T.uv[2][1];
                out.t = t;
                out.N = normalize(alpha * T.n[0] + beta * T.n[1] + gamma * T.n[2]);
                out.P = alpha * T.p[0] + beta * T.p[1] + gamma * T.p[2];
                out.tri = idx;
                out.beta = beta;
                out.gamma = gamma;
                return true;
            }
        }
        return false;
    }

    // src/cs488.h:685-699 (only the centre is used later; per-triangle boxes are never read)
    void precalc() {
        for (Tri& T : tris) T.center = (T.p[0] + T.p[1] + T.p[2]) * (1.0f / 3.0f);
    }

    // src/cs488.h:814-835
    void single_triangle() {
        tris.assign(1, Tri());
        mats.assign(1, Material());
        Tri& T = tris[0];
        T.mat = 0;
        T.p[0] = mk(-0.5f, -0.5f, 0.0f);
        T.p[1] = mk(0.5f, -0.5f, 0.0f);
        T.p[2] = mk(0.0f, 0.5f, 0.0f);
        V3 n = normalize(cross(T.p[1] - T.p[0], T.p[2] - T.p[0]));
        T.n[0] = T.n[1] = T.n[2] = n;
        T.uv[0][0] = 0; T.uv[0][1] = 0;
        T.uv[1][0] = 0; T.uv[1][1] = 1;
        T.uv[2][0] = 1; T.uv[2][1] = 0;
    }

    // src/cs488.h:854-915.  Lines are read through an 80-byte fgets; names keep their newline.
    void load_mtl(const std::string& path) {
        FILE* fp = fopen(path.c_str(), "r");
        if (!fp) return;  // the reference dereferences the null FILE* here
        Material cur;
        cur.name = "";
        char line[81];
        while (fgets(line, 80, fp) != nullptr) {
            std::string s = line;
            float r, g, b, k;
            if (s.compare(0, 6, "newmtl", 0, 6) == 0) {
                s.erase(0, 7);
                if (cur.name != "") mats.push_back(cur);
                cur = Material();
                cur.name = s;
            } else if (s.compare(0, 2, "Ka", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f %f %f", &r, &g, &b) == 3) cur.Ka = mk(r, g, b);
            } else if (s.compare(0, 2, "Kd", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f %f %f", &r, &g, &b) == 3) cur.Kd = mk(r, g, b);
            } else if (s.compare(0, 2, "Ks", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f %f %f", &r, &g, &b) == 3) cur.Ks = mk(r, g, b);
            } else if (s.compare(0, 2, "Ns", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f", &k) == 1) cur.Ns = k;
            } else if (s.compare(0, 6, "map_Kd", 0, 6) == 0) {
                s.erase(0, 7);
                s.erase(s.size() - 1, 1);
                cur.textured = true;
                cur.tex_file = base_dir + s;
            } else if (s.compare(0, 2, "Ke", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f %f %f", &r, &g, &b) == 3) cur.Ke = mk(r, g, b);
            } else if (s.compare(0, 2, "Op", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f", &k) == 1) cur.opacity = k;
            } else if (s.compare(0, 2, "Ro", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f", &k) == 1) cur.roughness = k;
            } else if (s.compare(0, 2, "Ri", 0, 2) == 0) {
                if (sscanf(s.c_str() + 3, "%f", &k) == 1) cur.eta = k;
            }
        }
        if (cur.name != "") mats.push_back(cur);
        fclose(fp);
    }

    // "v/t/n" splitter of src/cs488.h:919-948 (atoi of each field, missing fields read as 0)
    static void split_ref(char* word, int& vi, int& ti, int& ni) {
        const char* tp = " ";
        const char* np = " ";
        bool seen = false;
        for (char* q = word; *q != '\0'; q++) {
            if (*q == '/') {
                if (!seen) { tp = q + 1; seen = true; } else { np = q + 1; }
                *q = '\0';
            }
        }
        vi = atoi(word);
        ti = atoi(tp);
        ni = atoi(np);
    }

    // src/cs488.h:703-788 (load) and :917-1194 (ParseOBJ), merged: the reference expands every
    // face into three private vertices and then copies them into Triangle records.
    bool load(const char* path) {
        std::string p = path;
        size_t slash = p.find_last_of("/\\");
        base_dir = (slash != std::string::npos ? p.substr(0, slash) : std::string("")) + "/";
        FILE* fp = fopen(path, "r");
        if (!fp) return false;

        // first pass: material library (loaded as soon as its line is met) -- :965-986
        char line[81];
        while (fgets(line, 80, fp) != NULL) {
            std::string s = line;
            if (s.compare(0, 6, "mtllib", 0, 6) == 0) {
                s.erase(0, 7);
                s.erase(s.size() - 1, 1);
                load_mtl(base_dir + s);
            }
        }
        fseek(fp, 0, SEEK_SET);

        // second pass -- :1008-1113
        std::vector<float> v, n, t;
        std::vector<int> vi, ni, ti, mi;
        bool no_normals = false, no_uv = false, no_mats = true;
        int cur_mat = 0;
        while (fgets(line, 80, fp) != NULL) {
            std::string s = line;
            if (line[0] == 'v') {
                if (line[1] == 'n') {
                    float x = 0, y = 0, z = 0;
                    sscanf(&line[2], "%f %f %f\n", &x, &y, &z);
                    float l = sqrtf(x * x + y * y + z * z);
                    n.push_back(x / l); n.push_back(y / l); n.push_back(z / l);
                } else if (line[1] == 't') {
                    float a = 0, b = 0;
                    sscanf(&line[2], "%f %f\n", &a, &b);
                    t.push_back(a); t.push_back(b);
                } else {
                    float x = 0, y = 0, z = 0;
                    sscanf(&line[1], "%f %f %f\n", &x, &y, &z);
                    v.push_back(x); v.push_back(y); v.push_back(z);
                }
            }
            if (s.compare(0, 6, "usemtl", 0, 6) == 0) {
                s.erase(0, 7);
                for (size_t i = 0; i < mats.size(); i++) {
                    if (s.compare(mats[i].name) == 0) { cur_mat = (int)i; no_mats = false; break; }
                }
            } else if (line[0] == 'f') {
                char w[3][32];
                w[0][0] = w[1][0] = w[2][0] = '\0';
                sscanf(&line[1], "%31s %31s %31s\n", w[0], w[1], w[2]);
                mi.push_back(cur_mat);
                for (int k = 0; k < 3; k++) {
                    int a, b, c;
                    split_ref(w[k], a, b, c);
                    vi.push_back(a - 1);
                    if (c) ni.push_back(c - 1); else { ni.push_back(0); no_normals = true; }
                    if (b) ti.push_back(b - 1); else { ti.push_back(0); no_uv = true; }
                }
            }
        }
        fclose(fp);

        const int ntri = (int)mi.size();
        if (ntri == 0) return false;  // :714

        // material typing, :717-733
        if (!no_mats) {
            for (Material& m : mats) {
                m.type = LAMBERTIAN;
                if (m.Ns == 100.0f) m.type = METAL;
                if (m.name.compare(0, 5, "glass", 0, 5) == 0) { m.type = GLASS; m.eta = 1.5f; }
            }
        } else {
            mats.resize(1);
        }

        tris.assign(ntri, Tri());
        for (int i = 0; i < ntri; i++) {
            Tri& T = tris[i];
            for (int k = 0; k < 3; k++) {
                const int a = vi[3 * i + k];
                T.p[k] = mk(v[3 * a], v[3 * a + 1], v[3 * a + 2]);
            }
            if (!no_normals) {
                for (int k = 0; k < 3; k++) {
                    const int c = ni[3 * i + k];
                    T.n[k] = mk(n[3 * c], n[3 * c + 1], n[3 * c + 2]);
                }
            } else {  // :748-757
                V3 fn = normalize(cross(T.p[1] - T.p[0], T.p[2] - T.p[0]));
                T.n[0] = T.n[1] = T.n[2] = fn;
            }
            T.mat = 0;
            for (int k = 0; k < 3; k++) T.uv[k][0] = T.uv[k][1] = 0.0f;
            if (!no_mats) {
                if (!no_uv && mats[mi[i]].textured) {  // :763-771
                    for (int k = 0; k < 3; k++) {
                        const int b = ti[3 * i + k];
                        T.uv[k][0] = t[2 * b];
                        T.uv[k][1] = t[2 * b + 1];
                    }
                }
                T.mat = mi[i];
            }
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// SAH BVH, src/cs488.h:1409-1706
// ---------------------------------------------------------------------------------------------
struct BvhNode {
    bool leaf = false;
    int left = -1, right = -1;
    std::vector<int> list;
    Box box;
};

struct Bvh {
    const Mesh* mesh = nullptr;
    std::vector<BvhNode> nodes;

    // src/cs488.h:1455-1486 -- Hoare partition quicksort on centroids, pivot = middle element
    void sort_axis(int* idx, int axis, int li, int ri) const {
        int i = li, j = ri;
        const float pivot = mesh->tris[idx[(li + ri) / 2]].center[axis];
        while (true) {
            while (mesh->tris[idx[i]].center[axis] < pivot) ++i;
            while (mesh->tris[idx[j]].center[axis] > pivot) --j;
            if (i >= j) break;
            int s = idx[i]; idx[i] = idx[j]; idx[j] = s;
            ++i; --j;
        }
        if (li < (i - 1)) sort_axis(idx, axis, li, i - 1);
        if ((j + 1) < ri) sort_axis(idx, axis, j + 1, ri);
    }

    void fit_tri(Box& b, int tri) const {
        const Tri& T = mesh->tris[tri];
        b.fit(T.p[0]); b.fit(T.p[1]); b.fit(T.p[2]);
    }

    // src/cs488.h:1490-1634 (SAHBVH branch)
    int split(std::vector<int>& idx, const Box& box) {
        const int n = (int)idx.size();
        int best_axis = -1, best_i = -1;
        const float outer = box.area();
        float cmin = INFINITY;
        std::vector<float> areaL(n), areaR(n);
        std::vector<int> sorted(n);
        for (int axis = 0; axis <= 2; ++axis) {
            sort_axis(idx.data(), axis, 0, n - 1);
            sorted = idx;
            Box bl, br;
            for (int s = 0; s < n; ++s) {
                fit_tri(bl, sorted[s]);
                if (s > 0) fit_tri(br, sorted[n - s]);
                areaL[s] = bl.area();
                areaR[n - s - 1] = br.area();
            }
            for (int i = 0; i < n; ++i) {
                // cost exactly as written at :1551 (object counts i and n-i)
                float c = 2 * 1.0f + (areaL[i] * i + areaR[i] * (n - i)) * 1.0f / outer;
                if (c < cmin) { cmin = c; best_i = i; best_axis = axis; }
            }
        }
        if (best_axis >= 0) sort_axis(idx.data(), best_axis, 0, n - 1);  // :1561
        sorted = idx;
        const bool make_leaf = (n <= 4 || cmin > n);  // :1567

        const int id = (int)nodes.size();
        nodes.push_back(BvhNode());
        nodes[id].box = box;
        if (make_leaf) {
            nodes[id].leaf = true;
            nodes[id].list = idx;
            return id;
        }
        Box bl, br;
        std::vector<int> L(sorted.begin(), sorted.begin() + best_i + 1);
        std::vector<int> R(sorted.begin() + best_i + 1, sorted.end());
        for (int k : L) fit_tri(bl, k);
        for (int k : R) fit_tri(br, k);
        nodes[id].leaf = false;
        int l = split(L, bl);
        nodes[id].left = l;
        int r = split(R, br);
        nodes[id].right = r;
        return id;
    }

    // src/cs488.h:1638-1667
    void build(const Mesh* m) {
        mesh = m;
        nodes.clear();
        const int n = (int)m->tris.size();
        std::vector<int> idx(n);
        Box box;
        for (int i = 0; i < n; ++i) { idx[i] = i; fit_tri(box, i); }
        nodes.reserve(2 * n);
        split(idx, box);
    }

    // src/cs488.h:1671-1706
    void traverse(Hit& best, const Ray& r, int id, float tmin, float tmax) const {
        const BvhNode& nd = nodes[id];
        if (nd.leaf) {
            Hit h;
            for (int k : nd.list) {
                if (mesh->hit_tri(h, r, k, tmin, tmax)) {
                    if (h.t < best.t) best = h;
                }
            }
        } else {
            float tl = 0, tr = 0;
            bool hl = nodes[nd.left].box.hit(r, tl);
            bool hr = nodes[nd.right].box.hit(r, tr);
            hl = hl && (tl < best.t);
            hr = hr && (tr < best.t);
            if (hl && hr) {
                if (tl < tr) {
                    traverse(best, r, nd.left, tmin, tmax);
                    traverse(best, r, nd.right, tmin, tmax);
                } else {
                    traverse(best, r, nd.right, tmin, tmax);
                    traverse(best, r, nd.left, tmin, tmax);
                }
            } else if (hl) {
                traverse(best, r, nd.left, tmin, tmax);
            } else if (hr) {
                traverse(best, r, nd.right, tmin, tmax);
            }
        }
    }

    // src/cs488.h:1432-1444
    bool intersect(Hit& out, const Ray& r, float tmin, float tmax) const {
        out.t = FLT_MAX;
        float t0;
        if (nodes[0].box.hit(r, t0)) traverse(out, r, 0, tmin, tmax);
        return out.t != FLT_MAX;
    }
};

// ---------------------------------------------------------------------------------------------
// SAH kd-tree, src/cs488.h:1199-1407
// ---------------------------------------------------------------------------------------------
struct KdNode {
    Box box;
    std::vector<int> list;
    int left = -1, right = -1;  // indices into Kd::nodes; -1 = leaf
};

struct Kd {
    const Mesh* mesh = nullptr;
    std::vector<KdNode> nodes;  // creation order: parent, then (left,right) pair, then subtrees
    std::vector<float> tmin_[3], tmax_[3];

    // src/cs488.h:1239-1246
    static void split_box(const Box& b, Box& l, Box& r, int axis, float v) {
        V3 lmax = b.hi, rmin = b.lo;
        lmax[axis] = v;
        rmin[axis] = v;
        l.fit(b.lo); l.fit(lmax);
        r.fit(rmin); r.fit(b.hi);
    }
    // src/cs488.h:1229-1233
    static float cost_split(float ap, float a1, float a2, int n1, int n2) {
        float cost = 2 * 0.1f + (a1 / ap) * n1 * 1.0f + (a2 / ap) * n2 * 1.0f;
        float lambda = ((n1 < n2 ? n1 : n2) == 0 ? 0.8f : 1.0f);
        return lambda * cost;
    }
    // src/cs488.h:1249-1294: candidates in (axis, triangle, min-then-max) order, first strict
    // minimum wins; every candidate recounts the node's triangles
    void find_plane(const KdNode& nd, int& best_axis, float& best_v, float& best_cost) const {
        best_axis = 0; best_v = 0.0f; best_cost = FLT_MAX;
        const float ap = nd.box.area();
        for (int axis = 0; axis < 3; ++axis) {
            const std::vector<float>& lo = tmin_[axis];
            const std::vector<float>& hi = tmax_[axis];
            for (int k : nd.list) {
                for (int side = 0; side < 2; ++side) {
                    const float v = side == 0 ? lo[k] : hi[k];
                    if (v - kEps < nd.box.lo[axis] || v + kEps > nd.box.hi[axis]) continue;
                    Box b1, b2;
                    split_box(nd.box, b1, b2, axis, v);
                    int n1 = 0, n2 = 0;
                    for (int q : nd.list) {
                        if (lo[q] <= v) ++n1;
                        if (hi[q] > v) ++n2;
                    }
                    float c = cost_split(ap, b1.area(), b2.area(), n1, n2);
                    if (c < best_cost) { best_cost = c; best_axis = axis; best_v = v; }
                }
            }
        }
    }
    // src/cs488.h:1296-1320
    void build_rec(int id) {
        if (nodes[id].list.size() <= 1) return;
        int axis; float v, cost;
        find_plane(nodes[id], axis, v, cost);
        if (cost >= (float)nodes[id].list.size() * 1.0f) return;
        const int l = (int)nodes.size();
        nodes.push_back(KdNode());
        nodes.push_back(KdNode());
        const int r = l + 1;
        nodes[id].left = l;
        nodes[id].right = r;
        split_box(nodes[id].box, nodes[l].box, nodes[r].box, axis, v);
        for (int k : nodes[id].list) {
            if (tmin_[axis][k] <= v) nodes[l].list.push_back(k);
            if (tmax_[axis][k] > v) nodes[r].list.push_back(k);
        }
        build_rec(l);
        build_rec(r);
    }
    // src/cs488.h:1388-1400
    void build(const Mesh* m) {
        mesh = m;
        nodes.clear();
        const int n = (int)m->tris.size();
        for (int a = 0; a < 3; a++) {
            tmin_[a].resize(n); tmax_[a].resize(n);
            for (int i = 0; i < n; i++) {  // :1221-1227
                const Tri& T = m->tris[i];
                // std::min / std::max over a list keep the FIRST of equal elements (matters for +-0)
                tmin_[a][i] = std::min({T.p[0][a], T.p[1][a], T.p[2][a]});
                tmax_[a][i] = std::max({T.p[0][a], T.p[1][a], T.p[2][a]});
            }
        }
        nodes.push_back(KdNode());
        for (int i = 0; i < n; ++i) {
            const Tri& T = m->tris[i];
            nodes[0].box.fit(T.p[0]); nodes[0].box.fit(T.p[1]); nodes[0].box.fit(T.p[2]);
            nodes[0].list.push_back(i);
        }
        build_rec(0);
    }
    // src/cs488.h:1344-1379: like the BVH walk, but the far child is re-checked against the
    // current best before it is entered
    void traverse(Hit& best, const Ray& r, int id, float tmin, float tmax) const {
        const KdNode& nd = nodes[id];
        if (nd.left < 0) {
            Hit h;
            for (int k : nd.list) {
                if (mesh->hit_tri(h, r, k, tmin, tmax)) {
                    if (h.t < best.t) best = h;
                }
            }
        } else {
            float tl = 0, tr = 0;
            bool hl = nodes[nd.left].box.hit(r, tl);
            bool hr = nodes[nd.right].box.hit(r, tr);
            hl = hl && (tl < best.t);
            hr = hr && (tr < best.t);
            if (hl && hr) {
                if (tl < tr) {
                    traverse(best, r, nd.left, tmin, tmax);
                    if (tr < best.t) traverse(best, r, nd.right, tmin, tmax);
                } else {
                    traverse(best, r, nd.right, tmin, tmax);
                    if (tl < best.t) traverse(best, r, nd.left, tmin, tmax);
                }
            } else if (hl) {
                traverse(best, r, nd.left, tmin, tmax);
            } else if (hr) {
                traverse(best, r, nd.right, tmin, tmax);
            }
        }
    }
    // src/cs488.h:1402-1406
    bool intersect(Hit& out, const Ray& r, float tmin, float tmax) const {
        out.t = FLT_MAX;
        float t0;
        if (nodes.empty() || !nodes[0].box.hit(r, t0)) return false;
        traverse(out, r, 0, tmin, tmax);
        return out.t != FLT_MAX;
    }
};

// ---------------------------------------------------------------------------------------------
// scene, camera, shading: src/cs488.h:1709-1899 and :1902-2172
// ---------------------------------------------------------------------------------------------
struct Camera {
    V3 eye = mk(0.0f, 0.0f, 1.5f);     // src/cs488.h:83
    V3 lookat = mk(0.0f, 0.0f, 0.0f);  // :84
    V3 up = mk(0.0f, 1.0f, 0.0f);      // :85 (normalize of a unit vector)
    float fov = 45.0f;                 // :75
    float film = 0.032f;               // :78
};

struct Scene {
    Mesh mesh;
    Bvh bvh;
    Kd kd;
    int accel = 0;  // 0 = BVH, 1 = kd-tree
    bool built = false;
    std::vector<V3> env;  // row 0 = bottom row, as Image::load stores it (:210-215)
    int env_w = 0, env_h = 0;
    Camera cam;
    int max_level = 15;  // :60
    int width = 512, height = 384;  // :48-49
    std::vector<V3> accum, frame;
    int total_samples = 0;

    // src/cs488.h:1717-1727
    V3 ibl(const Ray& r) const {
        if (env_h == 0) return mk(0.0f);
        float rr = (1.0f / kPi) * acosf(r.d.z) /
                   sqrtf((float)(pow((double)r.d.x, 2.0) + pow((double)r.d.y, 2.0)));
        float u = r.d.x * rr;
        float v = r.d.y * rr;
        float fi = (u + 1.0f) * 0.5f * env_w;
        float fj = (v + 1.0f) * 0.5f * env_h;
        int i = (fi == fi) ? (int)fi : 0;  // the reference converts NaN/out-of-range unguarded
        int j = (fj == fj) ? (int)fj : 0;
        if (i < 0) i = 0; if (i > env_w - 1) i = env_w - 1;
        if (j < 0) j = 0; if (j > env_h - 1) j = env_h - 1;
        return env[i + j * env_w];
    }

    // src/cs488.h:1750-1777
    bool intersect(Hit& out, const Ray& r, float tmin = 0.0f, float tmax = FLT_MAX) const {
        if (tl_cnt) tl_cnt->rays++;
        bool hit = false;
        Hit tmp;
        out.t = FLT_MAX;
        bool got = accel == 1 ? kd.intersect(tmp, r, tmin, tmax) : bvh.intersect(tmp, r, tmin, tmax);
        if (got) {
            if (tmp.t < out.t) { hit = true; out = tmp; }
        }
        if (hit) {
            out.backface = dot(r.d, out.N) > 0;
            if (out.backface) out.N = -out.N;
        }
        return hit;
    }

    // src/cs488.h:1825-1839 with :74-79 (aspect, film distance); r1 is drawn before r2
    Ray camera_ray(int x, int y, bool jitter) const {
        const V3 view = normalize(cam.lookat - cam.eye);  // :2284
        const V3 w = normalize(-view);
        const V3 u = normalize(cross(cam.up, w));
        const V3 v = cross(w, u);
        const float aspect = float(width / float(height));
        const float dist = cam.film / (2.0f * tanf(cam.fov * kDegToRad * 0.5f));
        float U, V;
        if (jitter) {
            U = (x + (rnd() * 2 - 1)) / float(width) - 0.5f;
            V = (y + (rnd() * 2 - 1)) / float(height) - 0.5f;
        } else {  // pixel-centre ray of :1788-1789
            U = (x + 0.5f) / float(width) - 0.5f;
            V = (y + 0.5f) / float(height) - 0.5f;
        }
        const V3 pix = cam.eye + float(aspect * cam.film * U) * u + float(cam.film * V) * v - dist * w;
        Ray r;
        r.o = cam.eye;
        r.d = normalize(pix - cam.eye);
        return r;
    }

    V3 shade(const Hit& h, V3 view, int level) const;
    V3 shade_lambert(const Hit& h, V3 view, int level) const;
    V3 shade_metal(const Hit& h, V3 view, int level) const;
    V3 shade_glass(V3 view, const Hit& h, int level) const;
    V3 reflect_ray(V3 view, const Hit& h, int level) const;
};

static inline float saturate(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }  // :1902-1904

// src/cs488.h:1906-1910
static float fresnel(float e1, float e2, float ci, float co) {
    float rs = (e1 * ci - e2 * co) / (e1 * ci + e2 * co);
    float rt = (e1 * co - e2 * ci) / (e1 * co + e2 * ci);
    return (powf(rs, 2) + powf(rt, 2)) / 2;
}
// src/cs488.h:1912-1914
static inline V3 reflect_vec(V3 v, V3 axis) { return (-2 * dot(v, axis)) * axis + v; }

// src/cs488.h:1916-1924
V3 Scene::reflect_ray(V3 view, const Hit& h, int level) const {
    Ray r;
    r.d = reflect_vec(-view, h.N);
    r.o = h.P + kEps * h.N;
    Hit nh;
    bool got = intersect(nh, r);
    return got ? shade(nh, -r.d, level + 1) : ibl(r);
}

// src/cs488.h:1926-1963
V3 Scene::shade_glass(V3 view, const Hit& h, int level) const {
    V3 wi = -view;
    float e1, e2;
    float wn = dot(wi, h.N);
    if (h.backface) { e1 = h.mat->eta; e2 = kEtaAir; } else { e1 = kEtaAir; e2 = h.mat->eta; }
    float under = 1 - powf(e1 / e2, 2) * (1 - powf(wn, 2));
    if (under < 0) return reflect_ray(view, h, level);
    Ray rr;
    rr.d = (e1 / e2) * (wi - wn * h.N) - (sqrtf(under) * h.N);
    rr.o = h.P - kEps * h.N;
    Hit nh;
    bool got = intersect(nh, rr);  // traced before the Fresnel decision, as in the reference
    float ci = wn / (length(wi) * length(h.N));
    float co = dot(h.N, rr.d) / (length(h.N) * length(rr.d));
    float R = fresnel(e1, e2, ci, co);
    if (rnd() < R) return reflect_ray(view, h, level);
    return got ? shade(nh, -rr.d, level + 1) : ibl(rr);
}

// src/cs488.h:1994-2016
static V3 cosine_sample(V3 w) {
    float u1 = rnd();
    float u2 = rnd();
    float r = sqrtf(u1);
    float theta = 2.0f * kPi * u2;
    float x = r * cosf(theta);
    float y = r * sinf(theta);
    float z = sqrtf(1.0f - u1);
    V3 up = fabsf(w.z) < 0.999f ? mk(0.0f, 0.0f, 1.0f) : mk(1.0f, 0.0f, 0.0f);
    V3 tg = normalize(cross(up, w));
    V3 bt = cross(w, tg);
    return normalize(tg * x + bt * y + w * z);
}

// src/cs488.h:2018-2048
V3 Scene::shade_lambert(const Hit& h, V3 view, int level) const {
    V3 brdf = h.color();
    Ray nr;
    nr.o = h.P + kEps * h.N;
    nr.d = cosine_sample(h.N);
    const float ct = dot(nr.d, h.N);
    if (ct < kEps) return mk(0.0f);
    const float p = ct / kPi;
    Hit nh;
    bool got = intersect(nh, nr);
    V3 next = got ? shade(nh, -nr.d, level + 1) : ibl(nr) * p;
    V3 cur = h.mat->Ke + next * brdf * ct / p;
    if (h.mat->opacity < 1.0f - kEps) {
        nr.o = h.P - kEps * h.N;
        nr.d = -view;
        got = intersect(nh, nr);
        next = got ? shade(nh, view, level + 1) : ibl(nr);
        return h.mat->opacity * cur + (1 - h.mat->opacity) * next;
    }
    return cur;
}

// src/cs488.h:2050-2055
static float schlick(float eta, float c) {
    float a = eta - 1, b = eta + 1;
    float f0 = (a * a) / (b * b);
    return f0 + (1 - f0) * powf(1 - c, 5);
}
// src/cs488.h:2057-2061
static float geom_att(float nh, float vh, float nl, float nv) {
    float g1 = 2 * nh * nv / vh;
    float g2 = 2 * nh * nl / vh;
    return fminf(1.0f, fminf(g1, g2));
}
// src/cs488.h:2063-2069
static float ggx(float nh, float rough) {
    float a = rough * rough;
    float a2 = a * a;
    float den = nh * nh * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * den * den);
}
// src/cs488.h:2071-2088
static float cook_torrance(V3 view, V3 light, const Hit& h) {
    V3 n = h.N;
    V3 hv = normalize(view + light);
    float nl = saturate(dot(n, light));
    float nv = saturate(dot(n, view));
    float nh = saturate(dot(n, hv));
    float vh = saturate(dot(view, hv));
    float F = schlick(h.mat->eta, vh);
    float D = ggx(nh, h.mat->roughness);
    float G = geom_att(nh, vh, nl, nv);
    float spec = (F * D * G) / (4.0f * nv * nl + kEps);
    return saturate(spec);
}

struct M3 { V3 cx, cy, cz; };  // columns, as linalg's float3x3
static inline V3 mul(const M3& m, V3 v) { return m.cx * v.x + m.cy * v.y + m.cz * v.z; }  // linalg.h:536
static inline M3 transpose(const M3& m) {  // linalg.h:557
    M3 t;
    t.cx = mk(m.cx.x, m.cy.x, m.cz.x);
    t.cy = mk(m.cx.y, m.cy.y, m.cz.y);
    t.cz = mk(m.cx.z, m.cy.z, m.cz.z);
    return t;
}
// src/cs488.h:2090-2098
static M3 rotation_to(V3 z) {
    const V3 ref = fabsf(dot(z, mk(0, 1, 0))) > 0.99f ? mk(0, 0, 1) : mk(0, 1, 0);
    M3 m;
    m.cx = normalize(cross(ref, z));
    m.cy = cross(z, m.cx);
    m.cz = z;
    return m;
}
// src/cs488.h:2100-2130
static V3 sample_half(V3 n, V3 view, float rough) {
    const M3 wfg = rotation_to(n);
    const M3 gfw = transpose(wfg);
    const float ra = rnd();
    const float rb = rnd();
    const float alpha = rough * rough;
    const V3 vg = mul(gfw, view);
    const V3 vs = normalize(mk(alpha * vg.x, alpha * vg.y, vg.z));
    const float area_blue = 1;
    const float area_green = vs.z;
    const float pb = 1 / (area_blue + area_green);
    const float pg = 1 - pb;
    const float phi = (rb < pb) ? rb / pb * kPi : kPi + (rb - pb) / pg * kPi;
    const float r = sqrtf(ra);
    const float x = r * cosf(phi);
    const float y = r * sinf(phi) * (rb < pb ? area_blue : area_green);
    const float z = sqrtf(fmaxf(0.0f, 1.0f - x * x - y * y));
    const V3 ms = mul(rotation_to(vs), mk(x, y, z));
    const V3 mg = mk(alpha * ms.x, alpha * ms.y, fmaxf(0.0f, ms.z));
    return normalize(mul(wfg, mg));
}
// src/cs488.h:2132-2147
V3 Scene::shade_metal(const Hit& h, V3 view, int level) const {
    V3 hv = sample_half(h.N, view, h.mat->roughness);
    V3 light = normalize(reflect_vec(-view, hv));
    if (dot(light, h.N) < kEps) return mk(0.0f);
    Ray nr;
    nr.o = h.P + kEps * h.N;
    nr.d = light;
    Hit nh;
    bool got = intersect(nh, nr);
    V3 next = got ? shade(nh, -nr.d, level + 1) : ibl(nr);
    float spec = cook_torrance(view, light, h);
    return h.mat->Kd * next * dot(light, h.N) + next * h.mat->Ks * spec;
}
// src/cs488.h:2149-2172
V3 Scene::shade(const Hit& h, V3 view, int level) const {
    if (level > max_level) return mk(0.0f);
    if (tl_cnt && level < 20) tl_cnt->shade_calls[level]++;
    if (level > 4) {
        float prob = 0.6f;
        if (h.mat->type == LAMBERTIAN) {
            V3 f = h.color();
            prob = fmaxf(fmaxf(f.x, fmaxf(f.y, f.z)), 0.05f);
        }
        if (rnd() > prob) return mk(0.0f);
    }
    if (h.mat->type == LAMBERTIAN) return shade_lambert(h, view, level);
    if (h.mat->type == METAL) return shade_metal(h, view, level);
    if (h.mat->type == GLASS) return shade_glass(view, h, level);
    return mk(100.0f, 0.0f, 100.0f);
}

// src/cs488.h:1845-1860 -- one column strip of one progressive frame
static void render_strip(Scene* S, int xmin, int xmax, int spp) {
    for (int j = 0; j < S->height; ++j) {
        for (int i = xmin; i <= xmax; ++i) {
            V3& acc = S->accum[i + j * S->width];
            for (int k = 0; k < spp; ++k) {
                const Ray r = S->camera_ray(i, j, true);
                Hit h;
                V3 s = S->intersect(h, r) ? S->shade(h, -r.d, 0) : S->ibl(r);
                if (!(std::isnan(s.x) || std::isnan(s.y) || std::isnan(s.z))) {
                    acc = acc + s;
                } else if (tl_cnt) {
                    tl_cnt->nan_samples++;
                }
            }
            S->frame[i + j * S->width] = acc / (float)S->total_samples;
        }
    }
}

}  // namespace orc

// ---------------------------------------------------------------------------------------------
// C interface for the tests (ctypes)
// ---------------------------------------------------------------------------------------------
using namespace orc;

struct OrcMaterial {
    char name[96];
    int type;
    float eta;
    float Ka[3], Kd[3], Ks[3], Ke[3];
    float Ns, opacity, roughness;
    int textured;
    char tex_file[512];
};

static void add_counters(uint64_t* out, const Counters& c) {
    if (!out) return;
    out[0] += c.rays; out[1] += c.box_tests; out[2] += c.tri_tests; out[3] += c.nan_samples;
    for (int i = 0; i < 20; i++) out[4 + i] += c.shade_calls[i];
}

extern "C" {

void* orc_create() { return new Scene(); }
void orc_destroy(void* h) { delete (Scene*)h; }

// returns the triangle count, or -1 when the file is unusable (src/main.cpp:8-13 then falls
// back to orc_single_triangle)
int orc_load_obj(void* h, const char* path) {
    Scene* S = (Scene*)h;
    S->mesh = Mesh();
    S->built = false;
    if (!S->mesh.load(path)) return -1;
    return (int)S->mesh.tris.size();
}
void orc_single_triangle(void* h) {
    Scene* S = (Scene*)h;
    S->mesh = Mesh();
    S->mesh.single_triangle();
    S->built = false;
}
// explicit geometry (for synthetic scenes): arrays laid out as orc_get_tris returns them
void orc_set_tris(void* h, int n, const float* pos9, const float* nrm9, const float* uv6, const int* mat) {
    Scene* S = (Scene*)h;
    S->mesh.tris.assign(n, Tri());
    for (int i = 0; i < n; i++) {
        Tri& T = S->mesh.tris[i];
        for (int k = 0; k < 3; k++) {
            T.p[k] = mk(pos9[9 * i + 3 * k], pos9[9 * i + 3 * k + 1], pos9[9 * i + 3 * k + 2]);
            T.n[k] = mk(nrm9[9 * i + 3 * k], nrm9[9 * i + 3 * k + 1], nrm9[9 * i + 3 * k + 2]);
            T.uv[k][0] = uv6 ? uv6[6 * i + 2 * k] : 0.0f;
            T.uv[k][1] = uv6 ? uv6[6 * i + 2 * k + 1] : 0.0f;
        }
        T.mat = mat ? mat[i] : 0;
    }
    if (S->mesh.mats.empty()) S->mesh.mats.resize(1);
    S->built = false;
}
int orc_num_tris(void* h) { return (int)((Scene*)h)->mesh.tris.size(); }
int orc_num_materials(void* h) { return (int)((Scene*)h)->mesh.mats.size(); }
void orc_get_tris(void* h, float* pos9, float* nrm9, float* uv6, int* mat) {
    Scene* S = (Scene*)h;
    for (size_t i = 0; i < S->mesh.tris.size(); i++) {
        const Tri& T = S->mesh.tris[i];
        for (int k = 0; k < 3; k++) {
            for (int c = 0; c < 3; c++) {
                pos9[9 * i + 3 * k + c] = T.p[k][c];
                nrm9[9 * i + 3 * k + c] = T.n[k][c];
            }
            uv6[6 * i + 2 * k] = T.uv[k][0];
            uv6[6 * i + 2 * k + 1] = T.uv[k][1];
        }
        mat[i] = T.mat;
    }
}
void orc_get_material(void* h, int i, OrcMaterial* o) {
    const Material& m = ((Scene*)h)->mesh.mats[i];
    memset(o, 0, sizeof(*o));
    snprintf(o->name, sizeof(o->name), "%s", m.name.c_str());
    o->type = m.type; o->eta = m.eta;
    for (int c = 0; c < 3; c++) { o->Ka[c] = m.Ka[c]; o->Kd[c] = m.Kd[c]; o->Ks[c] = m.Ks[c]; o->Ke[c] = m.Ke[c]; }
    o->Ns = m.Ns; o->opacity = m.opacity; o->roughness = m.roughness;
    o->textured = m.textured ? 1 : 0;
    snprintf(o->tex_file, sizeof(o->tex_file), "%s", m.tex_file.c_str());
}
void orc_set_material(void* h, int i, const OrcMaterial* o) {
    Scene* S = (Scene*)h;
    if ((int)S->mesh.mats.size() <= i) S->mesh.mats.resize(i + 1);
    Material& m = S->mesh.mats[i];
    m.name = o->name; m.type = o->type; m.eta = o->eta;
    for (int c = 0; c < 3; c++) { m.Ka[c] = o->Ka[c]; m.Kd[c] = o->Kd[c]; m.Ks[c] = o->Ks[c]; m.Ke[c] = o->Ke[c]; }
    m.Ns = o->Ns; m.opacity = o->opacity; m.roughness = o->roughness;
    m.textured = o->textured != 0;
}
// decoded 8-bit RGB texture, first row of the file first (what stbi_load returns, :841)
int orc_set_texture(void* h, int mat, const uint8_t* rgb, int w, int ht) {
    Scene* S = (Scene*)h;
    if (mat < 0 || mat >= (int)S->mesh.mats.size()) return -1;
    Material& m = S->mesh.mats[mat];
    m.tex.assign(rgb, rgb + (size_t)w * ht * 3);
    m.tw = w; m.th = ht;
    return 0;
}
// decoded float RGB environment image, first row of the file first; stored flipped (:208-215)
void orc_set_env(void* h, const float* rgb, int w, int ht) {
    Scene* S = (Scene*)h;
    S->env.assign((size_t)w * ht, mk(0.0f));
    S->env_w = w; S->env_h = ht;
    size_t k = 0;
    for (int j = ht - 1; j >= 0; j--)
        for (int i = 0; i < w; i++) { S->env[i + (size_t)j * w] = mk(rgb[k], rgb[k + 1], rgb[k + 2]); k += 3; }
}
void orc_set_camera(void* h, const float* eye, const float* lookat, const float* up, float fov, float film) {
    Scene* S = (Scene*)h;
    S->cam.eye = mk(eye[0], eye[1], eye[2]);
    S->cam.lookat = mk(lookat[0], lookat[1], lookat[2]);
    S->cam.up = mk(up[0], up[1], up[2]);
    S->cam.fov = fov; S->cam.film = film;
}
void orc_set_max_level(void* h, int lvl) { ((Scene*)h)->max_level = lvl; }

// src/cs488.h:1733-1747
void orc_build(void* h, int accel) {
    Scene* S = (Scene*)h;
    S->accel = accel;
    S->mesh.precalc();
    if (accel == 1) S->kd.build(&S->mesh); else S->bvh.build(&S->mesh);
    S->built = true;
}

// tree dumps in the reference's own node numbering (BVH: pre-order array index, :1576-1627)
int orc_bvh_num_nodes(void* h) { return (int)((Scene*)h)->bvh.nodes.size(); }
int orc_bvh_num_refs(void* h) {
    int n = 0;
    for (const BvhNode& nd : ((Scene*)h)->bvh.nodes) n += (int)nd.list.size();
    return n;
}
void orc_bvh_dump(void* h, int* is_leaf, int* left, int* right, int* first, int* count, float* box6, int* refs) {
    Scene* S = (Scene*)h;
    int off = 0;
    for (size_t i = 0; i < S->bvh.nodes.size(); i++) {
        const BvhNode& nd = S->bvh.nodes[i];
        is_leaf[i] = nd.leaf; left[i] = nd.left; right[i] = nd.right;
        first[i] = off; count[i] = (int)nd.list.size();
        for (int k : nd.list) refs[off++] = k;
        for (int c = 0; c < 3; c++) { box6[6 * i + c] = nd.box.lo[c]; box6[6 * i + 3 + c] = nd.box.hi[c]; }
    }
}
int orc_kd_num_nodes(void* h) { return (int)((Scene*)h)->kd.nodes.size(); }
int orc_kd_num_refs(void* h) {
    int n = 0;
    for (const KdNode& nd : ((Scene*)h)->kd.nodes) if (nd.left < 0) n += (int)nd.list.size();
    return n;
}
// kd nodes in creation order (node, its two children as a pair, left subtree, right subtree);
// only leaves report their triangle lists
void orc_kd_dump(void* h, int* left, int* right, int* first, int* count, float* box6, int* refs) {
    Scene* S = (Scene*)h;
    int off = 0;
    for (size_t i = 0; i < S->kd.nodes.size(); i++) {
        const KdNode& nd = S->kd.nodes[i];
        left[i] = nd.left; right[i] = nd.right;
        first[i] = off; count[i] = 0;
        if (nd.left < 0) {
            count[i] = (int)nd.list.size();
            for (int k : nd.list) refs[off++] = k;
        }
        for (int c = 0; c < 3; c++) { box6[6 * i + c] = nd.box.lo[c]; box6[6 * i + 3 + c] = nd.box.hi[c]; }
    }
}

// Takes a kd-tree from outside instead of building one (arrays as orc_kd_dump writes them, same node
// numbering).  The reference's builder is quadratic per node (src/cs488.h:1249-1294) and so is the one
// above; for meshes too large for it the tests hand over the tree of the renderer's O(n log n) builder,
// which they hold node for node equal to this file's builder on every mesh small enough to compare.
// What is checked with an imported tree is the traversal (src/cs488.h:1344-1379), not the builder.
void orc_kd_import(void* h, int n_nodes, const int* left, const int* right, const int* first, const int* count,
                   const float* box6, const int* refs) {
    Scene* S = (Scene*)h;
    S->accel = 1;
    S->mesh.precalc();
    S->kd.mesh = &S->mesh;
    S->kd.nodes.assign((size_t)n_nodes, KdNode());
    for (int i = 0; i < n_nodes; i++) {
        KdNode& nd = S->kd.nodes[i];
        nd.left = left[i]; nd.right = right[i];
        for (int c = 0; c < 3; c++) { nd.box.lo[c] = box6[6 * i + c]; nd.box.hi[c] = box6[6 * i + 3 + c]; }
        if (nd.left < 0) nd.list.assign(refs + first[i], refs + first[i] + count[i]);
    }
    S->built = true;
}

// closest hit for a batch of rays (src/cs488.h:1750-1777); tri = -1 and t = FLT_MAX on a miss.
// counters (optional, 24 x u64) accumulate rays, box tests, triangle tests.
void orc_trace(void* h, int n, const float* rays6, int* tri, float* t, float* bary2, float* P3, float* N3,
               int* backface, uint64_t* counters) {
    Scene* S = (Scene*)h;
    Counters c;
    tl_cnt = &c;
    for (int i = 0; i < n; i++) {
        Ray r;
        r.o = mk(rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2]);
        r.d = mk(rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5]);
        Hit hh;
        bool got = S->intersect(hh, r);
        tri[i] = got ? hh.tri : -1;
        t[i] = got ? hh.t : FLT_MAX;
        if (bary2) { bary2[2 * i] = got ? hh.beta : 0; bary2[2 * i + 1] = got ? hh.gamma : 0; }
        if (P3) for (int k = 0; k < 3; k++) P3[3 * i + k] = got ? hh.P[k] : 0;
        if (N3) for (int k = 0; k < 3; k++) N3[3 * i + k] = got ? hh.N[k] : 0;
        if (backface) backface[i] = got ? (hh.backface ? 1 : 0) : 0;
    }
    tl_cnt = nullptr;
    add_counters(counters, c);
}

// all triangles whose individual test returns exactly t_ref (tie set of a closest hit);
// returns the number written (at most cap)
int orc_tie_set(void* h, const float* ray6, float t_ref, int* out, int cap) {
    Scene* S = (Scene*)h;
    Ray r;
    r.o = mk(ray6[0], ray6[1], ray6[2]);
    r.d = mk(ray6[3], ray6[4], ray6[5]);
    int n = 0;
    Hit hh;
    for (size_t i = 0; i < S->mesh.tris.size(); i++) {
        if (S->mesh.hit_tri(hh, r, (int)i, 0.0f, FLT_MAX) && hh.t == t_ref) {
            if (n < cap) out[n] = (int)i;
            n++;
        }
    }
    return n;
}

// camera rays (src/cs488.h:1825-1839, or the pixel-centre form of :1780-1794 when jitter == 0);
// with jitter the stream starts at *rng_state and the advanced state is written back
void orc_camera_rays(void* h, int width, int height, int jitter, uint64_t* rng_state, float* rays6) {
    Scene* S = (Scene*)h;
    S->width = width; S->height = height;
    Rng rng; rng.state = rng_state ? *rng_state : kDefaultSeed;
    tl_rng = &rng;
    size_t k = 0;
    for (int j = 0; j < height; j++)
        for (int i = 0; i < width; i++) {
            Ray r = S->camera_ray(i, j, jitter != 0);
            rays6[k++] = r.o.x; rays6[k++] = r.o.y; rays6[k++] = r.o.z;
            rays6[k++] = r.d.x; rays6[k++] = r.d.y; rays6[k++] = r.d.z;
        }
    tl_rng = nullptr;
    if (rng_state) *rng_state = rng.state;
}

// radiance of single rays: intersect ? shade(hit, -d, 0) : ibl  (src/cs488.h:1853); the RNG
// stream starts at *rng_state and continues from ray to ray
void orc_radiance(void* h, int n, const float* rays6, uint64_t* rng_state, float* rgb) {
    Scene* S = (Scene*)h;
    Rng rng; rng.state = *rng_state;
    tl_rng = &rng;
    for (int i = 0; i < n; i++) {
        Ray r;
        r.o = mk(rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2]);
        r.d = mk(rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5]);
        Hit hh;
        V3 s = S->intersect(hh, r) ? S->shade(hh, -r.d, 0) : S->ibl(r);
        rgb[3 * i] = s.x; rgb[3 * i + 1] = s.y; rgb[3 * i + 2] = s.z;
    }
    tl_rng = nullptr;
    *rng_state = rng.state;
}

void orc_ibl(void* h, int n, const float* dirs3, float* rgb) {
    Scene* S = (Scene*)h;
    for (int i = 0; i < n; i++) {
        Ray r; r.o = mk(0.0f); r.d = mk(dirs3[3 * i], dirs3[3 * i + 1], dirs3[3 * i + 2]);
        V3 s = S->ibl(r);
        rgb[3 * i] = s.x; rgb[3 * i + 1] = s.y; rgb[3 * i + 2] = s.z;
    }
}

// clear both buffers and the sample count (src/cs488.h:2288-2290)
void orc_reset(void* h, int width, int height) {
    Scene* S = (Scene*)h;
    S->width = width; S->height = height;
    S->accum.assign((size_t)width * height, mk(0.0f));
    S->frame.assign((size_t)width * height, mk(0.0f));
    S->total_samples = 0;
}

// `frames` progressive frames of `spp` samples each (src/cs488.h:1878-1897 inside the loop of
// :2278-2324, including the one extra RNG draw after each frame, :2317).  threads == 1 follows
// the reference's single global stream from *rng_state; threads > 1 gives strip s the stream
// seeded (*rng_state + 2*s*0x9E3779B97F4A7C15) | 1.  Returns total samples per pixel so far.
int orc_render(void* h, int spp, int frames, int threads, uint64_t* rng_state, uint64_t* counters) {
    Scene* S = (Scene*)h;
    if (S->accum.empty()) orc_reset(h, S->width, S->height);
    if (threads < 1) threads = 1;
    std::vector<Rng> rngs(threads);
    for (int s = 0; s < threads; s++) {
        rngs[s].state = threads == 1 ? *rng_state : ((*rng_state + 2ull * s * 0x9E3779B97F4A7C15ull) | 1ull);
    }
    std::vector<Counters> cnts(threads);
    for (int f = 0; f < frames; f++) {
        S->total_samples += spp;  // :1882
        if (threads == 1) {
            tl_rng = &rngs[0]; tl_cnt = &cnts[0];
            render_strip(S, 0, S->width - 1, spp);
            tl_rng = nullptr; tl_cnt = nullptr;
        } else {
            std::vector<std::thread> pool;
            for (int s = 0; s < threads; s++) {
                pool.emplace_back([S, s, threads, spp, &rngs, &cnts]() {
                    tl_rng = &rngs[s]; tl_cnt = &cnts[s];
                    render_strip(S, s * S->width / threads, (s + 1) * S->width / threads - 1, spp);  // :1888
                    tl_rng = nullptr; tl_cnt = nullptr;
                });
            }
            for (auto& th : pool) th.join();
        }
        for (int s = 0; s < threads; s++) rngs[s].next();  // :2317
    }
    *rng_state = rngs[0].state;
    for (int s = 0; s < threads; s++) add_counters(counters, cnts[s]);
    return S->total_samples;
}
void orc_get_image(void* h, int which, float* rgb) {  // 0 = accumulation, 1 = frame; row 0 = bottom
    Scene* S = (Scene*)h;
    const std::vector<V3>& src = which == 0 ? S->accum : S->frame;
    memcpy(rgb, src.data(), src.size() * sizeof(V3));
}

// first n outputs of the generator from *state (src/cs488.h:133-142)
void orc_pcg32(uint64_t* state, int n, uint32_t* out_u32, float* out_f32) {
    Rng r; r.state = *state;
    for (int i = 0; i < n; i++) {
        Rng c = r;
        uint32_t u = r.next_u32();
        if (out_u32) out_u32[i] = u;
        if (out_f32) out_f32[i] = c.next();
    }
    *state = r.state;
}

}  // extern "C"
