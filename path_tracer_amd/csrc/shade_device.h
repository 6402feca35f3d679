// shade_device.h -- device code of the shading stage (gfx950), shared by the per-level shade
// kernel and the fused kernel that finishes the deep bounce levels.
//
// The reference shades by recursion (shade, src/cs488.h:2149-2172); here every path carries a
// throughput and each bounce is one pass over a queue.  With beta the product of the factors the
// recursion would multiply onto the value returned from below:
//   level > maxLevel             -> the path ends (src/cs488.h:2150)
//   level > 4                    -> Russian roulette, survival leaves beta unchanged (:2153-2160)
//   Lambertian (:2018-2048)      -> cos < eps ends the path with nothing added; otherwise
//                                   radiance += beta*o*Ke, bounce child beta*o*(c*cos/p) whose miss
//                                   adds env*p (miss_scale = p); if opacity o < 1-eps a second
//                                   child continues straight through with beta*(1-o)
//   metal (:2132-2147)           -> beta * (Kd*dot(L,N) + Ks*spec); ends if dot(L,N) < eps
//   glass (:1926-1963)           -> beta unchanged; reflect on total internal reflection or when
//                                   a uniform draw is below the Fresnel reflectance, else refract
//                                   (the reference traces the refracted ray before deciding;
//                                   here only the chosen ray is traced)
//   miss                         -> radiance += beta * env(dir) * miss_scale (:1717-1727)
// A path that ends with a non-finite throughput adds NaN to its sample, as the recursion's
// 0 * NaN would; the accumulate stage drops samples with a NaN component but still counts them
// (src/cs488.h:1841-1860).
//
// Random numbers are counter based: value = squares32(counter, key) with key derived from the
// seed and the GLOBAL sample index and counter = pixel << 32 | path id << 8 | dimension, so an
// image does not depend on how samples are split into batches or over GPUs.  Each draw is
// turned into a float as the reference does, float(double(u32) / 2^32) (src/cs488.h:140-142).
//
// Surviving rays are appended to the workgroup's own region of the next queue: ballot of the
// lanes that emit, population count for the wave's block of entries (one LDS atomic per wave),
// prefix count for each lane's entry.  No global atomic is involved.
#pragma once

#include <float.h>

#include "device_math.h"
#include "device_types.h"
#include "pt_math.h"
#include "trace_device.h"  // what the fused kernel stages of the tree, for the LDS budget below

namespace pt {

namespace shade_dev {


__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t squares32(uint64_t ctr, uint64_t key) {
    uint64_t x = ctr * key, y = x, z = y + key;
    x = x * x + y; x = (x >> 32) | (x << 32);
    x = x * x + z; x = (x >> 32) | (x << 32);
    x = x * x + y; x = (x >> 32) | (x << 32);
    return (uint32_t)((x * x + z) >> 32);
}

struct Sampler {
    uint64_t key;
    uint64_t base;  // pixel << 32 | path id << 8
    uint32_t dim;
    __device__ __forceinline__ float next() {
        const uint32_t u = squares32(base | (dim & 0xffu), key);
        dim++;
        return (float)((double)u / 4294967296.0);
    }
};

__device__ __forceinline__ uint64_t sample_key(uint64_t seed, int global_sample) {
    return mix64(seed ^ mix64((uint64_t)(uint32_t)global_sample)) | 1ull;
}

__device__ __forceinline__ Sampler make_sampler(uint64_t key, int pixel, uint32_t bits) {
    Sampler s;
    s.key = key;
    s.base = ((uint64_t)(uint32_t)pixel << 32) | (uint64_t)(bits & 0xffffff00u);
    s.dim = bits & 0xffu;
    return s;
}

// Position, within dealing round l, of the run that workgroup blk receives (see raygen_kernel).  The product
// stays below the number of runs in a batch (at most 2^22), so 32-bit arithmetic is exact.
__device__ __forceinline__ int dealt_position(int blk, int l, const BatchGpu& b) {
    return (int)(((unsigned int)blk + (unsigned int)l * (unsigned int)b.deal_shift) % (unsigned int)b.blocks);
}

// slot -> (pixel, index of the sample within the batch's chunk)
__device__ __forceinline__ void slot_to_sample(const BatchGpu& b, int slot, int& pixel, int& k) {
    const int pl = slot / b.chunk;
    pixel = b.pixel0 + pl;
    k = slot - pl * b.chunk;
}

// Reserves queue entries for the lanes of one wave: every lane passes how many records it emits
// first (0/1) and second (0/1); the wave takes one block of the workgroup's LDS counter and each
// lane gets its two indices (relative to the workgroup's region).  Must be reached by all 64 lanes.
__device__ __forceinline__ void wave_reserve(bool e0, bool e1, int* lds_counter, int& idx0, int& idx1) {
    const unsigned long long m0 = __ballot(e0), m1 = __ballot(e1);
    const int lane = __lane_id();
    const int c0 = __popcll(m0), c1 = __popcll(m1);
    int base = 0;
    if (lane == 0 && (c0 + c1) > 0) base = atomicAdd(lds_counter, c0 + c1);
    base = __shfl(base, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    idx0 = base + __popcll(m0 & below);
    idx1 = base + c0 + __popcll(m1 & below);
}

// Adds a path's contribution to its camera sample.  A scene without translucent surfaces has one path per sample,
// whose contributions arrive level by level: float atomics, always in the same order.  Where paths fork
// (src/cs488.h:2038-2044) several paths share a sample and the order of their additions depends on timing, so such a
// scene (sc.fixed_radiance) sums in 64-bit fixed point instead -- units of 2^-32, exact and therefore independent of
// the order; the image is then the same bits run after run, as the reference's is with one thread.  A record is two
// float4 there: three sums and a word of flags for what fixed point cannot hold (bits 1-3 / 4-6: +inf / -inf or beyond
// +-2^30 in r, g, b; bits 7-9: NaN in r, g, b), which accumulate_kernel turns back into the float it stands for,
// component by component as the float path and the reference's recursion would have produced it (the sample is
// dropped for a NaN in any component either way, src/cs488.h:1841-1858).
enum { RAD_POS_INF = 2, RAD_NEG_INF = 16, RAD_NAN = 128 };
__device__ __forceinline__ void add_radiance(const SceneGpu& sc, float4* sample_rad, int slot, vec3 v) {
    if (!sc.fixed_radiance) {
        float* p = reinterpret_cast<float*>(sample_rad + slot);
        atomicAdd(p + 0, v.x);
        atomicAdd(p + 1, v.y);
        atomicAdd(p + 2, v.z);
        return;
    }
    unsigned long long* p = reinterpret_cast<unsigned long long*>(sample_rad + 2 * (size_t)slot);
    const float c[3] = {v.x, v.y, v.z};
    unsigned int flags = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (fabsf(c[k]) < 1073741824.0f) {  // false for NaN
            const long long q = __float2ll_rn(c[k] * 4294967296.0f);
            if (q != 0) atomicAdd(p + k, (unsigned long long)q);
        } else {
            flags |= (isnan(c[k]) ? (unsigned int)RAD_NAN : (c[k] > 0.0f ? (unsigned int)RAD_POS_INF : (unsigned int)RAD_NEG_INF)) << k;
        }
    }
    if (flags) atomicOr(reinterpret_cast<unsigned int*>(p + 3), flags);
}

// The float a fixed-point record stands for (see add_radiance).
__device__ __forceinline__ float4 radiance_of_fixed(const float4& lo, const float4& hi) {
    const long long r = (long long)(((unsigned long long)__float_as_uint(lo.y) << 32) | __float_as_uint(lo.x));
    const long long g = (long long)(((unsigned long long)__float_as_uint(lo.w) << 32) | __float_as_uint(lo.z));
    const long long b = (long long)(((unsigned long long)__float_as_uint(hi.y) << 32) | __float_as_uint(hi.x));
    const unsigned int flags = __float_as_uint(hi.z);
    float out[3] = {(float)((double)r * (1.0 / 4294967296.0)), (float)((double)g * (1.0 / 4294967296.0)), (float)((double)b * (1.0 / 4294967296.0))};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const bool pos = (flags & ((unsigned int)RAD_POS_INF << k)) != 0, neg = (flags & ((unsigned int)RAD_NEG_INF << k)) != 0;
        if (pos || neg) out[k] = pos && neg ? NAN : (pos ? INFINITY : -INFINITY);
        if (flags & ((unsigned int)RAD_NAN << k)) out[k] = NAN;
    }
    return make_float4(out[0], out[1], out[2], 0.0f);
}

__device__ __forceinline__ bool finite3(vec3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }

// Scene::ibl (src/cs488.h:1717-1727); the texel index is clamped where the reference would
// index outside the image (direction along +-z).
__device__ __forceinline__ vec3 env_lookup(const SceneGpu& sc, vec3 d) {
    if (sc.env_h == 0) return v3(0.0f);
    const float r = (1.0f / PT_PI) * acosf(d.z) / sqrtf((float)((double)d.x * (double)d.x + (double)d.y * (double)d.y));
    const float u = d.x * r, v = d.y * r;
    const float fi = (u + 1.0f) * 0.5f * sc.env_w;
    const float fj = (v + 1.0f) * 0.5f * sc.env_h;
    int i = (fi == fi) ? (int)fi : 0;
    int j = (fj == fj) ? (int)fj : 0;
    i = i < 0 ? 0 : (i > sc.env_w - 1 ? sc.env_w - 1 : i);
    j = j < 0 ? 0 : (j > sc.env_h - 1 ? sc.env_h - 1 : j);
    const float* p = sc.env + 3 * ((size_t)i + (size_t)j * sc.env_w);
    return v3(p[0], p[1], p[2]);
}

__device__ __forceinline__ float saturate(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }

// -2 * dot(v, axis) * axis + v  (src/cs488.h:1912-1914)
__device__ __forceinline__ vec3 reflect_about(vec3 v, vec3 axis) { return (-2 * dot(v, axis)) * axis + v; }

struct Frame3 { vec3 cx, cy, cz; };
__device__ __forceinline__ vec3 apply(const Frame3& m, vec3 v) { return m.cx * v.x + m.cy * v.y + m.cz * v.z; }
__device__ __forceinline__ vec3 apply_transposed(const Frame3& m, vec3 v) {
    // rows of the transpose are the columns of m; product formed column by column like the
    // reference's mul(transpose(m), v) (src/linalg.h:536,557)
    return v3(m.cx.x, m.cy.x, m.cz.x) * v.x + v3(m.cx.y, m.cy.y, m.cz.y) * v.y + v3(m.cx.z, m.cy.z, m.cz.z) * v.z;
}
// MakeRotationMatrix (src/cs488.h:2090-2098)
__device__ __forceinline__ Frame3 frame_about(vec3 z) {
    const vec3 ref = fabsf(dot(z, v3(0.0f, 1.0f, 0.0f))) > 0.99f ? v3(0.0f, 0.0f, 1.0f) : v3(0.0f, 1.0f, 0.0f);
    Frame3 m;
    m.cx = normalize_fast(cross(ref, z));
    m.cy = cross(z, m.cx);
    m.cz = z;
    return m;
}

// cosineWeightedHemisphereSample (src/cs488.h:1994-2016)
__device__ __forceinline__ vec3 cosine_direction(vec3 w, float u1, float u2) {
    const float r = sqrtf(u1);
    const float x = r * cos_turns(u2);  // theta = 2 pi u2
    const float y = r * sin_turns(u2);
    const float z = sqrtf(1.0f - u1);
    const vec3 up = fabsf(w.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    const vec3 tg = normalize_fast(cross(up, w));
    const vec3 bt = cross(w, tg);
    return normalize_fast(tg * x + bt * y + w * z);
}

// sampleHalfVector (src/cs488.h:2100-2130)
__device__ __forceinline__ vec3 half_vector(vec3 n, vec3 view, float rough, float ra, float rb) {
    const Frame3 wfg = frame_about(n);
    const float alpha = rough * rough;
    const vec3 vg = apply_transposed(wfg, view);
    const vec3 vs = normalize_fast(v3(alpha * vg.x, alpha * vg.y, vg.z));
    const float area_green = vs.z;
    const float pb = exact_reciprocal(1.0f + area_green);
    const float pg = 1 - pb;
    const float phi = (rb < pb) ? divide1(rb, pb) * PT_PI : PT_PI + divide1(rb - pb, pg) * PT_PI;
    const float r = sqrtf(ra);
    const float turns = phi * (0.5f / PT_PI);
    const float x = r * cos_turns(turns);
    const float y = r * sin_turns(turns) * (rb < pb ? 1.0f : area_green);
    const float z = sqrtf(fmaxf(0.0f, 1.0f - x * x - y * y));
    const vec3 ms = apply(frame_about(vs), v3(x, y, z));
    const vec3 mg = v3(alpha * ms.x, alpha * ms.y, fmaxf(0.0f, ms.z));
    return normalize_fast(apply(wfg, mg));
}

// cookTorranceReflectance with schlickFresnel, normalDistributionGGX, geometricAttenuation
// (src/cs488.h:2050-2088)
__device__ __forceinline__ float cook_torrance(vec3 view, vec3 light, vec3 n, float rough, float eta) {
    const vec3 hv = normalize_fast(view + light);
    const float nl = saturate(dot(n, light));
    const float nv = saturate(dot(n, view));
    const float nh = saturate(dot(n, hv));
    const float vh = saturate(dot(view, hv));
    const float es = eta - 1, ep = eta + 1;
    const float f0 = divide1(es * es, ep * ep);
    const float omv = 1 - vh, omv2 = omv * omv;
    const float F = f0 + (1 - f0) * (omv2 * omv2 * omv);
    const float a = rough * rough;
    const float a2 = a * a;
    const float den = nh * nh * (a2 - 1.0f) + 1.0f;
    const float D = divide1(a2, PT_PI * den * den);
    const float rvh = exact_reciprocal(vh);
    const float g1 = div_by_recip(2 * nh * nv, vh, rvh);
    const float g2 = div_by_recip(2 * nh * nl, vh, rvh);
    const float G = fminf(1.0f, fminf(g1, g2));
    const float spec = divide1(F * D * G, 4.0f * nv * nl + PT_EPSILON);
    return saturate(spec);
}

// fresnel (src/cs488.h:1906-1910)
__device__ __forceinline__ float fresnel_unpolarised(float e1, float e2, float ci, float co) {
    const float rs = divide1(e1 * ci - e2 * co, e1 * ci + e2 * co);
    const float rt = divide1(e1 * co - e2 * ci, e1 * co + e2 * ci);
    return (rs * rs + rt * rt) / 2;
}

// Material::fetchTexture (src/cs488.h:402-414)
__device__ __forceinline__ vec3 texel(const SceneGpu& sc, int tw, int th, int off, float u, float v) {
    int x = int(u * tw) % tw;
    int y = int(v * th) % th;
    if (x < 0) x += tw;
    if (y < 0) y += th;
    const uint8_t* p = sc.texels + (size_t)off + 3 * ((size_t)x + (size_t)y * tw);
    return v3((float)p[0], (float)p[1], (float)p[2]) / 255.0f;
}

// ---- one bounce of shading for the n paths at [region, region + n) of the queue ----
// Called by all 256 threads of the workgroup.  Survivors go to the same region of `out`; the
// number produced (clamped to the region's capacity) is returned to every thread, with a barrier
// passed, and an overflow raises the batch's flag.
// What the host saw in the scene decides which instantiation runs, so that branches that cannot be
// taken (and the registers they keep live) are compiled out:
//   SHADE_PLAIN       only opaque, untextured Lambertian materials and no environment image
//   SHADE_LAMBERTIAN  only Lambertian materials (textures, translucency and an environment allowed)
//   SHADE_GLASS       Lambertian and glass materials;  SHADE_METAL  Lambertian and metal materials
//   SHADE_ALL         anything
enum { SHADE_ALL = 0, SHADE_LAMBERTIAN = 1, SHADE_PLAIN = 2, SHADE_GLASS = 3, SHADE_METAL = 4 };
// Shading geometry of a small scene staged in LDS.  A hit's three positions, three normals and its material record
// are otherwise eleven gathered 16-byte loads per path, and it is the rate of such lane-addresses, not bytes, that
// the shading of a scene this size waits for (cornellbox: shade_kernel 0.125 -> 0.104 ms per launch).  GEO_NONE:
// nothing staged; GEO_ALL: material records, positions and normals of a scene of at most GEO_LDS_MAX_TRIS
// triangles and GEO_LDS_MAX_MATS materials.  (Staging only the materials of larger scenes, and the batch's sampler
// keys, was measured too: nothing gained, the teapot lost 0.6 %.)  Layout (float4 units): MATERIAL_STRIDE records
// per material, 3 positions per triangle, 3 normals per triangle.
enum { GEO_NONE = 0, GEO_ALL = 2 };
enum { GEO_LDS_MAX_TRIS = 256, GEO_LDS_MAX_MATS = 32 };
struct ShadeLds {
    const float4* mats;
    const float4* pos;
    const float4* nrm;
};
// The fused kernel stages the geometry behind its part of the tree and beside its stacks, of which the stacks and
// counters take less than GEO_LDS_STATIC; a gfx950 workgroup can have 160 KiB of LDS.  (With the staged part of the
// tree chosen per scene the sum stays below 50 KiB; PT_OPT_LDS_NODES can ask for a stage that the trace kernel can just
// launch with, and the fused kernel then leaves the geometry in memory.)
enum { GEO_LDS_STATIC = 17 * 1024, GEO_LDS_WORKGROUP = 160 * 1024 };
__host__ __device__ inline int geometry_kind(const SceneGpu& sc) {
    if (sc.n_mats > GEO_LDS_MAX_MATS || sc.n_tris > GEO_LDS_MAX_TRIS) return GEO_NONE;
    const size_t geo = ((size_t)MATERIAL_STRIDE * sc.n_mats + (size_t)6 * sc.n_tris) * sizeof(float4);
    return GEO_LDS_STATIC + trace_dev::dynamic_lds_bytes(sc) + 16 + geo <= GEO_LDS_WORKGROUP ? GEO_ALL : GEO_NONE;
}
__host__ __device__ inline size_t geometry_lds_bytes(const SceneGpu& sc) {
    return geometry_kind(sc) == GEO_ALL ? ((size_t)MATERIAL_STRIDE * sc.n_mats + (size_t)6 * sc.n_tris) * sizeof(float4) : 0;
}
// Every thread of the workgroup calls it with the same `kind` (geometry_kind); a barrier must follow before the
// first shade_region.
__device__ __forceinline__ ShadeLds stage_geometry(const SceneGpu& sc, float4* geo, int kind) {
    ShadeLds l;
    float4* mats = geo;
    float4* pos = mats + MATERIAL_STRIDE * sc.n_mats;
    float4* nrm = pos + 3 * sc.n_tris;
    l.mats = mats; l.pos = pos; l.nrm = nrm;
    if (kind == GEO_ALL) {
        const int n_m = MATERIAL_STRIDE * sc.n_mats, n_p = 3 * sc.n_tris;
        for (int k = threadIdx.x; k < n_m + 2 * n_p; k += blockDim.x)
            geo[k] = k < n_m ? sc.mats[k] : (k < n_m + n_p ? sc.tri_pos[k - n_m] : sc.tri_nrm[k - n_m - n_p]);
    }
    return l;
}

// CAMERA: the queue holds camera rays, which carry no throughput record (see below).
// GEO: what is read from LDS (stage_geometry) instead of memory.
template <int KIND, bool CAMERA, int GEO = GEO_NONE>
__device__ __forceinline__ int shade_region(const SceneGpu& sc, const BatchGpu& b, int level, int max_level, const QueueSet& in,
                                            const float4* __restrict__ hit, const QueueSet& out,
                                            float4* __restrict__ sample_rad, int n, int* s_out_ptr, const ShadeLds& lds = ShadeLds()) {
    int& s_out = *s_out_ptr;
    const int blk = blockIdx.x;
    if (threadIdx.x == 0) s_out = 0;
    __syncthreads();
    const size_t region = (size_t)blk * b.cap_block;
    for (int base = 0; base < n; base += blockDim.x) {
        const int j = base + threadIdx.x;
        const size_t i = region + j;
        bool emit0 = false, emit1 = false;
        vec3 o0 = v3(0.0f), d0 = v3(0.0f), b0 = v3(0.0f);
        vec3 o1 = v3(0.0f), d1 = v3(0.0f), b1 = v3(0.0f);
        float scale0 = 1.0f;
        uint32_t bits0 = 0, bits1 = 0;
        int slot = 0;

        if (j < n) {
            const float4 rd = in.ray_d[i], h = hit[i];
            float4 bt;
            if (!CAMERA) {
                bt = in.beta[i];
            } else {
                // camera rays carry no throughput record: it is (1, 1, 1) and the sample slot follows from the
                // position in the workgroup's region, the inverse of raygen_kernel's dealing
                const int l = j >> 6;
                const int slot0 = (l * b.blocks + dealt_position(blk, l, b)) * 64 + (j & 63);
                bt = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(slot0));
            }
            const vec3 d = v3(rd.x, rd.y, rd.z);
            const vec3 beta = v3(bt.x, bt.y, bt.z);
            slot = __float_as_int(bt.w);
            const uint32_t bits = __float_as_uint(rd.w);
            const int tri = __float_as_int(h.x);
            bool ended = false;  // path stops here without any contribution of its own

            if (tri < 0) {
                // miss: environment radiance (zero without a probe; the product keeps a NaN alive)
                if (KIND == SHADE_PLAIN) {  // no environment: radiance zero, times a throughput that may be NaN
                    if (!finite3(beta)) add_radiance(sc, sample_rad, slot, beta * 0.0f);
                } else {
                    const vec3 e = env_lookup(sc, d) * h.z;  // miss record: h.z = the ray's miss scale
                    if (sc.env_h != 0 || !finite3(beta)) add_radiance(sc, sample_rad, slot, beta * e);
                }
            } else if (level > max_level) {
                ended = true;
            } else {
                int pixel, k;
                slot_to_sample(b, slot, pixel, k);
                Sampler rng = make_sampler(b.keys[k], pixel, bits);

                // hit record: interpolated position, shading normal, texcoords (src/cs488.h:671-675)
                const float hb = h.z, hg = h.w, ha = 1 - hb - hg;
                const float4* pos = GEO == GEO_ALL ? lds.pos + 3 * tri : sc.tri_pos + 3 * tri;
                const float4* nrm = GEO == GEO_ALL ? lds.nrm + 3 * tri : sc.tri_nrm + 3 * tri;
                const float4 P0 = pos[0], P1 = pos[1], P2 = pos[2];
                const float4 N0 = nrm[0], N1 = nrm[1], N2 = nrm[2];
                vec3 N = normalize_fast(ha * v3(N0.x, N0.y, N0.z) + hb * v3(N1.x, N1.y, N1.z) + hg * v3(N2.x, N2.y, N2.z));
                const vec3 P = ha * v3(P0.x, P0.y, P0.z) + hb * v3(P1.x, P1.y, P1.z) + hg * v3(P2.x, P2.y, P2.z);
                const bool backface = dot(d, N) > 0;  // src/cs488.h:1770-1773
                if (backface) N = -N;
                const vec3 view = -d;

                const int mid = __float_as_int(P0.w);
                const float4* mat = GEO != GEO_NONE ? lds.mats + MATERIAL_STRIDE * mid : sc.mats + MATERIAL_STRIDE * mid;
                const float4 m0 = mat[0], m1 = mat[1], m2 = mat[2], m3 = mat[3];
                const int type = __float_as_int(m0.w);
                const vec3 Kd = v3(m0.x, m0.y, m0.z), Ks = v3(m1.x, m1.y, m1.z), Ke = v3(m2.x, m2.y, m2.z);
                const float eta = m1.w, opacity = m2.w, rough = m3.x;
                const int tw = __float_as_int(m3.y), th = __float_as_int(m3.z), toff = __float_as_int(m3.w);

                // HitInfo::surfaceColor (src/cs488.h:473-475), Lambertian only (Material::BRDF :417-428)
                vec3 colour = v3(0.0f);
                if (type == 0) {
                    const float4 m4 = mat[4];
                    colour = v3(m4.x, m4.y, m4.z);  // Kd / pi, divided on the host
                    if (KIND != SHADE_PLAIN && tw > 0) {
                        const float2 T0 = sc.tri_uv[3 * tri], T1 = sc.tri_uv[3 * tri + 1], T2 = sc.tri_uv[3 * tri + 2];
                        const float tu = ha * T0.x + hb * T1.x + hg * T2.x;
                        const float tv = ha * T0.y + hb * T1.y + hg * T2.y;
                        colour = colour * texel(sc, tw, th, toff, tu, tv);
                    }
                }

                bool alive = true;
                if (level > 4) {  // Russian roulette, src/cs488.h:2153-2160
                    float prob = 0.6f;
                    if (type == 0) prob = fmaxf(fmaxf(colour.x, fmaxf(colour.y, colour.z)), 0.05f);
                    if (rng.next() > prob) alive = false;
                }
                if (!alive) {
                    ended = true;
                } else if (type == 0) {
                    const float u1 = rng.next();
                    const float u2 = rng.next();
                    const vec3 wo = cosine_direction(N, u1, u2);
                    const float ct = dot(wo, N);
                    if (ct < PT_EPSILON) {
                        ended = true;
                    } else {
                        const float p = div_by_recip(ct, PT_PI, 1.0f / PT_PI);
                        const bool translucent = KIND != SHADE_PLAIN && opacity < 1.0f - PT_EPSILON;
                        const vec3 weight = translucent ? beta * opacity : beta;
                        if (Ke.x != 0.0f || Ke.y != 0.0f || Ke.z != 0.0f) add_radiance(sc, sample_rad, slot, weight * Ke);
                        emit0 = true;
                        o0 = P + PT_EPSILON * N;
                        d0 = wo;
                        b0 = weight * divide3(colour * ct, p);
                        scale0 = p;
                        if (translucent) {
                            emit1 = true;
                            o1 = P - PT_EPSILON * N;
                            d1 = d;
                            b1 = beta * (1 - opacity);
                        }
                    }
                } else if ((KIND == SHADE_ALL || KIND == SHADE_METAL) && type == 1) {
                    const float ra = rng.next();
                    const float rb = rng.next();
                    const vec3 hv = half_vector(N, view, rough, ra, rb);
                    const vec3 light = normalize_fast(reflect_about(-view, hv));
                    const float ln = dot(light, N);
                    if (ln < PT_EPSILON) {
                        ended = true;
                    } else {
                        const float spec = cook_torrance(view, light, N, rough, eta);
                        emit0 = true;
                        o0 = P + PT_EPSILON * N;
                        d0 = light;
                        b0 = beta * (Kd * ln + Ks * spec);
                    }
                } else if ((KIND == SHADE_ALL || KIND == SHADE_GLASS) && type == 2) {
                    const vec3 wi = d;
                    const float wn = dot(wi, N);
                    const float e1 = backface ? eta : PT_ETA_AIR, e2 = backface ? PT_ETA_AIR : eta;
                    const float ratio = divide1(e1, e2);
                    const float under = 1 - (ratio * ratio) * (1 - wn * wn);
                    bool reflect = under < 0;
                    vec3 rdir = v3(0.0f);
                    if (!reflect) {
                        rdir = ratio * (wi - wn * N) - (sqrtf(under) * N);
                        const float ci = divide1(wn, length(wi) * length(N));
                        const float co = divide1(dot(N, rdir), length(N) * length(rdir));
                        const float R = fresnel_unpolarised(e1, e2, ci, co);
                        if (rng.next() < R) reflect = true;
                    }
                    emit0 = true;
                    b0 = beta;
                    if (reflect) {
                        d0 = reflect_about(wi, N);
                        o0 = P + PT_EPSILON * N;
                    } else {
                        d0 = rdir;
                        o0 = P - PT_EPSILON * N;
                    }
                } else if (KIND == SHADE_ALL) {
                    // unknown material type: the reference's error colour (src/cs488.h:2170)
                    add_radiance(sc, sample_rad, slot, beta * v3(100.0f, 0.0f, 100.0f));
                }

                // random-number bookkeeping of the children: a fork doubles the path id
                const uint32_t pid = bits >> 8;
                if (emit1) {
                    bits0 = (((pid << 1) & 0xffffffu) << 8);
                    bits1 = ((((pid << 1) | 1u) & 0xffffffu) << 8);
                } else {
                    bits0 = (pid << 8) | (rng.dim & 0xffu);
                }
            }
            if (ended && !finite3(beta)) add_radiance(sc, sample_rad, slot, v3(NAN));
        }

        int k0, k1;
        wave_reserve(emit0, emit1, &s_out, k0, k1);
        if (emit0 && k0 < b.cap_block) {
            out.ray_o[region + k0] = make_float4(o0.x, o0.y, o0.z, scale0);
            out.ray_d[region + k0] = make_float4(d0.x, d0.y, d0.z, __uint_as_float(bits0));
            out.beta[region + k0] = make_float4(b0.x, b0.y, b0.z, __int_as_float(slot));
        }
        if (emit1 && k1 < b.cap_block) {
            out.ray_o[region + k1] = make_float4(o1.x, o1.y, o1.z, 1.0f);
            out.ray_d[region + k1] = make_float4(d1.x, d1.y, d1.z, __uint_as_float(bits1));
            out.beta[region + k1] = make_float4(b1.x, b1.y, b1.z, __int_as_float(slot));
        }
    }
    __syncthreads();
    const int produced = s_out;
    if (threadIdx.x == 0 && produced > b.cap_block) *b.overflow = 1;  // paths were dropped: the batch will be repeated
    return produced < b.cap_block ? produced : b.cap_block;
}


}  // namespace shade_dev

}  // namespace pt
