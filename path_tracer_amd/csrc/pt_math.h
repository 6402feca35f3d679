// pt_math.h -- 3-vector arithmetic shared by host and device code.
//
// Operation order follows the vector library the reference uses (src/linalg.h): a dot product is
// a left-to-right sum started from zero (src/linalg.h:367,401,501), normalize divides every
// component by the square root of the dot product (src/linalg.h:503-504), cross is the textbook
// form (src/linalg.h:500).  Code that must agree bit for bit with x86 scalar arithmetic is built
// with floating-point contraction off, so none of these expressions turns into an FMA.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace pt {

struct vec3 {
    float x, y, z;
};

PT_HD vec3 v3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD vec3 v3(float s) { return v3(s, s, s); }
PT_HD vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD vec3 operator-(vec3 a) { return v3(-a.x, -a.y, -a.z); }
PT_HD vec3 operator*(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD vec3 operator*(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD vec3 operator*(float s, vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PT_HD vec3 operator/(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PT_HD float dot(vec3 a, vec3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
PT_HD vec3 cross(vec3 a, vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_HD float length(vec3 a) { return sqrtf(dot(a, a)); }
PT_HD vec3 normalize(vec3 a) { return a / length(a); }
PT_HD float comp(vec3 a, int axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }

// constants of src/cs488.h:53-61
#define PT_PI 3.14159265358979f
#define PT_EPSILON 5e-5f
#define PT_ETA_AIR 1.0f

}  // namespace pt
