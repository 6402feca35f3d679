// device_math.h -- division without the divider, shared by the GPU stages (gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include "pt_math.h"

namespace pt {

// The correctly rounded reciprocal from the hardware approximation and one Newton step:
// fma(fma(-d, r0, 1), r0, r0) with r0 = v_rcp_f32(d) equals RN(1 / d) for every binary32 d with
// 2^-60 <= |d| <= 2^60 on gfx950 (tools/verify_fast_reciprocal.hip checks all 2.03e9 of them).
__device__ __forceinline__ float exact_reciprocal(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    return __fmaf_rn(__fmaf_rn(-d, r0, 1.0f), r0, r0);
}

// RN(a / d) from r = RN(1 / d):  q0 = RN(a r);  e = RN(fma(-q0, d, a)), which is exact;
// q = RN(fma(e, r, q0)).  Equal to the IEEE quotient for EVERY pair of binary32 significands
// (2^46 pairs, tools/verify_recip_division.hip), hence for all operands as long as no intermediate
// overflows or becomes subnormal.
__device__ __forceinline__ float div_by_recip(float a, float d, float r) {
    const float q0 = a * r;
    const float e = __fmaf_rn(-q0, d, a);
    return __fmaf_rn(e, r, q0);
}

// a / s and normalize(a) for the shading stages: the quotients are the IEEE ones whenever s and the
// results are normal numbers, which is every case that yields a finite direction or colour.
__device__ __forceinline__ vec3 divide3(vec3 a, float s) {
    const float r = exact_reciprocal(s);
    return v3(div_by_recip(a.x, s, r), div_by_recip(a.y, s, r), div_by_recip(a.z, s, r));
}
__device__ __forceinline__ float divide1(float a, float s) { return div_by_recip(a, s, exact_reciprocal(s)); }
__device__ __forceinline__ vec3 normalize_fast(vec3 a) { return divide3(a, sqrtf(dot(a, a))); }

// sin and cos of 2 pi x from the hardware (v_sin_f32 / v_cos_f32 take revolutions); used for
// sampling angles only, where an absolute error around 1e-6 is far below the Monte Carlo noise.
__device__ __forceinline__ float sin_turns(float x) { return __builtin_amdgcn_sinf(x); }
__device__ __forceinline__ float cos_turns(float x) { return __builtin_amdgcn_cosf(x); }

}  // namespace pt
