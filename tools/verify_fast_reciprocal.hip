// verify_fast_reciprocal.hip -- does one Newton step on the hardware reciprocal give the correctly
// rounded reciprocal?      r0 = v_rcp_f32(d);   r1 = fma(fma(-d, r0, 1), r0, r0)
// is compared with RN(1 / d) (IEEE division) for EVERY binary32 d with 2^-60 <= |d| <= 2^60, both
// signs (121 exponents x 2^23 significands x 2).  Prints the number of mismatches and the first few.
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/verify_fast_reciprocal.hip -o tools/verify_fast_rcp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void check(unsigned exp_first, unsigned long long* bad, unsigned* first_bad, int raw) {
    const unsigned sig = blockIdx.x * blockDim.x + threadIdx.x;          // 2^23 significands
    const unsigned e = exp_first + blockIdx.y;                           // biased exponent
    for (unsigned sign = 0; sign < 2; sign++) {
        const float d = __uint_as_float((sign << 31) | (e << 23) | sig);
        const float r0 = __builtin_amdgcn_rcpf(d);
        const float r1 = raw ? r0 : __fmaf_rn(__fmaf_rn(-d, r0, 1.0f), r0, r0);
        const float ref = 1.0f / d;
        if (__float_as_uint(r1) != __float_as_uint(ref)) {
            atomicAdd(bad, 1ull);
            const unsigned slot = atomicAdd(&first_bad[0], 1u);
            if (slot < 16) first_bad[1 + slot] = __float_as_uint(d);
        }
    }
}

int main(int argc, char** argv) {
    const int raw = argc > 1 ? atoi(argv[1]) : 0;  // 1: test the unrefined v_rcp_f32 (expected to fail: checks the checker)
    unsigned long long* d_bad;
    unsigned* d_first;
    hipMalloc((void**)&d_bad, sizeof(unsigned long long));
    hipMalloc((void**)&d_first, 17 * sizeof(unsigned));
    hipMemset(d_bad, 0, sizeof(unsigned long long));
    hipMemset(d_first, 0, 17 * sizeof(unsigned));
    const unsigned e_lo = 127 - 60, e_hi = 127 + 60;
    hipLaunchKernelGGL(check, dim3((1u << 23) / 256, e_hi - e_lo + 1), dim3(256), 0, 0, e_lo, d_bad, d_first, raw);
    hipDeviceSynchronize();
    unsigned long long bad = 0;
    unsigned first[17];
    hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    hipMemcpy(first, d_first, sizeof(first), hipMemcpyDeviceToHost);
    for (unsigned k = 0; k < first[0] && k < 16; k++) {
        float f;
        memcpy(&f, &first[1 + k], 4);
        printf("mismatch: d = 0x%08x (%g)\n", first[1 + k], f);
    }
    printf("%s: %llu values checked, %llu mismatches\n", raw ? "raw v_rcp_f32" : "v_rcp_f32 + one Newton step",
           2ull * (e_hi - e_lo + 1) * (1ull << 23), bad);
    return 0;
}
