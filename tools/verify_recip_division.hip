// verify_recip_division.hip -- exhaustive check of the three-operation division used by the trace
// kernel's slab test:
//     r  = RN(1 / d)                 (IEEE division, once per ray and axis)
//     q0 = RN(a * r)
//     e  = RN(fma(-q0, d, a))        (exact: the residual is representable)
//     q  = RN(fma(e, r, q0))
// against q_ref = RN(a / d) (IEEE division), for EVERY pair of binary32 significands
// (a, d in [1, 2), 2^23 x 2^23 pairs).  Scaling a or d by a power of two scales every intermediate
// by a power of two, so the significand pairs decide all cases in which nothing overflows or
// becomes subnormal.  Prints the number of mismatches and the first few offending pairs.
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/verify_recip_division.hip -o tools/verify_recip
// run:   tools/verify_recip [first_d_block] [n_d_blocks] [1 = test the uncorrected product instead]   (a d-block = 2^13 significands of d)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float div3(float a, float d, float r) {
    const float q0 = a * r;
    const float e = __fmaf_rn(-q0, d, a);
    return __fmaf_rn(e, r, q0);
}

// one thread = one d significand and one run of a significands (blockIdx.y picks the run)
__global__ void check(unsigned d_first, unsigned a_count, unsigned long long* bad, unsigned* first_bad, int uncorrected) {
    const unsigned di = d_first + blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned a_first = blockIdx.y * a_count;
    const float d = __uint_as_float(0x3f800000u | di);
    const float r = 1.0f / d;
    unsigned long long local = 0;
    for (unsigned k = 0; k < a_count; k++) {
        const unsigned ai = a_first + k;
        const float a = __uint_as_float(0x3f800000u | ai);
        const float q = uncorrected ? a * r : div3(a, d, r);  // the uncorrected product must FAIL (checks the checker)
        const float ref = a / d;
        if (__float_as_uint(q) != __float_as_uint(ref)) {
            local++;
            const unsigned slot = atomicAdd(&first_bad[0], 1u);
            if (slot < 16) { first_bad[1 + 2 * slot] = ai; first_bad[2 + 2 * slot] = di; }
        }
    }
    if (local) atomicAdd(bad, local);
}

int main(int argc, char** argv) {
    const unsigned first_block = argc > 1 ? (unsigned)atoi(argv[1]) : 0;
    const unsigned n_blocks = argc > 2 ? (unsigned)atoi(argv[2]) : 1024;  // 1024 x 2^13 = all 2^23 d significands
    const int uncorrected = argc > 3 ? atoi(argv[3]) : 0;
    unsigned long long* d_bad;
    unsigned* d_first;
    hipMalloc((void**)&d_bad, sizeof(unsigned long long));
    hipMalloc((void**)&d_first, 33 * sizeof(unsigned));
    hipMemset(d_bad, 0, sizeof(unsigned long long));
    hipMemset(d_first, 0, 33 * sizeof(unsigned));
    unsigned long long total_pairs = 0;
    for (unsigned b = first_block; b < first_block + n_blocks && b < 1024; b++) {
        // 2^13 values of d per launch, all 2^23 values of a in 128 runs of 2^16
        hipLaunchKernelGGL(check, dim3((1u << 13) / 256, 128), dim3(256), 0, 0, b << 13, 1u << 16, d_bad, d_first, uncorrected);
        total_pairs += (1ull << 13) * (1ull << 23);
        if ((b - first_block) % 16 == 15 || b == first_block + n_blocks - 1) {
            unsigned long long bad = 0;
            hipDeviceSynchronize();
            hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
            printf("d blocks [%u, %u]: %llu pairs checked, %llu mismatches\n", first_block, b, total_pairs, bad);
            fflush(stdout);
        }
    }
    unsigned first[33];
    unsigned long long bad = 0;
    hipDeviceSynchronize();
    hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    hipMemcpy(first, d_first, sizeof(first), hipMemcpyDeviceToHost);
    for (unsigned k = 0; k < first[0] && k < 16; k++)
        printf("mismatch: a significand 0x%06x  d significand 0x%06x\n", first[1 + 2 * k], first[2 + 2 * k]);
    printf("TOTAL %llu pairs, %llu mismatches\n", total_pairs, bad);
    return bad ? 1 : 0;
}
