// issue_rate_probe.hip -- how many scalar-ALU and vector-ALU instructions a gfx950 SIMD issues per
// cycle when eight waves are resident, alone and mixed.  Tells whether a kernel with a given
// SALU : VALU ratio is limited by scalar issue.
// build: hipcc --offload-arch=gfx950 -O2 tools/issue_rate_probe.hip -o gpurun_out/issue_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define S4(a, b, c, d) "s_add_u32 %" #a ", %" #a ", 1\n s_add_u32 %" #b ", %" #b ", 1\n s_add_u32 %" #c ", %" #c ", 1\n s_add_u32 %" #d ", %" #d ", 1\n"
#define V4(a, b, c, d) "v_add_f32 %" #a ", %" #a ", %" #a "\n v_add_f32 %" #b ", %" #b ", %" #b "\n v_add_f32 %" #c ", %" #c ", %" #c "\n v_add_f32 %" #d ", %" #d ", %" #d "\n"
#define M4(v0, v1, s0, s1) "v_add_f32 %" #v0 ", %" #v0 ", %" #v0 "\n s_add_u32 %" #s0 ", %" #s0 ", 1\n v_add_f32 %" #v1 ", %" #v1 ", %" #v1 "\n s_add_u32 %" #s1 ", %" #s1 ", 1\n"
#define M3(v0, v1, s0) "v_add_f32 %" #v0 ", %" #v0 ", %" #v0 "\n s_add_u32 %" #s0 ", %" #s0 ", 1\n v_add_f32 %" #v1 ", %" #v1 ", %" #v1 "\n"

// MODE 0: 32 SALU / iteration; 1: 32 VALU; 2: 32 VALU + 32 SALU interleaved; 3: 32 VALU + 16 SALU
template <int MODE>
__global__ void __launch_bounds__(256) probe(int iters, float* out, unsigned* out_s) {
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile(S4(0, 1, 2, 3) S4(0, 1, 2, 3) S4(0, 1, 2, 3) S4(0, 1, 2, 3) S4(0, 1, 2, 3) S4(0, 1, 2, 3) S4(0, 1, 2, 3) S4(0, 1, 2, 3)
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 1) {
            asm volatile(V4(0, 1, 2, 3) V4(0, 1, 2, 3) V4(0, 1, 2, 3) V4(0, 1, 2, 3) V4(0, 1, 2, 3) V4(0, 1, 2, 3) V4(0, 1, 2, 3) V4(0, 1, 2, 3)
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        } else if (MODE == 2) {
            asm volatile(M4(0, 1, 4, 5) M4(2, 3, 6, 7) M4(0, 1, 4, 5) M4(2, 3, 6, 7) M4(0, 1, 4, 5) M4(2, 3, 6, 7) M4(0, 1, 4, 5) M4(2, 3, 6, 7)
                         M4(0, 1, 4, 5) M4(2, 3, 6, 7) M4(0, 1, 4, 5) M4(2, 3, 6, 7) M4(0, 1, 4, 5) M4(2, 3, 6, 7) M4(0, 1, 4, 5) M4(2, 3, 6, 7)
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else {
            asm volatile(M3(0, 1, 4) M3(2, 3, 5) M3(0, 1, 6) M3(2, 3, 7) M3(0, 1, 4) M3(2, 3, 5) M3(0, 1, 6) M3(2, 3, 7)
                         M3(0, 1, 4) M3(2, 3, 5) M3(0, 1, 6) M3(2, 3, 7) M3(0, 1, 4) M3(2, 3, 5) M3(0, 1, 6) M3(2, 3, 7)
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
    }
    if (v0 + v1 + v2 + v3 == 12345.0f) out[threadIdx.x] = v0;       // never true for these inputs; keeps the chains alive
    if (s0 + s1 + s2 + s3 == 7u && threadIdx.x == 0) out_s[0] = s0;
}

template <int MODE>
static double run(int blocks, int iters, float* d, unsigned* ds) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(iters / 10, d, ds);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<blocks, 256>>>(iters, d, ds);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3;
}

int main() {
    float* d; unsigned* ds;
    hipMalloc(&d, 4096); hipMalloc(&ds, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clock = p.clockRate * 1e3;
    const int iters = 20000;
    for (int per_cu = 1; per_cu <= 8; per_cu *= 2) {
        const int blocks = cus * per_cu;  // per_cu workgroups of 4 waves: per_cu waves on every SIMD
        const double t0 = run<0>(blocks, iters, d, ds), t1 = run<1>(blocks, iters, d, ds), t2 = run<2>(blocks, iters, d, ds), t3 = run<3>(blocks, iters, d, ds);
        // cycles a SIMD spends per iteration of all its waves, per instruction of the named kind
        auto cyc = [&](double t, int n) { return t * clock / ((double)iters * per_cu * n); };
        printf("waves/SIMD %d: SALU only %.2f cyc/inst | VALU only %.2f cyc/inst | 32V+32S %.2f cyc per V+S pair | 32V+16S %.2f cyc per V (+S/2)\n",
               per_cu, cyc(t0, 32), cyc(t1, 32), cyc(t2, 32), cyc(t3, 32));
    }
    printf("clock %.0f MHz, %d CUs\n", clock * 1e-6, cus);
    return 0;
}
