// Microbenchmark: what does it cost to fetch 64-byte tree nodes at random, (A) one node per lane as four 16-byte loads,
// against (B) one node per group of four lanes, each lane loading one 16-byte quarter?  Both variants chase pointers:
// the next node index comes out of the fetched data, as in a tree walk.  Table sizes: 13 MB (the sponza stand-in's BVH
// nodes; beyond one XCD's L2) and 1 MB.  usage: node_fetch_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ __launch_bounds__(256) void per_lane(const float4* __restrict__ nodes, unsigned int n_nodes, int steps, unsigned int* out) {
    unsigned int cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_nodes;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        const float4* p = nodes + 4 * (size_t)cur;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        cur = (__float_as_uint(a.w) ^ __float_as_uint(d.x)) % n_nodes;  // data-dependent next node
    }
    out[blockIdx.x * 256 + threadIdx.x] = cur + (unsigned int)acc;
}

__global__ __launch_bounds__(256) void per_quad(const float4* __restrict__ nodes, unsigned int n_nodes, int steps, unsigned int* out) {
    const unsigned int piece = threadIdx.x & 3u;
    unsigned int cur = ((blockIdx.x * 256u + threadIdx.x) >> 2) * 2654435761u % n_nodes;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        const float4 a = nodes[4 * (size_t)cur + piece];
        acc += a.x + a.y;
        // the quad's lanes exchange what the next index is made of: quarter 0's w and quarter 3's x
        const unsigned int w0 = (unsigned int)__builtin_amdgcn_mov_dpp((int)__float_as_uint(a.w), 0x00, 0xf, 0xf, true);
        const unsigned int x3 = (unsigned int)__builtin_amdgcn_mov_dpp((int)__float_as_uint(a.x), 0xff, 0xf, 0xf, true);
        cur = (w0 ^ x3) % n_nodes;
    }
    out[blockIdx.x * 256 + threadIdx.x] = cur + (unsigned int)acc;
}

int main() {
    const int blocks = 256 * 5, steps = 64;
    unsigned int* d_out;
    hipMalloc(&d_out, blocks * 256 * sizeof(unsigned int));
    for (unsigned int n_nodes : {16384u, 200000u, 1500000u}) {
        std::vector<float4> h(4 * (size_t)n_nodes);
        srand(7);
        for (auto& v : h) { unsigned int r[4]; for (int k = 0; k < 4; k++) r[k] = (unsigned int)rand() * 2654435761u; for (int k = 0; k < 4; k++) r[k] &= 0x3fffffffu; float f[4]; memcpy(f, r, 16); v = make_float4(f[0], f[1], f[2], f[3]); }
        float4* d_nodes;
        hipMalloc(&d_nodes, h.size() * sizeof(float4));
        hipMemcpy(d_nodes, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int variant = 0; variant < 2; variant++) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(per_lane, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, steps, d_out);
                else hipLaunchKernelGGL(per_quad, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, steps, d_out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double fetches = (double)blocks * 256 * steps / (variant == 0 ? 1 : 4);
            printf("%8u nodes (%6.1f MB)  %s: %7.3f ms, %7.1f G node fetches/s, %6.2f TB/s of node bytes, %7.1f G wave-steps/s\n", n_nodes, n_nodes * 64e-6,
                   variant == 0 ? "one node per lane (4 loads)" : "one node per quad (1 load) ", best, fetches / best * 1e-6, fetches * 64 / best * 1e-9,
                   (double)blocks * 4 * steps / best * 1e-6);
        }
        hipFree(d_nodes);
    }
    return 0;
}
