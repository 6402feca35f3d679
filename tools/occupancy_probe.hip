// occupancy_probe.hip -- how many 256-lane workgroups does a CU of this GPU really hold at once?
// Every workgroup notes the 100 MHz clock when it starts and then spins for a fixed time; workgroups that start
// within the first tenth of the spin were resident together.  Swept over the dynamic LDS size and, through an
// inline-asm clobbers, the numbers of vector and scalar registers the kernel claims.
// build: hipcc --offload-arch=gfx950 -O2 tools/occupancy_probe.hip -o tools/occupancy_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int VGPRS, int SGPRS = 0>
__global__ __launch_bounds__(256) void spin(unsigned long long* start, int spin_ticks) {
    extern __shared__ int lds[];
    const unsigned long long t0 = wall_clock64();
    if (VGPRS > 64) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    else if (VGPRS > 56) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    else if (VGPRS > 32) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if (SGPRS > 96) asm volatile("s_mov_b32 s99, 0" ::: "s99");
    else if (SGPRS > 88) asm volatile("s_mov_b32 s91, 0" ::: "s91");
    else if (SGPRS > 80) asm volatile("s_mov_b32 s83, 0" ::: "s83");
    else if (SGPRS > 72) asm volatile("s_mov_b32 s75, 0" ::: "s75");
    else if (SGPRS > 64) asm volatile("s_mov_b32 s67, 0" ::: "s67");
    lds[threadIdx.x] = (int)t0;
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) start[blockIdx.x] = t0 + (lds[(threadIdx.x + 1) & 255] & 0);
}

template <int VGPRS, int SGPRS = 0>
void run(int cus, int lds_bytes) {
    const int per_cu = 12, n = cus * per_cu, ticks = 20000;  // 200 us
    unsigned long long* d;
    (void)hipMalloc(&d, n * sizeof(unsigned long long));
    int api = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, spin<VGPRS, SGPRS>, 256, lds_bytes);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((spin<VGPRS, SGPRS>), dim3(n), dim3(256), lds_bytes, 0, d, ticks);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const unsigned long long first = *std::min_element(h.begin(), h.end());
    int together = 0;
    for (auto t : h) together += (t - first) < (unsigned long long)ticks / 10;
    printf("vgprs<=%3d sgprs<=%3d lds %6d B: occupancy query %2d per CU, started together %5d = %.2f per CU\n", VGPRS, SGPRS, lds_bytes, api, together,
           together / (double)cus);
    (void)hipFree(d);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, LDS per CU %zu B\n", p.name, p.multiProcessorCount, (size_t)p.maxSharedMemoryPerMultiProcessor);
    const int lds[] = {0, 8192, 16384, 18432, 20112, 20480, 22528, 26388, 32768};
    for (int b : lds) run<32>(p.multiProcessorCount, b);
    for (int b : {0, 20112}) run<64>(p.multiProcessorCount, b);
    for (int b : {0, 20112}) run<56>(p.multiProcessorCount, b);
    for (int b : {0}) run<96>(p.multiProcessorCount, b);
    run<56, 68>(p.multiProcessorCount, 20112);
    run<56, 76>(p.multiProcessorCount, 20112);
    run<56, 84>(p.multiProcessorCount, 20112);
    run<56, 92>(p.multiProcessorCount, 20112);
    run<56, 100>(p.multiProcessorCount, 20112);
    return 0;
}
