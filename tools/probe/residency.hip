// Probe: are the first `nact` single-wave workgroups of a larger grid co-resident when the rest of
// the grid exits at once?  (The chain launches cover 2 items per block; the second half is idle.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(64) void spin(unsigned long long ticks, int nact, unsigned *sink, unsigned *where) {
    extern __shared__ unsigned lds[];
    if ((int)blockIdx.x >= nact) return;
    lds[threadIdx.x] = threadIdx.x;
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    if (threadIdx.x == 0) { where[3 * blockIdx.x] = xcc; where[3 * blockIdx.x + 1] = hw; where[3 * blockIdx.x + 2] = (unsigned)t0; }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == 12345) sink[0] = 1;
}
int main(int argc, char **argv) {
    unsigned *sink, *where; (void)hipMalloc(&sink, 4); (void)hipMalloc(&where, 3 * 4096 * 4);
    (void)hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int lds = argc > 1 ? atoi(argv[1]) : 81920;
    for (int i = 2; i + 1 < argc; i += 2) {
        const int grid = atoi(argv[i]), nact = atoi(argv[i + 1]);
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(grid), dim3(64), lds, 0, 200000ull /*2 ms*/, nact, sink, where);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        static unsigned h[3 * 4096];
        (void)hipMemcpy(h, where, sizeof h, hipMemcpyDeviceToHost);
        // per XCC: how many active workgroups started within the first 0.5 ms
        unsigned tmin = ~0u; for (int w = 0; w < nact && w < 4096; w++) if (h[3 * w + 2] < tmin) tmin = h[3 * w + 2];
        int early[16] = {0}, late[16] = {0};
        for (int w = 0; w < nact && w < 4096; w++) { const unsigned x = h[3 * w] & 15; if (h[3 * w + 2] - tmin < 50000) early[x]++; else late[x]++; }
        printf("lds=%d grid=%d active=%d : %.2f ms  early/late per XCC:", lds, grid, nact, best);
        for (int x = 0; x < 8; x++) printf(" %d/%d", early[x], late[x]);
        printf("\n");
    }
    return 0;
}
