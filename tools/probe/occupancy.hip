// Probe: how many single-wave workgroups with X bytes of dynamic LDS are resident per CU on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(64) void spin(unsigned long long ticks, unsigned *sink) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == 12345) sink[0] = 1;
}
int main(int argc, char **argv) {
    unsigned *sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    int dflt[] = {7232, 7296, 7680, 8192, 14464, 14592, 14848, 15360, 16384, 17920, 18176, 18432, 20480, 28928, 30720, 36160};
    int sizes[64], nsz = 0;
    for (int i = 1; i < argc && nsz < 64; i++) sizes[nsz++] = atoi(argv[i]);      // sizes to probe on the command line
    if (!nsz) for (int v : dflt) sizes[nsz++] = v;
    for (int si = 0; si < nsz; si++) {
        const int s = sizes[si];
        int best = 0;
        for (int k = 1; k <= 40; k++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(256 * k), dim3(64), s, 0, 200000ull /*2 ms*/, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < 3.2f) best = k; else break;
        }
        printf("lds=%6d B  resident WGs/CU >= %d\n", s, best);
    }
    return 0;
}
