// LDS atomic-add throughput on one CU, by pattern (what bounds k_enc_front's pair-count pass?).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_atomic_rate.bin tools/micro/lds_atomic_rate.hip (git-ignored, travels with gpurun)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
// mode 0: 64 lanes, 64 different dwords (no conflict); 1: all lanes one dword; 2: 16 different dwords (4 lanes each);
// 3: as 0 with every second lane idle; 4: as 0 with 3 of 4 lanes idle; 5: eight dwords; 6: 64 dwords, stride 2 (bank pairs)
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cyc, int mode, int waves)
{
    __shared__ uint32_t c[8192];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (uint32_t j = tid; j < 8192; j += blockDim.x) c[j] = 0;
    __syncthreads();
    uint32_t idx;
    bool on = true;
    switch (mode) {
    case 0: idx = lane; break;
    case 1: idx = 0; break;
    case 2: idx = lane & 15; break;
    case 3: idx = lane; on = (lane & 1) == 0; break;
    case 4: idx = lane; on = (lane & 3) == 0; break;
    case 5: idx = lane & 7; break;
    default: idx = 2 * lane; break;
    }
    idx += 128 * w;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    if (on) {
#pragma unroll 16
        for (int i = 0; i < ITERS; i++) { atomicAdd(&c[idx], 1u); idx ^= 64u * (i & 1); }
    }
    __syncthreads();
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
    uint32_t s = 0;
    for (uint32_t j = tid; j < 8192; j += blockDim.x) s += c[j];
    out[blockIdx.x * blockDim.x + tid] = s;
    (void)waves;
}
int main()
{
    uint32_t *out; unsigned long long *cyc, h[2];
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    const char *names[] = {"64 dwords", "1 dword", "16 dwords", "64 dwords, half the lanes", "64 dwords, a quarter of the lanes", "8 dwords", "stride 2"};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int mode = 0; mode < 7; mode++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, cyc, mode, waves);
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, cyc, mode, waves);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            printf("waves %d  %-34s %8.2f cycles per wave instruction (shader clock), %7.1f ns total\n", waves, names[mode],
                   (double)h[0] / ITERS / waves, (double)h[1] * 10.0);
        }
    return 0;
}
