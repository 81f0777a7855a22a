// LDS round trip seen by ONE wave (one workgroup on an idle chip): dependent chains of ds_read_b32 / ds_read_u8 /
// ds_read2_b32, each with k vector instructions between the load and the next address.  Cycles by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_latency.bin tools/micro/lds_latency.hip && ./tools/micro/lds_latency.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 20000
template <int KIND, int K>
__global__ void chase(uint64_t *out, uint32_t seed)
{
    __shared__ uint32_t t[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) t[i] = (uint32_t)((i * 1103515245u + seed) >> 8) & 4095u;
    __syncthreads();
    uint32_t v = threadIdx.x & 3;                  // four lanes live, like one stream's quad; the others follow lane 0..3 too
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; i++) {
        uint32_t r;
        if (KIND == 0) r = t[v];
        else if (KIND == 1) r = ((volatile uint8_t *)t)[v * 4] | (v & 0xf00u);
        else { r = t[v] ^ t[(v + 1) & 4095]; }
#pragma unroll
        for (int k = 0; k < K; k++) r = (r * 5u + 1u) & 4095u;     // K dependent vector instructions (mad + and: 2 each)
        v = r & 4095u;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
int main()
{
    uint64_t *d, h[2];
    hipMalloc(&d, 16);
    auto run = [&](const char *name, void (*k)(uint64_t *, uint32_t)) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 12345u); hipDeviceSynchronize(); }
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-34s %7.1f cycles per iteration\n", name, (double)h[0] / N);
    };
    run("ds_read_b32, 0 valu", chase<0, 0>);
    run("ds_read_b32, 2 valu", chase<0, 1>);
    run("ds_read_b32, 8 valu", chase<0, 4>);
    run("ds_read_b32, 16 valu", chase<0, 8>);
    run("ds_read_u8, 0 valu", chase<1, 0>);
    run("ds_read_u8, 8 valu", chase<1, 4>);
    run("2 x ds_read_b32 (one wait), 0 valu", chase<2, 0>);
    return 0;
}
