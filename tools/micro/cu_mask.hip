// Do CU masks partition the chip?  (round 4: would make the scheduler's shares physical)
// Streams created with hipExtStreamCreateWithCUMask over sets of mask bits; a kernel of 1,024 workgroups on each records the
// XCC and the (SE, SH, CU) of the CU every workgroup ran on.  Prints, per stream, how many distinct CUs were used per XCC -
// which says (a) whether the mask is honoured and (b) how mask bits map to XCCs on this part.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/cu_mask.bin tools/micro/cu_mask.hip && ./tools/micro/cu_mask.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <set>
__global__ void where(uint32_t *out, uint64_t ticks)
{
    const uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11));
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) << 16 | (hw & 0xff00u) | ((hw >> 13) & 7u) << 4 | ((hw >> 12) & 1u);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t v = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) v++;
    if (v == 0xffffffffu) out[0] = v;
}
int main()
{
    const int NW = 1024;
    uint32_t *d; hipMalloc(&d, 4 * NW * 8);
    auto run = [&](const char *what, std::vector<uint32_t> mask) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: create failed: %s\n", what, hipGetErrorString(e)); return; }
        hipLaunchKernelGGL(where, dim3(NW), dim3(64), 0, s, d, (uint64_t)20000);
        hipStreamSynchronize(s);
        std::vector<uint32_t> h(NW);
        hipMemcpy(h.data(), d, 4 * NW, hipMemcpyDeviceToHost);
        std::set<uint32_t> cus[16];
        for (uint32_t v : h) cus[(v >> 16) & 15].insert(v & 0xffff);
        printf("%-44s CUs used per XCC:", what);
        for (int x = 0; x < 8; x++) printf(" %2zu", cus[x].size());
        printf("\n");
        hipStreamDestroy(s);
    };
    run("all 256 bits", std::vector<uint32_t>(8, 0xffffffffu));
    run("bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    run("bits 128..255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("every eighth bit (0, 8, 16, ..)", std::vector<uint32_t>(8, 0x01010101u));
    run("even bits", std::vector<uint32_t>(8, 0x55555555u));
    return 0;
}
