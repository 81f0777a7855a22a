// Do kernels overlap?  (round 4: how the chain kernel's class launches can run side by side)
//   (a) two spinning kernels on ONE stream, plain launches                       -> expected: serial
//   (b) the same with hipExtAnyOrderLaunch on the second                         -> overlap if the barrier bit is dropped
//   (c) k spinning kernels on k streams of one priority / of mixed priorities    -> how many hardware queues there are
// Each kernel is one workgroup per CU spinning for ~5 ms on s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/launch_overlap.bin tools/micro/launch_overlap.hip && ./tools/micro/launch_overlap.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void spin(uint64_t ticks, uint32_t *sink)
{
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t v = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) v++;
    if (v == 0xffffffffu) *sink = v;
}
int main()
{
    uint32_t *d;
    hipMalloc(&d, 4);
    const uint64_t ticks = 500000;                   // s_memtime runs at 100 MHz: 5 ms
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto timed = [&](const char *what, auto body) {
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            body();
            hipEventRecord(b, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%-60s %7.2f ms\n", what, ms);
        }
    };
    hipStream_t s0;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreateWithFlags(&e0, hipEventDisableTiming); hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    auto on_stream = [&](auto body) {                // bracket work on s0 between the null-stream events
        hipEventRecord(e0, 0); hipStreamWaitEvent(s0, e0, 0);
        body();
        hipEventRecord(e1, s0); hipStreamWaitEvent(0, e1, 0);
    };
    timed("one kernel", [&] { on_stream([&] { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, ticks, d); }); });
    timed("two kernels, one stream, plain", [&] { on_stream([&] {
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, ticks, d);
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, ticks, d); }); });
    timed("two kernels, one stream, second hipExtAnyOrderLaunch", [&] { on_stream([&] {
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, ticks, d);
        hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d); }); });
    timed("eight kernels, one stream, all but the first any-order", [&] { on_stream([&] {
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, ticks, d);
        for (int i = 0; i < 7; i++) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d); }); });
    timed("any-order pair, then a plain one (must wait for both)", [&] { on_stream([&] {
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, ticks, d);
        hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, ticks, d); }); });
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("stream priorities: least %d, greatest %d\n", lo, hi);
    for (int mixed = 0; mixed < 2; mixed++)
        for (int k : {2, 3, 4, 6, 8, 12}) {
            std::vector<hipStream_t> ss(k);
            for (int i = 0; i < k; i++) {
                const int levels = lo - hi + 1;
                const int prio = mixed ? hi + i % levels : (lo + hi) / 2;
                hipStreamCreateWithPriority(&ss[i], hipStreamNonBlocking, prio);
            }
            std::vector<hipEvent_t> ev(k);
            for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
            char what[128];
            snprintf(what, sizeof what, "%d kernels on %d streams, %s", k, k, mixed ? "priorities cycling" : "one priority");
            timed(what, [&] {
                hipEventRecord(e0, 0);
                for (int i = 0; i < k; i++) {
                    hipStreamWaitEvent(ss[i], e0, 0);
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, ss[i], ticks, d);
                    hipEventRecord(ev[i], ss[i]);
                    hipStreamWaitEvent(0, ev[i], 0);
                }
            });
            for (auto s : ss) hipStreamDestroy(s);
            for (auto e : ev) hipEventDestroy(e);
        }
    return 0;
}
