// r4x16_multi.hip - several GPUs of one node behind one call (include/rans4x16_hip.h part 3).
// Blocks are independent, so this is a static split: contiguous ranges of near-equal uncompressed bytes, one host
// thread and one context per device, every range through the ordinary host-buffer pipeline (r4x16_host.hip).
// No collective and no peer-to-peer traffic: the reference's batch loop (tests/rANS_static4x16pr_test.c:191-206)
// cut into `ndev` loops.
#include "r4x16_host.h"

extern "C" int rans4x16_hip_partition(int n, const unsigned int *weight, int parts, int *bounds)
{
    if (n < 0 || parts < 1 || !bounds) return -1;
    double total = 0;
    for (int i = 0; i < n; i++) total += weight ? (double)weight[i] : 1.0;
    bounds[0] = 0;
    double acc = 0;
    int i = 0;
    for (int r = 1; r < parts; r++) {
        const double target = total * (double)r / (double)parts;
        // block i belongs to the first range whose end lies beyond the block's midpoint
        while (i < n) {
            const double w = weight ? (double)weight[i] : 1.0;
            if (acc + w / 2 > target) break;
            acc += w;
            i++;
        }
        bounds[r] = i;
    }
    bounds[parts] = n;
    return 0;
}

struct rans4x16_hip_multi {
    std::vector<rans4x16_hip_ctx *> ctx;
    std::string err;
};

extern "C" rans4x16_hip_multi *rans4x16_hip_multi_create(int ndev, const int *devices)
{
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        static std::once_flag once;
        std::call_once(once, [] { fprintf(stderr, "rans4x16_hip: no HIP device available; this library has no CPU path\n"); });
        return nullptr;
    }
    if (ndev <= 0) { ndev = visible; devices = nullptr; }
    if (ndev > 64) return nullptr;
    rans4x16_hip_multi *m = new rans4x16_hip_multi();
    for (int d = 0; d < ndev; d++) {
        rans4x16_hip_ctx *c = rans4x16_hip_create(devices ? devices[d] : d);
        if (!c) { rans4x16_hip_multi_destroy(m); return nullptr; }
        m->ctx.push_back(c);
    }
    return m;
}

extern "C" void rans4x16_hip_multi_destroy(rans4x16_hip_multi *m)
{
    if (!m) return;
    for (auto *c : m->ctx) rans4x16_hip_destroy(c);
    delete m;
}

extern "C" int rans4x16_hip_multi_devices(const rans4x16_hip_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" const char *rans4x16_hip_multi_last_error(const rans4x16_hip_multi *m) { return m ? m->err.c_str() : "no context"; }

static int run_multi(rans4x16_hip_multi *m, int n, bool decode,
                     const unsigned char *const *in, const unsigned int *in_size,
                     unsigned char *const *out, unsigned int *out_size, const int *order, int *status)
{
    if (!m || m->ctx.empty()) return -1;
    if (n <= 0) return n == 0 ? 0 : -1;
    if (!in || !in_size || !out || !out_size) { m->err = "batch_multi: bad arguments"; return -1; }
    const int P = (int)m->ctx.size();
    std::vector<int> bounds((size_t)P + 1);
    // weight = uncompressed bytes: what the chain kernels' time follows (decode: the capacities, which are the
    // stored sizes for every caller of the reference API)
    if (rans4x16_hip_partition(n, decode ? out_size : in_size, P, bounds.data()) != 0) return -1;
    std::vector<int> rc((size_t)P, 0);
    auto work = [&](int p) {
        const int lo = bounds[(size_t)p], hi = bounds[(size_t)p + 1];
        if (hi <= lo) return;
        rc[(size_t)p] = r4x16_run_host_batch(m->ctx[(size_t)p], hi - lo, decode, in + lo, in_size + lo, out + lo, out_size + lo,
                                             order ? order + lo : nullptr, status ? status + lo : nullptr);
    };
    std::vector<std::thread> th;
    for (int p = 1; p < P; p++) th.emplace_back(work, p);
    work(0);
    for (auto &t : th) t.join();
    int failed = 0;
    for (int p = 0; p < P; p++) {
        if (rc[(size_t)p] < 0) {
            m->err = "device " + std::to_string(m->ctx[(size_t)p]->device) + ": " + m->ctx[(size_t)p]->err;
            return -1;
        }
        failed += rc[(size_t)p];
    }
    return failed;
}

extern "C" int rans4x16_hip_compress_batch_multi(rans4x16_hip_multi *m, int n,
                                                 const unsigned char *const *in, const unsigned int *in_size,
                                                 unsigned char *const *out, unsigned int *out_size,
                                                 const int *order, int *status)
{
    return run_multi(m, n, false, in, in_size, out, out_size, order, status);
}

extern "C" int rans4x16_hip_uncompress_batch_multi(rans4x16_hip_multi *m, int n,
                                                   const unsigned char *const *in, const unsigned int *in_size,
                                                   unsigned char *const *out, unsigned int *out_size, int *status)
{
    return run_multi(m, n, true, in, in_size, out, out_size, nullptr, status);
}
