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

// ---- host feed: each device's worker on the CPUs of the device's NUMA node (include/rans4x16_hip.h part 3) ----------
#include <pthread.h>
#include <sched.h>

extern "C" int rans4x16_hip_cpulist_parse(const char *list, unsigned char *mask, int mask_bytes)
{
    if (!list || !mask || mask_bytes <= 0) return -1;
    memset(mask, 0, (size_t)mask_bytes);
    int count = 0;
    const char *p = list;
    auto skip = [&] { while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++; };
    auto number = [&](long *v) -> bool {
        if (*p < '0' || *p > '9') return false;
        long x = 0;
        while (*p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); if (x > 1000000) return false; p++; }
        *v = x;
        return true;
    };
    skip();
    if (!*p) return 0;                                    // an empty list: a node without CPUs
    for (;;) {
        long a, b;
        if (!number(&a)) return -1;
        b = a;
        if (*p == '-') { p++; if (!number(&b) || b < a) return -1; }
        if (b >= 8L * mask_bytes) return -1;
        for (long c = a; c <= b; c++)
            if (!(mask[c >> 3] & (1u << (c & 7)))) { mask[c >> 3] |= (unsigned char)(1u << (c & 7)); count++; }
        skip();
        if (*p == ',') { p++; skip(); continue; }
        if (!*p) break;
        return -1;
    }
    return count;
}

static bool read_text(const std::string &path, char *buf, size_t cap)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}
static int device_numa_node(int device)
{
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof id, device) != hipSuccess) return -1;
    for (char *q = id; *q; q++) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');     // sysfs spells it in lower case
    char buf[64];
    if (!read_text(std::string("/sys/bus/pci/devices/") + id + "/numa_node", buf, sizeof buf)) return -1;
    return atoi(buf);
}
// the CPUs of `node` as a cpu_set_t; false if unknown, empty or the whole machine anyway
static bool node_cpus(int node, cpu_set_t *set)
{
    if (node < 0) return false;
    char buf[4096];
    if (!read_text("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, sizeof buf)) return false;
    unsigned char mask[CPU_SETSIZE / 8];
    const int n = rans4x16_hip_cpulist_parse(buf, mask, (int)sizeof mask);
    if (n <= 0) return false;
    CPU_ZERO(set);
    for (int c = 0; c < CPU_SETSIZE; c++) if (mask[c >> 3] & (1u << (c & 7))) CPU_SET(c, set);
    return true;
}
// RAII: the calling thread on the device's node (threads it starts inherit the mask), its old mask back at the end
struct NodeBinding {
    cpu_set_t old;
    bool bound = false;
    explicit NodeBinding(int node)
    {
        const bool enabled = r4x16_opts_defaults()->v[OPT_NUMA] != 0;
        cpu_set_t want;
        if (!enabled || !node_cpus(node, &want)) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof old, &old) != 0) return;
        cpu_set_t both;
        CPU_AND(&both, &want, &old);                         // never outside what the process is allowed (cgroups, taskset)
        if (CPU_COUNT(&both) == 0 || CPU_EQUAL(&both, &old)) return;
        bound = pthread_setaffinity_np(pthread_self(), sizeof both, &both) == 0;
    }
    ~NodeBinding() { if (bound) (void)pthread_setaffinity_np(pthread_self(), sizeof old, &old); }
};

struct rans4x16_hip_multi {
    std::vector<rans4x16_hip_ctx *> ctx;
    std::vector<int> node;                  // NUMA node of each device (-1: unknown)
    std::string err;
};

extern "C" int rans4x16_hip_multi_numa_node(const rans4x16_hip_multi *m, int index)
{
    return (m && index >= 0 && index < (int)m->node.size()) ? m->node[(size_t)index] : -1;
}

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
        m->node.push_back(device_numa_node(c->device));
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
        // this device's copier threads and bounce buffers on the socket the device hangs off (header, part 3)
        NodeBinding bind(m->node[(size_t)p]);
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
