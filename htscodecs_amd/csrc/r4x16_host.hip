// r4x16_host.hip - the host-buffer batch machinery behind rans4x16_hip_{compress,uncompress}_batch and
// rans4x16_hip_compress_best_batch: staging through device arenas, the copier-thread pipeline for large batches,
// and X_STRIPE orchestration.  Orchestration only: every byte of codec work runs in the kernels of
// r4x16_encode.hip / r4x16_decode.hip through the *_dev entry points of r4x16_api.hip.
#include "r4x16_host.h"

// ---------------------------------------------------------------------------------------------
// host-buffer batches: stage through one device arena, run the *_dev path, copy results back.
// ---------------------------------------------------------------------------------------------
int r4x16_ensure_stage(rans4x16_hip_ctx *c, size_t bytes)
{
    if (bytes <= c->stage_bytes) return 0;
    if (c->stage) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(c->stage)); c->stage = nullptr; c->stage_bytes = 0; }
    // never more than half of what the card has free: sizes come from the caller's arrays, and through the
    // out == NULL decode entry from a size field of the stream itself (hostile input must not exhaust the device)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b / 2) {
        c->err = "host batch: staging of " + std::to_string(bytes >> 20) + " MiB exceeds half of the free device memory";
        return -1;
    }
    const hipError_t e = hipMalloc((void **)&c->stage, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->stage = nullptr;
        c->err = std::string("hipMalloc of the staging arena: ") + hipGetErrorString(e);
        return -1;
    }
    c->stage_bytes = bytes;
    return 0;
}

// Give back device memory above `keep` bytes (the single-block entry points call this after an unusually large
// block, so that one call - or one hostile size field - does not pin gigabytes to the calling thread for good).
void r4x16_trim(rans4x16_hip_ctx *c, size_t keep)
{
    if (!c || c->stage_bytes + c->ws_bytes <= keep) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->stage) { (void)hipFree(c->stage); c->stage = nullptr; c->stage_bytes = 0; }
    if (c->ws) { (void)hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
    if (c->xs) { (void)hipFree(c->xs); c->xs = nullptr; c->xs_bytes = 0; }
    c->ws_busy = false;
}

static int stripe_compress_many(rans4x16_hip_ctx *, const std::vector<int> &, const unsigned char *const *, const unsigned int *,
                                unsigned char *const *, unsigned int *, const int *, int *);
static int stripe_uncompress_many(rans4x16_hip_ctx *, const std::vector<int> &, const unsigned char *const *, const unsigned int *,
                                  unsigned char *const *, unsigned int *, int *);

static int stripe_many_dev(rans4x16_hip_ctx *, bool, const std::vector<int> &, const unsigned char *const *, const unsigned int *,
                           unsigned char *const *, unsigned int *, const int *, int *);
static int run_plain_batch(rans4x16_hip_ctx *c, int n, bool decode,
                           const unsigned char *const *in, const unsigned int *in_size,
                           unsigned char *const *out, unsigned int *out_size, const int *order, int *status);

static bool is_stripe(bool decode, const unsigned char *in, unsigned int in_size, int order)
{
    if (decode) return in_size > 0 && (in[0] & X_STRIPE);
    return (order & X_STRIPE) && in_size > 20;                         // :1151
}

int r4x16_run_host_batch(rans4x16_hip_ctx *c, int n, bool decode,
                          const unsigned char *const *in, const unsigned int *in_size,
                          unsigned char *const *out, unsigned int *out_size, const int *order, int *status)
{
    if (!c) return -1;
    if (n <= 0) return n == 0 ? 0 : -1;
    HIPCHK(c, hipSetDevice(c->device));
    // stripe blocks are expanded into one device batch of all their planes (and candidate methods)
    std::vector<int> plain, striped;
    int failed = 0;
    for (int i = 0; i < n; i++) {
        const int o = order ? order[i] : 0;
        if (is_stripe(decode, in[i], in_size[i], o)) striped.push_back(i); else plain.push_back(i);
    }
    if (!striped.empty()) {
        std::vector<int> sst(striped.size(), 0);
        const bool dev_route = c->opts.v[OPT_HOST_STRIPE_DEV] != 0;
        const int rc = dev_route ? stripe_many_dev(c, decode, striped, in, in_size, out, out_size, order, sst.data())
                     : decode ? stripe_uncompress_many(c, striped, in, in_size, out, out_size, sst.data())
                              : stripe_compress_many(c, striped, in, in_size, out, out_size, order, sst.data());
        if (rc < 0) return -1;
        for (size_t k = 0; k < striped.size(); k++) {
            const int i = striped[k];
            if (status) status[i] = sst[k] ? R4X16_E_SIZE : 0;
            if (sst[k]) { out_size[i] = 0; failed++; }
        }
    }
    if (plain.empty()) return failed;
    if ((int)plain.size() == n) {
        const int f = run_plain_batch(c, n, decode, in, in_size, out, out_size, order, status);
        return f < 0 ? -1 : failed + f;
    }
    const int m = (int)plain.size();
    std::vector<const unsigned char *> pin(m);
    std::vector<unsigned char *> pout(m);
    std::vector<unsigned int> pis(m), pos(m);
    std::vector<int> pord(m), pst(m);
    for (int k = 0; k < m; k++) {
        const int i = plain[k];
        pin[k] = in[i]; pout[k] = out[i]; pis[k] = in_size[i]; pos[k] = out_size[i]; pord[k] = order ? order[i] : 0;
    }
    const int f = run_plain_batch(c, m, decode, pin.data(), pis.data(), pout.data(), pos.data(), pord.data(), pst.data());
    if (f < 0) return -1;
    for (int k = 0; k < m; k++) {
        out_size[plain[k]] = pos[k];
        if (status) status[plain[k]] = pst[k];
    }
    return failed + f;
}

// One slab of blocks: copy in, run the device path, copy out, all on the context's own stream.
static int run_slab(rans4x16_hip_ctx *c, int n, bool decode,
                    const unsigned char *const *in, const unsigned int *in_size,
                    unsigned char *const *out, unsigned int *out_size, const int *order, int *status)
{
    hipStream_t s = c->stream;
    // arena: [in blocks][out slots][offset/size/status arrays]
    std::vector<u64> in_off(n), out_off(n);
    std::vector<u32> cap(n);
    std::vector<i32> ord(n);
    size_t in_tot = 0, out_tot = 0;
    u32 max_in = 0, max_cap = 0;
    u64 sum_in = 0, sum_cap = 0;
    for (int i = 0; i < n; i++) {
        in_off[i] = in_tot; in_tot += align_up((size_t)in_size[i] + 16, 256);
        cap[i] = out_size[i];
        out_off[i] = out_tot; out_tot += align_up((size_t)cap[i] + 16, 256);
        if (in_size[i] > max_in) max_in = in_size[i];
        // decode: only blocks with PACK / RLE need the stage buffers that max_out_cap sizes (r4x16_api.hip)
        const bool xfb = !decode || (in_size[i] && (in[i][0] & (X_PACK | X_RLE)));
        if (cap[i] > max_cap && xfb) max_cap = cap[i];
        sum_in += in_size[i];
        if (xfb) sum_cap += cap[i];
        ord[i] = order ? order[i] : 0;
    }
    const size_t arr = align_up((size_t)n * 8, 256);
    const size_t total = in_tot + out_tot + 6 * arr;
    if (r4x16_ensure_stage(c, total) != 0) return -1;
    u8 *d_in = c->stage, *d_out = d_in + in_tot, *meta = d_out + out_tot;
    u64 *d_in_off = (u64 *)meta, *d_out_off = (u64 *)(meta + arr);
    u32 *d_in_size = (u32 *)(meta + 2 * arr), *d_cap = (u32 *)(meta + 3 * arr), *d_osz = (u32 *)(meta + 4 * arr);
    i32 *d_status = (i32 *)(meta + 5 * arr);
    i32 *d_order = (i32 *)(meta + 5 * arr + arr / 2);

    for (int i = 0; i < n; i++)
        if (in_size[i]) HIPCHK(c, hipMemcpyAsync(d_in + in_off[i], in[i], in_size[i], hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_in_off, in_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_out_off, out_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_in_size, in_size, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_cap, cap.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_order, ord.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));

    int rc;
    if (decode)
        rc = rans4x16_hip_uncompress_dev_sized(c, n, d_in, d_in_off, d_in_size, d_out, d_out_off, d_cap, d_osz,
                                               d_status, max_in, max_cap, sum_cap, s);
    else
        rc = rans4x16_hip_compress_dev_sized(c, n, d_in, d_in_off, d_in_size, d_out, d_out_off, d_cap, d_osz,
                                             d_status, 0, d_order, max_in, sum_in, s);
    if (rc != 0) return -1;

    std::vector<u32> osz(n);
    std::vector<i32> st(n);
    HIPCHK(c, hipMemcpyAsync(osz.data(), d_osz, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(st.data(), d_status, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    int failed = 0;
    for (int i = 0; i < n; i++) {
        if (status) status[i] = st[i];
        if (st[i] != 0) { out_size[i] = 0; failed++; continue; }
        out_size[i] = osz[i];
        if (osz[i]) HIPCHK(c, hipMemcpyAsync(out[i], d_out + out_off[i], osz[i], hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return failed;
}

// ---------------------------------------------------------------------------------------------
// Large host batches: a staged pipeline instead of one copy-in / compute / copy-out pass.
//   * Copier threads move the callers' (pageable) buffers through their own pinned bounce buffers: a CPU
//     memcpy per core feeding true asynchronous DMA, both PCIe directions at once.  (Copies from pageable
//     memory issued straight to the runtime are staged by one runtime thread: ~20 GB/s both ways together.)
//     All copy-in DMA shares one stream and all copy-out DMA another: the runtime multiplexes streams onto
//     four hardware queues, and a copy that lands in the queue of a running 50 ms chain kernel waits for it
//     (measured: with a stream per copier thread, slabs were launched only as their predecessors finished).
//   * The batch is cut into a few slabs; the thread that queues a slab's last copy-in launches the slab's
//     kernels on one of the lane contexts (own stream + workspace), after events on the copier streams.
//   * A slab's results are copied out as soon as its kernels finish, beside later slabs' kernels.
// A chain kernel needs its 25-55 ms per MiB of block size however few blocks it is given, so slabs are
// large (up to 2 GiB of input + output capacity) and all lanes run at once.
// Options of the context (rans4x16_hip_set_option): host_pipe_mb (batches of at least this size, or of 32 blocks and
//        more, take this route; default 64; 0 = never), host_threads (default 8), host_lanes (default 2),
//        host_slab_min_mb (default 32).
// ---------------------------------------------------------------------------------------------

#define PIPE_CHUNK ((size_t)8 << 20)          // bytes per pinned bounce buffer

struct PackDesc { u64 src, dst; u32 len, pad; };      // one result to gather: slot offset, packed offset, bytes
struct PipeSlot {
    u8 *pin = nullptr;
    hipEvent_t ev = nullptr;
    bool busy = false;                        // a DMA batch is in flight, `ev` marks its end
    size_t fill = 0;
    struct Out { u8 *dst; size_t off, len; };
    std::vector<Out> outs;                    // copy-out: pinned -> caller once the DMA has landed
};
struct PipeCopier {
    PipeSlot slot[2];
    int k = 0;
};
struct HostPipe {
    std::vector<PipeCopier> cp;
    hipStream_t s_out = nullptr;              // copy-out DMA (copy-in uses the context's own stream)
    std::vector<rans4x16_hip_ctx *> lanes;
    std::mutex *lane_mu = nullptr;
    std::vector<hipEvent_t> events;           // grows; reused by every call
    u32 *h_osz = nullptr;                     // pinned: per-block output sizes and statuses
    i32 *h_st = nullptr;
    size_t h_n = 0;
    PackDesc *h_pk = nullptr;                 // pinned: one descriptor per block
    size_t h_pk_n = 0;
};

void r4x16_pipe_destroy(HostPipe *hp)
{
    if (!hp) return;
    if (hp->s_out) { (void)hipStreamSynchronize(hp->s_out); (void)hipStreamDestroy(hp->s_out); }
    for (auto &c : hp->cp) {
        for (auto &sl : c.slot) { if (sl.pin) (void)hipHostFree(sl.pin); if (sl.ev) (void)hipEventDestroy(sl.ev); }
    }
    for (auto *l : hp->lanes) rans4x16_hip_destroy(l);
    delete[] hp->lane_mu;
    for (auto e : hp->events) (void)hipEventDestroy(e);
    if (hp->h_osz) (void)hipHostFree(hp->h_osz);
    if (hp->h_st) (void)hipHostFree(hp->h_st);
    if (hp->h_pk) (void)hipHostFree(hp->h_pk);
    delete hp;
}

static int pipe_prepare(rans4x16_hip_ctx *c, int threads, int nlanes, size_t nevents, size_t n, size_t nblocks)
{
    if (!c->pipe) c->pipe = new HostPipe();
    HostPipe *hp = c->pipe;
    if (!hp->s_out) HIPCHK(c, hipStreamCreateWithFlags(&hp->s_out, hipStreamNonBlocking));
    while ((int)hp->cp.size() < threads) {
        PipeCopier pc;
        for (auto &sl : pc.slot) {
            HIPCHK(c, hipHostMalloc((void **)&sl.pin, PIPE_CHUNK, hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        }
        hp->cp.push_back(std::move(pc));
    }
    if ((int)hp->lanes.size() < nlanes) {
        if (!hp->lane_mu) hp->lane_mu = new std::mutex[16];
        while ((int)hp->lanes.size() < nlanes) {
            rans4x16_hip_ctx *l = rans4x16_hip_create(c->device);
            if (!l) { c->err = "host batch: cannot create a lane context"; return -1; }
            l->no_fork = true;          // (side streams of equal priority on two lanes would share a hardware queue)
            l->opts = c->opts;
            // The runtime keeps one pool of hardware queues per stream priority and multiplexes the streams of
            // a priority onto it; two lane streams of equal priority were seen sharing a queue, which runs their
            // kernels one after the other.  Lanes therefore take different priorities: different queues.
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least > greatest) {
                // (the copy streams have the default priority, the middle of the range: the first two lanes take
                //  the two ends, so that no copy shares a queue with a lane's kernels either)
                const int levels = least - greatest + 1;
                const int li = (int)(hp->lanes.size() % (size_t)levels);
                const int prio = li == 0 ? greatest : li == 1 ? least : greatest + li - 1;
                hipStream_t ps = nullptr;
                if (hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, prio) == hipSuccess) {
                    (void)hipStreamDestroy(l->stream);
                    l->stream = ps;
                }
            }
            hp->lanes.push_back(l);
        }
    }
    while (hp->events.size() < nevents) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hp->events.push_back(e);
    }
    if (hp->h_n < n) {
        if (hp->h_osz) { (void)hipHostFree(hp->h_osz); hp->h_osz = nullptr; }
        if (hp->h_st) { (void)hipHostFree(hp->h_st); hp->h_st = nullptr; }
        hp->h_n = 0;
        const size_t want = n + n / 2 + 1024;
        HIPCHK(c, hipHostMalloc((void **)&hp->h_osz, want * 4, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void **)&hp->h_st, want * 4, hipHostMallocDefault));
        hp->h_n = want;
    }
    if (hp->h_pk_n < nblocks) {
        if (hp->h_pk) { (void)hipHostFree(hp->h_pk); hp->h_pk = nullptr; }
        hp->h_pk_n = 0;
        const size_t want = nblocks + nblocks / 2 + 1024;
        HIPCHK(c, hipHostMalloc((void **)&hp->h_pk, want * sizeof(PackDesc), hipHostMallocDefault));
        hp->h_pk_n = want;
    }
    return 0;
}

struct PipeSlab {
    int lo = 0, hi = 0;                       // blocks [lo, hi)
    int gin = 1, gout = 1;                    // blocks per copy-in / copy-out unit
    int nin = 0, nout = 0;                    // units
    u32 max_in = 0, max_cap = 0;
    u64 sum_in = 0, sum_cap = 0;              // the slab's input bytes / the output capacities of its X_PACK and X_RLE blocks
    std::atomic<int> in_next{0}, in_queued{0}, out_next{0};
    std::atomic<int> launched{0}, finished{0};
    // copy-out plan, made once by the first thread that sees the slab's kernels finished:
    // 0 none yet, 1 being made, 4 pack kernel in flight, 2 ready (results packed), 3 ready (copied from their slots)
    std::atomic<int> plan{0};
    u64 pk_total = 0;                         // bytes of the packed results
    std::vector<u64> poff;                    // start of each block's result in the packed region
};

// Sparse results (encode: every block owns a bound-sized slot and fills a fraction of it) are gathered by the device
// into the slab's input region, which is dead once the slab's kernels have run, and cross PCIe as a few dense DMAs
// instead of one small DMA per block (15,000 small blocks: 140 ms of DMA calls before).
__global__ __launch_bounds__(256) void k_pack_results(const u8 *out, u8 *in, const PackDesc *d)
{
    const PackDesc p = d[blockIdx.x];
    const u8 *s = out + p.src;                // slots are 256-byte aligned; a packed result may start at any byte
    u8 *t = in + p.dst;
    const u32 n16 = p.len >> 4;
    for (u32 i = threadIdx.x; i < n16; i += 256) ((u32x4_unaligned *)t)[i] = ((const u32x4 *)s)[i];
    for (u32 i = (n16 << 4) + threadIdx.x; i < p.len; i += 256) t[i] = s[i];
}

// K > 1 (encode only) is the "try K methods, keep the smallest" mode of SURVEY §8f-3: the K candidates of a
// block share its one copy of the input, only the winner is copied out (first candidate wins ties,
// tokenise_name3.c:1283-1286), and win_k[i] receives its index in methods[].
static int run_pipelined(rans4x16_hip_ctx *c, int n, bool decode,
                         const unsigned char *const *in, const unsigned int *in_size,
                         unsigned char *const *out, unsigned int *out_size, const int *order, int *status,
                         int threads, int nlanes, int K = 1, const int *methods = nullptr, int *win_k = nullptr)
{
    // ---- layout: the same arena as run_slab; K output slots per block --------------------------------
    const size_t ni = (size_t)n * (size_t)K;
    if (ni > (size_t)INT_MAX) { c->err = "host batch: too many candidates"; return -1; }
    std::vector<u64> in_off(n), in_off_it(ni), out_off(ni);
    std::vector<u32> cap(ni), in_size_it(ni);
    std::vector<i32> ord(ni);
    size_t in_tot = 0, out_tot = 0;
    for (int i = 0; i < n; i++) {
        in_off[i] = in_tot; in_tot += align_up((size_t)in_size[i] + 16, 256);
        for (int k = 0; k < K; k++) {
            const size_t it = (size_t)i * K + k;
            in_off_it[it] = in_off[i];
            in_size_it[it] = in_size[i];
            cap[it] = out_size[i];
            out_off[it] = out_tot; out_tot += align_up((size_t)cap[it] + 16, 256);
            ord[it] = methods ? methods[k] : (order ? order[i] : 0);
        }
    }
    const size_t arr = align_up(ni * 8, 256);
    const size_t pk_bytes = align_up((size_t)n * sizeof(PackDesc), 256);
    if (r4x16_ensure_stage(c, in_tot + out_tot + 6 * arr + pk_bytes) != 0) return -1;
    u8 *d_in = c->stage, *d_out = d_in + in_tot, *meta = d_out + out_tot;
    u64 *d_in_off = (u64 *)meta, *d_out_off = (u64 *)(meta + arr);
    u32 *d_in_size = (u32 *)(meta + 2 * arr), *d_cap = (u32 *)(meta + 3 * arr), *d_osz = (u32 *)(meta + 4 * arr);
    i32 *d_status = (i32 *)(meta + 5 * arr);
    i32 *d_order = (i32 *)(meta + 5 * arr + arr / 2);
    PackDesc *d_pk = (PackDesc *)(meta + 6 * arr);
    std::vector<int> win(n, -1);                              // winning candidate of each block (K > 1)

    // ---- slabs: a multiple of the lane count, as few as the lane workspaces allow (8 GiB of input + capacity per
    // slab).  Every kernel of a slab has a latency floor that does not shrink with the slab - a chain kernel needs
    // the same time for one block as for a few thousand, the table kernels one wave-lifetime per block - so many
    // small slabs cost many floors (60,000 blocks of <= 64 KiB in slabs of 1,900: 717 ms; in four slabs: see §6).
    const size_t tot = in_tot + out_tot;
    const size_t rounds = (tot + (size_t)nlanes * ((size_t)8 << 30) - 1) / ((size_t)nlanes * ((size_t)8 << 30));
    // One slab per lane for both directions (R4X16_HOST_DEC_SLABS / R4X16_HOST_ENC_SLABS per lane).  Measured on 3,072 x
    // 1 MiB q40 blocks, round 3: encode 30.1 GB/s with one slab per lane against 26.8 with two and 19.0 with three (round
    // 2's default was two: its chain kernel needed 28 ms per slab whatever the slab, now 22 for slabs of up to 1,024
    // blocks and still a floor per slab); decode 23.3 with one, 19.7 with two, 14.9 with three.  Three decode slabs of
    // exactly 1,024 blocks on ONE lane reach 25.3 GB/s (each takes the short-step rows, 31 ms, and its output leaves
    // while the next runs) - but only for alphabets whose direct rows fit four streams per CU, which the host cannot
    // know; with any other data three serial slabs cost three full chain latencies.  Not adopted.
    // Round 4 (profiles/r04_host_timeline.md): what bounds a large batch is the copy-out (8.6 GB of decoded blocks at
    // ~48 GB/s against ~40 GB/s of copy-in for half as many bytes), and it cannot start before the first slab's kernels
    // are done - so slabs of about 2.3 GB of arena (input + capacity) each, however many that makes: 8,192 x 1 MiB decode
    // 27.9 -> 33.5 GB/s with six slabs instead of two, encode 37.3 -> 39.0 with four; a 3,072-block batch keeps its two.
    // host_dec_slabs / host_enc_slabs > 0 fix the slabs per lane instead.
    const long per_lane = decode ? c->opts.v[OPT_HOST_DEC_SLABS] : c->opts.v[OPT_HOST_ENC_SLABS];
    size_t nslab = per_lane > 0 ? (rounds ? rounds : 1) * (size_t)nlanes * (size_t)per_lane
                                : std::max<size_t>((size_t)nlanes, (tot + ((size_t)1150 << 20)) / ((size_t)2300 << 20));
    const long slab_min_mb = c->opts.v[OPT_HOST_SLAB_MIN_MB];
    while (nslab > 1 && tot / nslab < ((size_t)(slab_min_mb > 0 ? slab_min_mb : 1) << 20)) nslab--;
    if (nslab > (size_t)n) nslab = (size_t)n;
    std::vector<PipeSlab> slabs(nslab);
    {
        const size_t per = (tot + nslab - 1) / nslab;
        size_t acc = 0, j = 0;
        slabs[0].lo = 0;
        for (int i = 0; i < n; i++) {
            acc += align_up((size_t)in_size[i] + 16, 256) + (size_t)K * align_up((size_t)out_size[i] + 16, 256);
            const int left = n - (i + 1);
            if (j + 1 < nslab && (acc >= per * (j + 1) || left == (int)(nslab - j - 1))) {
                slabs[j].hi = i + 1;
                slabs[++j].lo = i + 1;
            }
        }
        slabs[j].hi = n;
        nslab = j + 1;
    }
    for (size_t j = 0; j < nslab; j++) {
        PipeSlab &S = slabs[j];
        u32 widest = 0;
        for (int i = S.lo; i < S.hi; i++) {
            if (in_size[i] > S.max_in) S.max_in = in_size[i];
            if (out_size[i] > widest) widest = out_size[i];
            // decode: only blocks with PACK / RLE need the stage buffers that max_out_cap sizes (r4x16_api.hip)
            const bool xfb = !decode || (in_size[i] && (in[i][0] & (X_PACK | X_RLE)));
            if (out_size[i] > S.max_cap && xfb) S.max_cap = out_size[i];
            S.sum_in += in_size[i];
            if (xfb) S.sum_cap += out_size[i];
        }
        const size_t gi = PIPE_CHUNK / (align_up((size_t)S.max_in + 16, 256));
        const size_t go = PIPE_CHUNK / ((size_t)widest + 64);
        S.gin = (int)(gi < 1 ? 1 : gi > 512 ? 512 : gi);
        S.gout = (int)(go < 1 ? 1 : go > 512 ? 512 : go);
        S.nin = (S.hi - S.lo + S.gin - 1) / S.gin;
        S.nout = (S.hi - S.lo + S.gout - 1) / S.gout;
    }
    if (pipe_prepare(c, threads, nlanes, 3 * nslab, ni, (size_t)n) != 0) return -2;      // nothing started: the caller may take the single-pass route
    HostPipe *hp = c->pipe;

    hipStream_t s0 = c->stream;
    const hipStream_t s_in = c->stream, s_out = hp->s_out;
    HIPCHK(c, hipMemcpyAsync(d_in_off, in_off_it.data(), ni * 8, hipMemcpyHostToDevice, s0));
    HIPCHK(c, hipMemcpyAsync(d_out_off, out_off.data(), ni * 8, hipMemcpyHostToDevice, s0));
    HIPCHK(c, hipMemcpyAsync(d_in_size, in_size_it.data(), ni * 4, hipMemcpyHostToDevice, s0));
    HIPCHK(c, hipMemcpyAsync(d_cap, cap.data(), ni * 4, hipMemcpyHostToDevice, s0));
    HIPCHK(c, hipMemcpyAsync(d_order, ord.data(), ni * 4, hipMemcpyHostToDevice, s0));
    HIPCHK(c, hipStreamSynchronize(s0));

    std::atomic<int> broken{0};
    std::mutex err_mu;
    const bool trace = c->opts.v[OPT_HOST_TRACE] != 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    if (trace) fprintf(stderr, "[pipe] %s n=%d slabs=%zu threads=%d lanes=%d in=%.1f MB cap=%.1f MB\n", decode ? "dec" : "enc", n, nslab, threads, nlanes, in_tot / 1e6, out_tot / 1e6);
    auto fail = [&](const char *what, hipError_t e) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!broken.exchange(1)) c->err = std::string("host batch pipeline: ") + what + ": " + hipGetErrorString(e);
    };
#define PIPECHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(#call, e_); return false; } } while (0)

    // wait for the slot's DMA batch, then hand copy-out bytes to the caller's buffers
    auto drain = [&](PipeSlot &sl) -> bool {
        if (sl.busy) {
            PIPECHK(hipEventSynchronize(sl.ev));
            for (auto &o : sl.outs) memcpy(o.dst, sl.pin + o.off, o.len);
            sl.busy = false;
        }
        sl.outs.clear();
        sl.fill = 0;
        return true;
    };
    // close the current slot (its DMAs are queued), move to the other one and make it free
    auto flip = [&](PipeCopier &pc, hipStream_t st) -> bool {
        PipeSlot &sl = pc.slot[pc.k];
        if (sl.fill) { PIPECHK(hipEventRecord(sl.ev, st)); sl.busy = true; }
        pc.k ^= 1;
        return drain(pc.slot[pc.k]);
    };

    auto launch_slab = [&](size_t j) -> bool {
        PipeSlab &S = slabs[j];
        const size_t li = j % (size_t)nlanes;
        rans4x16_hip_ctx *l = hp->lanes[li];
        std::lock_guard<std::mutex> g(hp->lane_mu[li]);
        {
            hipEvent_t e = hp->events[3 * j];              // every copy-in of this slab is queued on s_in
            PIPECHK(hipEventRecord(e, s_in));
            PIPECHK(hipStreamWaitEvent(l->stream, e, 0));
        }
        const int lo = S.lo * K, m = (S.hi - S.lo) * K;          // items
        int rc;
        if (decode)
            rc = rans4x16_hip_uncompress_dev_sized(l, m, d_in, d_in_off + lo, d_in_size + lo, d_out, d_out_off + lo, d_cap + lo,
                                                   d_osz + lo, d_status + lo, S.max_in, S.max_cap, S.sum_cap * (u64)K, l->stream);
        else
            rc = rans4x16_hip_compress_dev_sized(l, m, d_in, d_in_off + lo, d_in_size + lo, d_out, d_out_off + lo, d_cap + lo,
                                                 d_osz + lo, d_status + lo, 0, d_order + lo, S.max_in, S.sum_in * (u64)K, l->stream);
        if (rc != 0) {
            std::lock_guard<std::mutex> g2(err_mu);
            if (!broken.exchange(1)) c->err = l->err;
            return false;
        }
        PIPECHK(hipMemcpyAsync(hp->h_osz + lo, d_osz + lo, (size_t)m * 4, hipMemcpyDeviceToHost, l->stream));
        PIPECHK(hipMemcpyAsync(hp->h_st + lo, d_status + lo, (size_t)m * 4, hipMemcpyDeviceToHost, l->stream));
        PIPECHK(hipEventRecord(hp->events[3 * j + 1], l->stream));
        S.launched.store(1, std::memory_order_release);
        if (trace) fprintf(stderr, "[pipe] slab %zu (%d blocks) launched on lane %zu at %.1f ms\n", j, S.hi - S.lo, li, now_ms());
        return true;
    };

    // one copy-in unit: blocks [b0, b1) of slab j
    auto copy_in_unit = [&](PipeCopier &pc, int b0, int b1) -> bool {
        const u64 base = in_off[b0];
        const u64 extent = in_off[b1 - 1] + in_size[b1 - 1] - base;
        if (extent <= PIPE_CHUNK) {
            PipeSlot &sl = pc.slot[pc.k];                      // free: flip() drained it
            for (int i = b0; i < b1; i++)
                if (in_size[i]) memcpy(sl.pin + (in_off[i] - base), in[i], in_size[i]);
            if (extent) {
                PIPECHK(hipMemcpyAsync(d_in + base, sl.pin, extent, hipMemcpyHostToDevice, s_in));
                sl.fill = extent;
            }
            return flip(pc, s_in);
        }
        // a single block larger than a bounce buffer: piece by piece
        for (u64 p = 0; p < extent; p += PIPE_CHUNK) {
            const size_t len = (size_t)(extent - p < PIPE_CHUNK ? extent - p : PIPE_CHUNK);
            PipeSlot &sl = pc.slot[pc.k];
            memcpy(sl.pin, in[b0] + p, len);
            PIPECHK(hipMemcpyAsync(d_in + base + p, sl.pin, len, hipMemcpyHostToDevice, s_in));
            sl.fill = len;
            if (!flip(pc, s_in)) return false;
        }
        return true;
    };
    // one copy-out unit (the slab's kernels have finished, sizes and statuses are in pinned memory)
    auto copy_out_unit = [&](PipeCopier &pc, int b0, int b1) -> bool {
        // results that nearly fill their slots (decode: the capacity is the size) travel as one DMA over the
        // whole extent, like copy-in; sparse ones (encode: bound-sized slots) block by block
        if (K == 1) {
            size_t sum = 0;
            u64 end = out_off[b0];
            for (int i = b0; i < b1; i++)
                if (hp->h_st[i] == 0 && hp->h_osz[i]) { sum += hp->h_osz[i]; end = out_off[i] + hp->h_osz[i]; }
            const u64 base = out_off[b0], extent = end - base;
            if (sum && extent <= PIPE_CHUNK && sum * 4 >= extent * 3) {
                PipeSlot &sl = pc.slot[pc.k];                  // free: the previous unit ended with flip()
                PIPECHK(hipMemcpyAsync(sl.pin, d_out + base, extent, hipMemcpyDeviceToHost, s_out));
                for (int i = b0; i < b1; i++)
                    if (hp->h_st[i] == 0 && hp->h_osz[i]) sl.outs.push_back({out[i], (size_t)(out_off[i] - base), hp->h_osz[i]});
                sl.fill = extent;
                return flip(pc, s_out);
            }
        }
        for (int i = b0; i < b1; i++) {
            size_t it = (size_t)i;
            if (K > 1) {
                if (win[i] < 0) continue;                           // chosen by plan_slab
                it = (size_t)i * K + win[i];
            }
            if (hp->h_st[it] != 0) continue;
            const size_t sz = hp->h_osz[it];
            for (size_t p = 0; p < sz;) {
                PipeSlot *sl = &pc.slot[pc.k];
                size_t at = (sl->fill + 63) & ~(size_t)63;
                if (at >= PIPE_CHUNK || (PIPE_CHUNK - at < sz - p && at != 0)) {
                    if (!flip(pc, s_out)) return false;
                    sl = &pc.slot[pc.k];
                    at = 0;
                }
                const size_t len = sz - p < PIPE_CHUNK - at ? sz - p : PIPE_CHUNK - at;
                PIPECHK(hipMemcpyAsync(sl->pin + at, d_out + out_off[it] + p, len, hipMemcpyDeviceToHost, s_out));
                sl->outs.push_back({out[i] + p, at, len});
                sl->fill = at + len;
                p += len;
            }
        }
        return flip(pc, s_out);
    };

    // The copy-out plan of a slab whose kernels have finished (one thread makes it; sizes and statuses are in pinned
    // memory by now): pick the winners, then either pack the results on the device or leave them in their slots.
    auto plan_slab = [&](size_t j) -> bool {
        PipeSlab &S = slabs[j];
        const int m = S.hi - S.lo;
        S.poff.resize((size_t)m + 1);
        u64 T = 0, sum = 0;
        for (int i = S.lo; i < S.hi; i++) {
            size_t it = (size_t)i;
            if (K > 1) {
                int w = -1;
                for (int k = 0; k < K; k++) {
                    const size_t cand = (size_t)i * K + k;
                    if (hp->h_st[cand] == 0 && (w < 0 || hp->h_osz[cand] < hp->h_osz[(size_t)i * K + w])) w = k;
                }
                win[i] = w;
                it = (size_t)i * K + (w < 0 ? 0 : w);
            }
            const u32 sz = hp->h_st[it] == 0 ? hp->h_osz[it] : 0u;
            hp->h_pk[i].src = out_off[it];
            hp->h_pk[i].dst = in_off[S.lo] + T;
            hp->h_pk[i].len = sz;
            hp->h_pk[i].pad = 0;
            S.poff[(size_t)(i - S.lo)] = T;
            T += ((u64)sz + 63u) & ~(u64)63u;
            sum += sz;
        }
        S.poff[(size_t)m] = T;
        S.pk_total = T;
        const u64 in_region = in_off[S.hi - 1] + align_up((size_t)in_size[S.hi - 1] + 16, 256) - in_off[S.lo];
        const u64 out_extent = out_off[(size_t)(S.hi - 1) * K + (K - 1)] + cap[(size_t)(S.hi - 1) * K + (K - 1)] - out_off[(size_t)S.lo * K];
        const bool dense = K == 1 && sum * 4 >= out_extent * 3;       // decode: capacity = size, slots are adjacent
        if (dense || T > in_region || c->opts.v[OPT_HOST_PACK] == 0) {
            S.plan.store(3, std::memory_order_release);
            return true;
        }
        S.nout = (int)((T + PIPE_CHUNK - 1) / PIPE_CHUNK);
        if (T == 0) { S.plan.store(2, std::memory_order_release); return true; }
        const size_t li = j % (size_t)nlanes;
        rans4x16_hip_ctx *l = hp->lanes[li];
        std::lock_guard<std::mutex> g(hp->lane_mu[li]);
        PIPECHK(hipMemcpyAsync(d_pk + S.lo, hp->h_pk + S.lo, (size_t)m * sizeof(PackDesc), hipMemcpyHostToDevice, l->stream));
        hipLaunchKernelGGL(k_pack_results, dim3((unsigned)m), dim3(256), 0, l->stream, (const u8 *)d_out, d_in, (const PackDesc *)(d_pk + S.lo));
        PIPECHK(hipGetLastError());
        PIPECHK(hipEventRecord(hp->events[3 * j + 2], l->stream));
        S.plan.store(4, std::memory_order_release);
        return true;
    };
    // one copy-out unit of a packed slab: PIPE_CHUNK bytes of the packed region, handed out to the blocks they belong to
    auto copy_out_packed = [&](PipeCopier &pc, PipeSlab &S, int u) -> bool {
        const u64 a = (u64)u * PIPE_CHUNK, b = a + PIPE_CHUNK < S.pk_total ? a + PIPE_CHUNK : S.pk_total;
        PipeSlot &sl = pc.slot[pc.k];                              // free: the previous unit ended with flip()
        PIPECHK(hipMemcpyAsync(sl.pin, d_in + in_off[S.lo] + a, (size_t)(b - a), hipMemcpyDeviceToHost, s_out));
        // first block whose result ends after a
        size_t i = (size_t)(std::upper_bound(S.poff.begin(), S.poff.end(), a) - S.poff.begin());
        i = i ? i - 1 : 0;
        const size_t m = (size_t)(S.hi - S.lo);
        for (; i < m && S.poff[i] < b; i++) {
            const u64 r0 = S.poff[i], r1 = r0 + hp->h_pk[(size_t)S.lo + i].len;
            const u64 x0 = r0 > a ? r0 : a, x1 = r1 < b ? r1 : b;
            if (x1 > x0) sl.outs.push_back({out[(size_t)S.lo + i] + (x0 - r0), (size_t)(x0 - a), (size_t)(x1 - x0)});
        }
        sl.fill = (size_t)(b - a);
        return flip(pc, s_out);
    };

    auto worker = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess) { fail("hipSetDevice", hipErrorInvalidDevice); return; }
        PipeCopier &pc = hp->cp[t];
        pc.k = 0;
        for (auto &sl : pc.slot) { sl.busy = false; sl.fill = 0; sl.outs.clear(); }
        size_t in_low = 0, out_low = 0;
        // Copy-in feeds the kernels and copy-out drains them; the even threads look for copy-in work first, the
        // odd ones for copy-out, and each takes the other kind when its own has nothing ready.
        const bool in_first = (t & 1) == 0;
        auto take_out = [&](bool &err) -> bool {
            auto drained = [&](PipeSlab &S) { return S.plan.load(std::memory_order_acquire) >= 2 && S.plan.load() != 4 && S.out_next.load(std::memory_order_relaxed) >= S.nout; };
            while (out_low < nslab && drained(slabs[out_low])) out_low++;
            for (size_t j = out_low; j < nslab; j++) {
                PipeSlab &S = slabs[j];
                int pl = S.plan.load(std::memory_order_acquire);
                if (pl == 0) {
                    if (!S.launched.load(std::memory_order_acquire)) continue;
                    if (!S.finished.load(std::memory_order_acquire)) {
                        const hipError_t q = hipEventQuery(hp->events[3 * j + 1]);
                        if (q == hipErrorNotReady) continue;
                        if (q != hipSuccess) { fail("hipEventQuery", q); err = true; return false; }
                        if (!S.finished.exchange(1) && trace) fprintf(stderr, "[pipe] slab %zu kernels seen finished at %.1f ms\n", j, now_ms());
                    }
                    int expect = 0;
                    if (!S.plan.compare_exchange_strong(expect, 1)) continue;
                    if (!plan_slab(j)) { err = true; return false; }
                    pl = S.plan.load(std::memory_order_acquire);
                }
                if (pl == 1) continue;
                if (pl == 4) {
                    const hipError_t q = hipEventQuery(hp->events[3 * j + 2]);
                    if (q == hipErrorNotReady) continue;
                    if (q != hipSuccess) { fail("hipEventQuery", q); err = true; return false; }
                    int expect = 4;
                    if (S.plan.compare_exchange_strong(expect, 2) && trace)
                        fprintf(stderr, "[pipe] slab %zu results packed (%.1f MB) at %.1f ms\n", j, S.pk_total / 1e6, now_ms());
                    pl = 2;
                }
                if (S.out_next.load(std::memory_order_relaxed) >= S.nout) continue;
                const int u = S.out_next.fetch_add(1);
                if (u >= S.nout) continue;
                if (pl == 2) {
                    if (!copy_out_packed(pc, S, u)) { err = true; return false; }
                } else {
                    const int b0 = S.lo + u * S.gout, b1 = b0 + S.gout < S.hi ? b0 + S.gout : S.hi;
                    if (!copy_out_unit(pc, b0, b1)) { err = true; return false; }
                }
                return true;
            }
            return false;
        };
        auto take_in = [&](bool &err) -> bool {
            for (;;) {
                while (in_low < nslab && slabs[in_low].in_next.load(std::memory_order_relaxed) >= slabs[in_low].nin) in_low++;
                if (in_low >= nslab) return false;
                PipeSlab &S = slabs[in_low];
                const int u = S.in_next.fetch_add(1);
                if (u >= S.nin) continue;
                const int b0 = S.lo + u * S.gin, b1 = b0 + S.gin < S.hi ? b0 + S.gin : S.hi;
                if (!copy_in_unit(pc, b0, b1)) { err = true; return false; }
                if (S.in_queued.fetch_add(1, std::memory_order_acq_rel) + 1 == S.nin)
                    if (!launch_slab(in_low)) { err = true; return false; }
                return true;
            }
        };
        while (!broken.load(std::memory_order_relaxed)) {
            bool err = false;
            bool did = in_first ? take_in(err) : take_out(err);
            if (err) return;
            if (!did) did = in_first ? take_out(err) : take_in(err);
            if (err) return;
            if (did) continue;
            if (out_low >= nslab) break;                       // every unit has been taken
            // nothing to take yet: finish what this thread has in flight, then wait for a kernel
            if (!flip(pc, s_in) || !flip(pc, s_in)) return;
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        if (!broken.load()) { if (flip(pc, s_in)) (void)flip(pc, s_in); }
    };

    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto &t : th) t.join();
    if (trace) fprintf(stderr, "[pipe] done at %.1f ms\n", now_ms());
#undef PIPECHK
    if (broken.load()) {
        (void)hipDeviceSynchronize();
        return -1;
    }
    int failed = 0;
    for (int i = 0; i < n; i++) {
        size_t it = (size_t)i;
        if (K > 1) {
            if (win_k) win_k[i] = win[i];
            it = (size_t)i * K + (win[i] < 0 ? 0 : win[i]);
        }
        if (status) status[i] = hp->h_st[it];
        if (hp->h_st[it] != 0) { out_size[i] = 0; failed++; }
        else out_size[i] = hp->h_osz[it];
    }
    return failed;
}

static int run_plain_batch(rans4x16_hip_ctx *c, int n, bool decode,
                           const unsigned char *const *in, const unsigned int *in_size,
                           unsigned char *const *out, unsigned int *out_size, const int *order, int *status)
{
    const long pipe_mb = c->opts.v[OPT_HOST_PIPE_MB];
    long threads = c->opts.v[OPT_HOST_THREADS], nlanes = c->opts.v[OPT_HOST_LANES];
    threads = threads < 1 ? 1 : threads > 32 ? 32 : threads;
    nlanes = nlanes < 1 ? 1 : nlanes > 16 ? 16 : nlanes;
    size_t tot = 0;
    for (int i = 0; i < n; i++) tot += (size_t)in_size[i] + out_size[i];
    // small batches of few blocks keep the single pass (no threads to start); many small blocks take the pipeline
    // whatever their total, because the single pass issues one driver copy per block and direction
    // (5,000 x 4 KiB: 197 ms single pass, 35 ms pipelined; tools/small_batch_routes.py)
    if (pipe_mb <= 0 || (tot < ((size_t)pipe_mb << 20) && n < 32))
        return run_slab(c, n, decode, in, in_size, out, out_size, order, status);
    const long by_work = (long)(tot >> 21) + n / 64 + 1;           // a copier thread per 2 MiB / 64 blocks is plenty
    if (threads > by_work) threads = by_work;
    const int rc = run_pipelined(c, n, decode, in, in_size, out, out_size, order, status, (int)threads, (int)nlanes);
    // -2: the pipeline's resources (pinned buffers, lane contexts) could not be set up, e.g. a locked-memory limit
    return rc == -2 ? run_slab(c, n, decode, in, in_size, out, out_size, order, status) : rc;
}

// ---------------------------------------------------------------------------------------------
// X_STRIPE (rANS_static4x16pr.c:1154-1216, :1360-1433) is orchestration around N ordinary
// sub-blocks, so it lives on the host side of the batch machinery: the byte planes are split /
// joined by a device kernel, every (plane, candidate method) pair is one block of a device batch,
// and the host only compares the resulting sizes and lays out the header.
// ---------------------------------------------------------------------------------------------
static int var_put_host(unsigned char *cp, u32 v)
{
    int groups = 1;
    for (u32 t = v >> 7; t; t >>= 7) groups++;
    for (int g = groups - 1; g >= 0; g--) *cp++ = (unsigned char)(((v >> (7 * g)) & 0x7f) | (g ? 0x80 : 0));
    return groups;
}
static int var_get_host(const unsigned char *cp, const unsigned char *endp, u32 *v)
{
    const unsigned char *op = cp;
    u32 j = 0;
    unsigned char ch;
    if (cp >= endp) { *v = 0; return 0; }
    do { ch = *cp++; j = (j << 7) | (ch & 0x7f); } while ((ch & 0x80) && cp < endp);
    *v = j;
    return (int)(cp - op);
}

// ---------------------------------------------------------------------------------------------
// X_STRIPE blocks of a host batch through the DEVICE-RESIDENT stripe route (r4x16_stripe.hip, round 3): the blocks are
// staged like any others and the prepare / pick / join kernels do on the device what stripe_compress_many /
// stripe_uncompress_many below do with a read-back of the plane sizes in the middle of the call (those stay as the
// route of R4X16_HOST_STRIPE_DEV=0).  Encode: the device route takes one `order` per call (N and the candidate methods
// follow from it), so the blocks are grouped by their order value - one call per value, nearly always one.  Decode:
// the host has the streams, so it reads every block's plane count from its header and reserves the largest.
// ---------------------------------------------------------------------------------------------
static int stripe_many_dev(rans4x16_hip_ctx *c, bool decode, const std::vector<int> &which,
                           const unsigned char *const *in, const unsigned int *in_size,
                           unsigned char *const *out, unsigned int *out_size, const int *order, int *fail)
{
    const size_t nb = which.size();
    std::vector<u8> ok(nb, 0);
    std::vector<int> key(nb, 0);                             // encode: the order value; decode: 0
    u32 max_planes = 0, max_ulen = 0;
    for (size_t k = 0; k < nb; k++) {
        const int i = which[k];
        fail[k] = 1;
        if (!decode) {
            const int o = order ? order[i] : 0;
            int N = o >> 8; if (N == 0) N = 4;
            if (N > 255 || out_size[i] < r4x16_compress_bound(in_size[i], o)) continue;          // :1158
            ok[k] = 1; key[k] = o;
        } else {
            // the checks the reference makes before it touches a plane (:1360-1400), on the host as before
            const unsigned char *p = in[i], *end = p + in_size[i];
            u32 ulen, hdr = 1;
            hdr += var_get_host(p + hdr, end, &ulen);
            if (hdr >= in_size[i]) continue;                               // :1367
            const u32 N = p[hdr++];
            if (ulen != out_size[i]) continue;                             // :1379 (caller sized the buffer)
            if (N == 0) { if (ulen == 0) { fail[k] = 0; out_size[i] = 0; } continue; }   // the reference spins forever here
            u64 ctot = 0;
            bool good = true;
            for (u32 j = 0; j < N; j++) {
                u32 cl;
                hdr += var_get_host(p + hdr, end, &cl);
                ctot += cl;
                if (hdr > in_size[i] || cl > in_size[i] || cl < 1) { good = false; break; }   // :1389
            }
            if (!good || hdr + ctot > in_size[i]) continue;                // :1398
            ok[k] = 1;
            if (N > max_planes) max_planes = N;
            if (ulen > max_ulen) max_ulen = ulen;
        }
    }
    // groups of equal key, in order of first appearance
    std::vector<int> keys;
    for (size_t k = 0; k < nb; k++)
        if (ok[k] && std::find(keys.begin(), keys.end(), key[k]) == keys.end()) keys.push_back(key[k]);
    hipStream_t s = c->stream;
    for (const int kv : keys) {
        std::vector<size_t> g;
        for (size_t k = 0; k < nb; k++) if (ok[k] && key[k] == kv) g.push_back(k);
        const int m = (int)g.size();
        std::vector<u64> in_off(m), out_off(m);
        std::vector<u32> isz(m), cap(m);
        size_t in_tot = 0, out_tot = 0;
        u32 max_in = 0, max_cap = 0;
        for (int e = 0; e < m; e++) {
            const int i = which[g[e]];
            in_off[e] = in_tot; in_tot += align_up((size_t)in_size[i] + 16, 256);
            isz[e] = in_size[i]; cap[e] = out_size[i];
            out_off[e] = out_tot; out_tot += align_up((size_t)cap[e] + 16, 256);
            if (isz[e] > max_in) max_in = isz[e];
            if (cap[e] > max_cap) max_cap = cap[e];
        }
        const size_t arr = align_up((size_t)m * 8, 256);
        if (r4x16_ensure_stage(c, in_tot + out_tot + 6 * arr) != 0) return -1;
        u8 *d_in = c->stage, *d_out = d_in + in_tot, *meta = d_out + out_tot;
        u64 *d_in_off = (u64 *)meta, *d_out_off = (u64 *)(meta + arr);
        u32 *d_isz = (u32 *)(meta + 2 * arr), *d_cap = (u32 *)(meta + 3 * arr), *d_osz = (u32 *)(meta + 4 * arr);
        i32 *d_st = (i32 *)(meta + 5 * arr);
        for (int e = 0; e < m; e++)
            HIPCHK(c, hipMemcpyAsync(d_in + in_off[e], in[which[g[e]]], isz[e], hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(d_in_off, in_off.data(), (size_t)m * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(d_out_off, out_off.data(), (size_t)m * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(d_isz, isz.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(d_cap, cap.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
        int rc;
        if (decode) {
            const int keep_planes = c->dev_stripe_planes;
            const unsigned int keep_out = c->dev_stripe_out;
            c->dev_stripe_planes = (int)max_planes;
            c->dev_stripe_out = max_ulen;
            rc = rans4x16_hip_uncompress_dev(c, m, d_in, d_in_off, d_isz, d_out, d_out_off, d_cap, d_osz, d_st, max_in, max_cap, s);
            c->dev_stripe_planes = keep_planes;
            c->dev_stripe_out = keep_out;
        } else
            rc = rans4x16_hip_compress_dev(c, m, d_in, d_in_off, d_isz, d_out, d_out_off, d_cap, d_osz, d_st, kv, nullptr, max_in, s);
        if (rc != 0) return -1;
        std::vector<u32> osz(m);
        std::vector<i32> st(m);
        HIPCHK(c, hipMemcpyAsync(osz.data(), d_osz, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(st.data(), d_st, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        // results: decode slots are full, so the slot region comes back as it is; encode results fill a fraction of
        // their bound-sized slots and are first gathered into the (by now dead) input region by k_pack_results - one
        // dense transfer instead of a DMA per block (1,000 blocks: 85 ms of driver calls)
        std::vector<u8> host;
        if (decode) { host.resize(out_tot); HIPCHK(c, hipMemcpyAsync(host.data(), d_out, out_tot, hipMemcpyDeviceToHost, s)); }
        HIPCHK(c, hipStreamSynchronize(s));
        std::vector<PackDesc> pk;
        u64 T = 0;
        if (!decode) {
            for (int e = 0; e < m; e++) {
                if (st[e] != 0 || !osz[e]) continue;
                PackDesc d; d.src = out_off[e]; d.dst = T; d.len = osz[e]; d.pad = (u32)e;
                pk.push_back(d);
                T += ((u64)osz[e] + 63u) & ~(u64)63u;
            }
            const size_t pk_bytes = pk.size() * sizeof(PackDesc);
            if (!pk.empty() && T <= in_tot && pk_bytes <= 6 * arr) {
                // (the descriptors go where the offset / size arrays were: nothing reads those any more)
                PackDesc *d_pk = (PackDesc *)meta;
                HIPCHK(c, hipMemcpyAsync(d_pk, pk.data(), pk_bytes, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_pack_results, dim3((unsigned)pk.size()), dim3(256), 0, s, (const u8 *)d_out, d_in, (const PackDesc *)d_pk);
                HIPCHK(c, hipGetLastError());
                host.resize(T);
                HIPCHK(c, hipMemcpyAsync(host.data(), d_in, T, hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
            } else {
                for (const PackDesc &d : pk) HIPCHK(c, hipMemcpyAsync(out[which[g[d.pad]]], d_out + d.src, d.len, hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
                pk.clear();
            }
        }
        for (int e = 0; e < m; e++) {
            const size_t k = g[e];
            const int i = which[k];
            if (st[e] != 0) continue;                                          // fail[k] stays 1
            if (decode) {
                if (osz[e] != cap[e]) continue;
                if (osz[e]) memcpy(out[i], host.data() + out_off[e], osz[e]);
            }
            out_size[i] = osz[e];
            fail[k] = 0;
        }
        for (const PackDesc &d : pk) memcpy(out[which[g[d.pad]]], host.data() + d.dst, d.len);
    }
    return 0;
}

// returns 0 on success (out/out_size filled), 1 if the block failed, -1 on runtime errors
// All X_STRIPE blocks of a batch at once.  `which` lists their indices in the caller's arrays; fail[k] != 0 marks
// block which[k] as failed (the caller reports R4X16_E_SIZE, as for every stripe-level inconsistency).
// Encode (:1154-1216): every block is split into its N byte planes by a device kernel, every (plane, candidate
// method) pair of every block is one item of ONE device batch, the host compares the sizes (smallest wins, the first
// on ties, :1199), writes the headers, and the winners - gathered into one dense region by k_pack_results - come back
// in a single transfer.  (Block by block, with a device round trip each, 1,000 blocks of 64 KiB took 1.2 s.)
static int stripe_compress_many(rans4x16_hip_ctx *c, const std::vector<int> &which,
                                const unsigned char *const *in, const unsigned int *in_size,
                                unsigned char *const *out, unsigned int *out_size, const int *order, int *fail)
{
    static const int methods[4] = {1, 64, 128, 0};                     // :1192
    struct Blk { int N, K, item0; int cand[4]; u64 boff; bool ok; };
    const size_t nb = which.size();
    std::vector<Blk> B(nb);
    size_t items = 0, in_tot = 0;
    for (size_t k = 0; k < nb; k++) {
        const int i = which[k], o = order ? order[i] : 0;
        Blk &b = B[k];
        b.N = o >> 8; if (b.N == 0) b.N = 4;
        b.K = 0;
        for (int j = 0; j < 4; j++) if ((o & methods[j]) == methods[j]) b.cand[b.K++] = methods[j];
        b.ok = b.N <= 255 && out_size[i] >= r4x16_compress_bound(in_size[i], o);     // :1158
        fail[k] = b.ok ? 0 : 1;
        b.item0 = (int)items;
        b.boff = in_tot;
        if (b.ok) { items += (size_t)b.N * b.K; in_tot += align_up((size_t)in_size[i] + 16, 256); }
    }
    if (items == 0) return 0;
    if (items > (size_t)INT_MAX) { c->err = "stripe batch: too many planes"; return -1; }
    std::vector<u64> in_off(items), out_off(items);
    std::vector<u32> isz(items), cap(items);
    std::vector<i32> ord(items);
    size_t out_tot = 0;
    u32 max_in = 0;
    for (size_t k = 0; k < nb; k++) {
        const Blk &b = B[k];
        if (!b.ok) continue;
        const u32 n = in_size[which[k]];
        u32 first = 0;
        for (int j = 0; j < b.N; j++) {
            const u32 part = n / (u32)b.N + ((n % (u32)b.N) > (u32)j);
            for (int q = 0; q < b.K; q++) {
                const size_t it = (size_t)b.item0 + (size_t)j * b.K + q;
                in_off[it] = b.boff + first; isz[it] = part; ord[it] = b.cand[q] | X_NOSZ;
                cap[it] = r4x16_compress_bound(part, ord[it]);
                out_off[it] = out_tot; out_tot += align_up((size_t)cap[it] + 16, 256);
            }
            if (part > max_in) max_in = part;
            first += part;
        }
    }
    const size_t arr = align_up(items * 8, 256), pk_bytes = align_up(items * sizeof(PackDesc), 256);
    if (r4x16_ensure_stage(c, 2 * in_tot + out_tot + 6 * arr + pk_bytes) != 0) return -1;
    u8 *d_in = c->stage, *d_pl = d_in + in_tot, *d_out = d_pl + in_tot, *meta = d_out + out_tot;
    u64 *d_in_off = (u64 *)meta, *d_out_off = (u64 *)(meta + arr);
    u32 *d_isz = (u32 *)(meta + 2 * arr), *d_cap = (u32 *)(meta + 3 * arr), *d_osz = (u32 *)(meta + 4 * arr);
    i32 *d_st = (i32 *)(meta + 5 * arr), *d_ord = (i32 *)(meta + 5 * arr + arr / 2);
    PackDesc *d_pk = (PackDesc *)(meta + 6 * arr);
    hipStream_t s = c->stream;
    for (size_t k = 0; k < nb; k++) {
        if (!B[k].ok) continue;
        const int i = which[k];
        HIPCHK(c, hipMemcpyAsync(d_in + B[k].boff, in[i], in_size[i], hipMemcpyHostToDevice, s));
        r4x16_launch_stripe(d_in + B[k].boff, d_pl + B[k].boff, in_size[i], (u32)B[k].N, 0, s);
    }
    HIPCHK(c, hipMemcpyAsync(d_in_off, in_off.data(), items * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_out_off, out_off.data(), items * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_isz, isz.data(), items * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_cap, cap.data(), items * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_ord, ord.data(), items * 4, hipMemcpyHostToDevice, s));
    if (rans4x16_hip_compress_dev(c, (int)items, d_pl, d_in_off, d_isz, d_out, d_out_off, d_cap, d_osz, d_st,
                                  0, d_ord, max_in, s) != 0) return -1;
    std::vector<u32> osz(items);
    std::vector<i32> st(items);
    HIPCHK(c, hipMemcpyAsync(osz.data(), d_osz, items * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(st.data(), d_st, items * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));

    // headers on the host, winners packed on the device (into the input + plane regions, dead by now)
    std::vector<PackDesc> pk;
    std::vector<unsigned int> hdr_len(nb, 0);
    pk.reserve(items);
    u64 T = 0;
    for (size_t k = 0; k < nb; k++) {
        const Blk &b = B[k];
        if (!b.ok) continue;
        const int i = which[k];
        bool good = true;
        for (int it = 0; it < b.N * b.K; it++) if (st[(size_t)b.item0 + it] != 0) good = false;
        if (!good) { fail[k] = 1; continue; }
        unsigned char *o = out[i];
        unsigned int hdr = 1;
        o[0] = (unsigned char)((order ? order[i] : 0) & ~X_NOSZ);       // :1185
        hdr += var_put_host(o + hdr, in_size[i]);
        o[hdr++] = (unsigned char)b.N;
        u64 body = 0;
        for (int j = 0; j < b.N; j++) {                                // smallest wins, first on ties (:1199)
            u32 best_sz = in_size[i] + 10;
            int best = 0;
            for (int q = 0; q < b.K; q++) {
                const u32 z = osz[(size_t)b.item0 + (size_t)j * b.K + q];
                if (best_sz > z) { best_sz = z; best = q; }
            }
            const size_t it = (size_t)b.item0 + (size_t)j * b.K + best;
            hdr += var_put_host(o + hdr, osz[it]);
            PackDesc d; d.src = out_off[it]; d.dst = T + body; d.len = osz[it]; d.pad = (u32)k;
            pk.push_back(d);
            body += osz[it];
        }
        hdr_len[k] = hdr;
        T += (body + 63u) & ~(u64)63u;
    }
    if (pk.empty()) return 0;
    std::vector<u8> host(T);
    if (T <= 2 * (u64)in_tot) {
        HIPCHK(c, hipMemcpyAsync(d_pk, pk.data(), pk.size() * sizeof(PackDesc), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pack_results, dim3((unsigned)pk.size()), dim3(256), 0, s, (const u8 *)d_out, d_in, (const PackDesc *)d_pk);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(host.data(), d_in, T, hipMemcpyDeviceToHost, s));
    } else {
        for (const PackDesc &d : pk) HIPCHK(c, hipMemcpyAsync(host.data() + d.dst, d_out + d.src, d.len, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    // planes of a block are adjacent in the packed region, in order
    size_t q = 0;
    while (q < pk.size()) {
        const size_t k = pk[q].pad;
        const int i = which[k];
        const u64 start = pk[q].dst;
        u64 body = 0;
        while (q < pk.size() && pk[q].pad == (u32)k) { body += pk[q].len; q++; }
        memcpy(out[i] + hdr_len[k], host.data() + start, body);
        out_size[i] = hdr_len[k] + (unsigned int)body;
    }
    return 0;
}

// Decode (:1360-1433): headers are parsed on the host, every plane of every block is one item of ONE device batch,
// a device kernel per block interleaves its planes, and the results come back in a single transfer.
static int stripe_uncompress_many(rans4x16_hip_ctx *c, const std::vector<int> &which,
                                  const unsigned char *const *in, const unsigned int *in_size,
                                  unsigned char *const *out, unsigned int *out_size, int *fail)
{
    struct Blk { u32 N, ulen, used, hdr; int item0; u64 boff, poff; bool ok; };
    const size_t nb = which.size();
    std::vector<Blk> B(nb);
    std::vector<u32> clen_all;
    size_t items = 0, in_tot = 0, pl_tot = 0;
    for (size_t k = 0; k < nb; k++) {
        const int i = which[k];
        Blk &b = B[k];
        b.ok = false; fail[k] = 1; b.item0 = (int)items; b.boff = in_tot; b.poff = pl_tot; b.N = 0;
        const unsigned char *p = in[i], *end = p + in_size[i];
        u32 ulen, hdr = 1;
        hdr += var_get_host(p + hdr, end, &ulen);
        if (hdr >= in_size[i]) continue;                               // :1367
        const u32 N = p[hdr++];
        if (ulen != out_size[i]) continue;                             // :1379 (caller sized the buffer)
        if (N == 0) { if (ulen == 0) { fail[k] = 0; out_size[i] = 0; } continue; }   // the reference spins forever here
        u64 ctot = 0;
        bool good = true;
        const size_t c0 = clen_all.size();
        for (u32 j = 0; j < N; j++) {
            u32 cl;
            hdr += var_get_host(p + hdr, end, &cl);
            clen_all.push_back(cl);
            ctot += cl;
            if (hdr > in_size[i] || cl > in_size[i] || cl < 1) { good = false; break; }   // :1389
        }
        if (!good || hdr + ctot > in_size[i]) { clen_all.resize(c0); continue; }          // :1398
        b.N = N; b.ulen = ulen; b.hdr = hdr; b.used = (u32)(hdr + ctot); b.ok = true;
        items += N;
        in_tot += align_up((size_t)b.used + 16, 256);
        pl_tot += align_up((size_t)ulen + 16, 256);
    }
    if (items == 0) return 0;
    if (items > (size_t)INT_MAX) { c->err = "stripe batch: too many planes"; return -1; }
    // sub-block j starts at hdr + sum(clen[<j]) and may read to the end of the stripe block (:1419)
    std::vector<u64> in_off(items), out_off(items);
    std::vector<u32> isz(items), cap(items);
    u32 max_in = 0, max_cap = 0;
    {
        size_t ci = 0;
        for (size_t k = 0; k < nb; k++) {
            const Blk &b = B[k];
            if (!b.ok) continue;
            u32 off = b.hdr, first = 0;
            for (u32 j = 0; j < b.N; j++) {
                const size_t it = (size_t)b.item0 + j;
                const u32 plen = b.ulen / b.N + ((b.ulen % b.N) > j);
                in_off[it] = b.boff + off; isz[it] = b.used - off; cap[it] = plen; out_off[it] = b.poff + first;
                if (isz[it] > max_in) max_in = isz[it];
                if (plen > max_cap) max_cap = plen;
                off += clen_all[ci++];
                first += plen;
            }
        }
    }
    const size_t arr = align_up(items * 8, 256);
    if (r4x16_ensure_stage(c, in_tot + 2 * pl_tot + 6 * arr) != 0) return -1;
    u8 *d_in = c->stage, *d_pl = d_in + in_tot, *d_out = d_pl + pl_tot, *meta = d_out + pl_tot;
    u64 *d_in_off = (u64 *)meta, *d_out_off = (u64 *)(meta + arr);
    u32 *d_isz = (u32 *)(meta + 2 * arr), *d_cap = (u32 *)(meta + 3 * arr), *d_osz = (u32 *)(meta + 4 * arr);
    i32 *d_st = (i32 *)(meta + 5 * arr);
    hipStream_t s = c->stream;
    for (size_t k = 0; k < nb; k++)
        if (B[k].ok) HIPCHK(c, hipMemcpyAsync(d_in + B[k].boff, in[which[k]], B[k].used, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_in_off, in_off.data(), items * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_out_off, out_off.data(), items * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_isz, isz.data(), items * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_cap, cap.data(), items * 4, hipMemcpyHostToDevice, s));
    if (rans4x16_hip_uncompress_dev(c, (int)items, d_in, d_in_off, d_isz, d_pl, d_out_off, d_cap, d_osz, d_st,
                                    max_in, max_cap, s) != 0) return -1;
    std::vector<u32> osz(items);
    std::vector<i32> st(items);
    HIPCHK(c, hipMemcpyAsync(osz.data(), d_osz, items * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(st.data(), d_st, items * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    bool any = false;
    for (size_t k = 0; k < nb; k++) {
        Blk &b = B[k];
        if (!b.ok) continue;
        for (u32 j = 0; j < b.N; j++)
            if (st[(size_t)b.item0 + j] != 0 || osz[(size_t)b.item0 + j] != cap[(size_t)b.item0 + j]) b.ok = false;   // :1419-1420
        if (!b.ok) continue;
        r4x16_launch_stripe(d_pl + b.poff, d_out + b.poff, b.ulen, b.N, 1, s);
        any = true;
    }
    if (any) {
        std::vector<u8> host(pl_tot);
        HIPCHK(c, hipMemcpyAsync(host.data(), d_out, pl_tot, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        for (size_t k = 0; k < nb; k++) {
            const Blk &b = B[k];
            if (!b.ok) continue;
            const int i = which[k];
            if (b.ulen) memcpy(out[i], host.data() + b.poff, b.ulen);
            out_size[i] = b.ulen;
            fail[k] = 0;
        }
    }
    return 0;
}

// "Try K methods, keep the smallest" (tokenise_name3.c:1246-1300 compress()): every block is
// encoded with each of methods[0..k), the smallest result is delivered, the first method wins ties
// (:1283-1286), and X_STRIPE methods are skipped for blocks whose size is not a multiple of four (:1271-1272).
// The plain candidates of a block share one copy of its input on the device and only the winner crosses PCIe
// back; X_STRIPE candidates go block by block through the host-orchestrated stripe path.
extern "C" int rans4x16_hip_compress_best_batch(rans4x16_hip_ctx *c, int n,
                                                const unsigned char *const *in, const unsigned int *in_size,
                                                unsigned char *const *out, unsigned int *out_size,
                                                int k, const int *methods, int *chosen, int *status)
{
    if (!c) return -1;
    if (n < 0 || k <= 0 || !methods || (n && (!in || !in_size || !out || !out_size))) {
        c->err = "compress_best_batch: bad arguments";
        return -1;
    }
    if (n == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> plain_m, plain_idx, stripe_idx;
    for (int j = 0; j < k; j++) {
        if (methods[j] & X_STRIPE) stripe_idx.push_back(j);
        else { plain_m.push_back(methods[j]); plain_idx.push_back(j); }
    }
    std::vector<int> best(n, -1), st(n, R4X16_E_UNSUPPORTED);
    std::vector<unsigned int> capv(out_size, out_size + n);
    if (!plain_m.empty()) {
        long threads = c->opts.v[OPT_HOST_THREADS], nlanes = c->opts.v[OPT_HOST_LANES];
        threads = threads < 1 ? 1 : threads > 32 ? 32 : threads;
        nlanes = nlanes < 1 ? 1 : nlanes > 16 ? 16 : nlanes;
        size_t tot = 0;
        for (int i = 0; i < n; i++) tot += in_size[i];
        const long by_size = (long)(tot >> 22) + 1;                  // a copier thread per 4 MiB of input is plenty
        if (threads > by_size) threads = by_size;
        std::vector<int> wk(n, 0);
        const int K = (int)plain_m.size();
        if (run_pipelined(c, n, false, in, in_size, out, out_size, nullptr, st.data(), (int)threads, (int)nlanes,
                          K, plain_m.data(), wk.data()) < 0)
            return -1;                                               // (-2 included: this mode has no single-pass route)
        for (int i = 0; i < n; i++)
            if (st[i] == 0) best[i] = plain_idx[K > 1 ? wk[i] : 0];
    }
    if (!stripe_idx.empty()) {
        // every eligible block with one stripe method is one more batch (its planes and their candidates all in one
        // device batch); results land in scratch buffers and replace the winner so far where they are smaller
        std::vector<int> elig;
        for (int i = 0; i < n; i++) if (in_size[i] % 4 == 0) elig.push_back(i);      // :1271-1272
        const int m = (int)elig.size();
        if (m) {
            std::vector<size_t> toff(m);
            size_t ttot = 0;
            for (int e = 0; e < m; e++) { toff[e] = ttot; ttot += (size_t)capv[elig[e]] + 1; }
            std::vector<unsigned char> tmp(ttot);
            std::vector<const unsigned char *> bin(m);
            std::vector<unsigned char *> bout(m);
            std::vector<unsigned int> bisz(m), bosz(m);
            std::vector<int> bord(m), bst(m);
            for (int j : stripe_idx) {
                for (int e = 0; e < m; e++) {
                    const int i = elig[e];
                    bin[e] = in[i]; bisz[e] = in_size[i]; bout[e] = tmp.data() + toff[e]; bosz[e] = capv[i]; bord[e] = methods[j]; bst[e] = 0;
                }
                if (r4x16_run_host_batch(c, m, false, bin.data(), bisz.data(), bout.data(), bosz.data(), bord.data(), bst.data()) < 0) return -1;
                for (int e = 0; e < m; e++) {
                    const int i = elig[e];
                    if (bst[e] != 0) continue;
                    if (best[i] < 0 || bosz[e] < out_size[i] || (bosz[e] == out_size[i] && j < best[i])) {
                        memcpy(out[i], bout[e], bosz[e]);
                        out_size[i] = bosz[e];
                        best[i] = j;
                        st[i] = 0;
                    }
                }
            }
        }
    }
    int failed = 0;
    for (int i = 0; i < n; i++) {
        if (best[i] < 0) { out_size[i] = 0; failed++; if (st[i] == 0) st[i] = R4X16_E_UNSUPPORTED; }
        if (chosen) chosen[i] = best[i] < 0 ? -1 : methods[best[i]];
        if (status) status[i] = st[i];
    }
    return failed;
}
