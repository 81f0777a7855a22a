// r4x16_api.hip — host side of librans4x16_hip.so (include/rans4x16_hip.h).
//
// Host code is orchestration only: argument checks, workspace, staging copies, kernel launches.
// Every byte of codec work (histograms, tables, rANS, transforms) runs in the HIP kernels of
// r4x16_encode.hip / r4x16_decode.hip; there is no CPU code path to fall back to.
#include "r4x16_host.h"

extern "C" const char *rans4x16_hip_version(void) { return "rans4x16_hip 0.2 (gfx950)"; }

// ---------------------------------------------------------------------------------------------
// Options (include/rans4x16_hip.h part 2b).  {name, environment variable that provides the default, built-in default}.
// ---------------------------------------------------------------------------------------------
static const struct { const char *name, *env; long dflt; } OPT_TAB[OPT_COUNT] = {
    /* OPT_DEC_DIRECT        */ {"dec_direct", "R4X16_DEC_DIRECT", 1},
    /* OPT_ENC_DIRECT        */ {"enc_direct", "R4X16_ENC_DIRECT", 1},
    /* OPT_BACK_WG_PER_CU    */ {"back_wg_per_cu", "R4X16_BACK_WG_PER_CU", 0},
    /* OPT_DEC_MID           */ {"dec_mid", "R4X16_DEC_MID", 0},
    /* OPT_DEC_SHORT_RING    */ {"dec_short_ring", "R4X16_DEC_SHORT_RING", 0},
    /* OPT_SCHED_SORT        */ {"sched_sort", "R4X16_SCHED_SORT", 1},
    /* OPT_SCHED_CLAIM       */ {"sched_claim", "R4X16_SCHED_CLAIM", 1},
    /* OPT_SCHED_CONCURRENT  */ {"sched_concurrent", "R4X16_SCHED_CONCURRENT", 1},
    /* OPT_SCHED_TRACE       */ {"sched_trace", "R4X16_SCHED_TRACE", 0},
    /* OPT_SCHED_LEARN       */ {"sched_learn", "R4X16_SCHED_LEARN", 2},
    /* OPT_MAX_WS_MB         */ {"max_workspace_mb", "R4X16_MAX_WS_MB", 160 << 10},
    /* OPT_HOST_STRIPE_DEV   */ {"host_stripe_dev", "R4X16_HOST_STRIPE_DEV", 1},
    /* OPT_HOST_PIPE_MB      */ {"host_pipe_mb", "R4X16_HOST_PIPE_MB", 64},
    /* OPT_HOST_THREADS      */ {"host_threads", "R4X16_HOST_THREADS", 8},
    /* OPT_HOST_LANES        */ {"host_lanes", "R4X16_HOST_LANES", 2},
    /* OPT_HOST_SLAB_MIN_MB  */ {"host_slab_min_mb", "R4X16_HOST_SLAB_MIN_MB", 32},
    /* OPT_HOST_DEC_SLABS    */ {"host_dec_slabs", "R4X16_HOST_DEC_SLABS", 0},
    /* OPT_HOST_ENC_SLABS    */ {"host_enc_slabs", "R4X16_HOST_ENC_SLABS", 0},
    /* OPT_HOST_PACK         */ {"host_pack", "R4X16_HOST_PACK", 1},
    /* OPT_HOST_TRACE        */ {"host_trace", "R4X16_HOST_TRACE", 0},
    /* OPT_DEC_QPW           */ {"dec_qpw", "R4X16_DEC_QPW", 0},
    /* OPT_DEC_QPW_SMALL     */ {"dec_qpw_small", "R4X16_DEC_QPW_SMALL", 0},
    /* OPT_DEC_QPW_PK        */ {"dec_qpw_pk", "R4X16_DEC_QPW_PK", 0},
    /* OPT_DEC_QPW_DIR       */ {"dec_qpw_dir", "R4X16_DEC_QPW_DIR", 0},
    /* OPT_ENC_QPW           */ {"enc_qpw", "R4X16_ENC_QPW", 0},
    /* OPT_ENC_WAVES         */ {"enc_waves", "R4X16_ENC_WAVES", 0},
    /* OPT_ENC_QPW_REC       */ {"enc_qpw_rec", "R4X16_ENC_QPW_REC", 0},
    /* OPT_ENC_QPW_CAP       */ {"enc_qpw_cap", "R4X16_ENC_QPW_CAP", 64},
    /* OPT_FRONT_LDS         */ {"front_lds", "R4X16_FRONT_LDS", 0},
    // process-wide (set with ctx == NULL before the first single-block call / multi-device call)
    /* OPT_COMBINE           */ {"combine", "R4X16_COMBINE", 1},
    /* OPT_COMBINE_WINDOW_US */ {"combine_window_us", "R4X16_COMBINE_WINDOW_US", -1},
    /* OPT_COMBINE_MAX       */ {"combine_max", "R4X16_COMBINE_MAX", 256},
    /* OPT_COMBINE_WORKERS   */ {"combine_workers", "R4X16_COMBINE_WORKERS", 1},
    /* OPT_COMBINE_MAX_MB    */ {"combine_max_mb", "R4X16_COMBINE_MAX_MB", 2048},
    /* OPT_NUMA              */ {"numa", "R4X16_NUMA", 1},
};
// the process-wide defaults: the environment is read here, once, and nowhere else
static std::mutex g_opts_mu;
static R4Opts *opts_defaults_rw()
{
    static R4Opts d;
    static std::once_flag once;
    std::call_once(once, [] {
        for (int i = 0; i < OPT_COUNT; i++) {
            const char *e = getenv(OPT_TAB[i].env);
            d.v[i] = e && *e ? atol(e) : OPT_TAB[i].dflt;
        }
    });
    return &d;
}
extern "C" const R4Opts *r4x16_opts_defaults(void) { return opts_defaults_rw(); }
static int opt_index(const char *name)
{
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; i++) if (!strcmp(name, OPT_TAB[i].name)) return i;
    return -1;
}
extern "C" int rans4x16_hip_set_option(rans4x16_hip_ctx *c, const char *name, long value)
{
    const int i = opt_index(name);
    if (i < 0) return -1;
    if (!c) {                                                    // the defaults of contexts created from now on, the contexts
        R4Opts *d = opts_defaults_rw();                           // behind the five drop-in symbols, and the process-wide options
        std::lock_guard<std::mutex> g(g_opts_mu);
        d->v[i] = value;
        return 0;
    }
    c->opts.v[i] = value;
    if (i == OPT_MAX_WS_MB) c->max_ws = value > 0 ? (size_t)value << 20 : (size_t)160 << 30;
    return 0;
}
// a consistent copy of the process-wide defaults (contexts at creation; the drop-in symbols' contexts at every call)
static void opts_snapshot(rans4x16_hip_ctx *c)
{
    const R4Opts *d = r4x16_opts_defaults();
    std::lock_guard<std::mutex> g(g_opts_mu);
    c->opts = *d;
    c->max_ws = d->v[OPT_MAX_WS_MB] > 0 ? (size_t)d->v[OPT_MAX_WS_MB] << 20 : (size_t)160 << 30;
}
extern "C" int rans4x16_hip_get_option(const rans4x16_hip_ctx *c, const char *name, long *value)
{
    const int i = opt_index(name);
    if (i < 0 || !value) return -1;
    if (c) { *value = c->opts.v[i]; return 0; }
    const R4Opts *d = r4x16_opts_defaults();
    std::lock_guard<std::mutex> g(g_opts_mu);
    *value = d->v[i];
    return 0;
}
extern "C" const char *rans4x16_hip_option_name(int index) { return index >= 0 && index < OPT_COUNT ? OPT_TAB[index].name : nullptr; }

extern "C" rans4x16_hip_ctx *rans4x16_hip_create(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        static std::once_flag once;
        std::call_once(once, [] { fprintf(stderr, "rans4x16_hip: no HIP device available; this library has no CPU path\n"); });
        return nullptr;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return nullptr; }
    if (device >= ndev) return nullptr;
    rans4x16_hip_ctx *c = new rans4x16_hip_ctx();
    c->device = device;
    opts_snapshot(c);
    if (hipSetDevice(device) != hipSuccess) { delete c; return nullptr; }
    // log(1024+k), log(4096+k) from the host libm: the same values the reference's compute_shift
    // obtains at rANS_static4x16pr.c:651-652 on this machine.
    double tab[2 * 257];
    for (int k = 0; k <= 256; k++) { tab[k] = log((double)(1024 + k)); tab[257 + k] = log((double)(4096 + k)); }
    if (hipMalloc((void **)&c->logtab, sizeof(tab)) != hipSuccess ||
        hipMemcpy(c->logtab, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "rans4x16_hip: cannot allocate on device %d\n", device);
        delete c;
        return nullptr;
    }
    // reciprocal of every frequency 1..4096 (rANS_word.h:220-259); index 0 is never used
    std::vector<u32> rcp(RCPTAB_ENTRIES, 0u);
    rcp[1] = ~0u;
    for (u32 f = 2; f < RCPTAB_ENTRIES; f++) {
        u32 shift = 0;
        while (f > (1u << shift)) shift++;
        rcp[f] = (u32)(((1ull << (shift + 31)) + f - 1) / f);
    }
    if (hipMalloc((void **)&c->rcptab, RCPTAB_ENTRIES * 4) != hipSuccess ||
        hipMemcpy(c->rcptab, rcp.data(), RCPTAB_ENTRIES * 4, hipMemcpyHostToDevice) != hipSuccess) {
        rans4x16_hip_destroy(c);
        return nullptr;
    }
    // (a stream alone codes 4 bytes per step: ~99 ns to encode, ~188 ns to decode - bytes per 10 ns tick)
    c->hint[0].pace = 0.40f; c->hint[1].pace = 0.21f;
    for (int w = 0; w < 2; w++)
        if (hipHostMalloc((void **)&c->hint[w].work, SCHED_HINT_BYTES, hipHostMallocDefault) == hipSuccess) memset(c->hint[w].work, 0, SCHED_HINT_BYTES);
        else c->hint[w].work = nullptr;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->stream = nullptr;
        rans4x16_hip_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void rans4x16_hip_destroy(rans4x16_hip_ctx *c)
{
    if (!c) return;
    r4x16_pipe_destroy(c->pipe);
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (int w = 0; w < 2; w++)
        for (auto &t : c->timed[w]) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    if (c->fork_made) {
        for (int i = 0; i < R4_FORK_STREAMS; i++) if (c->fork.aux[i]) { (void)hipStreamSynchronize(c->fork.aux[i]); (void)hipStreamDestroy(c->fork.aux[i]); }
        for (int i = 0; i <= R4_FORK_STREAMS; i++) if (c->fork.ev[i]) (void)hipEventDestroy(c->fork.ev[i]);
    }
    for (int w = 0; w < 2; w++) if (c->hint[w].work) (void)hipHostFree(c->hint[w].work);
    if (c->ws_done) (void)hipEventDestroy(c->ws_done);
    if (c->ws) (void)hipFree(c->ws);
    if (c->xs) (void)hipFree(c->xs);
    if (c->stage) (void)hipFree(c->stage);
    if (c->logtab) (void)hipFree(c->logtab);
    if (c->rcptab) (void)hipFree(c->rcptab);
    delete c;
}

extern "C" int rans4x16_hip_set_dev_stripe_planes(rans4x16_hip_ctx *c, int planes, unsigned int max_block_size)
{
    if (!c || planes < 0 || planes > 255) return -1;
    c->dev_stripe_planes = planes;
    c->dev_stripe_out = max_block_size;
    return 0;
}

extern "C" const char *rans4x16_hip_last_error(const rans4x16_hip_ctx *c) { return c ? c->err.c_str() : "no context"; }
extern "C" size_t rans4x16_hip_workspace_bytes(const rans4x16_hip_ctx *c) { return c ? c->ws_bytes : 0; }

extern "C" void rans4x16_hip_timing(rans4x16_hip_ctx *c, int enable) { if (c) c->timing = enable; }

extern "C" int rans4x16_hip_timing_read(rans4x16_hip_ctx *c, int which, double *ms_total, int *launches, int reset)
{
    if (!c || which < 0 || which > 1) return -1;
    double tot = 0;
    int n = 0;
    for (auto &t : c->timed[which]) {
        float ms = 0;
        if (hipEventSynchronize(t.b) != hipSuccess) return -1;
        if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) return -1;
        tot += ms;
        n++;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = n;
    if (reset) {
        for (auto &t : c->timed[which]) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
        c->timed[which].clear();
    }
    return 0;
}

static int ensure_ws(rans4x16_hip_ctx *c, size_t bytes)
{
    if (bytes <= c->ws_bytes) return 0;
    if (c->ws) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
    const hipError_t e = hipMalloc((void **)&c->ws, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();                   // not sticky: the caller retries with a smaller chunk
        c->ws = nullptr;
        c->err = std::string("hipMalloc of the workspace (") + std::to_string(bytes >> 20) + " MiB): " + hipGetErrorString(e);
        return -1;
    }
    c->ws_bytes = bytes;
    return 0;
}

// Blocks per workspace chunk: as many as the cap allows (the context's ceiling, and three quarters of what the device has
// free right now - other contexts and other processes share the card), in equal chunks rather than full ones and a rest.
// bytes(nb) = workspace of a chunk of nb blocks, monotone in nb.
template <class F>
static size_t plan_chunk(rans4x16_hip_ctx *c, size_t n, F bytes)
{
    size_t cap = c->max_ws;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t room = (free_b + c->ws_bytes) / 4 * 3;
        if (room < cap) cap = room;
    }
    if (bytes(n) <= cap) return n;
    size_t lo = 1, hi = n;                         // the largest chunk that fits: bytes(lo) <= cap < bytes(hi)
    while (lo + 1 < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (bytes(mid) <= cap) lo = mid; else hi = mid;
    }
    const size_t rounds = (n + lo - 1) / lo;
    return (n + rounds - 1) / rounds;
}

// The side streams of the chain kernels' class launches (R4Fork, r4x16_dev.h); nullptr where they are switched off
// (option sched_concurrent = 0), cannot be made, or the context is a lane of the host pipeline.
extern "C" int r4x16_cu_count(void);
#define FORK_ONE_BLOCK_BYTES (256u << 10)
static const R4Fork *fork_for(rans4x16_hip_ctx *c)
{
    if (!c->opts.v[OPT_SCHED_CONCURRENT] || c->no_fork) return nullptr;
    if (!c->fork_made) {
        c->fork_made = true;
        int lo = 0, hi = 0, n = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo: least priority (largest number), hi: greatest
        const int prio[R4_FORK_STREAMS] = {hi, lo, hi, lo, (lo + hi) / 2};
        for (; n < R4_FORK_STREAMS; n++)
            if (hipStreamCreateWithPriority(&c->fork.aux[n], hipStreamNonBlocking, prio[n]) != hipSuccess) { c->fork.aux[n] = nullptr; break; }
        for (int i = 0; i <= n; i++)
            if (hipEventCreateWithFlags(&c->fork.ev[i], hipEventDisableTiming) != hipSuccess) { c->fork.ev[i] = nullptr; n = 0; break; }
        c->fork.n = n;
    }
    return c->fork.n ? &c->fork : nullptr;
}

static int ws_order_begin(rans4x16_hip_ctx *c, hipStream_t s);
static int ws_order_end(rans4x16_hip_ctx *c, hipStream_t s);
// A context has ONE workspace: calls on different streams must not overlap on it.  Every *_dev call ends with an
// event on its stream; a call on another stream first waits for the previous call's event.
int r4x16_ws_order_begin(rans4x16_hip_ctx *c, hipStream_t s) { return ws_order_begin(c, s); }
int r4x16_ws_order_end(rans4x16_hip_ctx *c, hipStream_t s) { return ws_order_end(c, s); }
static int ws_order_begin(rans4x16_hip_ctx *c, hipStream_t s)
{
    if (c->ws_busy && s != c->ws_stream) HIPCHK(c, hipStreamWaitEvent(s, c->ws_done, 0));
    return 0;
}
static int ws_order_end(rans4x16_hip_ctx *c, hipStream_t s)
{
    if (!c->ws_done) HIPCHK(c, hipEventCreateWithFlags(&c->ws_done, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ws_done, s));
    c->ws_stream = s;
    c->ws_busy = true;
    return 0;
}

// carve helper
struct Carver {
    u8 *p; size_t off = 0;
    explicit Carver(u8 *base) : p(base) {}
    template <class T> T *take(size_t count, size_t elem = sizeof(T)) {
        off = align_up(off, 256);
        T *r = (T *)(p ? p + off : nullptr);
        off += count * elem;
        return r;
    }
};

static void sched_layout(Carver &cv, size_t nitems, SchedWs *w)
{
    w->key = cv.take<u32>(nitems);
    w->list = cv.take<u32>(nitems);
    w->cnt = cv.take<u32>(SCHED_CNT_WORDS);
    w->bins = cv.take<u32>(2 * SCHED_BINS);
    w->work = cv.take<u64>(2 * CLS_MAX);
}

static void time_begin(rans4x16_hip_ctx *c, int which, hipStream_t s, TimedLaunch &t)
{
    (void)hipEventCreate(&t.a); (void)hipEventCreate(&t.b);
    (void)hipEventRecord(t.a, s);
    (void)which;
}
static void time_end(rans4x16_hip_ctx *c, int which, hipStream_t s, TimedLaunch &t)
{
    (void)hipEventRecord(t.b, s);
    c->timed[which].push_back(t);
}

// ---------------------------------------------------------------------------------------------
// device-resident batches
// ---------------------------------------------------------------------------------------------
// var_bytes: the blocks' staging regions for X_PACK / X_RLE together (0: the batch cannot use the transforms)
static size_t enc_ws_layout(u8 *base, size_t nblk, u64 scratch_stride, u64 var_bytes, EncWs *w)
{
    Carver cv(base);
    w->desc = cv.take<EncDesc>(nblk);
    w->items = cv.take<EncItem>(3 * nblk);         // payload, RLE meta, nested order-1 table
    w->images = cv.take<u8>(nblk, ENC_IMG_BYTES);
    w->tab = cv.take<u8>(nblk, TAB_BYTES);
    w->scratch = cv.take<u8>(nblk, scratch_stride);
    w->scratch_stride = scratch_stride;
    w->var = var_bytes ? cv.take<u8>(var_bytes) : nullptr;
    w->voff = var_bytes ? cv.take<u64>(nblk + 1) : nullptr;
    w->var_bytes = var_bytes;
    w->metatab = cv.take<u8>(nblk, var_bytes ? META_TAB_BYTES : 0);
    w->stat = cv.take<EncStat>(nblk);
    w->dump = cv.take<u8>(1, ENC_DUMP_BYTES);
    sched_layout(cv, 3 * nblk, &w->sched);
    w->direct_budget = 0;
    w->meta_records = 0;
    w->pad = 0;
    return align_up(cv.off, 256);
}

extern "C" int rans4x16_hip_compress_dev(rans4x16_hip_ctx *c, int n,
                                         const unsigned char *d_in, const uint64_t *d_in_off,
                                         const uint32_t *d_in_size,
                                         unsigned char *d_out, const uint64_t *d_out_off,
                                         const uint32_t *d_out_cap, uint32_t *d_out_size,
                                         int32_t *d_status, int order, const int32_t *d_order,
                                         uint32_t max_in_size, void *stream)
{
    return rans4x16_hip_compress_dev_sized(c, n, d_in, d_in_off, d_in_size, d_out, d_out_off, d_out_cap, d_out_size, d_status,
                                           order, d_order, max_in_size, 0, stream);
}

extern "C" int rans4x16_hip_compress_dev_sized(rans4x16_hip_ctx *c, int n,
                                               const unsigned char *d_in, const uint64_t *d_in_off,
                                               const uint32_t *d_in_size,
                                               unsigned char *d_out, const uint64_t *d_out_off,
                                               const uint32_t *d_out_cap, uint32_t *d_out_size,
                                               int32_t *d_status, int order, const int32_t *d_order,
                                               uint32_t max_in_size, uint64_t total_in_size, void *stream)
{
    if (!c) return -1;
    if (n < 0 || (n && (!d_in || !d_in_off || !d_in_size || !d_out || !d_out_off || !d_out_cap || !d_out_size || !d_status))) {
        c->err = "compress_dev: bad arguments";
        return -1;
    }
    if (n == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (!d_order && (order & X_STRIPE)) {                 // one order for all blocks: N and the candidate methods are host knowledge
        BatchArgs sa;
        sa.in = d_in; sa.in_off = d_in_off; sa.in_size = d_in_size;
        sa.out = d_out; sa.out_off = d_out_off; sa.out_cap = d_out_cap; sa.out_size = d_out_size;
        sa.status = d_status; sa.d_order = nullptr; sa.order = order; sa.n = n;
        return r4x16_stripe_compress_dev(c, n, sa, order, max_in_size, s);
    }

    // Per block the workspace holds fixed-size records, tables and images, 256 KB for the pair counters and the nested
    // table stream, and - only for blocks that ask for X_PACK / X_RLE - a staging region sized from the block's own length
    // and laid out on the device (enc_var_layout).  The payload itself is written into the caller's slot (k_enc_tables).
    // The host only needs a bound for the staging regions together: from total_in_size when the caller gave it, from
    // n x max_in_size otherwise (per-block orders: every block may ask).
    const u64 scratch_stride = ENC_F_BYTES;
    const bool xf = d_order != nullptr || (order & (X_PACK | X_RLE));
    const u64 total_in = total_in_size ? total_in_size : (u64)n * max_in_size;
    auto var_for = [&](size_t nb) -> u64 { return xf ? enc_var_bound(nb, std::min<u64>(total_in, (u64)nb * max_in_size)) : 0; };
    EncWs w;
    size_t chunk = plan_chunk(c, (size_t)n, [&](size_t nb) { return enc_ws_layout(nullptr, nb, scratch_stride, var_for(nb), &w) + 4096; });
    if (ws_order_begin(c, s) != 0) return -1;
    for (;;) {                                     // out of memory: walk the batch in smaller chunks
        const size_t need = enc_ws_layout(nullptr, chunk, scratch_stride, var_for(chunk), &w);
        if (ensure_ws(c, need) == 0) break;
        if (chunk == 1) return -1;
        chunk = (chunk + 1) / 2;
    }
    enc_ws_layout(c->ws, chunk, scratch_stride, var_for(chunk), &w);
    w.logtab = c->logtab;
    w.rcptab = c->rcptab;

    BatchArgs a;
    a.in = d_in; a.in_off = d_in_off; a.in_size = d_in_size;
    a.out = d_out; a.out_off = d_out_off; a.out_cap = d_out_cap; a.out_size = d_out_size;
    a.status = d_status; a.d_order = d_order; a.order = order; a.n = n;

    for (size_t base = 0; base < (size_t)n; base += chunk) {
        const int nb = (int)((size_t)n - base < chunk ? (size_t)n - base : chunk);
        w.direct_budget = r4x16_enc_direct_budget(nb, &c->opts);       // few streams: LDS to spare, symbol records (r4x16_common.h)
        // (made at first use: not inside the timed region.  One small block is one payload class and a call of 0.4 ms: it
        //  stays on the caller's stream; one large block with X_RLE is two long streams - literals, run lengths - of
        //  different classes: 15.7 -> 12.0 ms per MiB of q8)
        const R4Fork *fk = (nb > 1 || max_in_size >= FORK_ONE_BLOCK_BYTES) ? fork_for(c) : nullptr;
        w.meta_records = fk == nullptr;
        r4x16_launch_enc_front(&a, &w, (int)base, nb, s, &c->opts);
        r4x16_launch_enc_tables(&a, &w, (int)base, nb, s);
        TimedLaunch t;
        if (c->timing) time_begin(c, 0, s, t);
        r4x16_launch_enc_chain(&w, 3 * nb, s, fk, &c->opts, &c->hint[0]);
        if (c->timing) time_end(c, 0, s, t);
        r4x16_launch_enc_finish(&a, &w, (int)base, nb, s);
    }
    HIPCHK(c, hipGetLastError());
    return ws_order_end(c, s);
}

// var_bytes: the staging regions of the blocks that carry X_PACK / X_RLE, together (0: the batch has none)
static size_t dec_ws_layout(u8 *base, size_t nblk, u64 var_bytes, u32 max_out_cap, DecWs *w)
{
    Carver cv(base);
    w->desc = cv.take<DecDesc>(nblk);
    w->items = cv.take<DecItem>(3 * nblk);
    w->resume = cv.take<DecResume>(nblk);
    w->images = cv.take<u8>(nblk, (size_t)DEC_IMG_SLOT);
    w->tbuf = cv.take<u8>(nblk, TBUF_BYTES);
    w->var = var_bytes ? cv.take<u8>(var_bytes) : nullptr;
    w->voff = var_bytes ? cv.take<u64>(nblk + 1) : nullptr;
    w->var_bytes = var_bytes;
    w->max_out_cap = max_out_cap;
    w->pad2 = 0;
    sched_layout(cv, 2 * nblk, &w->sched);
    w->direct_budget = 0;
    w->mid_budget = 0;
    return align_up(cv.off, 256);
}

extern "C" int rans4x16_hip_uncompress_dev(rans4x16_hip_ctx *c, int n,
                                           const unsigned char *d_in, const uint64_t *d_in_off,
                                           const uint32_t *d_in_size,
                                           unsigned char *d_out, const uint64_t *d_out_off,
                                           const uint32_t *d_out_cap, uint32_t *d_out_size,
                                           int32_t *d_status, uint32_t max_in_size, uint32_t max_out_cap,
                                           void *stream)
{
    return rans4x16_hip_uncompress_dev_sized(c, n, d_in, d_in_off, d_in_size, d_out, d_out_off, d_out_cap, d_out_size, d_status,
                                             max_in_size, max_out_cap, 0, stream);
}

extern "C" int rans4x16_hip_uncompress_dev_sized(rans4x16_hip_ctx *c, int n,
                                                 const unsigned char *d_in, const uint64_t *d_in_off,
                                                 const uint32_t *d_in_size,
                                                 unsigned char *d_out, const uint64_t *d_out_off,
                                                 const uint32_t *d_out_cap, uint32_t *d_out_size,
                                                 int32_t *d_status, uint32_t max_in_size, uint32_t max_out_cap,
                                                 uint64_t total_out_cap, void *stream)
{
    if (!c) return -1;
    if (n < 0 || (n && (!d_in || !d_in_off || !d_in_size || !d_out || !d_out_off || !d_out_cap || !d_out_size || !d_status))) {
        c->err = "uncompress_dev: bad arguments";
        return -1;
    }
    if (n == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (c->dev_stripe_planes > 0 && !c->in_stripe) {
        BatchArgs sa;
        sa.in = d_in; sa.in_off = d_in_off; sa.in_size = d_in_size;
        sa.out = d_out; sa.out_off = d_out_off; sa.out_cap = d_out_cap; sa.out_size = d_out_size;
        sa.status = d_status; sa.d_order = nullptr; sa.order = 0; sa.n = n;
        return r4x16_stripe_uncompress_dev(c, n, sa, max_in_size, max_out_cap, c->dev_stripe_out, s);
    }

    // PACK / RLE staging, only for the blocks whose flag byte asks for it: a buffer of the block's output size and room
    // for the decoded run-length meta (rANS_static4x16pr.c:1273: at most in_size + 257 bytes), laid out on the device
    // (dec_var_bytes, k_dec_voff).  The host bounds their sum: from total_out_cap when the caller gave it, from
    // n x max_out_cap otherwise; max_out_cap == 0 says the batch has no such block.
    const u64 total_out = total_out_cap ? total_out_cap : (u64)n * max_out_cap;
    auto var_for = [&](size_t nb) -> u64 { return max_out_cap ? dec_var_bound(nb, std::min<u64>(total_out, (u64)nb * max_out_cap)) : 0; };
    DecWs w;
    size_t chunk = plan_chunk(c, (size_t)n, [&](size_t nb) { return dec_ws_layout(nullptr, nb, var_for(nb), max_out_cap, &w) + 4096; });
    if (ws_order_begin(c, s) != 0) return -1;
    for (;;) {                                     // out of memory: walk the batch in smaller chunks
        const size_t need = dec_ws_layout(nullptr, chunk, var_for(chunk), max_out_cap, &w);
        if (ensure_ws(c, need) == 0) break;
        if (chunk == 1) return -1;
        chunk = (chunk + 1) / 2;
    }
    dec_ws_layout(c->ws, chunk, var_for(chunk), max_out_cap, &w);

    BatchArgs a;
    a.in = d_in; a.in_off = d_in_off; a.in_size = d_in_size;
    a.out = d_out; a.out_off = d_out_off; a.out_cap = d_out_cap; a.out_size = d_out_size;
    a.status = d_status; a.d_order = nullptr; a.order = 0; a.n = n;

    for (size_t base = 0; base < (size_t)n; base += chunk) {
        const int nb = (int)((size_t)n - base < chunk ? (size_t)n - base : chunk);
        w.direct_budget = r4x16_dec_direct_budget(nb, &c->opts);       // few streams: LDS to spare, the short-step rows (r4x16_common.h)
        w.mid_budget = r4x16_dec_mid_budget(nb, &c->opts);             // one partly filled round: the mid rows
        r4x16_launch_dec_front(&a, &w, (int)base, nb, s, &c->opts);
        const R4Fork *fk = (nb > 1 || max_out_cap >= FORK_ONE_BLOCK_BYTES) ? fork_for(c) : nullptr;   // (one large block: 28.2 -> 19.8 ms per MiB of q8 with X_RLE)
        TimedLaunch t;
        if (c->timing) time_begin(c, 1, s, t);
        r4x16_launch_dec_chain(&w, 2 * nb, s, fk, &c->opts, &c->hint[1]);
        if (c->timing) time_end(c, 1, s, t);
        r4x16_launch_dec_back(&a, &w, (int)base, nb, s, &c->opts);
    }
    HIPCHK(c, hipGetLastError());
    return ws_order_end(c, s);
}

extern "C" int r4x16_dec_residency(u32 nsym, int order, u32 bits, int *streams_per_wave, int *waves_per_cu, int short_ring);
extern "C" int r4x16_enc_residency(u32 nsym, int order, int *streams_per_wave, int *waves_per_cu);
extern "C" int r4x16_cu_count(void);

extern "C" int rans4x16_hip_residency(rans4x16_hip_ctx *c, int decode, unsigned int nsym, int order, unsigned int shift,
                                      int *streams_per_cu, int *lanes_live_per_wave, int *compute_units)
{
    if (!c) return -1;
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    int spw = 0, wpc = 0, total = 0;
    if (decode) {
        if (r4x16_dec_residency(nsym, order & 1, shift, &spw, &wpc, c->opts.v[OPT_DEC_SHORT_RING] != 0) != 0) return -1;
        total = spw * wpc;
    } else {
        total = r4x16_enc_residency(nsym, order & 1, &spw, &wpc);
        if (total < 0) return -1;
    }
    if (streams_per_cu) *streams_per_cu = total;
    if (lanes_live_per_wave) *lanes_live_per_wave = 4 * spw;
    if (compute_units) *compute_units = r4x16_cu_count();
    return 0;
}

extern "C" int rans4x16_hip_device_clock_khz(rans4x16_hip_ctx *c)
{
    int khz = 0;
    if (!c || hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, c->device) != hipSuccess) return -1;
    return khz;
}

extern "C" int rans4x16_hip_compress_batch(rans4x16_hip_ctx *c, int n,
                                           const unsigned char *const *in, const unsigned int *in_size,
                                           unsigned char *const *out, unsigned int *out_size,
                                           const int *order, int *status)
{
    return r4x16_run_host_batch(c, n, false, in, in_size, out, out_size, order, status);
}

extern "C" int rans4x16_hip_uncompress_batch(rans4x16_hip_ctx *c, int n,
                                             const unsigned char *const *in, const unsigned int *in_size,
                                             unsigned char *const *out, unsigned int *out_size, int *status)
{
    return r4x16_run_host_batch(c, n, true, in, in_size, out, out_size, nullptr, status);
}

// ---------------------------------------------------------------------------------------------
// The five htscodecs entry points (htscodecs/rANS_static4x16.h:41-50).
// ---------------------------------------------------------------------------------------------
// device memory a thread's context keeps between single-block calls; a call that needed more gives it back
#define SINGLE_CALL_KEEP ((size_t)1 << 30)

static rans4x16_hip_ctx *thread_ctx()
{
    // one context per host thread: the reference API is re-entrant from thread pools
    struct Holder { rans4x16_hip_ctx *c = nullptr; ~Holder() { rans4x16_hip_destroy(c); } };
    static thread_local Holder h;
    if (!h.c) h.c = rans4x16_hip_create(-1);
    if (h.c) opts_snapshot(h.c);                   // no context argument: these calls follow the process-wide options
    return h.c;
}

// ---------------------------------------------------------------------------------------------
// The combiner behind the single-block entry points.  A CRAM reader / writer calls rans_compress_to_4x16 or
// rans_uncompress_to_4x16 once per block from a pool of host threads.  One block is four chains: a GPU
// call for it costs its chain latency whatever else the card does, and thirty-two threads each driving their own
// copies and launches mostly wait for the runtime's locks (measured: 278 ms per call).  So calls that arrive together
// are served together: a caller queues its block and sleeps; ONE worker thread per direction takes what is queued
// (after a short, adaptive gathering window, see `window_us` below), runs one host batch on its own context and wakes
// each caller with its own result.  Callers stay independent: a batch that fails as a whole (staging refused, out of
// memory, a runtime error) is re-run block by block, so one hostile or oversized request cannot fail its neighbours;
// a batch gathers at most R4X16_COMBINE_MAX_MB of buffers; a worker that cannot start marks its direction dead and
// callers fall back to their own per-thread context.
// Process-wide options (rans4x16_hip_set_option(NULL, ..) before the first single-block call; defaults from R4X16_COMBINE*):
//   combine = 0               calls go straight to a per-thread context
//   combine_window_us         a fixed gathering window instead of the adaptive one
//   combine_max               blocks per batch (default 256)      combine_max_mb  buffer bytes per batch (default 2048)
// ---------------------------------------------------------------------------------------------
#include <condition_variable>
#include <deque>
#include <memory>
struct CombReq {
    const unsigned char *in; unsigned int in_size;
    unsigned char *out; unsigned int cap;
    int order;
    unsigned int result = 0;
    int rc = -1;                               // 0 ok, 1 block failed, -1 batch failed
    bool done = false;
};
struct Combiner {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<CombReq *> q[2];                // [0] encode, [1] decode
    int device = 0;
    bool started = false, stop = false;
    // One worker per direction: requests that arrive while a batch runs queue up and form the next batch, and the
    // callers of the batch that has just finished are given a moment to come back and join it - a twentieth of what
    // that batch took (20 us .. 2 ms; 200 us before the first), so that a pool calling in lock-step ends up in ONE batch
    // per round instead of two groups that alternate, each waiting for the other's (two workers with a fixed 200 us
    // window: 32 threads x 1 MiB decoded at 205-410 MB/s depending on how the groups fell; this way: see INTEGRATION 1).
    long window_us = -1, max_batch = 256, workers = 1;
    size_t max_bytes = (size_t)2048 << 20;     // input + output capacity gathered into one batch
    long last_us[2] = {4000, 4000};
    bool dead[2] = {false, false};             // the direction's worker could not start: callers use their own context

    static void finish(CombReq *r, int rc, unsigned int result) { r->rc = rc; r->result = result; r->done = true; }

    void worker(int dir)
    {
        rans4x16_hip_ctx *c = hipSetDevice(device) == hipSuccess ? rans4x16_hip_create(device) : nullptr;
        std::unique_lock<std::mutex> lk(mu);
        if (!c) {
            // nobody may wait for a worker that does not exist: hand back what is queued (rc -2 = "use your own
            // context") and refuse later submissions the same way
            dead[dir] = true;
            for (CombReq *r : q[dir]) finish(r, -2, 0u);
            q[dir].clear();
            cv_done.notify_all();
            return;
        }
        for (;;) {
            cv_work.wait(lk, [&] { return stop || !q[dir].empty(); });
            if (stop) break;
            const long win = window_us >= 0 ? window_us : std::min(2000L, std::max(20L, last_us[dir] / 20));
            if ((long)q[dir].size() < max_batch && win > 0) {
                // more callers may be a few microseconds behind: give them the window
                cv_work.wait_for(lk, std::chrono::microseconds(win), [&] { return stop || (long)q[dir].size() >= max_batch; });
                if (stop) break;
            }
            std::vector<CombReq *> batch;
            size_t bytes = 0;
            while (!q[dir].empty() && (long)batch.size() < max_batch) {
                CombReq *r = q[dir].front();
                const size_t need = (size_t)r->in_size + (size_t)r->cap;
                if (!batch.empty() && bytes + need > max_bytes) break;          // the rest forms the next batch
                bytes += need;
                batch.push_back(r);
                q[dir].pop_front();
            }
            lk.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            const int n = (int)batch.size();
            std::vector<const unsigned char *> in(n);
            std::vector<unsigned char *> out(n);
            std::vector<unsigned int> isz(n), osz(n);
            std::vector<int> ord(n), st(n, 0), rcs(n, 0);
            for (int i = 0; i < n; i++) { in[i] = batch[i]->in; out[i] = batch[i]->out; isz[i] = batch[i]->in_size; osz[i] = batch[i]->cap; ord[i] = batch[i]->order; }
            opts_snapshot(c);                  // the drop-in symbols have no context argument: they follow the process-wide options
            const int rc = r4x16_run_host_batch(c, n, dir == 1, in.data(), isz.data(), out.data(), osz.data(), dir == 0 ? ord.data() : nullptr, st.data());
            if (rc < 0 && n > 1) {
                // The batch failed as a whole (staging refused for one hostile size field, out of memory, a runtime
                // error): the callers are unrelated, so each block gets its own one-block batch and its own verdict.
                for (int i = 0; i < n; i++) {
                    osz[i] = batch[i]->cap;
                    st[i] = 0;
                    rcs[i] = r4x16_run_host_batch(c, 1, dir == 1, &in[i], &isz[i], &out[i], &osz[i], dir == 0 ? &ord[i] : nullptr, &st[i]);
                }
            } else
                for (int i = 0; i < n; i++) rcs[i] = rc;
            r4x16_trim(c, (size_t)4 << 30);
            const long took = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            lk.lock();
            last_us[dir] = took;
            for (int i = 0; i < n; i++) finish(batch[i], rcs[i] < 0 ? -1 : (st[i] != 0 ? 1 : 0), rcs[i] < 0 ? 0u : osz[i]);
            cv_done.notify_all();
        }
        lk.unlock();
        rans4x16_hip_destroy(c);
    }
    // 0 ok, 1 block failed, -1 call failed, -2 no worker for this direction (the caller uses its own context)
    int submit(CombReq &r, int dir)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (!started) {
            started = true;
            const R4Opts *o = r4x16_opts_defaults();
            if (o->v[OPT_COMBINE_WINDOW_US] >= 0) window_us = o->v[OPT_COMBINE_WINDOW_US];     // a fixed window instead of the adaptive one
            if (o->v[OPT_COMBINE_MAX] > 0) max_batch = o->v[OPT_COMBINE_MAX];
            if (o->v[OPT_COMBINE_WORKERS] > 0 && o->v[OPT_COMBINE_WORKERS] <= 4) workers = o->v[OPT_COMBINE_WORKERS];
            if (o->v[OPT_COMBINE_MAX_MB] > 0) max_bytes = (size_t)o->v[OPT_COMBINE_MAX_MB] << 20;
            // (detached, and the combiner itself is never destroyed: tearing GPU contexts down from static destructors
            //  at process exit races the runtime's own shutdown.  After fork() the child has no workers and no usable
            //  runtime either - HIP does not survive a fork - so nothing is done about that case.)
            for (int d = 0; d < 2; d++) for (long j = 0; j < workers; j++) std::thread([this, d] { worker(d); }).detach();
        }
        if (dead[dir]) return -2;
        q[dir].push_back(&r);
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return r.done; });
        return r.rc;
    }
};
// one combiner per device (a caller's current device decides, as it did for its per-thread context)
static Combiner *combiner_for_current_device()
{
    if (!r4x16_opts_defaults()->v[OPT_COMBINE]) return nullptr;
    int ndev = 0, dev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    static std::mutex mu;
    static Combiner *per_dev[64];
    std::lock_guard<std::mutex> g(mu);
    if (!per_dev[dev]) { per_dev[dev] = new Combiner(); per_dev[dev]->device = dev; }
    return per_dev[dev];
}
// one block through the combiner (or, with R4X16_COMBINE=0, through this thread's own context); 0 = ok
static int single_block(bool decode, const unsigned char *in, unsigned int in_size, unsigned char *out, unsigned int *out_size, int order)
{
    if (Combiner *cb = combiner_for_current_device()) {
        CombReq r;
        r.in = in; r.in_size = in_size; r.out = out; r.cap = *out_size; r.order = order;
        const int rc = cb->submit(r, decode ? 1 : 0);
        if (rc != -2) {
            *out_size = r.result;
            return rc;
        }
    }
    rans4x16_hip_ctx *c = thread_ctx();
    if (!c) return -1;
    const unsigned char *ins[1] = { in };
    unsigned char *outs[1] = { out };
    unsigned int isz[1] = { in_size };
    int ord[1] = { order };
    const int rc = r4x16_run_host_batch(c, 1, decode, ins, isz, outs, out_size, decode ? nullptr : ord, nullptr);
    r4x16_trim(c, SINGLE_CALL_KEEP);
    return rc;
}
extern "C" unsigned int rans_compress_bound_4x16(unsigned int size, int order)
{
    return r4x16_compress_bound(size, order);
}

extern "C" unsigned char *rans_compress_to_4x16(unsigned char *in, unsigned int in_size,
                                                unsigned char *out, unsigned int *out_size, int order)
{
    if (!out_size) return nullptr;
    unsigned char *mine = nullptr;
    if (!out) {
        *out_size = rans_compress_bound_4x16(in_size, order);
        if (!(out = mine = (unsigned char *)malloc(*out_size))) return nullptr;
    }
    if (single_block(false, in, in_size, out, out_size, order) != 0) { free(mine); return nullptr; }
    return out;
}

extern "C" unsigned char *rans_compress_4x16(unsigned char *in, unsigned int in_size,
                                             unsigned int *out_size, int order)
{
    return rans_compress_to_4x16(in, in_size, nullptr, out_size, order);
}

// uncompressed size stored in the container header (host read of <= 6 bytes; needed to size the
// malloc the reference API promises when out == NULL, rANS_static4x16pr.c:1459-1462)
static int peek_size(const unsigned char *in, unsigned int in_size, unsigned int *usz)
{
    if (in_size < 1 || (in[0] & X_NOSZ)) return -1;
    unsigned int v = 0, i = 1;
    unsigned char ch;
    if (i >= in_size) { *usz = 0; return 0; }
    do { ch = in[i++]; v = (v << 7) | (ch & 0x7f); } while ((ch & 0x80) && i < in_size);
    *usz = v;
    return 0;
}

extern "C" unsigned char *rans_uncompress_to_4x16(unsigned char *in, unsigned int in_size,
                                                  unsigned char *out, unsigned int *out_size)
{
    if (!out_size || in_size == 0) return nullptr;
    unsigned char *mine = nullptr;
    if (!out) {
        unsigned int usz;
        if (peek_size(in, in_size, &usz) != 0) return nullptr;       // X_NOSZ needs a caller buffer (:1456)
        if (usz >= INT_MAX) return nullptr;
        if (!(out = mine = (unsigned char *)malloc(usz ? usz : 1))) return nullptr;
        *out_size = usz;
    }
    if (single_block(true, in, in_size, out, out_size, 0) != 0) { free(mine); return nullptr; }
    return out;
}

extern "C" unsigned char *rans_uncompress_4x16(unsigned char *in, unsigned int in_size, unsigned int *out_size)
{
    return rans_uncompress_to_4x16(in, in_size, nullptr, out_size);
}

// =============================================================================================
// rANS 4x8 (CRAM 3.0): include/rans4x8_hip.h.  Same context, same workspace; kernels at the end of
// r4x16_encode.hip / r4x16_decode.hip.
// =============================================================================================
extern "C" size_t r4x8_dec_ws_bytes(size_t nblk);
extern "C" void r4x8_launch_decode(const BatchArgs *, u8 *, int, int, hipStream_t);
extern "C" void r4x8_launch_encode(const BatchArgs *, const EncWs *, int, int, hipStream_t);
extern "C" u32 r4x8_compress_bound(u32);

extern "C" unsigned int rans4x8_hip_compress_bound(unsigned int size) { return r4x8_compress_bound(size); }

extern "C" int rans4x8_hip_compress_dev(rans4x16_hip_ctx *c, int n,
                                        const unsigned char *d_in, const uint64_t *d_in_off, const uint32_t *d_in_size,
                                        unsigned char *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                                        uint32_t *d_out_size, int32_t *d_status, int order, const int32_t *d_order,
                                        uint32_t max_in_size, void *stream)
{
    if (!c) return -1;
    if (n < 0 || (n && (!d_in || !d_in_off || !d_in_size || !d_out || !d_out_off || !d_out_cap || !d_out_size || !d_status))) {
        c->err = "rans4x8 compress_dev: bad arguments";
        return -1;
    }
    if (n == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const u64 scratch_stride = align_up(std::max((size_t)r4x8_compress_bound(max_in_size) + 64, (size_t)ENC_F_BYTES), 256);
    EncWs w;
    size_t chunk = plan_chunk(c, (size_t)n, [&](size_t nb) { return enc_ws_layout(nullptr, nb, scratch_stride, 0, &w) + 4096; });
    if (ws_order_begin(c, s) != 0) return -1;
    for (;;) {
        if (ensure_ws(c, enc_ws_layout(nullptr, chunk, scratch_stride, 0, &w)) == 0) break;
        if (chunk == 1) return -1;
        chunk = (chunk + 1) / 2;
    }
    enc_ws_layout(c->ws, chunk, scratch_stride, 0, &w);
    w.logtab = c->logtab;
    w.rcptab = c->rcptab;
    BatchArgs a;
    a.in = d_in; a.in_off = d_in_off; a.in_size = d_in_size;
    a.out = d_out; a.out_off = d_out_off; a.out_cap = d_out_cap; a.out_size = d_out_size;
    a.status = d_status; a.d_order = d_order; a.order = order; a.n = n;
    for (size_t base = 0; base < (size_t)n; base += chunk) {
        const int nb = (int)((size_t)n - base < chunk ? (size_t)n - base : chunk);
        r4x8_launch_encode(&a, &w, (int)base, nb, s);
    }
    HIPCHK(c, hipGetLastError());
    return ws_order_end(c, s);
}

extern "C" int rans4x8_hip_uncompress_dev(rans4x16_hip_ctx *c, int n,
                                          const unsigned char *d_in, const uint64_t *d_in_off, const uint32_t *d_in_size,
                                          unsigned char *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                                          uint32_t *d_out_size, int32_t *d_status, void *stream)
{
    if (!c) return -1;
    if (n < 0 || (n && (!d_in || !d_in_off || !d_in_size || !d_out || !d_out_off || !d_out_cap || !d_out_size || !d_status))) {
        c->err = "rans4x8 uncompress_dev: bad arguments";
        return -1;
    }
    if (n == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    size_t chunk = plan_chunk(c, (size_t)n, [&](size_t nb) { return r4x8_dec_ws_bytes(nb) + 4096; });
    if (ws_order_begin(c, s) != 0) return -1;
    for (;;) {
        if (ensure_ws(c, r4x8_dec_ws_bytes(chunk)) == 0) break;
        if (chunk == 1) return -1;
        chunk = (chunk + 1) / 2;
    }
    BatchArgs a;
    a.in = d_in; a.in_off = d_in_off; a.in_size = d_in_size;
    a.out = d_out; a.out_off = d_out_off; a.out_cap = d_out_cap; a.out_size = d_out_size;
    a.status = d_status; a.d_order = nullptr; a.order = 0; a.n = n;
    for (size_t base = 0; base < (size_t)n; base += chunk) {
        const int nb = (int)((size_t)n - base < chunk ? (size_t)n - base : chunk);
        r4x8_launch_decode(&a, c->ws, (int)base, nb, s);
    }
    HIPCHK(c, hipGetLastError());
    return ws_order_end(c, s);
}

// host buffers: one staging pass (copy in, kernels, sizes back, copy out) on the context's stream
static int run_host8(rans4x16_hip_ctx *c, int n, bool decode,
                     const unsigned char *const *in, const unsigned int *in_size,
                     unsigned char *const *out, unsigned int *out_size, const int *order, int *status)
{
    if (!c) return -1;
    if (n <= 0) return n == 0 ? 0 : -1;
    if (!in || !in_size || !out || !out_size) { c->err = "rans4x8 batch: bad arguments"; return -1; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    std::vector<u64> in_off(n), out_off(n);
    std::vector<u32> cap(n);
    std::vector<i32> ord(n);
    size_t in_tot = 0, out_tot = 0;
    u32 max_in = 0;
    for (int i = 0; i < n; i++) {
        in_off[i] = in_tot; in_tot += align_up((size_t)in_size[i] + 16, 256);
        cap[i] = out_size[i];
        out_off[i] = out_tot; out_tot += align_up((size_t)cap[i] + 16, 256);
        if (in_size[i] > max_in) max_in = in_size[i];
        ord[i] = order ? order[i] : 0;
    }
    const size_t arr = align_up((size_t)n * 8, 256);
    if (r4x16_ensure_stage(c, in_tot + out_tot + 6 * arr) != 0) return -1;
    u8 *d_in = c->stage, *d_out = d_in + in_tot, *meta = d_out + out_tot;
    u64 *d_in_off = (u64 *)meta, *d_out_off = (u64 *)(meta + arr);
    u32 *d_in_size = (u32 *)(meta + 2 * arr), *d_cap = (u32 *)(meta + 3 * arr), *d_osz = (u32 *)(meta + 4 * arr);
    i32 *d_status = (i32 *)(meta + 5 * arr), *d_order = (i32 *)(meta + 5 * arr + arr / 2);
    for (int i = 0; i < n; i++)
        if (in_size[i]) HIPCHK(c, hipMemcpyAsync(d_in + in_off[i], in[i], in_size[i], hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_in_off, in_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_out_off, out_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_in_size, in_size, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_cap, cap.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_order, ord.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    const int rc = decode ? rans4x8_hip_uncompress_dev(c, n, d_in, d_in_off, d_in_size, d_out, d_out_off, d_cap, d_osz, d_status, s)
                          : rans4x8_hip_compress_dev(c, n, d_in, d_in_off, d_in_size, d_out, d_out_off, d_cap, d_osz, d_status,
                                                     0, d_order, max_in, s);
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

extern "C" int rans4x8_hip_compress_batch(rans4x16_hip_ctx *c, int n, const unsigned char *const *in, const unsigned int *in_size,
                                          unsigned char *const *out, unsigned int *out_size, const int *order, int *status)
{
    return run_host8(c, n, false, in, in_size, out, out_size, order, status);
}
extern "C" int rans4x8_hip_uncompress_batch(rans4x16_hip_ctx *c, int n, const unsigned char *const *in, const unsigned int *in_size,
                                            unsigned char *const *out, unsigned int *out_size, int *status)
{
    return run_host8(c, n, true, in, in_size, out, out_size, nullptr, status);
}

// htscodecs/rANS_static.h:41-44: malloc'd results, NULL on failure
extern "C" unsigned char *rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order)
{
    rans4x16_hip_ctx *c = thread_ctx();
    if (!c || !out_size || !in_size) return nullptr;
    unsigned int cap = r4x8_compress_bound(in_size);
    unsigned char *out = (unsigned char *)malloc(cap);
    if (!out) return nullptr;
    const unsigned char *ins[1] = { in };
    unsigned char *outs[1] = { out };
    unsigned int isz[1] = { in_size };
    int ord[1] = { order ? 1 : 0 };
    const int rc = run_host8(c, 1, false, ins, isz, outs, &cap, ord, nullptr);
    r4x16_trim(c, SINGLE_CALL_KEEP);
    if (rc != 0) { free(out); return nullptr; }
    *out_size = cap;
    return out;
}

extern "C" unsigned char *rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size)
{
    rans4x16_hip_ctx *c = thread_ctx();
    if (!c || !out_size || in_size < 9) return nullptr;                          // rANS_static.c:937
    // the reference's cheap header checks come BEFORE its allocation (rANS_static.c:241-253, :686-698): a hostile
    // nine-byte stream must not cost a 2 GiB malloc and a device staging arena
    if (in[0] > 1) return nullptr;
    if (in_size < (in[0] ? 27u : 26u)) return nullptr;
    const unsigned int csz = (unsigned int)in[1] | ((unsigned int)in[2] << 8) | ((unsigned int)in[3] << 16) | ((unsigned int)in[4] << 24);
    if (csz != in_size - 9) return nullptr;
    unsigned int usz = (unsigned int)in[5] | ((unsigned int)in[6] << 8) | ((unsigned int)in[7] << 16) | ((unsigned int)in[8] << 24);
    if (usz >= INT_MAX) return nullptr;
    unsigned char *out = (unsigned char *)malloc(usz ? usz : 1);
    if (!out) return nullptr;
    const unsigned char *ins[1] = { in };
    unsigned char *outs[1] = { out };
    unsigned int isz[1] = { in_size };
    const int rc = run_host8(c, 1, true, ins, isz, outs, &usz, nullptr, nullptr);
    r4x16_trim(c, SINGLE_CALL_KEEP);
    if (rc != 0) { free(out); return nullptr; }
    *out_size = usz;
    return out;
}
