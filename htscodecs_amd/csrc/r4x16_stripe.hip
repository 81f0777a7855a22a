// r4x16_stripe.hip - X_STRIPE for the device-resident calls (rANS_static4x16pr.c:1154-1216, :1360-1433).
//
// A stripe block is N ordinary sub-blocks (its byte planes), each coded with the best of up to four methods on
// encode.  Device-resident, without a host round trip, that is: a prepare kernel that transposes the planes and
// lays out one internal item per (plane, candidate method), the ordinary pipeline over the internal items (a
// recursive *_dev call on the same context and stream), and a finishing kernel per block - arg-min over the
// candidates, header, winners' payloads into the caller's slot (encode); status check and plane interleave (decode).
// The host cannot know which blocks of a device-resident batch are stripes (decode: the flag and N live in the
// stream; encode: in_size <= 20 drops the flag, :1151), so every block gets the same number of internal items:
// N x K on encode (uniform `order` only: N and K are then host knowledge), `planes` on decode
// (rans4x16_hip_set_dev_stripe_planes; blocks with more planes than that report UNSUPPORTED there and are left to
// the host entry points, which size their staging per block).  Blocks that are not stripes ride along as item 0.
#include "r4x16_host.h"

static __device__ __forceinline__ u32 sv_get(const u8 *p, u32 pos, u32 end, u32 *v)      // varint.h:131-160, bounded
{
    u32 j = 0, used = 0;
    u8 ch;
    if (pos >= end) { *v = 0; return 0; }
    do { ch = p[pos + used++]; j = (j << 7) | (ch & 0x7f); } while ((ch & 0x80) && pos + used < end);
    *v = j;
    return used;
}
static __device__ __forceinline__ u32 sv_put(u8 *cp, u32 v)
{
    u32 groups = 1;
    for (u32 t = v >> 7; t; t >>= 7) groups++;
    for (u32 g = groups; g-- > 0; ) *cp++ = (u8)(((v >> (7 * g)) & 0x7f) | (g ? 0x80 : 0));
    return groups;
}

struct StripeArrays {              // M = n * P internal items; every pointer into the context's stripe arena
    u8 *planes; u64 pl_stride;     // [n][pl_stride]: the blocks' byte planes, plane after plane
    u64 *in_off, *out_off;
    u32 *in_size, *out_cap, *out_size;
    i32 *status, *order;
    u8 *out; u64 oslot;            // encode: [M][oslot] candidate outputs
    u32 *blk;                      // [n][4]: block status, kind (0 plain ride-along, 1 stripe), N, ulen
};

// ---- encode ------------------------------------------------------------------------------------------
struct StripeEncArgs { int order, N, K; int methods[4]; };

__global__ __launch_bounds__(256) void k_stripe_enc_prepare(BatchArgs a, StripeArrays w, StripeEncArgs e)
{
    const u32 b = blockIdx.x, tid = threadIdx.x, P = (u32)(e.N * e.K);
    const u8 *src = a.in + a.in_off[b];
    const u32 n = a.in_size[b];
    u8 *pl = w.planes + (u64)b * w.pl_stride;
    const u64 item0 = (u64)b * P;
    i32 st = ST_OK;
    u32 kind = 1;
    if (a.out_cap[b] < r4x16_bound_hd(n, e.order)) st = ST_CAPACITY;                  // :1158
    else if (n <= 20) kind = 0;                                                       // :1151
    if (tid == 0) { w.blk[4 * b] = (u32)st; w.blk[4 * b + 1] = kind; w.blk[4 * b + 2] = (u32)e.N; w.blk[4 * b + 3] = n; }
    // items: everything idle first
    for (u32 t = tid; t < P; t += 256) {
        const u64 it = item0 + t;
        w.in_off[it] = (u64)(pl - w.planes); w.in_size[it] = 0; w.order[it] = X_CAT | X_NOSZ;
        w.out_off[it] = it * w.oslot; w.out_cap[it] = (u32)w.oslot;
    }
    __syncthreads();
    if (st != ST_OK) return;
    if (kind == 0) {
        if (tid == 0) { w.in_off[item0] = (u64)(src - w.planes); w.in_size[item0] = n; w.order[item0] = e.order & 0xff & ~X_STRIPE; }
        return;
    }
    const u32 N = (u32)e.N, base = n / N, extra = n % N;
    for (u32 i = tid; i < n; i += 256) {                                              // :1168-1180
        const u32 j = i % N, x = i / N;
        pl[j * base + (j < extra ? j : extra) + x] = src[i];
    }
    for (u32 t = tid; t < P; t += 256) {
        const u32 j = t / (u32)e.K, q = t % (u32)e.K;
        const u64 it = item0 + t;
        w.in_off[it] = (u64)(pl - w.planes) + j * base + (j < extra ? j : extra);
        w.in_size[it] = base + (extra > j ? 1u : 0u);
        w.order[it] = e.methods[q] | X_NOSZ;                                          // :1197
    }
}

__global__ __launch_bounds__(WAVE) void k_stripe_enc_pick(BatchArgs a, StripeArrays w, StripeEncArgs e)
{
    __shared__ u32 win[256], woff[256], hdr_len;
    const u32 b = blockIdx.x, lane = threadIdx.x, P = (u32)(e.N * e.K);
    const u64 item0 = (u64)b * P;
    u8 *out = a.out + a.out_off[b];
    i32 st = (i32)w.blk[4 * b];
    const u32 kind = w.blk[4 * b + 1], n = w.blk[4 * b + 3];
    if (st == ST_OK && kind == 0) {                                                   // not a stripe after all: item 0 is the block
        st = w.status[item0];
        const u32 sz = st == ST_OK ? w.out_size[item0] : 0u;
        wave_copy(out, w.out + w.out_off[item0], sz, lane);
        if (lane == 0) { a.status[b] = st; a.out_size[b] = sz; }
        return;
    }
    if (st == ST_OK && lane == 0) {
        u32 hl = 1, body = 0;
        out[0] = (u8)(e.order & ~X_NOSZ);                                             // :1185
        hl += sv_put(out + hl, n);
        out[hl++] = (u8)e.N;
        for (u32 j = 0; j < (u32)e.N && st == ST_OK; j++) {                           // smallest wins, the first on ties (:1199)
            u32 best_sz = n + 10, best = 0;
            for (u32 q = 0; q < (u32)e.K; q++) {
                const u64 it = item0 + j * (u32)e.K + q;
                if (w.status[it] != ST_OK) { st = w.status[it]; break; }
                if (best_sz > w.out_size[it]) { best_sz = w.out_size[it]; best = q; }
            }
            win[j] = j * (u32)e.K + best; woff[j] = body;
            hl += sv_put(out + hl, best_sz);
            body += best_sz;
        }
        hdr_len = hl;
        w.blk[4 * b] = (u32)st;
        woff[255] = body;
    }
    __syncthreads();
    st = (i32)w.blk[4 * b];
    if (st != ST_OK) { if (lane == 0) { a.status[b] = st; a.out_size[b] = 0; } return; }
    for (u32 j = 0; j < (u32)e.N; j++) {
        const u64 it = item0 + win[j];
        wave_copy(out + hdr_len + woff[j], w.out + w.out_off[it], w.out_size[it], lane);
    }
    if (lane == 0) { a.status[b] = ST_OK; a.out_size[b] = hdr_len + woff[255]; }
}

static int ensure_xs(rans4x16_hip_ctx *c, size_t bytes)
{
    if (bytes <= c->xs_bytes) return 0;
    if (c->xs) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(c->xs)); c->xs = nullptr; c->xs_bytes = 0; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b / 2) {
        c->err = "stripe staging of " + std::to_string(bytes >> 20) + " MiB exceeds half of the free device memory";
        return -1;
    }
    if (hipMalloc((void **)&c->xs, bytes) != hipSuccess) { (void)hipGetLastError(); c->xs = nullptr; c->err = "hipMalloc of the stripe arena failed"; return -1; }
    c->xs_bytes = bytes;
    return 0;
}

static size_t carve(StripeArrays *w, u8 *base, size_t n, size_t P, u64 pl_stride, u64 oslot)
{
    const size_t M = n * P;
    size_t off = 0;
    auto take = [&](size_t bytes) { u8 *p = base ? base + off : nullptr; off = align_up(off + bytes, 256); return p; };
    w->planes = take(n * pl_stride); w->pl_stride = pl_stride;
    w->in_off = (u64 *)take(M * 8); w->out_off = (u64 *)take(M * 8);
    w->in_size = (u32 *)take(M * 4); w->out_cap = (u32 *)take(M * 4); w->out_size = (u32 *)take(M * 4);
    w->status = (i32 *)take(M * 4); w->order = (i32 *)take(M * 4);
    w->blk = (u32 *)take(n * 16);
    w->out = take(M * oslot); w->oslot = oslot;
    return off;
}

int r4x16_stripe_compress_dev(rans4x16_hip_ctx *c, int n, const BatchArgs &a, int order, uint32_t max_in_size, hipStream_t s)
{
    StripeEncArgs e;
    e.order = order;
    e.N = order >> 8; if (e.N == 0) e.N = 4;
    static const int methods[4] = {1, 64, 128, 0};                                    // :1192
    e.K = 0;
    for (int j = 0; j < 4; j++) if ((order & methods[j]) == methods[j]) e.methods[e.K++] = methods[j];
    if (e.N > 255) { c->err = "compress_dev: more than 255 stripes"; return -1; }     // :1158 (the reference returns NULL)
    const u32 maxpart = (max_in_size + (u32)e.N - 1) / (u32)e.N;
    const u32 item_max = maxpart > 20 ? maxpart : 20;
    const u64 pl_stride = align_up((size_t)max_in_size + 64, 256);
    const u64 oslot = align_up((size_t)r4x16_compress_bound(item_max, 0xc1) + 64, 256);
    const size_t P = (size_t)e.N * e.K;
    if ((size_t)n * P > (size_t)INT_MAX) { c->err = "compress_dev: too many stripe candidates"; return -1; }
    StripeArrays w;
    if (ensure_xs(c, carve(&w, nullptr, (size_t)n, P, pl_stride, oslot)) != 0) return -1;
    carve(&w, c->xs, (size_t)n, P, pl_stride, oslot);
    // the stripe arena is ordered between calls on different streams like the workspace: nothing of this call touches it
    // before the previous call's last kernel is done, and the event recorded after the finishing kernel covers it
    if (r4x16_ws_order_begin(c, s) != 0) return -1;
    hipLaunchKernelGGL(k_stripe_enc_prepare, dim3(n), dim3(256), 0, s, a, w, e);
    if (rans4x16_hip_compress_dev(c, (int)((size_t)n * P), w.planes, w.in_off, w.in_size, w.out, w.out_off, w.out_cap, w.out_size,
                                  w.status, 0, w.order, item_max, s) != 0) return -1;
    hipLaunchKernelGGL(k_stripe_enc_pick, dim3(n), dim3(WAVE), 0, s, a, w, e);
    HIPCHK(c, hipGetLastError());
    return r4x16_ws_order_end(c, s);
}

// ---- decode ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_stripe_dec_prepare(BatchArgs a, StripeArrays w, u32 P)
{
    const u32 b = blockIdx.x, lane = threadIdx.x;
    const u8 *in = a.in + a.in_off[b];
    const u32 in_size = a.in_size[b], cap = a.out_cap[b];
    u8 *pl = w.planes + (u64)b * w.pl_stride;
    const u64 item0 = (u64)b * P;
    for (u32 t = lane; t < P; t += WAVE) {                                            // idle items: empty input (status EMPTY, ignored)
        const u64 it = item0 + t;
        w.in_off[it] = 0; w.in_size[it] = 0; w.out_off[it] = (u64)(pl - w.planes); w.out_cap[it] = 0;
    }
    __syncthreads();
    if (lane != 0) return;
    i32 st = ST_OK;
    u32 kind = 0, N = 0, ulen = 0;
    if (in_size == 0 || !(in[0] & X_STRIPE)) {
        // an ordinary block: item 0, straight into the caller's slot
        w.in_off[item0] = (u64)(in - a.in); w.in_size[item0] = in_size;
        w.out_off[item0] = (u64)((a.out + a.out_off[b]) - w.planes); w.out_cap[item0] = cap;
    } else {
        kind = 1;
        u32 hl = 1;
        hl += sv_get(in, hl, in_size, &ulen);
        if (hl >= in_size) st = ST_TRUNCATED;                                         // :1367
        else {
            N = in[hl++];
            if (ulen != cap) st = ST_CAPACITY;                                        // :1379: the caller's size must be the stored one
            else if (N == 0) { if (ulen != 0) st = ST_SIZE; }                         // (the reference never returns here)
            else if (N > P || (u64)ulen + 64u > w.pl_stride) st = ST_UNSUPPORTED;     // more planes / a larger block than this call was sized for
            else {
                u64 ctot = 0;
                u32 cl[255];
                for (u32 j = 0; j < N && st == ST_OK; j++) {
                    hl += sv_get(in, hl, in_size, &cl[j]);
                    ctot += cl[j];
                    if (hl > in_size || cl[j] > in_size || cl[j] < 1) st = ST_SIZE;   // :1389
                }
                if (st == ST_OK && hl + ctot > in_size) st = ST_SIZE;                 // :1398
                if (st == ST_OK) {
                    const u32 used = (u32)(hl + ctot);
                    u32 off = hl, first = 0;
                    for (u32 j = 0; j < N; j++) {                                     // sub-block j may read to the end of the block (:1419)
                        const u32 plen = ulen / N + ((ulen % N) > j);
                        w.in_off[item0 + j] = (u64)(in - a.in) + off; w.in_size[item0 + j] = used - off;
                        w.out_off[item0 + j] = (u64)(pl - w.planes) + first; w.out_cap[item0 + j] = plen;
                        off += cl[j]; first += plen;
                    }
                }
            }
        }
    }
    w.blk[4 * b] = (u32)st; w.blk[4 * b + 1] = kind; w.blk[4 * b + 2] = N; w.blk[4 * b + 3] = ulen;
}

__global__ __launch_bounds__(256) void k_stripe_dec_join(BatchArgs a, StripeArrays w, u32 P)
{
    const u32 b = blockIdx.x, tid = threadIdx.x;
    const u64 item0 = (u64)b * P;
    i32 st = (i32)w.blk[4 * b];
    const u32 kind = w.blk[4 * b + 1], N = w.blk[4 * b + 2], ulen = w.blk[4 * b + 3];
    if (kind == 0) {                                                                  // ordinary block: item 0's verdict
        if (tid == 0) { a.status[b] = w.status[item0]; a.out_size[b] = w.status[item0] == ST_OK ? w.out_size[item0] : 0u; }
        return;
    }
    if (st == ST_OK)
        for (u32 j = 0; j < N; j++) {                                                 // :1419-1420
            if (w.status[item0 + j] != ST_OK) { st = w.status[item0 + j]; break; }
            if (w.out_size[item0 + j] != w.out_cap[item0 + j]) { st = ST_SIZE; break; }
        }
    if (st != ST_OK || N == 0) { if (tid == 0) { a.status[b] = st; a.out_size[b] = 0; } return; }
    const u8 *pl = w.planes + (u64)b * w.pl_stride;
    u8 *out = a.out + a.out_off[b];
    const u32 base = ulen / N, extra = ulen % N;
    for (u32 i = tid; i < ulen; i += 256) {                                           // unstripe, utils.h:41-73
        const u32 j = i % N, x = i / N;
        out[i] = pl[j * base + (j < extra ? j : extra) + x];
    }
    if (tid == 0) { a.status[b] = ST_OK; a.out_size[b] = ulen; }
}

int r4x16_stripe_uncompress_dev(rans4x16_hip_ctx *c, int n, const BatchArgs &a, uint32_t max_in_size, uint32_t max_out_cap,
                                uint32_t max_stripe_out, hipStream_t s)
{
    const size_t P = (size_t)c->dev_stripe_planes;
    if ((size_t)n * P > (size_t)INT_MAX) { c->err = "uncompress_dev: too many stripe planes"; return -1; }
    const u64 pl_stride = align_up((size_t)max_stripe_out + 64, 256);
    StripeArrays w;
    if (ensure_xs(c, carve(&w, nullptr, (size_t)n, P, pl_stride, 0)) != 0) return -1;
    carve(&w, c->xs, (size_t)n, P, pl_stride, 0);
    if (r4x16_ws_order_begin(c, s) != 0) return -1;          // (as in r4x16_stripe_compress_dev)
    hipLaunchKernelGGL(k_stripe_dec_prepare, dim3(n), dim3(WAVE), 0, s, a, w, (u32)P);
    // the internal items: inputs relative to the caller's input arena, outputs relative to the plane buffer
    c->in_stripe = true;
    const int rc = rans4x16_hip_uncompress_dev(c, (int)((size_t)n * P), a.in, w.in_off, w.in_size, w.planes, w.out_off, w.out_cap,
                                               w.out_size, w.status, max_in_size, max_out_cap, s);
    c->in_stripe = false;
    if (rc != 0) return -1;
    hipLaunchKernelGGL(k_stripe_dec_join, dim3(n), dim3(256), 0, s, a, w, (u32)P);
    HIPCHK(c, hipGetLastError());
    return r4x16_ws_order_end(c, s);
}
