// r4x16_decode.hip — gfx950 kernels for rans_uncompress_to_4x16 (rANS_static4x16pr.c:1352-1636).
//
// Pipeline for a batch of blocks (one launch each, in stream order):
//   k_dec_front : one wave per block.  Parses the container header (flags, sizes, PACK / RLE
//                 meta), the order-0 / order-1 frequency tables (un-nesting an order-0
//                 compressed order-1 table on the spot) and writes a decode *image* plus a
//                 DecItem per stream.  Replaces :1435-1572, :500-561, :869-1013.
//   k_dec_chain : the hot loop.  A quad of lanes runs the 4 interleaved rANS states of one
//                 stream; QPW streams share a wave; images are staged in LDS when they fit.
//                 Replaces the loops at :574-607 and :1027-1114.
//   k_dec_back  : one wave per block.  RLE expansion, bit-unpacking, CAT copies, final size
//                 and status.  Replaces :1578-1629, rle.c:142-187, pack.c:211-348.
#include "r4x16_dev.h"

// ---- image access: global memory or LDS ------------------------------------------------------
struct GImg {
    gcu8 *p;
    __device__ __forceinline__ u32 ld8(u32 off) const { return p[off]; }
    __device__ __forceinline__ u32 ld32(u32 off) const { return *(gcu32 *)(p + off); }
};
struct LImg {
    const u8 *p;     // points into __shared__
    __device__ __forceinline__ u32 ld8(u32 off) const { return p[off]; }
    __device__ __forceinline__ u32 ld32(u32 off) const { return *(const u32 *)(p + off); }
};

// One table lookup + state update (rANS_static4x16pr.c:576-579 / :1033-1035).
//   r = coarse[m>>2]; refine over at most three following entries; x = freq*(x>>look) + m - start
// Entries carry the start in their HIGH half, so "m >= start_k" is one compare of the whole
// dword against (m<<16 | 0xffff).  Returns the chosen entry (its low half is the link).
template <class IMG>
__device__ __forceinline__ u32 lookup_step(const IMG &img, u32 row, u32 cells, u32 look, u32 mask, u32 &x)
{
    const u32 m = x & mask;
    const u32 r = img.ld8(row + 4 + (m >> 2));
    const u32 eb = row + 4 + cells + 4 * r;
    const u32 e0 = img.ld32(eb), e1 = img.ld32(eb + 4), e2 = img.ld32(eb + 8),
              e3 = img.ld32(eb + 12), e4 = img.ld32(eb + 16);
    const u32 key = (m << 16) | 0xffffu;
    u32 e = e0, en = e1;
    if (key >= e1) { e = e1; en = e2; }
    if (key >= e2) { e = e2; en = e3; }
    if (key >= e3) { e = e3; en = e4; }
    const u32 start = e >> 16;
    const u32 freq = (en >> 16) - start;
    x = __umul24(freq, x >> look) + (m - start);      // freq <= 2^15, x>>look < 2^22: exact mod 2^32
    return e;
}

// ---------------------------------------------------------------------------------------------
// The chain decoder, general form: image and words read straight from global memory.  Used for
// the small nested streams inside k_dec_front and for images too large for LDS.
// Every lane of the wave calls this; lane&3 selects the chain, lane>>2 the stream.
// Per step and chain (rANS_static4x16pr.c:576-597 / :1033-1059):
//     m = x & mask;  (start,freq,symbol) = lookup(context, m);
//     x = freq * (x >> look) + m - start;
//     if (x < 2^15 and two more bytes exist) x = (x << 16) | next word
// The four chains of a stream share one word stream consumed in chain order 0,1,2,3 each step;
// a chain's word index is the stream cursor plus the number of lower chains that also refill
// (4-bit ballot inside the quad) — the "prefix-sum compaction" of the renormalisation.
// Returns non-zero if a context without a table row was used.
// ---------------------------------------------------------------------------------------------
template <int ORDER, class IMG>
__device__ __forceinline__ u32 chain_decode(IMG img, gcu8 *words, u32 words_len, gu8 *out,
                                            u32 out_sz, u32 x, u32 look, bool active, u32 lane)
{
    const u32 k = lane & 3;
    const u32 mask = (1u << look) - 1;
    const u32 cells = 1u << (look - 2);
    const u32 nwords = words_len >> 1;
    u32 count, pos;
    if (ORDER == 0) {
        count = (out_sz + 3 - k) >> 2;            // bytes i with i%4 == k
        pos = k;
    } else {
        const u32 q = out_sz >> 2;                // :1015-1017
        count = q + (k == 3 ? out_sz - 4 * q : 0);
        pos = k * q;
    }
    if (!active) count = 0;

    u32 row = 0, cursor = 0, bad = 0, t = 0;
    if (ORDER == 1 && count) bad = img.ld32(0) & ROW_EMPTY;

    while (wave_any(t < count)) {
        const bool live = t < count;
        bool want = false;
        if (live) {
            const u32 e = lookup_step(img, row, cells, look, mask, x);
            if (ORDER == 0) {
                out[pos] = (u8)e;
                pos += 4;
            } else {
                row = (e & 0xffffu) << 4;
                const u32 hdr = img.ld32(row);
                if (t + 1 < count) bad |= hdr & ROW_EMPTY;
                out[pos] = (u8)hdr;
                pos += 1;
            }
            want = x < RANS_LOW;
        }
        // renormalise: chains refill in order 0..3 from the shared cursor
        const u32 wm = quad_ballot(want, lane);
        const u32 widx = cursor + __popc(wm & ((1u << k) - 1u));
        const bool take = want && widx < nwords;              // rANS_word.h:402-410
        if (take) {
            const u32 w = (u32)words[2 * widx] | ((u32)words[2 * widx + 1] << 8);
            x = (x << 16) | w;
        }
        cursor += __popc(quad_ballot(take, lane));
        t++;
    }
    return bad;
}

// ---------------------------------------------------------------------------------------------
// The chain decoder, hot form.  Same arithmetic; everything on the dependent path lives in LDS:
//   * the image (copied once per stream),
//   * a 128-byte ring of the compressed words per stream, refilled 64 bytes at a time with
//     16-byte global loads issued one refill ahead, so HBM/L2 latency never sits on the chain;
//     the next four candidate words are read from the ring at the top of each step, in parallel
//     with the table lookups, and the right one is picked once the quad ballot is known;
//   * decoded bytes are gathered four at a time per chain and stored as dwords (order-1), so a
//     wave issues one store per four symbols instead of four.
// LDS per stream: image, then RING_BYTES.
// ---------------------------------------------------------------------------------------------
#define RING_BYTES 144u      // 128-byte ring + 8-byte mirror of its head (+8 pad)

typedef u32 GAS __attribute__((aligned(1))) gu32_unaligned;   // global dword store at any byte address

template <int ORDER>
__device__ __forceinline__ u32 chain_decode_lds(const u8 *img_lds, u8 *ring, gcu8 *words, u32 words_len,
                                                gu8 *out, u32 out_sz, u32 x, u32 look, bool active, u32 lane)
{
    const LImg img{img_lds};
    const u32 k = lane & 3;
    const u32 mask = (1u << look) - 1;
    const u32 cells = 1u << (look - 2);
    const u32 nwords = words_len >> 1;
    const u32 below = (1u << k) - 1u;                      // quad lanes below this one
    const u32 qshift = lane & ~3u;
    u32 count;
    gu8 *op;                                               // next output byte of this chain
    if (ORDER == 0) {
        count = (out_sz + 3 - k) >> 2;
        op = out + k;
    } else {
        const u32 q = out_sz >> 2;
        count = q + (k == 3 ? out_sz - 4 * q : 0);
        op = out + (u64)k * q;
    }
    if (!active) count = 0;

    // ---- word ring: stream bytes relative to the 16-byte aligned address below `words` ----------
    gcu8 *abase = (gcu8 *)((u64)words & ~15ull);
    const u32 off0 = (u32)((u64)words & 15ull);
    const u32 avail = off0 + words_len;                   // bytes of abase[] that belong to the input
    auto load_chunk = [&](u32 c) -> u32x4 {               // 16-byte chunk c, zeros past the input
        u32x4 v = {0, 0, 0, 0};
        if (active && c * 16u < avail) v = *(gcu32x4 *)(abase + (u64)c * 16u);
        return v;
    };
    if (active) {
        const u32x4 c0 = load_chunk(k), c1 = load_chunk(k + 4);
        *(u32x4 *)(ring + 16 * k) = c0;
        *(u32x4 *)(ring + 64 + 16 * k) = c1;
        if (k == 0) *(u32x2 *)(ring + 128) = c0.xy;
    }
    u32x4 pend = load_chunk(8 + k);                       // half 2, written at the first crossing
    u32 half = 0;                                         // index of the 64-byte half holding the cursor
    __syncthreads();

    u32 row = 0, cursor = 0, bad = 0, t = 0;
    u32 acc = 0;                                          // order-1: the last (up to) 4 decoded bytes
    u32 hdr = 0;                                          // order-1: header of the row entered last step
    if (ORDER == 1 && count) bad = img.ld32(0) & ROW_EMPTY;

    // Four steps per trip: one loop test, one dword store and one ring check per trip.
    while (wave_any(t < count)) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool live = (t + u) < count;
            // next four candidate words (8 bytes at any byte alignment) from the ring; issued
            // before the table lookups so that their latency hides under them
            const u32 cb = off0 + 2 * cursor;
            const u32 ra = cb & 124u;
            // (volatile: keeps the compiler from sinking these reads into the refill branch, which
            //  would put their latency back on the dependent path)
            const u32 d0 = *(const volatile u32 *)(ring + ra), d1 = *(const volatile u32 *)(ring + ra + 4),
                      d2 = *(const volatile u32 *)(ring + ra + 8);
            const u32 sh = (cb & 3u) * 8u;

            u32 xn = x;
            const u32 e = lookup_step(img, row, cells, look, mask, xn);
            u32 byte0 = 0;
            if (ORDER == 0) {
                byte0 = e & 0xffu;
            } else {
                // the byte of the symbol decoded one step ago and the flags of the row in use now
                if (u > 0 || t > 0) {
                    if (live) bad |= hdr & ROW_EMPTY;
                    acc = (t + u <= count) ? __builtin_amdgcn_alignbit(hdr, acc, 8) : acc;
                }
                // symbols t-4 .. t-1 are now in acc, oldest in the low byte
                if (u == 0 && t >= 4 && t <= count) { *(gu32_unaligned *)op = acc; op += 4; }
                const u32 rown = (e & 0xffffu) << 4;
                const u32 hn = img.ld32(rown);
                hdr = live ? hn : hdr;
                row = live ? rown : row;
            }
            x = live ? xn : x;
            const bool want = live && x < RANS_LOW;

            // renormalise: chains refill in order 0..3 from the shared cursor.  After the first
            // refusal (stream exhausted) no later request can succeed either, so the cursor may
            // simply advance by the number of requests (rANS_word.h:402-410).
            const u64 wb = __ballot(want);
            const u32 wm = (u32)(wb >> qshift) & 0xfu;
            const u32 pre = __popc(wm & below);
            const bool take = want && cursor + pre < nwords;
            const u32 wlo = __builtin_amdgcn_alignbit(d1, d0, sh), whi = __builtin_amdgcn_alignbit(d2, d1, sh);
            const u32 w2 = (pre & 2u) ? whi : wlo;
            const u32 w = (w2 >> ((pre & 1u) * 16u)) & 0xffffu;
            x = take ? ((x << 16) | w) : x;
            cursor += __popc(wm);

            if (ORDER == 0) {
                // the quad's four bytes are consecutive: lane 0 stores them as one dword
                const u32 b1 = quad_bcast1(byte0), b2 = quad_bcast2(byte0), b3 = quad_bcast3(byte0);
                const u32 l3 = quad_bcast3(live ? 1u : 0u);
                const u32 dw = byte0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                if (l3) { if (k == 0) *(gu32_unaligned *)(op + 4 * (u64)(t + u)) = dw; }
                else if (live) op[4 * (u64)(t + u)] = (u8)byte0;
            }
        }
        t += 4;

        // ring refill when the cursor has entered a new 64-byte half (at most 32 bytes ago)
        const u32 nh = (off0 + 2 * cursor) >> 6;
        if (wave_any(nh != half)) {
            if (nh != half) {
                // half `nh+1` was requested at the previous crossing: park it in the slots just vacated
                const u32 slot = ((nh + 1) & 1u) * 64u + 16u * k;
                *(u32x4 *)(ring + slot) = pend;
                if (slot == 0) *(u32x2 *)(ring + 128) = pend.xy;
                pend = load_chunk(4 * (nh + 2) + k);
                half = nh;
            }
            __syncthreads();
        }
    }
    if (ORDER == 1 && count) {
        // t steps ran.  A chain whose count equals t still has its last byte in hdr and its last
        // four symbols unstored; any other chain has count%4 symbols left, already in acc.
        u32 stored = count & ~3u;
        if (count == t) { acc = __builtin_amdgcn_alignbit(hdr, acc, 8); stored = t - 4; }
        const u32 rem = count - stored;                    // 0..4, in the top `rem` bytes of acc
        for (u32 j = 0; j < rem; j++) op[j] = (u8)(acc >> (8 * (4 - rem + j)));
    }
    return bad;
}

// ---------------------------------------------------------------------------------------------
// Table parsing (one lane) and image building (whole wave).
// ---------------------------------------------------------------------------------------------
struct FrontShared {
    u32 F[256];        // frequencies of the row being parsed, by byte value
    u32 ent[264];      // entries of the row being built
    u8  present[256];  // alphabet of the stream (order-1: F0)
    u8  idx_of[256];   // byte -> compact index        (order-1)
    u8  alpha[256];    // compact index -> byte        (order-1)
    // scalars handed from lane 0 to the wave
    i32 status;
    u32 nnz, hdr, pos, nsym, bits, look, go;
    u32 R[4];
    u32 words_pos;
};

// rANS_static4x16pr.c:208-255 (see oracle/rans4x16_oracle.c get_alphabet for the derivation of
// the single-loop form).  Marks present[]; returns bytes consumed, 0 on failure.
__device__ u32 get_alphabet(ByteSrc &s, u32 pos, u32 end, u8 *present)
{
    if (pos >= end) return 0;
    u32 p = pos;
    u32 implicit = 0;
    u32 j = s.at(p++);
    bool more = (p + 2 < end) || j;
    while (more) {
        present[j] = 1;
        if (p >= end) return 0;
        const u32 nx = s.at(p);
        if (!implicit && j + 1 == nx) {
            if (p + 1 >= end) return 0;
            j = nx;
            p++;
            implicit = s.at(p++);
        } else if (implicit) {
            implicit--;
            if (++j > 255) return 0;
        } else {
            j = nx;
            p++;
        }
        more = j && p < end;
    }
    return p - pos;
}

// Turn S.F[] (by byte) into S.ent[] for one row.  `link_of(byte)` is the high half of an entry.
// Lane 0 only.  Mirrors the checks at :538-552 / :985-997.  Returns false on a bad table.
template <class LINK>
__device__ bool make_entries(FrontShared &S, const u8 *in_alphabet, u32 total, u32 bits, LINK link_of)
{
    // normalise_freq_shift :168-179
    u32 sh = 0;
    if (total != 0 && total != (1u << bits)) {
        u32 size = total;
        while (size < (1u << bits)) { size *= 2; sh++; }
    }
    u32 x = 0, nnz = 0;
    for (u32 j = 0; j < 256; j++) {
        if (!in_alphabet[j]) continue;
        const u32 f = S.F[j] << sh;
        if (!f) continue;
        if (f > (1u << bits) - x) return false;
        S.ent[nnz++] = (x << 16) | link_of(j);
        x += f;
    }
    if (x != (1u << bits)) return false;
    S.ent[nnz] = (1u << bits) << 16;
    S.ent[nnz + 1] = S.ent[nnz + 2] = S.ent[nnz + 3] = 0xffff0000u;
    S.nnz = nnz;
    return true;
}

// Whole wave: write one row (header, coarse map, entries) from S.ent / S.nnz / S.hdr.
__device__ void write_row(u8 *rowp, const FrontShared &S, u32 look, u32 lane)
{
    const u32 cells = 1u << (look - 2);
    const u32 nnz = S.nnz;
    u8 *coarse = rowp + 4;
    u32 *ent = (u32 *)(rowp + 4 + cells);
    if (lane == 0) *(u32 *)rowp = S.hdr | (nnz << 16);
    if (nnz == 0) {
        for (u32 c = lane; c < cells; c += WAVE) coarse[c] = 0;
        if (lane < 5) ent[lane] = lane == 0 ? 0u : 0xffff0000u;   // in-bounds filler, row is flagged EMPTY
        return;
    }
    for (u32 r = lane; r < nnz; r += WAVE) {
        const u32 lo = S.ent[r] >> 16, hi = S.ent[r + 1] >> 16;
        u32 c0 = (lo + 3) >> 2, c1 = (hi + 3) >> 2;
        if (c1 > cells) c1 = cells;
        for (u32 c = c0; c < c1; c++) coarse[c] = (u8)r;
    }
    for (u32 r = lane; r < nnz + 4; r += WAVE) ent[r] = S.ent[r];
}

// Order-0 stream front end: src[pos, pos+len) holds table, states, words.
// rANS_static4x16pr.c:500-561.  All lanes call; on return S.status / S.R / S.words_pos are set
// and the single-row image is at `img`.
__device__ void o0_front(ByteSrc &src, u32 pos, u32 len, u32 out_sz, u8 *img, FrontShared &S, u32 lane)
{
    for (u32 j = lane; j < 256; j += WAVE) { S.present[j] = 0; S.F[j] = 0; }
    __syncthreads();
    if (lane == 0) {
        S.status = ST_OK;
        S.hdr = 0;
        if (len < 16) S.status = ST_TRUNCATED;                        // :503
        else if (out_sz >= 0x7fffffffu) S.status = ST_SIZE;           // :506
        else {
            const u32 end = pos + len, tab_end = end - 8;             // :516
            u32 p = pos;
            // decode_freq :271-289 (a failed alphabet parse is not an error by itself)
            if (p == tab_end) S.status = ST_TABLE;
            else {
                p += get_alphabet(src, p, tab_end, S.present);
                u32 total = 0;
                for (u32 j = 0; j < 256; j++) {
                    if (!S.present[j]) continue;
                    u32 f;
                    p += var_get(src, p, tab_end, &f);
                    S.F[j] = f;
                    total += f;
                }
                if (p == pos) S.status = ST_TABLE;                    // fsz == 0 :531
                else if (!make_entries(S, S.present, total, O0_BITS, [](u32 j) { return j; }))
                    S.status = ST_TABLE;
                else if (p + 16 > end) S.status = ST_TRUNCATED;       // :554
                else {
                    for (u32 k = 0; k < 4; k++, p += 4) {
                        const u32 r = (u32)src.at(p) | ((u32)src.at(p + 1) << 8) |
                                      ((u32)src.at(p + 2) << 16) | ((u32)src.at(p + 3) << 24);
                        S.R[k] = r;
                        if (r < RANS_LOW) S.status = ST_STATE;        // :558-561
                    }
                    S.words_pos = p;
                }
            }
        }
    }
    __syncthreads();
    if (S.status == ST_OK) write_row(img, S, O0_BITS, lane);
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// k_dec_front
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_dec_front(BatchArgs a, DecWs ws, int base)
{
    __shared__ FrontShared S;
    __shared__ struct {
        i32 status;
        u32 order, pay_pos, pay_len, s1_size, compressed, usz, csz, tab_pos, after_table;
        u32 bits;
    } H;

    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    DecDesc *D = &ws.desc[b];
    DecItem *I0 = &ws.items[b], *I1 = &ws.items[gridDim.x + b];   // payload items first, meta items after
    const u8 *in = a.in + a.in_off[i];
    const u32 in_size = a.in_size[i];
    const u32 cap = a.out_cap[i];
    u8 *out = a.out + a.out_off[i];
    u8 *img = ws.images + (u64)b * DEC_IMG_SLOT;
    u8 *tbuf = ws.tbuf + (u64)b * TBUF_BYTES;
    ByteSrc src(in);

    // ---- container header: flags, sizes (:1435-1467) ----------------------------------------
    if (lane == 0) {
        I0->active = 0; I1->active = 0;
        I0->blk = b; I1->blk = b;
        D->status = ST_OK; D->cat_src = 0; D->cat_len = 0; D->osz = 0; D->s1_size = 0;
        D->pack_per = 1; D->rle_meta_len = 0;
        i32 st = ST_OK;
        u32 pos = 0, osz = 0, flags = 0;
        H.pay_len = 0;
        if (in_size == 0) st = ST_EMPTY;                               // :1357
        else {
            flags = src.at(pos++);
            if (flags & X_STRIPE) st = ST_UNSUPPORTED;                 // host entry points split stripes
            else if (flags & (X_PACK | X_RLE)) st = ST_UNSUPPORTED;    // TODO(milestone 2): k_dec_back transforms
            else {
                if (!(flags & X_NOSZ)) pos += var_get(src, pos, in_size, &osz);
                else osz = cap;
                if (cap < osz) st = ST_CAPACITY;                       // :1464
            }
        }
        D->flags = flags;
        if (st == ST_OK) {
            D->osz = osz;
            D->s1 = (u64)out; D->s2 = (u64)out; D->s3 = (u64)out;
            u32 s1_size = osz;
            const u32 left = in_size - pos;
            if (left == 0) {                                           // :1592-1595
                s1_size = 0;
            } else if (flags & X_CAT) {                                // :1578-1584
                if (s1_size > left || s1_size > osz) st = ST_SIZE;
                else { D->cat_src = (u64)(in + pos); D->cat_len = s1_size; }
            } else {
                H.pay_len = left;
            }
            D->s1_size = s1_size;
            H.pay_pos = pos; H.s1_size = s1_size; H.order = flags & 1;
        }
        H.status = st;
        D->status = st;
    }
    __syncthreads();
    if (H.status != ST_OK || H.pay_len == 0) return;

    const u32 pay_pos = H.pay_pos, pay_len = H.pay_len, s1_size = H.s1_size;

    if (H.order == 0) {
        // ---- order-0 payload ----------------------------------------------------------------
        o0_front(src, pay_pos, pay_len, s1_size, img, S, lane);
        if (lane == 0) {
            D->status = S.status;
            if (S.status == ST_OK) {
                I0->words = (u64)(in + S.words_pos);
                I0->words_len = pay_pos + pay_len - S.words_pos;
                I0->out = D->s1; I0->out_sz = s1_size; I0->image = (u64)img;
                I0->img_bytes = img_row_bytes(S.nnz, O0_BITS);
                I0->look = O0_BITS; I0->order = 0;
                for (int k = 0; k < 4; k++) I0->R[k] = S.R[k];
                __threadfence();
                I0->active = s1_size != 0;
            }
        }
        return;
    }

    // ---- order-1 payload (:869-1013) ------------------------------------------------------------
    if (lane == 0) {
        i32 st = ST_OK;
        const u32 end = pay_pos + pay_len;
        if (pay_len < 16) st = ST_TRUNCATED;                           // :872
        else if (s1_size >= 0x7fffffffu) st = ST_SIZE;                 // :875
        else {
            u32 p = pay_pos;
            const u32 hb = src.at(p++);
            H.bits = hb >> 4;
            H.compressed = hb & 1;
            if (H.bits < 10) st = ST_UNSUPPORTED;      // the reference reads unwritten slots here
            else if (H.compressed) {                                   // :944-955
                u32 usz, csz;
                p += var_get(src, p, end, &usz);
                p += var_get(src, p, end, &csz);
                if ((long)csz >= (long)(end - p) - 16) st = ST_TRUNCATED;
                else if (usz > TBUF_BYTES) st = ST_UNSUPPORTED;        // no valid table is this big
                else { H.usz = usz; H.csz = csz; H.tab_pos = p; H.after_table = p + csz; }
            } else {
                H.tab_pos = p;
            }
        }
        H.status = st;
    }
    __syncthreads();
    if (H.status != ST_OK) { if (lane == 0) D->status = H.status; return; }

    const u32 bits = H.bits;
    const u32 look = bits == 12 ? 12 : 10;                             // :1027, :1071
    const bool compressed = H.compressed != 0;

    if (compressed) {
        // un-nest the table: an order-0 stream of usz bytes inside src[tab_pos, tab_pos+csz)
        u8 *img0 = img + IMG_MAX_BYTES;
        o0_front(src, H.tab_pos, H.csz, H.usz, img0, S, lane);
        if (S.status != ST_OK) { if (lane == 0) D->status = S.status; return; }
        __threadfence();
        GImg g{to_global((const u8 *)img0)};
        chain_decode<0>(g, to_global(in + S.words_pos), H.tab_pos + H.csz - S.words_pos, to_global(tbuf), H.usz,
                        S.R[lane & 3], O0_BITS, lane < 4, lane);
        __threadfence();
        __syncthreads();
    }

    ByteSrc tsrc(compressed ? tbuf : in, compressed);
    const u32 tend = compressed ? H.usz : pay_pos + pay_len;

    // alphabet F0 (:958-965) and the compact alphabet F0 ∪ {0}
    for (u32 j = lane; j < 256; j += WAVE) S.present[j] = 0;
    __syncthreads();
    if (lane == 0) {
        u32 p = compressed ? 0 : H.tab_pos;
        const u32 used = get_alphabet(tsrc, p, tend, S.present);
        p += used;
        i32 st = ST_OK;
        if (!used || p >= tend) st = ST_TABLE;
        u32 n = 0;
        for (u32 j = 0; j < 256; j++)
            if (S.present[j] || j == 0) { S.idx_of[j] = (u8)n; S.alpha[n] = (u8)j; n++; }
        S.nsym = n;
        S.pos = p;
        H.status = st;
    }
    __syncthreads();
    if (H.status != ST_OK) { if (lane == 0) D->status = H.status; return; }

    const u32 nsym = S.nsym;
    const u32 stride = img_row_bytes(nsym, look);

    // rows, in byte order of the compact alphabet (:967-998)
    for (u32 ci = 0; ci < nsym; ci++) {
        if (lane == 0) {
            const u32 ctx = S.alpha[ci];
            S.hdr = ctx;
            S.nnz = 0;
            S.go = 1;
            if (!S.present[ctx]) {
                S.hdr |= ROW_EMPTY;                                    // byte 0 outside F0
            } else {
                // decode_freq_d :327-358
                u32 p = S.pos, total = 0, zeros = 0;
                bool ok = p != tend;
                for (u32 j = 0; j < 256; j++) S.F[j] = 0;
                for (u32 j = 0; ok && j < 256 && p < tend; j++) {
                    if (!S.present[j]) continue;
                    u32 f;
                    if (zeros) { f = 0; zeros--; }
                    else {
                        p += var_get(tsrc, p, tend, &f);
                        if (f == 0) {
                            if (p >= tend) { ok = false; break; }
                            zeros = tsrc.at(p++);
                        }
                    }
                    S.F[j] = f;
                    total += f;
                }
                if (!ok || p == S.pos) { H.status = ST_TABLE; S.go = 0; }
                else {
                    S.pos = p;
                    if (total == 0) S.hdr |= ROW_EMPTY;                // :977-980
                    else if (!make_entries(S, S.present, total, bits,
                                           [&](u32 j) { return (u32)S.idx_of[j] * (stride >> 4); })) {
                        H.status = ST_TABLE; S.go = 0;
                    }
                }
            }
        }
        __syncthreads();
        if (!S.go) break;
        write_row(img + (u64)ci * stride, S, look, lane);
        __syncthreads();
    }
    if (H.status != ST_OK) { if (lane == 0) D->status = H.status; return; }

    if (lane == 0) {
        u32 p = compressed ? H.after_table : S.pos;                    // :1000-1001
        const u32 end = pay_pos + pay_len;
        i32 st = ST_OK;
        if (p + 16 > end) st = ST_TRUNCATED;                           // :1005
        else {
            for (u32 k = 0; k < 4; k++, p += 4) {
                const u32 r = (u32)src.at(p) | ((u32)src.at(p + 1) << 8) |
                              ((u32)src.at(p + 2) << 16) | ((u32)src.at(p + 3) << 24);
                I0->R[k] = r;
                if (r < RANS_LOW) st = ST_STATE;                       // :1010-1013
            }
        }
        D->status = st;
        if (st == ST_OK) {
            I0->words = (u64)(in + p);
            I0->words_len = end - p;
            I0->out = D->s1; I0->out_sz = s1_size; I0->image = (u64)img;
            I0->img_bytes = nsym * stride;
            I0->look = look; I0->order = 1;
            __threadfence();
            I0->active = s1_size != 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_dec_chain: QPW streams per wave (one per quad).
// The host cannot know image sizes without a device->host round trip, so it launches one grid
// per LDS size class; a stream runs in the launch whose class (lo, hi] contains its LDS need
// (image + word ring) and every other wave exits at once.  LDS_IMG=false is the catch-all for
// images too big for LDS (lo = largest class).
// ---------------------------------------------------------------------------------------------
template <bool LDS_IMG>
__global__ __launch_bounds__(WAVE) void k_dec_chain(const DecItem *items, DecDesc *desc, int nitems,
                                                    int qpw, u32 lds_per_item, u32 cls_lo, u32 cls_hi)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 lane = threadIdx.x;
    const u32 quad = lane >> 2;
    const int it = (int)blockIdx.x * qpw + (int)quad;
    const bool mine = quad < (u32)qpw && it < nitems;
    const DecItem *I = &items[mine ? it : 0];
    bool active = mine && I->active;
    const u32 img_bytes = active ? I->img_bytes : 0u;
    const u32 need = img_bytes + RING_BYTES;
    active = active && need > cls_lo && need <= cls_hi;
    if (!wave_any(active)) return;

    // fields are read one by one (a register copy of the struct indexed by lane would spill)
    gcu8 *words = (gcu8 *)I->words;
    gu8 *out = (gu8 *)I->out;
    const u32 words_len = I->words_len, out_sz = I->out_sz, look = active ? I->look : 12u;
    const u32 order = active ? I->order : 2u;
    const u32 x0 = I->R[lane & 3];

    u32 bad;
    if (LDS_IMG) {
        // cooperative copy: the whole wave copies each quad's image in turn (16-byte pieces)
        const u64 my_img = active ? I->image : 0ull;
        for (int qd = 0; qd < qpw; qd++) {
            const u64 src = __shfl(my_img, qd * 4);
            const u32 nb = __shfl(img_bytes, qd * 4);
            if (!src) continue;
            gcu32x4 *s = (gcu32x4 *)src;
            u32x4 *dd = (u32x4 *)(lds + (u64)qd * lds_per_item);
            for (u32 j = lane; j < (nb >> 4); j += WAVE) dd[j] = s[j];
        }
        __syncthreads();
        const u8 *im = lds + (u64)quad * lds_per_item;
        u8 *ring = lds + (u64)quad * lds_per_item + (lds_per_item - RING_BYTES);
        // order-0 and order-1 streams may share a wave: run the two loops back to back
        bad = chain_decode_lds<1>(im, ring, words, words_len, out, out_sz, x0, look, active && order == 1, lane);
        bad |= chain_decode_lds<0>(im, ring, words, words_len, out, out_sz, x0, look, active && order == 0, lane);
    } else {
        GImg im{(gcu8 *)I->image};
        bad = chain_decode<1>(im, words, words_len, out, out_sz, x0, look, active && order == 1, lane);
        bad |= chain_decode<0>(im, words, words_len, out, out_sz, x0, look, active && order == 0, lane);
    }
    if (active && bad) desc[I->blk].status = ST_CONTEXT;
}

// ---------------------------------------------------------------------------------------------
// k_dec_back: CAT copies, final size and status.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_dec_back(BatchArgs a, DecWs ws, int base)
{
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    const DecDesc *D = &ws.desc[b];
    const i32 st = D->status;
    if (st == ST_OK && D->cat_src) wave_copy((u8 *)D->s1, (const u8 *)D->cat_src, D->cat_len, lane);
    if (lane == 0) {
        a.status[i] = st;
        a.out_size[i] = st == ST_OK ? D->s1_size : 0;
    }
}

// ---- host-callable launchers (r4x16_api.hip) ---------------------------------------------------
extern "C" void r4x16_launch_dec_front(const BatchArgs *a, const DecWs *ws, int base, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_dec_front, dim3(nblk), dim3(WAVE), 0, s, *a, *ws, base);
}
// LDS size classes: {bytes per stream, streams per wave}.  Streams per CU = floor(160 KB / (qpw*bytes)) * qpw.
static const struct { u32 bytes; int qpw; } DEC_CLASSES[] = {
    {2560, 16}, {5120, 8}, {10240, 4}, {16384, 1}, {22528, 1}, {40960, 1}, {81920, 1}, {163840, 1},
};
extern "C" void r4x16_launch_dec_chain(const DecWs *ws, int nitems, hipStream_t s)
{
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void *)k_dec_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        once = true;
    }
    u32 lo = 0;
    for (const auto &c : DEC_CLASSES) {
        const int grid = (nitems + c.qpw - 1) / c.qpw;
        hipLaunchKernelGGL(k_dec_chain<true>, dim3(grid), dim3(WAVE), (size_t)c.qpw * c.bytes, s,
                           ws->items, ws->desc, nitems, c.qpw, c.bytes, lo, c.bytes);
        lo = c.bytes;
    }
    const int grid = (nitems + 15) / 16;
    hipLaunchKernelGGL(k_dec_chain<false>, dim3(grid), dim3(WAVE), 0, s,
                       ws->items, ws->desc, nitems, 16, 0u, lo, 0xffffffffu);
}
extern "C" void r4x16_launch_dec_back(const BatchArgs *a, const DecWs *ws, int base, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_dec_back, dim3(nblk), dim3(WAVE), 0, s, *a, *ws, base);
}
