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
    const u8 *p;
    __device__ __forceinline__ u32 ld8(u32 off) const { return p[off]; }
    __device__ __forceinline__ u32 ld32(u32 off) const { return *(const u32 *)(p + off); }
};
struct LImg {
    const u8 *p;     // points into __shared__
    __device__ __forceinline__ u32 ld8(u32 off) const { return p[off]; }
    __device__ __forceinline__ u32 ld32(u32 off) const { return *(const u32 *)(p + off); }
};

// ---------------------------------------------------------------------------------------------
// The chain decoder.  Every lane of the wave calls this; lane&3 selects the chain, lane>>2 the
// stream.  Per step and chain (rANS_static4x16pr.c:576-597 / :1033-1059):
//     m = x & mask;  (start,freq,symbol) = lookup(context, m);
//     x = freq * (x >> look) + m - start;
//     if (x < 2^15 and two more bytes exist) x = (x << 16) | next word
// The four chains of a stream share one word stream consumed in chain order 0,1,2,3 each step;
// a chain's word index is the stream cursor plus the number of lower chains that also refill
// (4-bit ballot inside the quad) — the "prefix-sum compaction" of the renormalisation.
// Returns non-zero if a context without a table row was used.
// ---------------------------------------------------------------------------------------------
template <int ORDER, class IMG>
__device__ __forceinline__ u32 chain_decode(IMG img, const u8 *words, u32 words_len, u8 *out,
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
            const u32 m = x & mask;
            const u32 r = img.ld8(row + 4 + (m >> 2));
            const u32 eb = row + 4 + cells + 4 * r;
            const u32 e0 = img.ld32(eb), e1 = img.ld32(eb + 4), e2 = img.ld32(eb + 8),
                      e3 = img.ld32(eb + 12), e4 = img.ld32(eb + 16);
            u32 e = e0, en = e1;
            if (m >= (e1 & 0xffffu)) { e = e1; en = e2; }
            if (m >= (e2 & 0xffffu)) { e = e2; en = e3; }
            if (m >= (e3 & 0xffffu)) { e = e3; en = e4; }
            const u32 start = e & 0xffffu;
            const u32 freq = (en & 0xffffu) - start;
            x = freq * (x >> look) + m - start;
            u32 byte;
            if (ORDER == 0) {
                byte = e >> 16;
                out[pos] = (u8)byte;
                pos += 4;
            } else {
                row = (e >> 16) << 4;
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
        S.ent[nnz++] = x | (link_of(j) << 16);
        x += f;
    }
    if (x != (1u << bits)) return false;
    S.ent[nnz] = (1u << bits);
    S.ent[nnz + 1] = S.ent[nnz + 2] = S.ent[nnz + 3] = 0xffffu;
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
        if (lane < 4) ent[lane] = lane == 0 ? 0u : 0xffffu;       // in-bounds garbage, flagged EMPTY
        return;
    }
    for (u32 r = lane; r < nnz; r += WAVE) {
        const u32 lo = S.ent[r] & 0xffffu, hi = S.ent[r + 1] & 0xffffu;
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
    DecItem *I0 = &ws.items[2 * b], *I1 = &ws.items[2 * b + 1];
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
        GImg g{img0};
        chain_decode<0>(g, in + S.words_pos, H.tab_pos + H.csz - S.words_pos, tbuf, H.usz,
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
// k_dec_chain: QPW streams per wave (one per quad).  LDS_IMG: stage each stream's image in LDS.
// ---------------------------------------------------------------------------------------------
template <bool LDS_IMG>
__global__ __launch_bounds__(WAVE) void k_dec_chain(const DecItem *items, DecDesc *desc, int nitems,
                                                    int qpw, u32 lds_per_item)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 lane = threadIdx.x;
    const u32 quad = lane >> 2;
    const int it = (int)blockIdx.x * qpw + (int)quad;
    const bool mine = quad < (u32)qpw && it < nitems;
    const DecItem *I = &items[mine ? it : 0];
    const bool active = mine && I->active;
    // fields are read one by one (a register copy of the struct indexed by lane would spill)
    const u8 *words = (const u8 *)I->words;
    u8 *out = (u8 *)I->out;
    const u32 words_len = I->words_len, out_sz = I->out_sz, look = active ? I->look : 12u;
    const u32 order = active ? I->order : 2u;
    const u32 x0 = I->R[lane & 3];

    u32 bad;
    if (LDS_IMG) {
        // cooperative copy: the whole wave copies each quad's image in turn (16-byte pieces)
        const u64 my_img = active ? I->image : 0ull;
        const u32 my_nb = active ? I->img_bytes : 0u;
        for (int qd = 0; qd < qpw; qd++) {
            const u64 src = __shfl(my_img, qd * 4);
            const u32 nb = __shfl(my_nb, qd * 4);
            if (!src) continue;
            const uint4 *s = (const uint4 *)src;
            uint4 *d = (uint4 *)(lds + (u64)qd * lds_per_item);
            for (u32 j = lane; j < (nb >> 4); j += WAVE) d[j] = s[j];
        }
        __syncthreads();
        LImg im{lds + (u64)quad * lds_per_item};
        // order-0 and order-1 streams may share a wave: run the two loops back to back
        bad = chain_decode<1>(im, words, words_len, out, out_sz, x0, look, order == 1, lane);
        bad |= chain_decode<0>(im, words, words_len, out, out_sz, x0, look, order == 0, lane);
    } else {
        GImg im{(const u8 *)I->image};
        bad = chain_decode<1>(im, words, words_len, out, out_sz, x0, look, order == 1, lane);
        bad |= chain_decode<0>(im, words, words_len, out, out_sz, x0, look, order == 0, lane);
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
extern "C" void r4x16_launch_dec_chain(const DecWs *ws, int nitems, int qpw, u32 lds_per_item, hipStream_t s)
{
    const int grid = (nitems + qpw - 1) / qpw;
    if (lds_per_item)
        hipLaunchKernelGGL(k_dec_chain<true>, dim3(grid), dim3(WAVE), (size_t)qpw * lds_per_item, s,
                           ws->items, ws->desc, nitems, qpw, lds_per_item);
    else
        hipLaunchKernelGGL(k_dec_chain<false>, dim3(grid), dim3(WAVE), 0, s,
                           ws->items, ws->desc, nitems, qpw, 0u);
}
extern "C" void r4x16_launch_dec_back(const BatchArgs *a, const DecWs *ws, int base, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_dec_back, dim3(nblk), dim3(WAVE), 0, s, *a, *ws, base);
}
