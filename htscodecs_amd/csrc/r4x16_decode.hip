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
#include <stdlib.h>
#include <stdio.h>
#include <mutex>
#include <type_traits>
#include "r4x16_dev.h"
#include "r4x16_sched.h"

static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- image access: global memory or LDS ------------------------------------------------------
struct GImg {
    gcu8 *p;
    __device__ __forceinline__ u32 ld16(u32 off) const { return *(GAS const u16 *)(p + off); }
    __device__ __forceinline__ u32 ld32(u32 off) const { return *(gcu32 *)(p + off); }
    __device__ __forceinline__ u32x2 ld64(u32 off) const { return *(GAS const u32x2_a4 *)(p + off); }
    __device__ __forceinline__ u32x2 ld64_now(u32 off) const { return ld64(off); }
};
struct LImg {
    u32 base;        // LDS byte address (32-bit arithmetic keeps row addressing to one multiply-add)
    __device__ __forceinline__ u32 ld16(u32 off) const { return *(LAS const u16 *)(unsigned long)(base + off); }
    __device__ __forceinline__ u32 ld32(u32 off) const { return *(LAS const u32 *)(unsigned long)(base + off); }
    __device__ __forceinline__ u32x2 ld64(u32 off) const { return *(LAS const u32x2_a4 *)(unsigned long)(base + off); }
    // two dwords, read where the source says (volatile: not sunk below later arithmetic or into a branch)
    __device__ __forceinline__ u32x2 ld64_now(u32 off) const
    {
        return *(LAS const volatile u32x2_a4 *)(unsigned long)(base + off);
    }
};
__device__ __forceinline__ u32 lds_addr(const void *p) { return (u32)(unsigned long)(LAS const u8 *)p; }

// number of the four u16 separators in v that are <= m, without compares: separators are stored
// clamped to 0x7FFF (m < 4096, so the clamp changes no answer), hence in each 16-bit half
// (m | 0x8000) - sep cannot borrow from its neighbour and keeps bit 15 exactly when sep <= m.
__device__ __forceinline__ u32 count_le(u32 mmh /* (m | m << 16) | 0x80008000 */, u32x2 v)
{
    // the four flag bits sit in bytes 1 and 3 of the two differences: one byte permute gathers them
    const u32 g = __builtin_amdgcn_perm(mmh - v.y, mmh - v.x, 0x07050301u);
    return __popc(g & 0x80808080u);
}

// One table lookup + state update (rANS_static4x16pr.c:576-579 / :1033-1035) on the search tree
// described in r4x16_common.h.  `row` is the byte offset of the context's row in the image.
// Returns the compact symbol index; x becomes freq * (x >> look) + m - start.
// Order-1, two levels: the next step's root separators depend on the symbol found here, and their
// LDS round trip is the longest thing between two steps.  The symbol is e or e + 1, and e is known
// a dozen instructions before the last compare: both candidate roots are requested at that point.
struct RootSpec {
    u32 rows, roww;      // in: address of row 0, row stride
    u32 rowE;            // out: address of row e
    u32x2 ra, rb;        // out: roots of rows e and e + 1
    bool up;             // out: the symbol is e + 1
};

template <int LV, class IMG>
__device__ __forceinline__ u32 lookup_step(const IMG &img, u32 row, u32 look, u32 mask, u32 &x,
                                           const u32x2 *root = nullptr /* LV 2: img.ld64(row), read ahead */,
                                           RootSpec *spec = nullptr)
{
    const u32 m = x & mask;
    const u32 mm = __umul24(m, 0x10001u) + 0x80008000u;     // m | m << 16 | flags (m < 2^15: no overlap)
    u32 e, c0, c1, c2;
    if (LV == 3) {
        const u32 a = count_le(mm, img.ld64(row));
        const u32 b = count_le(mm, img.ld64(row + 8 + 8 * a));
        const u32 g = row + 48u + 60u * a + 12u * b;
        const u32 E0 = img.ld32(g), E1 = img.ld32(g + 4), E2 = img.ld32(g + 8), E3 = img.ld32(g + 12);
        const u32 p12 = __builtin_amdgcn_perm(E2, E1, 0x05040100u);     // cum[e0+2] | cum[e0+4] << 16
        const u32 dd = __popc((mm - p12) & 0x80008000u);
        u32 Ed = E0, En = E1;
        if (dd >= 1) { Ed = E1; En = E2; }
        if (dd >= 2) { Ed = E2; En = E3; }
        e = 30 * a + 6 * b + 2 * dd;
        c0 = Ed & 0xffffu; c1 = Ed >> 16; c2 = En & 0xffffu;
    } else if (LV == 4) {
        const u32 a = count_le(mm, img.ld64(row)) + count_le(mm, img.ld64(row + 8));
        const u32 b = count_le(mm, img.ld64(row + 16 + 8 * a));
        const u32 dd = count_le(mm, img.ld64(row + 64 + 8 * (5 * a + b)));
        e = 50 * a + 10 * b + 2 * dd;
        const u32 lo = row + 304u + 2 * e;
        const u32 c01 = img.ld32(lo), c23 = img.ld32(lo + 4);
        c0 = c01 & 0xffffu; c1 = c01 >> 16; c2 = c23 & 0xffffu;
    } else {
        // root, then the whole group of ten: dwords E_k = cum[10b+2k] | cum[10b+2k+1] << 16, k = 0..5
        const u32 b = count_le(mm, root ? *root : img.ld64(row));
        const u32 g = row + 8 + 20 * b;
        const u32 E0 = img.ld32(g), E1 = img.ld32(g + 4), E2 = img.ld32(g + 8), E3 = img.ld32(g + 12),
                  E4 = img.ld32(g + 16), E5 = img.ld32(g + 20);
        // even-ranked entries 2,4,6,8 of the group are its inner separators
        const u32 p01 = __builtin_amdgcn_perm(E2, E1, 0x05040100u);     // lo16(E1) | lo16(E2) << 16
        const u32 p23 = __builtin_amdgcn_perm(E4, E3, 0x05040100u);
        const u32 dd = __popc(__builtin_amdgcn_perm(mm - p23, mm - p01, 0x07050301u) & 0x80808080u);
        e = 10 * b + 2 * dd;
        if (spec) {
            spec->rowE = spec->rows + __umul24(e, spec->roww);
            spec->ra = img.ld64_now(spec->rowE);
            spec->rb = img.ld64_now(spec->rowE + spec->roww);
            __builtin_amdgcn_sched_barrier(0);      // the select chain below must not be scheduled ahead of these reads
        }
        u32 Ed = E0, En = E1;
        if (dd >= 1) { Ed = E1; En = E2; }
        if (dd >= 2) { Ed = E2; En = E3; }
        if (dd >= 3) { Ed = E3; En = E4; }
        if (dd >= 4) { Ed = E4; En = E5; }
        c0 = Ed & 0xffffu; c1 = Ed >> 16; c2 = En & 0xffffu;
    }
    const bool up = m >= c1;
    if (spec) spec->up = up;
    const u32 start = up ? c1 : c0;
    const u32 next = up ? c2 : c1;
    x = __umul24(next - start, x >> look) + (m - start);   // freq <= 2^15, x>>look < 2^22: exact mod 2^32
    asm("" : "+v"(e));                        // (keeps e + carry an add-with-carry: the compiler would otherwise
    return e + (up ? 1u : 0u);                  //  turn it into a select of 0/1 and an or, e being even)
}

// One lookup + state update on a packed row (r4x16_common.h, "level 1": 10-bit tables, <= 48 symbols).
// `row` is the LDS address of the context's row, `root` its first dword (read a step ahead), `first` the index of
// the row's first symbol of non-zero frequency.  Fields sit at bits 0, 11 and 22 of a dword with zero bits at 10 and
// 21: adding GM = guards - (m | m << 11 | m << 22) leaves guard bit i set exactly when field i >= m (a field plus
// 1024 minus m stays inside its eleven bits), so two "field < m" tests cost one add, one and, one popcount.
// Returns the compact symbol index; x becomes freq * (x >> 10) + m - start.
// WIDE: 49..96 symbols, root = eight u16 separators L[12 k] + 1 (r4x16_common.h), counted like the u16 rows' roots.
template <bool WIDE>
__device__ __forceinline__ u32 lookup_step_pk(u32 row, u32x2 rootv, u32x2 rootw, u32 first, u32 &x)
{
    const LImg img0{0u};
    const u32 GB = 0x00200400u;
    const u32 m = x & 1023u;
    const u32 GM = GB - __umul24(m, 0x400801u);
    u32 g, ga;
    if (WIDE) {
        const u32 mm = __umul24(m, 0x10001u) + 0x80008000u;
        g = count_le(mm, rootv) + count_le(mm, rootw);        // #{k : L[12 k] < m}
        ga = row + 16u + 16u * g;
    } else {
        // group of twelve: #{L[12], L[24], L[36]} below m
        const u32 root = rootv.x;
        const u32 gneg = __popc((root + GM) & GB);            // 2 - #{L[12], L[24] < m}
        const bool r3 = root < (m << 22);                     // L[36] < m: the top field needs no guard
        g = (r3 ? 3u : 2u) - gneg;
        ga = row + 16u * g;
    }
    u32 D0, D1, D2, D3, D4;
    if (WIDE) { D0 = img0.ld32(ga); D1 = img0.ld32(ga + 4); D2 = img0.ld32(ga + 8); D3 = img0.ld32(ga + 12); D4 = img0.ld32(ga + 16); }
    else {
        const u32x4 Dq = *(LAS const u32x4 *)(unsigned long)ga;          // the group: one 16-byte read (rows are 16-byte aligned)
        D0 = Dq.x; D1 = Dq.y; D2 = Dq.z; D3 = Dq.w; D4 = img0.ld32(ga + 16);
    }
    // dword of three.  A leaf dword holds L[3i + 1], L[3i + 2] in its guarded low fields and L[3i] in its TOP field, so
    // the pivots L[12g + 3], L[12g + 6], L[12g + 9] are tested without extraction: top field < m  <=>  dword < m << 22.
    const u32 m22 = m << 22;
    const bool s0 = D1 < m22, s1 = D2 < m22, s2 = D3 < m22;
    u32 D = D0, Dn = D1;
    if (s0) { D = D1; Dn = D2; }
    if (s1) { D = D2; Dn = D3; }
    if (s2) { D = D3; Dn = D4; }
    const u32 q = (u32)s0 + (u32)s1 + (u32)s2;
    // inside the dword: the two low fields against m by their guard bits
    const u32 rneg = __popc((D + GM) & GB);                   // 2 - r,  r = #{L[3q + 1], L[3q + 2] below m}
    const u32 r11 = 22u - 11u * rneg;
    // the candidates in stride-11 order: P = L[3q], L[3q + 1], L[3q + 2];  C = L[3q + 1], L[3q + 2], L[3q + 3]
    const u32 P = (D << 11) | (D >> 22);                      // (not a rotation: bit 21 is a guard bit, the top field must land on bit 0)
    const u32 Cc = (D & 0x003fffffu) | (Dn & 0xffc00000u);    // (one v_bfi_b32)
    const u32 prev = __builtin_amdgcn_ubfe(P, r11, 10);       // L[c]: end of the symbol before (1023 stands for -1)
    const u32 cur = __builtin_amdgcn_ubfe(Cc, r11, 10);       // L[c + 1]: end of this symbol
    const u32 np = ~prev;
    const u32 fm1 = (cur + np) & 1023u;                       // freq - 1
    const u32 off = (m + np) & 1023u;                         // m - start
    const u32 xs = x >> 10;
    x = __umul24(fm1, xs) + xs + off;                         // freq <= 1024, x >> 10 < 2^22: exact mod 2^32
    // `first` arrives with the + 2 already in it (the image stores first + 2): two multiply-adds and a subtraction
    // instead of two multiplies, a three-way add, a subtraction and an add
    u32 t;
    asm("v_mad_u32_u24 %0, %1, 12, %2" : "=v"(t) : "v"(g), "v"(first));
    asm("v_mad_u32_u24 %0, %1, 3, %2" : "=v"(t) : "v"(q), "v"(t));
    return t - rneg;
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
template <int ORDER, int LV, class IMG>
__device__ __forceinline__ u32 chain_decode(IMG img, u32 nsym, gcu8 *words, u32 words_len, gu8 *out,
                                            u32 out_sz, u32 x, u32 look, bool active, u32 lane)
{
    const u32 k = lane & 3;
    const u32 mask = (1u << look) - 1;
    const u32 nwords = words_len >> 1;
    const u32 rows = img_alpha_bytes(nsym), roww = img_row_bytes(nsym);
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

    u32 row = rows, cursor = 0, bad = 0, t = 0;
    if (ORDER == 1 && count) bad = img.ld16(0) & ROW_EMPTY;

    while (wave_any(t < count)) {
        const bool live = t < count;
        bool want = false;
        if (live) {
            const u32 s = lookup_step<LV>(img, row, look, mask, x);
            const u32 al = img.ld16(2 * s);
            if (ORDER == 0) {
                out[pos] = (u8)al;
                pos += 4;
            } else {
                row = rows + s * roww;
                if (t + 1 < count) bad |= al & ROW_EMPTY;
                out[pos] = (u8)al;
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
#ifndef PK_MIN_NSYM
#define PK_MIN_NSYM 13u      // smaller alphabets keep the u16 rows: their streams' LDS is mostly the word ring anyway
#endif
#define RING_BYTES 272u      // 256-byte ring (four 64-byte quarters) + 8-byte mirror of its head (+8 pad)
#define TRIP_STEPS 8         // steps per loop trip: at most 64 bytes of words, one quarter of the ring

typedef u32 GAS __attribute__((aligned(1))) gu32_unaligned;   // global dword store at any byte address

// TRIP: steps per loop trip.  8 with the 256-byte ring of four 64-byte quarters (three live, one being refilled: a trip
// moves the cursor by up to 64 bytes and reads 12 ahead).  4 with a 128-byte ring of TWO quarters, both live, the next
// one waiting in registers: a trip then moves the cursor by at most 32 bytes, so from anywhere in quarter h it stays
// inside h and h + 1, and the quarter just left is overwritten at once.  136 bytes less per stream: 46-symbol packed
// rows at 3,360 bytes, 3 x 16 streams per CU instead of 3 x 15 - for one more loop test per eight steps.
#define ROW_BAD  0x200u      // alpha[] flag (rANS 4x8 images): decoding this symbol is an error
#define RING_BYTES_SHORT 136u
// BYTE (round 4): rANS 4x8's renormalisation on the same loop - L = 2^23, a byte if x < 2^23 and a second one if x < 2^15,
// chains served in the order 0..3 (rANS_byte.h:541-551); `cursor` and `words_len` then count bytes, the look-up is the
// 12-bit one of the u16 rows, and the flag of the symbol that owns slot 4095 of a 4095-sum table (ROW_BAD) is kept.
template <int ORDER, int LV, int TRIP = TRIP_STEPS, bool BYTE = false>
__device__ __forceinline__ u32 chain_decode_lds(const u8 *img_lds, u32 nsym, u8 *ring, gcu8 *words, u32 words_len,
                                                gu8 *out, u32 out_sz, u32 x, u32 look, bool active, u32 lane)
{
    const LImg img{lds_addr(img_lds)};                     // alpha[] reads
    const LImg img0{0u};                                   // row reads: `row` is an absolute LDS address
    const u32 k = lane & 3;
    const u32 mask = (1u << look) - 1;
    constexpr bool PKD = LV == 1 || LV == 5, WIDE = LV == 5;       // packed rows (r4x16_common.h), with the 16-byte root
    static_assert(LV != 6 && LV != 10, "direct blocks and mid rows have their own loops: chain_decode_dir, chain_decode_mid");
    constexpr bool PK2 = LV == 1;                                  // layout 2 of the packed rows: 8-byte head entries (root, alpha word)
    const u32 rows = lds_addr(img_lds) + (PKD ? pk_head_bytes(nsym) : img_alpha_bytes(nsym)), roww = PKD ? pk_row_bytes(nsym) : img_row_bytes(nsym);
    const u32 nwords = BYTE ? words_len : words_len >> 1;  // units of the cursor: 16-bit words (bytes for rANS 4x8)
    const u32 below = (1u << k) - 1u;                      // quad lanes below this one
    constexpr u32 NQ = TRIP == 8 ? 4u : 2u, RB = 64u * NQ; // quarters and bytes of the ring
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
    // 16-byte chunk c of the stream; past the input the last chunk repeats (those ring bytes are never
    // taken: the fast trips stop 16 words before the end and the others check every word against nwords)
    const u32 lastc = avail ? (avail - 1u) >> 4 : 0u;
    const bool loadable = active && avail != 0;
    auto load_chunk = [&](u32 c) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (loadable) v = *(gcu32x4 *)(abase + 16ull * (c < lastc ? c : lastc));
        return v;
    };
    // Quarters h, h+1, h+2 of the stream are in the ring while the cursor is in quarter h (a trip moves it
    // by at most 64 bytes and reads 12 bytes ahead); quarter h+3 waits in `pend` for the next crossing.
    if (active) {
        const u32x4 c0 = load_chunk(k), c1 = load_chunk(k + 4);
        *(u32x4 *)(ring + 16 * k) = c0;
        *(u32x4 *)(ring + 64 + 16 * k) = c1;
        if (NQ == 4) *(u32x4 *)(ring + 128 + 16 * k) = load_chunk(k + 8);
        if (k == 0) *(u32x2 *)(ring + RB) = c0.xy;
    }
    u32x4 pend = load_chunk((NQ == 4 ? 12 : 8) + k);
    u32 half = 0;                                         // index of the 64-byte quarter holding the cursor
    __syncthreads();

    u32 row = rows, cursor = 0, bad = 0, t = 0;
    // root separators of `row`, read as soon as the row is known (one step ahead of their use,
    // so that this LDS round trip runs beside the renormalisation instead of after it)
    u32x2 root = PK2 ? u32x2{img.ld32(0), 0u} : LV == 1 ? u32x2{img0.ld32(row), 0u} : img0.ld64(row);
    u32x2 root2 = WIDE ? img0.ld64(row + 8) : u32x2{0u, 0u};
    u32 acc = 0;                                          // order-1: the last (up to) 4 decoded bytes
    u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;                   // order-1: completed dwords not yet stored (a3 newest)
    u32 hdr = 0, hdr_even = 0;                            // order-1: alpha[] word of the symbol decoded last step
    u32 badb = 0;                                         // BYTE: the alpha words of every symbol decoded (for ROW_BAD)
    if (ORDER == 1 && count) bad = img.ld16(PK2 ? 4 : 0);
    if (PKD) hdr = img.ld16(PK2 ? 4 : 0);                       // packed rows: bits 9.. of the context's alpha word = its `first`

    // Four steps per trip: one loop test, one store and one ring check per trip.  A trip in which
    // every stream of the wave is still running on all four chains and has at least 16 words left
    // takes the FAST body: no per-lane liveness selects and no end-of-stream test.
    auto trip = [&](auto fastc) {
        constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
        for (int u = 0; u < TRIP; u++) {
            const u32 T = t + (u32)u;                     // index of this step
            const bool live = FAST ? true : T < count;
            // next four candidate words (8 bytes at any byte alignment) from the ring; issued
            // before the table lookups so that their latency hides under them
            const u32 cb = BYTE ? off0 + cursor : off0 + 2 * cursor;
            const u32 ra = cb & (RB - 4u);
            // Three ALIGNED dwords and a funnel shift: a dword read at a misaligned LDS address costs far
            // more than the extra read (measured: +12 % on the whole kernel per misaligned read and step).
            // (volatile: keeps the compiler from sinking these reads into a branch, which would put their
            //  latency back on the dependent path; explicit LDS pointer: a volatile generic access goes FLAT)
            const u32x2 d01 = *(LAS const volatile u32x2_a4 *)(ring + ra);     // one ds_read2_b32
            const u32 d0 = d01.x, d1 = d01.y, d2 = *(lvcu32 *)(ring + ra + 8);

            u32 xn = x;
            RootSpec spec;
            spec.rows = rows; spec.roww = roww;
            const bool speculate = ORDER == 1 && LV == 2;
            u32 s, rown1 = 0;
            u32 hn_first = 0;
            u32x2 rootn1 = {0u, 0u}, rootn2 = {0u, 0u};
            if (PKD) {
                s = lookup_step_pk<WIDE>(row, root, root2, hdr >> PK_FIRST_SHIFT, xn);
                // the next row's root: requested as soon as the symbol is known, used at the top of the next step
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(rown1) : "v"(s), "v"(roww), "v"(rows));   // (the compiler's own choice is a 64-bit multiply-add)
                if (WIDE) {
                    rootn1 = *(LAS const volatile u32x2_a4 *)(unsigned long)rown1;
                    rootn2 = *(LAS const volatile u32x2_a4 *)(unsigned long)(rown1 + 8u);
                } else if (PK2) {
                    // the head entry of the symbol just decoded: the root of ITS row and its alpha word, one 8-byte read
                    const u32x2 h = *(LAS const volatile u32x2 *)(unsigned long)(lds_addr(img_lds) + 8u * s);
                    rootn1.x = h.x; hn_first = h.y;
                } else
                    rootn1.x = *(LAS const volatile u32 *)(unsigned long)rown1;
            } else {
                s = lookup_step<(PKD ? 2 : LV)>(img0, row, look, mask, xn, LV == 2 ? &root : nullptr, speculate ? &spec : nullptr);
            }
            const u32 hn = PK2 ? hn_first : img.ld16(2 * s);   // byte value | ROW_EMPTY of the new context
            if (BYTE) badb |= (FAST || live) ? hn : 0u;   // rANS 4x8: ROW_BAD on ANY decoded symbol, the last one included
            u32 byte0 = 0;
            if (ORDER == 0) {
                byte0 = hn & 0xffu;
            } else {
                // the byte of the symbol decoded one step ago and the flags of the row in use now
                if (u > 0 || t > 0) {
                    // ROW_EMPTY bit is tested after the loop; the fast body folds two steps into one three-way or
                    if (!FAST) bad |= live ? hdr : 0u;
                    else if (u & 1) bad |= hdr | hdr_even;
                    else hdr_even = hdr;
                    acc = (FAST || T <= count) ? __builtin_amdgcn_alignbit(hdr, acc, 8) : acc;
                }
                // symbols t-4 .. t-1 are now in acc, oldest in the low byte: queue the dword, and every
                // fourth trip store 16 bytes at once (4-byte stores to 16 different lines per wave made
                // the L2 write each line to HBM 3.5 times over; see profiles/)
                if ((u & 3) == 0 && T >= 4 && (FAST || T <= count)) {
                    a0 = a1; a1 = a2; a2 = a3; a3 = acc;
                    if (u == 0 && (t & 15u) == 0 && active) {       // t is a multiple of the trip: T % 16 == 0 only at u == 0
                        const u32x4 v = {a0, a1, a2, a3};
                        *(GAS u32x4_unaligned *)op = v;
                        op += 16;
                    }
                }
                hdr = live ? hn : hdr;
                if (PKD) {
                    row = live ? rown1 : row;
                    root.x = live ? rootn1.x : root.x;
                    if (WIDE) {
                        root.y = live ? rootn1.y : root.y;
                        root2.x = live ? rootn2.x : root2.x;
                        root2.y = live ? rootn2.y : root2.y;
                    }
                } else if (speculate) {
                    const u32 rown = spec.rowE + (spec.up ? roww : 0u);
                    const u32x2 rootn = {spec.up ? spec.rb.x : spec.ra.x, spec.up ? spec.rb.y : spec.ra.y};
                    row = live ? rown : row;
                    root.x = live ? rootn.x : root.x;
                    root.y = live ? rootn.y : root.y;
                } else {
                    const u32 rown = rows + __umul24(s, roww);
                    row = live ? rown : row;
                }
            }
            x = live ? xn : x;
            if (BYTE) {
                // a chain takes a byte if x < 2^23 and a second one if x < 2^15 (then x << 8 | b < 2^23 whatever b is)
                const bool w1 = live && x < (1u << 23), w2 = live && x < (1u << 15);
                const u32 m1 = quad_ballot(w1, lane), m2 = quad_ballot(w2, lane);
                const u32 pre = __popc(m1 & below) + __popc(m2 & below);            // bytes the chains before this one take: 0 .. 6
                u32 wlo, whi;
                if (ORDER == 1) { wlo = __builtin_amdgcn_alignbyte(d1, d0, cb); whi = __builtin_amdgcn_alignbyte(d2, d1, cb); }
                else { const u32 sh = (cb & 3u) * 8u; wlo = __builtin_amdgcn_alignbit(d1, d0, sh); whi = __builtin_amdgcn_alignbit(d2, d1, sh); }
                // the two bytes at `pre`, the first one on top: b0 << 8 | b1
                const u32 w = __builtin_amdgcn_perm(whi, wlo, __umul24(pre, 0x0101u) + 0x0c0c0001u);
                u32 x2 = (x << 16) | w, x1 = (x << 8) | (w >> 8);
                asm volatile("" : "+v"(x2), "+v"(x1));
                if (FAST) x = w2 ? x2 : w1 ? x1 : x;
                else {
                    // nothing is read past the end of the stream: a chain gets what is left, up to what it asks for
                    const u32 at = cursor + pre, room = at < nwords ? nwords - at : 0u, wantb = (w1 ? 1u : 0u) + (w2 ? 1u : 0u);
                    const u32 takeb = wantb < room ? wantb : room;
                    x = takeb > 1u ? x2 : takeb ? x1 : x;
                }
                cursor += __popc(m1) + __popc(m2);
            } else {
            const bool want = live && x < RANS_LOW;

            // renormalise: chains refill in order 0..3 from the shared cursor.  After the first
            // refusal (stream exhausted) no later request can succeed either, so the cursor may
            // simply advance by the number of requests (rANS_word.h:402-410).
            const u32 wm = quad_ballot(want, lane);
            const u32 pre = __popc(wm & below);
            const bool take = FAST ? want : (want && cursor + pre < nwords);
            // (byte shift = cb & 3.  Measured per loop: v_alignbyte saves the shift-amount instruction in the order-1 loop,
            //  -0.7 %, but the order-0 loop schedules 11 % worse with it - so each keeps the form that is faster)
            u32 wlo, whi;
            if (ORDER == 1) { wlo = __builtin_amdgcn_alignbyte(d1, d0, cb); whi = __builtin_amdgcn_alignbyte(d2, d1, cb); }
            else { const u32 sh = (cb & 3u) * 8u; wlo = __builtin_amdgcn_alignbit(d1, d0, sh); whi = __builtin_amdgcn_alignbit(d2, d1, sh); }
            // word `pre` of the four candidates: one byte permute over the 8 bytes {whi:wlo} with a per-lane
            // selector (bytes 2 pre and 2 pre + 1, then two zero bytes)
            const u32 w = __builtin_amdgcn_perm(whi, wlo, __umul24(pre, 0x0202u) + 0x0c0c0100u);
            u32 xr = (x << 16) | w;
            asm volatile("" : "+v"(xr));                  // keeps the refill arithmetic out of a branch
            x = take ? xr : x;
            cursor += __popc(wm);
            }

            if (ORDER == 0) {
                if (FAST) {
                    // Step T puts byte 4T + k on chain k.  Each lane gathers its own four bytes of four
                    // steps, then the quad transposes the 4 x 4 byte block (four DPP broadcasts, three byte
                    // permutes with a per-lane selector) so that lane j holds the dword of step T0 + j:
                    // one 4-byte store per lane, 16 contiguous bytes per stream, every fourth step.
                    acc = __builtin_amdgcn_alignbit(byte0, acc, 8);
                    if ((u & 3) == 3) {
                        const u32 A0 = quad_bcast0(acc), A1 = quad_bcast1(acc), A2 = quad_bcast2(acc), A3 = quad_bcast3(acc);
                        const u32 sel = k | ((4u + k) << 8);                        // byte k of either source
                        const u32 p01 = __builtin_amdgcn_perm(A1, A0, sel), p23 = __builtin_amdgcn_perm(A3, A2, sel);
                        const u32 dw = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                        if (active) *(gu32_unaligned *)(out + 4 * (u64)(T - 3u + k)) = dw;
                    }
                } else {
                    // the quad's four bytes are consecutive: lane 0 stores them as one dword
                    const u32 b1 = quad_bcast1(byte0), b2 = quad_bcast2(byte0), b3 = quad_bcast3(byte0);
                    const u32 l3 = quad_bcast3(live ? 1u : 0u);
                    const u32 dw = byte0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                    if (l3) { if (k == 0 && active) *(gu32_unaligned *)(op + 4 * (u64)T) = dw; }
                    else if (live) op[4 * (u64)T] = (u8)byte0;
                }
            }
        }
        t += TRIP;
    };
    while (wave_any(t < count)) {
        const bool slow = active && (t + TRIP > count || cursor + (BYTE ? 8 : 4) * TRIP > nwords);
        if (!wave_any(slow)) trip(std::true_type{});
        else trip(std::false_type{});

        // ring refill when the cursor has entered a new 64-byte quarter (at most 64 bytes ago)
        const u32 nh = (BYTE ? off0 + cursor : off0 + 2 * cursor) >> 6;
        if (wave_any(active && nh != half)) {
            if (active && nh != half) {                   // idle lanes hold garbage cursors: they must not write LDS
                // the quarter in `pend` was requested at the previous crossing: park it in the slot just vacated
                const u32 slot = (NQ == 4 ? ((nh + 2) & 3u) : ((nh + 1) & 1u)) * 64u + 16u * k;
                *(u32x4 *)(ring + slot) = pend;
                if (slot == 0) *(u32x2 *)(ring + RB) = pend.xy;
                pend = load_chunk(4 * (nh + (NQ == 4 ? 3 : 2)) + k);
                half = nh;
            }
            // (the chain kernels' workgroups are ONE wave, whose LDS operations execute in the order it issues them: the
            //  quarter just written is seen by the reads that follow without a barrier - and without the barrier's wait for
            //  everything in flight, the next step's look-ups included: 92.2 -> 92.0 ms)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (ORDER == 1 && count) {
        // t steps ran.  A chain whose count equals t still has its last byte in hdr and its last
        // four symbols unstored; any other chain has count%4 symbols left, already in acc.
        // dwords were queued at steps 4, 8, .. <= min(count, t-4) and stored 16 bytes at a time
        const u32 lastq = count < t - 4 ? count : t - 4;
        const u32 pushed = lastq >> 2, nd = pushed & 3u;   // nd queued dwords are still in a(4-nd)..a3
        if (count == t) acc = __builtin_amdgcn_alignbit(hdr, acc, 8);
        if (nd == 3) { *(gu32_unaligned *)op = a1; op += 4; }
        if (nd >= 2) { *(gu32_unaligned *)op = a2; op += 4; }
        if (nd >= 1) { *(gu32_unaligned *)op = a3; op += 4; }
        const u32 rem = count - 4 * pushed;                // 0..4, in the top `rem` bytes of acc
        for (u32 j = 0; j < rem; j++) op[j] = (u8)(acc >> (8 * (4 - rem + j)));
    }
    return active ? ((bad & ROW_EMPTY) | (BYTE ? badb & ROW_BAD : 0u)) : 0u;        // idle lanes ran on garbage in the fast trips
}

// ---------------------------------------------------------------------------------------------
// The chain decoder for direct blocks (r4x16_common.h, "level 6"): the short step.  Same ring, same renormalisation
// and the same output gathering as chain_decode_lds, but the step is laid out around its two dependent LDS reads -
// a lone wave issues in order, so whatever does not depend on a read has to stand IN ITS SHADOW, between the read
// and the first use of its result, or it costs its four cycles on top of the latency:
//     x -> slot pair -> [read: rank]            shadow: m, x >> look, ring address, ring reads, last step's byte
//       -> entry address -> [read: two entries]  shadow: the candidate words out of the ring dwords
//       -> pick, state update, next block       -> renormalise
// Measured against the same rows inside chain_decode_lds' step (ring reads first): see DESIGN 6.
// ---------------------------------------------------------------------------------------------
template <int ORDER, bool AFF>
__device__ __forceinline__ u32 chain_decode_dir(const u8 *img_lds, u32 nsym, u8 *ring, gcu8 *words, u32 words_len,
                                                gu8 *out, u32 out_sz, u32 x, u32 look, u32 aff, bool active, u32 lane)
{
    const u32 imga = lds_addr(img_lds);
    const u32 k = lane & 3;
    const u32 lookm1 = look - 1u;
    const u32 sh = 32u - look, lowmask = (1u << sh) - 1u;
    const u32 rows = imga + img_alpha_bytes(nsym), roww = dir_blk_bytes(nsym, look);
    const u32 fbb = dir_fb_bytes(nsym);
    const u32 nwords = words_len >> 1;
    const u32 below = (1u << k) - 1u;
    const u32 ringa = lds_addr(ring);
    const u32 c4 = __umul24(aff - 1u, 0x010101u) + ((aff - 1u) << 24);      // AFF: the alphabet's offset in every byte
    u32 count;
    gu8 *op;
    if (ORDER == 0) {
        count = (out_sz + 3 - k) >> 2;
        op = out + k;
    } else {
        const u32 q = out_sz >> 2;
        count = q + (k == 3 ? out_sz - 4 * q : 0);
        op = out + (u64)k * q;
    }
    if (!active) count = 0;

    // ---- word ring: as in chain_decode_lds ----------------------------------------------------
    gcu8 *abase = (gcu8 *)((u64)words & ~15ull);
    const u32 off0 = (u32)((u64)words & 15ull);
    const u32 avail = off0 + words_len;
    const u32 lastc = avail ? (avail - 1u) >> 4 : 0u;
    const bool loadable = active && avail != 0;
    auto load_chunk = [&](u32 c) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (loadable) v = *(gcu32x4 *)(abase + 16ull * (c < lastc ? c : lastc));
        return v;
    };
    if (active) {
        const u32x4 c0 = load_chunk(k), c1 = load_chunk(k + 4), c2 = load_chunk(k + 8);
        *(u32x4 *)(ring + 16 * k) = c0;
        *(u32x4 *)(ring + 64 + 16 * k) = c1;
        *(u32x4 *)(ring + 128 + 16 * k) = c2;
        if (k == 0) *(u32x2 *)(ring + 256) = c0.xy;
    }
    u32x4 pend = load_chunk(12 + k);
    u32 half = 0;
    __syncthreads();

    u32 row = rows, cursor = 0, bad = 0, t = 0;
    // acc gathers one byte per step: the output byte (order 1, !AFF: of the step before) or, AFF, the low byte of D
    // (compact index + one bit of the frequency, masked off when four of them become four bytes)
    u32 acc = 0;
    u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    u32 hdr = 0, hdr_even = 0;                             // !AFF: alpha[] word of the symbol decoded last step;  AFF: D of the last step
    if (ORDER == 1 && !AFF && count) bad = *(LAS const u16 *)(unsigned long)imga;
    auto to_bytes = [&](u32 a) -> u32 { return AFF ? (a & 0x7f7f7f7fu) + c4 : a; };

    u32 oq0 = 0, oq1 = 0, oq_t = 0;                         // order 0: the dwords of the last full trip, not yet stored
    bool oq_have = false;
    auto oq_flush = [&]() {
        if (ORDER == 0 && oq_have) {
            if (active) {
                *(gu32_unaligned *)(out + 4 * (u64)(oq_t + k)) = oq0;
                *(gu32_unaligned *)(out + 4 * (u64)(oq_t + 4u + k)) = oq1;
            }
            oq_have = false;
        }
    };
    auto trip = [&](auto fastc) {
        constexpr bool FAST = decltype(fastc)::value;
        oq_flush();
#pragma unroll
        for (int u = 0; u < TRIP_STEPS; u++) {
            const u32 T = t + (u32)u;
            const bool live = FAST ? true : T < count;
            // (1) the head of the dependent chain: rank of the owner of the even slot
            const u32 j = __builtin_amdgcn_ubfe(x, 1u, lookm1);
            const u32 rk = *(LAS const volatile u8 *)(unsigned long)(row + fbb + j);
            __builtin_amdgcn_sched_barrier(0);
            // (2) in its shadow: M, x >> look, the ring dwords, last step's byte
            const u32 M = (x << sh) | lowmask;
            const u32 xs = x >> look;
            const u32 cb = off0 + 2 * cursor;
            const u32 ra = ringa + (cb & 252u);
            const u32x2 d01 = *(LAS const volatile u32x2_a4 *)(unsigned long)ra;
            const u32 d0 = d01.x, d1 = d01.y, d2 = *(LAS const volatile u32 *)(unsigned long)(ra + 8u);
            if (ORDER == 1) {
                if (!AFF && (u > 0 || t > 0)) {
                    if (!FAST) bad |= live ? hdr : 0u;
                    else if (u & 1) bad |= hdr | hdr_even;
                    else hdr_even = hdr;
                    acc = (FAST || T <= count) ? __builtin_amdgcn_alignbit(hdr, acc, 8) : acc;
                }
                // AFF: acc took the symbol of step T - 1 in that step; either way symbols T-4 .. T-1 are in acc now
                if ((u & 3) == 0 && T >= 4 && (FAST || T <= count)) {
                    a0 = a1; a1 = a2; a2 = a3; a3 = to_bytes(acc);
                    if (u == 0 && (t & 15u) == 0 && active) {
                        const u32x4 v = {a0, a1, a2, a3};
                        *(GAS u32x4_unaligned *)op = v;
                        op += 16;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (3) the entry of that symbol and of the next one with a frequency
            const u32x2 e = *(LAS const volatile u32x2_a4 *)(unsigned long)(row + 4u * rk);
            __builtin_amdgcn_sched_barrier(0);
            // (4) in its shadow: the four candidate words
            const u32 wlo = __builtin_amdgcn_alignbyte(d1, d0, cb), whi = __builtin_amdgcn_alignbyte(d2, d1, cb);
            __builtin_amdgcn_sched_barrier(0);
            // (5) pick (r4x16_common.h: the smaller difference), state update, next block
            const u32 Da = M - e.x, Db = M - e.y;
            const u32 D = Da < Db ? Da : Db;
            const u32 xn = __umul24(__builtin_amdgcn_ubfe(D, 7u, 13u), xs) + (D >> sh);      // freq <= 4096, x >> look < 2^22: exact mod 2^32
            const u32 s = D & 127u;
            u32 rown1 = row;
            if (ORDER == 1) asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(rown1) : "v"(s), "v"(roww), "v"(rows));
            u32 byte0 = 0;
            if (AFF) {
                if (ORDER == 1) {
                    if (!FAST) bad |= live ? D : 0u;
                    else if (u & 1) bad |= D | hdr_even;
                    else hdr_even = D;
                    acc = live ? __builtin_amdgcn_alignbit(D, acc, 8) : acc;
                    row = live ? rown1 : row;
                } else byte0 = D;
            } else {
                const u32 hn = *(LAS const volatile u16 *)(unsigned long)(imga + 2u * s);  // byte | ROW_EMPTY: looked at in the next step
                if (ORDER == 0) byte0 = hn & 0xffu;
                else {
                    hdr = live ? hn : hdr;
                    row = live ? rown1 : row;
                }
            }
            x = live ? xn : x;
            // (6) renormalise: chains refill in order 0..3 from the shared cursor (see chain_decode_lds)
            const bool want = live && x < RANS_LOW;
            const u32 wm = quad_ballot(want, lane);
            const u32 pre = __popc(wm & below);
            const bool take = FAST ? want : (want && cursor + pre < nwords);
            const u32 w = __builtin_amdgcn_perm(whi, wlo, __umul24(pre, 0x0202u) + 0x0c0c0100u);
            u32 xr = (x << 16) | w;
            asm volatile("" : "+v"(xr));
            x = take ? xr : x;
            cursor += __popc(wm);

            if (ORDER == 0) {
                if (FAST) {
                    acc = __builtin_amdgcn_alignbit(byte0, acc, 8);
                    if ((u & 3) == 3) {
                        const u32 ab = to_bytes(acc);
                        const u32 A0 = quad_bcast0(ab), A1 = quad_bcast1(ab), A2 = quad_bcast2(ab), A3 = quad_bcast3(ab);
                        const u32 sel = k | ((4u + k) << 8);
                        const u32 p01 = __builtin_amdgcn_perm(A1, A0, sel), p23 = __builtin_amdgcn_perm(A3, A2, sel);
                        const u32 dw = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                        // (held back until the top of the next trip: a store issued here is what the ring refill behind
                        //  this trip would wait for - loads and stores share the counter - at the price of its whole
                        //  round trip: order-0 streams took ~590 cycles per step on this loop, order-1 ones 372)
                        if (u < 4) oq0 = dw; else oq1 = dw;
                    }
                } else {
                    if (AFF) byte0 = ((byte0 & 0x7fu) + c4) & 0xffu;
                    const u32 b1 = quad_bcast1(byte0), b2 = quad_bcast2(byte0), b3 = quad_bcast3(byte0);
                    const u32 l3 = quad_bcast3(live ? 1u : 0u);
                    const u32 dw = byte0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                    if (l3) { if (k == 0 && active) *(gu32_unaligned *)(op + 4 * (u64)T) = dw; }
                    else if (live) op[4 * (u64)T] = (u8)byte0;
                }
            }
        }
        if (ORDER == 0 && FAST) { oq_t = t; oq_have = true; }
        t += TRIP_STEPS;
    };
    static_assert(TRIP_STEPS == 8, "order 0 holds two dwords per trip back");
    while (wave_any(t < count)) {
        const bool slow = active && (t + TRIP_STEPS > count || cursor + 4 * TRIP_STEPS > nwords);
        if (!wave_any(slow)) trip(std::true_type{});
        else trip(std::false_type{});
        const u32 nh = (off0 + 2 * cursor) >> 6;
        if (wave_any(active && nh != half)) {
            if (active && nh != half) {
                const u32 slot = ((nh + 2) & 3u) * 64u + 16u * k;
                *(u32x4 *)(ring + slot) = pend;
                if (slot == 0) *(u32x2 *)(ring + 256) = pend.xy;
                pend = load_chunk(4 * (nh + 3) + k);
                half = nh;
            }
            __syncthreads();
        }
    }
    oq_flush();
    if (ORDER == 1 && count) {
        // t steps ran (see chain_decode_lds).  !AFF: a chain whose count equals t still has its last byte in hdr.
        // AFF: acc already holds every symbol decoded (the live ones only), the newest in its top byte.
        const u32 lastq = count < t - 4 ? count : t - 4;
        const u32 pushed = lastq >> 2, nd = pushed & 3u;
        if (!AFF && count == t) acc = __builtin_amdgcn_alignbit(hdr, acc, 8);
        if (nd == 3) { *(gu32_unaligned *)op = a1; op += 4; }
        if (nd >= 2) { *(gu32_unaligned *)op = a2; op += 4; }
        if (nd >= 1) { *(gu32_unaligned *)op = a3; op += 4; }
        const u32 rem = count - 4 * pushed;                // 0..4, in the top `rem` bytes of acc
        const u32 accb = to_bytes(acc);
        for (u32 jj = 0; jj < rem; jj++) op[jj] = (u8)(accb >> (8 * (4 - rem + jj)));
    }
    if (!active) return 0u;
    return AFF ? ((ORDER == 1 && (bad & DIR_EMPTY)) ? 1u : 0u) : (bad & ROW_EMPTY);
}

// ---------------------------------------------------------------------------------------------
// Table parsing (one lane) and image building (whole wave).
// ---------------------------------------------------------------------------------------------
struct FrontShared {
    u32 F[256];        // frequencies of the row being parsed, by byte value
    u16 cum[264];      // cum[0..n+3] of the row being built (cum[n] = total, then 0xFFFF)
    u8  present[256];  // alphabet of the stream (order-1: F0)
    u8  idx_of[256];   // byte -> compact index        (order-1)
    u8  alpha[256];    // compact index -> byte        (order-1)
    u16 rankof[256];   // compact index -> its rank among the alphabet members that have a row entry, or 0xffff
    u32 Fk[256];       // order-1: frequencies of the row being parsed, by that rank
    u32 np;            // number of ranked members
    u32 first;         // packed rows: index of the row's first symbol of non-zero frequency
    // scalars handed from lane 0 to the wave
    i32 status;
    u32 empty, pos, nsym, bits, look, go;
    u32 R[4];
    u32 words_pos;
};

// rANS_static4x16pr.c:208-255.  The reference has an unchecked fast loop (a do/while, so with three
// or more bytes left the FIRST symbol is accepted even when it is 0) followed by a checked loop
// with the same body; this is the checked body with that entry rule.  Marks present[]; returns
// bytes consumed, 0 on failure.
__device__ __forceinline__ u32 get_alphabet(ByteSrc &s, u32 pos, u32 end, u8 *present)
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

// Turn S.F[] (by byte) into the cumulative starts S.cum[0..n] of one row over the compact alphabet
// S.alpha[0..n).  Lane 0 only.  Mirrors normalise_freq_shift (:168-179) and the checks at
// :538-552 / :985-997.  Returns false on a bad table.
__device__ bool make_cum(FrontShared &S, u32 n, u32 total, u32 bits)
{
    u32 sh = 0;
    if (total != 0 && total != (1u << bits)) {
        u32 size = total;
        while (size < (1u << bits)) { size *= 2; sh++; }
    }
    u32 x = 0;
    for (u32 ci = 0; ci < n; ci++) {
        S.cum[ci] = (u16)x;
        const u32 f = S.F[S.alpha[ci]] << sh;
        if (!f) continue;
        if (f > (1u << bits) - x) return false;
        x += f;
    }
    if (x != (1u << bits)) return false;
    S.cum[n] = (u16)x;                                   // 1 << bits <= 32768
    S.cum[n + 1] = S.cum[n + 2] = S.cum[n + 3] = 0x7fffu;
    return true;
}

// Whole wave: write one row of the search tree (r4x16_common.h) from S.cum.  An empty row
// (context without a table) gets cum[0] = 0 and sentinels: any lookup lands on symbol 0 and the
// stream is failed through the ROW_EMPTY flag of that context.
__device__ void write_row(u8 *rowp, const FrontShared &S, u32 n, bool empty, u32 lane)
{
    const u32 lv = img_levels(n);
    auto C = [&](u32 r) -> u16 {
        if (empty) return r == 0 ? (u16)0 : (u16)0x7fffu;
        return r <= n + 3 ? S.cum[r] : (u16)0x7fffu;
    };
    auto N = [&](u32 r) -> u16 { const u16 v = C(r); return v > 0x7fffu ? (u16)0x7fffu : v; };   // node separator
    u16 *w = (u16 *)rowp;
    if (lv == 2) {
        // 4 root separators, then the cumulative array itself (sentinel-padded to whole groups)
        if (lane < 4) w[lane] = N(10 * (lane + 1));
        const u32 len = img_leaf_len(n);
        for (u32 t = lane; t < len; t += WAVE) w[4 + t] = C(t);
    } else if (lv == 3) {
        if (lane < 24) {
            const u32 t = lane;
            w[t] = t < 4 ? N(30 * (t + 1)) : N(30 * ((t - 4) >> 2) + 6 * (((t - 4) & 3) + 1));
        }
        const u32 len = img_leaf_len(n);
        for (u32 t = lane; t < len; t += WAVE) w[24 + t] = C(t);
    } else {
        for (u32 t = lane; t < 152; t += WAVE) {
            u16 v;
            if (t < 8) v = N(50 * (t + 1));
            else if (t < 32) { const u32 a = (t - 8) >> 2, i = (t - 8) & 3; v = N(50 * a + 10 * (i + 1)); }
            else { const u32 ab = (t - 32) >> 2, i = (t - 32) & 3; v = N(50 * (ab / 5) + 10 * (ab % 5) + 2 * (i + 1)); }
            w[t] = v;
        }
        for (u32 t = lane; t < n + 4; t += WAVE) w[152 + t] = C(t);
    }
}

// Whole wave: one packed row (r4x16_common.h, "level 1") from S.cum / S.first.  An empty row is all 1023: any
// lookup lands on its first symbol and the stream is failed through the ROW_EMPTY flag of that context.
__device__ void write_row_pk(u8 *rowp, u8 *headp, const FrontShared &S, u32 n, bool empty, u32 lane)
{
    const u32 first = S.first;
    auto L = [&](u32 j) -> u32 {
        if (empty || j == 0) return 1023u;
        const u32 idx = first + j;
        return idx <= n ? (u32)S.cum[idx] - 1u : 1023u;       // cum[first + 1 ..] >= 1: `first` has a frequency
    };
    if (n <= PK_MAX_NSYM) {                               // layout 2: the root in the context's head entry, the row = leaf dwords only
        if (lane == 0) ((u32 *)headp)[0] = L(12) | (L(24) << 11) | (L(36) << 22);
        if (lane < pk_row_bytes(n) / 4u) { const u32 i = 3u * lane; ((u32 *)rowp)[lane] = L(i + 1) | (L(i + 2) << 11) | (L(i) << 22); }
        return;
    }
    const u32 ndw = pk_row_bytes(n) / 4u, rdw = pk_root_bytes(n) / 4u;
    if (lane < ndw) {
        u32 v;
        if (lane < rdw) {
            if (rdw == 1) v = L(12) | (L(24) << 11) | (L(36) << 22);
            else {
                // wide root: u16 separators L[12 k] + 1 for k = 2 lane + 1, 2 lane + 2; beyond the last group: never <= m
                auto sep = [&](u32 k) -> u32 { return (!empty && first + 12u * k <= n) ? L(12u * k) + 1u : 0x7fffu; };
                v = sep(2u * lane + 1u) | (sep(2u * lane + 2u) << 16);
            }
        } else { const u32 i = 3u * (lane - rdw); v = L(i + 1) | (L(i + 2) << 11) | (L(i) << 22); }     // L[3i] on top: see lookup_step_pk
        ((u32 *)rowp)[lane] = v;
    }
}

// Whole wave: one mid row (r4x16_common.h, "level 10") from S.cum: lane j writes the entry of bucket j (slots 16 j ..
// 16 j + 15), all lanes the cumulative array.  owner(m) = #{t in 1 .. n - 1 : cum[t] <= m}: cum is non-decreasing, so
// this is the last symbol that starts at or below m, and among symbols of equal start the last one - the one with a
// frequency.  An empty row (context without a table): every slot owned by symbol 0 with the whole range; the stream is
// failed through the ROW_EMPTY flag of the context, as with the other row kinds.
__device__ void write_row_mid(u8 *rowp, const FrontShared &S, u32 n, bool empty, u32 lane)
{
    u16 *cw = (u16 *)(rowp + 64);
    const u32 len = mid_cum_len(n);
    for (u32 t = lane; t < len; t += WAVE) cw[t] = empty ? (t == 0 ? (u16)0 : t == 1 ? (u16)1024 : (u16)0x7fffu) : (t <= n ? S.cum[t] : (u16)0x7fffu);
    auto owner = [&](u32 m) -> u32 {
        u32 c = 0;
        for (u32 t = 1; t < n; t++) c += (u32)S.cum[t] <= m ? 1u : 0u;       // (n <= 64; S.cum is in LDS, the same address in every lane)
        return c;
    };
    u32 e = 0;
    if (!empty) {
        const u32 lo = owner(16u * lane), hi = owner(16u * lane + 15u);
        e = lo & ~1u;
        if (hi > e + 6u) e |= MID_OVF;                    // cum[hi + 1] must still lie in the window cum[e .. e + 7]
    }
    rowp[lane] = (u8)e;
}

// Whole wave: one direct block (r4x16_common.h, "level 6") from S.cum: fb[] = the symbols that have a frequency, by
// rank, tab[j] = rank of the owner of slot 2j.  Callers synchronise before (S.cum complete) and after.
__device__ void write_row_direct(u8 *blkp, FrontShared &S, u32 n, bool empty, u32 look, u32 lane, u32 &used)
{
    u32 *fb = (u32 *)blkp;
    u32 *tab = (u32 *)(blkp + dir_fb_bytes(n));
    const u32 T = 1u << (look - 1u);
    // staging in arrays that are dead once S.cum stands (n <= DIR_MAX_NSYM = 128: 129 entries at most): the entries in
    // S.F (the row's frequencies by byte), the starts of the symbols with a frequency in S.Fk (the same by rank)
    u32 *fbl = S.F;
    u16 *cumnz = (u16 *)S.Fk;
    if (empty) {
        if (lane < 2) fb[lane] = dir_entry(0u, 1u << look, 0u, look, look == 10u ? DIR_EMPTY : 0u);   // x stays as it is
        for (u32 j = lane; j < T / 4u; j += WAVE) tab[j] = 0u;
        return;
    }
    u32 nnz = 0;
    for (u32 c0 = 0; c0 < n; c0 += WAVE) {                   // (n <= DIR_MAX_NSYM: two rounds)
        const u32 c = c0 + lane;
        const u32 b = c < n ? S.cum[c] : 0u, e = c < n ? S.cum[c + 1] : 0u;
        const bool nz = e > b;
        const u64 mk = __ballot(nz);
        const u32 r = nnz + (u32)__popcll(mk & ((1ull << lane) - 1ull));
        if (nz) { fbl[r] = dir_entry(b, e - b, c, look, 0u); cumnz[r] = (u16)b; used |= 1u << (c0 / WAVE); }
        nnz += (u32)__popcll(mk);
    }
    __syncthreads();
    for (u32 r = lane; r <= nnz; r += WAVE) fb[r] = fbl[r < nnz ? r : nnz - 1u];     // (a valid row has nnz >= 1: its total is 1 << look)
    // each lane fills T / 64 consecutive pairs: one binary search for its first slot, then a merge walk
    const u32 P = T / WAVE;
    u32 slot = 2u * P * lane;
    u32 lo = 0, hi = nnz;                                   // cumnz[lo] <= slot < cumnz[hi]  (cumnz[0] = 0, "cumnz[nnz]" = 1 << look)
    while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (cumnz[mid] <= slot) lo = mid; else hi = mid; }
    u32 r = lo;
    u32 nxt = r + 1u < nnz ? cumnz[r + 1u] : 0xffffu;
    for (u32 q = 0; q < P; q += 4u) {
        u32 w = 0;
#pragma unroll
        for (u32 k = 0; k < 4u; k++) {
            while (nxt <= slot) { r++; nxt = r + 1u < nnz ? cumnz[r + 1u] : 0xffffu; }
            w |= r << (8u * k);
            slot += 2u;
        }
        tab[(P * lane + q) >> 2] = w;
    }
}
// Whole wave: c + 1 if byte = index + c for every compact symbol marked in `used` (bit i: symbol 64 i + lane), else 0
__device__ u32 affine_of(const FrontShared &S, u32 n, u32 used, u32 lane)
{
    u32 delta[2];
    bool u[2];
#pragma unroll
    for (u32 i = 0; i < 2; i++) {
        const u32 c = WAVE * i + lane;
        u[i] = c < n && ((used >> i) & 1u);
        delta[i] = c < n ? (u32)S.alpha[c] - c : 0u;
    }
    const u64 m0 = __ballot(u[0]), m1 = __ballot(u[1]);
    if (!m0 && !m1) return 0u;
    const u32 C = m0 ? (u32)__shfl((int)delta[0], __ffsll((unsigned long long)m0) - 1) : (u32)__shfl((int)delta[1], __ffsll((unsigned long long)m1) - 1);
    const bool bad = (u[0] && delta[0] != C) || (u[1] && delta[1] != C);
    return __ballot(bad) ? 0u : C + 1u;
}

// Order-0 stream front end: src[pos, pos+len) holds table, states, words.
// rANS_static4x16pr.c:500-561.  All lanes call; on return S.status / S.R / S.words_pos are set
// and the single-row image is at `img`.
__device__ __forceinline__ void o0_front(ByteSrc &src, u32 pos, u32 len, u32 out_sz, u8 *img, FrontShared &S, u32 lane,
                                         u32 dir_budget = 0u, u32 *direct = nullptr)      // *direct: 0, or 1 + DecItem.affine
{
    for (u32 j = lane; j < 256; j += WAVE) { S.present[j] = 0; S.F[j] = 0; }
    __syncthreads();
    if (lane == 0) {
        S.status = ST_OK;
        S.nsym = 0;
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
                u32 ns = 0;                                           // compact alphabet = listed symbols
                for (u32 j = 0; j < 256; j++) if (S.present[j]) S.alpha[ns++] = (u8)j;
                S.nsym = ns;
                if (p == pos) S.status = ST_TABLE;                    // fsz == 0 :531
                else if (!make_cum(S, ns, total, O0_BITS)) S.status = ST_TABLE;
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
    if (S.status == ST_OK) {
        const u32 ns = S.nsym;
        for (u32 j = lane; j < ns; j += WAVE) ((u16 *)img)[j] = S.alpha[j];
        // (direct rows pay a larger image - built here, copied into LDS by the chain kernel - for a shorter step: only
        //  for streams with at least a step per 16 bytes of image)
        const bool dir = ns <= DIR_MAX_NSYM && dir_img_bytes(ns, 1u, O0_BITS) + RING_BYTES <= dir_budget &&
                         out_sz >= dir_img_bytes(ns, 1u, O0_BITS) / 4u;      // (uniform)
        u32 used = 0;
        if (dir) write_row_direct(img + img_alpha_bytes(ns), S, ns, false, O0_BITS, lane, used);
        else write_row(img + img_alpha_bytes(ns), S, ns, false, lane);
        if (direct) *direct = dir ? 1u + affine_of(S, ns, used, lane) : 0u;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// k_dec_front
// ---------------------------------------------------------------------------------------------
// One row of an order-1 table (decode_freq_d :327-358: a varint per ranked member, zero runs as (0, count - 1)) parsed by
// the whole wave from the register window, 64 bytes at a time, instead of by one lane byte after byte (that was most of
// k_dec_front's time: 46 rows of ~40 values per quality block).  What a byte is - part of a varint or the count behind a
// zero - depends on the bytes before it: a three-state machine (START of a value, CONTinuation of a varint, RUN count)
// whose per-byte transition functions are composed by a wave scan.  Member indices are a prefix sum of "1 per value,
// count per run byte"; the row ends at the first finished value (a zero with its count byte) at which np members are
// reached, the next chunk starts after the last finished one.  Returns false - nothing is lost, the caller's one-lane
// parse then does the row - on what only a damaged table holds: varints of more than five bytes, zeros spelt in several
// bytes, no finished value in 64 bytes, or a table that ends within the chunk.
__device__ __forceinline__ bool row_parse_wave(WinSrc &win, u32 wlimit, u32 p, u32 np, u32 *Fk, u32 *pp, u32 *ptotal, u32 lane)
{
    u32 k0 = 0, total = 0;
    for (;;) {
        if (p < win.wbase || p + 80u > win.wbase + win.wlen) {
            win.fill(p, wlimit, lane);
            if (p + 80u > win.wbase + win.wlen) return false;
        }
        const u32 bo = p - win.wbase + lane;
        const int sl = (int)(bo >> 4);
        const u32 d0 = (u32)__shfl((int)win.held.x, sl), d1 = (u32)__shfl((int)win.held.y, sl),
                  d2 = (u32)__shfl((int)win.held.z, sl), d3 = (u32)__shfl((int)win.held.w, sl);
        const u32 dlo = (bo & 4u) ? d1 : d0, dhi = (bo & 4u) ? d3 : d2;
        const u32 b = (((bo & 8u) ? dhi : dlo) >> (8u * (bo & 3u))) & 0xffu;
        const bool cont = (b & 0x80u) != 0;
        // transition function of this byte, next state from START / CONT / RUN in bits 0-1 / 2-3 / 4-5
        // (START = 0, CONT = 1, RUN = 2): continuation byte -> CONT, CONT, START; zero -> RUN, START, START; other -> START
        u32 F = cont ? 5u : (b == 0 ? 2u : 0u);
        // (the scan's data moves on DPP row operations, as in wave_incl_scan: `g` = the composition of the bytes before)
        auto after = [&](u32 g, bool has) {
            if (has) {
                const u32 h0 = (F >> (2u * (g & 3u))) & 3u, h1 = (F >> (2u * ((g >> 2) & 3u))) & 3u, h2 = (F >> (2u * ((g >> 4) & 3u))) & 3u;
                F = h0 | (h1 << 2) | (h2 << 4);
            }
        };
        after(dpp_row<0x111, 0xf>(F), (lane & 15u) >= 1u);
        after(dpp_row<0x112, 0xf>(F), (lane & 15u) >= 2u);
        after(dpp_row<0x114, 0xf>(F), (lane & 15u) >= 4u);
        after(dpp_row<0x118, 0xf>(F), (lane & 15u) >= 8u);
        after(dpp_row<0x142, 0xa>(F), (lane & 16u) != 0u);       // lane 15 of rows 0 / 2 -> rows 1 / 3
        after(dpp_row<0x143, 0xc>(F), (lane & 32u) != 0u);       // lane 31 -> rows 2 and 3
        const u32 prevF = dpp_row<0x138, 0xf>(F);                // wave_shr:1
        const u32 pre = lane ? (prevF & 3u) : 0u;             // this byte's state: the chunk starts at a value
        const bool runb = pre == 2u, vend = pre != 2u && !cont, zero1 = pre == 0u && b == 0u;
        const bool complete = runb || (vend && !zero1);
        const u32 Mi = wave_incl_scan(runb ? b : (vend ? 1u : 0u), lane);
        // the varint that ends here: up to four bytes before this one
        const u32 pk = b | (pre << 8);
        const u32 q1 = dpp_row<0x138, 0xf>(pk), q2 = dpp_row<0x138, 0xf>(q1), q3 = dpp_row<0x138, 0xf>(q2), q4 = dpp_row<0x138, 0xf>(q3);   // wave_shr:1
        const bool c1 = pre == 1u, c2 = c1 && (q1 >> 8) == 1u, c3 = c2 && (q2 >> 8) == 1u, c4 = c3 && (q3 >> 8) == 1u, c5 = c4 && (q4 >> 8) == 1u;
        const u32 v = (b & 0x7fu) | (c1 ? (q1 & 0x7fu) << 7 : 0u) | (c2 ? (q2 & 0x7fu) << 14 : 0u) | (c3 ? (q3 & 0x7fu) << 21 : 0u) |
                      (c4 ? (q4 & 0x7fu) << 28 : 0u);
        const bool odd = vend && (c5 || (c1 && v == 0u));
        const u64 cm = __ballot(complete), cand = __ballot(complete && k0 + Mi >= np);
        if (!cm) return false;
        const u32 last = cand ? (u32)__ffsll((unsigned long long)cand) - 1u : 63u - (u32)__clzll((long long)cm);
        if (__ballot(odd && lane <= last)) return false;
        const bool mine = vend && !zero1 && lane <= last;
        if (mine) Fk[k0 + Mi - 1u] = v;
        total += wave_sum(mine ? v : 0u);
        k0 += (u32)__shfl((int)Mi, (int)last);
        p += last + 1u;
        if (cand) break;
    }
    *pp = p;
    *ptotal = total;
    return true;
}

// Order-1 frequency tables (:958-998) into the decoder image, then the payload item.  `tsrc` / `tend`: the table bytes -
// the input itself, or tbuf where the table came as a nested order-0 stream.  One wave.
__device__ void o1_tables(const u8 *in, ByteSrc &src, const u8 *tbuf, bool compressed, u32 bits, u32 tab_pos, u32 usz, u32 after_table,
                          u32 pay_pos, u32 pay_len, u32 s1_size, u8 *img, DecDesc *D, DecItem *I0, FrontShared &S, i32 *hst, u32 lane,
                          u32 dir_budget, u32 mid_budget)
{
    const u32 look = bits == 12 ? 12 : 10;                             // :1027, :1071
    ByteSrc tsrc(compressed ? tbuf : in);      // tbuf was never read by this CU before the fence above
    const u32 tend = compressed ? usz : pay_pos + pay_len;

    // alphabet F0 (:958-965) and the compact alphabet F0 ∪ {0}
    for (u32 j = lane; j < 256; j += WAVE) S.present[j] = 0;
    __syncthreads();
    if (lane == 0) {
        u32 p = compressed ? 0 : tab_pos;
        const u32 used = get_alphabet(tsrc, p, tend, S.present);
        p += used;
        i32 st = ST_OK;
        if (!used || p >= tend) st = ST_TABLE;
        u32 n = 0;
        for (u32 j = 0; j < 256; j++)
            if (S.present[j] || j == 0) { S.idx_of[j] = (u8)n; S.alpha[n] = (u8)j; n++; }
        S.nsym = n;
        S.pos = p;
        *hst = st;
    }
    __syncthreads();
    if (*hst != ST_OK) { if (lane == 0) D->status = *hst; return; }

    const u32 nsym = S.nsym;
    // 10-bit tables of quality-sized alphabets take the packed rows (smaller images: more streams per CU)
    // a batch that leaves LDS to spare takes the direct rows (the short step); else 10-bit tables of quality-sized
    // alphabets take the packed rows (smaller images: more streams per CU)
    // (table precisions other than 10 and 12 bits - damaged streams only - keep the u16 rows: the entry's 12-bit fields)
    const bool direct = (bits == 10 || bits == 12) && nsym <= DIR_MAX_NSYM && dir_img_bytes(nsym, nsym, look) + RING_BYTES <= dir_budget &&
                        s1_size >= dir_img_bytes(nsym, nsym, look) / 4u;     // (a step per 16 bytes of image at least: see o0_front)
    u32 dir_used = 0;                                               // symbols (64 i + lane) that have a frequency in some row
    // (the mid rows: a batch of one partly filled round - r4x16_common.h, level 10)
    const bool mid = !direct && bits == 10 && nsym >= MID_MIN_NSYM && nsym <= MID_MAX_NSYM && mid_img_bytes(nsym) + RING_BYTES <= mid_budget &&
                     s1_size >= mid_img_bytes(nsym) / 4u;
    const bool packed = !direct && !mid && bits == 10 && nsym >= PK_MIN_NSYM && nsym <= PKW_MAX_NSYM;
    const u32 stride = direct ? dir_blk_bytes(nsym, look) : mid ? mid_row_bytes(nsym) : packed ? pk_row_bytes(nsym) : img_row_bytes(nsym);
    u8 *rows0 = img + (packed ? pk_head_bytes(nsym) : img_alpha_bytes(nsym));
    const bool pk2 = packed && nsym <= PK_MAX_NSYM;       // the alpha word of context ci: in its 8-byte head entry
    auto alpha_word = [&](u32 ci) -> u16 * { return pk2 ? (u16 *)(img + 8u * ci + 4u) : (u16 *)img + ci; };

    // Every row lists a frequency for each member of F0 (decode_freq_d :327-358): rank them once.
    if (lane == 0) {
        u32 k = 0;
        for (u32 c = 0; c < nsym; c++) S.rankof[c] = S.present[S.alpha[c]] ? (u16)k++ : (u16)0xffff;
        S.np = k;
    }
    __syncthreads();
    const u32 np = S.np;
    // The parsing lane reads the table through a 1 KB register window (un-nested tables sit in tbuf, whose
    // slot may be read past the table's end); a raw table inside the input keeps the plain reader.
    WinSrc win(&tsrc);
    const u32 wlimit = compressed ? TBUF_BYTES : 0u;

    // rows, in byte order of the compact alphabet (:967-998)
    for (u32 ci = 0; ci < nsym; ci++) {
        {   // all lanes: clear the row's frequencies, keep the window ahead of the parse position
            for (u32 c = lane; c < np; c += WAVE) S.Fk[c] = 0;
            const u32 pos_now = S.pos;
            if (wlimit && (win.wlen == 0 || pos_now < win.wbase || pos_now + 800u > win.wbase + win.wlen)) win.fill(pos_now, wlimit, lane);
        }
        __syncthreads();
        u32 total = 0;
        const u32 ctx = S.alpha[ci];
        bool parsed = false;
        if (wlimit && S.present[ctx]) {                                // (uniform) the wave parses the row where no byte of it
            const u32 p0 = S.pos;                                      // needs a bounds check - the one-lane fast path's condition
            if (p0 != tend && p0 + 6u * np + 6u <= tend) {
                u32 pn = 0;
                parsed = row_parse_wave(win, wlimit, p0, np, S.Fk, &pn, &total, lane);
                __syncthreads();
                if (parsed && lane == 0) S.pos = pn;
            }
        }
        if (lane == 0) {
            S.empty = 0;
            S.go = 1;
            if (!S.present[ctx]) {
                S.empty = 1;                                           // byte 0 outside F0
            } else if (parsed) {
                if (total == 0) S.empty = 1;                           // :977-980
            } else {
                // decode_freq_d :327-358: one value per ranked member, zero runs as (0, count - 1)
                u32 p = S.pos, zeros = 0;
                bool ok = p != tend;
                // a row is at most 6 bytes per member; when that much lies inside the window and before the
                // end of the table, no byte of the row needs a bounds check
                if (ok && p >= win.wbase && p + 6u * np + 6u <= win.wbase + win.wlen && p + 6u * np + 6u <= tend) {
                    for (u32 k = 0; k < np; k++) {
                        u32 f = 0;
                        if (zeros) zeros--;
                        else {
                            u32 c;
                            do { c = win.at_inside(p++); f = (f << 7) | (c & 0x7fu); } while (c & 0x80u);   // varint.h:131-160
                            if (f == 0) zeros = win.at_inside(p++);
                        }
                        S.Fk[k] = f;
                        total += f;
                    }
                } else
                for (u32 k = 0; ok && k < np && p < tend; k++) {
                    u32 f;
                    if (zeros) { f = 0; zeros--; }
                    else {
                        p += var_get(win, p, tend, &f);
                        if (f == 0) {
                            if (p >= tend) { ok = false; break; }
                            zeros = win.at(p++);
                        }
                    }
                    S.Fk[k] = f;
                    total += f;
                }
                if (!ok || p == S.pos) { *hst = ST_TABLE; S.go = 0; }
                else {
                    S.pos = p;
                    if (total == 0) S.empty = 1;                       // :977-980
                }
            }
            if (pk2) *(u32 *)(img + 8u * ci + 4u) = ctx | (S.empty ? ROW_EMPTY : 0u);
            else ((u16 *)img)[ci] = (u16)(ctx | (S.empty ? ROW_EMPTY : 0u));
        }
        __syncthreads();
        if (!S.go) break;
        if (!S.empty) {
            // cumulative starts by the whole wave (normalise_freq_shift :168-179 and the checks at :985-997):
            // four compact symbols per lane, scaled, prefix-summed
            total = __shfl(total, 0);
            u32 sh = 0;
            if (total != (1u << bits)) { u32 size = total; while (size < (1u << bits)) { size *= 2; sh++; } }
            u32 f[4], sum = 0, lowest = 4;
            bool bad = false;
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                const u32 cc = 4 * lane + c;
                const u32 rk = cc < nsym ? S.rankof[cc] : 0xffffu;
                f[c] = rk != 0xffffu ? S.Fk[rk] << sh : 0u;
                bad |= f[c] > (1u << bits);
                sum += f[c];
                if (f[c] && lowest == 4) lowest = c;
            }
            if (packed) {                                              // first symbol of the row with a frequency
                const u64 has = __ballot(lowest != 4);
                const int fl = has ? __ffsll((unsigned long long)has) - 1 : 0;
                const u32 fst = (u32)__shfl((int)(4 * lane + (lowest & 3u)), fl);
                if (lane == 0) S.first = has ? fst : 0u;
            }
            const bool anybad = wave_any(bad);
            if (anybad) sum = 0;                                       // keeps the scan below from wrapping
            const u32 incl = wave_incl_scan(sum, lane);
            u32 x = incl - sum;
#pragma unroll
            for (u32 c = 0; c < 4; c++) {
                const u32 cc = 4 * lane + c;
                if (cc < nsym) S.cum[cc] = (u16)x;
                x += f[c];
            }
            const u32 tot = __shfl(incl, WAVE - 1);
            if (lane == 0) {
                if (anybad || tot != (1u << bits)) { *hst = ST_TABLE; S.go = 0; }
                S.cum[nsym] = (u16)tot;
                S.cum[nsym + 1] = S.cum[nsym + 2] = S.cum[nsym + 3] = 0x7fffu;
            }
            __syncthreads();
            if (!S.go) break;
        }
        if (direct)
            write_row_direct(rows0 + (u64)ci * stride, S, nsym, S.empty != 0, look, lane, dir_used);
        else if (mid)
            write_row_mid(rows0 + (u64)ci * stride, S, nsym, S.empty != 0, lane);
        else if (packed) {
            if (lane == 0) *alpha_word(ci) |= (u16)(((S.empty ? 0u : S.first) + 2u) << PK_FIRST_SHIFT);     // (first + 2: lookup_step_pk)
            write_row_pk(rows0 + (u64)ci * stride, img + 8u * ci, S, nsym, S.empty != 0, lane);
        } else
            write_row(rows0 + (u64)ci * stride, S, nsym, S.empty != 0, lane);
        __syncthreads();
    }
    if (*hst != ST_OK) { if (lane == 0) D->status = *hst; return; }
    // (12-bit order-1 entries have no room for the empty-row flag: those streams keep alpha[] and its ROW_EMPTY flags)
    const u32 affine = direct && look == 10u ? affine_of(S, nsym, dir_used, lane) : 0u;

    if (lane == 0) {
        u32 p = compressed ? after_table : S.pos;                    // :1000-1001
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
            I0->img_bytes = direct ? dir_img_bytes(nsym, nsym, look) : mid ? mid_img_bytes(nsym) : packed ? pk_img_bytes(nsym) : img_bytes(nsym, nsym);
            I0->nsym = nsym;
            I0->packed = direct ? 2u : mid ? 3u : packed ? 1u : 0u;
            I0->affine = affine;
            I0->look = look; I0->order = 1;
            I0->active = s1_size != 0;
        }
    }
}

// PHASE 0: container header, order-0 tables, order-1 tables that sit in the stream as they are; an order-1 table that is
// itself an order-0 stream becomes a chain item (four lanes decoding ~3 KB inside this one-wave kernel were a third of
// its time on 64 KiB quality blocks).  PHASE 1, after those items have run: the order-1 tables of their blocks.
template <int PHASE>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(PHASE ? 8 : 4))) void k_dec_front(BatchArgs a, DecWs ws, int base)
{
    __shared__ FrontShared S;
    __shared__ struct {
        i32 status;
        u32 order, pay_pos, pay_len, s1_size, compressed, usz, csz, tab_pos, after_table;
        u32 bits;
        u32 meta_nested, meta_pos, meta_slen, meta_len;
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

    if (PHASE == 1) {
        const DecResume R = ws.resume[b];
        if (!R.pending || D->status != ST_OK) return;
        if (lane == 0) H.status = ST_OK;
        __syncthreads();
        o1_tables(in, src, tbuf, true, R.bits, 0u, R.usz, R.after_table, R.pay_pos, R.pay_len, R.s1_size, img, D, I0, S, &H.status, lane,
                  ws.direct_budget, ws.mid_budget);
        return;
    }

    // ---- container header: flags, sizes, PACK map, RLE meta (:1435-1572) ---------------------------
    // the block's staging region for the inverse transforms (r4x16_common.h: dec_var_bytes; laid out by k_dec_voff
    // for the blocks whose first byte carries X_PACK or X_RLE): stage buffer, then the decoded run-length meta
    const u32 vcap = cap < ws.max_out_cap ? cap : ws.max_out_cap;
    const bool vhas = ws.var && ws.voff[b + 1] > ws.voff[b] && ws.voff[b + 1] <= ws.var_bytes;
    u8 *tmp = vhas ? ws.var + ws.voff[b] : nullptr;
    u8 *metabuf = vhas ? tmp + dec_var_tmp(vcap) : nullptr;
    const u32 tmp_room = vhas ? vcap : 0u, meta_room = vhas ? vcap + 256u : 0u;
    if (lane == 0) {
        DecItem *I2 = &ws.items[2 * gridDim.x + b];
        I0->active = 0; I1->active = 0; I2->active = 0;
        I0->blk = b; I1->blk = b; I2->blk = b;
        I0->packed = 0; I1->packed = 0; I2->packed = 0;
        I0->affine = 0; I1->affine = 0; I2->affine = 0;
        ws.resume[b].pending = 0;
        D->status = ST_OK; D->cat_src = 0; D->cat_len = 0; D->osz = 0; D->s1_size = 0;
        D->pack_per = 1; D->rle_meta_len = 0; D->rle_meta = 0;
        i32 st = ST_OK;
        u32 pos = 0, osz = 0, flags = 0;
        H.pay_len = 0; H.meta_nested = 0;
        if (in_size == 0) st = ST_EMPTY;                               // :1357
        else {
            flags = src.at(pos++);
            if (flags & X_STRIPE) st = ST_UNSUPPORTED;                 // host entry points split stripes
            else {
                if (!(flags & X_NOSZ)) pos += var_get(src, pos, in_size, &osz);
                else osz = cap;
                if (cap < osz) st = ST_CAPACITY;                       // :1464
                else if ((flags & (X_PACK | X_RLE)) && osz > tmp_room) st = ST_UNSUPPORTED;
            }
        }
        D->flags = flags;
        u32 s1_size = osz;
        if (st == ST_OK) {
            D->osz = osz;
            // stage buffers (:1480-1520): rans -> s1, un-RLE -> s2, un-PACK -> s3
            const bool pk = flags & X_PACK, rl = flags & X_RLE;
            u8 *s1 = out, *s2 = out, *s3 = out;
            if (pk && rl) { s1 = out; s2 = tmp; s3 = out; }
            else if (pk)  { s1 = tmp; s2 = tmp; s3 = out; }
            else if (rl)  { s1 = tmp; s2 = out; s3 = out; }
            D->s1 = (u64)s1; D->s2 = (u64)s2; D->s3 = (u64)s3;

            if (pk) {                                                  // hts_unpack_meta, pack.c:165-198
                const u32 left = in_size - pos;
                u32 used = 0, per = 1;
                if (left == 0) st = ST_TRUNCATED;
                else {
                    u32 n = src.at(pos);
                    if (n == 0) n = 256;
                    if (n <= 1) per = 0; else if (n <= 2) per = 8; else if (n <= 4) per = 4; else if (n <= 16) per = 2;
                    if (n > 16) { per = 1; used = 1; }
                    else if (left <= 1) st = ST_TRUNCATED;
                    else {
                        // the reference's map is `uint8_t map[16] = {0}` (rANS_static4x16pr.c:1524): codes past the
                        // listed symbols - only a damaged stream holds any - decode to byte 0
                        for (u32 q = 0; q < 16; q++) D->pack_map[q] = 0;
                        u32 j = 1, c = 0;
                        do { D->pack_map[c++] = src.at(pos + j); j++; } while (c < n && j < left);
                        if (c < n) st = ST_TRUNCATED; else used = j;
                    }
                }
                if (st == ST_OK) {
                    D->pack_per = per;
                    pos += used;
                    u32 psz;
                    pos += var_get(src, pos, in_size, &psz);            // :1539-1544
                    if (psz > s1_size) st = ST_SIZE; else s1_size = psz;
                }
            }
            if (st == ST_OK && rl) {                                   // :1549-1572
                u32 mlen, lit_len, c_meta, sz;
                const u32 left = in_size - pos;
                sz = var_get(src, pos, in_size, &mlen);
                sz += var_get(src, pos + sz, in_size, &lit_len);
                if (lit_len > s1_size) st = ST_SIZE;
                else if (mlen & 1) {                                   // raw meta lives in the input
                    const u32 avail = left - sz;
                    mlen = (mlen / 2 > avail) ? avail : mlen / 2;
                    c_meta = mlen;
                    D->rle_meta = (u64)(in + pos + sz);
                    D->rle_meta_len = mlen;
                } else {                                               // order-0 compressed meta
                    sz += var_get(src, pos + sz, in_size, &c_meta);
                    mlen /= 2;
                    if (mlen > meta_room) st = ST_UNSUPPORTED;         // larger than any valid meta for this batch
                    else {
                        H.meta_nested = 1; H.meta_pos = pos + sz; H.meta_slen = left - sz; H.meta_len = mlen;
                        D->rle_meta = (u64)metabuf;
                        D->rle_meta_len = mlen;
                    }
                }
                if (st == ST_OK) {
                    if (c_meta + sz > left) st = ST_SIZE;              // :1567 (32-bit arithmetic as there)
                    else { pos += c_meta + sz; s1_size = lit_len; }
                }
            }
        }
        if (st == ST_OK) {
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
    if (H.status != ST_OK) return;

    if (H.meta_nested) {
        // the RLE meta is itself an order-0 stream: it becomes a second item for the chain kernel
        // (with LDS to spare it takes the short-step rows like any order-0 payload: the run lengths of a 1 MiB q8 block
        //  are a stream of ~170,000 symbols - 10.5 ms at three look-ups per step on 4,096 such blocks)
        u8 *imgm = img + IMG_MAX_BYTES + IMG_O0_BYTES;
        __shared__ u32 meta_direct;
        o0_front(src, H.meta_pos, H.meta_slen, H.meta_len, imgm, S, lane, ws.direct_budget, &meta_direct);
        if (S.status != ST_OK) { if (lane == 0) D->status = S.status; return; }
        if (lane == 0) {
            I1->words = (u64)(in + S.words_pos);
            I1->words_len = H.meta_pos + H.meta_slen - S.words_pos;
            I1->out = (u64)metabuf; I1->out_sz = H.meta_len; I1->image = (u64)imgm;
            I1->img_bytes = meta_direct ? dir_img_bytes(S.nsym, 1u, O0_BITS) : img_bytes(S.nsym, 1); I1->nsym = S.nsym;
            I1->packed = meta_direct ? 2u : 0u;
            I1->affine = meta_direct ? meta_direct - 1u : 0u;
            I1->look = O0_BITS; I1->order = 0;
            for (int k = 0; k < 4; k++) I1->R[k] = S.R[k];
            I1->active = H.meta_len != 0;
        }
        __syncthreads();
    }
    if (H.pay_len == 0) return;

    const u32 pay_pos = H.pay_pos, pay_len = H.pay_len, s1_size = H.s1_size;

    if (H.order == 0) {
        // ---- order-0 payload ----------------------------------------------------------------
        __shared__ u32 o0_direct;
        o0_front(src, pay_pos, pay_len, s1_size, img, S, lane, ws.direct_budget, &o0_direct);
        if (lane == 0) {
            D->status = S.status;
            if (S.status == ST_OK) {
                I0->words = (u64)(in + S.words_pos);
                I0->words_len = pay_pos + pay_len - S.words_pos;
                I0->out = D->s1; I0->out_sz = s1_size; I0->image = (u64)img;
                I0->img_bytes = o0_direct ? dir_img_bytes(S.nsym, 1u, O0_BITS) : img_bytes(S.nsym, 1); I0->nsym = S.nsym;
                I0->packed = o0_direct ? 2u : 0u;
                I0->affine = o0_direct ? o0_direct - 1u : 0u;
                I0->look = O0_BITS; I0->order = 0;
                for (int k = 0; k < 4; k++) I0->R[k] = S.R[k];
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
    if (H.compressed) {
        // the table is an order-0 stream of usz bytes inside src[tab_pos, tab_pos+csz): its one-row image goes next to
        // the block's other images, its output to tbuf; k_dec_front<1> takes over from there
        u8 *imgn = img + IMG_MAX_BYTES;
        o0_front(src, H.tab_pos, H.csz, H.usz, imgn, S, lane);
        if (lane == 0) {
            if (S.status != ST_OK) D->status = S.status;
            else if (H.usz == 0) D->status = ST_TABLE;                  // (an empty table: get_alphabet has nothing to read)
            else {
                DecItem *I2 = &ws.items[2 * gridDim.x + b];
                I2->words = (u64)(in + S.words_pos);
                I2->words_len = H.tab_pos + H.csz - S.words_pos;
                I2->out = (u64)tbuf; I2->out_sz = H.usz; I2->image = (u64)imgn;
                I2->img_bytes = img_bytes(S.nsym, 1); I2->nsym = S.nsym;
                I2->look = O0_BITS; I2->order = 0;
                for (int k = 0; k < 4; k++) I2->R[k] = S.R[k];
                I2->active = 1;
                DecResume R;
                R.pending = 1; R.pay_pos = pay_pos; R.pay_len = pay_len; R.s1_size = s1_size; R.bits = bits;
                R.usz = H.usz; R.after_table = H.after_table; R.pad = 0;
                ws.resume[b] = R;
            }
        }
        return;
    }
    o1_tables(in, src, tbuf, false, bits, H.tab_pos, 0u, 0u, pay_pos, pay_len, s1_size, img, D, I0, S, &H.status, lane, ws.direct_budget, ws.mid_budget);
}

// ---------------------------------------------------------------------------------------------
// The chain decoder for mid rows (r4x16_common.h, "level 10"), order 1, 10-bit tables.  The same ring, renormalisation
// and output gathering as chain_decode_dir, laid out around its two dependent LDS reads:
//     x -> bucket (m >> 4) -> [read: e | overflow flag]     shadow: ring dwords, last step's byte
//       -> window address  -> [read: cum[e .. e + 7]]         shadow: the candidate words out of the ring dwords
//       -> count the entries <= m (SWAR, no compares), pick cum[s], cum[s + 1] out of the four dwords, state update
// A bucket flagged MID_OVF (more symbols share its sixteen slots than the window holds) takes a scan instead.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 chain_decode_mid(const u8 *img_lds, u32 nsym, u8 *ring, gcu8 *words, u32 words_len,
                                                gu8 *out, u32 out_sz, u32 x, bool active, u32 lane)
{
    const u32 imga = lds_addr(img_lds);
    const u32 k = lane & 3;
    const u32 rows = imga + img_alpha_bytes(nsym), roww = mid_row_bytes(nsym);
    const u32 nwords = words_len >> 1;
    const u32 below = (1u << k) - 1u;
    const u32 ringa = lds_addr(ring);
    const u32 q = out_sz >> 2;
    u32 count = q + (k == 3 ? out_sz - 4 * q : 0);
    gu8 *op = out + (u64)k * q;
    if (!active) count = 0;

    gcu8 *abase = (gcu8 *)((u64)words & ~15ull);
    const u32 off0 = (u32)((u64)words & 15ull);
    const u32 avail = off0 + words_len;
    const u32 lastc = avail ? (avail - 1u) >> 4 : 0u;
    const bool loadable = active && avail != 0;
    auto load_chunk = [&](u32 c) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (loadable) v = *(gcu32x4 *)(abase + 16ull * (c < lastc ? c : lastc));
        return v;
    };
    if (active) {
        const u32x4 c0 = load_chunk(k), c1 = load_chunk(k + 4), c2 = load_chunk(k + 8);
        *(u32x4 *)(ring + 16 * k) = c0;
        *(u32x4 *)(ring + 64 + 16 * k) = c1;
        *(u32x4 *)(ring + 128 + 16 * k) = c2;
        if (k == 0) *(u32x2 *)(ring + 256) = c0.xy;
    }
    u32x4 pend = load_chunk(12 + k);
    u32 half = 0;
    __syncthreads();

    u32 row = rows, cursor = 0, bad = 0, t = 0;
    u32 acc = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    u32 hdr = 0, hdr_even = 0;
    if (count) bad = *(LAS const u16 *)(unsigned long)imga;

    auto trip = [&](auto fastc) {
        constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
        for (int u = 0; u < TRIP_STEPS; u++) {
            const u32 T = t + (u32)u;
            const bool live = FAST ? true : T < count;
            // (1) head of the dependent chain: the bucket's entry
            const u32 m = x & 1023u;
            const u32 ent = *(LAS const volatile u8 *)(unsigned long)(row + (m >> 4));
            __builtin_amdgcn_sched_barrier(0);
            // (2) in its shadow: the ring dwords, the last step's byte
            const u32 cb = off0 + 2 * cursor;
            const u32 ra = ringa + (cb & 252u);
            const u32x2 d01 = *(LAS const volatile u32x2_a4 *)(unsigned long)ra;
            const u32 d0 = d01.x, d1 = d01.y, d2 = *(LAS const volatile u32 *)(unsigned long)(ra + 8u);
            const u32 mm = __umul24(m, 0x10001u) + 0x80008000u;       // m | m << 16 | flags: the u16 rows' compare-free count
            const u32 xs = x >> 10;
            if (u > 0 || t > 0) {
                if (!FAST) bad |= live ? hdr : 0u;
                else if (u & 1) bad |= hdr | hdr_even;
                else hdr_even = hdr;
                acc = (FAST || T <= count) ? __builtin_amdgcn_alignbit(hdr, acc, 8) : acc;
            }
            if ((u & 3) == 0 && T >= 4 && (FAST || T <= count)) {
                a0 = a1; a1 = a2; a2 = a3; a3 = acc;
                if (u == 0 && (t & 15u) == 0 && active) {
                    const u32x4 v = {a0, a1, a2, a3};
                    *(GAS u32x4_unaligned *)op = v;
                    op += 16;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (3) the window cum[e .. e + 7]
            const u32 e = ent & 0x7eu;
            const u32 wa = row + 64u + 2u * e;                        // (e is even: 4-byte aligned)
            const u32x2 w01 = *(LAS const volatile u32x2_a4 *)(unsigned long)wa, w23 = *(LAS const volatile u32x2_a4 *)(unsigned long)(wa + 8u);
            __builtin_amdgcn_sched_barrier(0);
            // (4) in its shadow: the four candidate words
            const u32 wlo = __builtin_amdgcn_alignbyte(d1, d0, cb), whi = __builtin_amdgcn_alignbyte(d2, d1, cb);
            __builtin_amdgcn_sched_barrier(0);
            // (5) s = e + #{window entries <= m} - 1; (cum[s], cum[s + 1]) out of the four dwords
            u32 kk = count_le(mm, w01) + count_le(mm, w23) - 1u;      // 0 .. 6 (cum[e] <= m always)
            const bool kb1 = (kk & 2u) != 0, kb2 = (kk & 4u) != 0;
            const u32 A = kb1 ? w01.y : w01.x, B = kb1 ? w23.y : w23.x;
            const u32 Wd = kb2 ? B : A;
            const u32 A1 = kb1 ? w23.x : w01.y;
            const u32 Wd1 = kb2 ? w23.y : A1;
            u32 pair = __builtin_amdgcn_alignbyte(Wd1, Wd, (kk & 1u) << 1);      // cum[s] | cum[s + 1] << 16
            u32 s = e + kk;
            const bool ovf = active && (ent & MID_OVF) != 0u;          // (idle lanes run the full trips on whatever their LDS holds:
            if (wave_any(ovf)) {                                      //  they must not scan it - and the scan is bounded anyway)
                // the scan: from the bucket's first owner on, until the next symbol starts beyond m
                if (ovf) {
                    u32 sc = e;
                    u32 c0 = *(LAS const u16 *)(unsigned long)(row + 64u + 2u * sc), c1 = *(LAS const u16 *)(unsigned long)(row + 64u + 2u * sc + 2u);
                    while (c1 <= m && sc + 1u < nsym) { sc++; c0 = c1; c1 = *(LAS const u16 *)(unsigned long)(row + 64u + 2u * sc + 2u); }
                    s = sc; pair = c0 | (c1 << 16);
                }
            }
            const u32 start = pair & 0xffffu, next = pair >> 16;
            const u32 xn = __umul24(next - start, xs) + (m - start);  // freq <= 1024, x >> 10 < 2^22: exact mod 2^32
            u32 rown1;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(rown1) : "v"(s), "v"(roww), "v"(rows));
            const u32 hn = *(LAS const volatile u16 *)(unsigned long)(imga + 2u * s);      // byte | ROW_EMPTY: looked at in the next step
            hdr = live ? hn : hdr;
            row = live ? rown1 : row;
            x = live ? xn : x;
            // (6) renormalise: chains refill in order 0..3 from the shared cursor (see chain_decode_lds)
            const bool want = live && x < RANS_LOW;
            const u32 wm = quad_ballot(want, lane);
            const u32 pre = __popc(wm & below);
            const bool take = FAST ? want : (want && cursor + pre < nwords);
            const u32 w = __builtin_amdgcn_perm(whi, wlo, __umul24(pre, 0x0202u) + 0x0c0c0100u);
            u32 xr = (x << 16) | w;
            asm volatile("" : "+v"(xr));
            x = take ? xr : x;
            cursor += __popc(wm);
        }
        t += TRIP_STEPS;
    };
    while (wave_any(t < count)) {
        const bool slow = active && (t + TRIP_STEPS > count || cursor + 4 * TRIP_STEPS > nwords);
        if (!wave_any(slow)) trip(std::true_type{});
        else trip(std::false_type{});
        const u32 nh = (off0 + 2 * cursor) >> 6;
        if (wave_any(active && nh != half)) {
            if (active && nh != half) {
                const u32 slot = ((nh + 2) & 3u) * 64u + 16u * k;
                *(u32x4 *)(ring + slot) = pend;
                if (slot == 0) *(u32x2 *)(ring + 256) = pend.xy;
                pend = load_chunk(4 * (nh + 3) + k);
                half = nh;
            }
            __syncthreads();
        }
    }
    if (count) {
        // t steps ran (see chain_decode_lds): a chain whose count equals t still has its last byte in hdr
        const u32 lastq = count < t - 4 ? count : t - 4;
        const u32 pushed = lastq >> 2, nd = pushed & 3u;
        if (count == t) acc = __builtin_amdgcn_alignbit(hdr, acc, 8);
        if (nd == 3) { *(gu32_unaligned *)op = a1; op += 4; }
        if (nd >= 2) { *(gu32_unaligned *)op = a2; op += 4; }
        if (nd >= 1) { *(gu32_unaligned *)op = a3; op += 4; }
        const u32 rem = count - 4 * pushed;
        for (u32 jj = 0; jj < rem; jj++) op[jj] = (u8)(acc >> (8 * (4 - rem + jj)));
    }
    return active ? (bad & ROW_EMPTY) : 0u;
}

// ---------------------------------------------------------------------------------------------
// k_dec_chain: QPW streams per wave (one per quad).
// The host cannot know image sizes without a device->host round trip, so it launches one grid
// per LDS size class; a stream runs in the launch whose class (lo, hi] contains its LDS need
// (image + word ring) and every other wave exits at once.  LDS_IMG=false is the catch-all for
// images too big for LDS (lo = largest class).
// ---------------------------------------------------------------------------------------------
template <bool LDS_IMG, int LV, int TRIP = TRIP_STEPS>
__global__ __launch_bounds__(WAVE) void k_dec_chain(const DecItem *items, DecDesc *desc, const u32 *list, u32 *count,
                                                    int qpw, u32 lds_per_item, int dyn)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 lane = threadIdx.x;
    const u32 quad = lane >> 2;
    // Persistent: the grid holds as many workgroups as are resident at once and each walks shares of the batch
    // (re-dispatching a second round of workgroups into slots as they free up left CUs under-filled: 15,360 streams
    // took 153 ms instead of 2 x 63).  The streams of this launch's class: count[SCHED_COUNT] of them, their item
    // indices at list[count[SCHED_START] ..], longest first; a share is `qpw` consecutive entries, claimed from the
    // class's counter; only the first count[SCHED_SEATS] workgroups of the grid work (r4x16_sched.h).
    const int nmine = (int)count[SCHED_COUNT];
    list += count[SCHED_START];
    SchedWalk walk(count, nmine, qpw, dyn != 0);
    for (int wg = walk.next_wave(); wg >= 0; wg = walk.next_wave()) {
    const int slot = wg * qpw + (int)quad;
    const bool mine = quad < (u32)qpw && slot < nmine;
    const DecItem *I = &items[mine ? list[slot] : list[wg * qpw]];
    bool active = mine && I->active;
    const u32 nsym = active ? I->nsym : 1u;
    const u32 img_bytes = active ? I->img_bytes : 0u;
    if (!wave_any(active)) continue;
    sched_setprio(sched_prio_of(active, active ? I->out_sz : 0u, (u32)qpw * lds_per_item >= 32768u));

    // fields are read one by one (a register copy of the struct indexed by lane would spill)
    gcu8 *words = (gcu8 *)I->words;
    gu8 *out = (gu8 *)I->out;
    const u32 words_len = I->words_len, out_sz = I->out_sz, look = active ? I->look : 12u;
    const u32 order = active ? I->order : 2u;
    const u32 x0 = I->R[lane & 3];

    u32 bad;
    if (LDS_IMG) {
        // A per-stream stride of whole 128-byte lines puts every stream's table on the same banks (the 3,584 bytes of a
        // 46-symbol packed stream, layout 2, are 28 lines to the byte, and the wave's 42 LDS granules leave no byte to
        // skew them with): every other stream of such a class keeps its word ring in FRONT of its image, which shifts
        // its tables by 272 bytes = 16 bytes mod 128.  Headline decode chain 93.4 -> 92.5 ms.  (The short ring's 136 bytes
        // would leave a flipped image 8 bytes off its 16-byte alignment: its class's stride, 3,360, is not a multiple of
        // 128, so it never flips; a class that would has to keep a ring of a multiple of 16 bytes.)
        const u32 ringb = TRIP == 8 ? RING_BYTES : RING_BYTES_SHORT;
        const u32 flip_odd = (lds_per_item & 127u) == 0u ? 1u : 0u;
        // cooperative copy: the whole wave copies each quad's image in turn (16-byte pieces)
        const u64 my_img = active ? I->image : 0ull;
        for (int qd = 0; qd < qpw; qd++) {
            const u64 src = __shfl(my_img, qd * 4);
            const u32 nb = __shfl(img_bytes, qd * 4);
            if (!src) continue;
            gcu32x4 *s = (gcu32x4 *)src;
            u32x4 *dd = (u32x4 *)(lds + (u64)qd * lds_per_item + (((u32)qd & flip_odd) ? ringb : 0u));
            for (u32 j = lane; j < ((nb + 15) >> 4); j += WAVE) dd[j] = s[j];
        }
        __syncthreads();
        const bool flip = (quad & flip_odd) != 0;
        const u8 *im = lds + (u64)quad * lds_per_item + (flip ? ringb : 0u);
        u8 *ring = lds + (u64)quad * lds_per_item + (flip ? 0u : lds_per_item - ringb);
        // order-0 and order-1 streams may share a wave: run the two loops back to back
        if constexpr (LV == 10) {
            bad = chain_decode_mid(im, nsym, ring, words, words_len, out, out_sz, x0, active && order == 1, lane);
        } else if constexpr (LV == 6) {
            // (affine alphabets - byte = index + c - skip the alpha[] read per symbol; a wave takes that body only if
            //  all its streams are affine)
            const u32 aff = active ? I->affine : 1u;
            if (!wave_any(aff == 0u)) {
                bad = chain_decode_dir<1, true>(im, nsym, ring, words, words_len, out, out_sz, x0, look, aff, active && order == 1, lane);
                bad |= chain_decode_dir<0, true>(im, nsym, ring, words, words_len, out, out_sz, x0, look, aff, active && order == 0, lane);
            } else {
                bad = chain_decode_dir<1, false>(im, nsym, ring, words, words_len, out, out_sz, x0, look, 0u, active && order == 1, lane);
                bad |= chain_decode_dir<0, false>(im, nsym, ring, words, words_len, out, out_sz, x0, look, 0u, active && order == 0, lane);
            }
        } else {
        bad = chain_decode_lds<1, LV, TRIP>(im, nsym, ring, words, words_len, out, out_sz, x0, look, active && order == 1, lane);
        if (LV != 1 && LV != 5)                               // packed rows exist for order-1 streams only
            bad |= chain_decode_lds<0, ((LV == 1 || LV == 5) ? 2 : LV)>(im, nsym, ring, words, words_len, out, out_sz, x0, look, active && order == 0, lane);
        }
    } else {
        GImg im{(gcu8 *)I->image};                            // (never level 1: packed images always fit a class)
        bad = chain_decode<1, ((LV == 1 || LV == 5) ? 2 : LV)>(im, nsym, words, words_len, out, out_sz, x0, look, active && order == 1, lane);
        bad |= chain_decode<0, ((LV == 1 || LV == 5) ? 2 : LV)>(im, nsym, words, words_len, out, out_sz, x0, look, active && order == 0, lane);
    }
    if (active && bad) desc[I->blk].status = ST_CONTEXT;
    __syncthreads();                                       // LDS is reused by the next share
    }
    walk.leave();
}

// ---------------------------------------------------------------------------------------------
// Run-length expansion, rle.c:142-187, by a whole workgroup: the route of SMALL batches (round 3).  One wave per block
// (rle_expand_wave below) walks the literals 64 at a time, each trip a chain of ballots, shuffles and LDS exchanges:
// 18.6 ms for a 1 MiB q8 block however few blocks there are.  Here every thread owns a contiguous chunk of the
// literals and walks it serially - twice (more instructions per literal, since the threads of a wave diverge, but 256
// threads per block.  Measured on 1 MiB q8 blocks with X_RLE: 64 blocks 4.5 ms against 11.0, 1,024 blocks 7.0, 4,096
// blocks 21.7 against 18.6 - hence the switch by batch size in the launcher.  Phases 1-3 take 0.12 ms, the counting
// walk 0.9 ms, the writing walk 3.5 ms: its byte-by-byte run fill, which a wave repeats for its longest run):
//   1  run-stream chunks: how many varints END in each (scan -> how many end before it)
//   2  literal chunks: how many literals are run-length symbols (scan -> the rank of a chunk's first varint)
//   3  each thread finds where its first varint starts: the run chunk that holds the end of the varint before it
//      (binary search over the scan of step 1), then a walk inside that chunk
//   4  walk: the bytes the chunk expands to (scan, 64-bit: a hostile run is up to 4 GiB -> the chunk's output offset)
//   5  walk again, writing: a literal, or a run as 16-byte pieces
// Bounds as in the reference: every literal needs room (rle.c:165), a run needs room for all of it (:173); a varint
// that would start past the end of the run stream reads as 0 (varint.h:136), of any length otherwise.
// ---------------------------------------------------------------------------------------------
#define BACK_THREADS 256u
#define BACK_FIFO 512u
#define BACK_TRIP 256u
#define BACK_FEED 1024u
struct BackShared {
    u8  is_rle[256];
    u32 scan_a[BACK_THREADS + 1];      // varint ends before each run chunk (+ total)
    u32 rank0[BACK_THREADS];           // rank of each literal chunk's first varint
    u32 wave_tot[8];
    u64 wave_tot64[8];
    u32 err;
    u8  map[16];
};

// exclusive prefix sum over the workgroup (every thread calls); *total = the sum
__device__ __forceinline__ u32 wg_excl_scan(u32 v, u32 *total, BackShared &B, u32 tid)
{
    const u32 lane = tid & (WAVE - 1), wv = tid / WAVE;
    const u32 incl = wave_incl_scan(v, lane);
    __syncthreads();
    if (lane == WAVE - 1) B.wave_tot[wv] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (u32 w = 0; w < BACK_THREADS / WAVE; w++) { const u32 t = B.wave_tot[w]; if (w < wv) base += t; tot += t; }
    *total = tot;
    return base + incl - v;
}
__device__ __forceinline__ u64 wg_excl_scan64(u64 v, u64 *total, BackShared &B, u32 tid)
{
    const u32 lane = tid & (WAVE - 1), wv = tid / WAVE;
    u64 incl = v;
#pragma unroll
    for (int dd = 1; dd < WAVE; dd <<= 1) {
        const u64 tt = __shfl_up(incl, dd);
        if (lane >= (u32)dd) incl += tt;
    }
    __syncthreads();
    if (lane == WAVE - 1) B.wave_tot64[wv] = incl;
    __syncthreads();
    u64 base = 0, tot = 0;
    for (u32 w = 0; w < BACK_THREADS / WAVE; w++) { const u64 t = B.wave_tot64[w]; if (w < wv) base += t; tot += t; }
    *total = tot;
    return base + incl - v;
}

// a thread's output: bytes gathered into 16-byte pieces (global stores need no alignment)
struct RunOut {
    gu8 *p;
    u32x4 acc;           // the last (up to) sixteen bytes, the newest in the top byte
    u32 cnt;
    __device__ __forceinline__ void put(u32 b)
    {
        acc.x = __builtin_amdgcn_alignbit(acc.y, acc.x, 8);
        acc.y = __builtin_amdgcn_alignbit(acc.z, acc.y, 8);
        acc.z = __builtin_amdgcn_alignbit(acc.w, acc.z, 8);
        acc.w = (acc.w >> 8) | (b << 24);
        if (++cnt == 16) { *(GAS u32x4_unaligned *)p = acc; p += 16; cnt = 0; }
    }
    __device__ __forceinline__ void flush()
    {
        for (u32 k = cnt; k < 16; k++) {
            acc.x = __builtin_amdgcn_alignbit(acc.y, acc.x, 8);
            acc.y = __builtin_amdgcn_alignbit(acc.z, acc.y, 8);
            acc.z = __builtin_amdgcn_alignbit(acc.w, acc.z, 8);
            acc.w >>= 8;
        }
        const u32 w[4] = {acc.x, acc.y, acc.z, acc.w};
        for (u32 k = 0; k < cnt; k++) p[k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
        p += cnt; cnt = 0;
    }
    // `n` copies of byte b
    __device__ __forceinline__ void fill(u32 b, u64 n)
    {
        if (n < 40) { for (u32 k = 0; k < (u32)n; k++) put(b); return; }
        while (cnt) { put(b); n--; }                       // completes the piece in hand (cnt < 16 <= n)
        const u32 b4 = b * 0x01010101u;
        const u32x4 v = {b4, b4, b4, b4};
        for (; n >= 16; n -= 16) { *(GAS u32x4_unaligned *)p = v; p += 16; }
        for (u32 k = 0; k < (u32)n; k++) put(b);
    }
};

__device__ bool rle_expand_wg(const u8 *lit, u32 lit_len, const u8 *runs, u32 run_len, const u8 *syms,
                           u32 nsyms, u8 *out, u32 cap, u32 &produced, BackShared &B, u32 tid)
{
    for (u32 j = tid; j < 256; j += BACK_THREADS) B.is_rle[j] = 0;
    if (tid == 0) B.err = 0;
    __syncthreads();
    for (u32 j = tid; j < nsyms; j += BACK_THREADS) B.is_rle[syms[j]] = 1;
    __syncthreads();

    // 1: varint ends per run chunk (a byte without the continuation bit, or the last byte of the stream)
    const u32 rc = ((run_len + BACK_THREADS - 1) / BACK_THREADS + 15u) & ~15u;       // chunk bytes, a multiple of 16
    const u32 r0 = tid * rc < run_len ? tid * rc : run_len, r1 = r0 + rc < run_len ? r0 + rc : run_len;
    u32 ends = 0;
    {
        gcu8 *g = to_global(runs);
        u32 i = r0;
        for (; i + 16 <= r1; i += 16) {
            const u32x4 v = *(GAS const u32x4_unaligned *)(g + i);
            ends += __popc(~v.x & 0x80808080u) + __popc(~v.y & 0x80808080u) + __popc(~v.z & 0x80808080u) + __popc(~v.w & 0x80808080u);
        }
        for (; i < r1; i++) ends += (g[i] & 0x80u) ? 0u : 1u;
        if (r1 == run_len && r1 > r0 && (g[r1 - 1] & 0x80u)) ends++;                  // the stream ends inside a varint: it ends there
    }
    u32 total_ends;
    const u32 ends_before = wg_excl_scan(ends, &total_ends, B, tid);
    B.scan_a[tid] = ends_before;
    if (tid == 0) B.scan_a[BACK_THREADS] = total_ends;

    // 2: run-length symbols per literal chunk
    const u32 lc = ((lit_len + BACK_THREADS - 1) / BACK_THREADS + 15u) & ~15u;
    const u32 l0 = tid * lc < lit_len ? tid * lc : lit_len, l1 = l0 + lc < lit_len ? l0 + lc : lit_len;
    u32 nr = 0;
    {
        gcu8 *g = to_global(lit);
        u32 i = l0;
        for (; i + 16 <= l1; i += 16) {
            const u32x4 v = *(GAS const u32x4_unaligned *)(g + i);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
                nr += (u32)B.is_rle[w[q] & 0xff] + (u32)B.is_rle[(w[q] >> 8) & 0xff] + (u32)B.is_rle[(w[q] >> 16) & 0xff] + (u32)B.is_rle[w[q] >> 24];
        }
        for (; i < l1; i++) nr += B.is_rle[g[i]];
    }
    u32 total_rle;
    const u32 rank = wg_excl_scan(nr, &total_rle, B, tid);       // (also orders scan_a's writes before the reads below)
    __syncthreads();

    // 3: where this chunk's first varint starts = one past the end of varint number `rank` (counting from 1)
    u32 rp = 0;
    if (rank != 0) {
        if (rank > total_ends) rp = run_len;                      // the stream has run out: every later varint reads as 0
        else {
            u32 lo = 0, hi = BACK_THREADS;                        // scan_a[lo] < rank <= scan_a[hi]
            while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (B.scan_a[mid] < rank) lo = mid; else hi = mid; }
            u32 need = rank - B.scan_a[lo];                       // the need-th end inside run chunk lo
            u32 i = lo * rc;
            const u32 iend = i + rc < run_len ? i + rc : run_len;
            ByteSrc rs0(runs);
            for (; i < iend; i++) {
                const bool e = !(rs0.at(i) & 0x80u) || i == run_len - 1u;
                if (e && --need == 0) break;
            }
            rp = i + 1u;
        }
    }

    // The literals of a chunk come through LDS, 64 per thread at a time: four 16-byte loads in flight instead of one
    // dependent 8-byte load per eight literals (a thread's walk is a chain of memory latencies otherwise).
    // (the tile lives in the workgroup route's kernel only: 17 KB of LDS in the one-wave kernel would cut its resident
    //  blocks per CU from 32 to 8 - measured: 16,384 x 1 MiB q4 blocks with X_PACK|X_RLE 130 -> 145 ms per step)
    __shared__ u8 tile[BACK_THREADS * 68u];               // 64 literals per thread at a stride of 17 dwords (no two threads on one bank)
    ByteSrc rs(runs);
    gcu8 *glit = to_global(lit);
    u8 *slot = tile + tid * 68u;
    auto stage = [&](u32 i0) {                              // literals [i0, i0 + 64) of this thread's chunk into its slot
        u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 at = i0 + 16u * (u32)q;
            v[q] = u32x4{0, 0, 0, 0};
            if (at + 16u <= l1) v[q] = *(GAS const u32x4_unaligned *)(glit + at);
            else for (u32 k = 0; at + k < l1 && k < 16u; k++) ((u8 *)&v[q])[k] = glit[at + k];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 *d = (u32 *)(slot + 16 * q);
            d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w;
        }
    };
    // 4: the bytes this chunk expands to
    u64 mine = 0;
    {
        u32 r = rp;
        for (u32 i0 = l0; i0 < l1; i0 += 64u) {
            stage(i0);
            const u32 m = l1 - i0 < 64u ? l1 - i0 : 64u;
            for (u32 k = 0; k < m; k++) {
                const u32 b = slot[k];
                u32 rlen = 0;
                if (B.is_rle[b]) r += var_get(rs, r, run_len, &rlen);
                mine += 1ull + rlen;
            }
        }
    }
    u64 total_out;
    const u64 obase = wg_excl_scan64(mine, &total_out, B, tid);

    // 5: write
    {
        u32 r = rp;
        u64 o = obase;
        RunOut ro{to_global(out) + (o < cap ? o : 0), {0, 0, 0, 0}, 0};
        bool bad = false;
        for (u32 i0 = l0; i0 < l1 && !bad; i0 += 64u) {
            stage(i0);
            const u32 m = l1 - i0 < 64u ? l1 - i0 : 64u;
            for (u32 k = 0; k < m; k++) {
                if (o >= cap) { bad = true; break; }                               // rle.c:165
                const u32 b = slot[k];
                u32 rlen = 0;
                if (B.is_rle[b]) r += var_get(rs, r, run_len, &rlen);
                if (rlen) {
                    if (o + rlen >= cap) { bad = true; break; }                    // rle.c:173
                    ro.fill(b, 1ull + rlen);
                } else ro.put(b);
                o += 1ull + rlen;
            }
        }
        ro.flush();
        if (bad) B.err = 1;
    }
    __syncthreads();
    produced = (u32)total_out;
    return B.err == 0;
}

// ---------------------------------------------------------------------------------------------
// Run-length expansion, rle.c:142-187, by one wave: the route of LARGE batches (every block resident at once; with
// thousands of blocks the wave slots are full and what counts is instruction issues per literal).  64 literals per
// trip.  Round 2's trip matched run lengths to literals straight from the run stream: a load whose address is this
// trip's result (the cursor), two barriers, and a byte store per lane and run byte - 18.6 ms for 4,096 x 1 MiB q8 blocks,
// 13.7 of them with the stores removed: the trip was a chain of memory latencies.  Round 3:
//   producer  the run stream is decoded 60 bytes at a time, in stream order and whatever the literals are, into a ring of
//             values in LDS (a varint's value is a function of its last five bytes: four bytes of history travel with
//             each batch) - its loads have addresses known trips ahead and are requested two batches ahead;
//   consumer  the k-th run-length literal takes the k-th value of the ring (varint.h:136: 0 once the stream has run out);
//   output    by OUTPUT position, a dword per lane and 256 bytes per pass: the lanes mark where their literals start
//             in an LDS tile, a prefix maximum over the marks names the literal every byte belongs to, ds_bpermute
//             fetches the values; a pass without any mark lies inside one run and is a plain fill.  Up to three
//             bytes that do not fill a dword wait for the next trip, so every store is a whole dword.
// A trip with a run of 2^22 bytes or more (256 of them would take a trip's prefix sums beyond 32 bits; only a hostile
// stream or a giant block has one) takes the plain route: byte stores, 64-bit sums.
// ---------------------------------------------------------------------------------------------
typedef u32 u32_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ u32 wave_incl_max(u32 v)
{
    v = max(v, dpp_row<0x111, 0xf>(v));
    v = max(v, dpp_row<0x112, 0xf>(v));
    v = max(v, dpp_row<0x114, 0xf>(v));
    v = max(v, dpp_row<0x118, 0xf>(v));
    v = max(v, dpp_row<0x142, 0xa>(v));
    v = max(v, dpp_row<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ u32 wave_shr1(u32 v) { return dpp_row<0x138, 0xf>(v); }   // lane l reads lane l-1, lane 0 reads 0

__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
template <class T> __device__ __forceinline__ T *uni_ptr(T *p)
{
    const u64 a = (u64)p;
    return (T *)(((u64)uni((u32)(a >> 32)) << 32) | uni((u32)a));
}

__device__ bool rle_expand_wave(const u8 *lit_, u32 lit_len_, const u8 *runs_, u32 run_len_, const u8 *syms,
                           u32 nsyms, u8 *out_, u32 cap_, u32 &produced, BackShared &B, u32 lane)
{
    // the arguments come out of the block's header, loaded by every lane: the compiler cannot know that they are the
    // same in all of them, and would keep the cursors below in vector registers and every branch on the execution mask
    // (and wait for the loads at their first use - inside the trip, where the wait also covers the trip's prefetches)
    const u8 *lit = uni_ptr(lit_), *runs = uni_ptr(runs_);
    u8 *out = uni_ptr(out_);
    const u32 lit_len = uni(lit_len_), run_len = uni(run_len_), cap = uni(cap_);
    for (u32 j = lane; j < 256; j += WAVE) B.is_rle[j] = 0;
    __syncthreads();
    for (u32 j = lane; j < nsyms; j += WAVE) B.is_rle[syms[j]] = 1;
    __syncthreads();

    const u64 lane_below = (1ull << lane) - 1ull;
    gcu8 *glit = to_global(lit), *gruns = to_global(runs);
    gu8 *gout = to_global(out);
    // (this route's buffers live in its own kernel only, like the workgroup route's tile)
    __shared__ __attribute__((aligned(16))) u8 litbuf_[16 + 2 * BACK_FEED];   // the literals: the KiB of the trip in hand, the KiB
    u8 *litbuf = litbuf_ + 16;                                                 // after it (and a byte in front that may be read)
    __shared__ __attribute__((aligned(16))) u8 rring[2 * BACK_FEED];     // the run stream around the producer's cursor
    __shared__ u32 fifo[BACK_FIFO];                                      // run lengths decoded ahead of the literals
    __shared__ __attribute__((aligned(8))) u32 tile[2 * WAVE];           // where literals start within 256 bytes of output (16-bit marks)
    u16 *tile16 = (u16 *)tile;

    // Both inputs reach the trips through LDS, a KiB at a time (sixteen bytes per lane, requested one refill ahead):
    // a load per trip, however far ahead it is requested, makes the trip wait for ALL memory operations in flight at the
    // first use of what it loaded - the counter is in order and the number of stores behind it is not known to the
    // compiler - i.e. for the request it has just made (measured: 12.4 ms for the 4,096 q8 blocks that way).
    auto load16 = [&](gcu8 *g, u32 at, u32 len) -> u32x4 {               // bytes [at + 16 lane, + 16) of a stream of len bytes
        const u32 o = at + 16u * lane;
        u32x4 v = {0, 0, 0, 0};
        if (o + 16u <= len) v = *(GAS const u32x4_unaligned *)(g + o);
        else if (o < len) {
            u32 w[4] = {0, 0, 0, 0};
            for (u32 c = 0; o + c < len; c++) w[c >> 2] |= (u32)g[o + c] << (8 * (c & 3));
            v = u32x4{w[0], w[1], w[2], w[3]};
        }
        return v;
    };

    // producer: batch q covers run bytes [q - 4, q + 60); lanes 0-3 carry history (before the stream: "a varint ended here")
    u32 rq = 0, made = 0, used = 0;                       // bytes decoded; values decoded / handed out
    u32 rl = 0;                                           // run bytes in the ring
    u32x4 rpre = load16(gruns, 0, run_len);
    auto produce = [&]() {
        if (rq + 60u > rl) {                              // (the half written held bytes before rq - 4)
            *(u32x4 *)(rring + (rl & BACK_FEED) + 16u * lane) = rpre;
            rl += BACK_FEED;
            rpre = load16(gruns, rl, run_len);
        }
        const u32 p = rq + lane - 4u;
        const bool have = p < run_len;
        const u32 c = have ? (u32)rring[p & (2u * BACK_FEED - 1u)] : 0u;
        const bool isend = (int)p < 0 || (have && (!(c & 0x80u) || p == run_len - 1u));
        const u64 Eall = __ballot(isend);
        const u64 prevE = Eall & lane_below;
        const u32 start = prevE ? 64u - (u32)__clzll(prevE) : 0u;     // the lane of this varint's first byte
        u32 v = c & 0x7fu, cd = c;
#pragma unroll
        for (int dd = 1; dd <= 4; dd++) {
            cd = wave_shr1(cd);
            if (lane >= start + (u32)dd) v |= (cd & 0x7fu) << (7 * dd);
        }
        const u64 E = Eall & ~0xfull;
        if (isend && lane >= 4u) fifo[(made + (u32)__popcll(E & lane_below)) & (BACK_FIFO - 1u)] = v;
        made += (u32)__popcll(E);
        rq += 60u;
    };

    u64 outp = 0;                                         // bytes expanded so far
    u32 cb = 0, carry_w = 0;                              // of which the last cb (< 4) wait in carry_w for their dword
    bool err = false;
    u32x4 lpre = load16(glit, 0, lit_len);
    // A trip is 256 literals, four consecutive ones per lane (the 64-literal trip of the first version spent 118 vector
    // instructions per trip, most of them the same whatever the literals: two scans, the ring bookkeeping, a pass).
    u32 wnext = 0, fnext = 0;                             // the next trip's literals of this lane, which of them are run-length symbols
    auto fetch = [&](u32 at) {                            // ... of the trip at literal `at` (< lit_len)
        if ((at & (BACK_FEED - 1u)) == 0) {               // (into the half the trip in hand does not read: its passes still look literals up)
            *(u32x4 *)(litbuf + (at & BACK_FEED) + 16u * lane) = lpre;
            lpre = load16(glit, at + BACK_FEED, lit_len);
        }
        wnext = *(const u32 *)(litbuf + (at & (2u * BACK_FEED - 1u)) + 4u * lane);
        fnext = (u32)B.is_rle[wnext & 0xffu] | ((u32)B.is_rle[(wnext >> 8) & 0xffu] << 1) |
                ((u32)B.is_rle[(wnext >> 16) & 0xffu] << 2) | ((u32)B.is_rle[wnext >> 24] << 3);
    };
    if (lit_len) fetch(0);
    for (u32 base = 0; base < lit_len; base += BACK_TRIP) {
        const u32 i0 = base + 4u * lane;
        const u32 nv = i0 >= lit_len ? 0u : (lit_len - i0 < 4u ? lit_len - i0 : 4u);     // this lane's literals in the trip
        const u32 vmask = (1u << nv) - 1u;
        const u32 w = wnext;
        const u32 fl = fnext & vmask;
        const u32 lb = base & (2u * BACK_FEED - 1u);      // the trip's literals are litbuf[lb .. lb + 256)
        if (base + BACK_TRIP < lit_len) fetch(base + BACK_TRIP);
        const u32 bv[4] = {w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, w >> 24};
        // the k-th run-length literal takes the k-th value of the ring
        const u32 nrl = (u32)__popc(fl);
        const u32 rincl = wave_incl_scan(nrl, lane);
        const u32 nr = (u32)__builtin_amdgcn_readlane((int)rincl, WAVE - 1);
        while (made - used < nr && rq < run_len) produce();            // (at most 255 values wait: 255 + 60 fit the ring)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // one wave: its LDS operations complete in order
        const u32 avail = made - used;
        u32 rv[4];
        {
            u32 rk = rincl - nrl;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool is = (fl >> k) & 1u;
                rv[k] = (is && rk < avail) ? fifo[(used + rk) & (BACK_FIFO - 1u)] : 0u;
                rk += is ? 1u : 0u;
            }
        }
        used += nr < avail ? nr : avail;

        if (__ballot((rv[0] | rv[1] | rv[2] | rv[3]) > 0x003fffffu)) {           // (256 lengths of up to 2^22 + 1 bytes: sums below 2^31)
            // ---- the plain route of a trip with a giant run: 64-bit sums, byte stores, the wave fills long runs together
            for (u32 k = lane; k < cb; k += WAVE) gout[outp - cb + k] = (u8)(carry_w >> (8 * k));
            cb = 0;
            u64 len[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { len[k] = ((vmask >> k) & 1u) ? 1ull + rv[k] : 0ull; sum += len[k]; }
            u64 incl = sum;
#pragma unroll
            for (int dd = 1; dd < WAVE; dd <<= 1) {
                const u64 tt = __shfl_up(incl, dd);
                if (lane >= (u32)dd) incl += tt;
            }
            const u64 total = __shfl(incl, WAVE - 1);
            if (outp + total > cap) { err = true; break; }              // rle.c:165, :173 (see below)
            u64 at = outp + incl - sum;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool v = (vmask >> k) & 1u;
                u8 *o = out + at;
                if (v) o[0] = (u8)bv[k];
                if (v && rv[k] && rv[k] <= 24u)
                    for (u32 j = 1; j <= rv[k]; j++) o[j] = (u8)bv[k];
                u64 longm = __ballot(v && rv[k] > 24u);
                while (longm) {
                    const int src = __ffsll((unsigned long long)longm) - 1;
                    longm &= longm - 1;
                    const u64 from = __shfl(at, src);
                    const u32 len1 = (u32)__shfl((int)rv[k], src), b1 = (u32)__shfl((int)bv[k], src);
                    u8 *ro = out + from + 1;
                    for (u32 j = lane; j < len1; j += WAVE) ro[j] = (u8)b1;
                }
                at += len[k];
            }
            outp += total;
            continue;
        }

        u32 len[4];
#pragma unroll
        for (int k = 0; k < 4; k++) len[k] = ((vmask >> k) & 1u) ? 1u + rv[k] : 0u;
        const u32 sum = len[0] + len[1] + len[2] + len[3];
        const u32 incl = wave_incl_scan(sum, lane);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
        // rle.c:165, :173: every literal needs room, a run needs room for all of it.  Positions rise with the literal, so the
        // trip's last byte decides for all of them (a literal at o needs o < cap, its run o + run < cap).
        if (outp + total > cap) { err = true; break; }

        // the trip's bytes behind the cb carried ones, as dwords from out + ob; e[k]: where this lane's k-th literal starts
        gu8 *ob = gout + (outp - cb);
        const u32 nbytes = cb + total, W = nbytes >> 2;
        u32 e[4];
        e[0] = cb + incl - sum; e[1] = e[0] + len[0]; e[2] = e[1] + len[1]; e[3] = e[2] + len[2];
#pragma unroll
        for (int k = 0; k < 4; k++) if (!((vmask >> k) & 1u)) e[k] = 0xffffffffu;       // (never inside a pass, never before one)
        const u32 keep = cb ? ~0u << (8 * cb) : ~0u;      // pass 0, lane 0: the bytes that are new
        u32 next_carry = 0;
        for (u32 P0 = 0; P0 < nbytes; P0 += 256u) {
            const bool in0 = e[0] - P0 < 256u, in1 = e[1] - P0 < 256u, in2 = e[2] - P0 < 256u, in3 = e[3] - P0 < 256u;
            const u64 inm = __ballot(in0 || in1 || in2 || in3);
            // literals that start before the pass (they are the trip's first: positions rise with the literal)
            const u32 nbefore = (u32)__popcll(__ballot(e[0] < P0)) + (u32)__popcll(__ballot(e[1] < P0)) +
                                (u32)__popcll(__ballot(e[2] < P0)) + (u32)__popcll(__ballot(e[3] < P0));
            u32 wout;
            if (!inm) wout = (u32)litbuf[lb + nbefore - 1u] * 0x01010101u;                // inside one run
            else {
                *(u32x2 *)(tile + 2u * lane) = u32x2{0u, 0u};
                const u32 id = 4u * lane + 1u;                                            // literal index in the trip + 1
                if (in0) tile16[e[0] - P0] = (u16)id;
                if (in1) tile16[e[1] - P0] = (u16)(id + 1u);
                if (in2) tile16[e[2] - P0] = (u16)(id + 2u);
                if (in3) tile16[e[3] - P0] = (u16)(id + 3u);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const u32x2 t = *(const u32x2 *)(tile + 2u * lane);
                u32 m0 = t.x & 0xffffu, m1 = max(m0, t.x >> 16), m2 = max(m1, t.y & 0xffffu), m3 = max(m2, t.y >> 16);
                const u32 pre = max(wave_shr1(wave_incl_max(m3)), nbefore);
                m0 = max(m0, pre); m1 = max(m1, pre); m2 = max(m2, pre); m3 = max(m3, pre);
                // (m = 0: a carried byte in front of the trip's first literal - read whatever is there, lane 0 replaces it)
                const u8 *lv = litbuf + lb - 1u;
                wout = (u32)lv[m0] | ((u32)lv[m1] << 8) | ((u32)lv[m2] << 16) | ((u32)lv[m3] << 24);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                    // the tile is rewritten by the next pass
            }
            if (P0 == 0 && lane == 0) wout = (wout & keep) | (carry_w & ~keep);
            const u32 d = (P0 >> 2) + lane;
            if (d < W) *(GAS u32_unaligned *)(ob + 4ull * d) = wout;
            if ((W >> 6) == (P0 >> 8)) next_carry = (u32)__builtin_amdgcn_readlane((int)wout, (int)(W & 63u));
        }
        carry_w = next_carry;
        cb = nbytes & 3u;
        outp += total;
    }
    if (!err)
        for (u32 k = lane; k < cb; k += WAVE) gout[outp - cb + k] = (u8)(carry_w >> (8 * k));
    produced = (u32)outp;
    return !err;
}

// hts_unpack, pack.c:211-348, by the workgroup.  Returns false when the packed data is too short.
template <u32 NT>
__device__ bool unpack(const u8 *data, u32 len, u8 *out, u32 out_len, u32 per, BackShared &B, u32 lane /* thread of the workgroup */)
{
    if (per == 1) { group_copy<NT>(out, data, len, lane); return true; }
    if (per == 0) {
        const u8 v = B.map[0];
        for (u32 i = lane; i < out_len; i += NT) out[i] = v;
        return true;
    }
    if ((out_len + per - 1) / per > len) return false;
    const u32 width = 8 / per, mask = (1u << width) - 1u;
    if (per == 4 || per == 8) {
        // <= 4 symbols: the map fits one register and v_perm_b32 is the look-up, its selector the packed byte's codes
        // spread one per byte.
        // Sixteen (per == 4) or thirty-two output bytes per lane and trip from one packed dword.
        const u32 m4 = (u32)B.map[0] | ((u32)B.map[1] << 8) | ((u32)B.map[2] << 16) | ((u32)B.map[3] << 24);
        const u32 in_per = 4u, out_per = in_per * per;                       // bytes in / out per lane and trip
        const u32 trips = out_len / out_per;
        // Four packed dwords per lane and trip; a trip's memory operations leave together at its top, behind an explicit
        // wait: the unpacked bytes of the trip before, then the requests for the trip after, then the trip's own arithmetic
        // on dwords that arrived during the last one.  (Loads and stores share one counter and the compiler takes them to
        // complete in any order: a request and a store per dword made each dword's first use wait for the store just issued.)
        gcu8 *gdata = to_global(data);
        gu8 *gout = to_global(out);
        auto ldw = [&](u32 t) -> u32 { return t < trips ? *(GAS const u32_unaligned *)(gdata + 4ull * t) : 0u; };
        // byte b -> selector bytes (b & 3, b >> 2 & 3, b >> 4 & 3, b >> 6): nibbles to bits 0 and 16, then pairs
        // to every byte (two shift-ors and two masks; the shifted copies never overlap)
        auto sel4 = [](u32 b) -> u32 { const u32 x = (b | (b << 12)) & 0x000f000fu; return (x | (x << 6)) & 0x03030303u; };
        auto expand4 = [&](u32 w) -> u32x4 {
            u32x4 v;
            v.x = __builtin_amdgcn_perm(m4, m4, sel4(w & 0xffu));
            v.y = __builtin_amdgcn_perm(m4, m4, sel4((w >> 8) & 0xffu));
            v.z = __builtin_amdgcn_perm(m4, m4, sel4((w >> 16) & 0xffu));
            v.w = __builtin_amdgcn_perm(m4, m4, sel4(w >> 24));
            return v;
        };
        // eight 1-bit codes per byte: two selector dwords per packed byte (b * 0x204081 spreads bits 0..3 to bytes)
        auto expand8 = [&](u32 w, u32 half) -> u32x4 {       // packed bytes 2 half, 2 half + 1 -> sixteen output bytes
            const u32 b0 = (w >> (16 * half)) & 0xffu, b1 = (w >> (16 * half + 8)) & 0xffu;
            u32x4 v;
            v.x = __builtin_amdgcn_perm(m4, m4, __umul24(b0 & 15u, 0x204081u) & 0x01010101u);
            v.y = __builtin_amdgcn_perm(m4, m4, __umul24(b0 >> 4, 0x204081u) & 0x01010101u);
            v.z = __builtin_amdgcn_perm(m4, m4, __umul24(b1 & 15u, 0x204081u) & 0x01010101u);
            v.w = __builtin_amdgcn_perm(m4, m4, __umul24(b1 >> 4, 0x204081u) & 0x01010101u);
            return v;
        };
        auto put = [&](u32 t, const u32x4 &lo, const u32x4 &hi) {
            if (t >= trips) return;
            gu8 *o = gout + (u64)out_per * t;
            *(GAS u32x4_unaligned *)o = lo;
            if (per == 8) *(GAS u32x4_unaligned *)(o + 16) = hi;
        };
        u32 wq[4] = {ldw(lane), ldw(lane + NT), ldw(lane + 2 * NT), ldw(lane + 3 * NT)};
        u32x4 rl[4] = {}, rh[4] = {};
        u32 rt = trips;                                      // the trip whose bytes wait in rl / rh (trips: none)
        for (u32 t0 = lane; t0 < trips; t0 += 4 * NT) {
            __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): wq[] is here, the last trip's stores are out
            __builtin_amdgcn_sched_barrier(0);
            if (rt < trips) {
#pragma unroll
                for (u32 qi = 0; qi < 4; qi++) put(rt + qi * NT, rl[qi], rh[qi]);
            }
            u32 nq[4];
#pragma unroll
            for (u32 qi = 0; qi < 4; qi++) nq[qi] = ldw(t0 + (4 + qi) * NT);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (u32 qi = 0; qi < 4; qi++) {
                if (per == 4) rl[qi] = expand4(wq[qi]);
                else { rl[qi] = expand8(wq[qi], 0); rh[qi] = expand8(wq[qi], 1); }
                wq[qi] = nq[qi];
            }
            rt = t0;
        }
        if (rt < trips) {
#pragma unroll
            for (u32 qi = 0; qi < 4; qi++) put(rt + qi * NT, rl[qi], rh[qi]);
        }
        for (u32 i = trips * out_per + lane; i < out_len; i += NT)
            out[i] = B.map[(data[i / per] >> ((i % per) * width)) & mask];
        return true;
    }
    if (per == 2) {
        // 5 .. 16 symbols, two codes per byte: the map is sixteen bytes in four registers, a code's byte comes from
        // v_perm_b32 over the lower or the upper eight entries, chosen by the code's bit 3.  Eight packed bytes in, sixteen
        // bytes out per lane and trip, the memory operations grouped as above.  (The loop below - four byte loads, four LDS
        // look-ups and a store per output dword, each store waited for by the next load's first use - took 3.7 ms for
        // 4,096 x 1 MiB q8 blocks.)
        const u32 m0 = (u32)B.map[0] | ((u32)B.map[1] << 8) | ((u32)B.map[2] << 16) | ((u32)B.map[3] << 24);
        const u32 m1 = (u32)B.map[4] | ((u32)B.map[5] << 8) | ((u32)B.map[6] << 16) | ((u32)B.map[7] << 24);
        const u32 m2 = (u32)B.map[8] | ((u32)B.map[9] << 8) | ((u32)B.map[10] << 16) | ((u32)B.map[11] << 24);
        const u32 m3 = (u32)B.map[12] | ((u32)B.map[13] << 8) | ((u32)B.map[14] << 16) | ((u32)B.map[15] << 24);
        const u32 trips = out_len / 16u;
        gcu8 *gdata = to_global(data);
        gu8 *gout = to_global(out);
        auto ld2 = [&](u32 t) -> u32x2 { u32x2 v = {0u, 0u}; if (t < trips) v = *(GAS const u32x2_unaligned *)(gdata + 8ull * t); return v; };
        auto four = [&](u32 x) -> u32 {                      // two packed bytes (bits 0..15) -> four output bytes
            const u32 t = (x | (x << 8)) & 0x00ff00ffu;
            const u32 sel = (t | (t << 4)) & 0x0f0f0f0fu;    // one code per byte, in output order
            const u32 lo = __builtin_amdgcn_perm(m1, m0, sel & 0x07070707u), hi = __builtin_amdgcn_perm(m3, m2, sel & 0x07070707u);
            const u32 up = ((sel >> 3) & 0x01010101u) * 0xffu;
            return (hi & up) | (lo & ~up);
        };
        auto expand = [&](const u32x2 w) -> u32x4 { return u32x4{four(w.x & 0xffffu), four(w.x >> 16), four(w.y & 0xffffu), four(w.y >> 16)}; };
        u32x2 wq[4] = {ld2(lane), ld2(lane + NT), ld2(lane + 2 * NT), ld2(lane + 3 * NT)};
        u32x4 rq[4] = {};
        u32 rt = trips;
        for (u32 t0 = lane; t0 < trips; t0 += 4 * NT) {
            __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): wq[] is here, the last trip's stores are out
            __builtin_amdgcn_sched_barrier(0);
            if (rt < trips) {
#pragma unroll
                for (u32 qi = 0; qi < 4; qi++) if (rt + qi * NT < trips) *(GAS u32x4_unaligned *)(gout + 16ull * (rt + qi * NT)) = rq[qi];
            }
            u32x2 nq[4];
#pragma unroll
            for (u32 qi = 0; qi < 4; qi++) nq[qi] = ld2(t0 + (4 + qi) * NT);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (u32 qi = 0; qi < 4; qi++) { rq[qi] = expand(wq[qi]); wq[qi] = nq[qi]; }
            rt = t0;
        }
        if (rt < trips) {
#pragma unroll
            for (u32 qi = 0; qi < 4; qi++) if (rt + qi * NT < trips) *(GAS u32x4_unaligned *)(gout + 16ull * (rt + qi * NT)) = rq[qi];
        }
        for (u32 i = trips * 16u + lane; i < out_len; i += NT)
            out[i] = B.map[(data[i / per] >> ((i % per) * width)) & mask];
        return true;
    }
    const u32 ndw = out_len >> 2;
    for (u32 w = lane; w < ndw; w += NT) {
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 i = 4 * w + j;
            v |= (u32)B.map[(data[i / per] >> ((i % per) * width)) & mask] << (8 * j);
        }
        *(u32_unaligned *)(out + 4 * (u64)w) = v;
    }
    const u32 i = 4 * ndw + lane;
    if (i < out_len) out[i] = B.map[(data[i / per] >> ((i % per) * width)) & mask];
    return true;
}

// ---------------------------------------------------------------------------------------------
// k_dec_back: CAT copies, un-RLE, un-PACK, final size and status (:1576-1629).
// ---------------------------------------------------------------------------------------------
// NT = 64: one wave per block (large batches); NT = BACK_THREADS: a workgroup per block (small batches).
template <u32 NT>
__global__ __launch_bounds__(NT) void k_dec_back(BatchArgs a, DecWs ws, int base)
{
    __shared__ BackShared B;
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    const DecDesc *D = &ws.desc[b];
    i32 st = D->status;
    u32 size = 0;
    if (st == ST_OK) {
        const u32 flags = D->flags;
        u8 *s1 = (u8 *)D->s1, *s2 = (u8 *)D->s2, *s3 = (u8 *)D->s3;
        u32 s1_size = D->s1_size;
        if (D->cat_src) group_copy<NT>(s1, (const u8 *)D->cat_src, D->cat_len, lane);
        if (flags & (X_PACK | X_RLE)) { wg_fence(); __syncthreads(); }
        u32 s2_size = s1_size;
        if (flags & X_RLE) {                                           // :1598-1613
            const u8 *meta = (const u8 *)D->rle_meta;
            const u32 mlen = D->rle_meta_len;
            u32 nsyms = 0;
            if (mlen == 0) st = ST_RLE;
            else {
                nsyms = meta[0] ? meta[0] : 256u;
                if (mlen < 1 + nsyms) st = ST_RLE;
            }
            if (st == ST_OK) {
                u32 produced = 0;
                bool ok;
                if constexpr (NT == WAVE) ok = rle_expand_wave(s1, s1_size, meta + 1 + nsyms, mlen - (1 + nsyms), meta + 1, nsyms, s2, D->osz, produced, B, lane);
                else ok = rle_expand_wg(s1, s1_size, meta + 1 + nsyms, mlen - (1 + nsyms), meta + 1, nsyms, s2, D->osz, produced, B, lane);
                if (!ok)
                    st = ST_RLE;
                else
                    s2_size = produced;
            }
            wg_fence();
            __syncthreads();
        }
        size = s2_size;
        if (st == ST_OK && (flags & X_PACK)) {                         // :1614-1623
            if (lane < 16) B.map[lane] = D->pack_map[lane];
            __syncthreads();
            const u32 per = D->pack_per;
            const u32 unpacked = per == 1 ? s2_size : D->osz;
            if (!unpack<NT>(s2, s2_size, s3, unpacked, per, B, lane)) st = ST_SIZE;
            size = unpacked;
        }
    }
    if (lane == 0) {
        a.status[i] = st;
        a.out_size[i] = st == ST_OK ? size : 0;
    }
}

// Where each block's staging region for the inverse transforms starts: blocks whose flag byte carries X_PACK or X_RLE
// get dec_var_bytes(min(capacity, the caller's bound)), the others nothing; r4x16_voff_scan makes offsets of the sizes.
__device__ __forceinline__ u64 dec_var_of(const BatchArgs &a, int i, u32 max_out_cap)
{
    if (!a.in_size[i]) return 0ull;
    const u8 flags = a.in[a.in_off[i]];
    if ((flags & X_STRIPE) || !(flags & (X_PACK | X_RLE))) return 0ull;
    const u32 cap = a.out_cap[i];
    return dec_var_bytes(cap < max_out_cap ? cap : max_out_cap);
}
__global__ __launch_bounds__(256) void k_dec_vsize(BatchArgs a, int base, int nblk, u64 *voff, u32 max_out_cap)
{
    const int b = (int)(blockIdx.x * 256u + threadIdx.x);
    if (b < nblk) voff[b] = dec_var_of(a, base + b, max_out_cap);
}

// ---- host-callable launchers (r4x16_api.hip) ---------------------------------------------------
static void launch_dec_chain_of(const DecWs *ws, const DecItem *items, int nitems, bool one_row_only, hipStream_t s, const R4Fork *fk, const R4Opts *o, SchedHint *hint);
extern "C" void r4x16_launch_dec_front(const BatchArgs *a, const DecWs *ws, int base, int nblk, hipStream_t s, const R4Opts *o)
{
    if (ws->var) {
        hipLaunchKernelGGL(k_dec_vsize, dim3((nblk + 255) / 256), dim3(256), 0, s, *a, base, nblk, ws->voff, ws->max_out_cap);
        r4x16_voff_scan(ws->voff, nblk, s);
    }
    hipLaunchKernelGGL(k_dec_front<0>, dim3(nblk), dim3(WAVE), 0, s, *a, *ws, base);
    launch_dec_chain_of(ws, ws->items + 2 * (size_t)nblk, nblk, true, s, nullptr, o, nullptr);        // nested order-1 tables
    hipLaunchKernelGGL(k_dec_front<1>, dim3(nblk), dim3(WAVE), 0, s, *a, *ws, base);
}
// LDS size classes: {bytes per stream (image + word ring), streams per wave, tree depth}.
// LDS is allocated in 1,280-byte granules; streams per CU = floor(160 KB / granules(qpw * bytes)) * qpw.
// Sizes are 16 mod 128, so that consecutive streams start four LDS banks apart.  For the 46-symbol
// quality alphabets three waves of ten streams measured best (2 x 15: -8 %, 4 x 7: -12 %, 5 x 6: -24 %).
// 2-read images are at most 6 KB, 4-level
// images at least 21 KB (or the lone 1.3 KB row of a large order-0 alphabet), so the two groups of
// classes are walked separately.
static const struct { u32 bytes; int qpw; int lv; } DEC_CLASSES[] = {
    // packed rows (level 1): 13..36 symbols in rows of up to 56 bytes, 37..48 of up to 72 (46 symbols: 68-byte rows,
    // 3,496 bytes with alphabet and ring: 3 x 15 streams per CU)
    // (layout 2: 46 symbols = 3,312 bytes of image, 3,584 with the ring: 15 streams = 42 LDS granules of 1,280 bytes exactly)
    {1424, 16, 1}, {2448, 16, 1}, {3360, 16, 1}, {3584, 15, 1}, {3840, 14, 1}, {4128, 13, 1}, {4880, 11, 1},
    // wide packed rows (level 5): 49..96 symbols, rows of 84..148 bytes; three workgroups per CU
    {5520, 9, 5}, {7184, 7, 5}, {8912, 6, 5}, {10640, 5, 5}, {13200, 4, 5}, {14736, 3, 5},
    {656, 16, 2}, {1296, 16, 2}, {2576, 16, 2}, {3856, 16, 2}, {5008, 16, 2}, {5360, 10, 2}, {5392, 15, 2}, {6416, 12, 2},
    // 51..150 symbols, 3 reads: one-row order-0 images, then order-1 images of 9..55 KB (one stream per wave,
    // as many waves per CU as LDS granules allow)
    {1296, 16, 3}, {10256, 1, 3}, {12816, 1, 3}, {16656, 1, 3}, {20496, 1, 3}, {25616, 1, 3}, {32016, 1, 3},
    {40976, 1, 3}, {53648, 1, 3}, {64016, 1, 3},
    {22528, 1, 4}, {32768, 1, 4}, {53248, 1, 4}, {81920, 1, 4}, {163840, 1, 4},
    // direct blocks (level 6; only batches that leave LDS to spare make such images, r4x16_dec_direct_budget): four
    // workgroups per CU and more - a wave per SIMD at least - with up to eight streams per wave, then one stream per wave.  Few
    // classes on purpose: every class is a launch, classes with streams run one after the other, and a batch that mixes
    // alphabets (q4 / q8 / q40) should not fall into more classes than it did with the compressed rows
    // (at most eight streams per wave: the step of this loop is LDS round trips, and a wave with more than 32 live
    //  lanes pays for both halves - 4,096 x 1 MiB q8 literal streams: 481 cycles per step at nine streams per wave,
    //  372 at eight; the compressed rows' loop gains 2 % from the same change.  Seven per wave and five workgroups per CU
    //  - 35 streams instead of 32 - put two waves on one SIMD: 26.1 ms again, and 8,192 q8 streams 39.5 instead of 33.5)
    {2576, 8, 6}, {4112, 8, 6}, {8080, 5, 6}, {13584, 3, 6}, {20368, 2, 6}, {32000, 1, 6}, {40960, 1, 6}, {53760, 1, 6},
    {81920, 1, 6}, {163840, 1, 6},
    // the same for order-0 streams (level 7 = level 6's kernel; images of at most 2,820 bytes).  A wave walks its order-1
    // quads, then its order-0 quads: the run lengths of a block with X_RLE in a launch of their own run beside the
    // block's literals where launches run side by side (one 1 MiB q8 block: 25.0 -> 18.7 ms), not behind them in the
    // same wave (only where launches do run side by side, DecClassTab.split_o0)
    {2576, 8, 7}, {4112, 8, 7},
    // order-0 streams with compressed rows of up to 50 symbols (level 8 = level 2's kernel; such an image and its ring
    // take at most 516 bytes).  The kernel runs a wave's order-1 streams, then its order-0 streams: in a class that
    // holds both kinds - q4 / q8 order-1 tables are this small too - a wave of long streams of both kinds took twice
    // the chain latency (round 4's heterogeneous batch: the class of its 1 MiB streams 104 ms instead of 50).
    {656, 16, 8},
    // packed rows of 43..46 symbols with the short ring (level 9 = level 1's rows, four-step trips: chain_decode_lds):
    // 3,224 bytes of image + 136 of ring = 3,360, 16 x 3,360 = 42 LDS granules exactly - three workgroups of SIXTEEN
    // streams per CU where the long ring allows fifteen
    {3360, 16, 9},
    // mid rows (level 10; only batches of one partly filled round make such images, r4x16_dec_mid_budget): four streams
    // per wave - the loop is two LDS round trips per step, few lanes per access - and four waves per CU, one per SIMD:
    // sixteen streams of up to 48 symbols (8,816 bytes) per CU, 4,096 per chip
    {8976, 4, 10},
};
// workgroups of `lds_bytes` each that one CU holds at once (1,280-byte LDS granules, 32 wave slots)
static int resident_per_cu(size_t lds_bytes, int waves_per_wg)
{
    const int granules = (int)((lds_bytes + 1279) / 1280);
    int n = granules ? 128 / granules : 32;
    if (n * waves_per_wg > 32) n = 32 / waves_per_wg;
    return n < 1 ? 1 : n;
}
// Per-device state of the launchers: a process may hold contexts on several devices, on several host threads.
#define MAX_DEVICES 64
static std::mutex g_dev_mu;
static int g_cu_count[MAX_DEVICES];
static u32 g_setup_done[MAX_DEVICES];
static int cu_count()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= MAX_DEVICES) return 256;
    std::lock_guard<std::mutex> g(g_dev_mu);
    int &n = g_cu_count[dev];
    if (!n && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)) n = 256;
    return n;
}
extern "C" int r4x16_cu_count(void) { return cu_count(); }
// true exactly once per (current device, bit): kernel attributes such as the dynamic-LDS limit are per device
extern "C" bool r4x16_first_on_device(u32 bit)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= MAX_DEVICES) return true;
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (g_setup_done[dev] & bit) return false;
    g_setup_done[dev] |= bit;
    return true;
}
static void lds_limit(const void *kernel, int bytes)
{
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) fprintf(stderr, "rans4x16_hip: cannot raise a kernel's dynamic LDS limit to %d bytes: %s\n", bytes, hipGetErrorString(e));
}
extern "C" int r4x16_resident_grid(size_t lds_bytes, int waves_per_wg, int wanted)
{
    const int cap = cu_count() * resident_per_cu(lds_bytes, waves_per_wg);
    return wanted < cap ? wanted : cap;
}
// ---- streams -> classes, on the device ---------------------------------------------------------
// class ids: index into DEC_CLASSES, then one catch-all per tree depth (images too large for LDS)
#define DEC_NCLS ((u32)(sizeof(DEC_CLASSES) / sizeof(DEC_CLASSES[0])))
struct DecClassTab { u32 n; u32 split_o0; u32 sort; u32 short_ring; u32 bytes[CLS_MAX]; u32 lv[CLS_MAX]; };
// class, length bucket and the class's work per item (r4x16_sched.h); per-class counts go through LDS first: a whole
// batch is usually one class, and 30,000 atomics on one global word took 0.18 ms
__global__ __launch_bounds__(256) void k_dec_classify(const DecItem *items, int nitems, DecClassTab tab, SchedWs sw)
{
    __shared__ u32 local[CLS_MAX];
    __shared__ u64 lwork[2 * CLS_MAX];
    if (threadIdx.x < CLS_MAX) { local[threadIdx.x] = 0; lwork[threadIdx.x] = 0ull; lwork[CLS_MAX + threadIdx.x] = 0ull; }
    __syncthreads();
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    u32 c = CLS_NONE, len = 0;
    if (i < nitems) {
        const DecItem *I = &items[i];
        if (I->active) {
            const u32 need = I->img_bytes + RING_BYTES, lv0 = item_levels(I->nsym, I->packed), lv = (lv0 == 6u && I->order == 0 && tab.split_o0) ? 7u : (lv0 == 2u && I->order == 0) ? 8u :
                                                                 (lv0 == 1u && tab.short_ring && need > 3344u && I->img_bytes + RING_BYTES_SHORT <= 3360u) ? 9u : lv0;
            c = tab.n + ((lv < 2u || lv > 4u) ? 0u : lv - 2u);  // catch-all of this depth (packed levels 1 and 5 always fit a class)
            for (u32 k = 0; k < tab.n; k++)
                if (tab.lv[k] == lv && (lv == 9u ? I->img_bytes + RING_BYTES_SHORT : need) <= tab.bytes[k]) { c = k; break; }   // classes of a depth ascend
            len = I->out_sz;
        }
    }
    sched_classify(sw, i, i < nitems, c, len, tab.sort != 0, local, lwork);
    __syncthreads();
    sched_classify_flush(sw, local, lwork);
}
__global__ void k_cls_zero(u32 *count) { if (threadIdx.x < CLS_MAX) count[threadIdx.x] = 0; }
__global__ void k_cls_scan(u32 *count)
{
    if (threadIdx.x == 0) {
        u32 at = 0;
        for (u32 c = 0; c < CLS_MAX; c++) { count[CLS_MAX + c] = at; at += count[c]; count[2 * CLS_MAX + c] = 0; }
    }
}
__global__ __launch_bounds__(256) void k_cls_scatter(const u32 *cls, int nitems, u32 *count, u32 *list)
{
    __shared__ u32 local[CLS_MAX], base[CLS_MAX];
    if (threadIdx.x < CLS_MAX) local[threadIdx.x] = 0;
    __syncthreads();
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const u32 c = i < nitems ? cls[i] : CLS_NONE;
    u32 rank = 0;
    if (c != CLS_NONE) rank = atomicAdd(&local[c], 1u);        // rank inside this workgroup
    __syncthreads();
    if (threadIdx.x < CLS_MAX && local[threadIdx.x])           // one reservation per class and workgroup
        base[threadIdx.x] = count[CLS_MAX + threadIdx.x] + atomicAdd(&count[2 * CLS_MAX + threadIdx.x], local[threadIdx.x]);
    __syncthreads();
    if (c != CLS_NONE) list[base[c] + rank] = (u32)i;
}

// positions and lists from per-item classes and per-class counts (also used by the encoder's launcher)
extern "C" void r4x16_launch_cls_zero(u32 *count, hipStream_t s)     // (a kernel, not hipMemsetAsync: the runtime's first
{                                                                      //  memset on a stream cost 0.5 s once per process)
    hipLaunchKernelGGL(k_cls_zero, dim3(1), dim3(64), 0, s, count);
}
extern "C" void r4x16_launch_cls_group(const u32 *cls, int nitems, u32 *count, u32 *list, hipStream_t s)
{
    hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(64), 0, s, count);
    hipLaunchKernelGGL(k_cls_scatter, dim3((nitems + 255) / 256), dim3(256), 0, s, cls, nitems, count, list);
}
extern "C" void r4x16_launch_dec_chain(const DecWs *ws, int nitems, hipStream_t s, const R4Fork *fk, const R4Opts *o, SchedHint *hint) { launch_dec_chain_of(ws, ws->items, nitems, false, s, fk, o, hint); }
static int dec_class_qpw(u32 ci, const R4Opts *o)
{
    const auto &c = DEC_CLASSES[ci];
    const int force_qpw = (int)o->v[OPT_DEC_QPW], force_small = (int)o->v[OPT_DEC_QPW_SMALL], force_pk = (int)o->v[OPT_DEC_QPW_PK],
              force_dir = (int)o->v[OPT_DEC_QPW_DIR];                                                    // tuning aids
    int qpw = (c.lv == 1 && force_pk && (c.bytes == 3496 || c.bytes == 3584)) ? force_pk :
              c.lv != 2 ? c.qpw : (force_qpw && c.bytes == 5360) ? force_qpw : (force_small && c.bytes < 5360) ? force_small : c.qpw;
    if ((c.lv == 6 || c.lv == 7) && force_dir > 0 && qpw > force_dir) qpw = force_dir;
    return qpw;
}
// one_row_only: the items are order-0 streams (one-row images): only the classes such an image can fall into are launched
static void launch_dec_chain_of(const DecWs *ws, const DecItem *items, int nitems, bool one_row_only, hipStream_t s0, const R4Fork *fk, const R4Opts *o, SchedHint *hint)
{
    // Classes side by side (r4x16_sched.h, PLAN): the class launches are dealt out over the caller's stream and the side
    // streams (fk), and the device-written plan gives each class its stream's share of the chip.  Without side streams
    // (a lane of the host pipeline, option sched_concurrent = 0) the launches go out in stream order, every class with
    // the whole chip - as up to round 3.
    const int nq = fk ? fk->n + 1 : 1;
    typedef void (*chain_fn)(const DecItem *, DecDesc *, const u32 *, u32 *, int, u32, int);
    struct Launch { chain_fn kern; int grid, qpw; size_t ldsb; u32 ci, bytes; };
    Launch todo[CLS_MAX];
    int ntodo = 0;
    SchedPlan plan;
    DecClassTab tab;
    tab.n = DEC_NCLS;
    tab.split_o0 = fk != nullptr;           // (launches in stream order: a wave takes both kinds, 4,096 x 1 MiB q8 with X_RLE 26.7 against 28.2 ms)
    tab.sort = o->v[OPT_SCHED_SORT] != 0;
    tab.short_ring = o->v[OPT_DEC_SHORT_RING] != 0;
    plan.ncls = DEC_NCLS; plan.concurrent = nq > 1 ? (u32)o->v[OPT_SCHED_CONCURRENT] : 0u; plan.claim = o->v[OPT_SCHED_CLAIM] != 0; plan.pad = 0;
    for (u32 ci = 0; ci < CLS_MAX; ci++) { plan.qpw[ci] = 16; plan.wgs_full[ci] = 0; plan.queue[ci] = 0xff; plan.rate[ci] = 0.f; }
    for (u32 ci = 0; ci < DEC_NCLS; ci++) {
        const auto &c = DEC_CLASSES[ci];
        tab.bytes[ci] = c.bytes; tab.lv[ci] = (u32)c.lv;
        const int qpw = dec_class_qpw(ci, o);
        const size_t ldsb = (size_t)qpw * c.bytes;
        plan.qpw[ci] = (u16)qpw;
        plan.wgs_full[ci] = (u16)(cu_count() * resident_per_cu(ldsb, 1));
        plan.rate[ci] = sched_rate(qpw, 1, resident_per_cu(ldsb, 1), cu_count());
        // (an order-0 image: at most IMG_O0_BYTES; depth 4 only as the lone row of an alphabet beyond 150 symbols)
        const bool skip = (c.lv == 9 && !tab.short_ring) || (c.lv == 10 && !ws->mid_budget) ||
                          (one_row_only && (c.lv == 1 || c.lv == 2 || (c.lv >= 5 && c.lv != 8) || c.bytes > (c.lv == 4 ? 22528u : IMG_O0_BYTES + RING_BYTES + 128u))) ||
                          ((c.lv == 6 || c.lv == 7) && !ws->direct_budget);      // (no stream of this batch was given direct blocks)
        if (skip) continue;
        chain_fn kern =
            c.lv == 10 ? k_dec_chain<true, 10> : c.lv == 9 ? k_dec_chain<true, 1, 4> : c.lv == 1 ? k_dec_chain<true, 1> : c.lv == 5 ? k_dec_chain<true, 5> : (c.lv == 2 || c.lv == 8) ? k_dec_chain<true, 2> : c.lv == 3 ? k_dec_chain<true, 3> :
            (c.lv == 6 || c.lv == 7) ? k_dec_chain<true, 6> : k_dec_chain<true, 4>;
        todo[ntodo++] = Launch{kern, r4x16_resident_grid(ldsb, 1, (nitems + qpw - 1) / qpw), qpw, ldsb, ci, c.bytes};
    }
    u8 qof[CLS_MAX];
    int lorder[CLS_MAX];
    {
        int cls_of[CLS_MAX];
        for (int k = 0; k < ntodo; k++) cls_of[k] = (int)todo[k].ci;
        if (hint) hint->learn = (o->v[OPT_SCHED_LEARN] & 2) != 0;
        sched_assign_queues(plan, cls_of, ntodo, nq, hint, qof, lorder, (hint && hint->work && o->v[OPT_SCHED_TRACE]) ? "decode" : nullptr);
        for (int k = 0; k < ntodo; k++) plan.queue[todo[k].ci] = qof[k];
    }
    r4x16_sched_zero(&ws->sched, s0);
    hipLaunchKernelGGL(k_dec_classify, dim3((nitems + 255) / 256), dim3(256), 0, s0, items, nitems, tab, ws->sched);
    r4x16_sched_group(&ws->sched, nitems, &plan, s0);
    if (r4x16_first_on_device(1u)) {
        lds_limit((const void *)k_dec_chain<true, 1>, 163840);
        lds_limit((const void *)k_dec_chain<true, 1, 4>, 163840);
        lds_limit((const void *)k_dec_chain<true, 10>, 163840);
        lds_limit((const void *)k_dec_chain<true, 5>, 163840);
        lds_limit((const void *)k_dec_chain<true, 2>, 163840);
        lds_limit((const void *)k_dec_chain<true, 3>, 163840);
        lds_limit((const void *)k_dec_chain<true, 4>, 163840);
        lds_limit((const void *)k_dec_chain<true, 6>, 163840);
    }
    const int dyn = o->v[OPT_SCHED_CLAIM] != 0;
    auto go = [&](const Launch &L, hipStream_t s) {
        DecDesc *desc = ws->desc;
        const u32 *list = ws->sched.list;
        u32 *cnt = ws->sched.cnt + L.ci;
        int qpw = L.qpw, dyn_ = dyn;
        u32 bytes = L.bytes;
        void *args[] = {(void *)&items, (void *)&desc, (void *)&list, (void *)&cnt, (void *)&qpw, (void *)&bytes, (void *)&dyn_};
        r4x16_sched_launch((const void *)L.kern, dim3(L.grid), dim3(WAVE), args, L.ldsb, s);
    };
    unsigned used = 0;
    for (int k = 0; k < ntodo; k++) used |= 1u << (qof[k] % (unsigned)nq);
    if (fk) fk->begin(s0, used);
    for (int j = 0; j < ntodo; j++) { const int k = lorder[j]; go(todo[k], fk ? fk->pick(s0, (unsigned)qof[k]) : s0); }
    if (fk) { fk->end(s0, used); r4x16_sched_hint_save(&ws->sched, hint, s0); }
    if (one_row_only) return;                 // (such an image always fits a class)
    // images that fit no LDS class: tables stay in global memory (L2); after the join, in stream order
    const int grid = (nitems + 15) / 16;
    go(Launch{k_dec_chain<false, 2>, grid, 16, 0, DEC_NCLS + 0, 0u}, s0);
    go(Launch{k_dec_chain<false, 3>, grid, 16, 0, DEC_NCLS + 1, 0u}, s0);
    go(Launch{k_dec_chain<false, 4>, grid, 16, 0, DEC_NCLS + 2, 0u}, s0);
}
// LDS bytes a stream may spend on direct blocks (level 6) when `nblk` streams are to be resident at once: the largest
// direct class that still holds the batch in ONE round of the chip (0: none does - the batch is large enough to be
// bound by resident streams, which is what the compressed rows are for).
//   R4X16_DEC_DIRECT=0  never;  =N (N >= 1)  accept up to N rounds of direct streams (default 1)
extern "C" u32 r4x16_dec_direct_budget(int nblk, const R4Opts *o)
{
    const int rounds = (int)o->v[OPT_DEC_DIRECT];
    if (rounds <= 0 || nblk <= 0) return 0u;
    const long cus = cu_count();
    const long per_cu = (nblk + cus * rounds - 1) / (cus * rounds);
    u32 best = 0;
    for (const auto &c : DEC_CLASSES)
        if (c.lv == 6 && (long)resident_per_cu((size_t)c.qpw * c.bytes, 1) * c.qpw >= per_cu && c.bytes > best) best = c.bytes;
    return best;
}
// LDS bytes a stream may spend on mid rows (level 10): the mid class's, if `nblk` streams fit `dec_mid` rounds of it
// (sixteen per CU), else 0.
extern "C" u32 r4x16_dec_mid_budget(int nblk, const R4Opts *o)
{
    const long rounds = o->v[OPT_DEC_MID];
    if (rounds <= 0 || nblk <= 0) return 0u;
    for (const auto &c : DEC_CLASSES)
        if (c.lv == 10 && (long)nblk <= rounds * cu_count() * resident_per_cu((size_t)c.qpw * c.bytes, 1) * c.qpw) return c.bytes;
    return 0u;
}
// Streams of one kind that a CU holds at once in the chain decoder (host arithmetic on the class table above).
extern "C" int r4x16_dec_residency(u32 nsym, int order, u32 bits, int *streams_per_wave, int *waves_per_cu, int short_ring)
{
    if (nsym == 0 || nsym > 256) return -1;
    const bool packed = order == 1 && bits == 10 && nsym >= PK_MIN_NSYM && nsym <= PKW_MAX_NSYM;
    u32 lv = packed ? item_levels(nsym, 1u) : (order == 0 && img_levels(nsym) == 2u) ? 8u : item_levels(nsym, 0u);
    const u32 img = packed ? pk_img_bytes(nsym) : img_bytes(nsym, order ? nsym : 1u);
    u32 need = img + RING_BYTES;
    if (lv == 1u && short_ring && need > 3344u && img + RING_BYTES_SHORT <= 3360u) { lv = 9u; need = img + RING_BYTES_SHORT; }
    for (const auto &c : DEC_CLASSES) {
        if ((u32)c.lv != lv || need > c.bytes) continue;
        *streams_per_wave = c.qpw;
        *waves_per_cu = resident_per_cu((size_t)c.qpw * c.bytes, 1);
        return 0;
    }
    *streams_per_wave = 16; *waves_per_cu = 8;              // tables in global memory: bounded by wave slots
    return 0;
}
extern "C" void r4x16_launch_dec_back(const BatchArgs *a, const DecWs *ws, int base, int nblk, hipStream_t s, const R4Opts *o)
{
    // One wave per block at every batch size since its trips became 256 literals of LDS-fed work (64 x 1 MiB q8 blocks with
    // X_RLE: step 38.0 ms either way; 1,024: 53.5 against 55.6; 2,048: 56.2 against 64.6); the workgroup-per-block
    // kernel, round 3's first answer to the old trip's fixed ~19 ms per MiB, stays selectable: R4X16_BACK_WG_PER_CU=N
    // takes it up to N blocks per CU.
    const int wg_per_cu = (int)o->v[OPT_BACK_WG_PER_CU];
    if (nblk <= wg_per_cu * cu_count()) hipLaunchKernelGGL(k_dec_back<BACK_THREADS>, dim3(nblk), dim3(BACK_THREADS), 0, s, *a, *ws, base);
    else hipLaunchKernelGGL(k_dec_back<WAVE>, dim3(nblk), dim3(WAVE), 0, s, *a, *ws, base);
}

// ---------------------------------------------------------------------------------------------
// X_STRIPE byte-plane transposition (rANS_static4x16pr.c:1168-1180 and unstripe, utils.h:41-73).
// Plane j holds bytes j, j+N, j+2N, ...; first[j] is its offset in the plane buffer.
// ---------------------------------------------------------------------------------------------
__global__ void k_stripe_split(const u8 *in, u8 *planes, u32 n, u32 N)
{
    const u32 base = n / N, extra = n % N;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u32 j = i % N, x = i / N;
        const u32 first = j * base + (j < extra ? j : extra);
        planes[first + x] = in[i];
    }
}
__global__ void k_stripe_join(const u8 *planes, u8 *out, u32 n, u32 N)
{
    const u32 base = n / N, extra = n % N;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u32 j = i % N, x = i / N;
        const u32 first = j * base + (j < extra ? j : extra);
        out[i] = planes[first + x];
    }
}
extern "C" void r4x16_launch_stripe(const u8 *src, u8 *dst, u32 n, u32 N, int join, hipStream_t s)
{
    if (!n) return;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (join) hipLaunchKernelGGL(k_stripe_join, dim3(grid), dim3(256), 0, s, src, dst, n, N);
    else      hipLaunchKernelGGL(k_stripe_split, dim3(grid), dim3(256), 0, s, src, dst, n, N);
}

// =============================================================================================
// rANS 4x8 decode (CRAM 3.0's codec: htscodecs/rANS_static.c:231-383, :676-916, rANS_byte.h).
// the same four interleaved chains per block, 12-bit tables, but BYTE-wise
// renormalisation (L = 2^23, up to two bytes per chain and step) and its own table grammar.
//
//   k8_dec_front : one wave per block: the 9-byte header, the frequency table(s) -> a lookup image in
//                  the 4x16 decoder's u16 format (r4x16_common.h) with one extra, flagged symbol for the
//                  slot a 4095-sum table leaves unowned.
//   k8_dec_chain : a quad per block, 16 blocks per wave; the general-form chain loop (tables read
//                  through L2) with the byte renormalisation: a chain takes (x < 2^23) + (x < 2^15) bytes
//                  - both known before any byte is read - at the quad's cursor plus the counts of the
//                  chains below it.
// Plain first version of this codec (no LDS images, no word ring): correctness and the boundary first.
//
// Stricter than the reference on damaged input (valid encoder output never gets here), all reported as
// R4X16_E_TABLE / R4X16_E_CONTEXT: symbols or contexts not listed in ascending order (the reference assigns
// slots in listing order, which a cumulative table over the sorted alphabet cannot express); slot 4095 of a
// table that sums to 4095 and contexts without a table (undefined in the reference, see oracle/rans4x8_oracle.c).
// =============================================================================================
#define X8_LOW   (1u << 23)
#define X8_BITS  12u
#define IMG8_MAX_NSYM 257u                                // 256 symbols + the flagged one
#define IMG8_SLOT ((528u + 257u * 832u + 255u) & ~255u)   // alpha[257] + 257 four-level rows

struct X8Item {
    u64 bytes;       // first renormalisation byte (just after the 4 states)
    u64 out;
    u64 image;
    u32 bytes_len;   // bytes from `bytes` to the end of the input
    u32 out_sz;
    u32 R[4];
    u32 order, nsym; // nsym counts the flagged extra symbol
    u32 active, pad;
};

struct Front8Shared {
    FrontShared S;
    u32 Fsym[256];       // frequency by byte value of the table being parsed (0 = not listed)
    u8 listed[256];      // listed in the table being parsed
    u8 ctx_seen[256];    // order-1: context has a table
    i32 status;
    u32 pos, total, n, go;
};

// One table of the 4x8 grammar (rANS_static.c:274-310 / :760-805) by one lane: symbols with a run-length
// shortcut, each followed by a one- or two-byte frequency, closed by a zero symbol.  Fills Fsym / listed;
// returns the new position or 0 with `st` set.  `collect` != nullptr: only mark the symbols seen (first pass
// of the order-1 parse, which needs the whole alphabet before it can lay out rows).
__device__ u32 x8_get_table(ByteSrc &src, u32 cp, u32 end, u32 *Fsym, u8 *listed, u32 *total, bool zero_is_total,
                            u8 *collect, i32 &st)
{
    u32 x = 0, rle = 0, prev = 0;
    bool first = true;
    u32 j = src.at(cp++);
    do {
        if (cp + 16 > end) { st = ST_TRUNCATED; return 0; }
        u32 F = src.at(cp++);
        if (F >= 128) F = ((F & 127) << 8) | src.at(cp++);
        if (!F && zero_is_total) F = 4096u;
        if (x + F > 4096u) { st = ST_TABLE; return 0; }
        if (!first && j <= prev) { st = ST_TABLE; return 0; }          // ascending listing only (see the header)
        first = false; prev = j;
        if (collect) collect[j] = 1;
        else { Fsym[j] = F; listed[j] = 1; }
        x += F;
        const u32 nx = src.at(cp);
        if (!rle && j + 1 == nx) { j = src.at(cp++); rle = src.at(cp++); }
        else if (rle) { rle--; j++; if (j > 255) { st = ST_TABLE; return 0; } }
        else j = src.at(cp++);
    } while (j);
    if (x < 4095u || x > 4096u) { st = ST_TABLE; return 0; }
    *total = x;
    return cp;
}

// Whole wave: cumulative starts of the current table over the compact alphabet S.alpha[0..n) (the last entry is
// the flagged extra symbol, which owns [total, 4096)), then the row.
__device__ void x8_build_row(Front8Shared &Z, u8 *rowp, u32 n, u32 lane)
{
    FrontShared &S = Z.S;
    u32 carry = 0;
    for (u32 cb = 0; cb < n; cb += WAVE) {
        const u32 c = cb + lane;
        const u32 f = (c + 1 < n) ? Z.Fsym[S.alpha[c]] : 0u;
        const u32 incl = wave_incl_scan(f, lane);
        if (c < n) S.cum[c] = (u16)(carry + incl - f);
        carry += (u32)__shfl((int)incl, WAVE - 1);
    }
    if (lane == 0) { S.cum[n - 1] = (u16)Z.total; S.cum[n] = 4096; S.cum[n + 1] = S.cum[n + 2] = S.cum[n + 3] = 0x7fffu; }
    __syncthreads();
    write_row(rowp, S, n, false, lane);
    __syncthreads();
}

__global__ __launch_bounds__(WAVE) void k8_dec_front(BatchArgs a, X8Item *items, u8 *images, int base)
{
    __shared__ Front8Shared Z;
    FrontShared &S = Z.S;
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    X8Item *I = &items[b];
    const u8 *in = a.in + a.in_off[i];
    const u32 in_size = a.in_size[i];
    u8 *img = images + (u64)b * IMG8_SLOT;
    ByteSrc src(in);

    for (u32 j = lane; j < 256; j += WAVE) { Z.Fsym[j] = 0; Z.listed[j] = 0; Z.ctx_seen[j] = 0; S.present[j] = 0; }
    if (lane == 0) {
        I->active = 0;
        i32 st = ST_OK;
        u32 order = 0, out_sz = 0;
        if (in_size == 0) st = ST_EMPTY;
        else if (in_size < 9) st = ST_TRUNCATED;                                 // rANS_static.c:937
        else {
            order = src.at(0);
            if (order > 1) st = ST_UNSUPPORTED;                                  // :715 (anything non-zero goes to the order-1 decoder, which wants 1)
            else if (in_size < (order ? 27u : 26u)) st = ST_TRUNCATED;           // :245, :711
            else {
                const u32 in_sz = (u32)src.at(1) | ((u32)src.at(2) << 8) | ((u32)src.at(3) << 16) | ((u32)src.at(4) << 24);
                out_sz = (u32)src.at(5) | ((u32)src.at(6) << 8) | ((u32)src.at(7) << 16) | ((u32)src.at(8) << 24);
                if (in_sz != in_size - 9) st = ST_SIZE;                          // :252
                else if (out_sz >= 0x7fffffffu) st = ST_SIZE;                    // :255
                else if (a.out_cap[i] < out_sz) st = ST_CAPACITY;
            }
        }
        Z.status = st;
        I->order = order; I->out_sz = out_sz;
    }
    __syncthreads();
    if (Z.status != ST_OK) { if (lane == 0) { a.status[i] = Z.status; a.out_size[i] = 0; } return; }
    const u32 order = I->order, end = in_size;

    if (order == 0) {
        if (lane == 0) {
            i32 st = ST_OK;
            const u32 cp = x8_get_table(src, 9, end, Z.Fsym, Z.listed, &Z.total, false, nullptr, st);
            u32 n = 0;
            if (st == ST_OK) {
                for (u32 j = 0; j < 256; j++) if (Z.listed[j]) S.alpha[n++] = (u8)j;
                S.alpha[n++] = 0;                                                // the flagged extra symbol
                if (cp + 16 > end) st = ST_TRUNCATED;                            // :313
            }
            Z.status = st; Z.pos = cp; Z.n = n;
        }
        __syncthreads();
        if (Z.status == ST_OK) {
            const u32 n = Z.n;
            for (u32 j = lane; j < n; j += WAVE) ((u16 *)img)[j] = (u16)(S.alpha[j] | (j + 1 == n ? ROW_BAD : 0u));
            x8_build_row(Z, img + img_alpha_bytes(n), n, lane);
        }
    } else {
        // pass 1 (lane 0): the alphabet = every byte listed as a context or as a symbol, byte 0 forced in (the first
        // byte of each quarter is decoded in context 0, :843-850)
        if (lane == 0) {
            i32 st = ST_OK;
            u32 cp = 9, rle_i = 0, prev = 0;
            bool first = true;
            u32 c = src.at(cp++);
            do {
                if (!first && c <= prev) { st = ST_TABLE; break; }
                first = false; prev = c;
                S.present[c] = 1;
                u32 dummy;
                cp = x8_get_table(src, cp, end, nullptr, nullptr, &dummy, true, S.present, st);
                if (st != ST_OK) break;
                const u32 nx = src.at(cp);
                if (!rle_i && c + 1 == nx) { c = src.at(cp++); rle_i = src.at(cp++); }
                else if (rle_i) { rle_i--; c++; if (c > 255) { st = ST_TABLE; break; } }
                else c = src.at(cp++);
            } while (c);
            u32 n = 0;
            if (st == ST_OK) {
                S.present[0] = 1;
                for (u32 j = 0; j < 256; j++) if (S.present[j]) { S.idx_of[j] = (u8)n; S.alpha[n++] = (u8)j; }
                S.alpha[n++] = 0;
                if (cp + 16 > end) st = ST_TRUNCATED;                            // :826
            }
            Z.status = st; Z.pos = cp; Z.n = n;
        }
        __syncthreads();
        if (Z.status == ST_OK) {
            const u32 n = Z.n, stride = img_row_bytes(n);
            u8 *rows0 = img + img_alpha_bytes(n);
            // every context starts without a table; the flagged symbol has none either
            for (u32 j = lane; j < n; j += WAVE) ((u16 *)img)[j] = (u16)(S.alpha[j] | ROW_EMPTY | (j + 1 == n ? ROW_BAD : 0u));
            for (u32 r = 0; r < n; r++) {                                        // rows of contexts without a table: all-sentinel
                u16 *w = (u16 *)(rows0 + (u64)r * stride);
                for (u32 t = lane; t < stride / 2; t += WAVE) w[t] = t == (img_leaf_off(img_levels(n)) / 2) ? (u16)0 : (u16)0x7fffu;
            }
            __syncthreads();
            // pass 2: the tables, context by context
            u32 cp = 9, rle_i = 0;
            u32 c = 0;
            if (lane == 0) { Z.pos = cp + 1; Z.go = 1; }
            c = src.at(cp);
            for (;;) {
                for (u32 j = lane; j < 256; j += WAVE) { Z.Fsym[j] = 0; Z.listed[j] = 0; }
                __syncthreads();
                if (lane == 0) {
                    i32 st = ST_OK;
                    const u32 np = x8_get_table(src, Z.pos, end, Z.Fsym, Z.listed, &Z.total, true, nullptr, st);
                    Z.status = st; Z.pos = np;
                }
                __syncthreads();
                if (Z.status != ST_OK) break;
                const u32 ci = S.idx_of[c];
                x8_build_row(Z, rows0 + (u64)ci * stride, n, lane);
                if (lane == 0) ((u16 *)img)[ci] &= (u16)~ROW_EMPTY;
                // next context (same grammar as the symbols; order was checked in pass 1)
                cp = Z.pos;
                const u32 nx = src.at(cp);
                if (!rle_i && c + 1 == nx) { c = src.at(cp); rle_i = src.at(cp + 1); cp += 2; }
                else if (rle_i) { rle_i--; c++; }
                else { c = src.at(cp); cp += 1; }
                if (lane == 0) Z.pos = cp;
                __syncthreads();
                if (!c) break;
            }
        }
    }
    __syncthreads();
    if (Z.status != ST_OK) { if (lane == 0) { a.status[i] = Z.status; a.out_size[i] = 0; } return; }
    if (lane == 0) {
        u32 p = Z.pos;
        i32 st = ST_OK;
        for (u32 k = 0; k < 4; k++, p += 4) {
            const u32 r = (u32)src.at(p) | ((u32)src.at(p + 1) << 8) | ((u32)src.at(p + 2) << 16) | ((u32)src.at(p + 3) << 24);
            I->R[k] = r;
            if (r < X8_LOW) st = ST_STATE;                                        // :316-319, :829-832
        }
        I->bytes = (u64)(in + p); I->bytes_len = end - p;
        I->out = (u64)(a.out + a.out_off[i]);
        I->image = (u64)img; I->nsym = Z.n;
        a.status[i] = st;
        a.out_size[i] = st == ST_OK ? I->out_sz : 0;
        I->active = st == ST_OK && I->out_sz != 0;
    }
}

// The 4x8 chain loop (rANS_static.c:323-372, :857-916): the lookups of chain_decode, byte renormalisation.
template <int ORDER, int LV>
__device__ __forceinline__ u32 chain_decode8(GImg img, u32 nsym, gcu8 *bytes, u32 bytes_len, gu8 *out, u32 out_sz, u32 x,
                                             bool active, u32 lane)
{
    const u32 k = lane & 3;
    const u32 rows = img_alpha_bytes(nsym), roww = img_row_bytes(nsym);
    const u32 below = (1u << k) - 1u;
    u32 count, pos;
    if (ORDER == 0) { count = (out_sz + 3 - k) >> 2; pos = k; }
    else { const u32 q = out_sz >> 2; count = q + (k == 3 ? out_sz - 4 * q : 0); pos = k * q; }
    if (!active) count = 0;
    u32 row = rows, cursor = 0, bad = 0, t = 0;
    if (ORDER == 1 && count) bad = img.ld16(0) & ROW_EMPTY;
    while (wave_any(t < count)) {
        const bool live = t < count;
        if (live) {
            const u32 s = lookup_step<LV>(img, row, X8_BITS, 4095u, x);
            const u32 al = img.ld16(2 * s);
            bad |= al & ROW_BAD;
            if (ORDER == 0) { out[pos] = (u8)al; pos += 4; }
            else {
                row = rows + s * roww;
                if (t + 1 < count) bad |= al & ROW_EMPTY;
                out[pos] = (u8)al; pos += 1;
            }
        }
        // a chain takes one byte if x < 2^23 and a second one if x < 2^15 (x << 8 | byte < 2^23 whatever the byte
        // is); chains are served in the order 0..3, and nothing is read past the end (rANS_byte.h:541-551)
        const bool w1 = live && x < X8_LOW, w2 = live && x < (1u << 15);
        const u32 m1 = quad_ballot(w1, lane), m2 = quad_ballot(w2, lane);
        const u32 at = cursor + __popc(m1 & below) + __popc(m2 & below);
        const u32 want = (w1 ? 1u : 0u) + (w2 ? 1u : 0u);
        const u32 room = at < bytes_len ? bytes_len - at : 0u;
        const u32 take = want < room ? want : room;
        if (take) {
            x = (x << 8) | bytes[at];
            if (take > 1) x = (x << 8) | bytes[at + 1];
        }
        cursor += __popc(m1) + __popc(m2);
        t++;
    }
    return bad;
}

// The same loop with the image and a 256-byte window of the stream's bytes in LDS (as k_dec_chain has them for 4x16):
// the first version read its tables through L2 and fetched every renormalisation byte with a load that depended on the
// step before (19-25 GB/s).  Ring: quarters h, h+1, h+2 of the stream are resident while the cursor is in quarter h (a
// trip of eight steps moves it by at most 64 bytes), quarter h+3 waits in registers.
#define X8_RING 272u
#define X8_TRIP 8
template <int ORDER, int LV>
__device__ __forceinline__ u32 chain_decode8_lds(const u8 *img_lds, u32 nsym, u8 *ring, gcu8 *bytes, u32 bytes_len, gu8 *out, u32 out_sz,
                                                 u32 x, bool active, u32 lane)
{
    const LImg img{lds_addr(img_lds)};
    const u32 k = lane & 3;
    const u32 rows = img_alpha_bytes(nsym), roww = img_row_bytes(nsym);
    const u32 below = (1u << k) - 1u;
    u32 count, pos;
    if (ORDER == 0) { count = (out_sz + 3 - k) >> 2; pos = k; }
    else { const u32 q = out_sz >> 2; count = q + (k == 3 ? out_sz - 4 * q : 0); pos = k * q; }
    if (!active) count = 0;
    // stream bytes relative to the 16-byte aligned address below `bytes`
    gcu8 *abase = (gcu8 *)((u64)bytes & ~15ull);
    const u32 off0 = (u32)((u64)bytes & 15ull);
    const u32 avail = off0 + bytes_len;
    const u32 lastc = avail ? (avail - 1u) >> 4 : 0u;
    const bool loadable = active && avail != 0;
    auto load_chunk = [&](u32 c) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (loadable) v = *(gcu32x4 *)(abase + 16ull * (c < lastc ? c : lastc));   // past the input the last chunk repeats: never taken
        return v;
    };
    if (active) {
        const u32x4 c0 = load_chunk(k);
        *(u32x4 *)(ring + 16 * k) = c0;
        *(u32x4 *)(ring + 64 + 16 * k) = load_chunk(k + 4);
        *(u32x4 *)(ring + 128 + 16 * k) = load_chunk(k + 8);
        if (k == 0) *(u32x2 *)(ring + 256) = c0.xy;           // the first 8 bytes once more behind the ring: no wrap inside a window
    }
    u32x4 pend = load_chunk(12 + k);
    u32 half = 0;
    __syncthreads();
    const u32 rbase = lds_addr(ring);
    u32 row = rows, cursor = 0, bad = 0, t = 0;
    u32x2 root = LV == 2 ? img.ld64(row) : u32x2{0u, 0u};
    if (ORDER == 1 && count) bad = img.ld16(0) & ROW_EMPTY;
    // Output: a trip's bytes are gathered in registers and leave at the top of the NEXT trip - the wave's vector-memory
    // counter retires in order, so a store issued just before the ring refill would make the refill's wait for its
    // prefetched chunk a wait for that store's acknowledgement (byte stores every step: 330 ns per step instead of 250).
    // Order 1: a chain's eight consecutive bytes, one 8-byte store.  Order 0: step T puts byte 4T + k on chain k; the
    // quad transposes four steps' bytes (k_dec_chain's scheme) and each lane stores the dword of one step.
    u32 accA = 0, accB = 0, pendA = 0, pendB = 0;
    u64 pend_at = 0;
    bool pending = false;
    const u32 qcount = ORDER == 0 ? (out_sz >> 2) : 0u;       // order 0: steps in which all four chains produce a byte
    while (wave_any(t < count)) {
        if (pending) {
            if (ORDER == 1) *(GAS u32x2_unaligned *)(out + pend_at) = u32x2{pendA, pendB};
            else { *(GAS u32_unaligned *)(out + pend_at) = pendA; *(GAS u32_unaligned *)(out + pend_at + 16) = pendB; }
            pending = false;
        }
        // gathered route for this trip: every step of it produces a byte on this chain (order 1) / on all four chains
        const bool whole = active && (ORDER == 1 ? t + X8_TRIP <= count : t + X8_TRIP <= qcount);   // (idle quads point at another's item)
#pragma unroll
        for (int u = 0; u < X8_TRIP; u++) {
            const bool live = t + (u32)u < count;
            // the eight bytes at the quad's cursor - all it can take in one step - are requested before the table
            // look-ups, as three aligned dwords (k_dec_chain does the same with its words)
            const u32 P0 = off0 + cursor, ra = rbase + (P0 & 252u);
            const u32 d0 = *(LAS const volatile u32 *)(unsigned long)ra, d1 = *(LAS const volatile u32 *)(unsigned long)(ra + 4u),
                      d2 = *(LAS const volatile u32 *)(unsigned long)(ra + 8u);
            u32 xn = x;
            const u32 s = lookup_step<LV>(img, row, X8_BITS, 4095u, xn, LV == 2 ? &root : nullptr);
            // the next row's root: requested as soon as the symbol is known, used at the top of the next step
            const u32 rown = ORDER == 1 ? rows + __umul24(s, roww) : row;
            const u32x2 rootn = (LV == 2 && ORDER == 1) ? img.ld64_now(rown) : root;
            const u32 al = img.ld16(2 * s);
            x = live ? xn : x;
            bad |= live ? (al & ROW_BAD) : 0u;
            if (ORDER == 1) {
                row = live ? rown : row;
                root.x = live ? rootn.x : root.x;
                root.y = live ? rootn.y : root.y;
                bad |= (t + (u32)u + 1 < count) ? (al & ROW_EMPTY) : 0u;
            }
            if (live && !whole) { out[pos] = (u8)al; pos += ORDER == 0 ? 4 : 1; }
            if (u < 4) accA = __builtin_amdgcn_alignbit(al, accA, 8); else accB = __builtin_amdgcn_alignbit(al, accB, 8);
            // a chain takes one byte if x < 2^23 and a second one if x < 2^15; chains are served in the order 0..3, and
            // nothing is read past the end (rANS_byte.h:541-551)
            const bool w1 = live && x < X8_LOW, w2 = live && x < (1u << 15);
            const u32 m1 = quad_ballot(w1, lane), m2 = quad_ballot(w2, lane);
            const u32 pre = __popc(m1 & below) + __popc(m2 & below);        // bytes the chains before this one take: 0..6
            const u32 at = cursor + pre;
            const u32 want = (w1 ? 1u : 0u) + (w2 ? 1u : 0u);
            const u32 room = at < bytes_len ? bytes_len - at : 0u;
            const u32 take = want < room ? want : room;
            const u32 wlo = __builtin_amdgcn_alignbyte(d1, d0, P0), whi = __builtin_amdgcn_alignbyte(d2, d1, P0);
            const u64 win = (((u64)whi << 32) | wlo) >> (8u * pre);
            const u32 b0 = (u32)win & 0xffu, b1 = ((u32)win >> 8) & 0xffu;
            const u32 x1 = (x << 8) | b0, x2 = (x1 << 8) | b1;
            x = take > 1 ? x2 : take ? x1 : x;
            cursor += __popc(m1) + __popc(m2);
        }
        if (ORDER == 1) {
            if (whole) { pendA = accA; pendB = accB; pend_at = pos; pos += X8_TRIP; pending = true; }
        } else {
            // (uniform over a quad: out_sz is) 4 x 4 byte transposes: lane j gets the dword of step t + j / t + 4 + j
            const u32 sel = k | ((4u + k) << 8);
            const u32 A0 = quad_bcast0(accA), A1 = quad_bcast1(accA), A2 = quad_bcast2(accA), A3 = quad_bcast3(accA);
            const u32 B0 = quad_bcast0(accB), B1 = quad_bcast1(accB), B2 = quad_bcast2(accB), B3 = quad_bcast3(accB);
            const u32 dA = __builtin_amdgcn_perm(__builtin_amdgcn_perm(A3, A2, sel), __builtin_amdgcn_perm(A1, A0, sel), 0x05040100u);
            const u32 dB = __builtin_amdgcn_perm(__builtin_amdgcn_perm(B3, B2, sel), __builtin_amdgcn_perm(B1, B0, sel), 0x05040100u);
            if (whole) { pendA = dA; pendB = dB; pend_at = 4ull * (t + k); pos += 4 * X8_TRIP; pending = true; }
        }
        t += X8_TRIP;
        const u32 nh = (off0 + cursor) >> 6;
        if (wave_any(active && nh != half)) {
            if (active && nh != half) {
                const u32 slot = ((nh + 2) & 3u) * 64u + 16u * k;
                *(u32x4 *)(ring + slot) = pend;
                if (slot == 0) *(u32x2 *)(ring + 256) = pend.xy;
                pend = load_chunk(4 * (nh + 3) + k);
                half = nh;
            }
            __syncthreads();
        }
    }
    if (pending) {
        if (ORDER == 1) *(GAS u32x2_unaligned *)(out + pend_at) = u32x2{pendA, pendB};
        else { *(GAS u32_unaligned *)(out + pend_at) = pendA; *(GAS u32_unaligned *)(out + pend_at + 16) = pendB; }
    }
    return bad;
}

// streams grouped by LDS need (k_cls_scan / k_cls_scatter as for 4x16): class 0 - one-row images (order 0), 16 per
// wave; class 1 - images up to X8_SLOT1 (order 1, up to 50 symbols: the quality alphabets), 12 per wave; class 2 - the
// rest, tables through L2
#define X8_SLOT0 (IMG_O0_BYTES + X8_RING)
#define X8_SLOT1A 5456u      // round 4: up to 47 symbols (the 45-46 of the quality alphabets and the flagged extra one) - fifteen per wave, two waves per CU (30 streams)
#define X8_SLOT1 6416u
__global__ __launch_bounds__(256) void k8_classify(const X8Item *items, int nitems, u32 *cls, u32 *count)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= nitems) return;
    const X8Item *I = &items[i];
    u32 c = CLS_NONE;
    if (I->active) {
        const u32 need = img_bytes(I->nsym, I->order ? I->nsym : 1u) + X8_RING;
        c = need <= X8_SLOT0 ? 0u : need <= X8_SLOT1A ? 1u : need <= X8_SLOT1 ? 2u : 3u;
        atomicAdd(&count[c], 1u);
    }
    cls[i] = c;
}

template <bool LDS_IMG>
__global__ __launch_bounds__(WAVE) void k8_dec_chain(const X8Item *items, BatchArgs a, int base, const u32 *list, const u32 *count,
                                                     int qpw, u32 slot_bytes)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 lane = threadIdx.x, quad = lane >> 2;
    const int nmine = (int)count[0];
    list += count[CLS_MAX];
    const int slot = (int)blockIdx.x * qpw + (int)quad;
    if ((int)blockIdx.x * qpw >= nmine) return;
    const bool mine = quad < (u32)qpw && slot < nmine;
    const u32 idx = list[mine ? slot : (int)blockIdx.x * qpw];
    const X8Item *I = &items[idx];
    const bool active = mine && I->active;
    const u32 nsym = active ? I->nsym : 1u, order = active ? I->order : 2u;
    const u32 lv = img_levels(nsym);
    gcu8 *bytes = (gcu8 *)I->bytes;
    gu8 *out = (gu8 *)I->out;
    const u32 blen = I->bytes_len, osz = I->out_sz, x0 = I->R[lane & 3];
    u32 bad = 0;
    if (LDS_IMG) {
        const u64 my_img = active ? I->image : 0ull;
        const u32 nbytes = active ? img_bytes(nsym, order ? nsym : 1u) : 0u;
        for (int qd = 0; qd < qpw; qd++) {
            const u64 src = __shfl(my_img, qd * 4);
            const u32 nb = __shfl(nbytes, qd * 4);
            if (!src) continue;
            gcu32x4 *sp = (gcu32x4 *)src;
            u32x4 *dd = (u32x4 *)(lds + (u64)qd * slot_bytes);
            for (u32 j = lane; j < ((nb + 15) >> 4); j += WAVE) dd[j] = sp[j];
        }
        __syncthreads();
        const u8 *im = lds + (u64)quad * slot_bytes;
        u8 *ring = lds + (u64)quad * slot_bytes + (slot_bytes - X8_RING);
        // (round 4: the 4x16 decoder's loop with rANS 4x8's byte renormalisation - full trips without liveness selects,
        //  speculative roots, gathered stores - instead of round 2's chain_decode8_lds)
#define X8_RUN(O, L) bad |= chain_decode_lds<O, L, TRIP_STEPS, true>(im, nsym, ring, bytes, blen, out, osz, x0, X8_BITS, active && order == O && lv == L, lane)
        X8_RUN(0, 2); X8_RUN(0, 3); X8_RUN(0, 4);
        X8_RUN(1, 2);                                          // (X8_SLOT1 holds 2-read images only: up to 50 symbols)
#undef X8_RUN
    } else {
        GImg im{(gcu8 *)I->image};
#define X8_RUN(O, L) bad |= chain_decode8<O, L>(im, nsym, bytes, blen, out, osz, x0, active && order == O && lv == L, lane)
        X8_RUN(0, 2); X8_RUN(0, 3); X8_RUN(0, 4);
        X8_RUN(1, 2); X8_RUN(1, 3); X8_RUN(1, 4);
#undef X8_RUN
    }
    if (active && bad) {
        a.status[base + (int)idx] = (bad & ROW_BAD) ? ST_TABLE : ST_CONTEXT;
        a.out_size[base + (int)idx] = 0;
    }
}

extern "C" size_t r4x8_dec_ws_bytes(size_t nblk)
{
    return align_up_sz(nblk * sizeof(X8Item), 256) + nblk * (size_t)IMG8_SLOT + 2 * align_up_sz(nblk * 4, 256) + align_up_sz(3 * CLS_MAX * 4, 256);
}
extern "C" void r4x8_launch_decode(const BatchArgs *a, u8 *ws, int base, int nblk, hipStream_t s)
{
    X8Item *items = (X8Item *)ws;
    u8 *images = ws + align_up_sz((size_t)nblk * sizeof(X8Item), 256);
    u32 *cls = (u32 *)(images + (size_t)nblk * IMG8_SLOT);
    u32 *cls_list = (u32 *)((u8 *)cls + align_up_sz((size_t)nblk * 4, 256));
    u32 *cls_count = (u32 *)((u8 *)cls_list + align_up_sz((size_t)nblk * 4, 256));
    hipLaunchKernelGGL(k8_dec_front, dim3(nblk), dim3(WAVE), 0, s, *a, items, images, base);
    r4x16_launch_cls_zero(cls_count, s);
    hipLaunchKernelGGL(k8_classify, dim3((nblk + 255) / 256), dim3(256), 0, s, (const X8Item *)items, nblk, cls, cls_count);
    r4x16_launch_cls_group(cls, nblk, cls_count, cls_list, s);
    if (r4x16_first_on_device(16u)) lds_limit((const void *)k8_dec_chain<true>, 163840);
    // (measured with the shape as a launch parameter, gpurun_out/r04_ab_x8*.txt, 11,520 x 1 MiB q40: a round takes ~58 ms whatever the wave holds - 12, 14 or 15 streams in
    //  each of two waves per CU: 100 GB/s; 8 x 3 waves: 87; 10 per wave does NOT make three waves: 54,560 bytes round up
    //  past a third of the LDS - so the class holds as many streams as two waves can: 2 x 15 x 5,456 = 163,680 bytes)
    const int q0 = 16, q1a = 15, q1 = 12;
    hipLaunchKernelGGL(k8_dec_chain<true>, dim3((nblk + q0 - 1) / q0), dim3(WAVE), (size_t)q0 * X8_SLOT0, s, (const X8Item *)items, *a, base,
                       (const u32 *)cls_list, (const u32 *)(cls_count + 0), q0, X8_SLOT0);
    hipLaunchKernelGGL(k8_dec_chain<true>, dim3((nblk + q1a - 1) / q1a), dim3(WAVE), (size_t)q1a * X8_SLOT1A, s, (const X8Item *)items, *a, base,
                       (const u32 *)cls_list, (const u32 *)(cls_count + 1), q1a, X8_SLOT1A);
    hipLaunchKernelGGL(k8_dec_chain<true>, dim3((nblk + q1 - 1) / q1), dim3(WAVE), (size_t)q1 * X8_SLOT1, s, (const X8Item *)items, *a, base,
                       (const u32 *)cls_list, (const u32 *)(cls_count + 2), q1, X8_SLOT1);
    hipLaunchKernelGGL(k8_dec_chain<false>, dim3((nblk + 15) / 16), dim3(WAVE), 0, s, (const X8Item *)items, *a, base,
                       (const u32 *)cls_list, (const u32 *)(cls_count + 3), 16, 0u);
}
