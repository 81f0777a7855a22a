// r4x16_enc_chain.h - k_enc_chain, the encoder's hot loop (rANS_static4x16pr.c:442-485, :794-839): device code shared by
// r4x16_enc_chain.hip (u16 images, order 0 and 1; the launcher) and r4x16_enc_chain_pk.hip (the packed-row
// instantiation, compiled with another instruction scheduler).
#pragma once
#include <stdlib.h>
#include "r4x16_dev.h"
#include "r4x16_sched.h"
#include "r4x16_enc_step.h"

// General form: image in global memory, byte loads.  Used for the small nested streams inside
// k_enc_front and for alphabets whose tables do not fit LDS.
template <int ORDER>
__device__ __forceinline__ u32 chain_encode(gcu8 *data, u32 n, gcu8 *image, u32 ns, u32 bits, gcu32 *rcptab,
                                            gu8 *scratch_end, bool active, u32 lane)
{
    const u32 k = lane & 3;
    GAS const u16 *cum = (GAS const u16 *)(image + ENC_IMG_IDX);
    const u32 rs = ns + 1;
    u32 x = RANS_LOW;
    u32 written = 0;                 // words emitted by the quad so far
    u32 nsteps, first;               // this lane takes part in steps [first, nsteps)
    u32 p;                           // position of the symbol coded at this lane's next step
    const u32 q = n >> 2;
    if (ORDER == 0) {
        const u32 gtop = n ? (n - 1) >> 2 : 0;
        nsteps = n ? gtop + 1 : 0;
        first = (4 * gtop + k < n) ? 0 : 1;      // the top group may be partial (:442-448)
        p = 4 * (gtop - (first ? 1 : 0)) + k;    // unused when nsteps <= first
    } else {
        const u32 tail = n - 4 * q;              // extra bytes on chain 3 (:806-811)
        nsteps = tail + q;
        first = (k == 3) ? 0 : tail;
        p = (k == 3) ? n - 1 : k * q + q - 1;
    }
    if (!active) { nsteps = 0; first = 0; }

    u32 cur = 0;                                 // compact index of the symbol coded next
    if (nsteps > first) cur = image[data[p]];

    for (u32 s = 0; wave_any(s < nsteps); s++) {
        const bool live = s >= first && s < nsteps;
        bool emit = false;
        u32 rcp = 0, pk = 0, nextc = 0;
        if (live) {
            u32 row = 0;
            if (ORDER == 0) {
                if (p >= 4) nextc = image[data[p - 4]];
            } else if (s != nsteps - 1) {        // context = previous byte; quarter start: context 0
                nextc = image[data[p - 1]];
                row = nextc;
            }
            const u32 c0 = cum[row * rs + cur], c1 = cum[row * rs + cur + 1];
            pk = c0 | ((c1 - c0) << 16);
            rcp = enc_rcp(rcptab, c1 - c0);
            emit = enc_wants_emit(x, pk, bits);
        }
        const u32 em = quad_ballot(emit, lane);
        if (emit) {
            const u32 above = __popc(em >> (k + 1));
            *(gu16 *)(scratch_end - 2 * (written + above + 1)) = (u16)x;
            x >>= 16;
        }
        written += __popc(em);
        if (live) {
            x = enc_advance(x, rcp, pk, bits);
            cur = nextc;
            p -= (ORDER == 0) ? 4 : 1;
        }
    }
    // RansEncFlush x4 in order 3,2,1,0 (:482-485): R0 ends up lowest in memory
    if (active) *(gu32 *)(scratch_end - 2 * written - 16 + 4 * k) = x;
    return active ? 2 * written + 16 : 0;
}

// ---------------------------------------------------------------------------------------------
// Hot form, order-1: the image sits in LDS and nothing on the dependent path (the state x)
// touches memory.  Symbols are known in advance, so their table entries are fetched one trip
// (four steps) ahead: a trip issues the global load of the input dword two trips ahead, the LDS
// index lookups of the next trip's bytes and their entry reads, and then runs four state updates
// on entries that were loaded during the previous trip.
// The schedule is phased per wave: (A) up to three tail steps, chain 3 only; (B) whole trips of
// the backward walk; (B') its last 0..3 steps; (C) the quarter starts in context 0.  Streams of
// different lengths in one wave simply drop out of (B) at different trips.
// ---------------------------------------------------------------------------------------------
// Emitted 16-bit words are staged in a 128-byte LDS ring per stream and copied out 64 bytes at a
// time (16 bytes per lane).  Word j of a stream (emission order) belongs at scratch_end - 2 (j + 1);
// in the ring it sits at byte 126 - 2 (j & 63), which keeps each 64-byte half in memory order.
//
// Why: the wave's vector-memory counter retires in order, so ONE outstanding HBM access (an
// input prefetch, a store waiting for its acknowledgement) stalls every later wait on that
// counter.  The hot loop therefore keeps everything it consumes per trip in LDS (tables, the
// reciprocal table, the emitted words) and touches global memory exactly twice per eight steps,
// unconditionally and in a fixed order: one 8-byte input load three double-trips ahead and one
// 16-byte store (a completed half of the ring, or a dump slot nobody reads).
// (ENC_RING_BYTES, r4x16_common.h: the ring + a 2-byte dump slot for lanes that do not emit, padded)
#define ENC_LRCP_BYTES 16400u        // RCPTAB_ENTRIES dwords, padded to 16
// BYTE: rANS 4x8's renormalisation (rANS_byte.h:320-402: L = 2^23, up to TWO bytes per chain and step, 12-bit
// frequencies) on the same pipeline, ring and flush - round 4; the unit of `written` / the ring is then the byte.
template <bool BYTE>
struct EncOutT {
    u8 *ring;            // LDS
    u32 ring126;         // LDS address of the ring's last slot (word 126 / byte 127)
    gu8 *send;           // scratch_end of the stream
    gu8 *dump;           // this lane's 16 bytes of the dump area
    u32 written;         // words (bytes) emitted by the quad so far
    u32 flushed;         // 64-byte halves already read out of the ring
    u32 k, lane;
    bool active;
    u32x4 held;          // a half read out of the ring, stored one double-trip later
    gu8 *held_dst;
    static constexpr u32 HALF_SHIFT = BYTE ? 6u : 5u;    // units per 64-byte half, as a shift
    // rANS_word.h:281-321 for one symbol; x is this lane's state.  pk = start | freq << 16.
    // q = x / freq < 2^21 once x < x_max, so q * (M - freq) is a 24-bit multiply (mod 2^32).
    __device__ __forceinline__ void step(u32 &x, bool live, u32 rcp, u32 pk, u32 bits)
    {
        const u32 f = pk >> 16, start = pk & 0xffffu;
        u32 xs;
        if (BYTE) {
            // x_max = ((2^23 >> 12) << 8) * f; one byte while x >= x_max, a second one if x >> 8 still is
            const u32 x_max = f << 19;
            const bool o1 = x >= x_max, o2 = (x >> 8) >= x_max;
            const u64 lv = __ballot(live);
            const u32 e1 = (u32)((__ballot(o1) & lv) >> (lane & ~3u)) & 0xfu, e2 = (u32)((__ballot(o2) & lv) >> (lane & ~3u)) & 0xfu;
            const bool emit1 = live && o1, emit2 = live && o2;
            const u32 j = written + __popc(e1 >> (k + 1)) + __popc(e2 >> (k + 1));
            const u32 j1 = emit1 ? (j & 127u) : ~0u, j2 = emit2 ? ((j + 1u) & 127u) : ~0u;       // -1: the dump slot at ring + 128
            *(LAS u8 *)(unsigned long)(ring126 - j1) = (u8)x;
            *(LAS u8 *)(unsigned long)(ring126 - j2) = (u8)(x >> 8);
            xs = emit2 ? x >> 16 : emit1 ? x >> 8 : x;
            written += __popc(e1) + __popc(e2);
        } else {
            // the compare's own lane mask, and-ed with the live lanes on the scalar side (a ballot of the
            // combined predicate would be rebuilt through a select and a second compare)
            const bool over = x >= (f << (31u - bits));
            const u64 m = __ballot(over) & __ballot(live);
            const u32 em = (u32)(m >> (lane & ~3u)) & 0xfu;
            const bool emit = live && over;
            const u32 j = written + __popc(em >> (k + 1));
            const u32 j63 = emit ? (j & 63u) : ~0u;                          // -1: the dump slot at ring + 128
            *(LAS u16 *)(unsigned long)(ring126 - 2u * j63) = (u16)x;
            xs = emit ? x >> 16 : x;
            written += __popc(em);
        }
        // exact x / f: Alverson reciprocal for f >= 2; f == 1 has rcp = 2^32 - 1 and shift 0, which
        // gives x - 1: the compare's carry puts the 1 back (an add-with-carry, no select)
        const u32 fm1 = f - 1u;
        const u32 rsh = 31u - (u32)__clz((int)(fm1 | 1u));
        const u32 q = (__umulhi(xs, rcp) >> rsh) + (fm1 == 0u ? 1u : 0u);
        const u32 cmpl = (1u << bits) - f;
        const u32 xn = __umul24(q, cmpl) + (xs + start);
        x = live ? xn : xs;
    }
    // conditional form: copy out the half that has just been completed, if any
    __device__ __forceinline__ void flush()
    {
        const bool due = active && (written >> HALF_SHIFT) != flushed;
        if (wave_any(due)) {
            if (due) {
                const u32x4 v = *(const u32x4 *)(ring + ((flushed & 1u) ? 0u : 64u) + 16u * k);
                *(GAS u32x4_unaligned *)(send - 64ull * (flushed + 1u) + 16u * k) = v;
                flushed++;
            }
        }
    }
    // unconditional form for the main loop: store what was read out last time, read out the next
    __device__ __forceinline__ void flush_pipelined()
    {
        *(GAS u32x4_unaligned *)held_dst = held;
        const bool due = active && (written >> HALF_SHIFT) != flushed;
        held = *(const u32x4 *)(ring + ((flushed & 1u) ? 0u : 64u) + 16u * k);
        held_dst = due ? send - 64ull * (flushed + 1u) + 16u * k : dump;
        flushed += due ? 1u : 0u;
    }
    __device__ __forceinline__ void flush_drain()
    {
        *(GAS u32x4_unaligned *)held_dst = held;
        held_dst = dump;
    }
    // the words (bytes) still in the ring, then the four states (RansEncFlush in order 3,2,1,0, :482-485)
    __device__ __forceinline__ u32 finish(u32 x)
    {
        flush();
        if (BYTE) { flush(); }                           // (a double trip of byte steps can complete two halves)
        const u32 first = (BYTE ? 64u : 32u) * flushed;
        const u32 rem = active ? written - first : 0u;
        for (u32 i = k; wave_any(i < rem); i += 4) {
            if (i < rem) {
                const u32 j = first + i;
                if (BYTE) send[-(long)(j + 1u)] = *(const u8 *)(ring + (127u - (j & 127u)));
                else *(gu16 *)(send - 2ull * (j + 1u)) = *(const u16 *)(ring + ((~j << 1) & 126u));
            }
        }
        if (active) {
            if (BYTE) { gu8 *d = send - written - 16 + 4 * k; d[0] = (u8)x; d[1] = (u8)(x >> 8); d[2] = (u8)(x >> 16); d[3] = (u8)(x >> 24); }
            else *(gu32 *)(send - 2ull * written - 16 + 4 * k) = x;
        }
        return active ? (BYTE ? written : 2 * written) + 16 : 0;
    }
};
typedef EncOutT<false> EncOut;

// PK: packed rows (r4x16_common.h) and a reciprocal table of the 1,025 frequencies a 10-bit table can hold.
template <bool PK, bool BYTE = false>
__device__ __forceinline__ u32 chain_encode_o1_lds(const u8 *img_lds, u8 *ring, const u32 *lrcp, gcu8 *data, u32 n, u32 ns,
                                                   u32 bits, gcu8 *safe, gu8 *scratch_end, gu8 *dump, bool active, u32 lane)
{
    const u32 k = lane & 3;
    const u8 *idx = img_lds;
    const u8 *cumb = img_lds + ENC_IMG_IDX;
    const u32 rs = PK ? 4u * enc_pk_row_dwords(ns) : ns + 1;       // bytes (packed) / u16 entries per context row
    const u32 cumb_lds = (u32)(unsigned long)(LAS const u8 *)cumb;
    // the (start, next) pair of symbol si in context ci.  u16 rows: start | next << 16, one dword read at a 2-byte
    // aligned LDS address.  Packed rows: 22 bits at bit 11 si of the row, from two aligned dwords and a funnel shift.
    auto pair = [&](u32 ci, u32 si) -> u32 {
        if (PK) {
            const u32 b = __umul24(si, 11u);
            const u32 a = cumb_lds + __umul24(ci, rs) + ((b >> 5) << 2);
            const u32x2 d = *(LAS const u32x2_a4 *)(unsigned long)a;
            return __builtin_amdgcn_alignbit(d.y, d.x, b);                   // (the shift uses the low five bits of b)
        }
        return *(LAS const u32 *)(cumb + 2u * (__umul24(ci, rs) + si));
    };
    const u32 rcp_last = PK ? 1024u : RCPTAB_ENTRIES - 1u;
    auto rcpof = [&](u32 pk) -> u32 { const u32 f = pk >> 16; return lrcp[f < rcp_last ? f : rcp_last]; };   // (clamp: idle lanes hold garbage)
    // pair -> start | freq << 16.  u16 rows: a shift and a subtract (the empty asm keeps it from becoming a
    // quarter-rate multiply by 0xFFFF0001); packed rows: two field extractions, a subtract, a shift-or
    auto topk = [&](u32 p) -> u32 {
        if (PK) { const u32 st = p & 2047u; return st | ((__builtin_amdgcn_ubfe(p, 11, 11) - st) << 16); }
        u32 hi = p << 16; asm("" : "+v"(hi)); return p - hi;
    };
    auto fetch = [&](u32 ci, u32 si) -> u32x2 {           // {rcp, start | freq << 16}
        const u32 pk = topk(pair(ci, si));
        u32x2 r = {rcpof(pk), pk};
        return r;
    };
    EncOutT<BYTE> o{ring, (u32)(unsigned long)(LAS u8 *)ring + (BYTE ? 127u : 126u), scratch_end, dump, 0u, 0u, k, lane, active, {0, 0, 0, 0}, dump};
    u32 x = BYTE ? (1u << 23) : RANS_LOW;
    const u32 q = active ? n >> 2 : 0;
    const u32 tail = active ? n - 4 * q : 0;

    // (A) tail bytes n-1 .. 4q on chain 3, context = previous byte (:806-811)
    u32 cur = 0;
    if (active && k == 3 && tail) cur = idx[data[n - 1]];
    for (u32 s = 0; wave_any(s < tail); s++) {
        const bool live = k == 3 && s < tail;
        u32 rcp = 0, pk = 0;
        if (live) {
            const u32 ci = idx[data[n - 2 - s]];
            const u32x2 e = fetch(ci, cur);
            rcp = e.x; pk = e.y;
            cur = ci;
        }
        o.step(x, live, rcp, pk, bits);
    }

    // (B) backward walk over offsets q-1 .. 1 of each quarter (:813-829); chain k codes byte
    // k*q + r in context byte k*q + r - 1.  Trip t codes offsets r0-4t .. r0-4t-3; the pipelined
    // loop takes an even number of trips, the rest goes to (B').
    gcu8 *qbase = data + (u64)k * q;
    const u32 r0 = q ? q - 1 : 0;
    const u32 main = r0;                         // steps in (B)+(B')
    const u32 npair = main >> 3;                 // double trips
    const u32 ntrip = 2 * npair;
    cur = (active && q) ? idx[qbase[r0]] : 0u;
    if (wave_any(npair > 0)) {
        // Software pipeline, every access issued at least one trip before its first use:
        //   input piece of double-trip D+3 (HBM, 8 bytes)   byte -> compact index of trip t+3 (LDS)
        //   cumulative pair of trip t+2 (LDS)               reciprocal of trip t+1 (LDS)      trip t: 4 state updates
        auto load8 = [&](u32 j) -> u32x2 {       // bytes r0-8j-8 .. r0-8j-1: .y = contexts of trip 2j, .x = of trip 2j+1
            gcu8 *p = j < npair ? qbase + (r0 - 8 * j) - 8 : safe;
            return *(GAS const u32x2_unaligned *)p;
        };
        struct I4 { u32 c0, c1, c2, c3; };
        auto idx4 = [&](u32 ww) -> I4 {
            I4 r = {idx[ww >> 24], idx[(ww >> 16) & 0xff], idx[(ww >> 8) & 0xff], idx[ww & 0xff]};
            return r;
        };
        auto cum4 = [&](const I4 &c, u32 sym) -> u32x4 {     // raw pairs start | next << 16
            u32x4 r = {pair(c.c0, sym), pair(c.c1, c.c0), pair(c.c2, c.c1), pair(c.c3, c.c2)};
            return r;
        };
        auto topk4 = [&](u32x4 p) -> u32x4 {
            u32x4 r = {topk(p.x), topk(p.y), topk(p.z), topk(p.w)};
            return r;
        };
        auto rcp4 = [&](u32x4 p) -> u32x4 {
            u32x4 r = {rcpof(p.x), rcpof(p.y), rcpof(p.z), rcpof(p.w)};
            return r;
        };
        u32 cur1, cur2;
        I4 I2;
        u32x4 P0, Praw, R0;
        // input pieces live in a ring of four register pairs, piece j in Q[j % 4]; the loop is unrolled
        // four double-trips so that no piece is ever copied (a copy would have to wait for the load)
        u32x2 Q0, Q1, Q2, Q3;
        {
            Q0 = load8(0); Q1 = load8(1); Q2 = load8(2); Q3 = load8(3);
            const I4 i0 = idx4(Q0.y), i1 = idx4(Q0.x);
            I2 = idx4(Q1.y);
            P0 = topk4(cum4(i0, cur));
            Praw = cum4(i1, i0.c3);
            cur1 = i0.c3; cur2 = i1.c3;
            R0 = rcp4(P0);
        }
        u32 t = 0;
        auto trip = [&](u32 wnext3) {
            const bool live = t < ntrip;
            const I4 In = idx4(wnext3);              // bytes of trip t+3
            const u32x4 Pn = cum4(I2, cur2);         // pairs of trip t+2
            const u32x4 P1 = topk4(Praw);            // trip t+1, read during the previous trip
            const u32x4 Rn = rcp4(P1);
            // the look-ups above belong to later trips: keep the scheduler from pulling next trip's
            // (which depend on them) up behind them, which would put their latency on this trip
            __builtin_amdgcn_sched_barrier(0);
            o.step(x, live, R0.x, P0.x, bits);
            o.step(x, live, R0.y, P0.y, bits);
            o.step(x, live, R0.z, P0.z, bits);
            o.step(x, live, R0.w, P0.w, bits);
            if (live) cur = cur1;
            cur1 = cur2; cur2 = I2.c3;
            I2 = In; P0 = P1; Praw = Pn; R0 = Rn;
            t++;
            __builtin_amdgcn_sched_barrier(0);
        };
        // double-trip d: trip 2d looks up the bytes of trip 2d+3 (piece d+1, low dword), trip 2d+1
        // those of trip 2d+4 (piece d+2, high dword); piece d+4 is requested into the slot of piece d
        for (u32 d = 0; wave_any(d < npair); d += 4) {
            Q0 = load8(d + 4); o.flush_pipelined(); trip(Q1.x); trip(Q2.y);
            Q1 = load8(d + 5); o.flush_pipelined(); trip(Q2.x); trip(Q3.y);
            Q2 = load8(d + 6); o.flush_pipelined(); trip(Q3.x); trip(Q0.y);
            Q3 = load8(d + 7); o.flush_pipelined(); trip(Q0.x); trip(Q1.y);
        }
        o.flush_drain();
        o.flush();                               // fewer than 32 words may stay in the ring from here on
    }
    // (B') remaining walk steps, one at a time
    u32 r = r0 - 4 * ntrip, done = 4 * ntrip;
    for (; wave_any(done < main); ) {
        const bool live = done < main;
        u32 rcp = 0, pk = 0;
        if (live) {
            const u32 ci = idx[qbase[r - 1]];
            const u32x2 e = fetch(ci, cur);
            rcp = e.x; pk = e.y;
            cur = ci; r--; done++;
        }
        o.step(x, live, rcp, pk, bits);
        o.flush();
    }
    // (C) first byte of each quarter in context 0 (:831-834)
    {
        const bool live = active && q > 0;
        u32 rcp = 0, pk = 0;
        if (live) { const u32x2 e = fetch(0, cur); rcp = e.x; pk = e.y; }
        o.step(x, live, rcp, pk, bits);
    }
    return o.finish(x);
}

// Hot form, order-0, for the chain kernel: the same software pipeline as chain_encode_o1_lds over
// the one-row image (:442-459: step s codes group g = gtop - s, chain k takes byte 4g + k; the top
// group may be partial).  A trip of four steps covers four whole groups = 16 contiguous bytes, of
// which this lane uses byte k of each dword.
template <bool BYTE = false>
__device__ __forceinline__ u32 chain_encode_o0_pipe(const u8 *img_lds, u8 *ring, const u32 *lrcp, gcu8 *data, u32 n,
                                                    u32 bits, gcu8 *safe, gu8 *scratch_end, gu8 *dump, bool active, u32 lane)
{
    const u32 k = lane & 3;
    const u8 *idx = img_lds;
    const u8 *cumb = img_lds + ENC_IMG_IDX;
    auto pair = [&](u32 si) -> u32 { return *(LAS const u32 *)(cumb + 2u * si); };
    auto rcpof = [&](u32 pk) -> u32 { const u32 f = pk >> 16; return lrcp[f < RCPTAB_ENTRIES - 1u ? f : RCPTAB_ENTRIES - 1u]; };
    auto topk = [&](u32 p) -> u32 { u32 hi = p << 16; asm("" : "+v"(hi)); return p - hi; };
    auto fetch = [&](u32 si) -> u32x2 {
        const u32 pk = topk(pair(si));
        u32x2 r = {rcpof(pk), pk};
        return r;
    };
    EncOutT<BYTE> o{ring, (u32)(unsigned long)(LAS u8 *)ring + (BYTE ? 127u : 126u), scratch_end, dump, 0u, 0u, k, lane, active, {0, 0, 0, 0}, dump};
    u32 x = BYTE ? (1u << 23) : RANS_LOW;
    const u32 Q = active ? n >> 2 : 0;                    // whole groups
    const u32 rem = active ? n & 3u : 0;                  // bytes of the partial top group

    // (A) the partial top group: chains k < rem code byte 4Q + k
    if (wave_any(rem != 0)) {
        const bool live = k < rem;
        u32 rcp = 0, pk = 0;
        if (live) { const u32x2 e = fetch(idx[data[4 * Q + k]]); rcp = e.x; pk = e.y; }
        o.step(x, live, rcp, pk, bits);
    }

    // (B) whole groups Q-1 .. 0; trip t covers groups Q-1-4t .. Q-4-4t = bytes [4 (Q-4t-4), 4 (Q-4t))
    const u32 npair = Q >> 3;                             // double trips
    const u32 ntrip = 2 * npair;
    if (wave_any(npair > 0)) {
        struct DT { u32x4 a, b; };                        // the pieces of trips 2j and 2j+1
        auto load_dt = [&](u32 j) -> DT {
            gcu8 *p = j < npair ? data + 4ull * (Q - 8 * j) - 32 : safe;
            DT r = {*(GAS const u32x4_unaligned *)(p + 16), *(GAS const u32x4_unaligned *)p};
            return r;
        };
        struct I4 { u32 c0, c1, c2, c3; };
        const u32 sh = 8 * k;
        auto idx4 = [&](u32x4 v) -> I4 {                  // steps run from the highest group (v.w) down
            I4 r = {idx[(v.w >> sh) & 0xff], idx[(v.z >> sh) & 0xff], idx[(v.y >> sh) & 0xff], idx[(v.x >> sh) & 0xff]};
            return r;
        };
        auto cum4 = [&](const I4 &c) -> u32x4 {
            u32x4 r = {pair(c.c0), pair(c.c1), pair(c.c2), pair(c.c3)};
            return r;
        };
        auto topk4 = [&](u32x4 p) -> u32x4 {
            u32x4 r = {topk(p.x), topk(p.y), topk(p.z), topk(p.w)};
            return r;
        };
        auto rcp4 = [&](u32x4 p) -> u32x4 {
            u32x4 r = {rcpof(p.x), rcpof(p.y), rcpof(p.z), rcpof(p.w)};
            return r;
        };
        I4 I2;
        u32x4 P0, Praw, R0;
        DT Q0, Q1, Q2, Q3;                                // piece pair j lives in Q[j % 4]
        {
            Q0 = load_dt(0); Q1 = load_dt(1); Q2 = load_dt(2); Q3 = load_dt(3);
            const I4 i0 = idx4(Q0.a), i1 = idx4(Q0.b);
            I2 = idx4(Q1.a);
            P0 = topk4(cum4(i0));
            Praw = cum4(i1);
            R0 = rcp4(P0);
        }
        u32 t = 0;
        auto trip = [&](u32x4 wnext3) {
            const bool live = t < ntrip;
            const I4 In = idx4(wnext3);                   // bytes of trip t+3
            const u32x4 Pn = cum4(I2);                    // pairs of trip t+2
            const u32x4 P1 = topk4(Praw);                 // trip t+1, read during the previous trip
            const u32x4 Rn = rcp4(P1);
            __builtin_amdgcn_sched_barrier(0);
            o.step(x, live, R0.x, P0.x, bits);
            o.step(x, live, R0.y, P0.y, bits);
            o.step(x, live, R0.z, P0.z, bits);
            o.step(x, live, R0.w, P0.w, bits);
            I2 = In; P0 = P1; Praw = Pn; R0 = Rn;
            t++;
            __builtin_amdgcn_sched_barrier(0);
        };
        for (u32 d = 0; wave_any(d < npair); d += 4) {
            Q0 = load_dt(d + 4); o.flush_pipelined(); trip(Q1.b); trip(Q2.a);
            Q1 = load_dt(d + 5); o.flush_pipelined(); trip(Q2.b); trip(Q3.a);
            Q2 = load_dt(d + 6); o.flush_pipelined(); trip(Q3.b); trip(Q0.a);
            Q3 = load_dt(d + 7); o.flush_pipelined(); trip(Q0.b); trip(Q1.a);
        }
        o.flush_drain();
        o.flush();
    }
    // (B') the groups below the pipelined trips, one step each
    for (u32 g = Q - 4 * ntrip; wave_any(g > 0); ) {
        const bool live = g > 0;
        u32 rcp = 0, pk = 0;
        if (live) { g--; const u32x2 e = fetch(idx[data[4 * g + k]]); rcp = e.x; pk = e.y; }
        o.step(x, live, rcp, pk, bits);
        o.flush();
    }
    return o.finish(x);
}


// ---------------------------------------------------------------------------------------------
// k_enc_chain: QPW streams per wave, one launch per LDS size class (see k_dec_chain).
// ---------------------------------------------------------------------------------------------
// LDS_IMG: a workgroup of up to four waves shares one LDS copy of the reciprocal table; each quad
// owns lds_per_item bytes (image, then the word ring).  Waves never meet again after the set-up.
// PK: the class holds packed order-1 streams only (10-bit tables): a 1,025-entry reciprocal table suffices.
#define ENC_LRCP_PK_BYTES 4112u      // 1,025 dwords, padded to 16
template <bool LDS_IMG, bool PK>
__global__ __launch_bounds__(256) void k_enc_chain(EncItem *items, const u32 *rcptab_, u8 *dump_, const u32 *list, u32 *count,
                                                   int qpw, int spw, u32 lds_per_item, int dyn)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 tid = threadIdx.x;
    const u32 lane = tid & (WAVE - 1);
    // qpw streams per workgroup, spw per wave (the first spw quads of each wave)
    const u32 wq = lane >> 2;
    const u32 quad = (tid >> 6) * (u32)spw + wq;
    // persistent: as many workgroups as are resident at once, each walking shares of its class's list (see k_dec_chain
    // and r4x16_sched.h: longest streams first, shares claimed from the class's counter, count[SCHED_SEATS] workgroups work)
    const int nmine = (int)count[SCHED_COUNT];
    list += count[SCHED_START];
    SchedWalk walk(count, nmine, qpw, dyn != 0);
    for (;;) {
    const int wg = LDS_IMG ? walk.next_wg((volatile u32 *)lds + 1) : walk.next_wave();
    if (wg < 0) break;
    const int slot = wg * qpw + (int)quad;
    const bool mine = wq < (u32)spw && quad < (u32)qpw && slot < nmine;
    EncItem *I = &items[mine ? list[slot] : list[wg * qpw]];
    bool active = mine && I->active;
    const u32 img_bytes = active ? I->img_bytes : 0u;
    if (LDS_IMG) {
        // workgroup-wide "does anybody have work here" through one dword of the dynamic LDS
        // (__syncthreads_or would bring 256 bytes of static LDS with it: one stream's worth of room)
        volatile u32 *flag = (volatile u32 *)lds;
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (active) *flag = 1;
        __syncthreads();
        const u32 any = *flag;
        __syncthreads();
        if (!any) continue;
    } else if (!wave_any(active)) continue;
    sched_setprio(sched_prio_of(active, active ? I->n : 0u));

    const u32 order = active ? I->order : 2u;
    gcu32 *rcptab = to_global(rcptab_);
    gcu8 *data = (gcu8 *)I->data;
    gu8 *send = (gu8 *)I->scratch_end;
    const u32 n = I->n, ns = I->ns, bits = active ? I->bits : 12u;
    u32 pay;
    if (LDS_IMG) {
        u32 *lrcp = (u32 *)lds;
        u8 *slots = lds + (PK ? ENC_LRCP_PK_BYTES : ENC_LRCP_BYTES);
        for (u32 j = tid; j < (PK ? 1025u : RCPTAB_ENTRIES); j += blockDim.x) lrcp[j] = rcptab[j];
        // each wave copies the images of its own quads (16-byte pieces)
        const u64 my_img = active ? I->image : 0ull;
        const u32 wq0 = (tid >> 6) * (u32)spw;             // first stream slot of this wave
        for (u32 qd = 0; qd < WAVE / 4; qd++) {
            const u64 src = __shfl(my_img, (int)qd * 4);
            const u32 nb = __shfl(img_bytes, (int)qd * 4);
            if (!src) continue;
            gcu32x4 *s = (gcu32x4 *)src;
            u32x4 *dd = (u32x4 *)(slots + (u64)(wq0 + qd) * lds_per_item);
            for (u32 j = lane; j < ((nb + 15) >> 4); j += WAVE) dd[j] = s[j];
        }
        __syncthreads();
        // lanes without a stream read and write the LDS of stream 0 (nothing of theirs is ever used)
        const u32 slot = active ? quad : 0u;
        const u8 *im = slots + (u64)slot * lds_per_item;
        u8 *ring = slots + (u64)slot * lds_per_item + (lds_per_item - ENC_RING_BYTES);
        gu8 *dump = to_global(dump_) + 16u * ((blockIdx.x * blockDim.x + tid) & (ENC_DUMP_BYTES / 16u - 1u));
        pay = chain_encode_o1_lds<PK>(im, ring, lrcp, data, n, ns, bits, (gcu8 *)rcptab, send, dump, order == 1, lane);
        if (!PK) pay |= chain_encode_o0_pipe(im, ring, lrcp, data, n, bits, (gcu8 *)rcptab, send, dump, order == 0, lane);
    } else {
        gcu8 *im = (gcu8 *)I->image;
        pay = chain_encode<1>(data, n, im, ns, bits, rcptab, send, order == 1, lane);
        pay |= chain_encode<0>(data, n, im, ns, bits, rcptab, send, order == 0, lane);
    }
    if (active && (lane & 3) == 0) I->pay_len = pay;
    if (LDS_IMG) __syncthreads();                          // LDS is reused by the next share
    }
    walk.leave();
}

