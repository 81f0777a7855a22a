// r4x16_encode.hip — gfx950 kernels for rans_compress_to_4x16 (rANS_static4x16pr.c:1138-1345).
//
// Pipeline for a batch of blocks (one launch each, in stream order):
//   k_enc_front  : one wave per block.  Container header, histograms (hist8 / hist1_4,
//                  utils.h:80-202) with LDS counters, frequency normalisation (:116-179), the
//                  10/12-bit decision (:629-691, double precision, evaluation order kept),
//                  table serialisation (:182-325) incl. the nested order-0 compression of a
//                  large order-1 table (:767-780), and the encoder symbol table ("image").
//   k_enc_chain  : the hot loop (:442-485, :794-839).  A quad runs the 4 states of a stream
//                  backwards over the input; emitted 16-bit words are placed with a 4-bit
//                  ballot prefix inside the quad.
//   k_enc_finish : one wave per block.  Chooses CAT fall-back (:1332-1337), assembles
//                  header + table + payload into the caller's slot, writes size and status.
#include <stdlib.h>
#include "r4x16_dev.h"

#define FRONT_DYN_LDS  36864u                           // LDS counters: alphabets up to 96 symbols (three workgroups per CU;
                                                        // 61,440 bytes / two per CU: +4 % on PACK|RLE blocks, +2 % on 64 KiB ones)
#define FRONT_LDS_NSYM 96u

// ---------------------------------------------------------------------------------------------
// rANS_static4x16pr.c:360-372, same expression, same evaluation order, in double.
// ---------------------------------------------------------------------------------------------
// (the expression itself lives in r4x16_common.h: the stripe kernels need it too)
__host__ __device__ static inline u32 compress_bound(u32 size, int order) { return r4x16_bound_hd(size, order); }

// ---------------------------------------------------------------------------------------------
// normalise_freq, rANS_static4x16pr.c:116-163, over `cnt` counters F[0..cnt) (zero = absent).
// Integer types as in the reference.  One lane.
// ---------------------------------------------------------------------------------------------
__device__ int normalise_freq(u32 *F, u32 cnt, int size, u32 tot)
{
    int retried = 0;
    if (!size) return 0;
    for (;;) {
        const u64 scale = ((u64)tot << 31) / (u64)(long)size + (u64)(long)((1 << 30) / size);
        u32 best = 0, arg = 0;
        int sum = 0;
        for (u32 j = 0; j < cnt; j++) {
            u32 f = F[j];
            if (!f) continue;
            if (best < f) { best = f; arg = j; }
            f = (u32)(((u64)f * scale) >> 31);
            if (f == 0) f = 1;
            F[j] = f;
            sum += (int)f;
        }
        int adjust = (int)(tot - (u32)sum);
        if (adjust > 0) {
            F[arg] += (u32)adjust;
        } else if (adjust < 0) {
            const u32 need = (u32)(-adjust);
            if (F[arg] > need && (retried || F[arg] / 2 >= need)) {
                F[arg] -= need;
            } else if (!retried) {
                retried = 1;
                size = sum;
                continue;
            } else {
                adjust += (int)(F[arg] - 1);
                F[arg] = 1;
                for (u32 j = 0; adjust && j < cnt; j++) {
                    if (F[j] < 2) continue;
                    const int take = (F[j] > (u32)(-adjust)) ? adjust : (int)(1 - F[j]);
                    F[j] += (u32)take;
                    adjust -= take;
                }
            }
        }
        return F[arg] > 0 ? 0 : -1;
    }
}

// ---------------------------------------------------------------------------------------------
// The chain encoder.  lane&3 = chain, lane>>2 = stream.  Returns the number of bytes written
// backwards from scratch_end (16 bytes of states + 2 per emitted word), same in all 4 lanes.
//
// Per step and chain (rANS_word.h:281-321): if x >= x_max emit the low 16 bits and shift;
// then x += bias + ((x * rcp) >> rcp_shift) * cmpl_freq.  Within one step the reference emits
// in chain order 3,2,1,0 onto a descending pointer, so chain k's word lands
// 2 * (1 + #emitting chains above k) below the step's starting pointer.
//
// Step schedule of a stream of n bytes (lock-step for its four chains):
//   order-0 (:442-459): step s codes group g = gtop - s, chain k takes byte 4g+k; the top group
//                       may be partial.
//   order-1 (:794-834): chain k owns quarter k; chain 3 first codes the n - 4q tail bytes alone,
//                       then all four walk their quarters backwards with the previous byte as
//                       context, and the first byte of each quarter is coded in context 0.
// ---------------------------------------------------------------------------------------------

// A prefetched symbol is two words: rcp = rcptab[freq] and sf = start | freq << 16.
// RansEncPutSymbol (rANS_word.h:281-321) with RansEncSymbolInit's parameters (:190-266) derived on
// the fly: x_max = freq << (31 - bits); q = (x * rcp) >> (32 + ceil(log2 freq) - 1) is the exact
// quotient x / freq for freq >= 2 (Alverson), and freq == 1 takes q = x; then
// x' = x + start + q * (M - freq)  ==  ((x / freq) << bits) + x % freq + start.
__device__ __forceinline__ bool enc_wants_emit(u32 x, u32 sf, u32 bits)
{
    return x >= ((sf >> 16) << (31 - bits));
}
__device__ __forceinline__ u32 enc_rcp(gcu32 *rcptab, u32 f)
{
    return rcptab[f < RCPTAB_ENTRIES ? f : 0u];          // idle lanes may hold garbage: stay inside the table
}
__device__ __forceinline__ u32 enc_advance(u32 x, u32 rcp, u32 sf, u32 bits)
{
    const u32 f = sf >> 16, start = sf & 0xffffu;
    const u32 rsh = 31u - (u32)__clz((int)(f - 1u));     // ceil(log2 f) - 1 for f >= 2
    u32 q = __umulhi(x, rcp) >> (rsh & 31u);
    q = f < 2u ? x : q;
    return x + start + q * ((1u << bits) - f);
}

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
#define ENC_RING_BYTES 144u          // ring + a 2-byte dump slot for lanes that do not emit (+ pad)
#define ENC_LRCP_BYTES 16400u        // RCPTAB_ENTRIES dwords, padded to 16
struct EncOut {
    u8 *ring;            // LDS
    u32 ring126;         // LDS address of the ring's last word slot
    gu8 *send;           // scratch_end of the stream
    gu8 *dump;           // this lane's 16 bytes of the dump area
    u32 written;         // words emitted by the quad so far
    u32 flushed;         // 64-byte halves already read out of the ring
    u32 k, lane;
    bool active;
    u32x4 held;          // a half read out of the ring, stored one double-trip later
    gu8 *held_dst;
    // rANS_word.h:281-321 for one symbol; x is this lane's state.  pk = start | freq << 16.
    // q = x / freq < 2^21 once x < x_max, so q * (M - freq) is a 24-bit multiply (mod 2^32).
    __device__ __forceinline__ void step(u32 &x, bool live, u32 rcp, u32 pk, u32 bits)
    {
        const u32 f = pk >> 16, start = pk & 0xffffu;
        // the compare's own lane mask, and-ed with the live lanes on the scalar side (a ballot of the
        // combined predicate would be rebuilt through a select and a second compare)
        const bool over = x >= (f << (31u - bits));
        const u64 m = __ballot(over) & __ballot(live);
        const u32 em = (u32)(m >> (lane & ~3u)) & 0xfu;
        const bool emit = live && over;
        const u32 j = written + __popc(em >> (k + 1));
        const u32 j63 = emit ? (j & 63u) : ~0u;                          // -1: the dump slot at ring + 128
        *(LAS u16 *)(unsigned long)(ring126 - 2u * j63) = (u16)x;
        const u32 xs = emit ? x >> 16 : x;
        written += __popc(em);
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
        const bool due = active && (written >> 5) != flushed;
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
        const bool due = active && (written >> 5) != flushed;
        held = *(const u32x4 *)(ring + ((flushed & 1u) ? 0u : 64u) + 16u * k);
        held_dst = due ? send - 64ull * (flushed + 1u) + 16u * k : dump;
        flushed += due ? 1u : 0u;
    }
    __device__ __forceinline__ void flush_drain()
    {
        *(GAS u32x4_unaligned *)held_dst = held;
        held_dst = dump;
    }
    // the words still in the ring, then the four states (RansEncFlush in order 3,2,1,0, :482-485)
    __device__ __forceinline__ u32 finish(u32 x)
    {
        flush();
        const u32 first = 32u * flushed;
        const u32 rem = active ? written - first : 0u;
        for (u32 i = k; wave_any(i < rem); i += 4) {
            if (i < rem) {
                const u32 j = first + i;
                *(gu16 *)(send - 2ull * (j + 1u)) = *(const u16 *)(ring + ((~j << 1) & 126u));
            }
        }
        if (active) *(gu32 *)(send - 2ull * written - 16 + 4 * k) = x;
        return active ? 2 * written + 16 : 0;
    }
};

// PK: packed rows (r4x16_common.h) and a reciprocal table of the 1,025 frequencies a 10-bit table can hold.
template <bool PK>
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
    EncOut o{ring, (u32)(unsigned long)(LAS u8 *)ring + 126u, scratch_end, dump, 0u, 0u, k, lane, active, {0, 0, 0, 0}, dump};
    u32 x = RANS_LOW;
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
    EncOut o{ring, (u32)(unsigned long)(LAS u8 *)ring + 126u, scratch_end, dump, 0u, 0u, k, lane, active, {0, 0, 0, 0}, dump};
    u32 x = RANS_LOW;
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

// Hot form, order-0: one row of entries in LDS; chain k takes bytes 4g+k for descending g.
template <class DP>     // DP: gcu8* (stream in HBM) or const u8* (small stream staged in LDS)
__device__ __forceinline__ u32 chain_encode_o0_lds(const u8 *img_lds, DP data, u32 n, u32 bits,
                                                   gcu32 *rcptab, gu8 *scratch_end, bool active, u32 lane,
                                                   const u32 *rcp_by_sym = nullptr /* LDS: reciprocal per compact symbol */)
{
    const u32 k = lane & 3;
    const u8 *idx = img_lds;
    const u16 *cum = (const u16 *)(img_lds + ENC_IMG_IDX);
    u32 x = RANS_LOW, written = 0;
    const u32 gtop = (active && n) ? (n - 1) >> 2 : 0;
    u32 nsteps = (active && n) ? gtop + 1 : 0;
    const u32 first = (active && n && 4 * gtop + k >= n) ? 1 : 0;    // top group may be partial
    const u32 p0 = 4 * (gtop - first) + k;                           // byte of this lane's first step
    // Three look-ups ahead of the state update, one stage per step (symbols are known in advance):
    // byte of step s+3, its compact index for step s+2, table entry and reciprocal for step s+1.
    auto ldbyte = [&](u32 st) -> u32 { return (st >= first && st < nsteps) ? (u32)data[p0 - 4 * (st - first)] : 0u; };
    auto entry = [&](u32 si) -> u32x2 {
        const u32 c0 = cum[si], c1 = cum[si + 1];
        u32x2 e = {rcp_by_sym ? rcp_by_sym[si] : enc_rcp(rcptab, c1 - c0), c0 | ((c1 - c0) << 16)};
        return e;
    };
    u32 b2 = ldbyte(2);
    u32 si1 = idx[ldbyte(1)];
    u32x2 e0 = entry(idx[ldbyte(0)]);
    for (u32 s = 0; wave_any(s < nsteps); s++) {
        const bool live = s >= first && s < nsteps;
        const u32 b3 = ldbyte(s + 3);
        const u32 si2 = idx[b2];
        const u32x2 e1 = entry(si1);
        const bool emit = live && enc_wants_emit(x, e0.y, bits);
        const u32 em = quad_ballot(emit, lane);
        if (emit) {
            const u32 above = __popc(em >> (k + 1));
            *(gu16 *)(scratch_end - 2 * (written + above + 1)) = (u16)x;
            x >>= 16;
        }
        written += __popc(em);
        const u32 xn = enc_advance(x, e0.x, e0.y, bits);
        if (live) x = xn;
        e0 = e1; si1 = si2; b2 = b3;
    }
    if (active) *(gu32 *)(scratch_end - 2 * written - 16 + 4 * k) = x;
    return active ? 2 * written + 16 : 0;
}

// ---------------------------------------------------------------------------------------------
// Shared state of k_enc_front.
// ---------------------------------------------------------------------------------------------
struct EncShared {
    alignas(16) u32 F[256];   // order-0 counters / scratch row (k_enc_tables, order 1: 64 pairs of doubles)
    u32 T[256];          // order-1: context totals (by compact index)
    int S[256];          // order-1: per-context target from compute_shift (by compact index)
    u32 rowlen[256];     // order-1: serialised length of each row
    u8  present[256];
    u8  idx_of[256];     // byte -> compact
    u8  alpha[256];      // compact -> byte
    u8  pmask[256];      // terms present in this row
    u32 nsym, tab_len, bits;
    i32 status;
    u32 pk_n, pk_meta_len, pk_len;       // wg_pack results
    u32 rl_nsyms, rl_lits, rl_runs;      // wg_rle_split results
};

// A block's order-1 pair counters in global memory (alphabets beyond the LDS limit count there; all are handed from
// k_enc_front to k_enc_tables there): up to 256 x 256 dwords at the bottom of the block's backward-write area, which
// nothing else touches before the chain kernel runs (r4x16_api.hip sizes that area to at least ENC_F_BYTES).
__device__ __forceinline__ u32 *enc_pair_counters(const EncWs &ws, u32 b) { return (u32 *)(ws.scratch + (u64)b * ws.scratch_stride); }

// ---- wave histogram of bytes (hist8, utils.h:80-102) into S.F ---------------------------------
__device__ void wave_hist8(const u8 *data, u32 n, u32 *F, u32 lane)
{
    for (u32 j = lane; j < 256; j += WAVE) F[j] = 0;
    wsync();
    u32 head = (u32)((16 - ((u64)data & 15)) & 15);
    if (head > n) head = n;
    if (lane < head) atomicAdd(&F[data[lane]], 1u);
    const u32 body = (n - head) >> 4;
    const uint4 *v = (const uint4 *)(data + head);
    for (u32 i = lane; i < body; i += WAVE) {
        const uint4 w = v[i];
        const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            atomicAdd(&F[ww[c] & 0xff], 1u);
            atomicAdd(&F[(ww[c] >> 8) & 0xff], 1u);
            atomicAdd(&F[(ww[c] >> 16) & 0xff], 1u);
            atomicAdd(&F[ww[c] >> 24], 1u);
        }
    }
    const u32 done = head + body * 16;
    if (done + lane < n) atomicAdd(&F[data[done + lane]], 1u);
    wsync();
}

// put_alphabet, rANS_static4x16pr.c:182-206.  One lane.
__device__ u32 put_alphabet(u8 *cp, const u8 *present)
{
    u8 *start = cp;
    u32 implicit = 0;
    for (u32 j = 0; j < 256; j++) {
        if (!present[j]) continue;
        if (implicit) { implicit--; continue; }
        *cp++ = (u8)j;
        if (j && present[j - 1]) {
            u32 kk = j + 1;
            while (kk < 256 && present[kk]) kk++;
            implicit = kk - (j + 1);
            *cp++ = (u8)implicit;
        }
    }
    *cp++ = 0;
    return (u32)(cp - start);
}

// Order-0 stream front end (rANS_static4x16pr.c:405-435): histogram, two normalisations, table
// bytes to `tab`, encoder row to `imgrow`.  All lanes call.  Sets S.tab_len / S.status.
// enc_o0_tables expects the byte histogram of the data in S.F.
// normalise_freq over 256 counters by the whole wave, four counters per lane.  Same integer arithmetic as
// the one-lane form (sums are exact, the arg-max keeps the reference's "first largest" rule); the rarely
// taken tail that spreads a deficit over the symbols stays with one lane.  Returns 0 / -1 in every lane.
__device__ int wave_normalise_freq256(u32 *F, int size, u32 tot, u32 lane)
{
    int retried = 0;
    if (!size) return 0;
    for (;;) {
        const u64 scale = ((u64)tot << 31) / (u64)(long)size + (u64)(long)((1 << 30) / size);
        u32 best = 0, arg = 0;
        int sum = 0;
#pragma unroll
        for (u32 c = 0; c < 4; c++) {
            const u32 j = 4 * lane + c;
            u32 f = F[j];
            if (!f) continue;
            if (best < f) { best = f; arg = j; }
            f = (u32)(((u64)f * scale) >> 31);
            if (f == 0) f = 1;
            F[j] = f;
            sum += (int)f;
        }
        sum = (int)wave_sum((u32)sum);
        u32 wbest = best;
#pragma unroll
        for (int d = WAVE / 2; d; d >>= 1) { const u32 t = __shfl_xor(wbest, d); wbest = t > wbest ? t : wbest; }
        u32 warg = (best == wbest && best) ? arg : 0xffffu;          // first index holding the largest count
#pragma unroll
        for (int d = WAVE / 2; d; d >>= 1) { const u32 t = __shfl_xor(warg, d); warg = t < warg ? t : warg; }
        if (warg == 0xffffu) warg = 0;                               // all counters zero (the reference keeps arg = 0)
        wsync();
        int ret = 0, again = 0;
        if (lane == 0) {
            int adjust = (int)(tot - (u32)sum);
            if (adjust > 0) {
                F[warg] += (u32)adjust;
            } else if (adjust < 0) {
                const u32 need = (u32)(-adjust);
                if (F[warg] > need && (retried || F[warg] / 2 >= need)) {
                    F[warg] -= need;
                } else if (!retried) {
                    again = 1;
                } else {
                    adjust += (int)(F[warg] - 1);
                    F[warg] = 1;
                    for (u32 j = 0; adjust && j < 256; j++) {
                        if (F[j] < 2) continue;
                        const int take = (F[j] > (u32)(-adjust)) ? adjust : (int)(1 - F[j]);
                        F[j] += (u32)take;
                        adjust -= take;
                    }
                }
            }
            ret = F[warg] > 0 ? 0 : -1;
        }
        again = __shfl(again, 0);
        ret = __shfl(ret, 0);
        wsync();
        if (again) { retried = 1; size = sum; continue; }
        return ret;
    }
}

// encode_alphabet (:182-206) from a 256-bit presence mask held as four scalars: walks the set bits, no
// memory reads.  One lane.  Returns bytes written.
__device__ u32 put_alphabet_mask(u8 *cp, const u64 pm[4])
{
    u8 *start = cp;
    auto has = [&](u32 j) -> bool { return j < 256 && ((pm[j >> 6] >> (j & 63)) & 1ull); };
    u32 j = 0;
    for (;;) {
        // next present symbol at or after j
        u32 w = j >> 6;
        u64 m = w < 4 ? pm[w] & (~0ull << (j & 63)) : 0ull;
        while (!m && ++w < 4) m = pm[w];
        if (!m) break;
        j = 64 * w + (u32)__ffsll((unsigned long long)m) - 1;
        *cp++ = (u8)j;
        if (j && has(j - 1)) {
            u32 kk = j + 1;
            while (has(kk)) kk++;
            *cp++ = (u8)(kk - (j + 1));
            j = kk;                                                    // the implicit run is skipped
        } else j++;
    }
    *cp++ = 0;
    return (u32)(cp - start);
}

__device__ void enc_o0_tables(u32 n, u8 *tab, u8 *image, EncShared &S, u32 lane)
{
    u16 *imgrow = (u16 *)(image + ENC_IMG_IDX);      // cum[0..256]
    for (u32 j = lane; j < 256; j += WAVE) image[j] = (u8)j;       // order-0: symbols index the row directly
    u32 target = pow2_ceil(n);
    if (target > (1u << O0_BITS)) target = 1u << O0_BITS;
    if (lane == 0) S.status = ST_OK;
    wsync();
    if (wave_normalise_freq256(S.F, (int)n, target, lane) < 0) { if (lane == 0) S.status = ST_TABLE; }
    // table bytes: alphabet from the presence mask, then the frequencies as varints at prefix offsets
    u32 f[4], vl[4], mine = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { f[c] = S.F[lane * 4 + c]; vl[c] = f[c] ? var_len(f[c]) : 0u; mine += vl[c]; }
    u64 pm[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        // symbols 64w .. 64w+63 live in lanes 16w .. 16w+15, four per lane
        u64 bits4 = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            u64 x = (__ballot(f[c] != 0) >> (16 * w)) & 0xffffull;     // lanes 16w .. 16w+15
            x = (x | (x << 24)) & 0x000000ff000000ffull;               // spread 16 bits to every fourth position
            x = (x | (x << 12)) & 0x000f000f000f000full;
            x = (x | (x << 6)) & 0x0303030303030303ull;
            x = (x | (x << 3)) & 0x1111111111111111ull;
            bits4 |= x << c;
        }
        pm[w] = bits4;
    }
    for (u32 j = lane; j < 256; j += WAVE) S.present[j] = (pm[j >> 6] >> (j & 63)) & 1ull;
    u32 alen = 0;
    if (lane == 0) alen = put_alphabet_mask(tab, pm);
    alen = __shfl(alen, 0);
    {
        u32 off = alen + wave_incl_scan(mine, lane) - mine;
#pragma unroll
        for (int c = 0; c < 4; c++) if (f[c]) { var_put(tab + off, f[c]); off += vl[c]; }
        const u32 total = __shfl(wave_incl_scan(mine, lane), WAVE - 1);
        if (lane == 0) S.tab_len = alen + total;
    }
    wsync();
    if (wave_normalise_freq256(S.F, (int)target, 1u << O0_BITS, lane) < 0) { if (lane == 0) S.status = ST_TABLE; }   // :426
    wsync();
    // cumulative starts by a wave scan, 4 symbols per lane
    u32 sum = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { f[c] = S.F[lane * 4 + c]; sum += f[c]; }
    u32 start = wave_incl_scan(sum, lane) - sum;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        imgrow[lane * 4 + c] = (u16)start;
        start += f[c];
    }
    if (lane == WAVE - 1) imgrow[256] = (u16)start;
    wsync();
}

__device__ void enc_o0_front(const u8 *data, u32 n, u8 *tab, u8 *image, EncShared &S, u32 lane)
{
    wave_hist8(data, n, S.F, lane);
    enc_o0_tables(n, tab, image, S, lane);
}

// ---------------------------------------------------------------------------------------------
// Histograms of a whole block by all FRONT_THREADS threads of the workgroup (LDS counters).
// hist8: utils.h:80-102.  hist1_4: utils.h:136-202 — every adjacent pair, the first byte seen in
// context 0 — on compact symbol indices, so that an alphabet of n symbols needs n*n counters.
// ---------------------------------------------------------------------------------------------
#define FRONT_THREADS 256u

// `priv`: 16 x 256 scratch counters; threads spread over 16 private copies (tid & 15) so that the
// few hot symbols of quality data do not serialise the LDS atomics of a whole wave.
__device__ __forceinline__ void wg_hist8(const u8 *data, u32 n, u32 *Fout, u32 *priv, u32 tid)
{
    // (copy stride 257 dwords: each copy starts one LDS bank further; with 256 all sixteen sat on the same banks)
    for (u32 j = tid; j < 16 * 257; j += FRONT_THREADS) priv[j] = 0;
    __syncthreads();
    u32 *F = priv + 257 * (tid & 15);
    // 16-byte pieces, four in flight per thread (each is requested three pieces before it is counted: the
    // passes are memory-latency bound otherwise, eight waves per CU cannot hide an HBM round trip per piece)
    const u32 full = n >> 4;
    auto ld = [&](u32 pi) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (pi < full) v = *(GAS const u32x4_unaligned *)(to_global(data) + 16ull * pi);
        return v;
    };
    auto count = [&](u32x4 w, u32 pi) {
        if (pi >= full) return;
        const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            atomicAdd(&F[ww[c] & 0xff], 1u);
            atomicAdd(&F[(ww[c] >> 8) & 0xff], 1u);
            atomicAdd(&F[(ww[c] >> 16) & 0xff], 1u);
            atomicAdd(&F[ww[c] >> 24], 1u);
        }
    };
    const u32 T = FRONT_THREADS;
    u32x4 q0 = ld(tid), q1 = ld(tid + T), q2 = ld(tid + 2 * T), q3 = ld(tid + 3 * T);
    for (u32 pi = tid; pi < full; pi += 4 * T) {
        count(q0, pi);         q0 = ld(pi + 4 * T);
        count(q1, pi + T);     q1 = ld(pi + 5 * T);
        count(q2, pi + 2 * T); q2 = ld(pi + 6 * T);
        count(q3, pi + 3 * T); q3 = ld(pi + 7 * T);
    }
    const u32 done = full * 16;
    if (done + tid < n) atomicAdd(&F[data[done + tid]], 1u);
    __syncthreads();
    {
        u32 t = 0;
#pragma unroll
        for (int c = 0; c < 16; c++) t += priv[257 * c + tid];
        Fout[tid] = t;                                   // FRONT_THREADS == 256
    }
    __syncthreads();
}

// present8 (utils.h:108-131): which byte values occur.  Order-1 needs only that of the byte histogram, and
// plain byte stores do not serialise on the hot symbols the way counting atomics do.  F[b] becomes 0 / 1.
__device__ __forceinline__ void wg_present8(const u8 *data, u32 n, u32 *F, u8 *flags, u32 tid)
{
    flags[tid] = 0;                                       // FRONT_THREADS == 256
    __syncthreads();
    const u32 full = n >> 4;
    auto ld = [&](u32 pi) -> u32x4 {
        u32x4 v = {0, 0, 0, 0};
        if (pi < full) v = *(GAS const u32x4_unaligned *)(to_global(data) + 16ull * pi);
        return v;
    };
    auto mark = [&](u32x4 w, u32 pi) {
        if (pi >= full) return;
        const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            flags[ww[c] & 0xff] = 1; flags[(ww[c] >> 8) & 0xff] = 1;
            flags[(ww[c] >> 16) & 0xff] = 1; flags[ww[c] >> 24] = 1;
        }
    };
    const u32 T = FRONT_THREADS;
    u32x4 q0 = ld(tid), q1 = ld(tid + T), q2 = ld(tid + 2 * T), q3 = ld(tid + 3 * T);
    for (u32 pi = tid; pi < full; pi += 4 * T) {
        mark(q0, pi);         q0 = ld(pi + 4 * T);
        mark(q1, pi + T);     q1 = ld(pi + 5 * T);
        mark(q2, pi + 2 * T); q2 = ld(pi + 6 * T);
        mark(q3, pi + 3 * T); q3 = ld(pi + 7 * T);
    }
    const u32 done = full * 16;
    if (done + tid < n) flags[data[done + tid]] = 1;
    __syncthreads();
    F[tid] = flags[tid];
    __syncthreads();
}

// Fp0 points at `copies` x ns*ns zeroed counters (LDS, or global with copies == 1); threads spread
// over the copies by tid, and the copies are summed into the first one at the end.
// counters of one copy: ns*ns, padded so that the copies start eight LDS banks apart
__device__ __forceinline__ u32 hist1_copy_stride(u32 ns, u32 copies) { return copies > 1 ? ((ns * ns + 31u) & ~31u) + 8u : ns * ns; }

template <class FP>
__device__ __forceinline__ void wg_hist1(const u8 *data, u32 n, FP Fp0, u32 ns, u32 copies, const u8 *idx_of, u32 tid)
{
    const u32 cs = hist1_copy_stride(ns, copies);
    FP Fp = Fp0 + (tid & (copies - 1)) * cs;
    // 16-byte pieces with the byte before them, four in flight per thread (see wg_hist8)
    const u32 full = n >> 4;
    struct Piece { u32x4 w; u32 before; };
    auto ld = [&](u32 pi) -> Piece {
        Piece p = {{0, 0, 0, 0}, 0};
        if (pi < full) { p.w = *(GAS const u32x4_unaligned *)(to_global(data) + 16ull * pi); p.before = pi ? to_global(data)[16ull * pi - 1] : 0u; }
        return p;
    };
    auto count = [&](const Piece &p, u32 pi) {
        if (pi >= full) return;
        const u32 ww[4] = {p.w.x, p.w.y, p.w.z, p.w.w};
        u32 ci[16];                                  // compact indices of this thread's bytes
#pragma unroll
        for (int c = 0; c < 4; c++) {
            ci[4 * c] = idx_of[ww[c] & 0xff]; ci[4 * c + 1] = idx_of[(ww[c] >> 8) & 0xff];
            ci[4 * c + 2] = idx_of[(ww[c] >> 16) & 0xff]; ci[4 * c + 3] = idx_of[ww[c] >> 24];
        }
        u32 prev = pi ? idx_of[p.before] : 0u;       // the first byte of the block is seen in context 0
#pragma unroll
        for (int c = 0; c < 16; c++) {
            atomicAdd(&Fp[prev * ns + ci[c]], 1u);
            prev = ci[c];
        }
    };
    const u32 T = FRONT_THREADS;
    Piece q0 = ld(tid), q1 = ld(tid + T), q2 = ld(tid + 2 * T), q3 = ld(tid + 3 * T);
    for (u32 pi = tid; pi < full; pi += 4 * T) {
        count(q0, pi);         q0 = ld(pi + 4 * T);
        count(q1, pi + T);     q1 = ld(pi + 5 * T);
        count(q2, pi + 2 * T); q2 = ld(pi + 6 * T);
        count(q3, pi + 3 * T); q3 = ld(pi + 7 * T);
    }
    if (tid == 0) {                                  // the last n % 16 bytes
        u32 prev = full ? idx_of[data[16 * full - 1]] : 0u;
        for (u32 i = 16 * full; i < n; i++) {
            const u32 cur = idx_of[data[i]];
            atomicAdd(&Fp[(i ? prev : 0u) * ns + cur], 1u);
            prev = cur;
        }
    }
    __syncthreads();
    // the three quarter starts are coded in context 0 (rANS_static4x16pr.c:720-723)
    if (tid >= 1 && tid < 4) atomicAdd(&Fp0[idx_of[data[tid * (n >> 2)]]], 1u);
    __syncthreads();
    if (copies > 1) {
        for (u32 j = tid; j < ns * ns; j += FRONT_THREADS) {
            u32 t = Fp0[j];
            for (u32 c = 1; c < copies; c++) t += Fp0[c * cs + j];
            Fp0[j] = t;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// The two transforms, by all FRONT_THREADS threads of the workgroup.  (One-wave forms were latency-bound:
// one dependent byte load per 64 bytes, 86 ms for 4,096 x 1 MiB blocks with X_PACK|X_RLE against 15 ms.)
// ---------------------------------------------------------------------------------------------
// hts_pack, pack.c:56-151.  S.F holds the byte histogram of data[0..n).  Each thread packs 16-byte
// pieces of the input (2 / 4 / 8 output bytes).  Ends on a workgroup barrier.
// `provisional`: S.F comes from the head of the block only.  Bytes it has not seen map to 0x80, and the return value says
// whether one turned up (or the head alone cannot tell): the caller then repeats the call on the full presence flags.
__device__ bool wg_pack(const u8 *data, u32 n, u8 *meta, u8 *out, EncShared &S, u32 tid, bool provisional)
{
    if (tid == 0) {
        u32 ns = 0;
        for (u32 j = 0; j < 256; j++)
            if (S.F[j]) { S.idx_of[j] = (u8)ns; S.alpha[ns] = (u8)j; ns++; }
            else S.idx_of[j] = 0x80;
        meta[0] = (u8)ns;                                 // 256 wraps to 0 (pack.c:74)
        if (ns <= 16) for (u32 j = 0; j < ns; j++) meta[1 + j] = S.alpha[j];
        S.pk_n = ns;
        S.pk_meta_len = ns > 16 ? 1 : ns + 1;
        const u32 per = ns > 16 ? 1 : ns > 4 ? 2 : ns > 2 ? 4 : ns > 1 ? 8 : 0;
        S.pk_len = ns > 16 ? n : per ? (n + per - 1) / per : 0;
    }
    __syncthreads();
    const u32 ns = S.pk_n;
    if (ns > 16) return provisional;                      // copy case (caller keeps `data`) - but exactly 256 symbols keep the
                                                          // flag (pack.c:74), and only the whole block can tell
    if (ns <= 1) return provisional;                      // constant input - as far as the head goes
    u32 seen = 0;
    const u32 per = ns > 4 ? 2 : ns > 2 ? 4 : 8;
    const u32 width = 8 / per;
    const u32 pieces = n >> 4;
    // (global-address-space accesses: a FLAT load or store also counts on the LDS counter, and every idx_of look-up
    //  below would wait for it; the next piece is requested before this one is packed)
    gu8 *gout = to_global(out);
    auto piece = [&](u32 pi) -> u32x4 { u32x4 v = {0, 0, 0, 0}; if (pi < pieces) v = *(GAS const u32x4_unaligned *)(to_global(data) + 16ull * pi); return v; };
    u32x4 ahead = piece(tid);
    for (u32 pi = tid; pi < pieces; pi += FRONT_THREADS) {
        const u32x4 v = ahead;
        ahead = piece(pi + FRONT_THREADS);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        u64 acc = 0;                                      // 16 symbols of `width` bits, first in the low bits
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const u32 i0 = S.idx_of[w[c] & 0xff], i1 = S.idx_of[(w[c] >> 8) & 0xff],
                      i2 = S.idx_of[(w[c] >> 16) & 0xff], i3 = S.idx_of[w[c] >> 24];
            const u64 four = (u64)(i0 | (i1 << width) | (i2 << (2 * width)) | (i3 << (3 * width)));
            acc |= four << (4 * width * c);
            seen |= i0 | i1 | i2 | i3;
        }
        gu8 *o = gout + (u64)pi * (16 / per);
        if (per == 2)      *(GAS u64_unaligned *)o = acc;
        else if (per == 4) *(GAS u32_unaligned *)o = (u32)acc;
        else               *(GAS u16_unaligned *)o = (u16)acc;
    }
    if (tid == 0) {                                       // the last n % 16 bytes
        const u32 nout = S.pk_len;
        for (u32 ob = (pieces * 16) / per; ob < nout; ob++) {
            u32 v = 0;
            const u32 i0 = ob * per;
            for (u32 k = 0; k < per && i0 + k < n; k++) { const u32 ix = S.idx_of[data[i0 + k]]; seen |= ix; v |= ix << (k * width); }
            out[ob] = (u8)v;
        }
    }
    wg_fence();
    return __syncthreads_or((int)(seen & 0x80u)) != 0;
}

// Phase timing of k_enc_front for variant builds (-DR4X16_PROF_FRONT; tools/front_phases.py): cycles of thread 0 between
// stamps, summed over the blocks.  Not compiled into the product.
#ifdef R4X16_PROF_FRONT
__device__ unsigned long long g_front_prof[16];
#define PROF_INIT unsigned long long prof_t = tid == 0 ? (unsigned long long)wall_clock64() : 0ull
#define PROF(k) do { if (tid == 0) { const unsigned long long n_ = (unsigned long long)wall_clock64(); atomicAdd(&g_front_prof[k], n_ - prof_t); prof_t = n_; } } while (0)
extern "C" __attribute__((visibility("default"))) int rans4x16_hip_debug_front_prof(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_front_prof), sizeof(g_front_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_front_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
__device__ unsigned long long g_tables_prof[16];
#define TPROF_INIT unsigned long long tprof_t = lane == 0 ? (unsigned long long)wall_clock64() : 0ull
#define TPROF(k) do { if (lane == 0) { const unsigned long long n_ = (unsigned long long)wall_clock64(); atomicAdd(&g_tables_prof[k], n_ - tprof_t); tprof_t = n_; } } while (0)
extern "C" __attribute__((visibility("default"))) int rans4x16_hip_debug_tables_prof(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tables_prof), sizeof(g_tables_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tables_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define PROF_INIT
#define PROF(k)
#define TPROF_INIT
#define TPROF(k)
#endif

// A thread's private byte stream to global memory, sixteen bytes per store.  (In the chunked RLE split every lane writes
// its own region: byte stores would be sixty-four one-byte transactions per instruction, 0.46 ms per 256 KiB block.)
struct ByteOut {
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
        for (u32 k = cnt; k < 16; k++) {         // bring the oldest byte down to byte 0
            acc.x = __builtin_amdgcn_alignbit(acc.y, acc.x, 8);
            acc.y = __builtin_amdgcn_alignbit(acc.z, acc.y, 8);
            acc.z = __builtin_amdgcn_alignbit(acc.w, acc.z, 8);
            acc.w >>= 8;
        }
        const u32 w[4] = {acc.x, acc.y, acc.z, acc.w};
        for (u32 k = 0; k < cnt; k++) p[k] = (u8)(k < 4 ? w[0] >> (8 * k) : k < 8 ? w[1] >> (8 * (k - 4)) : k < 12 ? w[2] >> (8 * (k - 8)) : w[3] >> (8 * (k - 12)));
        p += cnt; cnt = 0;
    }
    __device__ __forceinline__ void put_var(u32 v)          // var_put_u32, varint.h:85-104
    {
        u32 groups = 1;
        for (u32 t = v >> 7; t; t >>= 7) groups++;
        for (u32 g = groups; g-- > 0; ) put(((v >> (7 * g)) & 0x7f) | (g ? 0x80u : 0u));
    }
};

// var_put_u32 (varint.h:85-104) through a global-address-space pointer (see wg_pack on FLAT accesses)
__device__ __forceinline__ u32 var_put_g(gu8 *cp, u32 v)
{
    u32 groups = 1;
    for (u32 t = v >> 7; t; t >>= 7) groups++;
    for (u32 g = groups; g-- > 0; ) *cp++ = (u8)(((v >> (7 * g)) & 0x7f) | (g ? 0x80 : 0));
    return groups;
}

// rle_encode with automatic symbol choice, rle.c:48-138.  S.F holds the byte histogram.  The repeat
// counts and the split are taken by all threads, each on its own chunk of the input.  `tiles`: 5 KB of LDS.
// Results: S.rl_nsyms, S.rl_lits, S.rl_runs (bytes); symbols in S.alpha[0..nsyms).  Ends on a workgroup barrier.
#define RLE_TILE 16384u
__device__ void wg_rle_split(const u8 *data, u32 n, u8 *lits_end, u8 *runs_end, EncShared &S, u8 *tiles, u32 tid)
{
    const u32 lane = tid & (WAVE - 1);
    PROF_INIT;
    u32 *rep = S.T;                                      // repeats per symbol
    rep[tid] = 0;                                        // FRONT_THREADS == 256
    __syncthreads();
    {
        const u32 pieces = (n + 15) >> 4;
        // (a thread's next piece and the byte before it are requested before this piece is counted)
        struct Pc { u32x4 v; u32 before; };
        auto piece = [&](u32 pi) -> Pc {
            Pc r = {{0, 0, 0, 0}, 256u};
            const u32 off = pi * 16;
            if (pi >= pieces) return r;
            if (n - off >= 16) r.v = *(GAS const u32x4_unaligned *)(to_global(data) + off);
            else { u32 w[4] = {0, 0, 0, 0}; for (u32 c = 0; c < n - off; c++) w[c >> 2] |= (u32)to_global(data)[off + c] << (8 * (c & 3)); r.v = u32x4{w[0], w[1], w[2], w[3]}; }
            if (off) r.before = to_global(data)[off - 1];
            return r;
        };
        Pc ahead = piece(tid);
        for (u32 pi = tid; pi < pieces; pi += FRONT_THREADS) {
            const u32 off = pi * 16;
            const u32 cnt = n - off < 16 ? n - off : 16;
            const Pc cur_p = ahead;
            ahead = piece(pi + FRONT_THREADS);
            const u32 w[4] = {cur_p.v.x, cur_p.v.y, cur_p.v.z, cur_p.v.w};
            u32 prev = cur_p.before;
            u32 run = 0;                                  // repeats of `prev` not yet added
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const u32 cur = (w[c >> 2] >> (8 * (c & 3))) & 0xff;
                if (c < (int)cnt) {
                    if (cur == prev) run++;
                    else { if (run) atomicAdd(&rep[prev], run); run = 0; prev = cur; }
                }
            }
            if (run) atomicAdd(&rep[prev], run);
        }
    }
    __syncthreads();
    PROF(11);
    if (tid == 0) {
        u32 ns = 0;
        for (u32 j = 0; j < 256; j++) {
            const bool use = 2 * (u64)rep[j] > (u64)S.F[j];
            S.present[j] = use;
            if (use) S.alpha[ns++] = (u8)j;
        }
        S.rl_nsyms = ns;
    }
    __syncthreads();

    // The split itself: every thread owns one contiguous chunk of the input and walks it twice.
    //   walk A: literals in the chunk, the position of its first literal, the varint bytes of the runs that END inside
    //           the chunk, and the run left open at its end (the chunk's last literal, if that is an RLE symbol);
    //   between: the next literal after each chunk (a suffix minimum over the chunks' first literals) closes the open
    //           runs, and exclusive sums over the chunks place every chunk's literals and run bytes;
    //   walk B: the same walk, now writing.
    // A byte is a literal unless it repeats an RLE symbol (rle.c:121-133); an RLE-symbol literal is followed, in the run
    // stream, by varint(number of repeats behind it).  (The first version swept 16 KB LDS tiles from the top with five
    // workgroup barriers per tile: 1.2 ms per 256 KiB, 69 % of k_enc_front on q4 with X_PACK|X_RLE; this form: see DESIGN 6.)
    u32 *cF = (u32 *)tiles, *cL = cF + 256, *cV = cL + 256, *cP = cV + 256, *cN = cP + 256;   // first / literals / run bytes / open run / next literal
    const u32 NONE = 0xffffffffu;
    const u32 csz = ((n + FRONT_THREADS - 1) / FRONT_THREADS + 15u) & ~15u;            // chunk bytes, a multiple of 16
    const u32 c0 = tid * csz < n ? tid * csz : n, c1 = c0 + csz < n ? c0 + csz : n;
    // one walk; EMIT = false: count, EMIT = true: write at (lp, vp).  `open` = position of the RLE-symbol literal whose run is running.
    auto walk = [&](auto emitc, u32 &nlit, u32 &first, u32 &vbytes, u32 &open, gu8 *lp, gu8 *vp) {
        constexpr bool EMIT = decltype(emitc)::value;
        ByteOut lo{lp, {0, 0, 0, 0}, 0}, vo{vp, {0, 0, 0, 0}, 0};
        u32 prev = c0 ? to_global(data)[c0 - 1] : 256u;
        // a thread's pieces are consecutive, so their loads are dependent round trips to memory unless the next one
        // is requested before this one is looked at
        auto piece = [&](u32 p0) -> u32x4 {
            u32x4 v = {0, 0, 0, 0};
            if (p0 + 16 <= c1) v = *(GAS const u32x4_unaligned *)(to_global(data) + p0);
            else if (p0 < c1) {
                u32 w[4] = {0, 0, 0, 0};
                for (u32 c = 0; c < c1 - p0; c++) w[c >> 2] |= (u32)to_global(data)[p0 + c] << (8 * (c & 3));
                v = u32x4{w[0], w[1], w[2], w[3]};
            }
            return v;
        };
        u32x4 ahead = piece(c0);
        for (u32 p0 = c0; p0 < c1; p0 += 16) {
            const u32 cnt = c1 - p0 < 16 ? c1 - p0 : 16;
            const u32x4 v = ahead;
            ahead = piece(p0 + 16);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const u32 cur = (w[c >> 2] >> (8 * (c & 3))) & 0xff;
                if (c < (int)cnt) {
                    const bool lit = cur != prev || !S.present[cur];
                    if (lit) {
                        const u32 at = p0 + (u32)c;
                        if (open != NONE) {                                   // the run behind `open` ends here
                            const u32 run = at - open - 1;
                            if (EMIT) vo.put_var(run); else vbytes += var_len(run);
                        }
                        open = S.present[cur] ? at : NONE;
                        if (first == NONE) first = at;
                        if (EMIT) lo.put(cur); else nlit++;
                    }
                    prev = cur;
                }
            }
        }
        if (EMIT) {
            if (open != NONE) vo.put_var(cN[tid] - open - 1);                 // the run that leaves the chunk
            lo.flush(); vo.flush();
        }
    };
    u32 nlit = 0, first = NONE, vbytes = 0, open = NONE;
    PROF(12);
    walk(std::false_type{}, nlit, first, vbytes, open, (gu8 *)nullptr, (gu8 *)nullptr);
    cF[tid] = first; cL[tid] = nlit; cV[tid] = vbytes; cP[tid] = open;
    __syncthreads();
    PROF(13);
    if (tid == 0) {
        u32 nx = n;                                                            // next literal after the chunk
        for (int t = FRONT_THREADS - 1; t >= 0; t--) { cN[t] = nx; if (cF[t] != NONE) nx = cF[t]; }
        u32 al = 0, av = 0;
        for (u32 t = 0; t < FRONT_THREADS; t++) {
            const u32 v = cV[t] + (cP[t] != NONE ? var_len(cN[t] - cP[t] - 1) : 0u);
            const u32 l = cL[t];
            cL[t] = al; cV[t] = av;                                            // exclusive sums
            al += l; av += v;
        }
        S.rl_lits = al; S.rl_runs = av;
    }
    __syncthreads();
    {
        const u32 nl_all = S.rl_lits, nv_all = S.rl_runs;
        u32 d0 = 0, d1 = NONE, d2 = 0, op = NONE;
        PROF(14);
        walk(std::true_type{}, d0, d1, d2, op, to_global(lits_end) - nl_all + cL[tid], to_global(runs_end) - nv_all + cV[tid]);
    }
    wg_fence();
    __syncthreads();
    PROF(15);
}

// ---------------------------------------------------------------------------------------------
// k_enc_front
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)b, src), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(b >> 32), src);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

__device__ __forceinline__ double approx_log(double a)            // fast_log :620-623
{
    const long long bits = __double_as_longlong(a);
    return (double)(bits - 4606921278410026770LL) * 1.539095918623324e-16;
}

__global__ __launch_bounds__(FRONT_THREADS) void k_enc_front(BatchArgs a, EncWs ws, int base, u32 dyn_bytes)
{
    extern __shared__ __attribute__((aligned(16))) u8 dyn[];
    __shared__ EncShared S;
    __shared__ struct { i32 status; u32 go, order, dlen, nested_len, flags, hl, run; u64 data; double e10, e12; int max_tot; u32 wcnt[4]; } H;

    // Wave 0 runs the whole front end; waves 1..3 join only for the two histogram passes over the
    // block (the bulk of the memory traffic).  Inside wave-0-only code the ordering points are
    // wsync(); __syncthreads() appears only where all four waves meet.
    const u32 tid = threadIdx.x;
    const u32 lane = tid & (WAVE - 1);
    const bool w0 = tid < WAVE;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    PROF_INIT;
    EncDesc *D = &ws.desc[b];
    EncItem *I0 = &ws.items[b], *I1 = &ws.items[gridDim.x + b];   // payload items first, meta items after
    const u8 *in = a.in + a.in_off[i];
    const u32 in_size = a.in_size[i];
    const u32 cap = a.out_cap[i];
    int order = a.d_order ? a.d_order[i] : a.order;
    u8 *img = ws.images + (u64)b * ENC_IMG_BYTES;
    u8 *tab = ws.tab + (u64)b * TAB_BYTES;
    u8 *scratch = ws.scratch + (u64)b * ws.scratch_stride;
    u8 *scratch_end = scratch + ws.scratch_stride;

    // ---- container header (:1144-1237) ---------------------------------------------------------
    if (tid == 0) {
        H.run = 0;
        ws.stat[b].run = 0;
        EncItem *I2 = &ws.items[2 * gridDim.x + b];                        // the order-1 table as an order-0 stream (k_enc_tables)
        I0->active = 0; I1->active = 0; I0->pay_len = 0; I1->pay_len = 0; I0->packed = 0; I1->packed = 0;
        I2->active = 0; I2->pay_len = 0; I2->packed = 0;
        I0->blk = b; I1->blk = b; I2->blk = b;
        D->cat = 0; D->rle_on = 0; D->tab_len = 0; D->hdr_len = 0; D->dlen = 0; D->tab = (u64)tab; D->nest_on = 0;
        i32 st = ST_OK;
        u32 go = 0;
        H.flags = 0; H.hl = 0;
        if (cap < compress_bound(in_size, order)) st = ST_CAPACITY;
        // the backward-write area was sized from the caller's max_in_size: a larger block would run the
        // table and chain coders past the start of its slot, into its neighbour's
        else if ((u64)compress_bound(in_size, 0xc1) + 64u > ws.scratch_stride) st = ST_UNSUPPORTED;
        else {
            if (in_size <= 20) order &= ~X_STRIPE;                         // :1151
            if (order & X_STRIPE) st = ST_UNSUPPORTED;                     // host entry points split stripes
            else if (order & X_CAT) {                                      // :1218-1225
                D->hdr[0] = X_CAT;
                D->hdr_len = 1 + var_put(D->hdr + 1, in_size);
                D->cat = 1; D->data = (u64)in; D->dlen = in_size; D->flags = X_CAT;
            } else if ((order & (X_PACK | X_RLE)) && (in_size > ws.xf_stride)) {
                st = ST_UNSUPPORTED;                                       // batch was sized without transform staging
            } else {
                u32 flags = (u32)order & 0xff;
                u32 hl = 1;
                D->nosz = flags & X_NOSZ;
                if (!(flags & X_NOSZ)) hl += var_put(D->hdr + 1, in_size); // :1234-1235
                H.flags = flags; H.hl = hl;
                go = 1;
            }
        }
        D->status = st;
        H.status = st; H.go = go;
    }
    __syncthreads();
    if (H.status != ST_OK || !H.go) return;

    // From here to the histograms every branch depends on shared values only (uniform over the
    // workgroup); the transforms use all four waves, the rest is wave 0 between barriers.
    const u8 *data = in;
    u32 n = in_size;
    u32 flags = H.flags, hl = H.hl;

    // ---- X_PACK (:1244-1267) ----------------------------------------------------------------------
    if (flags & X_PACK) {
        if (n == 0) flags &= ~(u32)X_PACK;
        else {
            u8 *pbuf = ws.packed + (u64)b * ws.xf_stride;
            // hts_pack only asks WHICH bytes occur (pack.c:62-75): the presence pass, plain byte stores, instead of the
            // counting histogram, whose LDS atomics all but serialise on the two to sixteen symbols PACK is made for
            // Blocks of 256 KiB and more take the symbol set from their first 64 KiB and pack at once; the packing
            // pass notices a byte from outside that set, and only then is the block looked at in full (as for the
            // order-1 alphabet below: one read of the input instead of two).
            PROF(0);
            const bool head_only = n >= 262144u;
            wg_present8(data, head_only ? 65536u : n, S.F, S.pmask, tid);
            PROF(1);
            if (wg_pack(data, n, D->hdr + hl, pbuf, S, tid, head_only)) {
                wg_present8(data, n, S.F, S.pmask, tid);
                wg_pack(data, n, D->hdr + hl, pbuf, S, tid, false);
            }
            PROF(2);
            if (S.pk_meta_len == 1 && S.pk_n != 256) flags &= ~(u32)X_PACK;    // > 16 symbols (:1249); 256 wraps to 0 and stays
            else {
                if (S.pk_n <= 16) data = pbuf;
                n = S.pk_len;
                hl += S.pk_meta_len;
                __syncthreads();
                if (tid == 0) H.hl = hl + var_put(D->hdr + hl, n);
                __syncthreads();
                hl = H.hl;
            }
        }
    }

    // ---- X_RLE (:1269-1319) -----------------------------------------------------------------------
    if (flags & X_RLE) {
        if (n == 0) flags &= ~(u32)X_RLE;
        else {
            u8 *lits_end = ws.lits + (u64)b * ws.xf_stride + ws.xf_stride;
            u8 *meta_end = ws.meta + (u64)b * (ws.xf_stride + 768) + (ws.xf_stride + 768);
            PROF(3);
            wg_hist8(data, n, S.F, (u32 *)dyn, tid);
            PROF(4);
            wg_rle_split(data, n, lits_end, meta_end, S, dyn, tid);
            PROF(5);
            const u32 nsy = S.rl_nsyms, nlits = S.rl_lits, nruns = S.rl_runs;
            const u32 mlen = nruns + nsy + 1;                              // :1282-1285
            if ((double)((u64)nlits + mlen) >= .99 * (double)n) {          // :1287
                flags &= ~(u32)X_RLE;
            } else {
                u8 *m = meta_end - mlen;
                if (tid == 0) m[0] = (u8)nsy;
                if (tid < nsy) m[1 + tid] = S.alpha[tid];                  // nsy <= 256 == FRONT_THREADS
                wg_fence();
                __syncthreads();
                // the meta is coded as an order-0 stream by the chain kernel (item I1)
                u8 *mtab = ws.metatab + (u64)b * META_TAB_BYTES;
                u8 *imgm = img + ENC_IMG_META;
                if (w0) {
                    enc_o0_front(m, mlen, mtab, imgm, S, lane);
                    if (lane == 0) {
                        D->rle_on = 1; D->rle_mlen = mlen; D->rle_lits = nlits;
                        D->rle_meta = (u64)m; D->meta_tab = (u64)mtab; D->meta_tab_len = S.tab_len;
                        if (S.status != ST_OK) D->status = S.status;
                        I1->data = (u64)m; I1->n = mlen; I1->image = (u64)imgm; I1->bits = O0_BITS; I1->order = 0;
                        I1->ns = 256; I1->img_bytes = ENC_IMG_IDX + 2u * 257u;
                        I1->scratch_end = (u64)(ws.scratch2 + (u64)b * ws.scratch2_stride + ws.scratch2_stride);
                        I1->active = S.status == ST_OK;
                    }
                }
                __syncthreads();
                if (S.status != ST_OK) return;
                data = lits_end - nlits;
                n = nlits;
            }
        }
    }

    {
        u32 o = order & 1;
        if (o && n < 8) { flags &= ~1u; o = 0; }                           // :1322-1325
        __syncthreads();
        if (tid == 0) {
            D->flags = flags; D->hdr[0] = (u8)flags; D->hdr_len = hl;
            D->data = (u64)data; D->dlen = n;
            H.order = o; H.data = (u64)data; H.dlen = n;
            H.run = n != 0;
        }
    }
    wg_fence();
    __syncthreads();                                                      // all four waves meet here
    if (!H.run) return;

    data = (const u8 *)H.data;
    n = H.dlen;
    PROF(6);

    // pass 1 over the block, all waves: byte histogram (hist8, utils.h:80-102) for order 0, presence only
    // (present8, :108-131) for order 1.  Order-1 blocks of 256 KiB and more look at their first 64 KiB only: the pair
    // counters get one extra "any other byte" symbol, and only if that one is ever hit (never, on quality data: a
    // block's alphabet is complete within its first few thousand bytes) is the exact two-pass route taken.  One read
    // of the input instead of two (the front end fetched 2.0 x the batch).
    // Smaller blocks, down to 64 KiB, look at their first quarter.
    const bool sampled = H.order == 1 && n >= 65536u;
    const u32 SAMPLE = n >= 262144u ? 65536u : (n >> 2) & ~15u;
    if (H.order == 0) wg_hist8(data, n, S.F, (u32 *)dyn, tid);
    else              wg_present8(data, sampled ? SAMPLE : n, S.F, S.pmask, tid);

    PROF(7);
    EncStat *ST = &ws.stat[b];
    if (H.order == 0) {
        ST->F0[tid] = S.F[tid];                                           // FRONT_THREADS == 256
        if (tid == 0) { ST->run = 1; ST->order = 0; ST->ns = 0; }
        return;
    }

    // ---- order-1 (:694-780) ---------------------------------------------------------------------
    u32 *Fg = enc_pair_counters(ws, b);
    u32 ns = 0, nsx = 0;
    bool f_in_lds = false;
    for (u32 attempt = sampled ? 0u : 1u; attempt < 2u; attempt++) {
        const bool prov = attempt == 0;                                   // provisional alphabet + overflow symbol
        if (attempt == 1 && sampled) wg_present8(data, n, S.F, S.pmask, tid);
        // compact alphabet F0 from the presence flags (0 forced in, :731); absent bytes map to the overflow symbol
        {   // one byte value per thread (FRONT_THREADS == 256): rank among the present ones by ballot + wave counts
            const bool pr = S.F[tid] != 0 || tid == 0;
            const u64 bal = __ballot(pr);
            if (lane == 0) H.wcnt[tid >> 6] = (u32)__popcll(bal);
            __syncthreads();
            u32 k = (u32)__popcll(bal & ((1ull << lane) - 1ull)), tot = 0;
            for (u32 w = 0; w < 4; w++) { const u32 c = H.wcnt[w]; tot += c; if (w < (tid >> 6)) k += c; }
            S.present[tid] = pr;
            if (pr) { S.idx_of[tid] = (u8)k; S.alpha[k] = (u8)tid; }
            else if (prov) S.idx_of[tid] = (u8)(tot < 255 ? tot : 255);
            if (tid == 0) S.nsym = tot;
        }
        __syncthreads();
        ns = S.nsym;
        nsx = prov ? ns + 1 : ns;                                         // row / column count of the counters
        f_in_lds = 4u * nsx * nsx <= dyn_bytes;
        if (prov && !f_in_lds) continue;                                  // large alphabets: straight to the exact route
        const u32 copies = !f_in_lds ? 1u : (16u * hist1_copy_stride(nsx, 4) <= dyn_bytes ? 4u : (8u * hist1_copy_stride(nsx, 2) <= dyn_bytes ? 2u : 1u));
        if (f_in_lds) { for (u32 j = tid; j < copies * hist1_copy_stride(nsx, copies); j += FRONT_THREADS) ((u32 *)dyn)[j] = 0; }
        else          { for (u32 j = tid; j < nsx * nsx; j += FRONT_THREADS) Fg[j] = 0; }
        __syncthreads();
        // pass 2 over the block: order-1 pair histogram, all waves
        PROF(8);
        if (f_in_lds) wg_hist1(data, n, (u32 *)dyn, nsx, copies, S.idx_of, tid);
        else          wg_hist1(data, n, Fg, nsx, 1u, S.idx_of, tid);
        PROF(9);
        if (!prov) break;
        // any pair with the overflow symbol (row ns or column ns of the (ns + 1)^2 counters)?
        bool hit = false;
        for (u32 j = tid; j <= ns; j += FRONT_THREADS) hit |= ((u32 *)dyn)[ns * nsx + j] != 0 || ((u32 *)dyn)[j * nsx + ns] != 0;
        if (!__syncthreads_or(hit)) break;
    }
    ST->F0[tid] = S.F[tid];
    // hand over to k_enc_tables: alphabet maps and the pair counters (compact, ns*ns)
    ST->present[tid] = S.present[tid]; ST->idx_of[tid] = S.idx_of[tid]; ST->alpha[tid] = S.alpha[tid];
    if (f_in_lds) for (u32 j = tid; j < ns * ns; j += FRONT_THREADS) Fg[j] = ((u32 *)dyn)[(j / ns) * nsx + j % ns];   // (nsx = ns + 1 on the sampled route)
    if (tid == 0) { ST->run = 1; ST->order = 1; ST->ns = ns; }
    PROF(10);
}

// ---------------------------------------------------------------------------------------------
// k_enc_tables: one wave per block, small LDS footprint so that thousands of blocks are resident:
// everything between the histograms and the chain kernel — normalisation, the 10/12-bit decision,
// table serialisation (nested coding included) and the encoder image.
// ---------------------------------------------------------------------------------------------
#define TABLES_DYN_LDS 10240u          // pair counters of alphabets up to 50 symbols
#define TABLES_LDS_NSYM 50u
// nested table stream: order-0 table (< 1 KB) at the bottom, payload written down from NEST_AREA.  The largest
// serialised order-1 table is 256 rows of 256 two-byte entries and the alphabet, ~132 KB; its order-0 coding stays
// below 1.05 x that + 16.
#define NEST_AREA 196608u

__global__ __launch_bounds__(WAVE) void k_enc_tables(BatchArgs a, EncWs ws, int base)
{
    extern __shared__ __attribute__((aligned(16))) u8 dyn[];
    __shared__ EncShared S;
    __shared__ struct { i32 status; double e10, e12; int max_tot; } H;
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    EncDesc *D = &ws.desc[b];
    EncItem *I0 = &ws.items[b];
    const EncStat *ST = &ws.stat[b];
    if (D->status != ST_OK || !ST->run) return;
    u8 *img = ws.images + (u64)b * ENC_IMG_BYTES;
    u8 *tab = ws.tab + (u64)b * TAB_BYTES;
    u8 *tabraw = tab + 1;                                 // the serialised order-1 table, behind its header byte
    u8 *scratch = ws.scratch + (u64)b * ws.scratch_stride;
    u8 *scratch_end = scratch + ws.scratch_stride;
    const u8 *data = (const u8 *)D->data;
    const u32 n = D->dlen;

    for (u32 j = lane; j < 256; j += WAVE) S.F[j] = ST->F0[j];
    if (lane == 0) H.status = ST_OK;
    wsync();
    if (ST->order == 0) {
        enc_o0_tables(n, tab, img, S, lane);
        if (lane == 0) {
            D->status = S.status;
            D->tab_len = S.tab_len;
            I0->data = (u64)data; I0->n = n; I0->image = (u64)img; I0->bits = O0_BITS; I0->order = 0;
            I0->ns = 256; I0->img_bytes = ENC_IMG_IDX + 2u * 257u;
            I0->scratch_end = (u64)scratch_end;
            I0->active = S.status == ST_OK;
        }
        return;
    }

    TPROF_INIT;
    const u32 ns = ST->ns;
    for (u32 j = lane; j < 256; j += WAVE) { S.present[j] = ST->present[j]; S.idx_of[j] = ST->idx_of[j]; S.alpha[j] = ST->alpha[j]; }
    if (lane == 0) S.nsym = ns;
    u32 *Fg = enc_pair_counters(ws, b);
    const bool f_in_lds = ns <= TABLES_LDS_NSYM;
    if (f_in_lds) for (u32 j = lane; j < ns * ns; j += WAVE) ((u32 *)dyn)[j] = Fg[j];
    wsync();
    u32 *Fp = f_in_lds ? (u32 *)dyn : Fg;
    // context totals = row sums
    for (u32 r = lane; r < ns; r += WAVE) {
        u32 t = 0;
        for (u32 j = 0; j < ns; j++) t += Fp[r * ns + j];
        S.T[r] = t;
    }
    wsync();

    TPROF(0);
    // ---- compute_shift (:629-691): row by row; terms in parallel, sums in reference order ------
    if (lane == 0) H.max_tot = 0;
    wsync();
    double e10 = 0, e12 = 0;                              // running sums, identical in every lane
    double2 *terms = (double2 *)S.F;                      // 64 pairs = the 1 KB of S.F, which the order-1 path does not use
    // log(1024 + k), log(4096 + k) for k < 64 sit one per lane (k is a count of "tiny" symbols in a row and
    // nearly always small); a row's pair is then a lane read instead of two dependent global loads
    const double ltab10 = ws.logtab[lane], ltab12 = ws.logtab[257 + lane];
    for (u32 r = 0; r < ns; r++) {
        const u32 Tr = S.T[r];
        const int target0 = (int)pow2_ceil(Tr);
        u32 tiny10 = 0, tiny12 = 0, nz = 0;
        for (u32 jb = 0; jb < ns; jb += WAVE) {                        // counts by ballot, no shuffle reductions
            const u32 j = jb + lane;
            const u32 f = j < ns ? Fp[r * ns + j] : 0u;
            const u32 qd = f ? (u32)target0 / f : 0u;
            nz += (u32)__popcll(__ballot(f != 0));
            tiny10 += (u32)__popcll(__ballot(f != 0 && qd > 1024u));
            tiny12 += (u32)__popcll(__ballot(f != 0 && qd > 4096u));
        }
        const double l10 = tiny10 < WAVE ? readlane_f64(ltab10, (int)tiny10) : ws.logtab[tiny10];
        const double l12 = tiny12 < WAVE ? readlane_f64(ltab12, (int)tiny12) : ws.logtab[257 + tiny12];
        // Terms in parallel, one symbol per lane; the sum must run in the reference's order (j ascending, one
        // accumulator over all rows).  The terms of the symbols with a count go to LDS side by side, and every lane
        // adds them up from there (same address in all lanes: a broadcast read, loads ahead of the dependent adds;
        // fetching them lane by lane with v_readlane took 60 % of this kernel on 64 KiB quality blocks).
        for (u32 jb = 0; jb < ns; jb += WAVE) {
            const u32 j = jb + lane;
            const u32 f = j < ns ? Fp[r * ns + j] : 0u;
            const u64 m = __ballot(f != 0);
            if (f) {
                int x = (int)((double)1024 * (double)f / (double)Tr);
                const double t10 = (double)f * (approx_log((double)(x > 1 ? x : 1)) - l10);
                x = (int)((double)4096 * (double)f / (double)Tr);
                const double t12 = (double)f * (approx_log((double)(x > 1 ? x : 1)) - l12);
                terms[__popcll(m & ((1ull << lane) - 1ull))] = double2{t10, t12};
            }
            wsync();
            const u32 cnt = (u32)__popcll(m);
#pragma unroll 4
            for (u32 k = 0; k < cnt; k++) {
                const double2 t = terms[k];
                e10 -= t.x;
                e12 -= t.y;
                e10 += 4;
                e12 += 6;
            }
            wsync();
        }
        if (lane == 0) {
            int target = target0;
            if (nz < 64 && target > 128) target /= 2;                  // :678-681
            if (target > 1024) target /= 2;
            if (target > 4096) target = 4096;
            S.S[r] = target;
            if (H.max_tot < target) H.max_tot = target;
        }
        wsync();
    }
    const u32 bits = (e10 / e12 < 1.01 || H.max_tot <= 1024) ? 10u : 12u;          // :685

    TPROF(1);
    // ---- per-context normalisation (:740-752), one context row per lane ---------------------------
    for (u32 rb = 0; rb < ns; rb += WAVE) {
        const u32 r = rb + lane;
        if (r < ns) {
            int target = S.S[r];
            if (bits == 10 && target > 1024) target = 1024;
            S.S[r] = target;
            if (normalise_freq(Fp + r * ns, ns, (int)S.T[r], (u32)target) < 0) H.status = ST_TABLE;
            // serialised length of the row (:295-325)
            u32 len = 0, zeros = 0;
            for (u32 j = 0; j < ns; j++) {
                const u32 f = Fp[r * ns + j];
                if (f) { if (zeros) { len += 2; zeros = 0; } len += var_len(f); }
                else zeros++;
            }
            if (zeros) len += 2;
            S.rowlen[r] = len;
        }
    }
    wsync();
    if (H.status != ST_OK) { if (lane == 0) D->status = H.status; return; }

    TPROF(2);
    // ---- serialise: alphabet, then rows at their prefix offsets -----------------------------------
    if (lane == 0) {
        u32 off = put_alphabet(tabraw, S.present);                    // :732
        for (u32 r = 0; r < ns; r++) { const u32 l = S.rowlen[r]; S.rowlen[r] = off; off += l; }
        S.tab_len = off;
    }
    wsync();
    for (u32 rb = 0; rb < ns; rb += WAVE) {
        const u32 r = rb + lane;
        if (r < ns) {
            // (a lane's row goes out in 16-byte pieces: byte stores, one per lane and instruction to forty-odd different
            //  lines, were 100 us of this kernel per block)
            ByteOut bo{to_global(tabraw) + S.rowlen[r], {0, 0, 0, 0}, 0};
            u32 zeros = 0;
            for (u32 j = 0; j < ns; j++) {
                const u32 f = Fp[r * ns + j];
                if (f) {
                    if (zeros) { bo.put(0); bo.put(zeros - 1); zeros = 0; }
                    bo.put_var(f);
                } else zeros++;
            }
            if (zeros) { bo.put(0); bo.put(zeros - 1); }
            bo.flush();
        }
    }
    const u32 tlen = S.tab_len;

    TPROF(3);
    // ---- encoder image: scale each row up to 1<<bits (:756) and build entries (:759-762) -----------
    for (u32 j = lane; j < 256; j += WAVE) img[j] = S.present[j] ? S.idx_of[j] : (u8)0;
    u16 *cumimg = (u16 *)(img + ENC_IMG_IDX);            // cum[r][0..ns]
    const bool packed = bits == 10 && ns >= ENC_PK_MIN_NS && ns <= ENC_PK_MAX_NS;
    const u32 W = enc_pk_row_dwords(ns);
    // one context row per lane, serial over its ns entries (a row per wave with a scan per 64 entries and two
    // barriers per packed row took 98 us per block for 46 rows)
    for (u32 rb = 0; rb < ns; rb += WAVE) {
        const u32 r = rb + lane;
        if (r >= ns) continue;
        u32 sh = 0;
        const u32 tgt = (u32)S.S[r];
        if (tgt != 0 && tgt != (1u << bits)) { u32 sz = tgt; while (sz < (1u << bits)) { sz *= 2; sh++; } }
        u32 x = 0;
        if (!packed) {
            u16 *row = cumimg + r * (ns + 1);
            for (u32 j = 0; j < ns; j++) { row[j] = (u16)x; x += Fp[r * ns + j] << sh; }
            row[ns] = (u16)x;
        } else {
            // packed row (r4x16_common.h): entry j at bit 11 j of the row's bit stream
            u32 *row = (u32 *)(img + ENC_IMG_IDX) + r * W;
            u64 acc = 0;                                  // bits not yet written, `have` of them
            u32 have = 0, wi = 0;
            for (u32 j = 0; j <= ns; j++) {
                acc |= (u64)x << have;
                have += 11;
                if (have >= 32) { row[wi++] = (u32)acc; acc >>= 32; have -= 32; }
                if (j < ns) x += Fp[r * ns + j] << sh;
            }
            if (wi < W) row[wi] = (u32)acc;
        }
    }
    if (packed && lane == 0) ((u32 *)(img + ENC_IMG_IDX))[ns * W] = 0;      // the pair window's second dword past the last row
    wsync();

    TPROF(4);
    // ---- table into the stream (:766-780) -----------------------------------------------------------
    // A table of 1,000 bytes and more is also coded as an order-0 stream, and the shorter form goes out.  That
    // stream is a chain item of its own (four lanes for ~3,000 steps inside this one-wave kernel were 38 % of it on
    // 64 KiB quality blocks); k_enc_finish compares the lengths.  Its byte histogram, table and image are made here.
    // Where it lives: the low NEST_AREA bytes of the block's backward-write area - the part of the order-1 bound
    // that stands for the table (257*257*3) and that the payload, written down from the top, never reaches.
    if (lane == 0) tab[0] = (u8)(bits << 4);
    if (1 + tlen > 1000) {
        u8 *ntab = scratch;                                           // the nested stream's own order-0 table
        u8 *img0 = img + ENC_IMG_NESTED;
        wsync();
        enc_o0_front(tabraw, tlen, ntab, img0, S, lane);
        if (lane == 0 && S.status == ST_OK) {
            EncItem *I2 = &ws.items[2 * gridDim.x + b];
            D->nest_on = 1; D->nest_tab = (u64)ntab; D->nest_tab_len = S.tab_len;
            I2->data = (u64)tabraw; I2->n = tlen; I2->image = (u64)img0; I2->bits = O0_BITS; I2->order = 0;
            I2->ns = 256; I2->img_bytes = ENC_IMG_IDX + 2u * 257u;
            I2->scratch_end = (u64)(scratch + NEST_AREA);
            I2->active = 1;
        }
    }
    TPROF(5);
    const u32 final_len = 1 + tlen;
    wsync();
    if (lane == 0) {
        D->tab_len = final_len;
        I0->data = (u64)data; I0->n = n; I0->image = (u64)img; I0->bits = bits; I0->order = 1;
        I0->ns = ns; I0->img_bytes = packed ? enc_pk_img_bytes(ns) : ENC_IMG_IDX + 2u * ns * (ns + 1);
        I0->packed = packed ? 1u : 0u;
        I0->scratch_end = (u64)scratch_end;
        I0->active = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// k_enc_chain: QPW streams per wave, one launch per LDS size class (see k_dec_chain).
// ---------------------------------------------------------------------------------------------
// LDS_IMG: a workgroup of up to four waves shares one LDS copy of the reciprocal table; each quad
// owns lds_per_item bytes (image, then the word ring).  Waves never meet again after the set-up.
// PK: the class holds packed order-1 streams only (10-bit tables): a 1,025-entry reciprocal table suffices.
#define ENC_LRCP_PK_BYTES 4112u      // 1,025 dwords, padded to 16
template <bool LDS_IMG, bool PK>
__global__ __launch_bounds__(256) void k_enc_chain(EncItem *items, const u32 *rcptab_, u8 *dump_, const u32 *list, const u32 *count,
                                                   int qpw, int spw, u32 lds_per_item)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 tid = threadIdx.x;
    const u32 lane = tid & (WAVE - 1);
    // qpw streams per workgroup, spw per wave (the first spw quads of each wave)
    const u32 wq = lane >> 2;
    const u32 quad = (tid >> 6) * (u32)spw + wq;
    // persistent: as many workgroups as are resident at once, each walking its share (see k_dec_chain)
    // the streams of this launch's class, grouped on the device (k_enc_classify, r4x16_launch_cls_group)
    const int nmine = (int)count[0];
    list += count[CLS_MAX];
    const int nwg = (nmine + qpw - 1) / qpw;
    for (int wg = (int)blockIdx.x; wg < nwg; wg += (int)gridDim.x) {
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
}

// ---------------------------------------------------------------------------------------------
// k_enc_finish
// ---------------------------------------------------------------------------------------------
#define FINISH_THREADS 256u
__global__ __launch_bounds__(FINISH_THREADS) void k_enc_finish(BatchArgs a, EncWs ws, int base)
{
    __shared__ u8 vbuf[16];
    __shared__ u32 vlen;
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    const EncDesc *D = &ws.desc[b];
    const EncItem *I0 = &ws.items[b], *I1 = &ws.items[gridDim.x + b], *I2 = &ws.items[2 * gridDim.x + b];
    u8 *out = a.out + a.out_off[i];
    const i32 st = D->status;
    if (st != ST_OK) {
        if (lane == 0) { a.status[i] = st; a.out_size[i] = 0; }
        return;
    }
    u32 pos = D->hdr_len;
    const u32 dlen = D->dlen;
    u32 flags = D->hdr[0];
    if (lane >= 1 && lane < D->hdr_len) out[lane] = D->hdr[lane];
    if (D->cat) {
        group_copy<FINISH_THREADS>(out + pos, (const u8 *)D->data, dlen, lane);
        pos += dlen;
    } else {
        if (D->rle_on) {                                              // :1294-1310
            const u32 mlen = D->rle_mlen;
            const u32 mpay = I1->pay_len;
            const u32 clen = D->meta_tab_len + mpay;
            const bool comp = clen < mlen;
            if (lane == 0) {
                u32 l = var_put(vbuf, comp ? mlen * 2 : mlen * 2 + 1);
                l += var_put(vbuf + l, D->rle_lits);
                if (comp) l += var_put(vbuf + l, clen);
                vlen = l;
            }
            __syncthreads();
            if (lane < vlen) out[pos + lane] = vbuf[lane];
            pos += vlen;
            if (comp) {
                group_copy<FINISH_THREADS>(out + pos, (const u8 *)D->meta_tab, D->meta_tab_len, lane);
                pos += D->meta_tab_len;
                group_copy<FINISH_THREADS>(out + pos, (const u8 *)I1->scratch_end - mpay, mpay, lane);
                pos += mpay;
            } else {
                group_copy<FINISH_THREADS>(out + pos, (const u8 *)D->rle_meta, mlen, lane);
                pos += mlen;
            }
        }
        const u32 pay = I0->active ? I0->pay_len : 0;
        // order-1 table: as serialised, or as the order-0 stream the chain kernel made of it (:766-780)
        u32 tab_len = D->tab_len;
        const u32 tlen = tab_len - 1;
        const u32 npay = D->nest_on ? I2->pay_len : 0;
        const u32 nlen = D->nest_tab_len + npay;
        const bool nested = D->nest_on && nlen + 6 < tab_len;         // :772
        if (nested) tab_len = 1 + var_len(tlen) + var_len(nlen) + nlen;
        const u32 plen = tab_len + pay;
        if (plen >= dlen) {                                           // :1332-1337
            flags = (flags & ~3u) | X_CAT | D->nosz;
            group_copy<FINISH_THREADS>(out + pos, (const u8 *)D->data, dlen, lane);
            pos += dlen;
        } else {
            if (nested) {
                __syncthreads();                                      // (vbuf may still be read for the RLE header)
                if (lane == 0) {
                    vbuf[0] = (u8)(((const u8 *)D->tab)[0] | 1);
                    u32 l = 1 + var_put(vbuf + 1, tlen);
                    l += var_put(vbuf + l, nlen);
                    vlen = l;
                }
                __syncthreads();
                if (lane < vlen) out[pos + lane] = vbuf[lane];
                pos += vlen;
                group_copy<FINISH_THREADS>(out + pos, (const u8 *)D->nest_tab, D->nest_tab_len, lane);
                pos += D->nest_tab_len;
                group_copy<FINISH_THREADS>(out + pos, (const u8 *)I2->scratch_end - npay, npay, lane);
                pos += npay;
            } else {
                group_copy<FINISH_THREADS>(out + pos, (const u8 *)D->tab, D->tab_len, lane);
                pos += D->tab_len;
            }
            group_copy<FINISH_THREADS>(out + pos, (const u8 *)I0->scratch_end - pay, pay, lane);
            pos += pay;
        }
    }
    if (lane == 0) {
        out[0] = (u8)flags;
        a.status[i] = ST_OK;
        a.out_size[i] = pos;
    }
}

// ---- host-callable launchers -------------------------------------------------------------------
extern "C" bool r4x16_first_on_device(u32 bit);                                          // r4x16_decode.hip
extern "C" void r4x16_launch_enc_front(const BatchArgs *a, const EncWs *ws, int base, int nblk, hipStream_t s)
{
    // static + dynamic LDS exceeds the 64 KB default; gfx950 has 160 KB per CU
    if (r4x16_first_on_device(2u))
        (void)hipFuncSetAttribute((const void *)k_enc_front, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    static const u32 dynb = getenv("R4X16_FRONT_LDS") ? (u32)atoi(getenv("R4X16_FRONT_LDS")) : FRONT_DYN_LDS;   // tuning aid
    hipLaunchKernelGGL(k_enc_front, dim3(nblk), dim3(FRONT_THREADS), dynb, s, *a, *ws, base, dynb);
}
// {LDS bytes per stream, streams per wave}; LDS is allocated in 1,280-byte granules.
// q4/q8 images are ~0.4 KB, an order-0 row 0.8 KB, q40 4.6 KB (16 x 4,800 = 60 granules: 2 waves, 32 streams per CU — fuller waves measured faster than more waves)
// LDS size classes: bytes per stream (image + word ring).  A workgroup takes as many streams as
// fit beside the shared reciprocal table, up to 64 (four waves); 1,280-byte allocation granules.
// (sizes are 16 mod 128: consecutive streams start four LDS banks apart, so that the eight streams of a
// 32-lane access group do not all hit the same bank when they touch the same offset)
static const u32 ENC_CLASSES[] = {656, 1296, 2576, 4752, 6416, 12816, 33296, 73616, 147344};
// packed rows (20..64 symbols, 10-bit tables): 46 symbols need 3,532 bytes -> 45 streams per CU beside the small
// reciprocal table (3,536 is 80 mod 128: consecutive streams start 20 banks apart)
static const u32 ENC_PK_CLASSES[] = {1168, 2064, 2832, 3536, 3728, 4752, 6416};
#define ENC_NCLS    ((u32)(sizeof(ENC_CLASSES) / sizeof(ENC_CLASSES[0])))
#define ENC_PK_NCLS ((u32)(sizeof(ENC_PK_CLASSES) / sizeof(ENC_PK_CLASSES[0])))
static int enc_class_qpw(u32 bytes, bool pk = false)
{
    const u32 room = 163840u - (pk ? ENC_LRCP_PK_BYTES : ENC_LRCP_BYTES);
    const u32 fit = room / bytes;
    static const int cap = getenv("R4X16_ENC_QPW_CAP") ? atoi(getenv("R4X16_ENC_QPW_CAP")) : 64;   // tuning aid
    return (int)(fit > (u32)cap ? (u32)cap : fit);
}
extern "C" void r4x16_launch_enc_tables(const BatchArgs *a, const EncWs *ws, int base, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_enc_tables, dim3(nblk), dim3(WAVE), TABLES_DYN_LDS, s, *a, *ws, base);
}
extern "C" int r4x16_resident_grid(size_t lds_bytes, int waves_per_wg, int wanted);      // r4x16_decode.hip
extern "C" int r4x16_cu_count(void);
struct EncClassTab { u32 n; u32 bytes[CLS_MAX]; u32 pk[CLS_MAX]; };     // classes: u16 images, then packed ones
__global__ __launch_bounds__(256) void k_enc_classify(const EncItem *items, int nitems, EncClassTab tab, u32 *cls, u32 *count)
{
    __shared__ u32 local[CLS_MAX];
    if (threadIdx.x < CLS_MAX) local[threadIdx.x] = 0;
    __syncthreads();
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < nitems) {
        u32 c = CLS_NONE;
        if (items[i].active) {
            const u32 need = items[i].img_bytes + ENC_RING_BYTES, pk = items[i].packed;
            c = tab.n;                                         // images too large for LDS (never a packed one)
            for (u32 k = 0; k < tab.n; k++) if (tab.pk[k] == pk && need <= tab.bytes[k]) { c = k; break; }
            atomicAdd(&local[c], 1u);
        }
        cls[i] = c;
    }
    __syncthreads();
    if (threadIdx.x < CLS_MAX && local[threadIdx.x]) atomicAdd(&count[threadIdx.x], local[threadIdx.x]);
}
extern "C" void r4x16_launch_cls_group(const u32 *cls, int nitems, u32 *count, u32 *list, hipStream_t s);   // r4x16_decode.hip
extern "C" void r4x16_launch_cls_zero(u32 *count, hipStream_t s);
extern "C" void r4x16_launch_enc_chain(const EncWs *ws, int nitems, hipStream_t s)
{
    {
        EncClassTab tab;
        tab.n = 0;
        for (const u32 bytes : ENC_CLASSES) { tab.pk[tab.n] = 0; tab.bytes[tab.n++] = bytes; }
        for (const u32 bytes : ENC_PK_CLASSES) { tab.pk[tab.n] = 1; tab.bytes[tab.n++] = bytes; }
        r4x16_launch_cls_zero(ws->cls_count, s);
        hipLaunchKernelGGL(k_enc_classify, dim3((nitems + 255) / 256), dim3(256), 0, s, (const EncItem *)ws->items, nitems, tab, ws->cls, ws->cls_count);
        r4x16_launch_cls_group(ws->cls, nitems, ws->cls_count, ws->cls_list, s);
    }
    if (r4x16_first_on_device(4u)) {
        (void)hipFuncSetAttribute((const void *)k_enc_chain<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void *)k_enc_chain<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    }
    u32 ci = 0;
    static const int force_qpw = getenv("R4X16_ENC_QPW") ? atoi(getenv("R4X16_ENC_QPW")) : 0;   // tuning aids
    static const int force_waves = getenv("R4X16_ENC_WAVES") ? atoi(getenv("R4X16_ENC_WAVES")) : 0;
    for (u32 cls = 0; cls < ENC_NCLS + ENC_PK_NCLS; cls++) {
        const bool pk = cls >= ENC_NCLS;
        const u32 bytes = pk ? ENC_PK_CLASSES[cls - ENC_NCLS] : ENC_CLASSES[cls];
        const u32 tuned = pk ? 3536u : 4752u;                // the class of the 46-symbol quality tables
        int qpw = (force_qpw && bytes == tuned) ? force_qpw : enc_class_qpw(bytes, pk);
        // One workgroup per CU is the best shape (measured: 30 streams per CU as 1 x 32 beat 2 x 16 by a
        // third and half-filled 64s by a fifth), so a batch that cannot fill the class's workgroups on
        // every CU gets smaller ones (items [0, n/3) are the payload streams).
        {
            const int cus = r4x16_cu_count();
            int want = (((nitems + 2) / 3 + cus - 1) / cus + 3) & ~3;
            if (want < 8) want = 8;
            if (qpw > want) qpw = want;
        }
        int waves = (qpw + 7) / 8;                         // about eight streams per wave measured best (fewer
        if (waves > 4) waves = 4;                          // lanes per LDS access, one wave per SIMD)
        if (force_waves && bytes == tuned) waves = force_waves;
        const int spw = (qpw + waves - 1) / waves;
        const size_t ldsb = (size_t)(pk ? ENC_LRCP_PK_BYTES : ENC_LRCP_BYTES) + (size_t)qpw * bytes;
        const int grid = r4x16_resident_grid(ldsb, waves, (nitems + qpw - 1) / qpw);
        void (*kern)(EncItem *, const u32 *, u8 *, const u32 *, const u32 *, int, int, u32) =
            pk ? k_enc_chain<true, true> : k_enc_chain<true, false>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * waves), ldsb, s,
                           ws->items, ws->rcptab, ws->dump, (const u32 *)ws->cls_list, (const u32 *)(ws->cls_count + ci), qpw, spw, bytes);
        ci++;
    }
    const int grid = (nitems + 15) / 16;
    hipLaunchKernelGGL((k_enc_chain<false, false>), dim3(grid), dim3(WAVE), 0, s, ws->items, ws->rcptab, ws->dump,
                       (const u32 *)ws->cls_list, (const u32 *)(ws->cls_count + ci), 16, 16, 0u);
}
extern "C" int r4x16_enc_residency(u32 nsym, int order, int *streams_per_wave, int *waves_per_cu)
{
    if (nsym == 0 || nsym > 256) return -1;
    // (the packed rows need a 10-bit table: what every BASELINE text chooses; a 12-bit stream keeps the u16 rows)
    const bool pk = order && nsym >= ENC_PK_MIN_NS && nsym <= ENC_PK_MAX_NS;
    const u32 need = (pk ? enc_pk_img_bytes(nsym) : order ? ENC_IMG_IDX + 2u * nsym * (nsym + 1) : ENC_IMG_IDX + 2u * 257u) + ENC_RING_BYTES;
    for (u32 cls = 0; cls < ENC_NCLS + ENC_PK_NCLS; cls++) {
        if ((cls >= ENC_NCLS) != pk) continue;
        const u32 bytes = pk ? ENC_PK_CLASSES[cls - ENC_NCLS] : ENC_CLASSES[cls];
        if (need > bytes) continue;
        const int qpw = enc_class_qpw(bytes, pk);
        int waves = (qpw + 7) / 8;
        if (waves > 4) waves = 4;
        *streams_per_wave = (qpw + waves - 1) / waves;
        *waves_per_cu = waves;                               // one workgroup per CU
        return qpw;
    }
    *streams_per_wave = 16; *waves_per_cu = 8;
    return 128;
}
extern "C" void r4x16_launch_enc_finish(const BatchArgs *a, const EncWs *ws, int base, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_enc_finish, dim3(nblk), dim3(FINISH_THREADS), 0, s, *a, *ws, base);
}
extern "C" u32 r4x16_compress_bound(u32 size, int order) { return compress_bound(size, order); }

// =============================================================================================
// rANS 4x8 encode (CRAM 3.0's codec: htscodecs/rANS_static.c:85-224, :409-615, rANS_byte.h).  SURVEY.md 8f-4.
//   k8_enc_front  : 256 threads per block: histograms (the 4x16 front end's passes: hist8 / present8 + hist1_4 with
//                   the three quarter starts, utils.h:80-202), normalisation to 4095 (order 0: 64-bit fixed point;
//                   order 1: double precision, one context row per thread, evaluation order of the reference),
//                   the table in the 4x8 grammar, the u16 cumulative image of the 4x16 encoder.
//   k8_enc_chain  : a quad per block; the general-form chain loop with BYTE renormalisation: a chain emits
//                   (x >= x_max) + (x >> 8 >= x_max) bytes, low byte first, on the quad's descending pointer,
//                   chains served in the order 3, 2, 1, 0 (rANS_byte.h:320-402).
//   k8_enc_finish : header (order, sizes), table, payload into the caller's slot.
// Plain first version (tables through L2, byte stores); in_size == 0 is refused (the reference divides by zero).
// =============================================================================================
#define X8_LOW_E (1u << 23)
__host__ __device__ static inline u32 compress_bound8(u32 size) { return (u32)((int)(1.05 * size) + 257 * 257 * 3 + 9); }   // :87
extern "C" u32 r4x8_compress_bound(u32 size) { return compress_bound8(size); }

// symbols j with F[j] != 0 (byte order = compact order) in the 4x8 table grammar (:143-169, :497-531); `F` is
// indexed by compact symbol, `alpha` maps to bytes.  cp == nullptr: only the length.
__device__ u32 x8_put_table(u8 *cp, const u32 *F, u32 fstride, const u8 *alpha, u32 ns)
{
    u32 len = 0, rle = 0;
    for (u32 c = 0; c < ns; c++) {
        const u32 f = F[c * fstride];
        if (!f) continue;
        if (rle) rle--;
        else {
            const u32 j = alpha[c];
            if (cp) cp[len] = (u8)j;
            len++;
            if (c && F[(c - 1) * fstride] && alpha[c - 1] + 1u == j) {
                u32 r = c + 1;
                while (r < ns && F[r * fstride] && alpha[r] == j + (r - c)) r++;
                rle = r - (c + 1);
                if (cp) cp[len] = (u8)rle;
                len++;
            }
        }
        if (f < 128) { if (cp) cp[len] = (u8)f; len++; }
        else { if (cp) { cp[len] = (u8)(128 | (f >> 8)); cp[len + 1] = (u8)(f & 0xff); } len += 2; }
    }
    if (cp) cp[len] = 0;
    return len + 1;
}

__global__ __launch_bounds__(FRONT_THREADS) void k8_enc_front(BatchArgs a, EncWs ws, int base)
{
    extern __shared__ __attribute__((aligned(16))) u8 dyn[];
    __shared__ EncShared S;
    __shared__ struct { i32 status; u32 order, ns, tab_len; } H;
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    EncItem *I0 = &ws.items[b];
    const u8 *data = a.in + a.in_off[i];
    const u32 n = a.in_size[i];
    int order = a.d_order ? a.d_order[i] : a.order;
    u8 *img = ws.images + (u64)b * ENC_IMG_BYTES;
    u8 *tab = ws.tab + (u64)b * TAB_BYTES;
    u8 *scratch_end = ws.scratch + (u64)b * ws.scratch_stride + ws.scratch_stride;

    if (tid == 0) {
        I0->active = 0; I0->pay_len = 0; I0->blk = b; I0->packed = 0;
        i32 st = ST_OK;
        if (n == 0) st = ST_EMPTY;
        else if (a.out_cap[i] < compress_bound8(n)) st = ST_CAPACITY;
        else if ((u64)compress_bound8(n) + 64u > ws.scratch_stride) st = ST_UNSUPPORTED;   // block larger than the call announced
        H.status = st;
        H.order = (order && n >= 4) ? 1u : 0u;                                  // :438
        ws.desc[b].status = st;
    }
    __syncthreads();
    if (H.status != ST_OK) return;

    if (H.order == 0) {
        wg_hist8(data, n, S.F, (u32 *)dyn, tid);
        if (tid == 0) {
            // normalise to 4095 (:106-133): the largest symbol absorbs the difference; one harder retry
            u64 tr = ((u64)4096 << 31) / n + (1u << 30) / n;
            for (;;) {
                int fsum = 0, m = 0, M = 0;
                for (int j = 0; j < 256; j++) {
                    int f = (int)S.F[j];
                    if (!f) continue;
                    if (m < f) { m = f; M = j; }
                    f = (int)(((u64)f * tr) >> 31);
                    if (f == 0) f = 1;
                    S.F[j] = (u32)f;
                    fsum += f;
                }
                fsum++;
                if (fsum < 4096) { S.F[M] += (u32)(4096 - fsum); break; }
                if (fsum - 4096 > (int)S.F[M] / 2) { tr = 2104533975u; continue; }
                S.F[M] -= (u32)(fsum - 4096);
                break;
            }
            for (u32 j = 0; j < 256; j++) S.alpha[j] = (u8)j;
            H.tab_len = x8_put_table(tab, S.F, 1, S.alpha, 256);
        }
        __syncthreads();
        img[tid] = (u8)tid;                                                      // identity index, one row of 257
        if (tid == 0) {
            u16 *cum = (u16 *)(img + ENC_IMG_IDX);
            u32 x = 0;
            for (u32 j = 0; j < 256; j++) { cum[j] = (u16)x; x += S.F[j]; }
            cum[256] = (u16)x;
        }
        H.ns = 256;
    } else {
        wg_present8(data, n, S.F, S.pmask, tid);
        if (tid == 0) {
            u32 ns = 0;
            for (u32 j = 0; j < 256; j++) {
                S.present[j] = (S.F[j] != 0) || j == 0;
                if (S.present[j]) { S.idx_of[j] = (u8)ns; S.alpha[ns] = (u8)j; ns++; }
            }
            S.nsym = ns;
        }
        __syncthreads();
        const u32 ns = S.nsym;
        const bool f_in_lds = ns <= FRONT_LDS_NSYM;
        u32 *Fg = enc_pair_counters(ws, b);
        u32 *Fp = f_in_lds ? (u32 *)dyn : Fg;
        for (u32 j = tid; j < ns * ns; j += FRONT_THREADS) Fp[j] = 0;
        __syncthreads();
        wg_hist1(data, n, Fp, ns, 1u, S.idx_of, tid);
        // one context row per thread: totals, normalisation in double precision (:470-495), serialised length
        u32 T = 0;
        if (tid < ns) {
            u32 *F = Fp + tid * ns;
            for (u32 j = 0; j < ns; j++) T += F[j];
            if (T) {
                double p = (double)4096 / (double)(int)T;
                for (;;) {
                    int t2 = 0, m = 0, M = 0;
                    for (u32 j = 0; j < ns; j++) {
                        int f = (int)F[j];
                        if (!f) continue;
                        if (m < f) { m = f; M = (int)j; }
                        f = (int)((double)f * p);
                        if (f == 0) f = 1;
                        F[j] = (u32)f;
                        t2 += f;
                    }
                    t2++;
                    if (t2 < 4096) { F[M] += (u32)(4096 - t2); break; }
                    if (t2 - 4096 >= (int)F[M] / 2) { p = .98; continue; }
                    F[M] -= (u32)(t2 - 4096);
                    break;
                }
            }
            S.T[tid] = T;
        }
        __syncthreads();
        if (tid < ns) {
            u32 len = 0;
            if (T) {
                // the context byte with the table grammar's run-length shortcut over the contexts that occur (:497-510)
                u32 pos = 0;
                for (u32 r = tid; r > 0 && S.T[r - 1] && S.alpha[r - 1] + 1u == S.alpha[r]; r--) pos++;
                len = (pos == 0 ? 1u : pos == 1 ? 2u : 0u) + x8_put_table(nullptr, Fp + tid * ns, 1, S.alpha, ns);
            }
            S.rowlen[tid] = len;
        }
        __syncthreads();
        if (tid == 0) {
            u32 off = 0;
            for (u32 r = 0; r < ns; r++) { const u32 l = S.rowlen[r]; S.rowlen[r] = off; off += l; }
            tab[off] = 0;                                                        // closes the context list (:534)
            H.tab_len = off + 1;
        }
        __syncthreads();
        if (tid < ns && T) {
            u8 *cp = tab + S.rowlen[tid];
            u32 pos = 0;
            for (u32 r = tid; r > 0 && S.T[r - 1] && S.alpha[r - 1] + 1u == S.alpha[r]; r--) pos++;
            if (pos == 0) *cp++ = S.alpha[tid];
            else if (pos == 1) {
                u32 r = tid + 1;
                while (r < ns && S.T[r] && S.alpha[r] == S.alpha[tid] + (r - tid)) r++;
                *cp++ = S.alpha[tid];
                *cp++ = (u8)(r - (tid + 1));
            }
            x8_put_table(cp, Fp + tid * ns, 1, S.alpha, ns);
        }
        // image: byte -> compact index, u16 cumulative rows
        img[tid] = S.present[tid] ? S.idx_of[tid] : (u8)0;
        if (tid < ns) {
            u16 *cum = (u16 *)(img + ENC_IMG_IDX) + tid * (ns + 1);
            u32 x = 0;
            for (u32 j = 0; j < ns; j++) { cum[j] = (u16)x; x += Fp[tid * ns + j]; }
            cum[ns] = (u16)x;
        }
        H.ns = ns;
    }
    wg_fence();
    __syncthreads();
    if (tid == 0) {
        ws.desc[b].tab_len = H.tab_len;
        I0->data = (u64)data; I0->n = n; I0->image = (u64)img; I0->bits = 12; I0->order = H.order;
        I0->ns = H.ns; I0->img_bytes = 0;
        I0->scratch_end = (u64)scratch_end;
        I0->active = 1;
    }
}

// One quad per block.  The step schedule is chain_encode's (both codecs walk a block the same way); what differs
// is the renormalisation: zero, one or two BYTES per chain and step.
template <int ORDER>
__device__ __forceinline__ u32 chain_encode8(gcu8 *data, u32 n, gcu8 *image, u32 ns, gcu32 *rcptab, gu8 *scratch_end,
                                             u32 room, bool active, u32 lane)
{
    const u32 k = lane & 3;
    GAS const u16 *cum = (GAS const u16 *)(image + ENC_IMG_IDX);
    const u32 rs = ns + 1;
    u32 x = X8_LOW_E;
    u32 written = 0;                 // bytes emitted by the quad so far
    u32 nsteps, first, p;
    const u32 q = n >> 2;
    if (ORDER == 0) {
        const u32 gtop = n ? (n - 1) >> 2 : 0;
        nsteps = n ? gtop + 1 : 0;
        first = (4 * gtop + k < n) ? 0 : 1;
        p = 4 * (gtop - (first ? 1 : 0)) + k;
    } else {
        const u32 tail = n - 4 * q;
        nsteps = tail + q;
        first = (k == 3) ? 0 : tail;
        p = (k == 3) ? n - 1 : k * q + q - 1;
    }
    if (!active) { nsteps = 0; first = 0; }
    u32 cur = 0;
    if (nsteps > first) cur = image[data[p]];
    for (u32 s = 0; wave_any(s < nsteps); s++) {
        const bool live = s >= first && s < nsteps;
        bool e1 = false, e2 = false;
        u32 rcp = 0, pk = 0, nextc = 0;
        if (live) {
            u32 row = 0;
            if (ORDER == 0) { if (p >= 4) nextc = image[data[p - 4]]; }
            else if (s != nsteps - 1) { nextc = image[data[p - 1]]; row = nextc; }
            const u32 c0 = cum[row * rs + cur], c1 = cum[row * rs + cur + 1];
            const u32 f = c1 - c0;
            pk = c0 | (f << 16);
            rcp = enc_rcp(rcptab, f);
            const u32 x_max = f << 19;                                           // rANS_byte.h:217 (f = 4096: 2^31)
            e1 = x >= x_max;
            e2 = (x >> 8) >= x_max;
        }
        const u32 m1 = quad_ballot(e1, lane), m2 = quad_ballot(e2, lane);
        if (e1) {
            const u32 at = written + __popc(m1 >> (k + 1)) + __popc(m2 >> (k + 1));
            if (at + 2 <= room) {                                                // (never false for data the bound covers)
                scratch_end[-(long)(at + 1)] = (u8)x;
                if (e2) scratch_end[-(long)(at + 2)] = (u8)(x >> 8);
            }
            x >>= e2 ? 16 : 8;
        }
        written += __popc(m1) + __popc(m2);
        if (live) {
            x = enc_advance(x, rcp, pk, 12);
            cur = nextc;
            p -= (ORDER == 0) ? 4 : 1;
        }
    }
    // RansEncFlush x4 in the order 3,2,1,0 (:199-202): state 0 ends up lowest in memory
    if (active && written + 16 <= room) {
        gu8 *dst = scratch_end - written - 16 + 4 * k;
        dst[0] = (u8)x; dst[1] = (u8)(x >> 8); dst[2] = (u8)(x >> 16); dst[3] = (u8)(x >> 24);
    }
    return active ? written + 16 : 0;
}

__global__ __launch_bounds__(WAVE) void k8_enc_chain(EncItem *items, const u32 *rcptab_, int nitems, u32 room)
{
    const u32 lane = threadIdx.x;
    const int slot = (int)blockIdx.x * 16 + (int)(lane >> 2);
    const bool mine = slot < nitems;
    EncItem *I = &items[mine ? slot : 0];
    const bool active = mine && I->active;
    if (!wave_any(active)) return;
    gcu32 *rcptab = to_global(rcptab_);
    gcu8 *data = (gcu8 *)I->data, *im = (gcu8 *)I->image;
    gu8 *send = (gu8 *)I->scratch_end;
    const u32 n = I->n, ns = I->ns, order = active ? I->order : 2u;
    u32 pay = chain_encode8<1>(data, n, im, ns, rcptab, send, room, order == 1, lane);
    pay |= chain_encode8<0>(data, n, im, ns, rcptab, send, room, order == 0, lane);
    if (active && (lane & 3) == 0) I->pay_len = pay;
}

__global__ __launch_bounds__(FINISH_THREADS) void k8_enc_finish(BatchArgs a, EncWs ws, int base, u32 room)
{
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    const int i = base + (int)b;
    const EncItem *I0 = &ws.items[b];
    const i32 st = ws.desc[b].status;
    if (st != ST_OK || !I0->active || I0->pay_len > room) {
        if (tid == 0) { a.status[i] = st != ST_OK ? st : ST_CAPACITY; a.out_size[i] = 0; }
        return;
    }
    u8 *out = a.out + a.out_off[i];
    const u32 tab_len = ws.desc[b].tab_len, pay = I0->pay_len, total = 9 + tab_len + pay;
    if (tid == 0) {                                                              // :204-214, :593-605
        const u32 csz = total - 9, n = I0->n;
        out[0] = (u8)I0->order;
        out[1] = (u8)csz; out[2] = (u8)(csz >> 8); out[3] = (u8)(csz >> 16); out[4] = (u8)(csz >> 24);
        out[5] = (u8)n; out[6] = (u8)(n >> 8); out[7] = (u8)(n >> 16); out[8] = (u8)(n >> 24);
    }
    group_copy<FINISH_THREADS>(out + 9, ws.tab + (u64)b * TAB_BYTES, tab_len, tid);
    group_copy<FINISH_THREADS>(out + 9 + tab_len, (const u8 *)I0->scratch_end - pay, pay, tid);
    if (tid == 0) { a.status[i] = ST_OK; a.out_size[i] = total; }
}

extern "C" void r4x8_launch_encode(const BatchArgs *a, const EncWs *ws, int base, int nblk, hipStream_t s)
{
    if (r4x16_first_on_device(8u))
        (void)hipFuncSetAttribute((const void *)k8_enc_front, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_DYN_LDS);
    const u32 room = (u32)(ws->scratch_stride > 0xffffffffull ? 0xffffffffu : ws->scratch_stride);
    hipLaunchKernelGGL(k8_enc_front, dim3(nblk), dim3(FRONT_THREADS), FRONT_DYN_LDS, s, *a, *ws, base);
    hipLaunchKernelGGL(k8_enc_chain, dim3((nblk + 15) / 16), dim3(WAVE), 0, s, ws->items, ws->rcptab, nblk, room);
    hipLaunchKernelGGL(k8_enc_finish, dim3(nblk), dim3(FINISH_THREADS), 0, s, *a, *ws, base, room);
}
