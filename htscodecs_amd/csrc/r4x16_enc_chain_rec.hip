// =============================================================================================
// r4x16_enc_chain_rec.hip - k_enc_chain_rec: the encoder's hot loop on symbol records (r4x16_common.h, "kind 2"), the
// short-step route for batches that leave LDS to spare (rANS_static4x16pr.c:442-485, :794-839; rANS_word.h:281-321).
//
// Same walk, same phases and the same staging of the emitted words as chain_encode_o1_lds / chain_encode_o0_pipe
// (r4x16_enc_chain.h), but a symbol costs one 16-byte LDS read instead of a cumulative pair, its unpacking and a
// reciprocal look-up, and the state update takes its operands as they come: a lone wave issues one instruction of
// any kind every four cycles, so instructions per symbol are the step time.
// Software pipeline, one stage per trip of four steps:  input piece (HBM, 8 or 16 bytes, three double-trips ahead)
//   -> byte -> compact index (LDS)  ->  record (LDS)  ->  four state updates.
// =============================================================================================
#include <stdlib.h>
#include <type_traits>
#include "r4x16_dev.h"
#include "r4x16_sched.h"

struct RecOut {
    u8 *ring;            // LDS: 128-byte ring of emitted words (word j of the stream at byte 126 - 2 (j & 63)) + dump slot
    u32 ring126;
    gu8 *send;           // scratch_end of the stream: word j belongs at send - 2 (j + 1)
    gu8 *dump;
    u32 written, flushed;
    u32 k, lane, himask; // himask: the quad lanes above this one
    u64 amask;           // lanes that hold a stream
    bool active;
    u32x4 held;
    gu8 *held_dst;
    // RansEncPutSymbol (rANS_word.h:281-321) on a record {rcp_freq, x_max, bias, cmpl_freq | rcp_shift << 24}.
    // q = x / freq < 2^21 once x < x_max, so q * cmpl_freq is a 24-bit multiply (mod 2^32) that ignores the top byte.
    template <bool ALL>
    __device__ __forceinline__ void step(u32 &x, bool live, const u32x4 r)
    {
        const bool over = x >= r.y;
        const u64 m = __ballot(over) & (ALL ? amask : __ballot(live));
        const u32 sh = (u32)(m >> (lane & ~3u));              // the quad's four bits at the bottom
        const bool emit = ALL ? (over && active) : (live && over);
        const u32 j = written + __popc(sh & himask);
        const u32 j63 = emit ? (j & 63u) : ~0u;               // -1: the dump slot at ring + 128
        *(LAS u16 *)(unsigned long)(ring126 - 2u * j63) = (u16)x;
        const u32 xs = emit ? x >> 16 : x;
        written += __popc(sh & 15u);
        const u32 q = __umulhi(xs, r.x) >> (r.w >> 24);
        const u32 xn = __umul24(q, r.w) + (xs + r.z);
        x = (ALL || live) ? xn : xs;
    }
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
    // unconditional form for the main loop (see EncOut::flush_pipelined): store what was read out last time, read out the next
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
        if (active) *(gu32 *)(send - 2ull * written - 16 + 4 * k) = x;      // RansEncFlush x4 (:482-485)
        return active ? 2 * written + 16 : 0;
    }
};

// ORDER 1: chain k codes quarter k backwards, byte p in the context of byte p - 1, the quarter's first byte in context 0
// (:794-839).  ORDER 0: step s codes group gtop - s, chain k its byte 4 g + k (:442-459).
// AFF: compact index = byte - c for every byte of the data (EncItem.affine = c + 1): no idx_of[] look-up per symbol.
// The pipelined look-ups also run for trips past a stream's end and for lanes without a stream; so that their record
// addresses stay inside the image whatever bytes they see, trips past the end re-read the stream's own first bytes
// (members of the alphabet) and lanes without a pipelined stream mask the index to 0.
template <int ORDER, bool AFF>
__device__ __forceinline__ u32 chain_encode_rec(const u8 *img_lds, u8 *ring, gcu8 *data, u32 n, u32 ns, u32 aff, gcu8 *safe,
                                                gu8 *scratch_end, gu8 *dump, bool active, u32 lane)
{
    const u32 k = lane & 3;
    const u32 idxa = (u32)(unsigned long)(LAS const u8 *)img_lds;
    const u32 reca = idxa + ENC_IMG_IDX;
    const u32 rowb = 16u * ns;
    const u32 coff = aff - 1u;
    auto idxof = [&](u32 byte) -> u32 {
        if (AFF) return (byte - coff) & 0xffu;
        return *(LAS const u8 *)(unsigned long)(idxa + byte);
    };
    u32 pmask = 0;                                            // AFF: 0xff on lanes whose stream takes the pipelined loop
    auto idxof_pipe = [&](u32 byte) -> u32 {
        if (AFF) return (byte - coff) & pmask;
        return *(LAS const u8 *)(unsigned long)(idxa + byte);
    };
    // record of symbol si in context ci (order 0: ci = 0)
    auto rec = [&](u32 ci, u32 si) -> u32x4 {
        const u32 a = (ORDER == 1 ? reca + __umul24(ci, rowb) : reca) + 16u * si;
        return *(LAS const u32x4 *)(unsigned long)a;
    };
    RecOut o{ring, (u32)(unsigned long)(LAS u8 *)ring + 126u, scratch_end, dump, 0u, 0u, k, lane, (0xeu << k) & 0xeu, __ballot(active), active,
             {0, 0, 0, 0}, dump};
    u32 x = RANS_LOW;
    const u32x4 none = {0u, ~0u, 0u, 0u};

    if (ORDER == 1) {
        const u32 q = active ? n >> 2 : 0;
        const u32 tail = active ? n - 4 * q : 0;
        // (A) tail bytes n-1 .. 4q on chain 3, context = previous byte (:806-811)
        u32 cur = 0;
        if (active && k == 3 && tail) cur = idxof(data[n - 1]);
        for (u32 s = 0; wave_any(s < tail); s++) {
            const bool live = k == 3 && s < tail;
            u32x4 r = none;
            if (live) { const u32 ci = idxof(data[n - 2 - s]); r = rec(ci, cur); cur = ci; }
            o.step<false>(x, live, r);
        }
        // (B) backward walk over offsets q-1 .. 1 of each quarter (:813-829)
        gcu8 *qbase = data + (u64)k * q;
        const u32 r0 = q ? q - 1 : 0;
        const u32 main = r0;
        const u32 npair = main >> 3;
        const u32 ntrip = 2 * npair;
        cur = (active && q) ? idxof(qbase[r0]) : 0u;
        if (wave_any(npair > 0)) {
            pmask = npair ? 0xffu : 0u;
            gcu8 *past = npair ? qbase : safe;       // what trips past the end read (the stream's own first eight bytes)
            auto load8 = [&](u32 j) -> u32x2 {       // bytes r0-8j-8 .. r0-8j-1: .y = contexts of trip 2j, .x = of trip 2j+1
                gcu8 *p = j < npair ? qbase + (r0 - 8 * j) - 8 : past;
                return *(GAS const u32x2_unaligned *)p;
            };
            struct I4 { u32 c0, c1, c2, c3; };
            struct R4 { u32x4 a, b, c, d; };
            auto idx4 = [&](u32 ww) -> I4 {
                I4 r = {idxof_pipe(ww >> 24), idxof_pipe((ww >> 16) & 0xff), idxof_pipe((ww >> 8) & 0xff), idxof_pipe(ww & 0xff)};
                return r;
            };
            auto rec4 = [&](const I4 &c, u32 sym) -> R4 {     // symbol `sym` in context c0, c0 in c1, c1 in c2, c2 in c3
                R4 r = {rec(c.c0, sym), rec(c.c1, c.c0), rec(c.c2, c.c1), rec(c.c3, c.c2)};
                return r;
            };
            // Two sets of pipeline registers used in turn (trip t reads set t & 1 and fills the other): a rotation
            // `P0 = Pn; I1 = In` at the end of a trip costs twenty register moves per four symbols.
            I4 IA, IB;                               // contexts of the trip after the one in hand
            R4 PA, PB;                               // records of the trip in hand
            u32 curA;                                // symbol coded first in the next trip = last context of this one
            u32x2 Q0, Q1, Q2, Q3;                    // input pieces, piece j in Q[j % 4]
            {
                Q0 = load8(0); Q1 = load8(1); Q2 = load8(2); Q3 = load8(3);
                const I4 i0 = idx4(Q0.y);
                IA = idx4(Q0.x);
                PA = rec4(i0, cur);
                curA = i0.c3;
            }
            u32 t = 0;
            // (the all-live / some-dead decision is taken once per eight trips, OUTSIDE the trips: a branch between a trip's
            //  look-ups and its steps would end the basic block there, and the compiler's wait-count pass then waits for
            //  everything in flight - the look-ups just issued - at the top of the steps: 240 cycles per trip, measured)
            auto trip = [&](auto allc, u32 wnext2, const I4 &Icur, I4 &Inext, const R4 &Pcur, R4 &Pnext) {
                constexpr bool ALL = decltype(allc)::value;
                const bool live = ALL ? true : t < ntrip;
                Inext = idx4(wnext2);                    // contexts of trip t+2
                Pnext = rec4(Icur, curA);                // records of trip t+1
                __builtin_amdgcn_sched_barrier(0);       // (the look-ups above belong to later trips: see chain_encode_o1_lds)
                o.step<ALL>(x, live, Pcur.a); o.step<ALL>(x, live, Pcur.b); o.step<ALL>(x, live, Pcur.c); o.step<ALL>(x, live, Pcur.d);
                if (live) cur = curA;
                curA = Icur.c3;
                t++;
                __builtin_amdgcn_sched_barrier(0);
            };
            // double-trip d: trip 2d looks up the contexts of trip 2d+2 (piece d+1, high dword), trip 2d+1 those of trip
            // 2d+3 (piece d+1, low dword); piece d+4 is requested into the slot of piece d
            auto eight = [&](auto allc, u32 d) {
                Q0 = load8(d + 4); o.flush_pipelined(); trip(allc, Q1.y, IA, IB, PA, PB); trip(allc, Q1.x, IB, IA, PB, PA);
                Q1 = load8(d + 5); o.flush_pipelined(); trip(allc, Q2.y, IA, IB, PA, PB); trip(allc, Q2.x, IB, IA, PB, PA);
                Q2 = load8(d + 6); o.flush_pipelined(); trip(allc, Q3.y, IA, IB, PA, PB); trip(allc, Q3.x, IB, IA, PB, PA);
                Q3 = load8(d + 7); o.flush_pipelined(); trip(allc, Q0.y, IA, IB, PA, PB); trip(allc, Q0.x, IB, IA, PB, PA);
            };
            for (u32 d = 0; wave_any(d < npair); d += 4) {
                if (!wave_any(active && t + 8 > ntrip)) eight(std::true_type{}, d);
                else eight(std::false_type{}, d);
            }
            o.flush_drain();
            o.flush();
        }
        // (B') remaining walk steps, one at a time
        u32 r = r0 - 4 * ntrip, done = 4 * ntrip;
        for (; wave_any(done < main); ) {
            const bool live = done < main;
            u32x4 rr = none;
            if (live) { const u32 ci = idxof(qbase[r - 1]); rr = rec(ci, cur); cur = ci; r--; done++; }
            o.step<false>(x, live, rr);
            o.flush();
        }
        // (C) first byte of each quarter in context 0 (:831-834)
        {
            const bool live = active && q > 0;
            u32x4 rr = none;
            if (live) rr = rec(0, cur);
            o.step<false>(x, live, rr);
        }
        return o.finish(x);
    } else {
        const u32 Q = active ? n >> 2 : 0;                    // whole groups
        const u32 rem = active ? n & 3u : 0;                  // bytes of the partial top group
        // (A) the partial top group: chains k < rem code byte 4Q + k
        if (wave_any(rem != 0)) {
            const bool live = k < rem;
            u32x4 rr = none;
            if (live) rr = rec(0, idxof(data[4 * Q + k]));
            o.step<false>(x, live, rr);
        }
        // (B) whole groups Q-1 .. 0; trip t covers groups Q-1-4t .. Q-4-4t = bytes [4 (Q-4t-4), 4 (Q-4t))
        const u32 npair = Q >> 3;
        const u32 ntrip = 2 * npair;
        if (wave_any(npair > 0)) {
            pmask = npair ? 0xffu : 0u;
            gcu8 *past = npair ? data : safe;                 // (the stream's own first 32 bytes)
            struct DT { u32x4 a, b; };                        // the pieces of trips 2j and 2j+1
            auto load_dt = [&](u32 j) -> DT {
                gcu8 *p = j < npair ? data + 4ull * (Q - 8 * j) - 32 : past;
                DT r = {*(GAS const u32x4_unaligned *)(p + 16), *(GAS const u32x4_unaligned *)p};
                return r;
            };
            struct I4 { u32 c0, c1, c2, c3; };
            struct R4 { u32x4 a, b, c, d; };
            const u32 sh = 8 * k;
            auto idx4 = [&](u32x4 v) -> I4 {                  // steps run from the highest group (v.w) down
                I4 r = {idxof_pipe((v.w >> sh) & 0xff), idxof_pipe((v.z >> sh) & 0xff), idxof_pipe((v.y >> sh) & 0xff), idxof_pipe((v.x >> sh) & 0xff)};
                return r;
            };
            auto rec4 = [&](const I4 &c) -> R4 {
                R4 r = {rec(0, c.c0), rec(0, c.c1), rec(0, c.c2), rec(0, c.c3)};
                return r;
            };
            I4 IA, IB;
            R4 PA, PB;
            DT Q0, Q1, Q2, Q3;
            {
                Q0 = load_dt(0); Q1 = load_dt(1); Q2 = load_dt(2); Q3 = load_dt(3);
                PA = rec4(idx4(Q0.a));
                IA = idx4(Q0.b);
            }
            u32 t = 0;
            auto trip = [&](auto allc, u32x4 wnext2, const I4 &Icur, I4 &Inext, const R4 &Pcur, R4 &Pnext) {
                constexpr bool ALL = decltype(allc)::value;
                const bool live = ALL ? true : t < ntrip;
                Inext = idx4(wnext2);                         // bytes of trip t+2
                Pnext = rec4(Icur);                           // records of trip t+1
                __builtin_amdgcn_sched_barrier(0);
                o.step<ALL>(x, live, Pcur.a); o.step<ALL>(x, live, Pcur.b); o.step<ALL>(x, live, Pcur.c); o.step<ALL>(x, live, Pcur.d);
                t++;
                __builtin_amdgcn_sched_barrier(0);
            };
            auto eight = [&](auto allc, u32 d) {
                Q0 = load_dt(d + 4); o.flush_pipelined(); trip(allc, Q1.a, IA, IB, PA, PB); trip(allc, Q1.b, IB, IA, PB, PA);
                Q1 = load_dt(d + 5); o.flush_pipelined(); trip(allc, Q2.a, IA, IB, PA, PB); trip(allc, Q2.b, IB, IA, PB, PA);
                Q2 = load_dt(d + 6); o.flush_pipelined(); trip(allc, Q3.a, IA, IB, PA, PB); trip(allc, Q3.b, IB, IA, PB, PA);
                Q3 = load_dt(d + 7); o.flush_pipelined(); trip(allc, Q0.a, IA, IB, PA, PB); trip(allc, Q0.b, IB, IA, PB, PA);
            };
            for (u32 d = 0; wave_any(d < npair); d += 4) {
                if (!wave_any(active && t + 8 > ntrip)) eight(std::true_type{}, d);
                else eight(std::false_type{}, d);
            }
            o.flush_drain();
            o.flush();
        }
        // (B') the groups below the pipelined trips, one step each
        for (u32 g = Q - 4 * ntrip; wave_any(g > 0); ) {
            const bool live = g > 0;
            u32x4 rr = none;
            if (live) { g--; rr = rec(0, idxof(data[4 * g + k])); }
            o.step<false>(x, live, rr);
            o.flush();
        }
        return o.finish(x);
    }
}

// ---------------------------------------------------------------------------------------------
// k_enc_chain_rec: one wave per workgroup, qpw streams per wave (one per quad), persistent, one launch per LDS size
// class (see k_dec_chain).  A stream owns lds_per_item bytes: image (idx_of + records), then the word ring.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_enc_chain_rec(EncItem *items, const u32 *safe_, u8 *dump_, const u32 *list, u32 *count,
                                                        int qpw, int spw_unused, u32 lds_per_item, int dyn)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 lane = threadIdx.x;
    const u32 quad = lane >> 2;
    const int nmine = (int)count[SCHED_COUNT];
    list += count[SCHED_START];
    SchedWalk walk(count, nmine, qpw, dyn != 0);          // (r4x16_sched.h)
    for (int wg = walk.next_wave(); wg >= 0; wg = walk.next_wave()) {
        const int slot = wg * qpw + (int)quad;
        const bool mine = quad < (u32)qpw && slot < nmine;
        EncItem *I = &items[mine ? list[slot] : list[wg * qpw]];
        const bool active = mine && I->active;
        if (!wave_any(active)) continue;
        sched_setprio(sched_prio_of(active, active ? I->n : 0u));
        const u32 img_bytes = active ? I->img_bytes : 0u;
        const u32 order = active ? I->order : 2u;
        gcu8 *data = (gcu8 *)I->data;
        gu8 *send = (gu8 *)I->scratch_end;
        const u32 n = I->n, ns = I->ns;
        // the whole wave copies each quad's image in turn (16-byte pieces)
        const u64 my_img = active ? I->image : 0ull;
        for (int qd = 0; qd < qpw; qd++) {
            const u64 src = __shfl(my_img, qd * 4);
            const u32 nb = __shfl(img_bytes, qd * 4);
            if (!src) continue;
            gcu32x4 *s = (gcu32x4 *)src;
            u32x4 *dd = (u32x4 *)(lds + (u64)qd * lds_per_item);
            for (u32 j = lane; j < ((nb + 15) >> 4); j += WAVE) dd[j] = s[j];
        }
        __syncthreads();
        // lanes without a stream read the image of stream 0 and never emit (RecOut::amask)
        const u32 sl = active ? quad : 0u;
        const u8 *im = lds + (u64)sl * lds_per_item;
        u8 *ring = lds + (u64)sl * lds_per_item + (lds_per_item - ENC_RING_BYTES);
        gu8 *dump = to_global(dump_) + 16u * ((blockIdx.x * blockDim.x + lane) & (ENC_DUMP_BYTES / 16u - 1u));
        const u32 aff = active ? I->affine : 1u;
        u32 pay;
        if (!wave_any(aff == 0u)) {                           // every stream of the wave has an affine alphabet
            pay = chain_encode_rec<1, true>(im, ring, data, n, ns, aff, (gcu8 *)safe_, send, dump, order == 1, lane);
            pay |= chain_encode_rec<0, true>(im, ring, data, n, ns, aff, (gcu8 *)safe_, send, dump, order == 0, lane);
        } else {
            pay = chain_encode_rec<1, false>(im, ring, data, n, ns, 1u, (gcu8 *)safe_, send, dump, order == 1, lane);
            pay |= chain_encode_rec<0, false>(im, ring, data, n, ns, 1u, (gcu8 *)safe_, send, dump, order == 0, lane);
        }
        if (active && (lane & 3) == 0) I->pay_len = pay;
        __syncthreads();                                      // LDS is reused by the next share
    }
    walk.leave();
}

extern "C" void r4x16_enc_chain_rec_lds_limit(int bytes)
{
    (void)hipFuncSetAttribute((const void *)k_enc_chain_rec, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
extern "C" const void *r4x16_enc_chain_rec_kernel(void) { return (const void *)k_enc_chain_rec; }
