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
#include "r4x16_sched.h"
#include "r4x16_enc_step.h"

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
    u32 wtmp[12];        // wg_rle_split: per-wave partial results of its scans
    u32 sel4[16];        // wg_rle_split: v_perm selectors that move the bytes of a 4-bit mask to the bottom of a dword
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

// One symbol record of the encoder's short-step images (r4x16_common.h, "kind 2"; RansEncSymbolInit, rANS_word.h:190-266).
// rcptab[f] = ceil(2^(31 + ceil(log2 f)) / f) for f >= 2, the reference's rcp_freq (:252).
__device__ __forceinline__ u32x4 enc_record(u32 start, u32 f, u32 bits, const u32 *rcptab)
{
    if (f == 0) return u32x4{0u, ~0u, 0u, 0u};                       // never coded in this context
    if (f == 1) return u32x4{~0u, 1u << (31u - bits), start + (1u << bits) - 1u, (1u << bits) - 1u};
    const u32 rsh = 31u - (u32)__clz((int)(f - 1u));                   // ceil(log2 f) - 1
    return u32x4{rcptab[f], f << (31u - bits), start, ((1u << bits) - f) | (rsh << 24)};
}

__device__ void enc_o0_tables(u32 n, u8 *tab, u8 *image, EncShared &S, u32 lane, const u32 *rec_rcptab = nullptr)
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
    if (rec_rcptab) {                                       // symbol records instead of the cumulative row
        u32x4 *rec = (u32x4 *)(image + ENC_IMG_IDX);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            rec[lane * 4 + c] = enc_record(start, f[c], O0_BITS, rec_rcptab);
            start += f[c];
        }
        wsync();
        return;
    }
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
// 16-bit masks over a 16-byte piece (bit c = byte c), computed on the four dwords at once.
// zero bytes of x -> bits 0..3
__device__ __forceinline__ u32 zero_bytes4(u32 x)
{
    const u32 nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bit 7 of every non-zero byte
    const u32 t = (nz ^ 0x80808080u) >> 7;                                  // bits 0, 8, 16, 24 for the zero bytes
    const u32 u = t | (t >> 7);
    return (u | (u >> 14)) & 15u;
}
// bytes equal to the byte before them (`before` for byte 0; 256 and more: byte 0 has no predecessor)
__device__ __forceinline__ u32 eq_prev_mask16(u32x4 v, u32 before)
{
    const u32 s0 = (v.x << 8) | (before & 0xffu), s1 = __builtin_amdgcn_alignbyte(v.y, v.x, 3),
              s2 = __builtin_amdgcn_alignbyte(v.z, v.y, 3), s3 = __builtin_amdgcn_alignbyte(v.w, v.z, 3);
    const u32 m = zero_bytes4(v.x ^ s0) | (zero_bytes4(v.y ^ s1) << 4) | (zero_bytes4(v.z ^ s2) << 8) | (zero_bytes4(v.w ^ s3) << 12);
    return before > 255u ? m & ~1u : m;
}
template <bool REP>
__device__ __forceinline__ void wg_hist8_t(const u8 *data, u32 n, u32 *Fout, u32 *Rout, u32 *priv, u32 tid)
{
    // (copy stride 257 dwords: each copy starts one LDS bank further; with 256 all sixteen sat on the same banks)
    for (u32 j = tid; j < (REP ? 32 : 16) * 257; j += FRONT_THREADS) priv[j] = 0;
    __syncthreads();
    u32 *F = priv + 257 * (tid & 15);
    u32 *R = F + 16 * 257;                               // REP: bytes that repeat the byte before them (rle.c:60-72)
    // 16-byte pieces, four in flight per thread (each is requested three pieces before it is counted: the
    // passes are memory-latency bound otherwise, eight waves per CU cannot hide an HBM round trip per piece)
    const u32 full = n >> 4;
    struct Pc { u32x4 v; u32 before; };
    auto ld = [&](u32 pi) -> Pc {
        Pc r = {{0, 0, 0, 0}, 256u};
        if (pi < full) {
            r.v = *(GAS const u32x4_unaligned *)(to_global(data) + 16ull * pi);
            if (REP && pi) r.before = to_global(data)[16ull * pi - 1];
        }
        return r;
    };
    auto count = [&](const Pc &q, u32 pi) {
        if (pi >= full) return;
        const u32 ww[4] = {q.v.x, q.v.y, q.v.z, q.v.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            atomicAdd(&F[ww[c] & 0xff], 1u);
            atomicAdd(&F[(ww[c] >> 8) & 0xff], 1u);
            atomicAdd(&F[(ww[c] >> 16) & 0xff], 1u);
            atomicAdd(&F[ww[c] >> 24], 1u);
        }
        if (REP) {
            const u32 eq = eq_prev_mask16(q.v, q.before);
#pragma unroll
            for (int c = 0; c < 16; c++)
                if ((eq >> c) & 1u) atomicAdd(&R[(ww[c >> 2] >> (8 * (c & 3))) & 0xff], 1u);
        }
    };
    const u32 T = FRONT_THREADS;
    Pc q0 = ld(tid), q1 = ld(tid + T), q2 = ld(tid + 2 * T), q3 = ld(tid + 3 * T);
    for (u32 pi = tid; pi < full; pi += 4 * T) {
        count(q0, pi);         q0 = ld(pi + 4 * T);
        count(q1, pi + T);     q1 = ld(pi + 5 * T);
        count(q2, pi + 2 * T); q2 = ld(pi + 6 * T);
        count(q3, pi + 3 * T); q3 = ld(pi + 7 * T);
    }
    const u32 done = full * 16;
    if (done + tid < n) {
        const u32 b = data[done + tid];
        atomicAdd(&F[b], 1u);
        if (REP && done + tid && data[done + tid - 1] == b) atomicAdd(&R[b], 1u);
    }
    __syncthreads();
    {
        u32 t = 0, r = 0;
#pragma unroll
        for (int c = 0; c < 16; c++) { t += priv[257 * c + tid]; if (REP) r += priv[257 * (16 + c) + tid]; }
        Fout[tid] = t;                                   // FRONT_THREADS == 256
        if (REP) Rout[tid] = r;
    }
    __syncthreads();
}
__device__ __forceinline__ void wg_hist8(const u8 *data, u32 n, u32 *Fout, u32 *priv, u32 tid)
{
    wg_hist8_t<false>(data, n, Fout, nullptr, priv, tid);
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

// Fp0 points at `copies` x ns*ns zeroed counters (LDS, or global with copies == 1).  The copies are interleaved -
// counter i of copy c is dword copies * i + c - and a thread counts in copy tid % copies: the lanes of a wave that use
// one copy then share a quarter (half) of the LDS banks among themselves only, and sixteen random addresses over
// sixteen banks collide less than sixty-four over sixty-four (an LDS atomic costs two cycles per lane on its worst
// bank or address, four at least per wave instruction: tools/micro/lds_atomic_rate.hip).  The copies are summed into
// compact form at the end.
template <class FP>
__device__ __forceinline__ void wg_hist1(const u8 *data, u32 n, FP Fp0, u32 ns, u32 copies, const u8 *idx_of, u32 tid)
{
    const u32 csh = copies == 4 ? 2u : copies == 2 ? 1u : 0u;
    FP Fp = Fp0 + (tid & (copies - 1));
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
            atomicAdd(&Fp[(__umul24(prev, ns) + ci[c]) << csh], 1u);       // (24-bit multiply: full rate, see wg_hist1_range)
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
            atomicAdd(&Fp[((i ? prev : 0u) * ns + cur) << csh], 1u);
            prev = cur;
        }
    }
    __syncthreads();
    // the three quarter starts are coded in context 0 (rANS_static4x16pr.c:720-723)
    if (tid >= 1 && tid < 4) atomicAdd(&Fp0[(u32)idx_of[data[tid * (n >> 2)]] << csh], 1u);
    __syncthreads();
    if (copies > 1) {
        // sum the copies (adjacent dwords), then write the sums in compact form: 4 copies only exist up to 48 x 48
        // counters, 2 up to 67 x 67 - at most 18 per thread
        u32 sum[18];
#pragma unroll
        for (u32 r = 0; r < 18; r++) {
            const u32 j = tid + r * FRONT_THREADS;
            u32 t = 0;
            if (j < ns * ns) for (u32 c = 0; c < copies; c++) t += Fp0[(j << csh) + c];
            sum[r] = t;
        }
        __syncthreads();
#pragma unroll
        for (u32 r = 0; r < 18; r++) {
            const u32 j = tid + r * FRONT_THREADS;
            if (j < ns * ns) Fp0[j] = sum[r];
        }
        __syncthreads();
    }
}

// wg_hist1 for alphabets that sit in a narrow range of byte values [lo, lo + R) - quality values do: the counters
// are indexed by (byte - lo) while counting, so that the sixteen byte -> compact index look-ups per 16-byte piece drop
// out of the LDS queue, which is what bounds this pass (32 LDS instructions per piece, 4+ cycles each: 297 -> 193 us
// per MiB and workgroup on 46-symbol quality data).  `nsa` = R, plus one "any other byte" index on the provisional
// route (values outside the range clamp to it).  Afterwards the sums of the copies go to the compact nsx x nsx layout
// every other pass uses; the four pairs that are coded in context 0 (the first byte and the three quarter starts,
// rANS_static4x16pr.c:720-723) are added there.  Returns true - provisional route only - if a byte was counted that is
// not in the alphabet (outside the range, or in a gap of it).  LDS counters only; nsa * nsa <= 18 * FRONT_THREADS.
// a * b + c for a, b < 2^24 in one full-rate instruction (the compiler splits the sum three ways - multiply, shift, v_add3 -
// when it is written in C)
__device__ __forceinline__ u32 mad24(u32 a, u32 b, u32 c) { u32 r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u32 lds_addr_of(const void *p) { return (u32)(unsigned long)(LAS const u8 *)p; }
__device__ __forceinline__ void lds_inc(u32 byte_addr)
{
    __hip_atomic_fetch_add((LAS u32 *)(unsigned long)byte_addr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool wg_hist1_range(const u8 *data, u32 n, u32 *Fp0, u32 nsx, u32 copies, const EncShared &S, u32 lo, u32 nsa,
                                               bool prov, u32 tid)
{
    const u32 csh = copies == 4 ? 2u : copies == 2 ? 1u : 0u;
    u32 *Fp = Fp0 + (tid & (copies - 1));
    const u32 top = nsa - 1, rsa = nsa | 1u;          // (odd row stride: an even one halves the banks a row pair can reach)
    const u32 cs = csh + 2u, rowb = rsa << cs, fbase = lds_addr_of(Fp);
    const u32 full = n >> 4;
    struct Piece { u32x4 w; u32 before; };
    auto ld = [&](u32 pi) -> Piece {
        Piece p = {{0, 0, 0, 0}, 0};
        if (pi < full) { p.w = *(GAS const u32x4_unaligned *)(to_global(data) + 16ull * pi); p.before = pi ? to_global(data)[16ull * pi - 1] : 0u; }
        return p;
    };
    auto ix = [&](u32 b) -> u32 { const u32 d = b - lo; return d < top ? d : top; };      // (bytes below lo wrap around and clamp too)
    auto count = [&](const Piece &p, u32 pi) {
        if (pi >= full) return;
        const u32 ww[4] = {p.w.x, p.w.y, p.w.z, p.w.w};
        u32 ci[16];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            ci[4 * c] = ix(ww[c] & 0xff); ci[4 * c + 1] = ix((ww[c] >> 8) & 0xff);
            ci[4 * c + 2] = ix((ww[c] >> 16) & 0xff); ci[4 * c + 3] = ix(ww[c] >> 24);
        }
        // counter (prev, cur) of this thread's copy, as an LDS byte address: one 24-bit multiply-add for the row, one
        // shift-add for the column (a plain `prev * rsa` is a 32-bit multiply, quarter rate: it was half of the loop's
        // vector cycles - 472 instructions per 64 bytes, 64 of them v_mul_lo_u32)
        u32 row = mad24(ix(p.before), rowb, fbase);                             // the row of the byte before, as an address
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const u32 at = (ci[c] << cs) + row;
            if (c || pi) lds_inc(at);                                           // (the block's first byte: context 0, below)
            row = mad24(ci[c], rowb, fbase);
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
        u32 prev = full ? ix(data[16 * full - 1]) : 0u;
        for (u32 i = 16 * full; i < n; i++) {
            const u32 cur = ix(data[i]);
            if (i) atomicAdd(&Fp[(prev * rsa + cur) << csh], 1u);
            prev = cur;
        }
    }
    __syncthreads();
    // sums of the copies, each with its place in the compact layout
    const u32 NONE = 0xffffffffu, R = prov ? top : nsa;
    u32 sum[18], dst[18];
    bool hit = false;
#pragma unroll
    for (u32 r = 0; r < 18; r++) {
        const u32 j = tid + r * FRONT_THREADS;
        sum[r] = 0; dst[r] = NONE;
        if (j < nsa * nsa) {
            const u32 ra = j / nsa, ca = j - ra * nsa;
            u32 t = 0;
            for (u32 c = 0; c < copies; c++) t += Fp0[((ra * rsa + ca) << csh) + c];
            const bool ok = ra < R && ca < R && S.present[(lo + ra) & 0xffu] && S.present[(lo + ca) & 0xffu];
            if (ok) { sum[r] = t; dst[r] = (u32)S.idx_of[(lo + ra) & 0xffu] * nsx + S.idx_of[(lo + ca) & 0xffu]; }
            else if (t) hit = true;
        }
    }
    __syncthreads();
    for (u32 j = tid; j < nsx * nsx; j += FRONT_THREADS) Fp0[j] = 0;
    __syncthreads();
#pragma unroll
    for (u32 r = 0; r < 18; r++) if (dst[r] != NONE) Fp0[dst[r]] = sum[r];
    __syncthreads();
    if (tid < 4) atomicAdd(&Fp0[S.idx_of[data[tid * (n >> 2)]]], 1u);             // row 0: the first byte, the quarter starts
    return __syncthreads_or((int)hit) != 0;
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
    // Four pieces per thread and trip, and a trip's memory operations leave together at its top, behind an explicit wait:
    // the packed results of the trip before, then the requests for the trip after - and only then the trip's own
    // arithmetic, on pieces that arrived during the last one.  (Loads and stores share one counter and the compiler
    // takes them to complete in any order: with a request and a store per piece, each piece's first use waited for
    // everything in flight - the acknowledgement of the store just issued included: 271 of k_enc_front's 752 us per
    // 1 MiB q4 block.)
    const u32 T = FRONT_THREADS;
    auto pack16 = [&](const u32x4 v) -> u64 {             // 16 symbols of `width` bits, first in the low bits
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        u64 acc = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const u32 i0 = S.idx_of[w[c] & 0xff], i1 = S.idx_of[(w[c] >> 8) & 0xff],
                      i2 = S.idx_of[(w[c] >> 16) & 0xff], i3 = S.idx_of[w[c] >> 24];
            const u64 four = (u64)(i0 | (i1 << width) | (i2 << (2 * width)) | (i3 << (3 * width)));
            acc |= four << (4 * width * c);
            seen |= i0 | i1 | i2 | i3;
        }
        return acc;
    };
    auto put16 = [&](u32 pi, u64 acc) {
        if (pi >= pieces) return;
        gu8 *o = gout + (u64)pi * (16 / per);
        if (per == 2)      *(GAS u64_unaligned *)o = acc;
        else if (per == 4) *(GAS u32_unaligned *)o = (u32)acc;
        else               *(GAS u16_unaligned *)o = (u16)acc;
    };
    u32x4 q0 = piece(tid), q1 = piece(tid + T), q2 = piece(tid + 2 * T), q3 = piece(tid + 3 * T);
    u64 r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    u32 rpi = pieces;                                     // the trip whose results wait in r0..r3 (pieces: none)
    for (u32 pi = tid; pi < pieces; pi += 4 * T) {
        __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0): q0..q3 are here, the last trip's stores are out
        __builtin_amdgcn_sched_barrier(0);
        if (rpi < pieces) { put16(rpi, r0); put16(rpi + T, r1); put16(rpi + 2 * T, r2); put16(rpi + 3 * T, r3); }
        const u32x4 n0 = piece(pi + 4 * T), n1 = piece(pi + 5 * T), n2 = piece(pi + 6 * T), n3 = piece(pi + 7 * T);
        __builtin_amdgcn_sched_barrier(0);
        r0 = pack16(q0); r1 = pack16(q1); r2 = pack16(q2); r3 = pack16(q3);
        rpi = pi;
        q0 = n0; q1 = n1; q2 = n2; q3 = n3;
    }
    if (rpi < pieces) { put16(rpi, r0); put16(rpi + T, r1); put16(rpi + 2 * T, r2); put16(rpi + 3 * T, r3); }
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
        if (v < 128u) { put(v); return; }
        u32 groups = 1;
        for (u32 t = v >> 7; t; t >>= 7) groups++;
        for (u32 g = groups; g-- > 0; ) put(((v >> (7 * g)) & 0x7f) | (g ? 0x80u : 0u));
    }
};

// The same stream fed up to four bytes at a time (wg_rle_split's literals: the bytes a 4-bit mask picks from a dword, already
// moved to the bottom of `bytes`): twenty bytes in five registers, the newest in the top byte of a4; every move is a
// v_perm_b32 of two neighbours with a computed selector - no loop over the bytes, the same instructions whatever the mask.
struct PieceOut {
    gu8 *p;
    u32 a0, a1, a2, a3, a4;
    u32 cnt;             // bytes not yet stored (< 16 between calls)
    __device__ __forceinline__ void append(u32 bytes, u32 c)          // c <= 4
    {
        const u32 sel = 0x03020100u + c * 0x01010101u;                // bytes c .. c + 3 of a register pair
        a0 = __builtin_amdgcn_perm(a1, a0, sel);
        a1 = __builtin_amdgcn_perm(a2, a1, sel);
        a2 = __builtin_amdgcn_perm(a3, a2, sel);
        a3 = __builtin_amdgcn_perm(a4, a3, sel);
        a4 = __builtin_amdgcn_perm(bytes, a4, sel);
        cnt += c;
        if (cnt >= 16u) {                                             // the oldest sixteen start at byte 20 - cnt (1 .. 4)
            const u32 so = 0x03020100u + (20u - cnt) * 0x01010101u;
            const u32x4 o = {__builtin_amdgcn_perm(a1, a0, so), __builtin_amdgcn_perm(a2, a1, so),
                             __builtin_amdgcn_perm(a3, a2, so), __builtin_amdgcn_perm(a4, a3, so)};
            *(GAS u32x4_unaligned *)p = o;
            p += 16; cnt -= 16u;
        }
    }
    __device__ __forceinline__ void flush()
    {
        const u32 w[5] = {a0, a1, a2, a3, a4};
        for (u32 k = 0; k < cnt; k++) { const u32 at = 20u - cnt + k; p[k] = (u8)(w[at >> 2] >> (8u * (at & 3u))); }
        p += cnt; cnt = 0;
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

// bytes whose flag in an LDS table of 256 bytes is set
__device__ __forceinline__ u32 flag_mask16(u32x4 v, const u8 *flags)
{
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    u32 m = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) m |= (u32)(flags[(w[c >> 2] >> (8 * (c & 3))) & 0xffu] != 0) << c;
    return m;
}
__device__ __forceinline__ u32 byte_of16(u32x4 v, u32 c)
{
    const u32 lo = (c & 4u) ? v.y : v.x, hi = (c & 4u) ? v.w : v.z;
    return (((c & 8u) ? hi : lo) >> (8u * (c & 3u))) & 0xffu;
}

// rle_encode with automatic symbol choice, rle.c:48-138.  S.F holds the byte histogram, S.T the repeats per symbol.  The repeat
// counts and the split are taken by all threads, each on its own chunk of the input.  `tiles`: RLE_LDS_BYTES of LDS.
// Results: S.rl_nsyms, S.rl_lits, S.rl_runs (bytes); symbols in S.alpha[0..nsyms).  Ends on a workgroup barrier.
#define RLE_SLOTS_AT 5120u
#define RLE_SLOT_BYTES 80u
#define RLE_LDS_BYTES (32u * 257u * 4u)                 // wg_hist8_t<true>'s counters; the split's slots need RLE_SLOTS_AT + 256 * RLE_SLOT_BYTES
__device__ void wg_rle_split(const u8 *data, u32 n, u8 *lits_end, u8 *runs_end, EncShared &S, u8 *tiles, u32 tid)
{
    const u32 lane = tid & (WAVE - 1);
    PROF_INIT;
    const u32 *rep = S.T;                                // repeats per symbol: counted with the histogram (wg_hist8_t<true>)
    if (tid < 16) {                                      // selector byte j = the j-th set bit of the mask (0x0c: a zero byte)
        u32 sel = 0, j = 0;
        for (u32 bit = 0; bit < 4; bit++) if ((tid >> bit) & 1u) sel |= bit << (8 * j++);
        for (; j < 4; j++) sel |= 0x0cu << (8 * j);
        S.sel4[tid] = sel;
    }
    __syncthreads();
    PROF(11);
    {
        // rle.c:74-84: a symbol is run-length coded if more than half of its occurrences repeat the byte before them; the
        // symbols in ascending order (FRONT_THREADS == 256: a thread per byte value)
        const bool use = 2 * (u64)rep[tid] > (u64)S.F[tid];
        const u64 um = __ballot(use);
        if (lane == 0) S.wtmp[tid / WAVE] = (u32)__popcll(um);
        S.present[tid] = use;
        __syncthreads();
        u32 at = (u32)__popcll(um & ((1ull << lane) - 1ull)), all = 0;
        for (u32 w2 = 0; w2 < FRONT_THREADS / WAVE; w2++) { const u32 c = S.wtmp[w2]; if (w2 < tid / WAVE) at += c; all += c; }
        if (use) S.alpha[at] = (u8)tid;
        if (tid == 0) S.rl_nsyms = all;
    }
    __syncthreads();

    // The split itself: every thread owns one contiguous chunk of the input and walks it twice.
    //   walk A: literals in the chunk, the position of its first literal, the varint bytes of the runs that END inside
    //           the chunk, and the run left open at its end (the chunk's last literal, if that is an RLE symbol);
    //   between: the next literal after each chunk (a suffix minimum over the chunks' first literals) closes the open
    //           runs, and exclusive sums over the chunks place every chunk's literals and run bytes;
    //   walk B: the same walk, now writing.
    // A byte is a literal unless it repeats an RLE symbol (rle.c:121-133); an RLE-symbol literal is followed, in the run
    // stream, by varint(number of repeats behind it).  Both walks work on 16-bit masks of a 16-byte piece - positions
    // that repeat their predecessor, positions holding an RLE symbol - so that walk A is mask arithmetic only and walk B
    // loops over the literals, not over the bytes (the byte-by-byte form cost ~45 instructions per input byte with every
    // lane on its own branch: 2.3 ms per MiB for both walks).
    // (The first version swept 16 KB LDS tiles from the top with five workgroup barriers per tile: 1.2 ms per 256 KiB.)
    u32 *cF = (u32 *)tiles, *cL = cF + 256, *cV = cL + 256, *cP = cV + 256, *cN = cP + 256;   // first / literals / run bytes / open run / next literal
    u8 *slot = tiles + RLE_SLOTS_AT + tid * RLE_SLOT_BYTES;                           // this thread's 64 bytes of input (stride 80: no two of eight lanes on one bank)
    const u32 NONE = 0xffffffffu;
    const u32 csz = ((n + FRONT_THREADS - 1) / FRONT_THREADS + 15u) & ~15u;            // chunk bytes, a multiple of 16
    const u32 c0 = tid * csz < n ? tid * csz : n, c1 = c0 + csz < n ? c0 + csz : n;
    // one walk; EMIT = false: count, EMIT = true: write at (lp, vp).  `open` = position of the RLE-symbol literal whose run is running.
    auto walk = [&](auto emitc, u32 &nlit, u32 &first, u32 &vbytes, u32 &open, gu8 *lp, gu8 *vp) {
        constexpr bool EMIT = decltype(emitc)::value;
        PieceOut lo{lp, 0, 0, 0, 0, 0, 0};
        ByteOut vo{vp, {0, 0, 0, 0}, 0};
        u32 prev = c0 ? to_global(data)[c0 - 1] : 256u;
        // A thread's pieces are consecutive: four at a time (64 bytes) come through the thread's own LDS slot, requested
        // one refill before they are looked at.  (Round 2 requested the next piece at the top of every trip: with stores
        // in flight behind it - their number unknown to the compiler, the counter in order - the first use of the piece
        // in hand waited for everything, the request just made included: a trip to memory per 16 bytes, 1.3 ms per
        // 1 MiB q8 block in the writing walk.)
        auto piece = [&](u32 q) -> u32x4 {
            u32x4 v = {0, 0, 0, 0};
            if (q + 16 <= c1) v = *(GAS const u32x4_unaligned *)(to_global(data) + q);
            else if (q < c1) {
                u32 w[4] = {0, 0, 0, 0};
                for (u32 c = 0; c < c1 - q; c++) w[c >> 2] |= (u32)to_global(data)[q + c] << (8 * (c & 3));
                v = u32x4{w[0], w[1], w[2], w[3]};
            }
            return v;
        };
        u32x4 n0 = piece(c0), n1 = piece(c0 + 16), n2 = piece(c0 + 32), n3 = piece(c0 + 48);
        for (u32 p0 = c0; p0 < c1; p0 += 16) {
            const u32 cnt = c1 - p0 < 16 ? c1 - p0 : 16;
            const u32 ph = (p0 - c0) & 63u;
            if (ph == 0) {
                u32x4 *sl = (u32x4 *)slot;
                sl[0] = n0; sl[1] = n1; sl[2] = n2; sl[3] = n3;
                n0 = piece(p0 + 64); n1 = piece(p0 + 80); n2 = piece(p0 + 96); n3 = piece(p0 + 112);
            }
            const u32x4 v = *(const u32x4 *)(slot + ph);
            const u32 valid = (1u << cnt) - 1u;
            const u32 P = flag_mask16(v, S.present);                       // positions holding an RLE symbol
            const u32 L = valid & ~(eq_prev_mask16(v, prev) & P);          // literals
            prev = byte_of16(v, cnt - 1);
            if (!L) continue;
            if (first == NONE) first = p0 + (u32)__ffs((int)L) - 1u;
            if (!EMIT) {
                nlit += (u32)__popc(L);
                const u32 hi = 31u - (u32)__clz((int)L);                   // the piece's last literal
                if (open != NONE) vbytes += var_len(p0 + (u32)__ffs((int)L) - 1u - open - 1u);
                vbytes += (u32)__popc(L & P & ~(1u << hi));                // runs that begin and end inside the piece: one byte each
                open = ((P >> hi) & 1u) ? p0 + hi : NONE;
            } else {
                // the literals: what L picks from each dword, moved to its bottom by a selector looked up under the
                // four mask bits; the run stream: the run open at the piece's first literal ends there, every RLE-symbol
                // literal but the piece's last has its run inside the piece (0 .. 14 repeats: one byte)
                // (round 2 looped over the literals one by one, every lane on its own count: 1.35 of k_enc_front's
                //  2.5 ms per 1 MiB q8 block)
                const u32 f = (u32)__ffs((int)L) - 1u, hi = 31u - (u32)__clz((int)L);
                if (open != NONE) vo.put_var(p0 + f - open - 1u);
                const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 Lk = (L >> (4 * k)) & 15u;
                    lo.append(__builtin_amdgcn_perm(0u, w[k], S.sel4[Lk]), (u32)__popc(Lk));
                }
                u32 m = L & P & ~(1u << hi);
                while (m) {
                    const u32 c = (u32)__ffs((int)m) - 1u;
                    m &= m - 1u;
                    vo.put((u32)__ffs((int)(L >> (c + 1u))) - 1u);
                }
                open = ((P >> hi) & 1u) ? p0 + hi : NONE;
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
    {
        // by all threads (one thread walking the 256 chunks twice: 39 us per block)
        // next literal after each chunk: the first of the later chunks' first literals (NONE is the largest value)
        const u32 wv = tid / WAVE;
        u32 v = first;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const u32 o = __shfl_down(v, d); if (lane + (u32)d < WAVE) v = v < o ? v : o; }
        if (lane == 0) S.wtmp[wv] = v;                                         // min over the wave's chunks
        u32 nx = __shfl_down(v, 1);
        if (lane == WAVE - 1) nx = NONE;
        __syncthreads();
        for (u32 w2 = wv + 1; w2 < FRONT_THREADS / WAVE; w2++) { const u32 o = S.wtmp[w2]; nx = nx < o ? nx : o; }
        if (nx == NONE) nx = n;
        cN[tid] = nx;
        // where each chunk's literals and run bytes go: exclusive sums
        const u32 l = nlit, vb = vbytes + (open != NONE ? var_len(nx - open - 1u) : 0u);
        const u32 li = wave_incl_scan(l, lane), vi = wave_incl_scan(vb, lane);
        if (lane == WAVE - 1) { S.wtmp[4 + wv] = li; S.wtmp[8 + wv] = vi; }
        __syncthreads();
        u32 lbase = 0, vbase = 0, lall = 0, vall = 0;
        for (u32 w2 = 0; w2 < FRONT_THREADS / WAVE; w2++) {
            const u32 a = S.wtmp[4 + w2], c = S.wtmp[8 + w2];
            if (w2 < wv) { lbase += a; vbase += c; }
            lall += a; vall += c;
        }
        cL[tid] = lbase + li - l; cV[tid] = vbase + vi - vb;
        if (tid == 0) { S.rl_lits = lall; S.rl_runs = vall; }
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
    __shared__ struct { i32 status; u32 go, order, dlen, nested_len, flags, hl, run; u64 data; double e10, e12; int max_tot; u32 wcnt[4], wlo[4], whi[4], lo, span; } H;

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
    // the block's staging region for the transforms (r4x16_common.h: enc_var_layout; laid out by k_enc_voff)
    const EncVar V = enc_var_layout(in_size, order);
    u8 *var = ws.var ? ws.var + ws.voff[b] : nullptr;

    // ---- container header (:1144-1237) ---------------------------------------------------------
    if (tid == 0) {
        H.run = 0;
        ws.stat[b].run = 0;
        EncItem *I2 = &ws.items[2 * gridDim.x + b];                        // the order-1 table as an order-0 stream (k_enc_tables)
        I0->active = 0; I1->active = 0; I0->pay_len = 0; I1->pay_len = 0; I0->packed = 0; I1->packed = 0;
        I0->affine = 0; I1->affine = 0; I2->affine = 0;
        I2->active = 0; I2->pay_len = 0; I2->packed = 0;
        I0->blk = b; I1->blk = b; I2->blk = b;
        D->cat = 0; D->rle_on = 0; D->tab_len = 0; D->hdr_len = 0; D->dlen = 0; D->tab = (u64)tab; D->nest_on = 0;
        i32 st = ST_OK;
        u32 go = 0;
        H.flags = 0; H.hl = 0;
        if (cap < compress_bound(in_size, order)) st = ST_CAPACITY;
        else {
            if (in_size <= 20) order &= ~X_STRIPE;                         // :1151
            if (order & X_STRIPE) st = ST_UNSUPPORTED;                     // host entry points split stripes
            else if (order & X_CAT) {                                      // :1218-1225
                D->hdr[0] = X_CAT;
                D->hdr_len = 1 + var_put(D->hdr + 1, in_size);
                D->cat = 1; D->data = (u64)in; D->dlen = in_size; D->flags = X_CAT;
            } else if (V.total && (!var || ws.voff[b] + V.total > ws.var_bytes)) {
                st = ST_UNSUPPORTED;                                       // batch was sized without transform staging, or for less data than it holds
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
            u8 *pbuf = var + V.packed;
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
            u8 *lits_end = var + V.meta;                                   // (the literals' region ends where the meta's begins)
            u8 *meta_end = var + V.scratch2;
            PROF(3);
            wg_hist8_t<true>(data, n, S.F, S.T, (u32 *)dyn, tid);          // + per symbol, the bytes that repeat their predecessor
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
                wg_hist8(m, mlen, S.F, (u32 *)dyn, tid);                   // (one wave counting ~170 KB of run lengths: 0.2 ms per 1 MiB q8 block)
                const bool mrec = ws.meta_records && enc_rec_img_bytes(256u, 1u) + ENC_RING_BYTES <= ws.direct_budget &&
                                  mlen >= enc_rec_img_bytes(256u, 1u) / 4u;                        // (uniform)
                if (w0) {
                    enc_o0_tables(mlen, mtab, imgm, S, lane, mrec ? ws.rcptab : nullptr);
                    if (lane == 0) {
                        D->rle_on = 1; D->rle_mlen = mlen; D->rle_lits = nlits;
                        D->rle_meta = (u64)m; D->meta_tab = (u64)mtab; D->meta_tab_len = S.tab_len;
                        if (S.status != ST_OK) D->status = S.status;
                        I1->data = (u64)m; I1->n = mlen; I1->image = (u64)imgm; I1->bits = O0_BITS; I1->order = 0;
                        I1->ns = 256; I1->img_bytes = mrec ? enc_rec_img_bytes(256u, 1u) : ENC_IMG_IDX + 2u * 257u;
                        I1->packed = mrec ? 2u : 0u; I1->affine = mrec ? 1u : 0u;
                        I1->scratch_end = (u64)(var + V.total);
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
            // the range of the byte values seen (byte 0 is in the alphabet by rule, not by being seen)
            const u64 seen = __ballot(S.F[tid] != 0);
            if (lane == 0) { H.wlo[tid >> 6] = seen ? (tid & ~63u) + (u32)__ffsll((unsigned long long)seen) - 1u : 256u;
                             H.whi[tid >> 6] = seen ? (tid & ~63u) + 63u - (u32)__clzll((long long)seen) : 0u; }
        }
        __syncthreads();
        if (tid == 0) {
            u32 l = 256, h = 0;
            for (u32 w = 0; w < 4; w++) { if (H.wlo[w] < l) l = H.wlo[w]; if (H.wlo[w] != 256u && H.whi[w] > h) h = H.whi[w]; }
            H.lo = l < 256u ? l : 0u; H.span = l < 256u ? h - l + 1u : 1u;
        }
        __syncthreads();
        ns = S.nsym;
        nsx = prov ? ns + 1 : ns;                                         // row / column count of the counters
        f_in_lds = 4u * nsx * nsx <= dyn_bytes;
        if (prov && !f_in_lds) continue;                                  // large alphabets: straight to the exact route
        u32 copies = (!f_in_lds || nsx * nsx > 18u * FRONT_THREADS) ? 1u : (16u * nsx * nsx <= dyn_bytes ? 4u : (8u * nsx * nsx <= dyn_bytes ? 2u : 1u));   // (wg_hist1 sums up to 18 counters per thread)
        // the alphabet as a range of byte values (wg_hist1_range), if that costs no counter copies
        const u32 lo = H.lo, nsa = H.span + (prov ? 1u : 0u);
        const u32 copies_r = (!f_in_lds || nsa * nsa > 18u * FRONT_THREADS) ? 0u : (16u * nsa * (nsa | 1u) <= dyn_bytes ? 4u : (8u * nsa * (nsa | 1u) <= dyn_bytes ? 2u : 1u));
        const bool ranged = copies_r >= copies && n >= 16;
        if (ranged) copies = copies_r;
        if (f_in_lds) { for (u32 j = tid; j < copies * (ranged ? nsa * (nsa | 1u) : nsx * nsx); j += FRONT_THREADS) ((u32 *)dyn)[j] = 0; }
        else          { for (u32 j = tid; j < nsx * nsx; j += FRONT_THREADS) Fg[j] = 0; }
        __syncthreads();
        // pass 2 over the block: order-1 pair histogram, all waves
        PROF(8);
        bool hit = false;
        if (ranged)        hit = wg_hist1_range(data, n, (u32 *)dyn, nsx, copies, S, lo, nsa, prov, tid);
        else if (f_in_lds) wg_hist1(data, n, (u32 *)dyn, nsx, copies, S.idx_of, tid);
        else               wg_hist1(data, n, Fg, nsx, 1u, S.idx_of, tid);
        PROF(9);
        if (!prov) break;
        // any pair with the overflow symbol (row ns or column ns of the (ns + 1)^2 counters)?
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
    u8 *scratch = ws.scratch + (u64)b * ws.scratch_stride;      // pair counters, then the nested table stream (NEST_AREA)
    // The payload is written backwards from the end of the caller's own slot - what the reference's coders do with their
    // output buffer (rANS_static4x16pr.c:396-402, :706-710: `ptr = out_end`, then a memmove behind the table) - so the
    // workspace holds no bound-sized area per block any more.  k_enc_front has checked the slot against
    // rans_compress_bound_4x16(in_size, order); the end used is 16-byte aligned inside it (the bound's own slack of
    // twenty bytes covers the difference).
    u8 *scratch_end;
    {
        const int i = base + (int)b;
        const int order_i = a.d_order ? a.d_order[i] : a.order;
        const u64 slot = (u64)(a.out + a.out_off[i]);
        scratch_end = (u8 *)((slot + compress_bound(a.in_size[i], order_i)) & ~15ull);
    }
    const u8 *data = (const u8 *)D->data;
    const u32 n = D->dlen;

    for (u32 j = lane; j < 256; j += WAVE) S.F[j] = ST->F0[j];
    if (lane == 0) H.status = ST_OK;
    wsync();
    if (ST->order == 0) {
        // (records pay a larger image for a shorter step: only for streams with at least a step per 16 bytes of image)
        const bool rec = enc_rec_img_bytes(256u, 1u) + ENC_RING_BYTES <= ws.direct_budget && n >= enc_rec_img_bytes(256u, 1u) / 4u;      // (uniform)
        enc_o0_tables(n, tab, img, S, lane, rec ? ws.rcptab : nullptr);
        if (lane == 0) {
            D->status = S.status;
            D->tab_len = S.tab_len;
            I0->data = (u64)data; I0->n = n; I0->image = (u64)img; I0->bits = O0_BITS; I0->order = 0;
            I0->ns = 256; I0->img_bytes = rec ? enc_rec_img_bytes(256u, 1u) : ENC_IMG_IDX + 2u * 257u;
            I0->packed = rec ? 2u : 0u;
            I0->affine = rec ? 1u : 0u;                    // order 0: symbols index the records directly
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
    // a batch that leaves LDS to spare takes symbol records (the short step); else quality-sized 10-bit tables pack
    const bool recs = enc_rec_img_bytes(ns, ns) + ENC_RING_BYTES <= ws.direct_budget && enc_rec_img_bytes(ns, ns) <= ENC_IMG_MAIN &&
                      n >= enc_rec_img_bytes(ns, ns) / 4u;
    const bool packed = !recs && bits == 10 && ns >= ENC_PK_MIN_NS && ns <= ENC_PK_MAX_NS;
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
        if (recs) {
            u32x4 *row = (u32x4 *)(img + ENC_IMG_IDX) + r * ns;
            for (u32 j = 0; j < ns; j++) { const u32 f = Fp[r * ns + j] << sh; row[j] = enc_record(x, f, bits, ws.rcptab); x += f; }
        } else if (!packed) {
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
    // symbol records: is the compact index an offset of the byte value?  (quality values are a run of consecutive bytes;
    // byte 0, which every order-1 alphabet lists, must then not occur in the data unless the run starts right behind it)
    u32 aff = 0;
    if (recs && ns >= 2) {
        const u32 c = (u32)S.alpha[1] - 1u;
        bool ok = true;
        for (u32 j = lane; j < ns; j += WAVE) if (j >= 1 && (u32)S.alpha[j] != j + c) ok = false;
        if (c != 0) for (u32 r = lane; r < ns; r += WAVE) if (Fp[r * ns] != 0) ok = false;      // byte 0 coded somewhere
        aff = __ballot(!ok) ? 0u : c + 1u;
    }
    const u32 final_len = 1 + tlen;
    wsync();
    if (lane == 0) {
        D->tab_len = final_len;
        I0->data = (u64)data; I0->n = n; I0->image = (u64)img; I0->bits = bits; I0->order = 1;
        I0->ns = ns; I0->img_bytes = recs ? enc_rec_img_bytes(ns, ns) : packed ? enc_pk_img_bytes(ns) : ENC_IMG_IDX + 2u * ns * (ns + 1);
        I0->packed = recs ? 2u : packed ? 1u : 0u;
        I0->affine = aff;
        I0->scratch_end = (u64)scratch_end;
        I0->active = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// k_enc_finish
// ---------------------------------------------------------------------------------------------
#define FINISH_THREADS 256u
// Move n bytes DOWN inside one buffer (dst <= src) by a workgroup of NT threads.  Where the regions do not overlap
// it is group_copy; where they do, a tile is read by all threads before any of them writes it (a thread that ran ahead
// by more than the distance between the regions would otherwise overwrite bytes another one has yet to read).
template <u32 NT>
__device__ __forceinline__ void slot_move(u8 *dst, const u8 *src, u32 n, u32 t)
{
    if (dst == src || n == 0) return;
    if (src >= dst + n) { group_copy<NT>(dst, src, n, t); return; }
    for (u32 base = 0; base < n; base += 32u * NT) {                      // 32 bytes per thread and tile
        const u32 at = base + 32u * t;
        u32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
        u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};                              // the one partial piece, at the very end
        const u32 left = at < n ? n - at : 0u;
        if (left >= 32u) { v0 = *(const u32x4_unaligned *)(src + at); v1 = *(const u32x4_unaligned *)(src + at + 16); }
        else {
#pragma unroll
            for (u32 j = 0; j < 32u; j++) if (j < left) w[j >> 2] |= (u32)src[at + j] << (8u * (j & 3u));
        }
        __syncthreads();
        if (left >= 32u) { *(u32x4_unaligned *)(dst + at) = v0; *(u32x4_unaligned *)(dst + at + 16) = v1; }
        else {
#pragma unroll
            for (u32 j = 0; j < 32u; j++) if (j < left) dst[at + j] = (u8)(w[j >> 2] >> (8u * (j & 3u)));
        }
        __syncthreads();
    }
}

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
            // the payload sits at the end of this very slot (k_enc_tables): moved down behind the table.  Header, meta
            // and table were written below `pos`, which the reference's own layout keeps below the payload's start
            // (the same bytes in a buffer of the same bound); a stream that did not would have been cut: reported.
            const u8 *psrc = (const u8 *)I0->scratch_end - pay;
            if (out + pos > psrc) { if (lane == 0) { a.status[i] = ST_CAPACITY; a.out_size[i] = 0; } return; }
            slot_move<FINISH_THREADS>(out + pos, psrc, pay, lane);
            pos += pay;
        }
    }
    if (lane == 0) {
        out[0] = (u8)flags;
        a.status[i] = ST_OK;
        a.out_size[i] = pos;
    }
}

// The size of each block's staging region for the transforms (enc_var_layout); r4x16_voff_scan turns the sizes into
// where the regions start.
__global__ __launch_bounds__(256) void k_enc_vsize(BatchArgs a, int base, int nblk, u64 *voff)
{
    const int b = (int)(blockIdx.x * 256u + threadIdx.x);
    if (b < nblk) voff[b] = enc_var_layout(a.in_size[base + b], a.d_order ? a.d_order[base + b] : a.order).total;
}

// ---- host-callable launchers -------------------------------------------------------------------
extern "C" bool r4x16_first_on_device(u32 bit);                                          // r4x16_decode.hip
extern "C" void r4x16_launch_enc_front(const BatchArgs *a, const EncWs *ws, int base, int nblk, hipStream_t s, const R4Opts *o)
{
    if (ws->var) {
        hipLaunchKernelGGL(k_enc_vsize, dim3((nblk + 255) / 256), dim3(256), 0, s, *a, base, nblk, ws->voff);
        r4x16_voff_scan(ws->voff, nblk, s);
    }
    // static + dynamic LDS exceeds the 64 KB default; gfx950 has 160 KB per CU
    if (r4x16_first_on_device(2u))
        (void)hipFuncSetAttribute((const void *)k_enc_front, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const u32 dynb0 = o->v[OPT_FRONT_LDS] > 0 ? (u32)o->v[OPT_FRONT_LDS] : FRONT_DYN_LDS;  // tuning aid
    const u32 dynb = dynb0 < RLE_LDS_BYTES ? RLE_LDS_BYTES : dynb0 > 65536u - 8192u ? 65536u - 8192u : dynb0;   // (the run-length split's slots; wg_hist8's 16 x 257 counters are smaller)
    hipLaunchKernelGGL(k_enc_front, dim3(nblk), dim3(FRONT_THREADS), dynb, s, *a, *ws, base, dynb);
}
extern "C" void r4x16_launch_enc_tables(const BatchArgs *a, const EncWs *ws, int base, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL(k_enc_tables, dim3(nblk), dim3(WAVE), TABLES_DYN_LDS, s, *a, *ws, base);
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

// (round 2's LDS form of this walk - image and reciprocal table in LDS, input in 16-byte windows - went in round 4: the
//  streams whose images fit LDS now run on the 4x16 encoder's pipeline, r4x16_enc_chain.hip: k8_enc_chain_pipe)
// streams by LDS need: class 0 - order-0 images (770 bytes), class 1 - order-1 images of up to 47 symbols, class 2 - the
// rest (tables through L2)
#define X8E_SLOT0 928u        // one-row image (770 bytes) + the ring of emitted bytes
#define X8E_SLOT1 4880u       // order-1 images of up to 46 symbols (4,580 bytes) + the ring
__global__ __launch_bounds__(256) void k8_enc_classify(const EncItem *items, int nitems, u32 *cls, u32 *count)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= nitems) return;
    const EncItem *I = &items[i];
    u32 c = CLS_NONE;
    if (I->active) {
        const u32 need = ENC_IMG_IDX + (I->order ? 2u * I->ns * (I->ns + 1u) : 2u * 257u) + ENC_RING_BYTES;     // (image + the ring of emitted bytes, r4x16_enc_chain.h)
        c = need <= X8E_SLOT0 ? 0u : need <= X8E_SLOT1 ? 1u : 2u;
        atomicAdd(&count[c], 1u);
    }
    cls[i] = c;
}

// the chain kernel for images in LDS: r4x16_enc_chain.hip (k8_enc_chain_pipe: the 4x16 encoder's software pipeline with
// rANS 4x8's byte renormalisation); images too large for LDS keep the general form:
__global__ __launch_bounds__(WAVE) void k8_enc_chain_gm(EncItem *items, const u32 *rcptab_, const u32 *list, const u32 *count, u32 room)
{
    const u32 lane = threadIdx.x, quad = lane >> 2;
    const int nmine = (int)count[0];
    list += count[CLS_MAX];
    if ((int)blockIdx.x * 16 >= nmine) return;
    const int slot = (int)blockIdx.x * 16 + (int)quad;
    const bool mine = slot < nmine;
    EncItem *I = &items[list[mine ? slot : (int)blockIdx.x * 16]];
    const bool active = mine && I->active;
    gcu32 *rcptab = to_global(rcptab_);
    gcu8 *data = (gcu8 *)I->data;
    gu8 *send = (gu8 *)I->scratch_end;
    const u32 n = I->n, ns = I->ns, order = active ? I->order : 2u;
    gcu8 *im = (gcu8 *)I->image;
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

extern "C" void r4x16_launch_cls_group(const u32 *cls, int nitems, u32 *count, u32 *list, hipStream_t s);   // r4x16_decode.hip
extern "C" void r4x16_launch_cls_zero(u32 *count, hipStream_t s);
extern "C" void r4x8_enc_chain_launch(EncItem *items, const u32 *rcptab, u8 *dump, const u32 *list, const u32 *count, int nblk, u32 slot_bytes,
                                      int qpw, int spw, hipStream_t s);    // r4x16_enc_chain.hip
extern "C" void r4x8_launch_encode(const BatchArgs *a, const EncWs *ws, int base, int nblk, hipStream_t s)
{
    if (r4x16_first_on_device(8u))
        (void)hipFuncSetAttribute((const void *)k8_enc_front, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_DYN_LDS);
    const u32 room = (u32)(ws->scratch_stride > 0xffffffffull ? 0xffffffffu : ws->scratch_stride);
    hipLaunchKernelGGL(k8_enc_front, dim3(nblk), dim3(FRONT_THREADS), FRONT_DYN_LDS, s, *a, *ws, base);
    // (the 4x16 encoder's grouping arrays, with round 2's plain grouping by class: key = class, cnt = count / start / cursor)
    r4x16_launch_cls_zero(ws->sched.cnt, s);
    hipLaunchKernelGGL(k8_enc_classify, dim3((nblk + 255) / 256), dim3(256), 0, s, (const EncItem *)ws->items, nblk, ws->sched.key, ws->sched.cnt);
    r4x16_launch_cls_group(ws->sched.key, nblk, ws->sched.cnt, ws->sched.list, s);
    // class 0: one-row images (order 0), 4 waves x 16 streams; class 1: order-1 images of up to 46 symbols, 30 streams
    // in four waves beside the one reciprocal table (16,400 + 30 x 4,880 = 162,800 bytes); class 2: tables through L2
    r4x8_enc_chain_launch(ws->items, ws->rcptab, ws->dump, ws->sched.list, ws->sched.cnt + 0, nblk, X8E_SLOT0, 64, 16, s);
    r4x8_enc_chain_launch(ws->items, ws->rcptab, ws->dump, ws->sched.list, ws->sched.cnt + 1, nblk, X8E_SLOT1, 30, 8, s);
    hipLaunchKernelGGL(k8_enc_chain_gm, dim3((nblk + 15) / 16), dim3(WAVE), 0, s, ws->items, ws->rcptab,
                       (const u32 *)ws->sched.list, (const u32 *)(ws->sched.cnt + 2), room);
    hipLaunchKernelGGL(k8_enc_finish, dim3(nblk), dim3(FINISH_THREADS), 0, s, *a, *ws, base, room);
}
