// r4x16_dev.h — device-side helpers shared by the decode and encode kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include "r4x16_common.h"

#define WAVE 64

// Pointers rebuilt from 64-bit fields are "generic" to the compiler, which then emits FLAT
// loads/stores.  FLAT operations count on lgkmcnt as well as vmcnt, so every wait for an LDS
// read would also wait for outstanding HBM stores.  The hot loops therefore use explicit
// global-address-space pointers (global_load / global_store, vmcnt only).
#define GAS __attribute__((address_space(1)))
typedef GAS u8        gu8;
typedef GAS const u8  gcu8;
typedef GAS u16       gu16;
typedef GAS u32       gu32;
typedef GAS const u32 gcu32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));   // two dwords at a 4-byte aligned address
typedef GAS const u32x4 gcu32x4;
typedef u32 __attribute__((aligned(1))) u32_unaligned;
typedef u32 u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef u32 u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));
typedef u64 __attribute__((aligned(1))) u64_unaligned;
typedef u16 __attribute__((aligned(1))) u16_unaligned;
#define LAS __attribute__((address_space(3)))                   // LDS
typedef LAS const volatile u32 lvcu32;          // volatile LDS dword (keep it 4-byte aligned: misaligned LDS reads are slow)
template <class T> __device__ __forceinline__ GAS T *to_global(T *p) { return (GAS T *)p; }
template <class T> __device__ __forceinline__ GAS const T *to_global(const T *p) { return (GAS const T *)p; }

// Ordering point for code that one wave runs on its own inside a larger workgroup: the lanes of a
// wave execute in lock-step, so all that is needed is that earlier LDS/global accesses have
// completed and that the compiler does not move accesses across this point.
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Ordering point for global memory that one thread of a workgroup wrote and another reads after the next barrier.
// (__threadfence() is device scope: on gfx950 an L2 write-back and a cache invalidate per call, `buffer_wbl2 sc1` /
// `buffer_inv sc1`, microseconds under load - and nothing here is read by another workgroup before the kernel ends.)
__device__ __forceinline__ void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// Side streams of a context for the chain kernels.  The chain kernels are launched once per LDS size class; in stream
// order a batch that mixes alphabets (q4 / q8 / q40 blocks: several classes) pays one chain latency per class, one after
// the other, although each launch leaves most of the chip idle.  So the class launches of a batch are dealt out over the
// caller's stream and these side streams, and joined again (events); a device-written plan gives every class its share
// of the chip (r4x16_sched.h).  Streams of ONE priority share two hardware queues, and the launches in a queue run one
// after the other (tools/micro/launch_overlap.hip: 4 streams of one priority = 2 at a time; 6 streams cycling through
// the three priorities = 6 at a time; hipExtAnyOrderLaunch on one stream = 1 at a time on gfx950) - hence five side
// streams: two of the highest priority, two of the lowest, one of the caller's presumed own.
#define R4_FORK_STREAMS 5
struct R4Fork {
    hipStream_t aux[R4_FORK_STREAMS];
    hipEvent_t ev[R4_FORK_STREAMS + 1];
    int n;                         // side streams in use (0: none)
    // fork: the side streams in `mask` (bit j = stream j of pick(); bit 0, the caller's own, is ignored) wait for what
    // `s` holds so far.  Only the streams a batch uses are forked and joined: an idle side stream still has to be
    // scheduled by the firmware to pass an event on, and a join over all five cost a millisecond per call.
    void begin(hipStream_t s, unsigned mask) const
    {
        if (!n || !(mask >> 1)) return;
        (void)hipEventRecord(ev[0], s);
        for (int i = 0; i < n; i++) if (mask & (2u << i)) (void)hipStreamWaitEvent(aux[i], ev[0], 0);
    }
    hipStream_t pick(hipStream_t s, unsigned k) const { const unsigned j = k % (unsigned)(n + 1); return j == 0 ? s : aux[j - 1]; }
    // join: `s` waits for every side stream in `mask`
    void end(hipStream_t s, unsigned mask) const
    {
        for (int i = 0; i < n; i++) if (mask & (2u << i)) { (void)hipEventRecord(ev[i + 1], aux[i]); (void)hipStreamWaitEvent(s, ev[i + 1], 0); }
    }
};

// Options of a context (include/rans4x16_hip.h: rans4x16_hip_set_option / rans4x16_hip_get_option).  The R4X16_*
// environment variables only provide the DEFAULTS, and are read once per process (r4x16_opts_defaults, r4x16_api.hip):
// nothing on a call path reads the environment.  The launchers take the calling context's options as an argument.
enum R4Opt {
    OPT_DEC_DIRECT, OPT_ENC_DIRECT, OPT_BACK_WG_PER_CU, OPT_DEC_MID, OPT_DEC_SHORT_RING,
    OPT_SCHED_SORT, OPT_SCHED_CLAIM, OPT_SCHED_CONCURRENT, OPT_SCHED_TRACE, OPT_SCHED_LEARN, OPT_MAX_WS_MB,
    OPT_HOST_STRIPE_DEV, OPT_HOST_PIPE_MB, OPT_HOST_THREADS, OPT_HOST_LANES, OPT_HOST_SLAB_MIN_MB,
    OPT_HOST_DEC_SLABS, OPT_HOST_ENC_SLABS, OPT_HOST_PACK, OPT_HOST_TRACE,
    OPT_DEC_QPW, OPT_DEC_QPW_SMALL, OPT_DEC_QPW_PK, OPT_DEC_QPW_DIR,
    OPT_ENC_QPW, OPT_ENC_WAVES, OPT_ENC_QPW_REC, OPT_ENC_QPW_CAP, OPT_FRONT_LDS,
    OPT_COMBINE, OPT_COMBINE_WINDOW_US, OPT_COMBINE_MAX, OPT_COMBINE_WORKERS, OPT_COMBINE_MAX_MB, OPT_NUMA,
    OPT_COUNT
};
struct R4Opts { long v[OPT_COUNT]; };
extern "C" const R4Opts *r4x16_opts_defaults(void);

// Arguments of a device-resident batch (include/rans4x16_hip.h, *_dev entry points).
struct BatchArgs {
    const u8  *in;
    const u64 *in_off;
    const u32 *in_size;
    u8        *out;
    const u64 *out_off;
    const u32 *out_cap;
    u32       *out_size;
    i32       *status;
    const i32 *d_order;
    int        order;
    int        n;
};

// ---------------------------------------------------------------------------------------------
// Sequential byte reader for the single lane that parses headers and tables.  Keeps an
// 8-byte window so that a run of dependent byte reads costs one global load per 8 bytes.
// `fresh` makes the loads bypass the CU's vector L1 (data written earlier by this launch).
// ---------------------------------------------------------------------------------------------
struct ByteSrc {
    const u8 *base;
    u64 win;
    u64 win_addr;
    bool fresh;
    __device__ ByteSrc(const u8 *b, bool fresh_ = false) : base(b), win(0), win_addr(~0ull), fresh(fresh_) {}
    __device__ __forceinline__ u8 at(u32 pos) {
        u64 a = (u64)(base + pos);
        u64 al = a & ~7ull;
        if (al != win_addr) {
            win_addr = al;
            win = fresh ? __hip_atomic_load((const u64 *)al, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                        : *(const u64 *)al;
        }
        return (u8)(win >> ((a & 7) * 8));
    }
};

// A window of 1 KB of a byte stream held in registers, 16 bytes per lane: the lane that parses reads any
// byte of it with a scalar lane read, so a run of dependent byte reads never waits for memory.  The window
// is (re)loaded by the whole wave with fill(); bytes outside it fall back to the ByteSrc.
struct WinSrc {
    ByteSrc *far;
    u32x4 held;
    u32 wbase;           // stream position of lane 0's first byte
    u32 wlen;            // bytes of the window that are loaded (0 = no window)
    __device__ WinSrc(ByteSrc *f) : far(f), held{0, 0, 0, 0}, wbase(0), wlen(0) {}
    // all lanes: load [pos & ~15, +1024) as far as it lies below `limit` (bytes of the stream that may be read)
    __device__ __forceinline__ void fill(u32 pos, u32 limit, u32 lane) {
        wbase = pos & ~15u;
        const u32 mine = wbase + 16u * lane;
        held = u32x4{0, 0, 0, 0};
        if (mine + 16u <= limit) held = *(const u32x4_unaligned *)(far->base + mine);
        const u32 room = limit > wbase ? (limit - wbase) & ~15u : 0u;
        wlen = room < 1024u ? room : 1024u;
    }
    __device__ __forceinline__ u8 at(u32 pos) {
        const u32 o = pos - wbase;
        if (o < wlen) {
            const u32 so = (u32)__builtin_amdgcn_readfirstlane((int)o);
            const int ln = (int)(so >> 4);
            // all four dwords of that lane, then scalar selects (a chain of conditional reads compiles to branches)
            const u32 d0 = (u32)__builtin_amdgcn_readlane((int)held.x, ln), d1 = (u32)__builtin_amdgcn_readlane((int)held.y, ln),
                      d2 = (u32)__builtin_amdgcn_readlane((int)held.z, ln), d3 = (u32)__builtin_amdgcn_readlane((int)held.w, ln);
            const u32 lo = (so & 4u) ? d1 : d0, hi = (so & 4u) ? d3 : d2;
            const u32 d = (so & 8u) ? hi : lo;
            return (u8)(d >> (8u * (so & 3u)));
        }
        return far->at(pos);
    }
    // the same without the range check, for callers that know the byte is inside the window
    __device__ __forceinline__ u32 at_inside(u32 pos) {
        const u32 so = (u32)__builtin_amdgcn_readfirstlane((int)(pos - wbase));
        const int ln = (int)(so >> 4);
        const u32 d0 = (u32)__builtin_amdgcn_readlane((int)held.x, ln), d1 = (u32)__builtin_amdgcn_readlane((int)held.y, ln),
                  d2 = (u32)__builtin_amdgcn_readlane((int)held.z, ln), d3 = (u32)__builtin_amdgcn_readlane((int)held.w, ln);
        const u32 lo = (so & 4u) ? d1 : d0, hi = (so & 4u) ? d3 : d2;
        const u32 d = (so & 8u) ? hi : lo;
        return (d >> (8u * (so & 3u))) & 0xffu;
    }
};

// varint.h:131-160 on a byte source: bytes [pos, end).  Returns bytes consumed (0 if pos >= end).
template <class SRC>
__device__ __forceinline__ u32 var_get(SRC &s, u32 pos, u32 end, u32 *v)
{
    u32 acc = 0, p = pos;
    u8 c;
    if (pos >= end) { *v = 0; return 0; }
    do {
        c = s.at(p++);
        acc = (acc << 7) | (c & 0x7f);
    } while ((c & 0x80) && p < end);
    *v = acc;
    return p - pos;
}

// varint.h:85-104
__device__ __forceinline__ u32 var_put(u8 *cp, u32 v)
{
    u32 groups = 1;
    for (u32 t = v >> 7; t; t >>= 7) groups++;
    for (int g = (int)groups - 1; g >= 0; g--)
        *cp++ = (u8)(((v >> (7 * g)) & 0x7f) | (g ? 0x80 : 0));
    return groups;
}

__device__ __forceinline__ u32 var_len(u32 v)
{
    u32 groups = 1;
    for (u32 t = v >> 7; t; t >>= 7) groups++;
    return groups;
}

// ---------------------------------------------------------------------------------------------
// Quad (4-lane) cross-lane helpers.  DPP quad_perm moves data inside each group of four
// adjacent lanes at VALU speed — no LDS round trip.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ u32 dpp_mov(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
__device__ __forceinline__ u32 quad_bcast0(u32 v) { return dpp_mov<QP(0, 0, 0, 0)>(v); }
__device__ __forceinline__ u32 quad_bcast1(u32 v) { return dpp_mov<QP(1, 1, 1, 1)>(v); }
__device__ __forceinline__ u32 quad_bcast2(u32 v) { return dpp_mov<QP(2, 2, 2, 2)>(v); }
__device__ __forceinline__ u32 quad_bcast3(u32 v) { return dpp_mov<QP(3, 3, 3, 3)>(v); }

typedef u16 u16x2 __attribute__((ext_vector_type(2)));

// For a per-lane predicate: 4-bit mask of the predicate over this lane's quad.  The compare writes a
// scalar pair, one 64-bit shift by the quad's first lane and one `and` bring it back: three
// instructions.  (Two DPP quad_perm ORs on a per-lane bit measured 2-5 % slower in the hot loops:
// more instructions, and each DPP read of a just-written register costs two wait states.)
__device__ __forceinline__ u32 quad_ballot(bool p, u32 lane)
{
    u64 m = __ballot(p);
    return (u32)(m >> (lane & ~3u)) & 0xfu;
}

__device__ __forceinline__ u32 wave_any(bool p) { return __ballot(p) != 0ull; }

// wave-wide inclusive prefix sum over 64 lanes on DPP row operations (VALU speed: four shifts inside the rows of
// sixteen lanes, then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into the upper half; the version on
// ds_bpermute shuffles cost six LDS-crossbar round trips on the dependent path of every table row and RLE trip)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_row(u32 v)          // lanes without a source (and rows outside ROW_MASK) read 0
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ u32 wave_incl_scan(u32 v, u32 lane)
{
    (void)lane;
    v += dpp_row<0x111, 0xf>(v);                       // row_shr:1
    v += dpp_row<0x112, 0xf>(v);                       // row_shr:2
    v += dpp_row<0x114, 0xf>(v);                       // row_shr:4
    v += dpp_row<0x118, 0xf>(v);                       // row_shr:8
    v += dpp_row<0x142, 0xa>(v);                       // row_bcast:15 into rows 1 and 3
    v += dpp_row<0x143, 0xc>(v);                       // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ u32 wave_sum(u32 v)
{
    return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), WAVE - 1);
}

__device__ __forceinline__ u32 pow2_ceil(u32 v)      // rANS_static4x16pr.c:105-114
{
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

// byte-granular copy by NT threads (thread index t), any alignment: head bytes until dst is 16-byte aligned, then
// 16-byte stores fed by 16-byte loads at whatever alignment the source has (global loads need none), two pieces per
// thread in flight.  (The first version fell back to dword pieces whenever source and destination were not co-aligned
// - nearly always for a payload behind a header and a table: k_enc_finish moved its 12 GB at 4 bytes per lane.)
template <u32 NT>
__device__ __forceinline__ void group_copy(u8 *dst, const u8 *src, u32 n, u32 t)
{
    u32 head = (u32)((16 - ((u64)dst & 15)) & 15);
    if (head > n) head = n;
    for (u32 i = t; i < head; i += NT) dst[i] = src[i];
    const u32 body = (n - head) >> 4;
    const u8 *sp = src + head;
    u32x4 *d16 = (u32x4 *)(dst + head);
    u32 i = t;
    for (; i + NT < body; i += 2 * NT) {
        const u32x4 a = *(const u32x4_unaligned *)(sp + 16ull * i), b = *(const u32x4_unaligned *)(sp + 16ull * (i + NT));
        d16[i] = a;
        d16[i + NT] = b;
    }
    if (i < body) d16[i] = *(const u32x4_unaligned *)(sp + 16ull * i);
    const u32 done = head + body * 16;
    for (u32 j = done + t; j < n; j += NT) dst[j] = src[j];
}

// byte-granular copy by one wave, any alignment (header / table / payload assembly): group_copy's scheme
__device__ __forceinline__ void wave_copy(u8 *dst, const u8 *src, u32 n, u32 lane) { group_copy<WAVE>(dst, src, n, lane); }
