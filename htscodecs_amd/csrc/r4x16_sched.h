// r4x16_sched.h - how the streams of a batch are handed to the persistent chain kernels (round 4).
//
// A chain kernel is launched once per LDS size class; a wave (the encoder: a workgroup) takes `qpw` consecutive
// entries of its class's list and runs until the LONGEST of them is done.  Up to round 3 a class list was in arrival
// order and the shares were dealt out with a fixed stride: fine for batches of equal blocks (every benchmark so far),
// but the reference's real callers do not make such batches - token columns of any size
// (htscodecs/tokenise_name3.c:1246-1300), a short last block in every file (tests/rANS_static4x16pr_test.c:139-176).
// With lengths log-uniform in 4 KiB .. 1 MiB a wave's longest stream is ~5x its mean one, and every class pays the
// chain latency of its longest stream, one class after the other.  Now, all on the device (the host never learns the
// sizes):
//   * SORT   each class list is ordered by chain length, longest first: a counting sort on (class, length bucket), eight
//            buckets per octave - the streams of a wave end within ~9 % of each other;
//   * CLAIM  shares are claimed from a per-class counter instead of a fixed stride: the longest shares start first,
//            the short ones fill the gaps (LPT scheduling), and a workgroup that becomes resident late - because another
//            class held its LDS - just finds less left to do;
//   * PLAN   the classes of a batch run SIDE BY SIDE: the class launches are dealt out over six streams (R4Fork), every
//            class is launched with the grid it would need alone, and a device-written plan gives it `seats` workgroups -
//            its stream's share of the chip by work (bytes / resident streams) - the others leave at once.
#pragma once
#include "r4x16_dev.h"

// what the host knows about each class: streams per workgroup and the workgroups a full chip holds
struct SchedPlan {
    u32 ncls, concurrent, claim, pad;
    u16 qpw[CLS_MAX];
    u16 wgs_full[CLS_MAX];
    float rate[CLS_MAX];     // streams' worth of progress a full chip of this class makes at once (sched_rate)
    u8  queue[CLS_MAX];      // the stream the class's launch goes out on (launches of one stream run one after the other); 0xff: not launched
};

#ifdef __HIPCC__
__device__ __forceinline__ u32 sched_bucket(u32 len)
{
    if (len < 8u) return len;
    const u32 msb = 31u - (u32)__clz((int)len);
    return 8u * (msb - 2u) + ((len >> (msb - 3u)) & 7u);                  // 8 .. 239
}
// one atomic per distinct key of the wave; returns this lane's rank among the wave's lanes with its key + the base
__device__ __forceinline__ u32 sched_wave_add(u32 *bins, u32 key, bool has)
{
    const u32 lane = threadIdx.x & (WAVE - 1);
    u64 todo = __ballot(has);
    u32 pos = 0;
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const u32 k0 = (u32)__builtin_amdgcn_readlane((int)key, lead);
        const u64 m = __ballot(has && key == k0);
        u32 base = 0;
        if ((int)lane == lead) base = atomicAdd(&bins[k0], (u32)__popcll(m));
        base = (u32)__builtin_amdgcn_readlane((int)base, lead);
        if (has && key == k0) pos = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return pos;
}
// called by the classify kernels (every thread of the launch, `has` = the item runs): key, bin count, class work
// (lds_cnt[CLS_MAX] and lds_work[2 * CLS_MAX] are the workgroup's own sums, flushed by sched_classify_flush)
__device__ __forceinline__ void sched_classify(const SchedWs &w, int i, bool in_range, u32 cls, u32 len, bool sort, u32 *lds_cnt, u64 *lds_work)
{
    const bool has = in_range && cls != CLS_NONE;
    const u32 key = has ? (cls << 8) | (sort ? 255u - sched_bucket(len) : 0u) : CLS_NONE;
    if (in_range) w.key[i] = key;
    (void)sched_wave_add(w.bins, has ? key : 0u, has);
    if (has) {
        atomicAdd(&lds_cnt[cls], 1u);
        atomicAdd((unsigned long long *)&lds_work[cls], (unsigned long long)len);
        atomicMax((unsigned long long *)&lds_work[CLS_MAX + cls], (unsigned long long)len);
    }
}
__device__ __forceinline__ void sched_classify_flush(const SchedWs &w, const u32 *lds_cnt, const u64 *lds_work)
{
    if (threadIdx.x < CLS_MAX && lds_cnt[threadIdx.x]) {
        atomicAdd((unsigned long long *)&w.work[threadIdx.x], (unsigned long long)lds_work[threadIdx.x]);
        atomicMax((unsigned long long *)&w.work[CLS_MAX + threadIdx.x], (unsigned long long)lds_work[CLS_MAX + threadIdx.x]);
    }
}
// Wave priority by chain length.  Long and short streams share a SIMD (other workgroups of the class, other classes):
// the issue arbiter serves the wave of higher priority first, so the streams that decide when the batch ends run at
// the pace of a lone wave and the short ones fill the slots they leave (a wave of the chain loops waits for LDS most
// of the time).  `p` is wave-uniform, 0 .. 3.
__device__ __forceinline__ void sched_setprio(u32 p)
{
    if (p >= 3u) __builtin_amdgcn_s_setprio(3);
    else if (p == 2u) __builtin_amdgcn_s_setprio(2);
    else if (p == 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
// `heavy`: the wave stands for a large piece of the CU's LDS (a class of large images: one wave per 50 KB where the
// smallest class has one per 10 KB).  Side by side the classes share a CU by LDS bytes, but its issue slots go to waves:
// without a lift the class of large images gets a fifth of the slots for half of the LDS (its launch took 85 ms where
// its share of the chip promised 37).
__device__ __forceinline__ u32 sched_prio_of(bool active, u32 len, bool heavy = false)
{
    const u32 p = wave_any(active && len >= (1u << 19)) ? 2u : wave_any(active && len >= (1u << 16)) ? 1u : 0u;
    return p + (heavy ? 1u : 0u);
}
// the walk of a workgroup over the shares of its class
struct SchedWalk {
    u32 *cnt;
    int nwg, stride, it;
    bool dyn, in;
    __device__ __forceinline__ SchedWalk(u32 *cnt_, int nmine, int qpw, bool dyn_) : cnt(cnt_), it(0), dyn(dyn_)
    {
        nwg = (nmine + qpw - 1) / qpw;
        const u32 seats = cnt[SCHED_SEATS];
        in = blockIdx.x < seats;
        stride = (int)(seats < gridDim.x ? seats : gridDim.x);
        if (in && nwg > 0 && threadIdx.x == 0) atomicMin(&cnt[SCHED_TSTART], (u32)__builtin_amdgcn_s_memrealtime());
    }
    // a workgroup that worked leaves: the class has run at least until now
    __device__ __forceinline__ void leave() const
    {
        if (in && nwg > 0 && threadIdx.x == 0) atomicMax(&cnt[SCHED_TEND], (u32)__builtin_amdgcn_s_memrealtime());
    }
    // next share of a ONE-WAVE workgroup (-1: none)
    __device__ __forceinline__ int next_wave()
    {
        if (!in) return -1;
        int wg;
        if (dyn) {
            u32 v = 0;
            if ((threadIdx.x & (WAVE - 1)) == 0) v = atomicAdd(&cnt[SCHED_CLAIM], 1u);
            wg = __builtin_amdgcn_readfirstlane((int)v);
        } else wg = (int)blockIdx.x + it * stride;
        it++;
        return wg < nwg ? wg : -1;
    }
    // next share of a workgroup of several waves: thread 0 claims, one LDS dword carries it (two barriers)
    __device__ __forceinline__ int next_wg(volatile u32 *slot)
    {
        if (!in) return -1;
        int wg;
        if (dyn) {
            if (threadIdx.x == 0) *slot = atomicAdd(&cnt[SCHED_CLAIM], 1u);
            __syncthreads();
            wg = (int)*slot;
            __syncthreads();
        } else wg = (int)blockIdx.x + it * stride;
        it++;
        return wg < nwg ? wg : -1;
    }
};
#endif

// How fast a class runs on a full chip, in "streams at a lone wave's pace".  Resident streams are not it: a CU issues for
// four waves at a time, and the fourteen 16-stream waves a CU holds of the smallest class each get a quarter of a SIMD's
// slots, not a whole one (order-0 q40: 409 GB/s with 224 streams per CU against 239 GB/s for the packed rows' 45) -
// further waves on a SIMD only fill the slots the others leave while they wait for LDS, about a quarter each.
static inline float sched_rate(int qpw, int waves_per_wg, int wgs_per_cu, int cus)
{
    const float waves = (float)waves_per_wg * (float)wgs_per_cu;                  // per CU
    const float per_wave = (float)qpw / (float)waves_per_wg;
    const float eff = waves <= 4.f ? waves : 4.f * (1.f + 0.25f * (waves / 4.f - 1.f));
    return per_wave * eff * (float)cus;
}
// What the last batch of a context looked like, as a hint for the next one: per class the sum of the chain lengths and
// the longest chain, copied to pinned host memory behind the chain kernels (no synchronisation - the next call reads
// whatever has arrived).  The host deals the class launches out over the streams with it.  The plan's shares make all
// streams end together whatever the deal - except that a class cannot end before its longest chain has run, at a lone
// wave's pace (1 MiB: ~50 ms to decode): two such classes behind each other on one stream are two such latencies.  So
// the classes go out longest first - by max(longest chain, work per resident stream) - each to the stream with the
// least so far.  Without a hint (first call) the launches are dealt out in turn.
struct SchedHint {
    u64 *work;
    // what the classes' launches really took, side by side, as rates in the plan's units (streams at a lone wave's pace),
    // smoothed over the batches of this context; 0 = not learned.  `pace`: bytes per 10 ns tick of one stream alone.
    float learned[CLS_MAX];
    float pace;
    bool learn;              // option sched_learn: bit 0 the encoder's classes, bit 1 the decoder's
    bool side_by_side;       // how the last batch's launches were dealt out (rates are learned from side-by-side runs only)
};                // pinned, [2 * CLS_MAX]: sums, then longest - and behind them the SCHED_CNT_WORDS
                                                // dwords of SchedWs.cnt (counts, seats: for the trace); nullptr: no hint kept
#define SCHED_HINT_BYTES (2 * CLS_MAX * sizeof(u64) + SCHED_CNT_WORDS * sizeof(u32))
// launch_order: the order in which the launches should go out - the classes the last batch used first (an EMPTY class's
// launch still has to get its workgroups through the dispatcher, which on a chip full of seated persistent workgroups
// takes until LDS frees up: 18 ms were seen - anything queued behind it on its stream waits that long)
void sched_assign_queues(SchedPlan &plan, const int *todo_cls, int ntodo, int nq, SchedHint *hint, u8 *queue_of_todo,
                         int *launch_order, const char *trace = nullptr);

extern "C" {
void r4x16_sched_hint_save(const SchedWs *w, SchedHint *hint, hipStream_t s);
void r4x16_voff_scan(u64 *v, int n, hipStream_t s);
void r4x16_sched_zero(const SchedWs *w, hipStream_t s);
void r4x16_sched_group(const SchedWs *w, int nitems, const SchedPlan *plan, hipStream_t s);
void r4x16_sched_launch(const void *kernel, dim3 grid, dim3 block, void **args, size_t lds, hipStream_t s);
}
