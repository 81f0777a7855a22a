// r4x16_sched.hip - the device-side grouping, ordering and launch plan of the chain kernels' streams (r4x16_sched.h).
#include "r4x16_sched.h"
#include <algorithm>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_sched_zero(SchedWs w)
{
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < 2u * SCHED_BINS) w.bins[i] = 0u;
    if (i < SCHED_CNT_WORDS) w.cnt[i] = (i >= SCHED_TSTART && i < SCHED_TEND) ? 0xffffffffu : 0u;
    if (i < 2u * CLS_MAX) w.work[i] = 0ull;
}

// One workgroup: exclusive scan of the (class, bucket) bins -> where each bin's streams start in the list; per-class
// counts and starts; and the plan - how many workgroups of each class's launch may work.
//   A class alone on the chip needs  t_c = work_c / (qpw_c x wgs_full_c)  (bytes per resident stream: the classes' step
//   times are within a factor of two of each other, the counts of resident streams differ by a factor of fifty).  The
//   launches of one stream run one after the other, the streams side by side: stream q gets the fraction
//   T_q / sum(T) of the chip, T_q = the sum of its classes' t_c, and each of its classes that fraction of wgs_full_c
//   while it runs - all streams then end together.  If every stream's largest class fits beside the others' as a
//   whole, nothing is rationed.
__global__ __launch_bounds__(1024) void k_sched_scan(SchedWs w, SchedPlan plan)
{
    __shared__ u32 part[1024];
    __shared__ float tq[CLS_MAX], fillq[CLS_MAX];
    const u32 t = threadIdx.x;
    constexpr u32 PER = SCHED_BINS / 1024u;                  // bins per thread (16): a class is 16 threads
    u32 v[PER], sum = 0;
#pragma unroll
    for (u32 j = 0; j < PER; j++) { v[j] = w.bins[t * PER + j]; sum += v[j]; }
    part[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {                    // inclusive scan, 10 steps
        const u32 add = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    u32 at = part[t] - sum;
#pragma unroll
    for (u32 j = 0; j < PER; j++) { w.bins[t * PER + j] = at; at += v[j]; }
    constexpr u32 TPC = SCHED_NB / PER;                      // threads per class
    if (t % TPC == 0) {
        const u32 c = t / TPC;
        const u32 first = part[t] - sum, last = part[t + TPC - 1];
        w.cnt[SCHED_COUNT + c] = last - first;
        w.cnt[SCHED_START + c] = first;
    }
    if (t < CLS_MAX) { tq[t] = 0.f; fillq[t] = 0.f; }
    __syncthreads();
    if (t == 0 && plan.concurrent == 1) {
        for (u32 c = 0; c < plan.ncls; c++) {
            const u32 n = w.cnt[SCHED_COUNT + c], q = plan.queue[c];
            if (!n || !plan.wgs_full[c] || q >= CLS_MAX) continue;
            tq[q] += (float)w.work[c] / plan.rate[c];
            const float fill = (float)((n + plan.qpw[c] - 1u) / plan.qpw[c]) / (float)plan.wgs_full[c];
            if (fill > fillq[q]) fillq[q] = fill;
        }
    }
    __syncthreads();
    if (t < CLS_MAX) {
        const u32 n = w.cnt[SCHED_COUNT + t];
        const u32 qpw = t < plan.ncls ? plan.qpw[t] : 16u, full = t < plan.ncls ? plan.wgs_full[t] : 0u;
        const u32 q = t < plan.ncls ? plan.queue[t] : 0xffu;
        u32 seats = full ? full : 0xffffffffu;               // (classes the host gave no figures for: every workgroup works)
        if (plan.concurrent == 1 && n && full && q < CLS_MAX) {
            float tot = 0.f, fills = 0.f;
            for (u32 c = 0; c < CLS_MAX; c++) { tot += tq[c]; fills += fillq[c]; }
            const u32 want = (n + qpw - 1u) / qpw;
            u32 s = want;
            if (fills > 1.f && tot > 0.f) {
                // (a stream-queue with a small share - the queue of the batch's small classes, one after the other - gets
                //  1.6 x its share: its classes then end well before the large ones instead of beside them, where their
                //  swing - a dozen workgroups among thousands - decided when the batch ends.  Heterogeneous 16 GiB batch,
                //  medians of six to eight passes, four alternations: encode 111 -> 103 ms, decode 91.3 -> 89.6)
                float share = tq[q] / tot;
                if (share < 0.2f) share *= 1.6f;
                s = (u32)ceilf(share * (float)full);
                if (s < 1u) s = 1u;
                if (s > want) s = want;
            }
            if (s > full) s = full;
            seats = s;
        }
        w.cnt[SCHED_SEATS + t] = n ? seats : 0u;
    }
}

__global__ __launch_bounds__(256) void k_sched_scatter(SchedWs w, int nitems)
{
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    const u32 key = i < nitems ? w.key[i] : CLS_NONE;
    const bool has = key != CLS_NONE;
    const u32 rank = sched_wave_add(w.bins + SCHED_BINS, has ? key : 0u, has);
    if (has) w.list[w.bins[key] + rank] = (u32)i;
}

extern "C" void r4x16_sched_zero(const SchedWs *w, hipStream_t s)
{
    hipLaunchKernelGGL(k_sched_zero, dim3((2u * SCHED_BINS + 255u) / 256u), dim3(256), 0, s, *w);
}
extern "C" void r4x16_sched_group(const SchedWs *w, int nitems, const SchedPlan *plan, hipStream_t s)
{
    hipLaunchKernelGGL(k_sched_scan, dim3(1), dim3(1024), 0, s, *w, *plan);
    hipLaunchKernelGGL(k_sched_scatter, dim3((nitems + 255) / 256), dim3(256), 0, s, *w, nitems);
}
extern "C" void r4x16_sched_launch(const void *kernel, dim3 grid, dim3 block, void **args, size_t lds, hipStream_t s)
{
    (void)hipLaunchKernel(kernel, grid, block, args, lds, s);
}

void sched_assign_queues(SchedPlan &plan, const int *todo_cls, int ntodo, int nq, SchedHint *hint, u8 *queue_of_todo,
                         int *launch_order, const char *trace)
{
    double load[R4_FORK_STREAMS + 1] = {0}, floor_q[R4_FORK_STREAMS + 1] = {0};
    double t[CLS_MAX], tl[CLS_MAX], fl[CLS_MAX];
    int order[CLS_MAX];
    bool any = false;
    double sum_tl = 0.0, seq = 0.0;
    const u32 *cnt = hint && hint->work ? (const u32 *)(hint->work + 2 * CLS_MAX) : nullptr;
    float model_rate[CLS_MAX];
    for (u32 c = 0; c < CLS_MAX; c++) model_rate[c] = plan.rate[c];
    for (int k = 0; k < ntodo; k++) {
        const int c = todo_cls[k];
        const double w = hint && hint->work ? (double)hint->work[c] : 0.0;
        fl[k] = hint && hint->work ? (double)hint->work[CLS_MAX + c] : 0.0;               // the longest chain: the class cannot end before it
        t[k] = plan.wgs_full[c] && plan.rate[c] > 0.f ? w / (double)plan.rate[c] : 0.0;   // the class alone on the chip, throughput-bound (model)
        // What the class's launch really took last time, side by side with the others: work / (duration x its share of
        // the chip) is its rate as it runs in company - where it was bound by throughput, not by its longest chain.
        // Smoothed over the batches of the context.
        if (cnt && hint->learn && hint->side_by_side && w > 0.0 && plan.wgs_full[c] && cnt[SCHED_SEATS + c] && hint->pace > 0.f) {
            const u32 t0 = cnt[SCHED_TSTART + c], t1 = cnt[SCHED_TEND + c], seats = cnt[SCHED_SEATS + c] < plan.wgs_full[c] ? cnt[SCHED_SEATS + c] : plan.wgs_full[c];
            const double ticks = (double)(u32)(t1 - t0), frac = (double)seats / plan.wgs_full[c];
            // (bound by throughput: it took clearly longer than its longest chain takes alone)
            if (t0 != 0xffffffffu && ticks > 1000.0 && ticks > 1.5 * fl[k] / hint->pace) {
                // (within 0.4 .. 1.25 of the model: a class that looks slower than that is bound by its longest chains running
                //  in a crowd, not by throughput, and more seats would only crowd it further - seen on the encoder's
                //  class of 64-stream workgroups, which took 85 ms with 263 seats and 68 with 98)
                float r = (float)(w / (ticks * frac) / hint->pace);
                const float lo = 0.4f * model_rate[c], hi = 1.25f * model_rate[c];
                r = r < lo ? lo : r > hi ? hi : r;
                hint->learned[c] = hint->learned[c] > 0.f ? 0.5f * hint->learned[c] + 0.5f * r : r;
            }
        }
        tl[k] = hint && hint->learned[c] > 0.f ? w / (double)hint->learned[c] : 1.3 * t[k];  // side by side (1.3: the model's allowance where nothing is learned)
        if (hint && hint->learned[c] > 0.f) plan.rate[c] = hint->learned[c];                 // the plan's shares follow what was learned
        sum_tl += tl[k];
        seq += fl[k] > t[k] ? fl[k] : t[k];
        if (t[k] > 0.0 || fl[k] > 0.0) any = true;
        order[k] = k;
    }
    for (int k = 0; k < ntodo; k++) launch_order[k] = k;
    if (hint) hint->side_by_side = nq > 1;
    if (nq <= 1 || !any) { for (int k = 0; k < ntodo; k++) queue_of_todo[k] = (u8)(nq > 1 ? k % nq : 0); return; }
    auto key = [&](int k) { return fl[k] > tl[k] ? fl[k] : tl[k]; };
    std::stable_sort(order, order + ntodo, [&](int a, int b) { return key(a) > key(b); });
    int rr = 0;
    for (int j = 0; j < ntodo; j++) {
        const int k = order[j];
        launch_order[j] = k;                             // (sorted: the used classes come first, the longest first)
        int q = 0;
        if (key(k) > 0.0) { for (int i = 1; i < nq; i++) if (load[i] < load[q]) q = i; load[q] += key(k); floor_q[q] += fl[k]; }
        else {
            // Classes the last batch did not use: in turn BEHIND the used classes, on their streams only.  On a stream of its
            // own an empty launch starts together with the used ones, and where its workgroups happen to be placed first
            // (they ask for a whole CU's LDS and leave at once) the dispatcher gives a used class's workgroup a seat beside
            // another one instead: 4,096 x 1 MiB blocks encoded in 26.5 or 38.6 ms, by the process (gpurun_out/r04_bim_kt*).
            int usedq[R4_FORK_STREAMS + 1], nu = 0;
            for (int i = 0; i < nq; i++) if (load[i] > 0.0) usedq[nu++] = i;
            q = nu ? usedq[rr++ % nu] : rr++ % nq;
        }
        queue_of_todo[k] = (u8)q;
    }
    // Side by side or one after the other?  Side by side hides the classes' chain latencies behind each other but costs
    // throughput (workgroups of several sizes share a CU's LDS and issue slots badly: 262,144 x 64 KiB mixed blocks
    // took 30 % longer that way); one after the other every class has the chip to itself but pays its own longest
    // chain.  With the last batch's figures both can be priced: whichever is shorter.
    double conc = sum_tl;
    for (int i = 0; i < nq; i++) if (floor_q[i] > conc) conc = floor_q[i];
    const bool in_order = seq <= conc;
    if (in_order) for (int k = 0; k < ntodo; k++) queue_of_todo[k] = 0;
    if (hint) hint->side_by_side = !in_order;
    if (trace) {                                         // option sched_trace: what the LAST batch looked like, and this deal
        fprintf(stderr, "rans4x16_hip sched %s: last batch seq %.0f conc %.0f (sum of side-by-side times %.0f) -> %s\n", trace, seq, conc, sum_tl, in_order ? "in stream order" : "side by side");
        for (int k = 0; k < ntodo; k++) {
            const int c = todo_cls[k];
            if (t[k] > 0.0 || fl[k] > 0.0)
                fprintf(stderr, "  class %2d qpw %2u full %4u rate %7.0f (learned %7.0f): streams %6u seats %4u took %8.3f ms work %12llu longest %8.0f t %9.0f -> stream %u\n", c,
                        plan.qpw[c], plan.wgs_full[c], plan.rate[c], hint->learned[c], cnt[SCHED_COUNT + c], cnt[SCHED_SEATS + c],
                        (double)(u32)(cnt[SCHED_TEND + c] - cnt[SCHED_TSTART + c]) / 1e5, (unsigned long long)hint->work[c], fl[k], t[k], queue_of_todo[k]);
        }
    }
}
extern "C" void r4x16_sched_hint_save(const SchedWs *w, SchedHint *hint, hipStream_t s)
{
    if (hint && hint->work) {
        (void)hipMemcpyAsync(hint->work, w->work, 2 * CLS_MAX * sizeof(u64), hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(hint->work + 2 * CLS_MAX, w->cnt, SCHED_CNT_WORDS * sizeof(u32), hipMemcpyDeviceToHost, s);
    }
}

// Exclusive prefix sum, in place, of the n u64 entries of v (the sizes of the blocks' staging regions, written by a
// kernel of the encoder / decoder); the total goes to v[n].  One workgroup, tiles of 1,024 entries with a carry.
__global__ __launch_bounds__(1024) void k_voff_scan(u64 *v, int n)
{
    __shared__ u64 part[1024];
    __shared__ u64 carry;
    const u32 t = threadIdx.x;
    if (t == 0) carry = 0ull;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + (int)t;
        const u64 mine = i < n ? v[i] : 0ull;
        part[t] = mine;
        __syncthreads();
        for (u32 d = 1; d < 1024u; d <<= 1) {
            const u64 add = t >= d ? part[t - d] : 0ull;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        const u64 c = carry;
        if (i < n) v[i] = c + part[t] - mine;
        __syncthreads();
        if (t == 1023) carry = c + part[1023];
        __syncthreads();
    }
    if (t == 0) v[n] = carry;
}
extern "C" void r4x16_voff_scan(u64 *v, int n, hipStream_t s) { hipLaunchKernelGGL(k_voff_scan, dim3(1), dim3(1024), 0, s, v, n); }
