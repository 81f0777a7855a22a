// =============================================================================================
// r4x16_enc_chain.hip - launcher of k_enc_chain (r4x16_enc_chain.h), the LDS size classes, and the instantiations for
// u16 images.  The packed-row instantiation lives in r4x16_enc_chain_pk.hip.
// =============================================================================================
#include "r4x16_enc_chain.h"

// r4x16_enc_chain_pk.hip: k_enc_chain<true, true>
extern "C" void r4x16_enc_chain_pk_lds_limit(int bytes);
extern "C" const void *r4x16_enc_chain_pk_kernel(void);

// ---- host-callable launcher ----------------------------------------------------------------------
extern "C" bool r4x16_first_on_device(u32 bit);                                          // r4x16_decode.hip
// {LDS bytes per stream, streams per wave}; LDS is allocated in 1,280-byte granules.
// q4/q8 images are ~0.4 KB, an order-0 row 0.8 KB, q40 4.6 KB (16 x 4,800 = 60 granules: 2 waves, 32 streams per CU — fuller waves measured faster than more waves)
// LDS size classes: bytes per stream (image + word ring).  A workgroup takes as many streams as
// fit beside the shared reciprocal table, up to 64 (four waves); 1,280-byte allocation granules.
// (sizes are 16 mod 128: consecutive streams start four LDS banks apart, so that the eight streams of a
// 32-lane access group do not all hit the same bank when they touch the same offset)
// (round 3: the steps above 12,816 were 33,296 and 73,616 - the 13 KB image of a 79-symbol alphabet, what X_PACK|X_RLE
//  makes of four-letter quality values, sat in the 33 KB class at four streams per CU: 11.8 ms for 4,096 such streams.
//  Now every count of streams per CU from nine to one has its class: 147,344 bytes beside the reciprocal table / k.)
static const u32 ENC_CLASSES[] = {656, 1296, 2576, 4752, 6416, 12816, 16272, 24464, 36752, 49040, 73616, 147344};
// packed rows (20..64 symbols, 10-bit tables): 46 symbols need 3,532 bytes -> 45 streams per CU beside the small
// reciprocal table (3,536 is 80 mod 128: consecutive streams start 20 banks apart)
static const u32 ENC_PK_CLASSES[] = {1168, 2064, 2832, 3536, 3728, 4752, 6416};
// symbol records (kind 2, r4x16_enc_chain_rec.hip; only batches that leave LDS to spare make such images): one wave
// per workgroup, {LDS bytes per stream, streams per wave}; four workgroups per CU, then fewer
struct EncRecClass { u32 bytes; int qpw; };
static const EncRecClass ENC_REC_CLASSES[] = {
    {2576, 15}, {4112, 9}, {8080, 5}, {13584, 3}, {20368, 2}, {32000, 1}, {40960, 1}, {53760, 1}, {81920, 1}, {163840, 1},
};
#define ENC_REC_NCLS ((u32)(sizeof(ENC_REC_CLASSES) / sizeof(ENC_REC_CLASSES[0])))
extern "C" void r4x16_enc_chain_rec_lds_limit(int bytes);
extern "C" const void *r4x16_enc_chain_rec_kernel(void);
#define ENC_NCLS    ((u32)(sizeof(ENC_CLASSES) / sizeof(ENC_CLASSES[0])))
#define ENC_PK_NCLS ((u32)(sizeof(ENC_PK_CLASSES) / sizeof(ENC_PK_CLASSES[0])))
static int enc_class_qpw(u32 bytes, bool pk, const R4Opts *o)
{
    const u32 room = 163840u - (pk ? ENC_LRCP_PK_BYTES : ENC_LRCP_BYTES);
    const u32 fit = room / bytes;
    const int cap = (int)o->v[OPT_ENC_QPW_CAP];                  // tuning aid (default 64)
    return (int)(fit > (u32)cap ? (u32)cap : fit);
}
extern "C" int r4x16_resident_grid(size_t lds_bytes, int waves_per_wg, int wanted);      // r4x16_decode.hip
extern "C" int r4x16_cu_count(void);
struct EncClassTab { u32 n; u32 sort; u32 bytes[CLS_MAX]; u32 pk[CLS_MAX]; };     // classes: u16 images, then packed ones, then records
__global__ __launch_bounds__(256) void k_enc_classify(const EncItem *items, int nitems, EncClassTab tab, SchedWs sw)
{
    __shared__ u32 local[CLS_MAX];
    __shared__ u64 lwork[2 * CLS_MAX];
    if (threadIdx.x < CLS_MAX) { local[threadIdx.x] = 0; lwork[threadIdx.x] = 0ull; lwork[CLS_MAX + threadIdx.x] = 0ull; }
    __syncthreads();
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    u32 c = CLS_NONE, len = 0;
    if (i < nitems && items[i].active) {
        // kinds: 0 u16 rows, 1 packed rows, 2 symbol records; the order-0 streams of kinds 0 and 2 have classes of their
        // own (kinds 3 and 4): the kernels run a wave's order-1 streams, then its order-0 streams, and a class that holds
        // both kinds - small order-1 alphabets make images as small as a one-row image - would pay both latencies
        const u32 need = items[i].img_bytes + ENC_RING_BYTES, raw = items[i].packed,
                  pk = items[i].order != 0 ? raw : raw == 0u ? 3u : raw == 2u ? 4u : raw;
        c = tab.n;                                         // images too large for LDS (never a packed one)
        for (u32 k = 0; k < tab.n; k++) if (tab.pk[k] == pk && need <= tab.bytes[k]) { c = k; break; }
        len = items[i].n;
    }
    sched_classify(sw, i, i < nitems, c, len, tab.sort != 0, local, lwork);
    __syncthreads();
    sched_classify_flush(sw, local, lwork);
}
// the shape of a class's launch: streams per workgroup, waves, streams per wave, LDS bytes
struct EncShape { int qpw, waves, spw; size_t ldsb; u32 bytes; };
#define ENC_O0_ROWS_BYTES 1296u      // order-0 streams with u16 rows: a 770-byte image and the ring
#define ENC_O0_REC_BYTES  8080u      // order-0 streams with symbol records: a 4,352-byte image and the ring
#define ENC_O0_REC_QPW    5
static EncShape enc_rows_shape(u32 cls, int nitems, const R4Opts *o, u32 bytes_o0 = 0)
{
    const bool pk = !bytes_o0 && cls >= ENC_NCLS;
    const u32 bytes = bytes_o0 ? bytes_o0 : pk ? ENC_PK_CLASSES[cls - ENC_NCLS] : ENC_CLASSES[cls];
    const u32 tuned = pk ? 3536u : 4752u;                // the class of the 46-symbol quality tables
    const int force_qpw = (int)o->v[OPT_ENC_QPW], force_waves = (int)o->v[OPT_ENC_WAVES];   // tuning aids
    int qpw = (force_qpw && bytes == tuned) ? force_qpw : enc_class_qpw(bytes, pk, o);
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
    EncShape sh;
    sh.qpw = qpw; sh.waves = waves; sh.spw = (qpw + waves - 1) / waves; sh.bytes = bytes;
    sh.ldsb = (size_t)(pk ? ENC_LRCP_PK_BYTES : ENC_LRCP_BYTES) + (size_t)qpw * bytes;
    return sh;
}
static EncShape enc_rec_shape(u32 r, const R4Opts *o, bool o0 = false)
{
    const EncRecClass c0 = {ENC_O0_REC_BYTES, ENC_O0_REC_QPW};
    const EncRecClass &c = o0 ? c0 : ENC_REC_CLASSES[r];
    const int force_rec = (int)o->v[OPT_ENC_QPW_REC];     // tuning aid
    EncShape sh;
    sh.qpw = (force_rec > 0 && c.qpw > force_rec) ? force_rec : c.qpw;
    sh.waves = 1; sh.spw = sh.qpw; sh.bytes = c.bytes;
    sh.ldsb = (size_t)sh.qpw * c.bytes;
    return sh;
}
static int enc_wgs_per_cu(const EncShape &sh)
{
    const long granules = ((long)sh.ldsb + 1279) / 1280;      // LDS is allocated in 1,280-byte granules
    long wgs = granules ? 128 / granules : 32;
    if (wgs * sh.waves > 32) wgs = 32 / sh.waves;
    return wgs < 1 ? 1 : (int)wgs;
}
extern "C" void r4x16_launch_enc_chain(const EncWs *ws, int nitems, hipStream_t s0, const R4Fork *fk, const R4Opts *o, SchedHint *hint)
{
    // classes side by side over the caller's stream and the side streams, each with its stream's share of the chip
    // (launch_dec_chain_of, r4x16_sched.h); class index ci = position in the classify table: u16 classes, packed
    // classes, record classes
    const int nq = fk ? fk->n + 1 : 1;
    struct Launch { const void *kern; int grid; EncShape sh; u32 ci; };
    Launch todo[CLS_MAX];
    int ntodo = 0;
    EncClassTab tab;
    tab.n = 0;
    tab.sort = o->v[OPT_SCHED_SORT] != 0;
    SchedPlan plan;
    plan.concurrent = nq > 1 ? (u32)o->v[OPT_SCHED_CONCURRENT] : 0u; plan.claim = o->v[OPT_SCHED_CLAIM] != 0; plan.pad = 0;
    for (u32 ci = 0; ci < CLS_MAX; ci++) { plan.qpw[ci] = 16; plan.wgs_full[ci] = 0; plan.queue[ci] = 0xff; plan.rate[ci] = 0.f; }
    const int cus = r4x16_cu_count();
    auto add = [&](u32 pk, u32 bytes, const EncShape &sh, const void *kern) {
        const u32 ci = tab.n++;
        tab.pk[ci] = pk; tab.bytes[ci] = bytes;
        plan.qpw[ci] = (u16)sh.qpw; plan.wgs_full[ci] = (u16)(cus * enc_wgs_per_cu(sh));
        plan.rate[ci] = sched_rate(sh.qpw, sh.waves, enc_wgs_per_cu(sh), cus);
        if (!kern) return;
        todo[ntodo++] = Launch{kern, r4x16_resident_grid(sh.ldsb, sh.waves, (nitems + sh.qpw - 1) / sh.qpw), sh, ci};
    };
    for (u32 k = 0; k < ENC_NCLS; k++) add(0, ENC_CLASSES[k], enc_rows_shape(k, nitems, o), (const void *)k_enc_chain<true, false>);
    for (u32 k = 0; k < ENC_PK_NCLS; k++) add(1, ENC_PK_CLASSES[k], enc_rows_shape(ENC_NCLS + k, nitems, o), r4x16_enc_chain_pk_kernel());
    // (record classes: only batches that leave LDS to spare make such images)
    for (u32 k = 0; k < ENC_REC_NCLS; k++) add(2, ENC_REC_CLASSES[k].bytes, enc_rec_shape(k, o), ws->direct_budget ? r4x16_enc_chain_rec_kernel() : nullptr);
    add(3, ENC_O0_ROWS_BYTES, enc_rows_shape(0, nitems, o, ENC_O0_ROWS_BYTES), (const void *)k_enc_chain<true, false>);
    add(4, ENC_O0_REC_BYTES, enc_rec_shape(0, o, true), ws->direct_budget ? r4x16_enc_chain_rec_kernel() : nullptr);
    plan.ncls = tab.n;
    u8 qof[CLS_MAX];
    int lorder[CLS_MAX];
    {
        int cls_of[CLS_MAX];
        for (int k = 0; k < ntodo; k++) cls_of[k] = (int)todo[k].ci;
        if (hint) hint->learn = (o->v[OPT_SCHED_LEARN] & 1) != 0;
        sched_assign_queues(plan, cls_of, ntodo, nq, hint, qof, lorder, (hint && hint->work && o->v[OPT_SCHED_TRACE]) ? "encode" : nullptr);
        for (int k = 0; k < ntodo; k++) plan.queue[todo[k].ci] = qof[k];
    }
    r4x16_sched_zero(&ws->sched, s0);
    hipLaunchKernelGGL(k_enc_classify, dim3((nitems + 255) / 256), dim3(256), 0, s0, (const EncItem *)ws->items, nitems, tab, ws->sched);
    r4x16_sched_group(&ws->sched, nitems, &plan, s0);
    if (r4x16_first_on_device(4u)) {
        (void)hipFuncSetAttribute((const void *)k_enc_chain<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        r4x16_enc_chain_pk_lds_limit(163840);
        r4x16_enc_chain_rec_lds_limit(163840);
    }
    const int dyn = o->v[OPT_SCHED_CLAIM] != 0;
    auto go = [&](const Launch &L, hipStream_t s) {
        EncItem *items = ws->items;
        const u32 *rcptab = ws->rcptab;
        u8 *dump = ws->dump;
        const u32 *list = ws->sched.list;
        u32 *cnt = ws->sched.cnt + L.ci;
        int qpw = L.sh.qpw, spw = L.sh.spw, dyn_ = dyn;
        u32 bytes = L.sh.bytes;
        void *args[] = {(void *)&items, (void *)&rcptab, (void *)&dump, (void *)&list, (void *)&cnt, (void *)&qpw, (void *)&spw, (void *)&bytes, (void *)&dyn_};
        r4x16_sched_launch(L.kern, dim3(L.grid), dim3(WAVE * L.sh.waves), args, L.sh.ldsb, s);
    };
    unsigned used = 0;
    for (int k = 0; k < ntodo; k++) used |= 1u << (qof[k] % (unsigned)nq);
    if (fk) fk->begin(s0, used);
    for (int j = 0; j < ntodo; j++) { const int k = lorder[j]; go(todo[k], fk ? fk->pick(s0, (unsigned)qof[k]) : s0); }
    if (fk) { fk->end(s0, used); r4x16_sched_hint_save(&ws->sched, hint, s0); }
    EncShape sh;
    sh.qpw = 16; sh.waves = 1; sh.spw = 16; sh.bytes = 0u; sh.ldsb = 0;
    go(Launch{(const void *)k_enc_chain<false, false>, (nitems + 15) / 16, sh, tab.n}, s0);     // images too large for LDS
}
// LDS bytes a stream may spend on symbol records when `nblk` streams are to be resident at once: the largest record
// class that still holds the batch in one round of the chip (0: none).  R4X16_ENC_DIRECT=0 never; =N up to N rounds.
extern "C" u32 r4x16_enc_direct_budget(int nblk, const R4Opts *o)
{
    const int rounds = (int)o->v[OPT_ENC_DIRECT];
    if (rounds <= 0 || nblk <= 0) return 0u;
    const long cus = r4x16_cu_count();
    const long per_cu = (nblk + cus * rounds - 1) / (cus * rounds);
    u32 best = 0;
    for (const auto &c : ENC_REC_CLASSES) {
        const long granules = ((long)c.qpw * c.bytes + 1279) / 1280;      // LDS is allocated in 1,280-byte granules
        long wgs = 128 / granules;
        if (wgs > 32) wgs = 32;
        if (wgs * c.qpw >= per_cu && c.bytes > best) best = c.bytes;
    }
    return best;
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
        const int qpw = enc_class_qpw(bytes, pk, r4x16_opts_defaults());
        int waves = (qpw + 7) / 8;
        if (waves > 4) waves = 4;
        *streams_per_wave = (qpw + waves - 1) / waves;
        *waves_per_cu = waves;                               // one workgroup per CU
        return qpw;
    }
    *streams_per_wave = 16; *waves_per_cu = 8;
    return 128;
}

// ---------------------------------------------------------------------------------------------
// rANS 4x8 (include/rans4x8_hip.h): the chain kernel for images that fit LDS.  Round 4: the 4x16 encoder's software
// pipeline (chain_encode_o1_lds / chain_encode_o0_pipe: input pieces, compact indices, cumulative pairs and reciprocals
// fetched trips ahead, emitted bytes through the LDS ring, 16-byte stores) with rANS 4x8's byte renormalisation
// (EncOutT<true>: up to two bytes per chain and step, rANS_byte.h:320-402) instead of round 2's step-at-a-time loop -
// the streams, the quarter maps and the u16 cumulative image are the same for both codecs.  A workgroup of up to four
// waves shares one LDS copy of the reciprocal table; qpw streams per workgroup, spw per wave.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k8_enc_chain_pipe(EncItem *items, const u32 *rcptab_, u8 *dump_, const u32 *list, const u32 *count,
                                                         u32 slot_bytes, int qpw, int spw)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 tid = threadIdx.x, lane = tid & (WAVE - 1), wq = lane >> 2;
    const u32 quad = (tid >> 6) * (u32)spw + wq;                 // stream slot inside the workgroup
    const int nmine = (int)count[0];
    list += count[CLS_MAX];
    if ((int)blockIdx.x * qpw >= nmine) return;
    const int slot = (int)blockIdx.x * qpw + (int)quad;
    const bool mine = wq < (u32)spw && quad < (u32)qpw && slot < nmine;
    EncItem *I = &items[list[mine ? slot : (int)blockIdx.x * qpw]];
    const bool active = mine && I->active;
    gcu32 *rcptab = to_global(rcptab_);
    gcu8 *data = (gcu8 *)I->data;
    gu8 *send = (gu8 *)I->scratch_end;
    const u32 n = I->n, ns = I->ns, order = active ? I->order : 2u;
    u32 *lrcp = (u32 *)lds;
    u8 *slots = lds + ENC_LRCP_BYTES;
    for (u32 j = tid; j < RCPTAB_ENTRIES; j += blockDim.x) lrcp[j] = rcptab[j];
    // each wave copies the images of its own quads (16-byte pieces)
    const u64 my_img = active ? I->image : 0ull;
    const u32 nbytes = active ? ENC_IMG_IDX + (order ? 2u * ns * (ns + 1u) : 2u * 257u) : 0u;
    const u32 wq0 = (tid >> 6) * (u32)spw;
    for (int qd = 0; qd < 16; qd++) {
        const u64 src = __shfl(my_img, qd * 4);
        const u32 nb = __shfl(nbytes, qd * 4);
        if (!src) continue;
        gcu32x4 *sp = (gcu32x4 *)src;
        u32x4 *dd = (u32x4 *)(slots + (u64)(wq0 + qd) * slot_bytes);
        for (u32 j = lane; j < ((nb + 15) >> 4); j += WAVE) dd[j] = sp[j];
    }
    __syncthreads();
    const u32 sl = active ? quad : 0u;                         // (lanes without a stream read and write the LDS of stream 0)
    const u8 *im = slots + (u64)sl * slot_bytes;
    u8 *ring = slots + (u64)sl * slot_bytes + (slot_bytes - ENC_RING_BYTES);
    gu8 *dump = to_global(dump_) + 16u * ((blockIdx.x * blockDim.x + tid) & (ENC_DUMP_BYTES / 16u - 1u));
    u32 pay = chain_encode_o1_lds<false, true>(im, ring, lrcp, data, n, ns, 12u, (gcu8 *)rcptab, send, dump, order == 1, lane);
    pay |= chain_encode_o0_pipe<true>(im, ring, lrcp, data, n, 12u, (gcu8 *)rcptab, send, dump, order == 0, lane);
    if (active && (lane & 3) == 0) I->pay_len = pay;
}
extern "C" void r4x8_enc_chain_launch(EncItem *items, const u32 *rcptab, u8 *dump, const u32 *list, const u32 *count, int nblk, u32 slot_bytes,
                                      int qpw, int spw, hipStream_t s)
{
    if (r4x16_first_on_device(32u))
        (void)hipFuncSetAttribute((const void *)k8_enc_chain_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int waves = (qpw + spw - 1) / spw;
    hipLaunchKernelGGL(k8_enc_chain_pipe, dim3((nblk + qpw - 1) / qpw), dim3(WAVE * waves), ENC_LRCP_BYTES + (size_t)qpw * slot_bytes, s,
                       items, rcptab, dump, list, count, slot_bytes, qpw, spw);
}
