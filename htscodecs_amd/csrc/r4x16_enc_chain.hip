// =============================================================================================
// r4x16_enc_chain.hip - launcher of k_enc_chain (r4x16_enc_chain.h), the LDS size classes, and the instantiations for
// u16 images.  The packed-row instantiation lives in r4x16_enc_chain_pk.hip.
// =============================================================================================
#include "r4x16_enc_chain.h"

// r4x16_enc_chain_pk.hip: k_enc_chain<true, true>
extern "C" void r4x16_enc_chain_pk_lds_limit(int bytes);
extern "C" void r4x16_enc_chain_pk_launch(int grid, int threads, size_t lds, hipStream_t s, EncItem *items, const u32 *rcptab, u8 *dump,
                                          const u32 *list, const u32 *count, int qpw, int spw, u32 lds_per_item);

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
static const struct { u32 bytes; int qpw; } ENC_REC_CLASSES[] = {
    {2576, 15}, {4112, 9}, {8080, 5}, {13584, 3}, {20368, 2}, {32000, 1}, {40960, 1}, {53760, 1}, {81920, 1}, {163840, 1},
};
#define ENC_REC_NCLS ((u32)(sizeof(ENC_REC_CLASSES) / sizeof(ENC_REC_CLASSES[0])))
extern "C" void r4x16_enc_chain_rec_lds_limit(int bytes);
extern "C" void r4x16_enc_chain_rec_launch(int grid, size_t lds, hipStream_t s, EncItem *items, const u32 *safe, u8 *dump,
                                           const u32 *list, const u32 *count, int qpw, u32 lds_per_item);
#define ENC_NCLS    ((u32)(sizeof(ENC_CLASSES) / sizeof(ENC_CLASSES[0])))
#define ENC_PK_NCLS ((u32)(sizeof(ENC_PK_CLASSES) / sizeof(ENC_PK_CLASSES[0])))
static int enc_class_qpw(u32 bytes, bool pk = false)
{
    const u32 room = 163840u - (pk ? ENC_LRCP_PK_BYTES : ENC_LRCP_BYTES);
    const u32 fit = room / bytes;
    static const int cap = getenv("R4X16_ENC_QPW_CAP") ? atoi(getenv("R4X16_ENC_QPW_CAP")) : 64;   // tuning aid
    return (int)(fit > (u32)cap ? (u32)cap : fit);
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
extern "C" void r4x16_launch_enc_chain(const EncWs *ws, int nitems, hipStream_t s0, const R4Fork *fk)
{
    hipStream_t s = s0;
    {
        EncClassTab tab;
        tab.n = 0;
        for (const u32 bytes : ENC_CLASSES) { tab.pk[tab.n] = 0; tab.bytes[tab.n++] = bytes; }
        for (const u32 bytes : ENC_PK_CLASSES) { tab.pk[tab.n] = 1; tab.bytes[tab.n++] = bytes; }
        for (const auto &c : ENC_REC_CLASSES) { tab.pk[tab.n] = 2; tab.bytes[tab.n++] = c.bytes; }
        r4x16_launch_cls_zero(ws->cls_count, s);
        hipLaunchKernelGGL(k_enc_classify, dim3((nitems + 255) / 256), dim3(256), 0, s, (const EncItem *)ws->items, nitems, tab, ws->cls, ws->cls_count);
        r4x16_launch_cls_group(ws->cls, nitems, ws->cls_count, ws->cls_list, s);
    }
    if (r4x16_first_on_device(4u)) {
        (void)hipFuncSetAttribute((const void *)k_enc_chain<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        r4x16_enc_chain_pk_lds_limit(163840);
        r4x16_enc_chain_rec_lds_limit(163840);
    }
    if (fk) fk->begin(s0);                     // a small batch: its classes run side by side (R4Fork, r4x16_dev.h)
    u32 launched = 0;
    static const int force_qpw = getenv("R4X16_ENC_QPW") ? atoi(getenv("R4X16_ENC_QPW")) : 0;   // tuning aids
    static const int force_waves = getenv("R4X16_ENC_WAVES") ? atoi(getenv("R4X16_ENC_WAVES")) : 0;
    // class index ci = position in the classify table: u16 classes, packed classes, record classes
    // (forked: classes of more than FORK_LDS_MAX bytes per workgroup wait for the join and go out in stream order -
    //  an EMPTY grid of workgroups that each ask for most of a CU's LDS competes for CUs with the class that does the
    //  work, see launch_dec_chain_of; pass 0 = the forked classes, pass 1 = the rest after the join)
    constexpr size_t FORK_LDS_MAX = 40960;
    int pass = 0;
    auto launch_rows = [&](u32 cls) {
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
        if (fk && (ldsb > FORK_LDS_MAX) != (pass == 1)) return;
        const int grid = r4x16_resident_grid(ldsb, waves, (nitems + qpw - 1) / qpw);
        if (fk && pass == 0) s = fk->pick(s0, launched++);
        if (pk)
            r4x16_enc_chain_pk_launch(grid, (int)(WAVE * waves), ldsb, s, ws->items, ws->rcptab, ws->dump, (const u32 *)ws->cls_list,
                                      (const u32 *)(ws->cls_count + cls), qpw, spw, bytes);
        else
            hipLaunchKernelGGL((k_enc_chain<true, false>), dim3(grid), dim3(WAVE * waves), ldsb, s,
                               ws->items, ws->rcptab, ws->dump, (const u32 *)ws->cls_list, (const u32 *)(ws->cls_count + cls), qpw, spw, bytes);
    };
    auto launch_rec = [&](u32 r) {
        const auto &c = ENC_REC_CLASSES[r];
        if (!ws->direct_budget) return;                       // (no stream of this batch was given records)
        static const int force_rec = getenv("R4X16_ENC_QPW_REC") ? atoi(getenv("R4X16_ENC_QPW_REC")) : 0;   // tuning aid
        const int qpw = (force_rec > 0 && c.qpw > force_rec) ? force_rec : c.qpw;
        const size_t ldsb = (size_t)qpw * c.bytes;
        if (fk && (ldsb > FORK_LDS_MAX) != (pass == 1)) return;
        const int grid = r4x16_resident_grid(ldsb, 1, (nitems + qpw - 1) / qpw);
        if (fk && pass == 0) s = fk->pick(s0, launched++);
        r4x16_enc_chain_rec_launch(grid, ldsb, s, ws->items, ws->rcptab, ws->dump, (const u32 *)ws->cls_list,
                                   (const u32 *)(ws->cls_count + ENC_NCLS + ENC_PK_NCLS + r), qpw, c.bytes);
    };
    for (pass = 0; pass < (fk ? 2 : 1); pass++) {
        if (fk && pass == 1) { s = s0; fk->end(s0); }
        for (u32 cls = 0; cls < ENC_NCLS + ENC_PK_NCLS; cls++) launch_rows(cls);
        for (u32 r = 0; r < ENC_REC_NCLS; r++) launch_rec(r);
    }
    const u32 ci = ENC_NCLS + ENC_PK_NCLS + ENC_REC_NCLS;
    s = s0;
    const int grid = (nitems + 15) / 16;
    hipLaunchKernelGGL((k_enc_chain<false, false>), dim3(grid), dim3(WAVE), 0, s, ws->items, ws->rcptab, ws->dump,
                       (const u32 *)ws->cls_list, (const u32 *)(ws->cls_count + ci), 16, 16, 0u);
}
// LDS bytes a stream may spend on symbol records when `nblk` streams are to be resident at once: the largest record
// class that still holds the batch in one round of the chip (0: none).  R4X16_ENC_DIRECT=0 never; =N up to N rounds.
extern "C" u32 r4x16_enc_direct_budget(int nblk)
{
    const char *ev = getenv("R4X16_ENC_DIRECT");           // (read per call: the tests switch it between calls)
    const int rounds = ev && *ev ? atoi(ev) : 1;
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
