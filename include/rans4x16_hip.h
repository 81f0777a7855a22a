/*
 * rans4x16_hip.h — C ABI of librans4x16_hip.so: the MI355X (gfx950) implementation of the
 * CRAM 3.1 rANS 4x16 codec, bit-exact with htscodecs 1.1.
 *
 * Two layers, both plain C (pointers + sizes, no C++/torch types):
 *
 *  1. The five entry points of htscodecs/rANS_static4x16.h:41-50, same names, same argument
 *     meaning, same ownership and error rules.  A program linked against
 *     libhtscodecs can be re-linked against this library for this codec without source changes.
 *     Buffers are HOST memory; each call stages through the GPU as a batch of one.
 *
 *  2. Batch entry points — the shape of the reference's own benchmark loop
 *     (tests/rANS_static4x16pr_test.c:191-206: a serial loop of rans_compress_to_4x16 /
 *     rans_uncompress_to_4x16 over independent blocks), which is what a GPU needs to be fed.
 *     *_batch take host buffers; *_dev take device-resident buffers (no PCIe in the call),
 *     enqueue on a caller-supplied HIP stream and do not synchronise.
 *
 * All GPU work is done by hand-written HIP kernels; there is no CPU fallback.  If no usable
 * GPU is present every entry point fails (NULL / negative return) and says why on stderr once.
 */
#ifndef RANS4X16_HIP_H
#define RANS4X16_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#pragma GCC visibility push(default)

/* ---- 1. drop-in replacements (htscodecs/rANS_static4x16.h:41-50) ------------------------- */

/* rANS_static4x16pr.c:360-372.  Pure host arithmetic. */
unsigned int rans_compress_bound_4x16(unsigned int size, int order);

/* rANS_static4x16pr.c:1138-1345.  out==NULL: malloc'd result (caller frees); else *out_size is
 * the capacity on entry (must be >= rans_compress_bound_4x16(in_size, order), otherwise NULL)
 * and the produced size on return.  order: bit0 order-1, 0x80 PACK, 0x40 RLE, 0x20 CAT,
 * 0x10 NOSZ, 0x08 STRIPE, bits 8..15 stripe count. */
unsigned char *rans_compress_to_4x16(unsigned char *in, unsigned int in_size,
                                     unsigned char *out, unsigned int *out_size, int order);
/* rANS_static4x16pr.c:1347-1350 */
unsigned char *rans_compress_4x16(unsigned char *in, unsigned int in_size,
                                  unsigned int *out_size, int order);
/* rANS_static4x16pr.c:1352-1636.  NULL on any malformed input. */
unsigned char *rans_uncompress_to_4x16(unsigned char *in, unsigned int in_size,
                                       unsigned char *out, unsigned int *out_size);
/* rANS_static4x16pr.c:1638-1641 */
unsigned char *rans_uncompress_4x16(unsigned char *in, unsigned int in_size,
                                    unsigned int *out_size);

/* ---- 2. batch interface ------------------------------------------------------------------ */

typedef struct rans4x16_hip_ctx rans4x16_hip_ctx;

/* Per-block status codes written to the status arrays (0 = success). */
enum {
    R4X16_OK            = 0,
    R4X16_E_CAPACITY    = 1,   /* output capacity too small (encode: < bound; decode: < stored size) */
    R4X16_E_TRUNCATED   = 2,   /* input ends inside a header, table or state words               */
    R4X16_E_TABLE       = 3,   /* frequency table does not sum to a power of two / overflows      */
    R4X16_E_STATE       = 4,   /* initial rANS state below 2^15                                   */
    R4X16_E_SIZE        = 5,   /* inconsistent size fields (pack / rle / cat)                     */
    R4X16_E_UNSUPPORTED = 6,   /* valid-looking stream this build does not handle */
    R4X16_E_CONTEXT     = 7,   /* order-1 stream used a context that has no table row             */
    R4X16_E_RLE         = 8,   /* run-length expansion overran the output                         */
    R4X16_E_EMPTY       = 9    /* zero-length compressed input                                    */
};

/* One context per (host thread, device).  device < 0 selects the current HIP device.
 * Returns NULL if the device or the code object is unusable. */
rans4x16_hip_ctx *rans4x16_hip_create(int device);
void              rans4x16_hip_destroy(rans4x16_hip_ctx *ctx);
const char       *rans4x16_hip_last_error(const rans4x16_hip_ctx *ctx);

/* Host-buffer batches: n independent blocks, semantics of n calls of the functions in part 1
 * with caller-provided output buffers (out[i] != NULL, out_size[i] = capacity in / size out).
 * Returns the number of failed blocks (their out_size[i] is set to 0 and status[i] != 0 if
 * status is not NULL), or -1 if the batch could not be run at all. */
int rans4x16_hip_compress_batch(rans4x16_hip_ctx *ctx, int n,
                                const unsigned char *const *in, const unsigned int *in_size,
                                unsigned char *const *out, unsigned int *out_size,
                                const int *order, int *status);
int rans4x16_hip_uncompress_batch(rans4x16_hip_ctx *ctx, int n,
                                  const unsigned char *const *in, const unsigned int *in_size,
                                  unsigned char *const *out, unsigned int *out_size,
                                  int *status);

/* "Try k methods, keep the smallest": the caller-side pattern of htscodecs/tokenise_name3.c:1246-1300
 * (compress(): up to nine `order` values per token column, smallest kept) and of CRAM block writers, as one
 * call.  Block i is encoded with every methods[j]; out[i] receives the smallest result, chosen[i] the
 * method that produced it (the first wins ties, tokenise_name3.c:1283-1286; -1 if every candidate
 * failed).  Methods with X_STRIPE (0x08) are skipped for blocks whose size is not a multiple of 4
 * (:1271-1272).  out_size[i] is the capacity on entry (at least the largest
 * rans_compress_bound_4x16(in_size[i], methods[j])) and the winner's size on return.
 * The candidates share one upload of the block; only the winner is copied back.
 * Returns the number of blocks without any successful candidate, or -1. */
int rans4x16_hip_compress_best_batch(rans4x16_hip_ctx *ctx, int n,
                                     const unsigned char *const *in, const unsigned int *in_size,
                                     unsigned char *const *out, unsigned int *out_size,
                                     int k, const int *methods, int *chosen, int *status);

/* Device-resident batches.  Every pointer below is a DEVICE pointer.
 *   d_in  + d_in_off[i]   : block i input,  d_in_size[i] bytes
 *   d_out + d_out_off[i]  : block i output slot, d_out_cap[i] bytes available
 *   d_out_size[i]         : bytes produced (0 on failure);  d_status[i]: code above
 * `order` applies to all blocks unless d_order != NULL (device array of n ints).
 * max_in_size / max_out_cap are host-side upper bounds on the per-block sizes, used only to
 * size the workspace (no device->host read-back happens inside these calls).  For decode, max_out_cap sizes
 * the stage buffers of X_PACK / X_RLE blocks only: it may be the largest output of THOSE blocks, 0 if the batch
 * has none (a transformed block larger than that reports UNSUPPORTED).
 * `stream` is a hipStream_t (NULL = default stream).  The call only enqueues work.
 * X_STRIPE (0x08), device-resident: encode accepts it when all blocks share one `order` (d_order == NULL): the
 * planes, the N x K candidate encodings, the choice of the smallest per plane and the header are all produced on the
 * device (rANS_static4x16pr.c:1154-1216); with per-block orders a stripe block reports UNSUPPORTED.  Decode accepts
 * stripe blocks after rans4x16_hip_set_dev_stripe_planes (below); without it they report UNSUPPORTED.  The host
 * entry points handle stripes in every case.
 * Returns 0 if enqueued, -1 on argument / allocation / launch errors. */
int rans4x16_hip_compress_dev(rans4x16_hip_ctx *ctx, int n,
                              const unsigned char *d_in, const uint64_t *d_in_off,
                              const uint32_t *d_in_size,
                              unsigned char *d_out, const uint64_t *d_out_off,
                              const uint32_t *d_out_cap, uint32_t *d_out_size,
                              int32_t *d_status, int order, const int32_t *d_order,
                              uint32_t max_in_size, void *stream);
int rans4x16_hip_uncompress_dev(rans4x16_hip_ctx *ctx, int n,
                                const unsigned char *d_in, const uint64_t *d_in_off,
                                const uint32_t *d_in_size,
                                unsigned char *d_out, const uint64_t *d_out_off,
                                const uint32_t *d_out_cap, uint32_t *d_out_size,
                                int32_t *d_status, uint32_t max_in_size, uint32_t max_out_cap,
                                void *stream);

/* The same calls with one more host-side figure, the SUM of the blocks' sizes: the workspace a block needs for
 * X_PACK / X_RLE depends on its own length, the device lays those regions out itself, and the host only has to bound
 * their total - total_in_size (encode: sum of d_in_size) / total_out_cap (decode: sum of d_out_cap over the blocks
 * that carry X_PACK or X_RLE; the sum over all blocks is a valid bound) instead of n x the largest block.  A batch of
 * 22,729 blocks of 4 KiB .. 1 MiB (4 GiB) then takes 29 GB of workspace instead of 62.  0 = unknown
 * (the plain calls above).  A batch that holds more than it announced fails the blocks that do not fit (UNSUPPORTED). */
int rans4x16_hip_compress_dev_sized(rans4x16_hip_ctx *ctx, int n,
                                    const unsigned char *d_in, const uint64_t *d_in_off,
                                    const uint32_t *d_in_size,
                                    unsigned char *d_out, const uint64_t *d_out_off,
                                    const uint32_t *d_out_cap, uint32_t *d_out_size,
                                    int32_t *d_status, int order, const int32_t *d_order,
                                    uint32_t max_in_size, uint64_t total_in_size, void *stream);
int rans4x16_hip_uncompress_dev_sized(rans4x16_hip_ctx *ctx, int n,
                                      const unsigned char *d_in, const uint64_t *d_in_off,
                                      const uint32_t *d_in_size,
                                      unsigned char *d_out, const uint64_t *d_out_off,
                                      const uint32_t *d_out_cap, uint32_t *d_out_size,
                                      int32_t *d_status, uint32_t max_in_size, uint32_t max_out_cap,
                                      uint64_t total_out_cap, void *stream);

/* Device-resident decode of X_STRIPE blocks: the flag and the plane count N live in the stream, so the host cannot
 * size the workspace per block; this sets what every block of later rans4x16_hip_uncompress_dev calls reserves:
 * `planes` internal sub-blocks (the default N is 4) and a plane buffer of `max_block_size` bytes.  A stripe block
 * with more planes, or larger than that, reports UNSUPPORTED; like the reference (:1379) a stripe block must be given
 * an output capacity equal to its stored size.  planes == 0 (the default) switches it off.  Returns 0, -1 on bad arguments. */
int rans4x16_hip_set_dev_stripe_planes(rans4x16_hip_ctx *ctx, int planes, unsigned int max_block_size);

/* ---- 2b. options ---------------------------------------------------------------------------
 * Everything that can be tuned or switched is an option of the context, set by name; the value is a long.
 * The R4X16_* environment variables named below only provide the DEFAULTS: they are read once per process, when the
 * first context is created (or the first option is asked for); no call path reads the environment.
 * ctx == NULL addresses the process-wide defaults: what contexts created from now on start with, and the
 * process-wide options at the end of the list.  Returns 0, or -1 for an unknown name.
 *
 *   name               default  environment default     meaning
 *   dec_direct            1     R4X16_DEC_DIRECT         decode: direct (short-step) rows for batches of up to N rounds of
 *                                                        resident direct streams; 0 = never
 *   enc_direct            1     R4X16_ENC_DIRECT         encode: the same for symbol records
 *   back_wg_per_cu        0     R4X16_BACK_WG_PER_CU     decode: run-length expansion by a workgroup per block up to N
 *                                                        blocks per compute unit (0: always one wave per block)
 *   dec_mid               0     R4X16_DEC_MID            decode: mid rows (bucket index + one 16-byte window of cumulative values) for
 *                                                        batches of up to N rounds of sixteen streams per compute unit; 0 = never
 *                                                        (built and measured in round 4, slower than the packed rows on quality data: off)
 *   dec_short_ring        0     R4X16_DEC_SHORT_RING     decode: packed rows of 43..44 symbols with a 128-byte word ring and four-step trips,
 *                                                        sixteen streams per wave instead of fifteen (measured slower per round: off)
 *   sched_sort            1     R4X16_SCHED_SORT         chain kernels: streams of a class ordered by length, longest first
 *   sched_claim           1     R4X16_SCHED_CLAIM        chain kernels: shares claimed from a counter (0: fixed stride)
 *   sched_concurrent      1     R4X16_SCHED_CONCURRENT   chain kernels: the classes of a batch side by side on six streams,
 *                                                        each with its share of the chip (0: one after the other)
 *   sched_trace           0     R4X16_SCHED_TRACE        the last batch's classes, and how this one's launches are dealt out, on stderr
 *   sched_learn           2     R4X16_SCHED_LEARN        bit 0 / bit 1: the encoder's / decoder's shares follow what the classes' launches of
 *                                                        the context's earlier batches really took (measured: the decoder gains 10 % on a
 *                                                        heterogeneous batch, the encoder's shares start to swing - so 2)
 *   max_workspace_mb  163840    R4X16_MAX_WS_MB          ceiling of the device workspace; larger batches are walked in chunks
 *   host_pipe_mb         64     R4X16_HOST_PIPE_MB       host batches of at least this many MiB (or 32 blocks) are pipelined
 *   host_threads          8     R4X16_HOST_THREADS       copier threads of the host pipeline
 *   host_lanes            2     R4X16_HOST_LANES         slabs of a host batch in flight at once
 *   host_slab_min_mb     32     R4X16_HOST_SLAB_MIN_MB   smallest slab
 *   host_dec_slabs / host_enc_slabs  0                   slabs per lane and round; 0 = by size (about 2.3 GB of input + capacity each)
 *   host_pack             1     R4X16_HOST_PACK          encode results gathered on the device before they cross PCIe
 *   host_stripe_dev       1     R4X16_HOST_STRIPE_DEV    X_STRIPE blocks of host batches through the device stripe kernels
 *   host_trace            0     R4X16_HOST_TRACE         timeline of a pipelined host batch on stderr
 *   dec_qpw, dec_qpw_small, dec_qpw_pk, dec_qpw_dir, enc_qpw, enc_waves, enc_qpw_rec, enc_qpw_cap, front_lds
 *                                                        tuning aids: streams per wave / workgroup of single classes
 *   process-wide (ctx == NULL, before the first single-block or multi-device call):
 *   combine               1     R4X16_COMBINE            the five drop-in symbols go through the combiner
 *   combine_window_us    -1     R4X16_COMBINE_WINDOW_US  fixed gathering window (-1: adaptive)
 *   combine_max         256     R4X16_COMBINE_MAX        blocks per combined batch
 *   combine_workers       1     R4X16_COMBINE_WORKERS    worker threads per direction
 *   combine_max_mb     2048     R4X16_COMBINE_MAX_MB     buffer bytes per combined batch
 *   numa                  1     R4X16_NUMA               multi-device calls bind each device's worker to its NUMA node
 */
int rans4x16_hip_set_option(rans4x16_hip_ctx *ctx, const char *name, long value);
int rans4x16_hip_get_option(const rans4x16_hip_ctx *ctx, const char *name, long *value);
/* Name of option number `index` (0, 1, ..), NULL past the last: lets a caller list what this build knows. */
const char *rans4x16_hip_option_name(int index);

/* Bytes of device workspace the context currently holds (grows on demand, never shrinks). */
size_t rans4x16_hip_workspace_bytes(const rans4x16_hip_ctx *ctx);

/* Timing hook for bench.py / rocprof cross-checks: when enabled, the *_dev calls bracket their
 * dominant ("chain") kernel with HIP events on the same stream; after synchronising, this
 * returns the accumulated milliseconds and launch count since the last reset. */
void rans4x16_hip_timing(rans4x16_hip_ctx *ctx, int enable);
int  rans4x16_hip_timing_read(rans4x16_hip_ctx *ctx, int which /*0 enc chain, 1 dec chain*/,
                              double *ms_total, int *launches, int reset);

/* How many streams of one kind the chain kernel of this build keeps resident per compute unit (the unit of
 * parallelism is the stream): `nsym` symbols in the alphabet, order 0 / 1, table precision `shift`
 * (10 or 12; ignored for order 0).  Host arithmetic on the kernels' LDS size classes; bench.py reports it next
 * to the measured step latency, and sizes its batch in whole rounds of it. */
int rans4x16_hip_residency(rans4x16_hip_ctx *ctx, int decode, unsigned int nsym, int order, unsigned int shift,
                           int *streams_per_cu, int *lanes_live_per_wave, int *compute_units);

/* Peak shader clock of the context's device in kHz (for cycles-per-step figures), -1 on error. */
int rans4x16_hip_device_clock_khz(rans4x16_hip_ctx *ctx);

/* Library/ABI version and the gfx target the code object was built for. */
const char *rans4x16_hip_version(void);

/* ---- 3. several GPUs of one node ---------------------------------------------------------------
 * Blocks are independent (every table travels in-band), so a batch is cut into contiguous ranges, one
 * per device, and each device runs the single-GPU pipeline on its range: no collective, no peer
 * traffic.  The split is the library's, not the caller's. */

/* Contiguous partition of n blocks into `parts` ranges of near-equal total weight (greedy on the
 * cumulative sum; a block goes to the range in which its midpoint falls).  Range r is
 * [bounds[r], bounds[r+1]); bounds has parts + 1 entries, bounds[0] = 0, bounds[parts] = n; ranges may
 * be empty.  weight == NULL means equal weights.  Pure host arithmetic, usable without a GPU (a
 * multi-process launcher calls it to find its rank's share).  Returns 0, or -1 on bad arguments. */
int rans4x16_hip_partition(int n, const unsigned int *weight, int parts, int *bounds);

typedef struct rans4x16_hip_multi rans4x16_hip_multi;

/* One context per listed device.  devices == NULL: devices 0 .. ndev-1; ndev <= 0: every visible device.
 * A device may be listed more than once (two pipelines on one card: how a one-GPU box rehearses the
 * multi-device path).  NULL if any context cannot be created. */
rans4x16_hip_multi *rans4x16_hip_multi_create(int ndev, const int *devices);
void                rans4x16_hip_multi_destroy(rans4x16_hip_multi *m);
int                 rans4x16_hip_multi_devices(const rans4x16_hip_multi *m);
const char         *rans4x16_hip_multi_last_error(const rans4x16_hip_multi *m);

/* rans4x16_hip_{compress,uncompress}_batch over all devices of `m`: the batch is partitioned by
 * uncompressed bytes (in_size for encode, the out_size capacities for decode), one host thread per
 * device runs its range through the host-buffer pipeline, and sizes / statuses land in the caller's
 * arrays in block order.  Same return value as the single-device calls. */
int rans4x16_hip_compress_batch_multi(rans4x16_hip_multi *m, int n,
                                      const unsigned char *const *in, const unsigned int *in_size,
                                      unsigned char *const *out, unsigned int *out_size,
                                      const int *order, int *status);
int rans4x16_hip_uncompress_batch_multi(rans4x16_hip_multi *m, int n,
                                        const unsigned char *const *in, const unsigned int *in_size,
                                        unsigned char *const *out, unsigned int *out_size,
                                        int *status);

/* Host feed of a multi-GPU node.  With eight devices the limiter is the host side: each device's
 * pipeline has eight copier threads moving the caller's buffers through pinned bounce buffers, and on a two-socket
 * node a copier on the wrong socket pushes every byte over the inter-socket link first.  The multi-device calls
 * therefore run each device's worker - and with it the copier threads it starts and the bounce buffers it
 * allocates - on the CPUs of the NUMA node the device hangs off (PCI bus id -> /sys/bus/pci/devices/<id>/numa_node
 * -> /sys/devices/system/node/node<N>/cpulist); the calling thread's own mask is restored afterwards.
 * R4X16_NUMA=0 switches it off; nothing happens where the node is unknown (-1) or the machine has one node. */

/* Parse a kernel "cpulist" ("0-15,32-47", "3", "0-3,8") into a bit mask of mask_bytes bytes (CPU c = bit c & 7 of byte
 * c >> 3).  Returns the number of CPUs set, or -1 on a malformed list or a CPU beyond the mask.  Pure text work,
 * usable (and tested) without a GPU. */
int rans4x16_hip_cpulist_parse(const char *list, unsigned char *mask, int mask_bytes);

/* NUMA node of a device of `m` (index into its device list) as the kernel reports it, -1 if unknown. */
int rans4x16_hip_multi_numa_node(const rans4x16_hip_multi *m, int index);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* RANS4X16_HIP_H */
