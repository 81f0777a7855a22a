/*
 * rans4x8_hip.h — rANS 4x8 (CRAM 3.0's codec) in librans4x16_hip.so, bit-exact with htscodecs 1.1's
 * rANS_static.c.  Same two layers as rans4x16_hip.h and the same context type (rans4x16_hip_create):
 *
 *  1. The two entry points of htscodecs/rANS_static.h:41-44, same names and ownership (results are malloc'd,
 *     the caller frees; NULL on failure).  Host buffers, a batch of one per call.
 *  2. Batch calls on host buffers and on device-resident buffers, with the argument meaning of the 4x16 ones.
 *
 * order: 0 / non-zero = order-1.  No CPU path: without a GPU every call fails.
 * On damaged input the device refuses a few streams the reference lets through with undefined results
 * (DESIGN.md 11); valid encoder output is never affected.
 */
#ifndef RANS4X8_HIP_H
#define RANS4X8_HIP_H

#include "rans4x16_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* ---- 1. drop-in replacements (htscodecs/rANS_static.h:41-44) ---------------------------------- */
/* rANS_static.c:927-932.  in_size == 0 returns NULL (the reference divides by zero there). */
unsigned char *rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order);
/* rANS_static.c:934-943 */
unsigned char *rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size);

/* ---- 2. batches -------------------------------------------------------------------------------- */
/* Capacity an output slot needs: what the reference allocates, (int)(1.05 * size) + 257*257*3 + 9
 * (rANS_static.c:87, :448). */
unsigned int rans4x8_hip_compress_bound(unsigned int size);

/* Host buffers; out_size[i] is the capacity on entry (encode: at least rans4x8_hip_compress_bound(in_size[i]);
 * decode: at least the uncompressed size stored in bytes 5..8 of the stream) and the size produced on return.
 * Returns the number of failed blocks (out_size 0, status != 0), -1 if the batch could not be run. */
int rans4x8_hip_compress_batch(rans4x16_hip_ctx *ctx, int n,
                               const unsigned char *const *in, const unsigned int *in_size,
                               unsigned char *const *out, unsigned int *out_size,
                               const int *order, int *status);
int rans4x8_hip_uncompress_batch(rans4x16_hip_ctx *ctx, int n,
                                 const unsigned char *const *in, const unsigned int *in_size,
                                 unsigned char *const *out, unsigned int *out_size, int *status);

/* Device-resident buffers (every pointer a DEVICE pointer; see rans4x16_hip_compress_dev for the layout).
 * The calls only enqueue work on `stream`.
 * PADDING: the decoder fetches a stream in aligned 16-byte pieces, so up to 15 bytes before a block's first byte and
 * after its last one are READ (never used): d_in must stay readable for 16 bytes beyond the end of its last block, and
 * a block must not start within 15 bytes of the start of the allocation unless that start is 16-byte aligned (any
 * hipMalloc'ed arena is).  The host-buffer calls above pad their own staging. */
int rans4x8_hip_compress_dev(rans4x16_hip_ctx *ctx, int n,
                             const unsigned char *d_in, const uint64_t *d_in_off, const uint32_t *d_in_size,
                             unsigned char *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                             uint32_t *d_out_size, int32_t *d_status, int order, const int32_t *d_order,
                             uint32_t max_in_size, void *stream);
int rans4x8_hip_uncompress_dev(rans4x16_hip_ctx *ctx, int n,
                               const unsigned char *d_in, const uint64_t *d_in_off, const uint32_t *d_in_size,
                               unsigned char *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                               uint32_t *d_out_size, int32_t *d_status, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RANS4X8_HIP_H */
