/*
 * rans4x16_oracle.h — CPU ORACLE for the rANS 4x16 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar C restatement of the algorithm in htscodecs 1.1
 * (htscodecs/rANS_static4x16pr.c, pack.c, rle.c, rANS_word.h, varint.h, utils.h).
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * can check the HIP path bit-for-bit.  The product library (htscodecs_amd/csrc) never
 * includes, links or calls anything in this directory.
 *
 * Parity pin: tests/test_oracle.py checks this restatement against all 24 reference-held
 * fixtures (tests/golden/r4x16/ = tests/dat/r4x16/ of the reference: decode AND byte-identical
 * re-encode) and the varint known-answer tables of tests/varint_test.c.  The reference itself
 * is not built here (its sources need an autotools-generated config.h, oracle/Makefile); the
 * generated edge vectors of tests/golden/edge.json carry THIS restatement's outputs and are
 * regression vectors for the HIP path, not a second pin.
 *
 * Symbols carry an orc_ prefix so the oracle and the product library can live in one process.  Signatures mirror htscodecs/rANS_static4x16.h:41-50.
 */
#ifndef RANS4X16_ORACLE_H
#define RANS4X16_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

unsigned int   orc_rans_compress_bound_4x16(unsigned int size, int order);
unsigned char *orc_rans_compress_to_4x16(unsigned char *in, unsigned int in_size,
                                         unsigned char *out, unsigned int *out_size, int order);
unsigned char *orc_rans_compress_4x16(unsigned char *in, unsigned int in_size,
                                      unsigned int *out_size, int order);
unsigned char *orc_rans_uncompress_to_4x16(unsigned char *in, unsigned int in_size,
                                           unsigned char *out, unsigned int *out_size);
unsigned char *orc_rans_uncompress_4x16(unsigned char *in, unsigned int in_size,
                                        unsigned int *out_size);

/* Building blocks exposed for unit tests (varint KATs, table exactness, transforms). */
int      orc_var_put_u32(uint8_t *cp, uint32_t v);
int      orc_var_get_u32(const uint8_t *cp, const uint8_t *endp, uint32_t *v);
int      orc_normalise_freq(uint32_t *F, int size, uint32_t tot);
int      orc_compute_shift(const uint32_t *F0, const uint32_t (*F)[256], const uint32_t *T, int *S);
/* Bare streams (no container byte / size): the O0stream / O1stream of SURVEY Appendix A. */
int      orc_o0_encode(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, uint32_t *out_len);
int      orc_o0_decode(const uint8_t *in, uint32_t in_size, uint8_t *out, uint32_t out_sz);
int      orc_o1_encode(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, uint32_t *out_len);
int      orc_o1_decode(const uint8_t *in, uint32_t in_size, uint8_t *out, uint32_t out_sz);
/* pack / rle transforms (pack.c:56-151, 165-348; rle.c:48-187). */
int      orc_pack(const uint8_t *in, uint64_t n, uint8_t *meta, int *meta_len, uint8_t *out, uint64_t *out_len);
int      orc_rle_encode(const uint8_t *in, uint64_t n, uint8_t *runs, uint64_t *runs_len,
                        uint8_t *syms, int *nsyms, uint8_t *lits, uint64_t *lits_len);

/* Multi-threaded CPU baseline helper used by bench.py: processes n blocks with nthreads
 * pthreads over disjoint block ranges (SURVEY §8d "all cores").  Returns #failed blocks. */
int orc_compress_many(int n, unsigned char *const *in, const unsigned int *in_size,
                      unsigned char *const *out, unsigned int *out_size, int order, int nthreads);
int orc_uncompress_many(int n, unsigned char *const *in, const unsigned int *in_size,
                        unsigned char *const *out, unsigned int *out_size, int nthreads);

/* rANS 4x8 (rans4x8_oracle.c): htscodecs/rANS_static.h:41-44 with an orc8_ prefix; malloc'd results. */
unsigned char *orc8_rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order);
unsigned char *orc8_rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size);

#ifdef __cplusplus
}
#endif
#endif
