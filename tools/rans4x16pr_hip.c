/*
 * rans4x16pr_hip — the reference's test/benchmark driver (tests/rANS_static4x16pr_test.c) rebuilt
 * on the drop-in library.  Same command line:
 *     rans4x16pr_hip [-o order[.N]] [-d] [-r] [-t] [-b] [infile [outfile]]
 *   -o N[.S]  order / flag bits (1 order-1, 8 stripe, 64 RLE, 128 PACK, ...), .S = stripe count
 *   -d        decode            -r   one raw block (the form the test fixtures use)
 *   -t        benchmark: split the input into BLK_SIZE blocks, 10 trials, print MB/s enc / dec
 *   -b        with -t: use the batch entry points instead of a serial loop over the five functions
 * Without -r the stream is the reference's framed format: 4-byte little-endian length, then block.
 * Only the five htscodecs entry points (and, with -b, the batch pair) are used: the file compiles
 * unchanged against libhtscodecs for the non -b modes.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <sys/time.h>

#include "../include/rans4x16_hip.h"

#ifndef BLK_SIZE
#  define BLK_SIZE (1039 * 251 * 4)      /* tests/rANS_static4x16pr_test.c:46-49 */
#endif
#ifndef NTRIALS
#  define NTRIALS 10
#endif

static unsigned char *load(FILE *fp, uint32_t *lenp)
{
    unsigned char *data = NULL;
    size_t cap = 0, len = 0, got;
    do {
        if (cap - len < BLK_SIZE) { cap = cap ? cap * 2 : BLK_SIZE; data = realloc(data, cap); }
        got = fread(data + len, 1, BLK_SIZE, fp);
        len += got;
    } while (got > 0);
    *lenp = (uint32_t)len;
    return data;
}

static double usec(struct timeval a, struct timeval b)
{
    return (double)(b.tv_sec - a.tv_sec) * 1e6 + (double)(b.tv_usec - a.tv_usec);
}

int main(int argc, char **argv)
{
    int opt, order = 0, decode = 0, test = 0, raw = 0, batch = 0;
    FILE *in = stdin, *out = stdout;
    while ((opt = getopt(argc, argv, "o:dtrb")) != -1) {
        switch (opt) {
        case 'o': {
            char *end;
            order = (int)strtol(optarg, &end, 0);
            if (*end == '.') order += atoi(end + 1) << 8;
            break;
        }
        case 'd': decode = 1; break;
        case 't': test = 1; break;
        case 'r': raw = 1; break;
        case 'b': batch = 1; break;
        default: return 2;
        }
    }
    if (optind < argc && !(in = fopen(argv[optind++], "rb"))) { perror("input"); return 1; }
    if (optind < argc && !(out = fopen(argv[optind++], "wb"))) { perror("output"); return 1; }

    if (test) {
        uint32_t total;
        unsigned char *data = load(in, &total);
        int nb = (int)((total + BLK_SIZE - 1) / BLK_SIZE), i;
        const unsigned char **src = calloc(nb, sizeof(*src));
        unsigned char **comp = calloc(nb, sizeof(*comp)), **back = calloc(nb, sizeof(*back));
        unsigned int *ssz = calloc(nb, 4), *csz = calloc(nb, 4), *ccap = calloc(nb, 4), *bsz = calloc(nb, 4);
        int *ord = calloc(nb, sizeof(int));
        rans4x16_hip_ctx *ctx = batch ? rans4x16_hip_create(-1) : NULL;
        if (batch && !ctx) { fprintf(stderr, "no GPU context\n"); return 1; }
        for (i = 0; i < nb; i++) {
            src[i] = data + (size_t)i * BLK_SIZE;
            ssz[i] = (uint32_t)(i == nb - 1 ? total - (size_t)i * BLK_SIZE : BLK_SIZE);
            ccap[i] = rans_compress_bound_4x16(BLK_SIZE, order);
            comp[i] = malloc(ccap[i]);
            back[i] = malloc(BLK_SIZE);
            ord[i] = order;
        }
        fprintf(stderr, "Testing %d blocks\n", nb);
        for (int trial = 0; trial < NTRIALS; trial++) {
            struct timeval t1, t2, t3, t4;
            size_t out_sz = 0;
            gettimeofday(&t1, NULL);
            if (batch) {
                for (i = 0; i < nb; i++) csz[i] = ccap[i];
                if (rans4x16_hip_compress_batch(ctx, nb, src, ssz, comp, csz, ord, NULL) != 0) return 1;
            } else {
                for (i = 0; i < nb; i++) {
                    csz[i] = ccap[i];
                    if (!rans_compress_to_4x16((unsigned char *)src[i], ssz[i], comp[i], &csz[i], order)) return 1;
                }
            }
            gettimeofday(&t2, NULL);
            for (i = 0; i < nb; i++) out_sz += 5 + csz[i];
            gettimeofday(&t3, NULL);
            if (batch) {
                for (i = 0; i < nb; i++) bsz[i] = ssz[i];
                if (rans4x16_hip_uncompress_batch(ctx, nb, (const unsigned char *const *)comp, csz, back, bsz, NULL) != 0) return 1;
            } else {
                for (i = 0; i < nb; i++) {
                    bsz[i] = ssz[i];
                    if (!rans_uncompress_to_4x16(comp[i], csz[i], back[i], &bsz[i])) return 1;
                }
            }
            gettimeofday(&t4, NULL);
            for (i = 0; i < nb; i++)
                if (bsz[i] != ssz[i] || memcmp(src[i], back[i], ssz[i]))
                    fprintf(stderr, "Mismatch in block %d, sz %u/%u\n", i, ssz[i], bsz[i]);
            fprintf(stderr, "%5.1f MB/s enc, %5.1f MB/s dec\t %ld bytes -> %ld bytes\n",
                    total / usec(t1, t2), total / usec(t3, t4), (long)total, (long)out_sz);
        }
        if (ctx) rans4x16_hip_destroy(ctx);
        return 0;
    }

    if (raw) {
        uint32_t n, m;
        unsigned char *buf = load(in, &n), *res;
        res = decode ? rans_uncompress_4x16(buf, n, &m) : rans_compress_4x16(buf, n, &m, order);
        if (!res) return 1;
        fwrite(res, 1, m, out);
        free(res);
        free(buf);
        return 0;
    }

    static unsigned char blk[BLK_SIZE + 257 * 257 * 3 + 65536];
    for (;;) {
        uint32_t n, m;
        unsigned char *res;
        if (decode) {
            if (fread(&n, 1, 4, in) != 4) break;
            if (n > sizeof(blk) || fread(blk, 1, n, in) != n) { fprintf(stderr, "Truncated input\n"); return 1; }
            if (!(res = rans_uncompress_4x16(blk, n, &m))) return 1;
        } else {
            n = (uint32_t)fread(blk, 1, BLK_SIZE, in);
            if (n == 0) break;
            if (!(res = rans_compress_4x16(blk, n, &m, n < 4 ? order & ~1 : order))) return 1;
            fwrite(&m, 1, 4, out);
        }
        fwrite(res, 1, m, out);
        free(res);
    }
    return 0;
}
