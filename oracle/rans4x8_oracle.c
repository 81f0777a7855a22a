/*
 * rans4x8_oracle.c — CPU ORACLE for rANS 4x8 (CRAM 3.0's codec).  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar C restatement of htscodecs 1.1's rANS_static.c + rANS_byte.h (8-bit renormalisation, L = 2^23,
 * 12-bit frequencies, four interleaved states), written from the format, not copied: division-form encoder
 * step, bounded readers, one renormalisation rule for the whole stream.  Used by tests/, smoke() and
 * bench.py's cpu_baseline leg only; the product library never includes, links or calls it.
 *
 * Parity pin: tests/test_oracle4x8.py checks it against the reference's eight fixtures
 * (tests/golden/r4x8/, byte-identical copies of the files under /root/reference/tests/dat/r4x8/: decode AND byte-identical
 * re-encode).  The reference itself is not built in this repository (its sources need an autotools-generated config.h,
 * oracle/Makefile): the fixtures are the pin.
 *
 * Stream (rANS_static.c:196-214, :590-607):
 *   byte 0      order (0 / 1)
 *   bytes 1-4   compressed size - 9, little endian        bytes 5-8   uncompressed size, little endian
 *   table       order 0: symbols in ascending order with a run-length shortcut, each followed by its frequency
 *               (one byte below 128, else 0x80 | high byte, low byte), closed by 0 (:143-169);
 *               order 1: the same for contexts, each followed by its own symbol table (:475-533)
 *   4 states    little endian, state 0 first; then the renormalisation bytes.
 * Frequencies of a table sum to 4095 as this encoder writes them ("historically we fill 4095", :807); the
 * decoder accepts 4095 and 4096.
 *
 * Where the reference is undefined on damaged input this restatement FAILS instead (and so does the device):
 *   - a lookup of slot 4095 in a table that sums to 4095 (the reference reads an unwritten table entry,
 *     rANS_static.c:257-259 / a byte written into another context's row, :807-808);
 *   - an order-1 context that has no table (the reference reads per-thread tables left by earlier calls, :685-706);
 *   - in_size == 0 on encode (the reference divides by zero, :108).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define X8_SHIFT 12
#define X8_TOT   4096u
#define X8_LOW   (1u << 23)                       /* rANS_byte.h:62 */

/* ---- shared helpers ----------------------------------------------------------------------------------- */

static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void wr32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* One encoder step (rANS_byte.h:281-318): up to two bytes leave the state, low byte first, on a descending
 * pointer; then x = (x / f) * 4096 + x % f + start.  x_max = 2^19 * f (:217). */
static inline uint32_t enc_put(uint32_t x, uint8_t **pp, uint32_t start, uint32_t f)
{
    const uint64_t x_max = (uint64_t)f << 19;
    uint8_t *p = *pp;
    if (x >= x_max) { *--p = (uint8_t)x; x >>= 8; if (x >= x_max) { *--p = (uint8_t)x; x >>= 8; } }
    *pp = p;
    return ((x / f) << X8_SHIFT) + (x % f) + start;
}

/* Renormalisation of one state (rANS_byte.h:439-551): bytes are taken while x < 2^23, at most two, never past
 * the end of the input.  The reference switches between an unchecked form (more than 8 bytes left) and a
 * checked one; both obey this rule. */
static inline uint32_t dec_renorm(uint32_t x, const uint8_t **pp, const uint8_t *end)
{
    const uint8_t *p = *pp;
    if (x < X8_LOW && p < end) {
        x = (x << 8) | *p++;
        if (x < X8_LOW && p < end) x = (x << 8) | *p++;
    }
    *pp = p;
    return x;
}

/* Frequency table writer shared by both orders (:143-169, :497-531): symbols j with F[j] != 0 in ascending
 * order; a symbol whose predecessor is also present is followed by the count of further consecutive present
 * symbols, which are then implicit. */
static uint8_t *put_table(uint8_t *cp, const int *F)
{
    int rle = 0;
    for (int j = 0; j < 256; j++) {
        if (!F[j]) continue;
        if (rle) rle--;
        else {
            *cp++ = (uint8_t)j;
            if (j && F[j - 1]) {
                int r = j + 1;
                while (r < 256 && F[r]) r++;
                rle = r - (j + 1);
                *cp++ = (uint8_t)rle;
            }
        }
        if (F[j] < 128) *cp++ = (uint8_t)F[j];
        else { *cp++ = (uint8_t)(128 | (F[j] >> 8)); *cp++ = (uint8_t)(F[j] & 0xff); }
    }
    *cp++ = 0;
    return cp;
}

/* ---- order 0 ------------------------------------------------------------------------------------------- */

static unsigned char *enc0(const uint8_t *in, uint32_t n, uint32_t *out_size)
{
    const size_t cap = (size_t)(int)(1.05 * n) + 257 * 257 * 3 + 9;          /* :87, :100 */
    uint8_t *out = malloc((size_t)(1.05 * n + 257 * 257 * 3 + 9) + 16);
    if (!out) return NULL;
    int F[256] = {0};
    for (uint32_t i = 0; i < n; i++) F[in[i]]++;
    /* normalise to 4095 (:106-133): fixed-point scale, the largest symbol absorbs the rest */
    uint64_t tr = ((uint64_t)X8_TOT << 31) / n + (1u << 30) / n;
    int fsum, M;
    for (;;) {
        int m = 0;
        fsum = 0; M = 0;
        for (int j = 0; j < 256; j++) {
            if (!F[j]) continue;
            if (m < F[j]) { m = F[j]; M = j; }
            F[j] = (int)(((uint64_t)F[j] * tr) >> 31);
            if (F[j] == 0) F[j] = 1;
            fsum += F[j];
        }
        fsum++;
        if (fsum < (int)X8_TOT) { F[M] += (int)X8_TOT - fsum; break; }
        if (fsum - (int)X8_TOT > F[M] / 2) { tr = 2104533975u; continue; }       /* :127 */
        F[M] -= fsum - (int)X8_TOT;
        break;
    }
    uint8_t *cp = put_table(out + 9, F);
    const uint32_t tab = (uint32_t)(cp - out);
    uint32_t C[256], x = 0;
    for (int j = 0; j < 256; j++) { C[j] = x; x += (uint32_t)F[j]; }

    uint8_t *end = out + cap, *p = end;
    uint32_t R[4] = {X8_LOW, X8_LOW, X8_LOW, X8_LOW};
    /* byte i belongs to state i & 3; the bytes are coded from the last to the first (:178-197) */
    for (uint32_t i = n; i-- > 0; ) R[i & 3] = enc_put(R[i & 3], &p, C[in[i]], (uint32_t)F[in[i]]);
    for (int k = 3; k >= 0; k--) { p -= 4; wr32(p, R[k]); }                        /* :199-202 */
    const uint32_t pay = (uint32_t)(end - p);
    *out_size = pay + tab;
    out[0] = 0;
    wr32(out + 1, *out_size - 9);
    wr32(out + 5, n);
    memmove(out + tab, p, pay);
    return out;
}

/* Table reader shared by both orders (:274-310, :760-805), in the reference's reading order: each listed symbol
 * takes the next `F` slots (symtab[x .. x+F) = symbol) and records its start and frequency - a symbol listed twice
 * (damaged input only) keeps both slot ranges but the later start / frequency, as the reference's tables do.
 * Returns the new position, NULL on failure.  `zero_is_total`: a frequency byte of 0 means 4096 (order 1, :770). */
static const uint8_t *get_table(const uint8_t *cp, const uint8_t *end, uint32_t *start, uint32_t *freq, uint8_t *symtab,
                                uint32_t *total, int zero_is_total)
{
    uint32_t x = 0;
    int rle = 0;
    int j = *cp++;
    do {
        if (cp > end - 16) return NULL;
        uint32_t F = *cp++;
        if (F >= 128) F = ((F & 127) << 8) | *cp++;
        if (!F && zero_is_total) F = X8_TOT;
        if (x + F > X8_TOT) return NULL;
        start[j] = x; freq[j] = F;
        memset(symtab + x, j, F);
        x += F;
        if (!rle && j + 1 == *cp) { j = *cp++; rle = *cp++; }
        else if (rle) { rle--; j++; if (j > 255) return NULL; }
        else j = *cp++;
    } while (j);
    if (x < X8_TOT - 1 || x > X8_TOT) return NULL;
    *total = x;
    return cp;
}

static unsigned char *dec0(const uint8_t *in, uint32_t in_size, uint32_t *out_size)
{
    if (in_size < 26 || in[0] != 0) return NULL;                                  /* :245-249 */
    const uint32_t in_sz = rd32(in + 1), out_sz = rd32(in + 5);
    if (in_sz != in_size - 9 || out_sz >= INT_MAX) return NULL;
    const uint8_t *end = in + in_size;
    uint32_t start[256] = {0}, freq[256] = {0}, total;
    uint8_t *sym = calloc(1, X8_TOT);                                             /* slot -> symbol (:292-296) */
    uint8_t *out = malloc(out_sz ? out_sz : 1);
    if (!sym || !out) { free(sym); free(out); return NULL; }
    const uint8_t *cp = get_table(in + 9, end, start, freq, sym, &total, 0);
    if (!cp || cp > end - 16) goto fail;
    uint32_t R[4];
    for (int k = 0; k < 4; k++, cp += 4) { R[k] = rd32(cp); if (R[k] < X8_LOW) goto fail; }   /* :316-319 */
    const uint32_t whole = out_sz & ~3u;
    for (uint32_t i = 0; i < whole; i += 4) {
        for (int k = 0; k < 4; k++) {
            const uint32_t m = R[k] & (X8_TOT - 1);
            if (m >= total) goto fail;                          /* slot 4095 of a 4095 table: see the header */
            const int s = sym[m];
            out[i + k] = (uint8_t)s;
            R[k] = freq[s] * (R[k] >> X8_SHIFT) + m - start[s];
        }
        for (int k = 0; k < 4; k++) R[k] = dec_renorm(R[k], &cp, end);
    }
    for (uint32_t k = 0; k < (out_sz & 3); k++) {              /* :363-373: looked up, not advanced */
        const uint32_t m = R[k] & (X8_TOT - 1);
        if (m >= total) goto fail;
        out[whole + k] = sym[m];
    }
    free(sym);
    *out_size = out_sz;
    return out;
fail:
    free(sym); free(out);
    return NULL;
}

/* ---- order 1 ------------------------------------------------------------------------------------------- */

static unsigned char *enc1(const uint8_t *in, uint32_t n, uint32_t *out_size)
{
    if (n < 4) return enc0(in, n, out_size);                                      /* :438 */
    const size_t cap = (size_t)(int)(1.05 * n) + 257 * 257 * 3 + 9;
    uint8_t *out = malloc((size_t)(1.05 * n + 257 * 257 * 3 + 9) + 16);
    int (*F)[256] = calloc(256, sizeof(*F));
    uint32_t (*C)[256] = calloc(256, sizeof(*C));
    if (!out || !F || !C) { free(out); free(F); free(C); return NULL; }
    int T[256] = {0};
    /* pair counts F[previous][current], previous of byte 0 = 0 (utils.h:137-202), plus the first byte of
     * quarters 1..3 in context 0 (:455-458) */
    {
        int prev = 0;
        for (uint32_t i = 0; i < n; i++) { F[prev][in[i]]++; T[prev]++; prev = in[i]; }
        const uint32_t q = n >> 2;
        F[0][in[q]]++; F[0][in[2 * q]]++; F[0][in[3 * q]]++;
        T[0] += 3;
    }
    uint8_t *cp = out + 9;
    int rle_i = 0;
    for (int i = 0; i < 256; i++) {
        if (!T[i]) continue;
        /* normalise the row to 4095 in double precision (:470-495) */
        double p = (double)X8_TOT / T[i];
        for (;;) {
            int t2 = 0, m = 0, M = 0;
            for (int j = 0; j < 256; j++) {
                if (!F[i][j]) continue;
                if (m < F[i][j]) { m = F[i][j]; M = j; }
                F[i][j] = (int)(F[i][j] * p);
                if (F[i][j] == 0) F[i][j] = 1;
                t2 += F[i][j];
            }
            t2++;
            if (t2 < (int)X8_TOT) { F[i][M] += (int)X8_TOT - t2; break; }
            if (t2 - (int)X8_TOT >= F[i][M] / 2) { p = .98; continue; }             /* :489 */
            F[i][M] -= t2 - (int)X8_TOT;
            break;
        }
        /* context byte with the same run-length shortcut, over T[] (:497-510) */
        if (rle_i) rle_i--;
        else {
            *cp++ = (uint8_t)i;
            if (i && T[i - 1]) {
                int r = i + 1;
                while (r < 256 && T[r]) r++;
                rle_i = r - (i + 1);
                *cp++ = (uint8_t)rle_i;
            }
        }
        cp = put_table(cp, F[i]);
        uint32_t x = 0;
        for (int j = 0; j < 256; j++) { C[i][j] = x; x += (uint32_t)F[i][j]; }
    }
    *cp++ = 0;
    const uint32_t tab = (uint32_t)(cp - out);

    uint8_t *end = out + cap, *p = end;
    uint32_t R[4] = {X8_LOW, X8_LOW, X8_LOW, X8_LOW};
    const uint32_t q = n >> 2;
    /* state k codes quarter k backwards, each byte in the context of its predecessor; state 3 first takes
     * the tail beyond 4q (:552-591).  Per step the states are served in the order 3, 2, 1, 0. */
    for (uint32_t i = n - 1; i > 4 * q - 1; i--) R[3] = enc_put(R[3], &p, C[in[i - 1]][in[i]], (uint32_t)F[in[i - 1]][in[i]]);
    for (uint32_t r = q - 1; r >= 1; r--)
        for (int k = 3; k >= 0; k--) {
            const uint32_t i = (uint32_t)k * q + r;
            R[k] = enc_put(R[k], &p, C[in[i - 1]][in[i]], (uint32_t)F[in[i - 1]][in[i]]);
        }
    for (int k = 3; k >= 0; k--) R[k] = enc_put(R[k], &p, C[0][in[(uint32_t)k * q]], (uint32_t)F[0][in[(uint32_t)k * q]]);
    for (int k = 3; k >= 0; k--) { p -= 4; wr32(p, R[k]); }
    const uint32_t pay = (uint32_t)(end - p);
    *out_size = pay + tab;
    out[0] = 1;
    wr32(out + 1, *out_size - 9);
    wr32(out + 5, n);
    memmove(out + tab, p, pay);
    free(F); free(C);
    return out;
}

static unsigned char *dec1(const uint8_t *in, uint32_t in_size, uint32_t *out_size)
{
    if (in_size < 27 || in[0] != 1) return NULL;                                  /* :711-715 */
    const uint32_t in_sz = rd32(in + 1), out_sz = rd32(in + 5);
    if (in_sz != in_size - 9 || out_sz >= INT_MAX) return NULL;
    const uint8_t *end = in + in_size;
    uint32_t (*start)[256] = calloc(256, sizeof(*start)), (*freq)[256] = calloc(256, sizeof(*freq));
    uint8_t (*sym)[X8_TOT] = calloc(256, sizeof(*sym));       /* slot -> symbol, one row per context byte */
    uint32_t total[256];
    uint8_t has_row[256] = {0};
    uint8_t *out = malloc(out_sz ? out_sz : 1);
    if (!start || !freq || !sym || !out) goto fail;
    {
        /* contexts (:757-823): the symbol / run-length grammar of a table, one level up */
        const uint8_t *cp = in + 9;
        int rle_i = 0;
        int i = *cp++;
        do {
            cp = get_table(cp, end, start[i], freq[i], sym[i], &total[i], 1);
            if (!cp) goto fail;
            has_row[i] = 1;
            if (!rle_i && i + 1 == *cp) { i = *cp++; rle_i = *cp++; }
            else if (rle_i) { rle_i--; i++; if (i > 255) goto fail; }
            else i = *cp++;
        } while (i);
        if (cp > end - 16) goto fail;
        uint32_t R[4];
        for (int k = 0; k < 4; k++, cp += 4) { R[k] = rd32(cp); if (R[k] < X8_LOW) goto fail; }   /* :829-832 */
        const uint32_t q = out_sz >> 2;
        int ctx[4] = {0, 0, 0, 0};
        /* quarters in lock step (:857-905), then state 3 alone over the tail (:908-916) */
        for (uint32_t t = 0; t < q + (out_sz - 4 * q); t++) {
            const int k0 = t < q ? 0 : 3;
            for (int k = k0; k < 4; k++) {
                const int c = ctx[k];
                if (!has_row[c]) goto fail;                     /* context without a table: see the header */
                const uint32_t m = R[k] & (X8_TOT - 1);
                if (m >= total[c]) goto fail;
                const int s = sym[c][m];
                out[t < q ? (uint32_t)k * q + t : 3 * q + t] = (uint8_t)s;
                R[k] = freq[c][s] * (R[k] >> X8_SHIFT) + m - start[c][s];
                ctx[k] = s;
            }
            for (int k = k0; k < 4; k++) R[k] = dec_renorm(R[k], &cp, end);
        }
    }
    free(start); free(freq); free(sym);
    *out_size = out_sz;
    return out;
fail:
    free(start); free(freq); free(sym); free(out);
    return NULL;
}

/* ---- the reference interface, htscodecs/rANS_static.h:41-44, with an orc8_ prefix ------------------------ */

unsigned char *orc8_rans_compress(unsigned char *in, unsigned int in_size, unsigned int *out_size, int order)
{
    if (in_size == 0) return NULL;                                                /* see the header */
    return order ? enc1(in, in_size, out_size) : enc0(in, in_size, out_size);
}

unsigned char *orc8_rans_uncompress(unsigned char *in, unsigned int in_size, unsigned int *out_size)
{
    if (in_size < 9) return NULL;                                                 /* :937 */
    return in[0] ? dec1(in, in_size, out_size) : dec0(in, in_size, out_size);
}
