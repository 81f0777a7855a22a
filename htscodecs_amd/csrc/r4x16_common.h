// r4x16_common.h — definitions shared by the host shim and the gfx950 kernels.
//
// Vocabulary (follows the reference / CRAM, not ML):
//   block  : one independent CRAM EXTERNAL block = one call of rans_(un)compress_to_4x16
//   stream : a bare O0stream / O1stream — table, 4 states, 16-bit words
//   chain  : one of the 4 interleaved rANS states of a stream; a *quad* (4 adjacent lanes of a
//            wave64) runs the 4 chains of one stream in lock-step, up to 16 streams per wave
//   item   : one stream scheduled on the chain kernel (a block's payload, or its RLE meta)
//   image  : the decode / encode lookup tables of one item in the layout the chain kernel reads
#pragma once
#include <stdint.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t  i32;

// container flag bits, rANS_static4x16pr.c:38-43
#define X_ORDER  0x01
#define X_STRIPE 0x08
#define X_NOSZ   0x10
#define X_CAT    0x20
#define X_RLE    0x40
#define X_PACK   0x80

#define RANS_LOW (1u << 15)   // rANS_word.h:63
#define O0_BITS  12           // rANS_static4x16pr.c:81

// rans_compress_bound_4x16, rANS_static4x16pr.c:360-372: same expression, same evaluation order, in double
static inline __host__ __device__ u32 r4x16_bound_hd(u32 size, int order)
{
    int N = order >> 8;
    if (!N) N = 4;
    order &= 0xff;
    double d = (order == 0 ? 1.05 * size + 257 * 3 + 4
                           : 1.05 * size + 257 * 257 * 3 + 4 + 257 * 3 + 4)
             + ((order & X_PACK) ? 1 : 0)
             + ((order & X_RLE) ? 1 + 257 * 3 + 4 : 0) + 20
             + ((order & X_STRIPE) ? 1 + 5 * N : 0);
    int sz = (int)d;
    return (u32)(sz + (sz & 1) + 2);
}

// status codes mirror include/rans4x16_hip.h
#define ST_OK 0
#define ST_CAPACITY 1
#define ST_TRUNCATED 2
#define ST_TABLE 3
#define ST_STATE 4
#define ST_SIZE 5
#define ST_UNSUPPORTED 6
#define ST_CONTEXT 7
#define ST_RLE 8
#define ST_EMPTY 9

// ---------------------------------------------------------------------------------------------
// Decode image (per item): a 5-ary search tree over the cumulative frequencies of each context.
//
//   [0, A)          u16 alpha[n]   : compact symbol index -> byte value | ROW_EMPTY (that symbol's
//                                    own context row has no table), A = 2n rounded up to 16
//   [A, A + R*W)    one row of W bytes per context (order-0: R = 1; order-1: R = n, row of symbol s
//                   at A + s*W).  A row holds, as u16, for cum[0..n] = cumulative starts
//                   (cum[0] = 0, cum[n] = 1 << bits, cum[i > n] = 0xFFFF):
//       2 reads (n <= 50):   root  : cum[10], cum[20], cum[30], cum[40]                (8 bytes)
//                            leaf  : cum[0 .. 10*((n-1)/10) + 11]  (the cum array itself, no copies)
//                            After the root narrows the symbol to the ten of group b, ONE read of the
//                            24 bytes cum[10b .. 10b+11] brings the whole group into registers and the
//                            rest is register work (a packed compare of the even-ranked entries, two
//                            select chains).  Rows are ~112 bytes for 46 symbols.
//       3 reads (n <= 150):  top   : cum[30], cum[60], cum[90], cum[120]               (8 bytes)
//                            mid a : cum[30a+6], cum[30a+12], cum[30a+18], cum[30a+24]   (a = 0..4, 8 bytes each)
//                            leaf  : cum[0 .. n+7]; the group's window cum[30a+6b .. 30a+6b+7] is one
//                                    16-byte read, its entries of rank 2 and 4 are the inner separators.
//                            Half the size of the 4-level rows for text-like alphabets (~100 symbols).
//       4 levels (n <= 256): top   : cum[50], cum[100], .., cum[400]                  (8 separators)
//                            mid a : cum[50a+10], .., cum[50a+40]                        (a = 0..5)
//                            low ab: cum[50a+10b+2], .., cum[50a+10b+8]                 (30 nodes)
//                            leaf  : cum[0 .. n+3]
//
// Lookup of slot m: at each inner level count the separators <= m (one 8-byte LDS read, compare-free
// SWAR count), which narrows the symbol to a pair e, e+1; cum[e..e+2] settles it:
// s = e + (m >= cum[e+1]), start = cum[s], freq = cum[s+1] - cum[s].  Symbols of zero frequency
// have equal neighbours and are skipped by construction.
//
// The reference's reverse tables (rANS_static4x16pr.c:538-549, :985-995) take 1-4 KB per context;
// this takes ~150 bytes for a 46-symbol alphabet whatever the table precision, so a CU's 160 KB of
// LDS holds the tables of ~20 such streams instead of 2 — and resident streams are what the
// decoder's throughput is made of (each stream is only four dependent chains wide).
// ---------------------------------------------------------------------------------------------
#define ROW_EMPTY 0x100u
static inline __host__ __device__ u32 img_levels(u32 n) { return n <= 50 ? 2u : n <= 150 ? 3u : 4u; }
static inline __host__ __device__ u32 img_alpha_bytes(u32 n) { return (2u * n + 15u) & ~15u; }
static inline __host__ __device__ u32 img_leaf_off(u32 lv) { return lv == 2 ? 8u : lv == 3 ? 48u : 304u; }
// u16 entries of the cumulative array.  Two levels: group b is read as the six dwords cum[10b .. 10b+11];
// the last dword of the LAST group is only ever looked at when n is a multiple of ten (otherwise no
// symbol of rank 10b+9 exists), so it is left out then and that read runs into whatever follows
// the row (the next row, or the word ring) — 4 bytes per row that decide whether 14 or 16 streams
// of a 45-symbol alphabet fit a wave's share of LDS.
static inline __host__ __device__ u32 img_leaf_len(u32 n)
{
    if (img_levels(n) == 3) return (n + 9u) & ~1u;       // cum[0 .. n+7]: the last group's window of eight
    if (img_levels(n) != 2) return n + 4u;
    const u32 q = (n - 1u) / 10u;
    return n % 10u == 0 ? 10u * q + 12u : 10u * q + 10u;
}
// rows are 4-byte aligned (two levels) or 8-byte aligned (four levels)
static inline __host__ __device__ u32 img_row_bytes(u32 n)
{
    if (img_levels(n) == 3) return 48u + 2u * img_leaf_len(n);
    return img_levels(n) == 2 ? 8u + 2u * img_leaf_len(n) : (304u + 2u * img_leaf_len(n) + 7u) & ~7u;
}
static inline __host__ __device__ u32 img_bytes(u32 n, u32 rows) { return img_alpha_bytes(n) + rows * img_row_bytes(n); }
// ---------------------------------------------------------------------------------------------
// Packed rows ("level 1"): order-1 streams with 10-bit tables (every total is 1024) and at most 48 symbols -
// the quality alphabets the headline benchmark is made of.  LDS bytes per stream decide how many streams a CU
// holds, and resident streams are the decoder's throughput, so these rows spend 10 bits per entry instead of 16:
//
//   alpha[s] (u16) additionally carries, in bits 9..15, `first` + 2 of the row of context s: `first` = the index of that row's
//   first symbol of non-zero frequency.
//   A row is a run of dwords of three 10-bit fields at bit 0, 11 and 22 (bits 10 and 21 are zero: guard bits for
//   a compare-free "field >= m" test, see lookup in r4x16_decode.hip).  Fields L[0..]:
//       L[0] = 1023 (stands for -1), L[j] = cum[first + j] - 1 for 1 <= j <= n - first, 1023 beyond:
//   the INCLUSIVE end of each symbol's slot range, so that every value fits ten bits (an exclusive end can be
//   1024, a start cannot say where the last symbol ends).  Symbol first + c, c = #{j >= 1 : L[j] < m}, owns slot m;
//   its range is L[c] + 1 .. L[c + 1].  Leading symbols of zero frequency (byte 0 is in every order-1 alphabet
//   and in almost no row) are skipped through `first`; later ones repeat their predecessor's end.
//       root         : L[12], L[24], L[36]               (which group of twelve)
//       leaf dword i : L[3i], L[3i + 1], L[3i + 2]       (group g is leaf dwords 4g .. 4g + 4)
//   Up to 48 symbols (round 4, "layout 2": what a step reads together sits together - the step is a lone wave's
//   instruction stream, and every LDS instruction in it is four issue cycles and a wait):
//       head[s], 8 bytes per context, IN PLACE OF alpha[]: dword 0 = the root of the row of context s, dword 1 = its
//                alpha word (byte value, ROW_EMPTY, first + 2) - the two things a step needs of the symbol it has just
//                decoded, in one 8-byte read;
//       rows     the leaf dwords alone, 16 G bytes per row (16 more when n = 12 G: the dword L[12 G ..]), 16-byte
//                aligned: a group is one 16-byte read and a dword.
//   46 symbols: 368 + 46 x 64 = 3,312 bytes (layout 1, root in front of each row: 3,224), 3,584 with the word ring =
//   fifteen streams in 42 LDS granules exactly; 67.2 -> 63.4 instructions and 6.1 -> 4.4 LDS operations per step,
//   headline decode chain 97.3 -> 93.5 ms.
//   49..96 symbols ("wide"): alpha[] and rows of root + leaf as before.
// ---------------------------------------------------------------------------------------------
#define PK_MAX_NSYM 48u                                  // one root dword: up to four groups of twelve
#define PKW_MAX_NSYM 96u                                 // "wide" packed rows: up to eight groups, 16-byte root
#define PK_FIRST_SHIFT 9                                 // `first` in bits 9..15 of the alpha word
static inline __host__ __device__ u32 pk_groups(u32 n) { return (n + 11u) / 12u; }
// Wide rows (49..96 symbols: what X_PACK makes of a four- or six-letter quality alphabet): the same leaf, and a
// root of eight u16 separators L[12 k] + 1 (0x7fff beyond the last group) counted with the u16 rows' compare-free
// count_le: two 8-byte reads instead of one dword.  80 symbols: 128-byte rows against 224 of the 3-read u16 rows.
static inline __host__ __device__ u32 pk_root_bytes(u32 n) { return n > PK_MAX_NSYM ? 16u : 4u; }
// (the last dword, L[12G ..], can only be selected when the alphabet fills its last group: left out otherwise, and
//  that read runs into the next row's root or the word ring - as with the u16 rows' last dword)
static inline __host__ __device__ u32 pk_head_bytes(u32 n) { return n > PK_MAX_NSYM ? img_alpha_bytes(n) : (8u * n + 15u) & ~15u; }
static inline __host__ __device__ u32 pk_row_bytes(u32 n)
{
    return n > PK_MAX_NSYM ? pk_root_bytes(n) + 4u + 16u * pk_groups(n) - (n % 12u ? 4u : 0u) : 16u * pk_groups(n) + (n % 12u ? 0u : 16u);
}
static inline __host__ __device__ u32 pk_img_bytes(u32 n) { return pk_head_bytes(n) + n * pk_row_bytes(n); }

// ---------------------------------------------------------------------------------------------
// Direct rows ("level 6"): the short-step route for batches that leave LDS to spare.  The rows above trade
// instructions for LDS bytes (a search over cumulative frequencies: ~35 vector instructions per symbol) because resident
// streams are the throughput of a FULL chip; a batch below one round of resident streams has the LDS of the idle
// slots to spend instead, and then the reference's own shape (ssym[ctx][m] + fb[ctx][sym], rANS_static4x16pr.c:538-549,
// :985-995) is the faster step - one byte read for the symbol, one read for (start, freq), no search:
//
//   per context a block of  fb[n + 1] (u32)  then  tab[T] (u8),  T = 1 << (look - 1):
//     fb[r]  = entry of the r-th symbol of this row that HAS a frequency; fb[nnz] = a copy of fb[nnz - 1]
//     tab[j] = rank r of the symbol that owns slot 2j.  Slot 2j + 1 belongs to the same symbol or to the next one
//              with a frequency, which then STARTS there: one 8-byte read brings fb[r] and fb[r + 1].
//   Entry, with sh = 32 - look:   start << sh  |  ~F & (2^sh - 1),   F = idx | freq << 7 [| DIR_EMPTY]
//   (compact symbol index in 7 bits - direct rows serve alphabets of up to 128 symbols -, freq <= 4096 in 13; bit 20
//   is free when look = 10).  The fields under `start` are stored COMPLEMENTED so that one subtraction from
//   M = m << sh | 2^sh - 1  (no borrow ever leaves the low part) gives  D = (m - start) << sh | F : the offset inside
//   the symbol's range and its fields at once.  Entries ascend with `start`, the first one never exceeds M, so
//   "the last entry that starts at or below m" is the unsigned minimum of the two differences (an entry beyond m
//   wraps to a huge one; the copy after the last entry ties): two subtractions and a minimum, no compare, no select.
//   Half-resolution tab: 512 bytes per context for 10-bit tables - a 46-symbol order-1 table takes 32 KB, four
//   streams per CU, 1,024 per chip, against 55 KB with the reference's full-resolution ssym.
// An empty row (context without a table) is one entry owning every slot with freq = 1 << look: the state is left as
// it is and the stream is failed through DIR_EMPTY (10-bit tables) or the ROW_EMPTY flag of alpha[], as with the other
// row kinds.
// Output bytes: alpha[idx] is one more LDS read per symbol - unless the alphabet is AFFINE, byte = idx + c for every
// symbol that has a frequency anywhere (quality values are a run of consecutive byte values; byte 0, which every
// order-1 alphabet lists, has none): then four indices are turned into four bytes by one masked add (DecItem.affine).
// ---------------------------------------------------------------------------------------------
#define DIR_MAX_NSYM 128u
#define DIR_EMPTY (1u << 20)
static inline __host__ __device__ u32 dir_entry(u32 start, u32 freq, u32 idx, u32 look, u32 flags)
{
    const u32 sh = 32u - look;
    return (start << sh) | (~(idx | (freq << 7) | flags) & ((1u << sh) - 1u));
}
static inline __host__ __device__ u32 dir_fb_bytes(u32 n) { return 4u * (n + 1u); }
static inline __host__ __device__ u32 dir_blk_bytes(u32 n, u32 look) { return dir_fb_bytes(n) + (1u << (look - 1u)); }
static inline __host__ __device__ u32 dir_img_bytes(u32 n, u32 rows, u32 look) { return img_alpha_bytes(n) + rows * dir_blk_bytes(n, look); }

// ---------------------------------------------------------------------------------------------
// Mid rows ("level 10", round 4): the short-ish step for batches of ONE partly filled round - more streams than the
// direct rows' 32 KB images allow (four per CU), fewer than would fill the chip with packed rows (45 per CU), e.g. the
// sixteen 1 MiB blocks per CU of a 4,096-block batch, which paid the packed rows' 453-cycle step with two thirds of the
// LDS empty.  Order-1 streams with 10-bit tables, per context:
//     idx[64] (u8)  one entry per bucket of sixteen slots: e = (owner of slot 16 j) & ~1, | 0x80 if the symbols that can
//                   own a slot of the bucket do not all lie in cum[e .. e + 6] (more than a handful of symbols that
//                   share sixteen slots: rare, takes a scan)
//     cum[n + 10] (u16)  the cumulative starts themselves, cum[n] = 1024, 0x7fff beyond
// A lookup is two dependent LDS reads like the direct rows' - the bucket's entry, then ONE 16-byte window cum[e .. e + 7],
// whose entries <= m are counted without compares (the u16 rows' SWAR count) - at a quarter of their footprint:
// 172 bytes per context for 46 symbols, 8.3 KB per stream with alphabet and word ring, sixteen streams per CU.
// ---------------------------------------------------------------------------------------------
#define MID_MIN_NSYM 13u
#define MID_MAX_NSYM 64u
#define MID_OVF 0x80u
static inline __host__ __device__ u32 mid_cum_len(u32 n) { return (n + 10u) & ~1u; }          // u16 entries, even: rows stay 4-byte aligned
static inline __host__ __device__ u32 mid_row_bytes(u32 n) { return 64u + 2u * mid_cum_len(n); }
static inline __host__ __device__ u32 mid_img_bytes(u32 n) { return img_alpha_bytes(n) + n * mid_row_bytes(n); }

#define IMG_O0_BYTES  1344u                       // 256 symbols, one row
#define IMG_MAX_BYTES (512u + 256u * 824u)        // 256 symbols, 256 rows

// One stream for the chain decoder.  80 bytes.
struct DecItem {
    u64 words;       // device address of the first 16-bit word (just after the 4 states)
    u64 out;         // device address of the first decoded byte
    u64 image;       // device address of the image; row of context 0 / the only row is at +0
    u32 words_len;   // bytes from `words` to the end of the stream's input window
    u32 out_sz;      // symbols to produce
    u32 R[4];        // initial states
    u32 img_bytes;   // size of the image
    u32 look;        // bits looked up per symbol: 12, or 10 (rANS_static4x16pr.c:1027, :1071)
    u32 order;       // 0: byte i on chain i&3;  1: chain k owns quarter k (+tail on chain 3)
    u32 active;      // 0 = nothing to do (failed block, CAT, empty)
    u32 blk;         // owning block (errors are reported there)
    u32 nsym;        // compact alphabet size n (decides the tree depth and the row size)
    u32 packed;      // 0: u16 rows of img_levels(nsym) levels, 1: packed 10-bit rows (level 1 / 5), 2: direct rows (level 6), 3: mid rows (level 10)
    u32 affine;      // direct rows: c + 1 when byte = compact index + c for every symbol with a frequency, else 0
};
static inline __host__ __device__ u32 item_levels(u32 nsym, u32 packed)
{
    return packed == 3u ? 10u : packed == 2u ? 6u : packed ? (nsym > PK_MAX_NSYM ? 5u : 1u) : img_levels(nsym);
}

// Per-block record of the decode pipeline.
struct DecDesc {
    i32 status;
    u32 flags;
    u32 osz;         // final size of the block
    u32 s1_size;     // bytes produced by the entropy stage (or copied for CAT)
    u64 s1;          // where the entropy stage writes
    u64 cat_src;     // CAT: source of the raw copy (0 if none)
    u32 cat_len;
    u32 pack_per;    // symbols per byte (8,4,2), 0 = constant, 1 = copy
    u8  pack_map[16];
    u32 rle_meta_len;    // decoded meta length
    u32 rle_meta_raw;    // 1: meta bytes live in the input at rle_meta; 0: decoded into workspace
    u64 rle_meta;        // device address of meta (nsyms, syms, run varints)
    u64 s2, s3;          // stage buffers after un-RLE / un-PACK
    u32 pad[2];
};

// ---------------------------------------------------------------------------------------------
// Encode image (per item), compact:
//   [0,256)                 u8  idx_of[256] : byte value -> compact symbol index (order-0: identity)
//   [256, 256+2*R*(ns+1))   u16 cum[R][ns+1]: cumulative starts of context row r (cum[r][ns] = 1 << bits);
//                           order-1: R = ns = size of the alphabet (byte 0 included, index 0);
//                           order-0: R = 1, ns = 256.
// Everything the coder needs for symbol s in context r follows from cum[r][s] and cum[r][s+1]:
// start, freq, x_max = freq << (31 - bits), the reciprocal shift ceil(log2 freq) - 1, and the
// reciprocal itself (rANS_word.h:252) from ONE 16 KB table rcptab[freq] shared by every stream (it
// stays L1-resident).  None of that is on the dependent path (symbols are known in advance), so two
// bytes per (context, symbol) are enough: a 46-symbol order-1 image is 4.6 KB and ~32 streams'
// tables fit one CU's LDS, against 2 with the reference's 24-byte RansEncSymbol records.
// ---------------------------------------------------------------------------------------------
#define ENC_IMG_IDX    256u
#define ENC_IMG_MAIN   132096u                                    // 256 + 2*256*257, rounded to 256
#define ENC_IMG_O0     1024u                                      // 256 + 2*257, rounded
#define ENC_IMG_NESTED ENC_IMG_MAIN                               // offset of the nested-table image
#define ENC_IMG_META   (ENC_IMG_NESTED + ENC_IMG_O0)              // offset of the RLE-meta image
#define ENC_IMG_META_BYTES 4352u                                  // one cumulative row, or 256 symbol records (enc_rec_img_bytes(256, 1))
#define ENC_IMG_BYTES  (ENC_IMG_META + ENC_IMG_META_BYTES)        // 137,472 per block
#define RCPTAB_ENTRIES 4097u

struct EncItem {
    u64 data;        // device address of the bytes to code
    u64 image;       // device address of the image
    u64 scratch_end; // device address one past the end of this item's backward-write area (even)
    u32 n;           // number of bytes
    u32 bits;        // 12 (order-0) or 10/12
    u32 order;
    u32 active;
    u32 pay_len;     // OUT: bytes written backwards (states + words)
    u32 blk;
    u32 ns;          // symbols per image row (a row is ns+1 u16)
    u32 img_bytes;   // bytes of the image (what must sit in LDS)
    u32 packed;      // 0: u16 rows, 1: rows are 11-bit bit streams (below), 2: 16-byte symbol records (further below)
    u32 affine;      // symbol records: c + 1 when compact index = byte - c for every byte of the data (no idx_of[] look-up), else 0
};
// Packed encoder rows: order-1 streams with 10-bit tables and 20..64 symbols (the quality alphabets).  Row r is a
// bit stream of 11-bit entries cum[r][0..ns] (11 bits hold the total 1024), entry j at bit 11 j; W = ceil(11 (ns + 1)
// / 32) dwords per row.  A symbol's (start, next) pair is 22 bits out of two adjacent dwords (one ds_read2 at a
// 4-byte aligned address and a funnel shift).  46 symbols: 68-byte rows instead of 94, 3.5 KB per stream instead
// of 4.7: 45 streams per CU - the decoder's count, so that a batch is a whole number of rounds for both.
#define ENC_PK_MIN_NS 20u
#define ENC_PK_MAX_NS 64u
static inline __host__ __device__ u32 enc_pk_row_dwords(u32 ns) { return (11u * (ns + 1u) + 31u) / 32u; }
static inline __host__ __device__ u32 enc_pk_img_bytes(u32 ns) { return ENC_IMG_IDX + 4u * ns * enc_pk_row_dwords(ns) + 4u; }

// Symbol records ("kind 2"): the encoder's short-step route for batches that leave LDS to spare - the twin of the
// decoder's direct rows.  The u16 / packed rows above make the coder derive start, freq, x_max, the reciprocal (one more
// LDS read) and its shift from two cumulative values per symbol: ~50 instructions per symbol, each four cycles of a
// lone wave's issue.  With LDS to spend, the image holds what the reference's RansEncSymbolInit (rANS_word.h:190-266)
// precomputes, one 16-byte record per (context, symbol), read with one ds_read_b128:
//     { rcp_freq,  x_max = freq << (31 - bits),  bias,  cmpl_freq | rcp_shift << 24 }
// (freq == 1: rcp_freq = 2^32 - 1, rcp_shift = 0, bias = start + (1 << bits) - 1, as there), and a step is
//     if (x >= x_max) emit; q = mulhi(x, rcp_freq) >> rcp_shift; x += bias + q * cmpl_freq      (:281-321)
// with the multiply by cmpl_freq a 24-bit one that ignores the shift in the top byte.  Image: idx_of[256], then
// rec[R][ns]: 34 KB for a 46-symbol order-1 table (four streams per CU), 4.4 KB for an order-0 one.
#define ENC_RING_BYTES 144u          // per stream behind its image: the 128-byte ring of emitted words + a dump slot (+ pad)
static inline __host__ __device__ u32 enc_rec_img_bytes(u32 ns, u32 rows) { return ENC_IMG_IDX + 16u * rows * ns; }

// Per-block record of the encode pipeline.
struct EncDesc {
    i32 status;
    u32 flags;       // first stream byte as decided so far
    u32 hdr_len;     // bytes of hdr[] in use (flag byte, size, pack meta ...)
    u8  hdr[44];
    u64 data;        // bytes handed to the entropy stage (after PACK / RLE)
    u32 dlen;
    u32 nosz;
    u32 cat;         // 1: copy `data` raw (explicit X_CAT)
    u32 tab_len;     // bytes of table (incl. order-1 header byte) staged in tab[]
    u64 tab;         // device address of the staged table bytes
    u32 nest_on;     // 1: the order-1 table is also coded as an order-0 stream (third chain item); k_enc_finish picks
    u32 nest_tab_len;//    the smaller form (rANS_static4x16pr.c:766-780).  Its own order-0 table: nest_tab[0..nest_tab_len)
    u64 nest_tab;
    u32 rle_on, rle_mlen, rle_lits;   // RLE meta bookkeeping (meta raw bytes at rle_meta)
    u32 meta_tab_len;
    u64 rle_meta;
    u64 meta_tab;
};

// ---------------------------------------------------------------------------------------------
// Device workspace carved per chunk of blocks by r4x16_api.hip.
// ---------------------------------------------------------------------------------------------
#define TBUF_BYTES     204800u                          // an un-nested order-1 table (257*257*3 = 198147 max)
#define IMG_META_BYTES 2832u                      // the RLE-meta image: one order-0 row, or its direct rows (dir_img_bytes(128, 1, 12) = 2,820)
#define DEC_IMG_SLOT   (IMG_MAX_BYTES + IMG_O0_BYTES + IMG_META_BYTES)   // payload image, nested-table image, RLE-meta image
#define TAB_BYTES      198656u                          // >= 1 + 257*257*3 (assert at rANS_static4x16pr.c:784)

#define CLS_MAX  64u
#define CLS_NONE 0xffffffffu
// streams grouped by LDS size class and ordered by chain length on the device (r4x16_sched.h)
#define SCHED_NB   256u                         // length buckets per class (8 per octave: 240 used)
#define SCHED_BINS (CLS_MAX * SCHED_NB)
// per-class arrays of SchedWs.cnt, CLS_MAX entries each
#define SCHED_COUNT 0u                          // streams of the class
#define SCHED_START (1u * CLS_MAX)              // first position in the list
#define SCHED_CLAIM (2u * CLS_MAX)              // next share to hand out
#define SCHED_SEATS (3u * CLS_MAX)              // workgroups of the class's launch that may work
#define SCHED_TSTART (4u * CLS_MAX)             // when the class's first workgroup started / its last one ended (low dword of the
#define SCHED_TEND  (5u * CLS_MAX)              // 100 MHz clock): how long the class really took, for the next batch's plan
#define SCHED_CNT_WORDS (6u * CLS_MAX)

struct SchedWs {
    u32 *key;        // [nitems]  class << 8 | 255 - bucket, or CLS_NONE
    u32 *list;       // [nitems]  item indices, by class, longest first
    u32 *cnt;        // [SCHED_CNT_WORDS]
    u32 *bins;       // [2 * SCHED_BINS]  per (class, bucket): count -> start, fill cursor
    u64 *work;       // [2 * CLS_MAX]  per class: sum of chain lengths, then the longest chain
};
// An order-1 block whose table is itself an order-0 stream (rANS_static4x16pr.c:944-955): k_dec_front<0> hands that
// stream to the chain kernel as an item of its own and leaves what k_dec_front<1> needs to carry on from the decoded
// table bytes.
struct DecResume { u32 pending, pay_pos, pay_len, s1_size, bits, usz, after_table, pad; };

struct DecWs {
    DecDesc *desc;     // [nblk]
    DecItem *items;    // [3*nblk]   [b] = payload stream of block b, [nblk+b] = its RLE meta stream, [2*nblk+b] = its nested table
    DecResume *resume; // [nblk]
    u8 *images;        // [nblk][DEC_IMG_SLOT]
    u8 *tbuf;          // [nblk][TBUF_BYTES]
    // X_PACK / X_RLE staging: a region per block that carries one of the flags, sized from the block's own output
    // capacity and laid out on the device (dec_var_bytes, k_dec_voff): block b owns var[voff[b] .. voff[b + 1])
    u8 *var;           // stage buffer of the inverse transforms, then the decoded run-length meta
    u64 *voff;         // [nblk + 1]
    u64 var_bytes;     // bytes behind `var` (a block whose region would end beyond them reports UNSUPPORTED)
    u32 max_out_cap;   // the caller's bound on a transformed block's size
    u32 pad2;
    // streams grouped by LDS size class on the device (k_dec_classify, r4x16_sched.hip), so that every chain
    // workgroup gets a full set of streams of its class whatever the mix of blocks in the batch
    SchedWs sched;     // key / list: [2*nblk]  (the nested tables' pass uses the first nblk)
    u32 direct_budget; // LDS bytes a stream of this batch may take for direct rows (0: never); set per chunk by the host
    u32 mid_budget;    // the same for mid rows (level 10)
};


// Per-block staging of the transforms (host bound and device layout use the same arithmetic).
static inline __host__ __device__ u64 var_align(u64 v) { return (v + 255u) & ~(u64)255u; }
struct EncVar { u64 packed, lits, meta, scratch2, total; };      // offsets inside the block's region, and its size
static inline __host__ __device__ EncVar enc_var_layout(u32 n, int order)
{
    EncVar v = {0, 0, 0, 0, 0};
    if ((order & X_CAT) || !(order & (X_PACK | X_RLE))) return v;
    u64 at = 0;
    if (order & X_PACK) { v.packed = at; at += var_align((u64)n + 64u); }
    if (order & X_RLE) {
        v.lits = at; at += var_align((u64)n + 64u);
        v.meta = at; at += var_align((u64)n + 64u + 768u);
        v.scratch2 = at; at += var_align((u64)r4x16_bound_hd(n + 768u, 0) + 64u);
    }
    v.total = at;
    return v;
}
// host: an upper bound of the sum of enc_var_layout(..).total over `nblk` blocks of `total_in` bytes together
static inline u64 enc_var_bound(u64 nblk, u64 total_in) { return nblk * 4096u + total_in * 4u + total_in / 16u + (1u << 20); }
static inline __host__ __device__ u64 dec_var_tmp(u32 cap) { return var_align((u64)cap + 64u); }
static inline __host__ __device__ u64 dec_var_bytes(u32 cap) { return dec_var_tmp(cap) + var_align((u64)cap + 512u); }
static inline u64 dec_var_bound(u64 nblk, u64 total_out) { return nblk * 1536u + 2u * total_out + (1u << 20); }

// What the histogram kernel hands to the table kernel (per block).
struct EncStat {
    u32 F0[256];        // byte histogram of the data handed to the entropy stage
    u8  present[256];   // order-1 alphabet F0 (byte 0 forced in)
    u8  idx_of[256];    // byte -> compact index
    u8  alpha[256];     // compact index -> byte
    u32 ns;             // alphabet size
    u32 run;            // 0: nothing to do for this block (failed, CAT, empty)
    u32 order;          // 0 / 1 after the "fewer than 8 bytes" rule
    u32 pad;
};

struct EncWs {
    EncDesc *desc;      // [nblk]
    EncItem *items;     // [3*nblk]  [b] = payload stream of block b, [nblk+b] = its RLE meta stream, [2*nblk+b] = its order-1 table
    u8 *images;         // [nblk][ENC_IMG_BYTES]
    u8 *tab;            // [nblk][TAB_BYTES]  table bytes as they go into the stream
    u8 *scratch;        // [nblk][scratch_stride]  backward-written states + words
    const double *logtab;   // [2][257]  log(1024+k), log(4096+k) from the host libm (:651-652)
    const u32 *rcptab;      // [4097]    reciprocal by frequency (rANS_word.h:252), shared by all streams
    u64 scratch_stride;
    // X_PACK / X_RLE staging: a region per block that asks for a transform, sized from the block's own length and laid
    // out on the device (enc_var_bytes, k_enc_voff): block b owns var[voff[b] .. voff[b + 1]) - bit-packed bytes, RLE
    // literals (filled from the end), RLE meta (nsyms, syms, run varints; filled from the end), the backward-written
    // meta stream.  var == nullptr when the batch cannot use transforms.
    u8 *var;
    u64 *voff;          // [nblk + 1]
    u64 var_bytes;      // bytes behind `var` (a block whose region would end beyond them reports UNSUPPORTED)
    u8 *metatab;        // [nblk][1024]        order-0 table of the meta stream
    EncStat *stat;      // [nblk]
    u8 *dump;           // [ENC_DUMP_BYTES]  target of the chain coder's idle output slots (never read)
    SchedWs sched;      // streams grouped by LDS size class (as in DecWs); key / list: [3*nblk]
    u32 direct_budget;  // LDS bytes a stream of this batch may take for symbol records (0: never); set per chunk by the host
    u32 meta_records;   // the RLE-meta streams take records too (set where the class launches go out in stream order: a wave then
                        // walks a block's literals and its run lengths together; side by side they are better off in two kernels)
    u32 pad;
};
#define META_TAB_BYTES 1024u
#define ENC_F_BYTES    262144u                          // 256 x 256 pair counters (bottom of a block's scratch area)
#define ENC_DUMP_BYTES 65536u

