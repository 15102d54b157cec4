/* mz_pack.c -- host side of the link formats of mz_yama_batch() (mz_host.c).
 *
 * The host-buffer path is bounded by the PCIe link and by host memory traffic, not by the kernels (DESIGN.md
 * section 5), so what crosses the link is what the device needs and nothing more:
 *
 *   up    the columns of A and B as byte CLASSES, two per byte.  The recurrence (reference mz_yama.c:113-242 through
 *         the tables of mz_scores.c:34-81) only distinguishes A/a, C/c, G/g, T/t, '-' and "anything else"; the device
 *         expands a nibble to a canonical letter of its class (k_unnib) and the kernels run unchanged;
 *         the band bounds as one byte per row -- LB[i]-LB[i-1] in the low nibble, RB[i]-RB[i-1] in the high one --
 *         where every step is 0..15, as two bytes per row where every step is 0..255, raw int32 otherwise (k_unband);
 *   down  the edit script, two bits per merged column (C = 0, I = 1, D = 2: reference mz_yama.c:24-26), in column
 *         order.  The merged columns themselves (reference mz_yama.c:293-313: C -> A-column over B-column, I -> dashes
 *         over B-column, D -> A-column over dashes) are assembled here from the caller's own A and B -- bytes the
 *         host already holds do not travel back.
 *
 * AVX2 forms are picked at run time (function multiversioning by hand: the library is built for plain x86-64).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#include <pthread.h>

#include "mz_pack.h"

/* ------------------------------------------------------------------------------------------------ byte classes */

static uint8_t g_cls[256];
static uint8_t g_dash[512] __attribute__((aligned(16)));
static int g_avx2;

__attribute__((constructor)) static void pack_init(void)
{
    int i;
    for (i = 0; i < 256; ++i) g_cls[i] = 5;
    g_cls['A'] = g_cls['a'] = 0; g_cls['C'] = g_cls['c'] = 1; g_cls['G'] = g_cls['g'] = 2; g_cls['T'] = g_cls['t'] = 3;
    g_cls['-'] = 4;
    memset(g_dash, '-', sizeof g_dash);
    __builtin_cpu_init();
    g_avx2 = __builtin_cpu_supports("avx2") && !(getenv("MZ_NO_AVX2") && atoi(getenv("MZ_NO_AVX2")));
}

static void classes_scalar(const uint8_t *src, size_t n, uint8_t *dst)
{
    size_t i;
    for (i = 0; i + 1 < n; i += 2) dst[i >> 1] = (uint8_t)(g_cls[src[i]] | (g_cls[src[i + 1]] << 4));
    if (n & 1) dst[n >> 1] = (uint8_t)(g_cls[src[n - 1]] | 0x50);
}

__attribute__((target("avx2"))) static inline __m256i classes32(__m256i v)
{
    const __m256i u = _mm256_or_si256(v, _mm256_set1_epi8(0x20));           /* case folded, as the device's byte_class() */
    __m256i c = _mm256_set1_epi8(5);
    c = _mm256_sub_epi8(c, _mm256_and_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('a')), _mm256_set1_epi8(5)));
    c = _mm256_sub_epi8(c, _mm256_and_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('c')), _mm256_set1_epi8(4)));
    c = _mm256_sub_epi8(c, _mm256_and_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('g')), _mm256_set1_epi8(3)));
    c = _mm256_sub_epi8(c, _mm256_and_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('t')), _mm256_set1_epi8(2)));
    c = _mm256_sub_epi8(c, _mm256_and_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('-')), _mm256_set1_epi8(1)));
    return c;
}

__attribute__((target("avx2"))) static void classes_avx2(const uint8_t *src, size_t n, uint8_t *dst)
{
    const __m256i w = _mm256_set1_epi16(0x1001);                            /* bytes (1, 16): even + 16 * odd */
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m256i c0 = classes32(_mm256_loadu_si256((const __m256i *)(src + i)));
        const __m256i c1 = classes32(_mm256_loadu_si256((const __m256i *)(src + i + 32)));
        const __m256i m0 = _mm256_maddubs_epi16(c0, w), m1 = _mm256_maddubs_epi16(c1, w);
        const __m256i r = _mm256_permute4x64_epi64(_mm256_packus_epi16(m0, m1), 0xD8);
        _mm256_storeu_si256((__m256i *)(dst + (i >> 1)), r);
    }
    classes_scalar(src + i, n - i, dst + (i >> 1));
}

/* n bytes of column text -> (n + 1) / 2 bytes of class nibbles (byte i = class[2i] | class[2i+1] << 4) */
void mz_pack_classes(const uint8_t *src, size_t n, uint8_t *dst)
{
    if (g_avx2 && n >= 64) classes_avx2(src, n, dst);
    else classes_scalar(src, n, dst);
}

/* Streaming stores.  Everything this file writes -- the staging block, the merged columns -- is written once, in order,
 * and read next by somebody else (the DMA engine, the caller), so it is produced in a small buffer that stays in L1
 * and goes to memory with non-temporal stores: a plain store to a new cache line first READS the line, which on a path
 * bound by host memory bandwidth is a third of its traffic (C4: 3 GB of writes per call). */
static inline void stream_lines(uint8_t *dst, const uint8_t *src, size_t n)      /* dst 16-byte aligned, n a multiple of 16 */
{
    size_t i;
    for (i = 0; i < n; i += 16) _mm_stream_si128((__m128i *)(dst + i), _mm_loadu_si128((const __m128i *)(src + i)));
}

__attribute__((target("avx2"))) static size_t classes_avx2_stream(const uint8_t *src, size_t n, uint8_t *dst)
{
    const __m256i w = _mm256_set1_epi16(0x1001);
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m256i c0 = classes32(_mm256_loadu_si256((const __m256i *)(src + i)));
        const __m256i c1 = classes32(_mm256_loadu_si256((const __m256i *)(src + i + 32)));
        const __m256i m0 = _mm256_maddubs_epi16(c0, w), m1 = _mm256_maddubs_epi16(c1, w);
        _mm256_stream_si256((__m256i *)(dst + (i >> 1)), _mm256_permute4x64_epi64(_mm256_packus_epi16(m0, m1), 0xD8));
    }
    return i;
}

/* the same to a 32-byte aligned slot of `slot` bytes (a multiple of 32, >= (n + 1) / 2), padded with class "other" */
void mz_pack_classes_stream(const uint8_t *src, size_t n, uint8_t *dst, size_t slot)
{
    uint8_t tail[64] __attribute__((aligned(32)));
    size_t done = 0, rest;
    if (g_avx2) done = classes_avx2_stream(src, n, dst);
    else for (; done + 64 <= n; done += 64) { classes_scalar(src + done, 64, tail); stream_lines(dst + (done >> 1), tail, 32); }
    rest = n - done;                                                  /* < 64 bytes: at most one more 32-byte line */
    if ((done >> 1) < slot) {
        memset(tail, 0x55, sizeof tail);
        classes_scalar(src + done, rest, tail);
        stream_lines(dst + (done >> 1), tail, slot - (done >> 1) < 32 ? slot - (done >> 1) : 32);
        for (done = (done >> 1) + 32; done < slot; done += 32) { memset(tail, 0x55, 32); stream_lines(dst + done, tail, 32); }
    }
}

/* ------------------------------------------------------------------------------------------------ band bounds */

/* LB[0..M], RB[0..M] -> LB[0], RB[0] (int32) + M bytes (LB step | RB step << 4).  Returns the OR of all steps (as
 * unsigned: a negative step sets the high bits): the format is valid iff the result is < 16. */
static uint32_t band_nib_scalar(const int *LB, const int *RB, int from, int M, uint8_t *dst)
{
    uint32_t acc = 0;
    int i;
    for (i = from; i <= M; ++i) {
        const uint32_t dl = (uint32_t)(LB[i] - LB[i - 1]), dr = (uint32_t)(RB[i] - RB[i - 1]);
        acc |= dl | dr;
        dst[i - 1] = (uint8_t)(dl | (dr << 4));
    }
    return acc;
}

__attribute__((target("avx2"))) static uint32_t band_nib_avx2(const int *LB, const int *RB, int M, uint8_t *dst)
{
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    __m256i acc = _mm256_setzero_si256();
    uint32_t a[8], r = 0;
    int i = 1, k;
    for (; i + 7 <= M; i += 8) {
        const __m256i dl = _mm256_sub_epi32(_mm256_loadu_si256((const __m256i *)(LB + i)), _mm256_loadu_si256((const __m256i *)(LB + i - 1)));
        const __m256i dr = _mm256_sub_epi32(_mm256_loadu_si256((const __m256i *)(RB + i)), _mm256_loadu_si256((const __m256i *)(RB + i - 1)));
        const __m256i x = _mm256_shuffle_epi8(_mm256_or_si256(dl, _mm256_slli_epi32(dr, 4)), pick);
        acc = _mm256_or_si256(acc, _mm256_or_si256(dl, dr));
        *(uint32_t *)(dst + i - 1) = (uint32_t)_mm256_cvtsi256_si32(x);
        *(uint32_t *)(dst + i + 3) = (uint32_t)_mm256_extract_epi32(x, 4);
    }
    _mm256_storeu_si256((__m256i *)a, acc);
    for (k = 0; k < 8; ++k) r |= a[k];
    return r | band_nib_scalar(LB, RB, i, M, dst);
}

uint32_t mz_pack_band_nib(const int *LB, const int *RB, int M, uint8_t *dst)
{
    ((int32_t *)dst)[0] = LB[0]; ((int32_t *)dst)[1] = RB[0];
    return (g_avx2 && M >= 16) ? band_nib_avx2(LB, RB, M, dst + 8) : band_nib_scalar(LB, RB, 1, M, dst + 8);
}

/* the M step bytes alone (LB[0], RB[0] travel in the chunk's header) to a 32-byte aligned slot of `slot` bytes (a
 * multiple of 32, >= M), through streaming stores; the OR of all steps as above */
uint32_t mz_pack_band_nib_stream(const int *LB, const int *RB, int M, uint8_t *dst, size_t slot)
{
    uint8_t buf[1024 + 32] __attribute__((aligned(32)));
    uint32_t acc = 0;
    size_t out = 0;
    int i0;
    for (i0 = 0; i0 < M; i0 += 1024) {                       /* rows i0+1 .. i0+m */
        const int m = M - i0 < 1024 ? M - i0 : 1024;
        size_t nb = ((size_t)m + 31) & ~(size_t)31;
        memset(buf + (m & ~31), 0, 32);
        acc |= (g_avx2 && m >= 16) ? band_nib_avx2(LB + i0, RB + i0, m, buf) : band_nib_scalar(LB + i0, RB + i0, 1, m, buf);
        if (out + nb > slot) nb = slot - out;
        stream_lines(dst + out, buf, nb);
        out += nb;
    }
    memset(buf, 0, 32);
    for (; out < slot; out += 32) stream_lines(dst + out, buf, 32);
    return acc;
}

/* the byte form: LB[0], RB[0], M bytes of LB steps, M bytes of RB steps (every step 0..255: the caller has checked) */
void mz_pack_band_bytes(const int *LB, const int *RB, int M, uint8_t *dst)
{
    int i;
    ((int32_t *)dst)[0] = LB[0]; ((int32_t *)dst)[1] = RB[0];
    for (i = 1; i <= M; ++i) { dst[8 + i - 1] = (uint8_t)(LB[i] - LB[i - 1]); dst[8 + M + i - 1] = (uint8_t)(RB[i] - RB[i - 1]); }
}

/* ------------------------------------------------------------------------------------------------ merged columns */

#define OP_AT(s, m) (((s)[(m) >> 2] >> (2 * ((m) & 3))) & 3u)

/* columns [m0, m1) of a pair into o (m0 a multiple of 4); *pa / *pb: the next unread column of A / B.
 * K and L are compile-time constants at every call site (the switch below): the copies become plain moves */
static inline __attribute__((always_inline)) void assemble_small(const int K, const int L, const uint8_t **pa, const uint8_t **pb,
                                                                 const uint8_t *s, int m0, int m1, uint8_t *o)
{
    const int W = K + L;
    const uint8_t *a = *pa, *b = *pb;
    int m = m0;
    for (; m + 4 <= m1; m += 4) {
        unsigned x = s[m >> 2];
        int k;
        for (k = 0; k < 4; ++k, x >>= 2) {
            const unsigned op = x & 3u;
            const int ta = op != 1u, tb = op != 2u;                  /* C and D take a column of A, C and I one of B */
            memcpy(o, ta ? a : g_dash, (size_t)K);
            memcpy(o + K, tb ? b : g_dash, (size_t)L);
            a += ta ? K : 0; b += tb ? L : 0; o += W;
        }
    }
    for (; m < m1; ++m) {
        const unsigned op = OP_AT(s, m);
        const int ta = op != 1u, tb = op != 2u;
        memcpy(o, ta ? a : g_dash, (size_t)K);
        memcpy(o + K, tb ? b : g_dash, (size_t)L);
        a += ta ? K : 0; b += tb ? L : 0; o += W;
    }
    *pa = a; *pb = b;
}

/* any K, L <= 255: 16 bytes at a time, running over the end of a column into the next one's place (written after it;
 * `o` has 32 spare bytes behind the last column) wherever the READ stays inside A and B; exact copies near their ends */
static void assemble_any(int K, int L, const uint8_t **pa, const uint8_t *aend, const uint8_t **pb, const uint8_t *bend,
                         const uint8_t *s, int m0, int m1, uint8_t *o)
{
    const int W = K + L, Kr = (K + 15) & ~15, Lr = (L + 15) & ~15;
    const uint8_t *a = *pa, *b = *pb;
    int m, k;
    for (m = m0; m < m1; ++m) {
        const unsigned op = OP_AT(s, m);
        const int ta = op != 1u, tb = op != 2u;
        const uint8_t *sa = ta ? a : g_dash, *sb = tb ? b : g_dash;
        if ((!ta || a + Kr <= aend) && (!tb || b + Lr <= bend)) {
            for (k = 0; k < Kr; k += 16) _mm_storeu_si128((__m128i *)(o + k), _mm_loadu_si128((const __m128i *)(sa + k)));
            for (k = 0; k < Lr; k += 16) _mm_storeu_si128((__m128i *)(o + K + k), _mm_loadu_si128((const __m128i *)(sb + k)));
        } else {
            memcpy(o, sa, (size_t)K);
            memcpy(o + K, sb, (size_t)L);
        }
        a += ta ? K : 0; b += tb ? L : 0; o += W;
    }
    *pa = a; *pb = b;
}

/* Blocks of at most 2 + 2 rows (C2): FOUR merged columns -- one byte of the edit script -- per step.  Their 4 (K + L)
 * <= 16 bytes are a byte shuffle of the next 4 K bytes of A beside the next 4 L bytes of B, with dashes where a column
 * takes none: the shuffle, the dash mask and the two advances are looked up by the script byte (tables built on first
 * use, 8.5 KB per shape).  The column-by-column form above spends ~14 instructions per column on selects and two-byte
 * moves -- 2.5 us per C2 pair, more than its packing; this one ~10 per FOUR columns. */
typedef struct shuf_tab { uint8_t sh[256][16], dash[256][16], adv_a[256], adv_b[256]; int ready; } __attribute__((aligned(16))) shuf_tab;
static shuf_tab g_shuf[3][3];                               /* [K][L], K, L <= 2 (eight bytes of each source per step) */
static int g_ssse3;

__attribute__((constructor)) static void shuf_init_cpu(void) { __builtin_cpu_init(); g_ssse3 = __builtin_cpu_supports("ssse3") && !(getenv("MZ_NO_SSSE3") && atoi(getenv("MZ_NO_SSSE3"))); }

static pthread_mutex_t g_shuf_mu = PTHREAD_MUTEX_INITIALIZER;
static void shuf_build(int K, int L)
{
    shuf_tab *T = &g_shuf[K][L];
    int x, k, r;
    pthread_mutex_lock(&g_shuf_mu);            /* ONE builder: a second one would blank entries a reader is using */
    if (T->ready) { pthread_mutex_unlock(&g_shuf_mu); return; }
    for (x = 0; x < 256; ++x) {
        int ia = 0, ib = 0, o = 0;
        memset(T->sh[x], 0x80, 16); memset(T->dash[x], 0, 16);
        for (k = 0; k < 4; ++k) {
            const unsigned op = ((unsigned)x >> (2 * k)) & 3u;
            const int ta = op != 1u, tb = op != 2u;
            for (r = 0; r < K; ++r, ++o) { if (ta) T->sh[x][o] = (uint8_t)(ia * K + r); else T->dash[x][o] = '-'; }
            for (r = 0; r < L; ++r, ++o) { if (tb) T->sh[x][o] = (uint8_t)(8 + ib * L + r); else T->dash[x][o] = '-'; }   /* B's bytes: upper half of the source register */
            ia += ta; ib += tb;
        }
        T->adv_a[x] = (uint8_t)(ia * K); T->adv_b[x] = (uint8_t)(ib * L);
    }
    __atomic_store_n(&T->ready, 1, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&g_shuf_mu);
}

/* columns [m0, m1), m0 a multiple of 4; returns the first column NOT done (the caller finishes the rest): the steps stop
 * where eight bytes can no longer be read from A or B */
__attribute__((target("ssse3"))) static int assemble_shuf(int K, int L, const uint8_t **pa, const uint8_t *aend, const uint8_t **pb,
                                                          const uint8_t *bend, const uint8_t *s, int m0, int m1, uint8_t **po)
{
    const shuf_tab *T = &g_shuf[K][L];
    const int W4 = 4 * (K + L);
    const uint8_t *a = *pa, *b = *pb;
    uint8_t *o = *po;
    int m = m0;
    for (; m + 4 <= m1 && a + 8 <= aend && b + 8 <= bend; m += 4) {
        const unsigned x = s[m >> 2];
        const __m128i src = _mm_unpacklo_epi64(_mm_loadl_epi64((const __m128i *)a), _mm_loadl_epi64((const __m128i *)b));
        const __m128i v = _mm_or_si128(_mm_shuffle_epi8(src, _mm_load_si128((const __m128i *)T->sh[x])), _mm_load_si128((const __m128i *)T->dash[x]));
        _mm_storeu_si128((__m128i *)o, v);                  /* (16 bytes: the columns after these four overwrite what is too much) */
        a += T->adv_a[x]; b += T->adv_b[x]; o += W4;
    }
    *pa = a; *pb = b; *po = o;
    return m;
}

static inline void assemble_piece(int K, int L, const uint8_t **pa, const uint8_t *aend, const uint8_t **pb, const uint8_t *bend,
                                  const uint8_t *s, int m0, int m1, uint8_t *o)
{
    if (K <= 2 && L <= 2 && g_ssse3) {
        if (!__atomic_load_n(&g_shuf[K][L].ready, __ATOMIC_ACQUIRE)) shuf_build(K, L);
        m0 = assemble_shuf(K, L, pa, aend, pb, bend, s, m0, m1, &o);
        if (m0 >= m1) return;
    }
    if (K <= 4 && L <= 4) {
        switch ((K - 1) * 4 + (L - 1)) {
#define CASE(k, l) case ((k) - 1) * 4 + ((l) - 1): assemble_small(k, l, pa, pb, s, m0, m1, o); return;
        CASE(1, 1) CASE(1, 2) CASE(1, 3) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 3) CASE(2, 4)
        CASE(3, 1) CASE(3, 2) CASE(3, 3) CASE(3, 4) CASE(4, 1) CASE(4, 2) CASE(4, 3) CASE(4, 4)
#undef CASE
        }
    }
    assemble_any(K, L, pa, aend, pb, bend, s, m0, m1, o);
}

/* The merged columns of one pair (reference mz_yama.c:293-313 + new_col :39-47) from its packed edit script: om
 * columns of K + L bytes to `out`, exactly om * (K + L) bytes.  A: M columns of K bytes, B: N columns of L bytes (the
 * job's own arrays).  The device has already checked that the script takes exactly M columns of A and N of B (status
 * MZ_E_EMIT otherwise).  The columns are put together a few KB at a time in a buffer that stays in L1 and leave it as
 * whole 64-byte lines through streaming stores; the lines `out` shares with its neighbours (the first when `out` is
 * not 64-byte aligned, the last) are written with plain stores of the pair's own bytes only. */
#define ASM_BUF 4096
void mz_assemble_cols(int K, int L, int M, int N, const uint8_t *A, const uint8_t *B, const uint8_t *script, int om, uint8_t *out)
{
    uint8_t buf[64 + ASM_BUF + 64] __attribute__((aligned(64)));
    const int W = K + L;
    const uint8_t *a = A, *b = B, *aend = A + (size_t)K * M, *bend = B + (size_t)L * N;
    const size_t phase = (size_t)((uintptr_t)out & 63);      /* buf[i] <-> line_base[i], line_base = out - phase */
    uint8_t *line = out - phase;
    size_t fill = phase, first = phase;                       /* bytes of buf in use; own bytes start at `first` (then 0) */
    int per = (ASM_BUF / W) & ~3, m0;
    if (per < 4) per = 4;                                     /* (W <= 510: four columns fit) */
    for (m0 = 0; m0 < om; m0 += per) {
        const int m1 = m0 + per < om ? m0 + per : om;
        size_t full;
        assemble_piece(K, L, &a, aend, &b, bend, script, m0, m1, buf + fill);
        fill += (size_t)(m1 - m0) * W;
        full = fill & ~(size_t)63;
        if (full) {
            size_t from = 0;
            if (first) { memcpy(line + first, buf + first, 64 - first); from = 64; first = 0; }     /* the line shared with the pair before */
            stream_lines(line + from, buf + from, full - from);
            line += full;
            memcpy(buf, buf + full, 64);                     /* the open line moves to the front */
            fill -= full;
        }
    }
    if (fill > first) memcpy(line + first, buf + first, fill - first);
}

/* ------------------------------------------------------------------------------------------------ merged rows
 * mz_preyama_batch() (mz_prebatch.c): the rows of a merged block, from the caller's own block text and what the device
 * sends back per merge (k_fin, kernels/prepost.inc) -- for every group of rows one bit per merged column (take the
 * row's next source byte, or a dash: reference mz_yama.c:293-313 through mafBuild, mz_preyama.c:61-66) and, where
 * rmColDash (mz_preyama.c:87-108) dropped columns of the slice, one bit per slice column (kept or not).  A row is its
 * source bytes, squeezed by the second kind of mask and spread out by the first: eight columns per table look-up
 * (pshufb), the rows of a block one after the other through the same L1 buffer and streaming stores as the columns above. */
static uint8_t g_exp[256][8] __attribute__((aligned(16)));      /* spread: source byte index per output byte, 0x80 where a dash goes */
static uint8_t g_expd[256][8] __attribute__((aligned(16)));     /* '-' where a dash goes */
static uint8_t g_cmp[256][8] __attribute__((aligned(16)));      /* squeeze: the kept bytes to the front */
static uint8_t g_pop8[256];

__attribute__((constructor)) static void rows_init(void)
{
    int x, k;
    for (x = 0; x < 256; ++x) {
        int i = 0;
        memset(g_cmp[x], 0x80, 8);
        for (k = 0; k < 8; ++k) {
            if (x >> k & 1) { g_exp[x][k] = (uint8_t)i; g_expd[x][k] = 0; g_cmp[x][i] = (uint8_t)k; ++i; }
            else { g_exp[x][k] = 0x80; g_expd[x][k] = '-'; }
        }
        g_pop8[x] = (uint8_t)i;
    }
}

#define MASK8(m, c) ((unsigned)((m)[(c) >> 6] >> ((c) & 63)) & 255u)

/* columns [c0, c1) of one row into o (c0 a multiple of 16; o may be written up to 15 bytes past the last column):
 * *ps = the row's next unread source byte, send = the end of its source */
__attribute__((target("ssse3"))) static void spread_ssse3(const uint8_t **ps, const uint8_t *send, const uint64_t *ops, int c0, int c1, uint8_t *o)
{
    const uint8_t *s = *ps;
    int c = c0;
    for (; c + 16 <= c1 && s + 16 <= send; c += 16) {
        const unsigned m0 = MASK8(ops, c), m1 = MASK8(ops, c + 8);
        const __m128i v0 = _mm_loadl_epi64((const __m128i *)s), v1 = _mm_loadl_epi64((const __m128i *)(s + g_pop8[m0]));
        const __m128i r0 = _mm_or_si128(_mm_shuffle_epi8(v0, _mm_loadl_epi64((const __m128i *)g_exp[m0])), _mm_loadl_epi64((const __m128i *)g_expd[m0]));
        const __m128i r1 = _mm_or_si128(_mm_shuffle_epi8(v1, _mm_loadl_epi64((const __m128i *)g_exp[m1])), _mm_loadl_epi64((const __m128i *)g_expd[m1]));
        _mm_storeu_si128((__m128i *)(o + (c - c0)), _mm_unpacklo_epi64(r0, r1));
        s += g_pop8[m0] + g_pop8[m1];
    }
    for (; c < c1; ++c) o[c - c0] = (ops[c >> 6] >> (c & 63) & 1) ? *s++ : (uint8_t)'-';
    *ps = s;
}
static void spread_scalar(const uint8_t **ps, const uint64_t *ops, int c0, int c1, uint8_t *o)
{
    const uint8_t *s = *ps;
    int c;
    for (c = c0; c < c1; ++c) o[c - c0] = (ops[c >> 6] >> (c & 63) & 1) ? *s++ : (uint8_t)'-';
    *ps = s;
}

/* src[0..n) without the bytes whose bit in `keep` is clear (dst: n + 16 bytes); returns the bytes kept */
__attribute__((target("ssse3"))) static size_t squeeze_mask_ssse3(const uint8_t *src, int n, const uint64_t *keep, uint8_t *dst)
{
    size_t j = 0;
    int c = 0;
    for (; c + 8 <= n; c += 8) {
        const unsigned m = MASK8(keep, c);
        _mm_storel_epi64((__m128i *)(dst + j), _mm_shuffle_epi8(_mm_loadl_epi64((const __m128i *)(src + c)), _mm_loadl_epi64((const __m128i *)g_cmp[m])));
        j += g_pop8[m];
    }
    for (; c < n; ++c) if (keep[c >> 6] >> (c & 63) & 1) dst[j++] = src[c];
    return j;
}
/* ... without its dashes (the first block's top row as the second stage of a v = 0 merge aligns it, mz_preyama.c:282-290) */
__attribute__((target("ssse3"))) static size_t squeeze_dash_ssse3(const uint8_t *src, int n, uint8_t *dst)
{
    const __m128i dash = _mm_set1_epi8('-');
    size_t j = 0;
    int c = 0;
    for (; c + 16 <= n; c += 16) {
        const __m128i v = _mm_loadu_si128((const __m128i *)(src + c));
        const unsigned m = ~(unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, dash)) & 0xffffu, m0 = m & 255u, m1 = m >> 8;
        _mm_storel_epi64((__m128i *)(dst + j), _mm_shuffle_epi8(v, _mm_loadl_epi64((const __m128i *)g_cmp[m0])));
        j += g_pop8[m0];
        _mm_storel_epi64((__m128i *)(dst + j), _mm_shuffle_epi8(_mm_srli_si128(v, 8), _mm_loadl_epi64((const __m128i *)g_cmp[m1])));
        j += g_pop8[m1];
    }
    for (; c < n; ++c) if (src[c] != '-') dst[j++] = src[c];
    return j;
}

/* nrows rows of om bytes, one after the other, to out (exactly nrows * om bytes).  tmp: room for the longest source row
 * that has to be squeezed + 16 bytes (NULL when none has).  The device has checked that every mask takes exactly the
 * bytes its row has (the closing check of mz_yama.c:310-312 on each stage's script); a spread never reads past `n`. */
#define ROW_PIECE 2048
void mz_assemble_rows(int nrows, const mz_rowspec *rows, int om, uint8_t *out)
{
    uint8_t buf[64 + ASM_BUF + 64] __attribute__((aligned(64)));
    const size_t phase = (size_t)((uintptr_t)out & 63);
    uint8_t *line = out - phase;
    size_t fill = phase, first = phase;
    int r, c0;
    for (r = 0; r < nrows; ++r) {
        const mz_rowspec *q = &rows[r];
        const uint8_t *s = q->src, *send = q->src + q->n;
        if (q->squeeze) {
            size_t kept;
            if (q->squeeze == 2) {
                if (g_ssse3) kept = squeeze_dash_ssse3(q->src, q->n, q->tmp);
                else { int c; for (c = 0, kept = 0; c < q->n; ++c) if (q->src[c] != '-') q->tmp[kept++] = q->src[c]; }
            } else if (g_ssse3) kept = squeeze_mask_ssse3(q->src, q->n, q->keep, q->tmp);
            else { int c; for (c = 0, kept = 0; c < q->n; ++c) if (q->keep[c >> 6] >> (c & 63) & 1) q->tmp[kept++] = q->src[c]; }
            s = q->tmp; send = q->tmp + kept;
        }
        for (c0 = 0; c0 < om; c0 += ROW_PIECE) {
            const int c1 = c0 + ROW_PIECE < om ? c0 + ROW_PIECE : om;
            size_t full;
            if (g_ssse3) spread_ssse3(&s, send, q->ops, c0, c1, buf + fill);
            else spread_scalar(&s, q->ops, c0, c1, buf + fill);
            fill += (size_t)(c1 - c0);
            full = fill & ~(size_t)63;
            if (full) {
                size_t from = 0;
                if (first) { memcpy(line + first, buf + first, 64 - first); from = 64; first = 0; }     /* the line shared with the merge before */
                stream_lines(line + from, buf + from, full - from);
                line += full;
                memcpy(buf, buf + full, 64);
                fill -= full;
            }
        }
    }
    if (fill > first) memcpy(line + first, buf + first, fill - first);
}
