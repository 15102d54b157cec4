/* mz_synth.c -- synthetic block-pair batches for the benchmark and the large-size tests
 * (SURVEY.md section 8d).  Not part of the reference; it only produces inputs in the packed
 * layout of include/mz_amd.h.
 *
 * Generator: xorshift64, base seed 88172645463325252; pair p of a batch uses its own stream
 * seeded from (seed, p) so that batches can be generated in shards (one per rank) and stay
 * identical to the unsharded batch.
 *   A: M columns of K bytes; a byte is '-' with probability 0.08, else uniform over ACGT;
 *      every column keeps at least one non-dash.  5 % of the columns carry lowercase / N bytes.
 *   B: column j <= M copies A[j][k mod K], is substituted with probability 0.10, then dashed
 *      with probability 0.08 (>= 1 non-dash per column); columns j > M are random.
 *   band "diag": LB[i] = RB[i] = floor(i*N/M), LB[0] = 0, then smooth(LB, RB, M, N, radius).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mz_amd.h"
#include "../../include/mz_preyama.h"

typedef struct { uint64_t s; } rng_t;

static uint64_t xs64(rng_t *r)
{
    uint64_t x = r->s;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return r->s = x;
}
static void rng_seed(rng_t *r, uint64_t seed, uint64_t stream)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (stream + 1);      /* splitmix64 scramble */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    r->s = z ? z : 88172645463325252ULL;
    xs64(r); xs64(r);
}
static unsigned below(rng_t *r, unsigned n) { return (unsigned)((xs64(r) >> 11) % n); }
static int chance(rng_t *r, unsigned per_mille) { return below(r, 1000) < per_mille; }

/* shapes of pair p: rows fixed, columns uniform in [lo, hi] */
void mz_synth_shapes(int n, uint64_t seed, int64_t first_pair, int K, int L, int mlo, int mhi,
                     int32_t *aK, int32_t *aL, int32_t *aM, int32_t *aN,
                     int64_t *offA, int64_t *offB, int64_t *offBand, int64_t totals[3])
{
    int64_t oa = 0, ob = 0, oband = 0;
    int p;
    for (p = 0; p < n; ++p) {
        rng_t r;
        rng_seed(&r, seed, (uint64_t)(first_pair + p));
        aK[p] = K; aL[p] = L;
        aM[p] = mlo + (int)below(&r, (unsigned)(mhi - mlo + 1));
        aN[p] = mlo + (int)below(&r, (unsigned)(mhi - mlo + 1));
        offA[p] = oa; offB[p] = ob; offBand[p] = oband;
        oa += (int64_t)aK[p] * aM[p]; ob += (int64_t)aL[p] * aN[p]; oband += aM[p] + 1;
    }
    totals[0] = oa; totals[1] = ob; totals[2] = oband;
}

/* BASELINE config 4: the block pairs of a 30-leaf guide tree.  The tree is a balanced
 * 16-leaf subtree with a 14-leaf caterpillar on top: 29 internal nodes, and the merge at a node with p and q leaves
 * below its two children aligns a p-row block with a q-row block (K = p, L = q; tba.c:177-255 walks such a tree
 * bottom-up, every node's merges are independent of its siblings').  Pair p of a batch belongs to a node drawn
 * from its own random stream, so all 29 shapes are mixed through any shard. */
static const unsigned char tree30[29][2] = {
    { 1, 1 }, { 1, 1 }, { 1, 1 }, { 1, 1 }, { 1, 1 }, { 1, 1 }, { 1, 1 }, { 1, 1 },      /* balanced part: 8 cherries */
    { 2, 2 }, { 2, 2 }, { 2, 2 }, { 2, 2 }, { 4, 4 }, { 4, 4 }, { 8, 8 },                /* ... up to its 16-leaf root */
    { 16, 1 }, { 17, 1 }, { 18, 1 }, { 19, 1 }, { 20, 1 }, { 21, 1 }, { 22, 1 },         /* caterpillar: one more leaf per node */
    { 23, 1 }, { 24, 1 }, { 25, 1 }, { 26, 1 }, { 27, 1 }, { 28, 1 }, { 29, 1 } };

int mz_synth_tree_nodes(int32_t *K, int32_t *L)
{
    int i;
    for (i = 0; i < 29; ++i) { if (K) K[i] = tree30[i][0]; if (L) L[i] = tree30[i][1]; }
    return 29;
}

/* shapes of pair p for the tree workload: (K, L) of a random node, columns uniform in [lo, hi] */
void mz_synth_shapes_tree(int n, uint64_t seed, int64_t first_pair, int mlo, int mhi,
                          int32_t *aK, int32_t *aL, int32_t *aM, int32_t *aN,
                          int64_t *offA, int64_t *offB, int64_t *offBand, int64_t totals[3])
{
    int64_t oa = 0, ob = 0, oband = 0;
    int p;
    for (p = 0; p < n; ++p) {
        rng_t r;
        unsigned node;
        rng_seed(&r, seed, (uint64_t)(first_pair + p));
        aM[p] = mlo + (int)below(&r, (unsigned)(mhi - mlo + 1));
        aN[p] = mlo + (int)below(&r, (unsigned)(mhi - mlo + 1));
        node = below(&r, 29);
        aK[p] = tree30[node][0]; aL[p] = tree30[node][1];
        offA[p] = oa; offB[p] = ob; offBand[p] = oband;
        oa += (int64_t)aK[p] * aM[p]; ob += (int64_t)aL[p] * aN[p]; oband += aM[p] + 1;
    }
    totals[0] = oa; totals[1] = ob; totals[2] = oband;
}

static unsigned char base_byte(rng_t *r, int odd)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    static const char other[6] = { 'a', 'c', 'g', 't', 'N', 'n' };
    if (odd && chance(r, 500)) return (unsigned char)other[below(r, 6)];
    return (unsigned char)acgt[below(r, 4)];
}

void mz_synth_fill(int n, uint64_t seed, int64_t first_pair, int radius,
                   const int32_t *aK, const int32_t *aL, const int32_t *aM, const int32_t *aN,
                   const int64_t *offA, const int64_t *offB, const int64_t *offBand,
                   uint8_t *poolA, uint8_t *poolB, int32_t *poolLB, int32_t *poolRB)
{
    int p;
#pragma omp parallel for schedule(dynamic, 64) if (n > 256)       /* (every pair has its own stream: any order) */
    for (p = 0; p < n; ++p) {
        const int K = aK[p], L = aL[p], M = aM[p], N = aN[p];
        uint8_t *A = poolA + offA[p], *B = poolB + offB[p];
        int32_t *LB = poolLB + offBand[p], *RB = poolRB + offBand[p];
        rng_t r;
        int i, k;

        rng_seed(&r, seed ^ 0xA5A5A5A5ULL, (uint64_t)(first_pair + p));
        for (i = 0; i < M; ++i) {
            const int odd = chance(&r, 50);
            int nd = 0;
            for (k = 0; k < K; ++k) {
                unsigned char ch = chance(&r, 80) ? '-' : base_byte(&r, odd);
                A[(size_t)i * K + k] = ch;
                nd += ch != '-';
            }
            if (!nd) A[(size_t)i * K + below(&r, (unsigned)K)] = base_byte(&r, 0);
        }
        for (i = 0; i < N; ++i) {
            const int odd = chance(&r, 50);
            int nd = 0;
            for (k = 0; k < L; ++k) {
                unsigned char ch;
                if (i < M) {
                    ch = A[(size_t)i * K + (k % K)];
                    if (ch == '-' || chance(&r, 100)) ch = base_byte(&r, odd);
                } else {
                    ch = base_byte(&r, odd);
                }
                if (chance(&r, 80)) ch = '-';
                B[(size_t)i * L + k] = ch;
                nd += ch != '-';
            }
            if (!nd) B[(size_t)i * L + below(&r, (unsigned)L)] = base_byte(&r, 0);
        }
        for (i = 0; i <= M; ++i)
            LB[i] = RB[i] = (int32_t)(((int64_t)i * N) / M);
        LB[0] = 0;
        smooth(LB, RB, M, N, radius);
    }
}

/* Bands as pre_yama derives them from blocks with indels against the shared reference row (mz_preyama.c:240-258, then
 * smooth): the band's centre walks down the diagonal, stands still over a run of columns only the first block has
 * and jumps over a run only the second block has -- `events` such runs per 1 000 columns, half of each kind, lengths
 * geometric with mean 3.  The pair's own stream decides, so shapes (N = where the walk ends) and bands agree. */
/* `events`: runs per 1 000 columns in the low 16 bits; bits 16..: how many of 1 000 runs are LONG -- 70 to 300 columns, uniform -- instead
 * of geometric with mean 3 (the heavy tail of real blocks: one such run makes the rows around it 100-330 columns wide) */
static int indel_run(rng_t *r, int tail)
{
    int g = 1;
    if (tail && (int)below(r, 1000) < tail) return 70 + (int)below(r, 231);
    while (below(r, 3) != 0) ++g;
    return g;
}
static int indel_walk(rng_t *r, int M, int events_tail, int32_t *centre)      /* centre: M+1 entries or NULL; returns N */
{
    const int events = events_tail & 0xffff, tail = events_tail >> 16;
    int i = 1, c = 0;
    if (centre) centre[0] = 0;
    while (i <= M) {
        const int u = (int)below(r, 1000);
        if (2 * u < events && i > 1) {
            int g = indel_run(r, tail);
            for (; g > 0 && i <= M; --g, ++i) if (centre) centre[i] = c;
            continue;
        }
        if (u < events) c += indel_run(r, tail);
        ++c;
        if (centre) centre[i] = c;
        ++i;
    }
    return c > 11 ? c : 11;
}
void mz_synth_shapes_indel(int n, uint64_t seed, int64_t first_pair, int K, int L, int mlo, int mhi, int events,
                           int32_t *aK, int32_t *aL, int32_t *aM, int32_t *aN,
                           int64_t *offA, int64_t *offB, int64_t *offBand, int64_t totals[3])
{
    int64_t oa = 0, ob = 0, oband = 0;
    int p;
    for (p = 0; p < n; ++p) {
        rng_t r, w;
        rng_seed(&r, seed, (uint64_t)(first_pair + p));
        rng_seed(&w, seed ^ 0x1D1D1D1DULL, (uint64_t)(first_pair + p));
        aM[p] = mlo + (int)below(&r, (unsigned)(mhi - mlo + 1));
        if (K == 0 && L == 0) {                          /* (K, L) of a random node of the C4 guide tree */
            const unsigned node = below(&r, 29);
            aK[p] = tree30[node][0]; aL[p] = tree30[node][1];
        } else { aK[p] = K; aL[p] = L; }
        aN[p] = indel_walk(&w, aM[p], events, NULL);
        offA[p] = oa; offB[p] = ob; offBand[p] = oband;
        oa += (int64_t)aK[p] * aM[p]; ob += (int64_t)aL[p] * aN[p]; oband += aM[p] + 1;
    }
    totals[0] = oa; totals[1] = ob; totals[2] = oband;
}
/* the bands of such a batch (after mz_synth_fill, whose diagonal bands they replace) */
void mz_synth_bands_indel(int n, uint64_t seed, int64_t first_pair, int radius, int events, const int32_t *aM, const int32_t *aN,
                          const int64_t *offBand, int32_t *poolLB, int32_t *poolRB)
{
    int p;
#pragma omp parallel for schedule(dynamic, 64) if (n > 256)
    for (p = 0; p < n; ++p) {
        const int M = aM[p], N = aN[p];
        int32_t *LB = poolLB + offBand[p], *RB = poolRB + offBand[p];
        rng_t w;
        int i;
        rng_seed(&w, seed ^ 0x1D1D1D1DULL, (uint64_t)(first_pair + p));
        indel_walk(&w, M, events, LB);
        for (i = 0; i <= M; ++i) { if (LB[i] > N) LB[i] = N; RB[i] = LB[i]; }
        LB[0] = 0; RB[M] = N;
        smooth(LB, RB, M, N, radius);
    }
}

/* ------------------------------------------------------------------------------------------------ block TEXT
 * Inputs of mz_preyama_batch(): two blocks topped by the same reference row, over their overlap (what the drivers hand
 * to pre_yama(), mz_preyama.c:152-172).  Merge p: R reference bases (uniform in [rlo, rhi]); block 1 has K rows (its top
 * row is the reference), block 2 has L rows below ITS copy of the reference row (L1 = L + 1), so that the yama() job of a
 * one-stage merge is K x L rows -- the shapes of the yama-level configurations.  Either block has runs of columns the
 * other lacks (the reference row holds dashes there): `events` runs per 1 000 reference bases, half in each block,
 * lengths geometric with mean 3 -- the band pre_yama derives walks around them exactly as mz_synth_bands_indel's does.
 * Rows below the top: the reference base, substituted with probability 0.10, a dash with 0.08; 5 columns per 1 000 are
 * all dashes below the top row (rmColDash, mz_preyama.c:87-108, has something to drop).  Text of merge p at
 * pool + offText[p]: K rows of Ma bytes, then L1 rows of Na bytes. */
typedef struct { int R, Ma, Na; } pre_shape;
static int pre_run(rng_t *w, int events) { return (int)below(w, 2000) < (events & 0xffff) ? indel_run(w, events >> 16) : 0; }
static pre_shape pre_walk(uint64_t seed, int64_t p, int K, int rlo, int rhi, int events, uint8_t *ins1, uint8_t *ins2)
{
    rng_t r, w;
    pre_shape s;
    int i;
    rng_seed(&r, seed, (uint64_t)p);
    rng_seed(&w, seed ^ 0x2E2E2E2EULL, (uint64_t)p);
    s.R = rlo + (int)below(&r, (unsigned)(rhi - rlo + 1));
    s.Ma = s.Na = s.R;
    for (i = 0; i < s.R; ++i) {                           /* runs of unshared columns in front of reference base i */
        const int g1 = (K > 1 && i > 0) ? pre_run(&w, events) : 0, g2 = i > 0 ? pre_run(&w, events) : 0;
        if (ins1) ins1[i] = (uint8_t)(g1 > 255 ? 255 : g1);
        if (ins2) ins2[i] = (uint8_t)(g2 > 255 ? 255 : g2);
        s.Ma += g1 > 255 ? 255 : g1; s.Na += g2 > 255 ? 255 : g2;
    }
    return s;
}

void mz_synth_pre_shapes(int n, uint64_t seed, int64_t first_pair, int K, int L, int rlo, int rhi, int events,
                         int32_t *aK, int32_t *aL1, int32_t *aMa, int32_t *aNa, int64_t *offText, int64_t *total)
{
    int64_t ot = 0;
    int p;
    for (p = 0; p < n; ++p) {
        pre_shape s;
        if (K == 0 && L == 0) {                          /* (K, L) of a random node of the C4 guide tree */
            rng_t r;
            unsigned node;
            rng_seed(&r, seed ^ 0x3C3C3C3CULL, (uint64_t)(first_pair + p));
            node = below(&r, 29);
            aK[p] = tree30[node][0]; aL1[p] = tree30[node][1] + 1;
        } else { aK[p] = K; aL1[p] = L + 1; }
        s = pre_walk(seed, first_pair + p, aK[p], rlo, rhi, events, NULL, NULL);
        aMa[p] = s.Ma; aNa[p] = s.Na;
        offText[p] = ot;
        ot += (int64_t)aK[p] * s.Ma + (int64_t)aL1[p] * s.Na;
    }
    *total = ot;
}

static void pre_rows(rng_t *r, uint8_t *T, int rows, int cols, const uint8_t *ref, const uint8_t *ins, int R)
{
    int i, c = 0, k, g;
    for (i = 0; i < R; ++i) {
        for (g = 0; g < ins[i]; ++g, ++c) {              /* a column the other block lacks: a dash in the reference row */
            int nd = 0;
            T[c] = '-';
            for (k = 1; k < rows; ++k) { const unsigned char ch = chance(r, 300) ? '-' : base_byte(r, 0); T[(size_t)k * cols + c] = ch; nd += ch != '-'; }
            if (!nd) T[(size_t)(1 + below(r, (unsigned)(rows - 1))) * cols + c] = base_byte(r, 0);
        }
        {
            const int odd = chance(r, 50), alldash = chance(r, 5);
            T[c] = ref[i];
            for (k = 1; k < rows; ++k) {
                unsigned char ch = ref[i];
                if (chance(r, 100)) ch = base_byte(r, odd);
                if (alldash || chance(r, 80)) ch = '-';
                T[(size_t)k * cols + c] = ch;
            }
            ++c;
        }
    }
}

void mz_synth_pre_fill(int n, uint64_t seed, int64_t first_pair, int rlo, int rhi, int events,
                       const int32_t *aK, const int32_t *aL1, const int32_t *aMa, const int32_t *aNa, const int64_t *offText, uint8_t *pool)
{
    int p;
#pragma omp parallel for schedule(dynamic, 64) if (n > 256)
    for (p = 0; p < n; ++p) {
        uint8_t *ref = (uint8_t *)malloc(3 * ((size_t)rhi + 1)), *ins1 = ref + rhi + 1, *ins2 = ins1 + rhi + 1;
        uint8_t *T1 = pool + offText[p], *T2 = T1 + (size_t)aK[p] * aMa[p];
        rng_t r;
        pre_shape s;
        int i;
        if (!ref) continue;
        s = pre_walk(seed, first_pair + p, aK[p], rlo, rhi, events, ins1, ins2);
        rng_seed(&r, seed ^ 0x5A5A5A5AULL, (uint64_t)(first_pair + p));
        for (i = 0; i < s.R; ++i) ref[i] = base_byte(&r, chance(&r, 50));
        pre_rows(&r, T1, aK[p], aMa[p], ref, ins1, s.R);
        pre_rows(&r, T2, aL1[p], aNa[p], ref, ins2, s.R);
        free(ref);
    }
}

/* src[off[i] .. off[i]+len[i]) (elements of `elem` bytes) to dst[pos[i] ..): the re-packing step of sharding a
 * batch (multiz_amd/shard.py) and of sampling one (bench.py), one memcpy per segment on the host threads.  `pos` is the
 * caller's exclusive prefix sum of len (no allocation here, nothing that can fail half-way).  Returns 0, -1 on bad
 * arguments. */
int mz_gather_segments(int64_t n, int64_t elem, const int64_t *off, const int64_t *len, const int64_t *pos, const void *src, void *dst)
{
    int64_t i;
    if (n < 0 || elem <= 0 || (n > 0 && (!off || !len || !pos || !src || !dst))) return -1;
#pragma omp parallel for schedule(static) if (n > 1024)
    for (i = 0; i < n; ++i)
        memcpy((char *)dst + pos[i] * elem, (const char *)src + off[i] * elem, (size_t)(len[i] * elem));
    return 0;
}
