/* oracle/yama_oracle.c -- faithful CPU restatement of the reference yama() recurrence.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity status: PINNED against
 * oracle/_ref/libref.so and tests/golden/ (see oracle.h).
 *
 * Written from the behavioural specification in SURVEY.md appendix A, which cites:
 *   band validity        reference mz_yama.c:58-71
 *   row 0                reference mz_yama.c:83-94
 *   I / C / D updates    reference mz_yama.c:113-166 / 168-205 / 207-242
 *   traceback byte       reference mz_yama.c:244-253
 *   traceback + emit     reference mz_yama.c:257-313
 *   score tables         reference mz_scores.c:9-14,23-29,34-81
 *   smooth()             reference mz_preyama.c:17-35
 * The arithmetic is the reference's: int32, table look-ups per (row-of-A,row-of-B)
 * pair, so the cost per cell is O(K*L) exactly like the CPU path being replaced.
 */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

/* ------------------------------------------------------------------ scratch memory
 * A bump arena so that the batch driver can reuse one block per thread instead of paying
 * malloc/free (and the mm system calls behind them) once per block pair. NULL arena = malloc. */
void *mzo__alloc(mzo_arena *ar, size_t bytes, int zero)
{
    void *p;
    bytes = (bytes + 63) & ~(size_t)63;
    if (ar == NULL) return zero ? calloc(bytes ? bytes : 64, 1) : malloc(bytes ? bytes : 64);
    if (ar->used + bytes > ar->cap) return NULL;
    p = ar->base + ar->used;
    ar->used += bytes;
    if (zero) memset(p, 0, bytes);
    return p;
}
void mzo__free(mzo_arena *ar, void *p) { if (ar == NULL) free(p); }

/* bytes of scratch one pair needs (both restatements), generously rounded */
size_t mzo__scratch_bytes(int K, int L, int M, int N, int64_t cells)
{
    size_t m = (size_t)M + 2, n = (size_t)N + 2;
    return m * K + n * L + (size_t)cells + 8 * m + 12 * n + (m + n) + (m + n) * (size_t)(K + L)
           + 72 * (m + n) + 64 * 16;
}

/* ------------------------------------------------------------------ score tables */

static void fill_scores(mzo_scores *sc, const int sub[4][4], int filler, int open_, int ext)
{
    static const char nt[4] = { 'A', 'C', 'G', 'T' };
    int a, b, x;

    for (a = 0; a < 128; ++a)
        for (b = 0; b < 128; ++b)
            sc->ss[a][b] = filler;
    for (a = 0; a < 4; ++a)
        for (b = 0; b < 4; ++b) {
            int U1 = nt[a], U2 = nt[b], l1 = tolower(U1), l2 = tolower(U2);
            sc->ss[U1][U2] = sc->ss[U1][l2] = sc->ss[l1][U2] = sc->ss[l1][l2] = sub[a][b];
        }
    for (x = 0; x < 128; ++x)
        sc->ss['-'][x] = sc->ss[x]['-'] = -ext;
    sc->ss['-']['-'] = 0;

    /* quasi-natural gap-open table, index = s<<3 | t<<2 | u<<1 | v where (s,t) are the
     * dash flags of the previous column pair and (u,v) those of the current one.
     * Non-zero exactly when the current pair has one dash and the previous pair was not
     * the same one-sided pattern: {0001,0010,0110,1001,1101,1110}. */
    for (x = 0; x < 16; ++x) {
        int s = (x >> 3) & 1, t = (x >> 2) & 1, u = (x >> 1) & 1, v = x & 1;
        sc->gop[x] = (u != v && !(s == u && t == v)) ? open_ : 0;
    }
    sc->gap_open = open_;
    sc->gap_extend = ext;
}

void mzo_scores_hoxd70(mzo_scores *sc)
{
    static const int m[4][4] = {
        {   91, -114,  -31, -123 },
        { -114,  100, -125,  -31 },
        {  -31, -125,  100, -114 },
        { -123,  -31, -114,   91 } };
    fill_scores(sc, m, -100, 400, 30);
}

void mzo_scores_hoxd85(mzo_scores *sc)
{
    static const int m[4][4] = {
        {   86, -135,  -68, -157 },
        { -135,  100, -148,  -68 },
        {  -68, -148,  100, -135 },
        { -157,  -68, -135,   86 } };
    fill_scores(sc, m, -100, 600, 50);
}

/* ------------------------------------------------------------------ helpers */

uint64_t mzo_fnv1a(const uint8_t *p, int64_t n, uint64_t h)
{
    int64_t i;
    if (h == 0) h = 1469598103934665603ULL;
    for (i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

int mzo_yama_check(int M, int N, const int *LB, const int *RB, int64_t *cells, int *bad_row)
{
    int r, need = N < 10 ? N : 10;
    int64_t tot = 0;

    if (bad_row) *bad_row = -1;
    if (LB[0] != 0 || RB[M] != N)
        return MZO_E_TERMINATION;
    for (r = 0; r <= M; ++r) {
        int w = RB[r] - LB[r];
        if (w < need) { if (bad_row) *bad_row = r; return MZO_E_NARROW; }
        tot += w + 1;
        if (r > 0 && LB[r] < LB[r-1]) { if (bad_row) *bad_row = r; return MZO_E_LB_MONO; }
        if (r > 0 && RB[r] < RB[r-1]) { if (bad_row) *bad_row = r; return MZO_E_RB_MONO; }
    }
    if (cells) *cells = tot;
    return MZO_OK;
}

void mzo_smooth(int *LB, int *RB, int M, int N, int radius)
{
    int i, run, rad = M < radius ? M : radius;

    for (i = 0, run = 0; i <= M; ++i) {          /* running max: LB non-decreasing */
        if (LB[i] > run) run = LB[i];
        LB[i] = run;
    }
    for (i = M, run = N; i >= 0; --i) {          /* running min from the right: RB non-decreasing */
        if (RB[i] < run) run = RB[i];
        RB[i] = run;
    }
    /* widen: high rows first for LB so that LB[i-rad] is still the un-widened value */
    for (i = M; i > rad; --i) {
        int a = LB[i] - rad;
        if (a < 0) a = 0;
        LB[i] = a < LB[i-rad] ? a : LB[i-rad];
    }
    for (; i >= 0; --i)
        LB[i] = 0;
    for (i = 0; i < M - rad; ++i) {
        int a = RB[i] + rad;
        if (a > N) a = N;
        RB[i] = a > RB[i+rad] ? a : RB[i+rad];
    }
    for (; i <= M; ++i)
        RB[i] = N;
}

typedef struct { int32_t C, D, I; } tri;
int mzo__faithful(mzo_arena *ar, const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                  const int *LB, const int *RB, const mzo_scores *sc,
                  uint8_t *out, int *OM, int32_t *final3, uint8_t *tb_out);
int mzo__profile(mzo_arena *ar, const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                 const int *LB, const int *RB, const mzo_scores *sc,
                 uint8_t *out, int *OM, int32_t *final3, uint8_t *tb_out);
static const tri TRI_NEG = { MZO_NEG, MZO_NEG, MZO_NEG };

/* three-way choice with the interior tie order: C-pred wins ties, then D only if
 * strictly greater than I (mz_yama.c:138-154,189-198,226-235) */
static inline int32_t choose(int32_t x, int32_t y, int32_t z, unsigned *flag)
{
    if (x >= y && x >= z) { *flag = MZO_FC; return x; }
    if (y > z)            { *flag = MZO_FD; return y; }
    *flag = MZO_FI;
    return z;
}

/* sum over (i<K, j<L) of gop[s_i t_j u_i v_j]; a NULL flag vector means "constant 1" */
static int32_t gap_sum(const mzo_scores *sc, const uint8_t *s, const uint8_t *t,
                       const uint8_t *u, const uint8_t *v, int K, int L)
{
    int32_t tot = 0;
    int i, j;
    for (i = 0; i < K; ++i) {
        int si = s ? s[i] : 1, ui = u ? u[i] : 1;
        for (j = 0; j < L; ++j) {
            int tj = t ? t[j] : 1, vj = v ? v[j] : 1;
            tot += sc->gop[(si << 3) | (tj << 2) | (ui << 1) | vj];
        }
    }
    return tot;
}

/* ------------------------------------------------------------------ traceback + emit
 * shared by both restatements (they differ only in how a cell's sums are evaluated) */
int mzo__trace_emit(mzo_arena *ar, const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                    const int *LB, const uint8_t *tb, const int64_t *rowoff,
                    tri last, uint8_t *out, int *OM)
{
    uint8_t *ops = (uint8_t *)mzo__alloc(ar, (size_t)(M + N) + 1, 0);
    int r = M, c = N, n = 0, i, j, m, k;
    unsigned node;

    /* final-cell tie order: C, then D if >= I (mz_yama.c:262-267) */
    if (last.C >= last.D && last.C >= last.I) node = MZO_FC;
    else if (last.D >= last.I)                node = MZO_FD;
    else                                      node = MZO_FI;

    while (r > 0 || c > 0) {
        unsigned st;
        if (r < 0 || c < 0 || n >= M + N) { mzo__free(ar, ops); return MZO_E_TRACEBACK; }
        st = tb[rowoff[r] + (c - LB[r])];
        ops[n++] = (uint8_t)node;
        if (node == MZO_FI)      { c--;      node = (st >> 4) & 3; }
        else if (node == MZO_FD) { r--;      node = (st >> 2) & 3; }
        else if (node == MZO_FC) { r--; c--; node = st & 3; }
        else { mzo__free(ar, ops); return MZO_E_TRACEBACK; }
    }
    if (r != 0 || c != 0) { mzo__free(ar, ops); return MZO_E_TRACEBACK; }

    *OM = n;
    i = j = m = 0;
    for (k = n - 1; k >= 0; --k, ++m) {
        uint8_t *col = out + (size_t)m * (K + L);
        if (ops[k] == MZO_FC) {
            memcpy(col, A + (size_t)i * K, K); memcpy(col + K, B + (size_t)j * L, L); ++i; ++j;
        } else if (ops[k] == MZO_FI) {
            memset(col, '-', K);               memcpy(col + K, B + (size_t)j * L, L); ++j;
        } else {
            memcpy(col, A + (size_t)i * K, K); memset(col + K, '-', L); ++i;
        }
    }
    mzo__free(ar, ops);
    return (i == M && j == N) ? MZO_OK : MZO_E_EMIT;
}

/* ------------------------------------------------------------------ faithful DP */
int mzo_yama_faithful(const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                      const int *LB, const int *RB, const mzo_scores *sc,
                      uint8_t *out, int *OM, int32_t *final3, uint8_t *tb_out)
{
    return mzo__faithful(NULL, A, K, M, B, L, N, LB, RB, sc, out, OM, final3, tb_out);
}

int mzo__faithful(mzo_arena *ar, const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                  const int *LB, const int *RB, const mzo_scores *sc,
                  uint8_t *out, int *OM, int32_t *final3, uint8_t *tb_out)
{
    int64_t cells, *rowoff;
    uint8_t *tb, *dashA, *dashB, *tp;
    tri *dp, left;
    int r, c, i, j, rc;
    const int ge = sc->gap_extend;

    rc = mzo_yama_check(M, N, LB, RB, &cells, NULL);
    if (rc) return rc;

    /* dash flags; index 0 is an all-zero sentinel column so that "previous column of
     * column 1" reads as non-dash -- that is what the r>1 / c>1 tests in
     * mz_yama.c:128,174,211 amount to */
    dashA = (uint8_t *)mzo__alloc(ar, (size_t)(M + 1) * K, 1);
    dashB = (uint8_t *)mzo__alloc(ar, (size_t)(N + 1) * L, 1);
    for (r = 1; r <= M; ++r)
        for (i = 0; i < K; ++i) dashA[(size_t)r*K + i] = (A[(size_t)(r-1)*K + i] == '-');
    for (c = 1; c <= N; ++c)
        for (j = 0; j < L; ++j) dashB[(size_t)c*L + j] = (B[(size_t)(c-1)*L + j] == '-');

    tb = tb_out ? tb_out : (uint8_t *)mzo__alloc(ar, (size_t)cells, 0);
    rowoff = (int64_t *)mzo__alloc(ar, sizeof(int64_t) * (size_t)(M + 1), 0);
    dp = (tri *)mzo__alloc(ar, sizeof(tri) * (size_t)(N + 1), 0);

    /* row 0: only insertions; extension charged, never an open (mz_yama.c:83-94) */
    tp = tb;
    rowoff[0] = 0;
    dp[0].C = dp[0].D = dp[0].I = 0;
    *tp++ = 0;
    for (c = 1; c <= RB[0]; ++c) {
        int nb = 0;
        for (j = 0; j < L; ++j) nb += !dashB[(size_t)c*L + j];
        dp[c].C = dp[c].D = MZO_NEG;
        dp[c].I = dp[c-1].I - nb * K * ge;
        *tp++ = (uint8_t)(MZO_FI << 4);
    }
    for (; c <= N; ++c) dp[c] = TRI_NEG;

    left = TRI_NEG;
    for (r = 1; r <= M; ++r) {
        const int lo = LB[r], hi = RB[r], lb1 = LB[r-1], lb2 = r > 1 ? LB[r-2] : 0;
        const uint8_t *ua = dashA + (size_t)r*K;          /* dash flags of A column r      */
        const uint8_t *sa = dashA + (size_t)(r-1)*K;      /* ... of A column r-1 (0s if r=1) */
        tri diag;
        int na = 0;

        for (i = 0; i < K; ++i) na += !ua[i];
        rowoff[r] = tp - tb;
        /* (r-1, lo-1) exists only if the previous row's band reaches that far left */
        diag = (lo - 1 >= lb1 && lo >= 1) ? dp[lo-1] : TRI_NEG;
        left = TRI_NEG;

        for (c = lo; c <= hi; ++c) {
            const uint8_t *vb = dashB + (size_t)c*L;                  /* column c   */
            const uint8_t *tbm = dashB + (size_t)(c > 0 ? c-1 : 0)*L; /* column c-1 */
            const tri up = dp[c];             /* (r-1,c): initial NEG beyond RB[r-1] */
            unsigned fc = 0, fd = 0, fi = 0;
            int32_t x, y, z;
            tri now;
            int nb = 0;

            for (j = 0; j < L; ++j) nb += (c > 0) ? !vb[j] : 0;

            /* ---- I: arrive from (r, c-1) */
            if (c > lo) {
                x = left.C; y = left.D; z = left.I;
                if (r < M) {                       /* no open for trailing end-gaps */
                    if (c > lb1 + 1) x -= gap_sum(sc, ua, tbm, NULL, vb, K, L);
                    y -= gap_sum(sc, ua, NULL, NULL, vb, K, L);
                    if (c > lo + 1)  z -= gap_sum(sc, NULL, tbm, NULL, vb, K, L);
                }
                now.I = choose(x, y, z, &fi) - nb * K * ge;
            } else {
                now.I = MZO_NEG;
            }

            /* ---- C: arrive from (r-1, c-1) */
            if (c > lb1) {
                x = diag.C; y = diag.D; z = diag.I;
                if (c > 1) {                       /* no open entering column 1 */
                    if (r > 1 && c > lb2 + 1) x -= gap_sum(sc, sa, tbm, ua, vb, K, L);
                    if (r > 1)                y -= gap_sum(sc, sa, NULL, ua, vb, K, L);
                    if (c > lb1 + 1)          z -= gap_sum(sc, NULL, tbm, ua, vb, K, L);
                }
                now.C = choose(x, y, z, &fc);
                for (i = 0; i < K; ++i)
                    for (j = 0; j < L; ++j)
                        now.C += sc->ss[A[(size_t)(r-1)*K + i] & 127][B[(size_t)(c-1)*L + j] & 127];
            } else {
                now.C = MZO_NEG;
            }

            /* ---- D: arrive from (r-1, c) */
            x = up.C; y = up.D; z = up.I;
            if (c > 0 && c < N) {                  /* no open in the first/last column */
                if (r > 1 && c > lb2) x -= gap_sum(sc, sa, vb, ua, NULL, K, L);
                if (r > 1)            y -= gap_sum(sc, sa, NULL, ua, NULL, K, L);
                if (c > lb1)          z -= gap_sum(sc, NULL, vb, ua, NULL, K, L);
            }
            now.D = choose(x, y, z, &fd) - na * L * ge;

            diag = up;
            dp[c] = now;
            left = now;
            *tp++ = (uint8_t)(fc | (fd << 2) | (fi << 4));
        }
    }

    if (final3) { final3[0] = left.C; final3[1] = left.D; final3[2] = left.I; }
    rc = mzo__trace_emit(ar, A, K, M, B, L, N, LB, tb, rowoff, left, out, OM);

    mzo__free(ar, dp); mzo__free(ar, rowoff); mzo__free(ar, dashA); mzo__free(ar, dashB);
    if (!tb_out) mzo__free(ar, tb);
    return rc;
}

/* ------------------------------------------------------------------ batch (CPU baseline) */
int mzo_yama_batch(int n, const int *K, const int *L, const int *M, const int *N,
                   const int64_t *offA, const int64_t *offB, const int64_t *offBand,
                   const uint8_t *poolA, const uint8_t *poolB, const int *poolLB, const int *poolRB,
                   const mzo_scores *sc, int variant, int threads,
                   int *om, uint64_t *hash, int64_t *cells_done)
{
    int bad = 0;
    int64_t total = 0;
    size_t need = 0;
    int p;

    for (p = 0; p < n; ++p) {          /* scratch for the largest pair (full-grid bound on cells is too big: use the band) */
        int64_t cells = 0;
        size_t b;
        if (mzo_yama_check(M[p], N[p], poolLB + offBand[p], poolRB + offBand[p], &cells, NULL)) cells = 0;
        b = mzo__scratch_bytes(K[p], L[p], M[p], N[p], cells);
        if (b > need) need = b;
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel reduction(+:bad, total)
    {
        mzo_arena ar;
        ar.base = (char *)malloc(need + 4096);
        ar.cap = need + 4096;
#pragma omp for schedule(dynamic, 1)
        for (p = 0; p < n; ++p) {
            uint8_t *out;
            int64_t cells = 0;
            int m_new = 0, rc;
            ar.used = 0;
            out = (uint8_t *)mzo__alloc(&ar, (size_t)(M[p] + N[p]) * (size_t)(K[p] + L[p]) + 1, 0);
            if (variant == 0)
                rc = mzo__faithful(&ar, poolA + offA[p], K[p], M[p], poolB + offB[p], L[p], N[p],
                                   poolLB + offBand[p], poolRB + offBand[p], sc, out, &m_new, NULL, NULL);
            else
                rc = mzo__profile(&ar, poolA + offA[p], K[p], M[p], poolB + offB[p], L[p], N[p],
                                  poolLB + offBand[p], poolRB + offBand[p], sc, out, &m_new, NULL, NULL);
            if (rc) {
                bad++;
                om[p] = -rc;
                hash[p] = 0;
            } else {
                uint64_t h = mzo_fnv1a((const uint8_t *)&m_new, 4, 0);
                om[p] = m_new;
                hash[p] = mzo_fnv1a(out, (int64_t)m_new * (K[p] + L[p]), h);
                mzo_yama_check(M[p], N[p], poolLB + offBand[p], poolRB + offBand[p], &cells, NULL);
                total += cells;
            }
        }
        free(ar.base);
    }
    if (cells_done) *cells_done = total;
    return bad;
}
