/* oracle/yama_profile_oracle.c -- integer-exact O(1)-per-cell restatement of yama().
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity status: PINNED (tests compare it with
 * mzo_yama_faithful, with oracle/_ref/libref.so and with tests/golden/).
 *
 * Every per-cell quantity of reference mz_yama.c:113-242 is a double sum over (row of A,
 * row of B) of a table entry that depends only on the byte *class* of each operand, so it
 * equals a small bilinear form of per-column counts.  Integer
 * addition is associative, hence the values -- and therefore every comparison, flag and
 * traceback byte -- are identical to the faithful evaluation as long as nothing overflows
 * (same condition as the reference itself).  This file is the executable specification
 * of what the HIP kernel computes per cell.
 *
 * Score model required (checked by mzo_profile_model): ss[][] must be constant on the six
 * byte classes {A/a, C/c, G/g, T/t, '-', everything else} and gop[] must be
 * open * [u != v] * [(s,t) != (u,v)] -- true for both reference tables
 * (mz_scores.c:34-81).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct { int32_t C, D, I; } tri;
#ifdef MZO_SPREAD_STATS              /* tests/tools/spread.c: how far apart the states of one band row lie (HISTORY section 10) */
#include "spread_hook.h"
#endif
static const tri TRI_NEG = { MZO_NEG, MZO_NEG, MZO_NEG };

void *mzo__alloc(mzo_arena *ar, size_t bytes, int zero);
void mzo__free(mzo_arena *ar, void *p);
int mzo__trace_emit(mzo_arena *ar, const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                    const int *LB, const uint8_t *tb, const int64_t *rowoff,
                    tri last, uint8_t *out, int *OM);

static inline int cls_of(unsigned ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-':           return 4;
    default:            return 5;
    }
}

/* per-column profile: class counts + the four gap counts of appendix A.4 */
typedef struct {
    int32_t cnt[6];
    int32_t n;      /* non-dash rows                                   */
    int32_t d;      /* dash rows                                       */
    int32_t p00;    /* non-dash here and non-dash in the previous col  */
    int32_t p11;    /* dash here and dash in the previous column       */
    int32_t w[6];   /* A side only: cnt^T * S6  (score row vector)     */
} prof;

static void build_profiles(const uint8_t *X, int rows, int cols, prof *P /* [cols+1], 1-based */)
{
    int c, i;
    memset(&P[0], 0, sizeof(prof));
    for (c = 1; c <= cols; ++c) {
        const uint8_t *cur = X + (size_t)(c-1)*rows;
        const uint8_t *prv = c > 1 ? cur - rows : NULL;
        prof *p = &P[c];
        memset(p, 0, sizeof(*p));
        for (i = 0; i < rows; ++i) {
            int dash = cur[i] == '-';
            int pdash = prv ? prv[i] == '-' : 0;    /* "previous of column 1" is non-dash */
            p->cnt[cls_of(cur[i])]++;
            if (dash) { p->d++; if (pdash) p->p11++; }
            else      { p->n++; if (!pdash) p->p00++; }
        }
    }
}

/* extract the 6x6 class matrix and the scalar open penalty; 0 if the tables do not
 * have the structure this restatement (and the GPU kernel) relies on */
int mzo_profile_model(const mzo_scores *sc, int S6[6][6], int *open_)
{
    static const unsigned char rep[6] = { 'A', 'C', 'G', 'T', '-', 'N' };
    int a, b, x;
    for (a = 0; a < 6; ++a)
        for (b = 0; b < 6; ++b)
            S6[a][b] = sc->ss[rep[a]][rep[b]];
    for (a = 0; a < 128; ++a)
        for (b = 0; b < 128; ++b)
            if (sc->ss[a][b] != S6[cls_of(a)][cls_of(b)]) return 0;
    *open_ = sc->gop[1];
    for (x = 0; x < 16; ++x) {
        int s = (x >> 3) & 1, t = (x >> 2) & 1, u = (x >> 1) & 1, v = x & 1;
        int want = (u != v && !(s == u && t == v)) ? *open_ : 0;
        if (sc->gop[x] != want) return 0;
    }
    return 1;
}

static inline int32_t choose(int32_t x, int32_t y, int32_t z, unsigned *flag)
{
    if (x >= y && x >= z) { *flag = MZO_FC; return x; }
    if (y > z)            { *flag = MZO_FD; return y; }
    *flag = MZO_FI;
    return z;
}

int mzo__profile(mzo_arena *ar, const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                 const int *LB, const int *RB, const mzo_scores *sc,
                 uint8_t *out, int *OM, int32_t *final3, uint8_t *tb_out);

int mzo_yama_profile(const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                     const int *LB, const int *RB, const mzo_scores *sc,
                     uint8_t *out, int *OM, int32_t *final3, uint8_t *tb_out)
{
    return mzo__profile(NULL, A, K, M, B, L, N, LB, RB, sc, out, OM, final3, tb_out);
}

int mzo__profile(mzo_arena *ar, const uint8_t *A, int K, int M, const uint8_t *B, int L, int N,
                 const int *LB, const int *RB, const mzo_scores *sc,
                 uint8_t *out, int *OM, int32_t *final3, uint8_t *tb_out)
{
    int S6[6][6], go, rc, r, c, a, b;
    const int ge = sc->gap_extend;
    int64_t cells, *rowoff;
    prof *PA, *PB;
    uint8_t *tb, *tp;
    tri *dp, left;

    if (!mzo_profile_model(sc, S6, &go)) return -1;
    rc = mzo_yama_check(M, N, LB, RB, &cells, NULL);
    if (rc) return rc;

    PA = (prof *)mzo__alloc(ar, sizeof(prof) * (size_t)(M + 1), 0);
    PB = (prof *)mzo__alloc(ar, sizeof(prof) * (size_t)(N + 1), 0);
    build_profiles(A, K, M, PA);
    build_profiles(B, L, N, PB);
    for (r = 1; r <= M; ++r)
        for (b = 0; b < 6; ++b) {
            int32_t w = 0;
            for (a = 0; a < 6; ++a) w += PA[r].cnt[a] * S6[a][b];
            PA[r].w[b] = w;
        }

    tb = tb_out ? tb_out : (uint8_t *)mzo__alloc(ar, (size_t)cells, 0);
    rowoff = (int64_t *)mzo__alloc(ar, sizeof(int64_t) * (size_t)(M + 1), 0);
    dp = (tri *)mzo__alloc(ar, sizeof(tri) * (size_t)(N + 1), 0);

    tp = tb;
    rowoff[0] = 0;
    dp[0].C = dp[0].D = dp[0].I = 0;
    *tp++ = 0;
    for (c = 1; c <= RB[0]; ++c) {
        dp[c].C = dp[c].D = MZO_NEG;
        dp[c].I = dp[c-1].I - PB[c].n * K * ge;
        *tp++ = (uint8_t)(MZO_FI << 4);
    }
    for (; c <= N; ++c) dp[c] = TRI_NEG;

    left = TRI_NEG;
    for (r = 1; r <= M; ++r) {
        const int lo = LB[r], hi = RB[r], lb1 = LB[r-1], lb2 = r > 1 ? LB[r-2] : 0;
        const prof *pa = &PA[r];
        /* A-side counts for the (r-1, r) column pair.  For r == 1 the previous column
         * counts as non-dash, which build_profiles already encodes in p00/p11. */
        const int32_t nA = pa->n, dA = pa->d, a00 = pa->p00, a11 = pa->p11;
        tri diag = (lo - 1 >= lb1 && lo >= 1) ? dp[lo-1] : TRI_NEG;

        rowoff[r] = tp - tb;
        left = TRI_NEG;
        for (c = lo; c <= hi; ++c) {
            const prof *pb = &PB[c];
            const int32_t nB = pb->n, dB = pb->d, b00 = pb->p00, b11 = pb->p11;
            const tri up = dp[c];
            unsigned fc = 0, fd = 0, fi = 0;
            int32_t x, y, z;
            tri now;

            if (c > lo) {
                x = left.C; y = left.D; z = left.I;
                if (r < M) {
                    if (c > lb1 + 1) x -= go * (K * nB - dA * b00);
                    y -= go * (K * nB);
                    if (c > lo + 1)  z -= go * (K * (nB - b00));
                }
                now.I = choose(x, y, z, &fi) - nB * K * ge;
            } else {
                now.I = MZO_NEG;
            }

            if (c > lb1) {
                int32_t sig = 0;
                x = diag.C; y = diag.D; z = diag.I;
                if (c > 1) {
                    if (r > 1 && c > lb2 + 1) x -= go * (nA * dB - a00 * b11 + dA * nB - a11 * b00);
                    if (r > 1)                y -= go * ((nA - a00) * dB + dA * nB);
                    if (c > lb1 + 1)          z -= go * (nA * dB + dA * (nB - b00));
                }
                for (b = 0; b < 6; ++b) sig += pa->w[b] * pb->cnt[b];
                now.C = choose(x, y, z, &fc) + sig;
            } else {
                now.C = MZO_NEG;
            }

            x = up.C; y = up.D; z = up.I;
            if (c > 0 && c < N) {
                if (r > 1 && c > lb2) x -= go * (nA * L - a00 * dB);
                if (r > 1)            y -= go * (L * (nA - a00));
                if (c > lb1)          z -= go * (L * nA);
            }
            now.D = choose(x, y, z, &fd) - nA * L * ge;

            diag = up;
            dp[c] = now;
            left = now;
            *tp++ = (uint8_t)(fc | (fd << 2) | (fi << 4));
#ifdef MZO_SPREAD_STATS
            MZO_SPREAD_CELL(r, c, lo, now);
#endif
        }
#ifdef MZO_SPREAD_STATS
        MZO_SPREAD_ROW(r);
#endif
    }

    if (final3) { final3[0] = left.C; final3[1] = left.D; final3[2] = left.I; }
    rc = mzo__trace_emit(ar, A, K, M, B, L, N, LB, tb, rowoff, left, out, OM);

    mzo__free(ar, dp); mzo__free(ar, rowoff); mzo__free(ar, PA); mzo__free(ar, PB);
    if (!tb_out) mzo__free(ar, tb);
    return rc;
}
