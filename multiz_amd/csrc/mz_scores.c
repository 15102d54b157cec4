/* mz_scores.c -- the score tables of the path, with the reference's names and globals
 * (reference mz_scores.c:9-122, mz_scores.h:8-19) so that stock drivers link unchanged, plus
 * mafScoreRange() (reference mz_scores.c:124-152) which mafBuild() uses to score a merged block.
 */
#include <ctype.h>
#include <stdlib.h>
#include "../../include/mz_scores.h"

__attribute__((noreturn)) void mz_fatalf(const char *fmt, ...);
extern int mz_scores_explicit;

int **ss, *gop;
int gap_open, gap_extend;

struct table { int **sub; int *open16; int built; };
static struct table t70, t85;

static void build(struct table *t, const int m[4][4], int filler, int open_, int ext)
{
    static const char nt[4] = { 'A', 'C', 'G', 'T' };
    int a, b, x;

    t->open16 = (int *)malloc(16 * sizeof(int));
    t->sub = (int **)malloc(128 * sizeof(int *));
    t->sub[0] = (int *)malloc(128 * 128 * sizeof(int));
    if (!t->open16 || !t->sub || !t->sub[0]) mz_fatalf("Ran out of memory trying to allocate %lu.", 128ul * 128 * sizeof(int));
    for (a = 1; a < 128; ++a) t->sub[a] = t->sub[0] + 128 * a;

    for (a = 0; a < 128; ++a)
        for (b = 0; b < 128; ++b)
            t->sub[a][b] = filler;                       /* unspecified pair, mz_scores.c:29 */
    for (a = 0; a < 4; ++a)
        for (b = 0; b < 4; ++b) {
            int U1 = nt[a], U2 = nt[b], l1 = tolower(U1), l2 = tolower(U2);
            t->sub[U1][U2] = t->sub[U1][l2] = t->sub[l1][U2] = t->sub[l1][l2] = m[a][b];
        }
    for (x = 0; x < 128; ++x)
        t->sub['-'][x] = t->sub[x]['-'] = -ext;          /* residue against a dash */
    t->sub['-']['-'] = 0;
    /* quasi-natural gap opens: the current column pair (u,v) has exactly one dash and the
     * previous pair (s,t) was not that same one-sided pattern (mz_scores.c:56-79) */
    for (x = 0; x < 16; ++x) {
        int s = (x >> 3) & 1, tt = (x >> 2) & 1, u = (x >> 1) & 1, v = x & 1;
        t->open16[x] = (u != v && !(s == u && tt == v)) ? open_ : 0;
    }
    t->built = 1;
}

void init_scores70(void)
{
    static const int hox70[4][4] = {
        {   91, -114,  -31, -123 }, { -114,  100, -125,  -31 },
        {  -31, -125,  100, -114 }, { -123,  -31, -114,   91 } };
    if (!t70.built) build(&t70, hox70, -100, 400, 30);
    ss = t70.sub; gop = t70.open16; gap_open = 400; gap_extend = 30;
    mz_scores_explicit = 0;
}

void init_scores85(void)
{
    static const int hox85[4][4] = {
        {   86, -135,  -68, -157 }, { -135,  100, -148,  -68 },
        {  -68, -148,  100, -135 }, { -157,  -68, -135,   86 } };
    if (!t85.built) build(&t85, hox85, -100, 600, 50);
    ss = t85.sub; gop = t85.open16; gap_open = 600; gap_extend = 50;
    mz_scores_explicit = 0;
}

/* ---- mafScoreRange() in O(rows) per column instead of O(rows^2).
 * When ss is constant on the byte classes {A,C,G,T,-,other} and symmetric, and gop is symmetric under
 * exchanging the two rows (all true of both reference tables), the sum over unordered row pairs depends only on
 * how many rows of a column fall into each class, and the gap term only on how many rows are in each of the
 * four (previous dash, current dash) states -- the same profile arithmetic the DP kernels use.  Every addend
 * is an integer and the reference adds them one at a time into a double, so any order gives the same double
 * below 2^53.  Tables without that structure (or bytes >= 128) take the literal double loop. */
static struct { int **ss_seen; int *gop_seen; int ok; long long S[6][6]; long long g[4][4]; unsigned char cls[128]; } prof;

static int class6(int ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-': return 4;
    default:  return 5;
    }
}

/* (re)derive the class model from the current ss / gop; call once outside any parallel region */
void mz_score_profile_sync(void)
{
    static const unsigned char rep[6] = { 'A', 'C', 'G', 'T', '-', 'N' };
    int a, b, ok = 1;
    if (prof.ss_seen == ss && prof.gop_seen == gop) return;
    for (a = 0; a < 128; ++a) prof.cls[a] = (unsigned char)class6(a);
    for (a = 0; a < 6; ++a)
        for (b = 0; b < 6; ++b) prof.S[a][b] = ss[rep[a]][rep[b]];
    for (a = 0; a < 128 && ok; ++a)
        for (b = 0; b < 128; ++b)
            if (ss[a][b] != prof.S[prof.cls[a]][prof.cls[b]] || ss[a][b] != ss[b][a]) { ok = 0; break; }
    for (a = 0; a < 4; ++a)                                /* state = previous dash << 1 | current dash */
        for (b = 0; b < 4; ++b) {
            prof.g[a][b] = GAP(a >> 1, b >> 1, a & 1, b & 1);
            if (prof.g[a][b] != GAP(b >> 1, a >> 1, b & 1, a & 1)) ok = 0;
        }
    prof.ok = ok;
    prof.ss_seen = ss; prof.gop_seen = gop;
}

static int score_range_profile(struct mafAli *maf, int start, int size, double *out)
{
    struct mafComp *p;
    unsigned short (*cnt)[6], (*st)[4];
    double total = 0.0;
    int i, k, l;
    if (!prof.ok) return 0;
    cnt = (unsigned short (*)[6])calloc((size_t)size, sizeof *cnt);
    st = (unsigned short (*)[4])calloc((size_t)size, sizeof *st);
    if (!cnt || !st) { free(cnt); free(st); return 0; }
    for (p = maf->components, k = 0; p != NULL; p = p->next, ++k) {
        const unsigned char *t = (const unsigned char *)p->text + start;
        int prev = (start > 0) ? (t[-1] == '-') : 0;
        if (k >= 65535) { free(cnt); free(st); return 0; }
        for (i = 0; i < size; ++i) {
            const unsigned ch = t[i];
            const int cur = ch == '-';
            if (ch >= 128) { free(cnt); free(st); return 0; }
            ++cnt[i][prof.cls[ch]];
            ++st[i][prev << 1 | cur];
            prev = cur;
        }
    }
    for (i = 0; i < size; ++i) {
        long long s = 0;
        for (k = 0; k < 6; ++k) {
            const long long ck = cnt[i][k];
            if (!ck) continue;
            s += prof.S[k][k] * (ck * (ck - 1) / 2);
            for (l = k + 1; l < 6; ++l) s += prof.S[k][l] * ck * cnt[i][l];
        }
        if (start + i > 0)
            for (k = 0; k < 4; ++k) {
                const long long ck = st[i][k];
                if (!ck) continue;
                s -= prof.g[k][k] * (ck * (ck - 1) / 2);
                for (l = k + 1; l < 4; ++l) s -= prof.g[k][l] * ck * st[i][l];
            }
        total += (double)s;
    }
    free(cnt); free(st);
    *out = total;
    return 1;
}

/* sum-of-pairs score of columns start..start+size-1 of a block: substitution score of every
 * unordered row pair minus the gap-open the column pair (i-1, i) incurs for it.  Accumulated in
 * double like the reference (every addend is an int, so the sum is exact below 2^53). */
double mafScoreRange(struct mafAli *maf, int start, int size)
{
    struct mafComp *p, *q;
    double total = 0.0;
    int i;

    if (start < 0 || size <= 0 || start + size > maf->textSize)
        mz_fatalf("mafScoreRange: start = %d, size = %d, textSize = %d\n", start, size, maf->textSize);
    if (ss == NULL)
        mz_fatalf("mafScoreRange: scores not initialized");
    if (prof.ss_seen != ss || prof.gop_seen != gop) {
#pragma omp critical(mz_score_profile)
        mz_score_profile_sync();
    }
    if (score_range_profile(maf, start, size, &total)) return total;
    total = 0.0;
    for (i = start; i < start + size; ++i)
        for (p = maf->components; p != NULL; p = p->next) {
            const unsigned char x = (unsigned char)p->text[i];
            for (q = p->next; q != NULL; q = q->next) {
                const unsigned char y = (unsigned char)q->text[i];
                total += SS(x, y);
                if (i > 0)
                    total -= GAP2(p->text[i-1], q->text[i-1], x, y);
            }
        }
    return total;
}
