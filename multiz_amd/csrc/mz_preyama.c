/* mz_preyama.c -- the block-pair adapter around yama(): band construction and (next commit)
 * pre_yama() itself, with the reference's names so that stock drivers link unchanged
 * (reference mz_preyama.c:17-35 here).
 */
#define _GNU_SOURCE            /* RTLD_DEFAULT (pre_yama2) */
#include <stdlib.h>
#include "../../include/mz_preyama.h"

/* the reference's 1-based arrays are freed as free(X + 1): gcc cannot see that X + 1 is the malloc()ed address */
#pragma GCC diagnostic ignored "-Wfree-nonheap-object"

/* Turn the raw per-row column range implied by the shared reference row into a legal DP band
 * (reference mz_preyama.c:17-35): make LB a running maximum and RB a running minimum from the
 * right, then widen both by rad = min(M, radius) rows/columns. */
void smooth(int *LB, int *RB, int M, int N, int radius)
{
    const int rad = M < radius ? M : radius;
    int i, run;

    for (i = 0, run = 0; i <= M; ++i) {
        if (LB[i] > run) run = LB[i];
        LB[i] = run;
    }
    for (i = M, run = N; i >= 0; --i) {
        if (RB[i] < run) run = RB[i];
        RB[i] = run;
    }
    /* descending i: LB[i-rad] is still un-widened when it is read */
    for (i = M; i > rad; --i) {
        int a = LB[i] - rad;
        if (a < 0) a = 0;
        LB[i] = a < LB[i - rad] ? a : LB[i - rad];
    }
    for (; i >= 0; --i) LB[i] = 0;
    /* ascending i: RB[i+rad] is still un-widened when it is read */
    for (i = 0; i < M - rad; ++i) {
        int a = RB[i] + rad;
        if (a > N) a = N;
        RB[i] = a > RB[i + rad] ? a : RB[i + rad];
    }
    for (; i <= M; ++i) RB[i] = N;
}

/* ------------------------------------------------------------------------------------------------
 * pre_yama() and its helpers.  Data structures are the reference's (1-based arrays of column
 * pointers over one contiguous buffer) because yama(), rmColDash(), mapping() and mafBuild() are
 * link-visible with exactly those conventions.
 * ------------------------------------------------------------------------------------------------ */
#include <string.h>
#include "../../include/mz_scores.h"

__attribute__((noreturn)) void mz_fatalf(const char *fmt, ...);

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)n);
    return p;
}

/* 1-based column pointers over `cols` contiguous columns of `rows` bytes.  One spare byte follows
 * the data and holds a non-dash: see the note on mapping() in pre_yama(). */
static uchar **cols_new(int cols, int rows)
{
    uchar **X = (uchar **)xmalloc((size_t)(cols > 0 ? cols : 1) * sizeof(uchar *)) - 1;
    int i;
    X[1] = (uchar *)xmalloc((size_t)cols * rows + 1);
    X[1][(size_t)cols * rows] = 'N';
    for (i = 2; i <= cols; ++i) X[i] = X[i-1] + rows;
    return X;
}
static void cols_free(uchar **X) { free(X[1]); free(X + 1); }

/* Turn yama()'s merged columns back into a block (reference mz_preyama.c:38-81): row i of the new
 * block takes its bookkeeping from the i-th row of a2, continuing into a3 (from a3's second row
 * unless top != 0); rows left without a base are dropped; the block is rescored. */
struct mafAli *mafBuild(uchar **A_new, int nrow, int ncol, struct mafAli *a2, int cbeg2,
                        struct mafAli *a3, int cbeg3, int top)
{
    struct mafAli *blk = (struct mafAli *)xmalloc(sizeof *blk);
    struct mafComp *src = a2->components, *tail = NULL, *nc;
    int skip = cbeg2, i, j;

    memset(blk, 0, sizeof *blk);
    blk->textSize = ncol;
    for (i = 0; i < nrow; ++i, src = src->next) {
        int start;
        if (src == NULL) {                         /* rows of the first block exhausted */
            src = top == 0 ? a3->components->next : a3->components;
            skip = cbeg3;
        }
        for (start = src->start, j = 0; j < skip; ++j)
            start += src->text[j] != '-';          /* bases of this row left of the slice */
        nc = mafCpyComp(src);
        nc->start = start;
        nc->size = 0;
        nc->text = (char *)xmalloc((size_t)ncol + 1);
        for (j = 0; j < ncol; ++j) {
            nc->text[j] = (char)A_new[j + 1][i];
            nc->size += nc->text[j] != '-';
        }
        nc->text[ncol] = '\0';
        if (nc->size == 0) { mafCompFree(&nc); continue; }
        if (tail) tail->next = nc; else blk->components = nc;
        tail = nc;
    }
    if (!blk->components) { free(blk); return NULL; }
    blk->score = mafScoreRange(blk, 0, ncol);
    return blk;
}

/* Delete the columns of X (1-based, `row` bytes each) that are all dashes, compacting in place;
 * *N becomes the new column count.  Returns map[1..oldN]: new index, or -1 if removed
 * (reference mz_preyama.c:87-108). */
int *rmColDash(uchar **X, int *N, int row)
{
    const int n = *N;
    int *map = (int *)xmalloc(((size_t)n + 1) * sizeof(int));
    int i, j, kept = 0;

    for (i = 1; i <= n; ++i) {
        for (j = 0; j < row && X[i][j] == '-'; ++j)
            ;
        if (j == row) { map[i] = -1; continue; }
        ++kept;
        if (kept != i) memcpy(X[kept], X[i], (size_t)row);
        map[i] = kept;
    }
    *N = kept;
    return map;
}

/* Pair up, in order, the columns of A (rows a_row1..a_row2, columns a_col1..a_col2) that are not
 * all-dash with the likewise non-all-dash columns of B; map[column of A] = column of B, -1 for
 * skipped columns (reference mz_preyama.c:111-148; indices into the result are A's own column
 * numbers, which the callers always start at 1). */
int *mapping(uchar **A, int a_row1, int a_row2, int a_col1, int a_col2,
             uchar **B, int b_row1, int b_row2, int b_col1, int b_col2)
{
    int *map, i, k, j, l;

    if (a_row2 - a_row1 != b_row2 - b_row1)
        mz_fatalf("not equal rows:!\n");
    map = (int *)xmalloc(((size_t)(a_col2 - a_col1) + 2) * sizeof(int));
    for (i = a_col1; i <= a_col2; ++i) map[i - a_col1 + 1] = -1;

    i = a_col1; k = b_col1;
    while (i <= a_col2 && k <= b_col2) {
        int a_live = 0, b_live = 0;
        for (; i <= a_col2; ++i) {                 /* next column of A with a base in the row range */
            for (j = a_row1; j <= a_row2 && A[i][j] == '-'; ++j)
                ;
            if (j <= a_row2) { a_live = 1; break; }
        }
        for (; k <= b_col2; ++k) {
            for (l = b_row1; l <= b_row2 && B[k][l] == '-'; ++l)
                ;
            if (l <= b_row2) { b_live = 1; break; }
        }
        if (a_live && b_live) map[i] = k;
        ++i; ++k;
    }
    return map;
}

/* fold the (row of first block -> column of second block) correspondences into raw bounds:
 * 0 / hi_unset mean "not set yet", exactly as the reference encodes it (mz_preyama.c:252-255) */
static void bound_note(int *LB, int *RB, int row, int col, int hi_unset)
{
    if (LB[row] == 0 || LB[row] > col) LB[row] = col;
    if (RB[row] == hi_unset || RB[row] < col) RB[row] = col;
}

/* ------------------------------------------------------------------------------------------------
 * pre_yama() in stages, so that a driver can run the yama() calls of many block pairs as one GPU
 * batch (mz_multiz.c).  A stage never looks at anything but its own inputs and the
 * merged columns handed to it, exactly as the reference's straight-line code does
 * (mz_preyama.c:152-359); pre_yama() below is the three stages with a batch of one in between.
 * ------------------------------------------------------------------------------------------------ */
#include "mz_py.h"

static void py_zero(mz_py *p) { memset(p, 0, sizeof *p); }

void mz_py_free(mz_py *p)
{
    if (p->A) cols_free(p->A);
    if (p->B) cols_free(p->B);
    if (p->merged) { if (p->borrowed) free(p->merged + 1); else cols_free(p->merged); }
    if (p->ref1) cols_free(p->ref1);
    if (p->ref2) cols_free(p->ref2);
    if (p->merged2) { if (p->borrowed) free(p->merged2 + 1); else cols_free(p->merged2); }
    free(p->map1); free(p->map2); free(p->LB); free(p->RB);
    py_zero(p);
}

/* wrap the contiguous merged columns of a yama result (malloc'ed, om columns of `rows` bytes) in the
 * reference's 1-based pointer array */
static uchar **cols_wrap(uchar *flat, int om, int rows)
{
    uchar **X = (uchar **)xmalloc((size_t)(om > 0 ? om : 1) * sizeof(uchar *)) - 1;
    int i;
    X[1] = flat;
    for (i = 2; i <= om; ++i) X[i] = X[i - 1] + rows;
    return X;
}

/* stage 1: slice, pack, derive the band.  MZ_PY_NULL: pre_yama() returns NULL (after writing a2's
 * slice to fpw2 when a1 has nothing left to align, mz_preyama.c:193-196); MZ_PY_JOB: run p->job. */
int mz_py_begin(mz_py *p, struct mafAli *a1, struct mafAli *a2, int beg, int end, int radius, int v, FILE *fpw2)
{
    struct mafComp *c;
    int K = 0, L = 0, M, N, i, j, r;
    int cend1;

    py_zero(p);
    p->a1 = a1; p->a2 = a2; p->radius = radius; p->v = v;
    for (c = a1->components; c; c = c->next) ++K;
    for (c = a2->components->next; c; c = c->next) ++L;     /* a2's top row only guides */

    p->cbeg1 = mafPos2Col(a1->components, beg, a1->textSize);
    cend1 = mafPos2Col(a1->components, end, a1->textSize);
    p->cbeg2 = mafPos2Col(a2->components, beg, a2->textSize);
    p->cend2 = mafPos2Col(a2->components, end, a2->textSize);
    M = p->M_all = cend1 - p->cbeg1 + 1;
    N = p->N_all = p->cend2 - p->cbeg2 + 1;

    /* second block without its reference row, column-major, all-dash columns removed */
    p->B = cols_new(N, L);
    for (i = 1; i <= N; ++i)
        for (r = 0, c = a2->components->next; r < L; ++r, c = c->next)
            p->B[i][r] = (uchar)c->text[p->cbeg2 + i - 1];
    p->map2 = rmColDash(p->B, &N, L);
    if (N < 1) { mz_py_free(p); return MZ_PY_NULL; }

    if (v == 0) --K;                                          /* a1's reference row is aligned later */
    if (K == 0) {
        if (fpw2) print_part_ali_col(a2, p->cbeg2, p->cend2, fpw2);
        mz_py_free(p);
        return MZ_PY_NULL;
    }
    p->A = cols_new(M, K);
    for (i = 1; i <= M; ++i)
        for (r = 0, c = v == 0 ? a1->components->next : a1->components; r < K; ++r, c = c->next)
            p->A[i][r] = (uchar)c->text[p->cbeg1 + i - 1];
    if (v == 0) {
        p->map1 = rmColDash(p->A, &M, K);
        if (M < 1) { mz_py_free(p); return MZ_PY_NULL; }
    } else {
        p->map1 = (int *)xmalloc(((size_t)M + 1) * sizeof(int));
        for (i = 1; i <= M; ++i) p->map1[i] = i;
    }

    /* band from the shared reference row: walk both copies of it base by base */
    p->LB = (int *)xmalloc(((size_t)M + 1) * sizeof(int));
    p->RB = (int *)xmalloc(((size_t)M + 1) * sizeof(int));
    for (i = 0; i <= M; ++i) { p->LB[i] = 0; p->RB[i] = N; }
    for (i = p->cbeg1, j = p->cbeg2; i <= cend1; ++i, ++j) {
        int ra, cb;
        while (a1->components->text[i] == '-') ++i;
        while (a2->components->text[j] == '-') ++j;
        ra = p->map1[i - p->cbeg1 + 1];
        cb = p->map2[j - p->cbeg2 + 1];
        if (ra != -1 && cb != -1) bound_note(p->LB, p->RB, ra, cb, N);
    }
    smooth(p->LB, p->RB, M, N, radius);
    p->K = K; p->L = L; p->M = M; p->N = N;
    p->stage = 1;
    p->job.K = K; p->job.L = L; p->job.M = M; p->job.N = N;
    p->job.A = p->A[1]; p->job.B = p->B[1]; p->job.LB = p->LB; p->job.RB = p->RB;
    return MZ_PY_JOB;
}

/* stage 2 / 3: `flat` = the merged columns of p->job (om columns; ownership passes to p unless p->borrowed).
 * MZ_PY_DONE: *result is pre_yama()'s return value; MZ_PY_JOB: (v == 0) run p->job once more. */
int mz_py_step(mz_py *p, uchar *flat, int om, struct mafAli **result)
{
    struct mafAli *a1 = p->a1, *a2 = p->a2;
    const int K = p->K, L = p->L, M = p->M, N = p->N, M_all = p->M_all, N_all = p->N_all;
    int i;

    if (p->stage == 1 && p->v == 1) {
        p->merged = cols_wrap(flat, om, K + L);
        *result = mafBuild(p->merged, K + L, om, a1, p->cbeg1, a2, p->cbeg2, 0);
        mz_py_free(p);
        return MZ_PY_DONE;
    }
    if (p->stage == 1) {
        /* second stage: align a1's reference row (dashes squeezed out) against the merged block.
         * Its band is the union of two estimates: where a1's other rows went (via map1 o map4a) and
         * where a2's rows went (via map2 o map4b). */
        int *m3a, *m4a, *m3b, *m4b, *LBa, *RBa, *LBb, *RBb;
        int M3 = M_all, N3 = N_all;
        const int M_new = om;

        p->merged = cols_wrap(flat, om, K + L);
        p->ref1 = cols_new(M_all, 1); p->ref2 = cols_new(N_all, 1);
        for (i = 1; i <= M_all; ++i) p->ref1[i][0] = (uchar)a1->components->text[p->cbeg1 + i - 1];
        m3a = rmColDash(p->ref1, &M3, 1);
        /* NOTE (reference mz_preyama.c:279): rows 1..K are scanned although A now has rows 0..K-1, so
         * "row K" of a column is really row 0 of the next column (and, for the last column, the byte
         * after the data: a stale pre-compaction byte, or the spare non-dash of cols_new()).  The
         * reference's outputs depend on it; reproduced literally. */
        m4a = mapping(p->A, 1, K, 1, M, p->merged, 0, K - 1, 1, M_new);
        LBa = (int *)xmalloc(((size_t)M3 + 1) * sizeof(int));
        RBa = (int *)xmalloc(((size_t)M3 + 1) * sizeof(int));
        for (i = 0; i <= M3; ++i) { LBa[i] = 0; RBa[i] = M_new; }
        for (i = 1; i <= M_all; ++i) {
            int t1 = m3a[i], t2;
            if (p->map1[i] == -1) continue;
            t2 = m4a[p->map1[i]];
            if (t1 != -1 && t2 != -1) bound_note(LBa, RBa, t1, t2, M_new);
        }
        smooth(LBa, RBa, M3, M_new, p->radius);
        free(m3a); free(m4a);

        for (i = 1; i <= N_all; ++i) p->ref2[i][0] = (uchar)a2->components->text[p->cbeg2 + i - 1];
        m3b = rmColDash(p->ref2, &N3, 1);
        m4b = mapping(p->B, 0, L - 1, 1, N, p->merged, K, K + L - 1, 1, M_new);
        LBb = (int *)xmalloc(((size_t)N3 + 1) * sizeof(int));
        RBb = (int *)xmalloc(((size_t)N3 + 1) * sizeof(int));
        for (i = 0; i <= N3; ++i) { LBb[i] = 0; RBb[i] = M_new; }
        for (i = 1; i <= N_all; ++i) {
            /* NOTE (reference mz_preyama.c:318-326): map2[i] is not tested for -1 here, so a removed
             * a2 column reads the word in front of the map; on glibc/x86-64 that is the upper half of
             * the chunk header, i.e. 0.  Reproduced as the value 0. */
            int t1 = m3b[i], t2 = p->map2[i] == -1 ? 0 : m4b[p->map2[i]];
            if (t1 != -1 && t2 != -1) bound_note(LBb, RBb, t1, t2, M_new);
        }
        smooth(LBb, RBb, N3, M_new, p->radius);
        if (M3 != N3) mz_fatalf("M3 not equals N3!!\n");
        for (i = 0; i <= M3; ++i) {
            if (LBa[i] < LBb[i]) LBb[i] = LBa[i];
            if (RBa[i] > RBb[i]) RBb[i] = RBa[i];
        }
        free(m3b); free(m4b); free(LBa); free(RBa);
        free(p->LB); free(p->RB);
        p->LB = LBb; p->RB = RBb;                           /* band of the second job */
        p->stage = 2;
        p->job.K = 1; p->job.L = K + L; p->job.M = M3; p->job.N = M_new;
        p->job.A = p->ref1[1]; p->job.B = p->merged[1]; p->job.LB = LBb; p->job.RB = RBb;
        return MZ_PY_JOB;
    }
    p->merged2 = cols_wrap(flat, om, K + L + 1);
    *result = mafBuild(p->merged2, K + L + 1, om, a1, p->cbeg1, a2, p->cbeg2, 0);
    mz_py_free(p);
    return MZ_PY_DONE;
}

/* one job on the GPU; failures end the program with the reference's messages (mz_host.c) */
void mz_py_run_one(mz_job *job, uchar **flat, int *om);

struct mafAli *pre_yama(struct mafAli *a1, struct mafAli *a2, int beg, int end, int radius, int v, FILE *fpw2)
{
    mz_py p;
    struct mafAli *result = NULL;
    uchar *flat;
    int om, st;

    st = mz_py_begin(&p, a1, a2, beg, end, radius, v, fpw2);
    while (st == MZ_PY_JOB) {
        mz_py_run_one(&p.job, &flat, &om);
        st = mz_py_step(&p, flat, om, &result);
    }
    return result;
}

/* ------------------------------------------------------------------------------------------------ pre_yama2
 * The three-block form (reference mz_preyama.c:360-521; declared in mz_preyama.h:27, no caller in the reference
 * tree): a1 is a PAIRWISE alignment of sequence X (top row) and sequence Y; a2 is a block topped by X, a3 a block
 * topped by Y; the slices of a2 over X[beg1..end1] and of a3 over Y[begN..endN] are merged, with the band taken from
 * how a1 pairs the positions of X and Y.  connectionAgreement2() (reference align_util.c:520-) decides from the
 * pairwise files in `pws` whether the merge is wanted at all; it lives in the caller's align_util.o (a weak
 * reference here: an executable that calls pre_yama2 links it, nothing else needs it). */
#include <ctype.h>
#include <stdio.h>
#include <dlfcn.h>

extern int connectionAgreement2(struct mafAli *a2, struct mafAli *a3, int cbeg2, int cend2, int cbeg3, int cend3,
                                struct pwuAliFiles *pws) __attribute__((weak));

/* ---- pre_yama2 in the same three steps as pre_yama(): describe the job, run it on the GPU, build the block.
 *
 * The band comes from the pairwise block a1, whose two rows are sequence X (the top row of a2) and sequence Y (the top
 * row of a3): column k of a1's slice has consumed tx(k) bases of X and ty(k) bases of Y, the tx-th base of X sits in
 * column colA[tx] of the a2 slice and the ty-th base of Y in column colB[ty] of the a3 slice, so column k ties row
 * colA[tx(k)] of the DP to column colB[ty(k)].  Per row the first non-zero and the last such column are the raw bounds
 * (the reference's running form of the same, mz_preyama.c:461-497, including its use of 0 as "not set"). */

/* the base columns of a row slice: col[t] (t = 1..) = 1-based slice column of its t-th base; returns the count */
static int base_columns(const char *text, int cbeg, int cend, int **cols)
{
    int *c = (int *)xmalloc(((size_t)(cend - cbeg) + 3) * sizeof(int)), n = 0, k;
    c[0] = 0;
    for (k = cbeg; k <= cend; ++k)
        if (text[k] != '-') c[++n] = k - cbeg + 1;
    *cols = c;
    return n;
}

/* the two copies of a sequence disagree at a base: the reference's report (mz_preyama.c:361-383: plain stderr, no
 * program name) and exit(1) */
static void report_disagreement(char x, char y, int pos, const struct mafComp *r1, int cb1, int ce1,
                                const struct mafComp *r2, int cb2, int ce2, int which_file)
{
    const struct mafComp *rows[2] = { r1, r2 };
    const int lo[2] = { cb1, cb2 }, hi[2] = { ce1, ce2 };
    int w, k;
    fprintf(stderr, "%c != %c\n", x, y);
    fprintf(stderr, "in file 1, positions %d... of %s are:\n  ", pos, r1->src);
    for (w = 0; w < 2; ++w) {
        if (w == 1) fprintf(stderr, "while in file %d they are:\n  ", which_file);
        for (k = lo[w]; k <= hi[w]; ++k)
            if (rows[w]->text[k] != '-') fputc(rows[w]->text[k], stderr);
        fputc('\n', stderr);
    }
    exit(1);
}

static int count_rows(const struct mafAli *a) { int n = 0; const struct mafComp *c; for (c = a->components; c; c = c->next) ++n; return n; }

/* top rows of the pairwise block and of a multi-row block must be the same sequence (messages of mz_preyama.c:403-418;
 * `second`: the wording for a1's second row against a3) */
static void same_sequence_or_die(const struct mafComp *p, const struct mafComp *q, const struct mafComp *size_shown, int second)
{
    if (strcmp(p->src, q->src) != 0)
        mz_fatalf(second ? "pre_yama: first rows (2) for sequences %s != %s" : "pre_yama: first rows for sequences %s != %s", p->src, q->src);
    if (p->srcSize != q->srcSize)
        mz_fatalf(second ? "pre_yama: first row (2) srcSizes %d != %d" : "pre_yama: first row srcSizes %d != %d", p->srcSize, size_shown->srcSize);
    if (p->strand != q->strand)
        mz_fatalf(second ? "pre_yama: first rows (2) on opposite strands" : "pre_yama: first rows on opposite strands");
}

struct mafAli *pre_yama2(struct mafAli *a1, struct mafAli *a2, struct mafAli *a3, int beg1, int end1,
                         int begN, int endN, int radius, struct pwuAliFiles *pws)
{
    typedef int (*agree_fn)(struct mafAli *, struct mafAli *, int, int, int, int, struct pwuAliFiles *);
    struct mafComp *X = a1->components, *Y = X ? X->next : NULL, *topA, *topB, *c;
    struct mafAli *blk;
    mz_job job;
    uchar **A, **B, *flat;
    int *colA, *colB, *LB, *RB;
    int cb1, ce1, cbA, ceA, cbB, ceB, K, L, M, N, nA, nB, tx = 0, ty = 0, k, r, om;
    agree_fn agree;

    if (!X || !Y) mz_fatalf("pre_yama: cannot find c and c1");
    if (Y->next) mz_fatalf("pre_yama: a1 is not a pairwise alignment");
    K = count_rows(a2); L = count_rows(a3);
    if (K == 0 || L == 0) mz_fatalf("pre_yama: an alignment has 0 rows");
    topA = a2->components; topB = a3->components;
    same_sequence_or_die(X, topA, Y, 0);          /* (the reference prints Y's size as the second number here) */
    same_sequence_or_die(Y, topB, topB, 1);

    cb1 = mafPos2Col(X, beg1, a1->textSize); ce1 = mafPos2Col(X, end1, a1->textSize);
    if (cb1 != mafPos2Col(Y, begN, a1->textSize)) mz_fatalf("pre_yama: mismatch of beg1 and begN");
    if (ce1 != mafPos2Col(Y, endN, a1->textSize)) mz_fatalf("pre_yama: mismatch of end1 and endN");
    cbA = mafPos2Col(topA, beg1, a2->textSize); ceA = mafPos2Col(topA, end1, a2->textSize);
    cbB = mafPos2Col(topB, begN, a3->textSize); ceB = mafPos2Col(topB, endN, a3->textSize);

    /* the caller's connectionAgreement2() (align_util.c): bound when the library was loaded, or found now */
    agree = connectionAgreement2 ? connectionAgreement2 : (agree_fn)dlsym(RTLD_DEFAULT, "connectionAgreement2");
    if (!agree) mz_fatalf("pre_yama2: connectionAgreement2() (align_util.c) is not linked in");
    if (agree(a2, a3, cbA, ceA, cbB, ceB, pws) == 0) return NULL;

    M = ceA - cbA + 1; N = ceB - cbB + 1;
    if (M < 2 && N < 2) return NULL;

    A = cols_new(M, K); B = cols_new(N, L);
    for (r = 0, c = topA; r < K; ++r, c = c->next)
        for (k = 1; k <= M; ++k) A[k][r] = (uchar)c->text[cbA + k - 1];
    for (r = 0, c = topB; r < L; ++r, c = c->next)
        for (k = 1; k <= N; ++k) B[k][r] = (uchar)c->text[cbB + k - 1];
    nA = base_columns(topA->text, cbA, ceA, &colA);
    nB = base_columns(topB->text, cbB, ceB, &colB);

    LB = (int *)xmalloc(((size_t)M + 1) * sizeof(int));
    RB = (int *)xmalloc(((size_t)M + 1) * sizeof(int));
    for (k = 0; k <= M; ++k) { LB[k] = 0; RB[k] = N; }
    for (k = cb1; k <= ce1; ++k) {
        const char x = X->text[k], y = Y->text[k];
        if (x != '-') {
            if (++tx > nA) mz_fatalf("pre_yama: bad scene");
            if (toupper((unsigned char)x) != toupper(A[colA[tx]][0])) report_disagreement(x, (char)A[colA[tx]][0], beg1, X, cb1, ce1, topA, cbA, ceA, 2);
        }
        if (y != '-') {
            if (++ty > nB) mz_fatalf("pre_yama: ouch");
            if (toupper((unsigned char)y) != toupper(B[colB[ty]][0])) report_disagreement(y, (char)B[colB[ty]][0], begN, Y, cb1, ce1, topB, cbB, ceB, 3);
        }
        if (LB[colA[tx]] == 0) LB[colA[tx]] = colB[ty];
        RB[colA[tx]] = colB[ty];
    }
    if (colA[tx] != M || colB[ty] != N) mz_fatalf("pre_yama: i = %d, M = %d, j = %d, N = %d", colA[tx], M, colB[ty], N);
    free(colA); free(colB);

    smooth(LB, RB, M, N, radius);
    job.K = K; job.L = L; job.M = M; job.N = N; job.A = A[1]; job.B = B[1]; job.LB = LB; job.RB = RB;
    mz_py_run_one(&job, &flat, &om);
    {
        uchar **merged = cols_wrap(flat, om, K + L);
        blk = mafBuild(merged, K + L, om, a2, cbA, a3, cbB, 1);
        cols_free(merged);
    }
    cols_free(A); cols_free(B);
    free(LB); free(RB);
    return blk;
}
