/* mz_multic.c -- the multic driver with all of its merges run as GPU batches (second caller of pre_yama();
 * SURVEY.md 8 f4, reference multic.c). */
#include "mz_drivers.h"

/* ================================================================================================== multic
 * The second caller of pre_yama() (reference multic.c:72): two block lists topped by the same reference, no
 * single-coverage requirement, every overlapping pair of blocks without a common species is merged over its
 * overlap; what no merge covered is printed afterwards.  As in multiz, which pairs are merged and over which
 * slices depends on the inputs' coordinates, species names and paralog marks only -- multih() never looks at an
 * alignment result -- so all merges of a run are enumerated first (multic.c:124-196), run as GPU batches with
 * the host stages one merge per thread, and replayed in order: merged blocks to stdout, the used columns marked
 * from the merged block's reference row (multic.c:100-122), then the unused stretches of every block
 * (:228-254).  A condition the stock program dies of in the middle of the enumeration (multic.c:139,161,176) is
 * raised at the same point of the output. */

typedef struct { struct mafAli *ali; char *unused; char *text; size_t len; } cnode;
typedef struct {
    mz_py py;
    cnode *a, *b;
    int beg, end;
    int state;
    int have;                 /* pre_yama() returned a block */
    int rbeg, rend;           /* reference interval of that block */
    char *text; size_t len;   /* as mafWrite() prints it, if wide enough */
    mz_job bad_job; mz_out bad_out;
} cmerge;
typedef struct { cnode *A, *B; int na, nb, m0, m1; char *fatal; } ccontig;
typedef struct { cmerge *mg; int nmg, capmg; ccontig *ct; int nct, capct; } crecord;

static int align_cate;        /* s=? of the command line (multic.c:37,297) */

static cnode *cnode_list(struct mafAli *list, int *count)      /* create_aliNode_list(), multic.c:198-225 */
{
    struct mafAli *a;
    cnode *v;
    int n = 0, i;
    for (a = list; a; a = a->next) ++n;
    *count = n;
    if (!n) return NULL;
    v = (cnode *)mz_xmalloc((size_t)n * sizeof *v);
    for (i = 0; i < n; ++i) {
        v[i].ali = mz_pop_first(&list);
        v[i].text = NULL; v[i].len = 0;
        v[i].unused = (char *)mz_xmalloc((size_t)v[i].ali->textSize + 1);
        memset(v[i].unused, 'u', (size_t)v[i].ali->textSize);
    }
    return v;
}

static int same_species(struct mafComp *A, struct mafComp *B)
{
    struct mafComp *x, *y;
    for (x = A; x; x = x->next)
        for (y = B; y; y = y->next)
            if (strcmp(x->name, y->name) == 0) return 1;
    return 0;
}

static char *fatal_text(const char *fmt, int arg)
{
    char buf[256];
    snprintf(buf, sizeof buf, fmt, arg);
    return mz_xstrdup(buf);
}

/* multih(), multic.c:124-196: record one merge per call of overlap_wrapper() */
static void enumerate_multic(crecord *R, ccontig *C, int v)
{
    int ia, ib, bk = 0;
    for (ia = 0; ia < C->na; ++ia) {
        struct mafAli *a = C->A[ia].ali;
        struct mafComp *c;
        int a_beg, a_end, copyA = 0;
        if (align_cate == 2 && a->components->paralog == 'a') continue;
        for (c = a->components->next; c; c = c->next) copyA += c->paralog == 'c';
        if (align_cate != 0 && copyA > 1) { C->fatal = fatal_text("A: each block shall contain at most one copy paralog: %d", copyA); return; }
        a_beg = a->components->start;
        a_end = a_beg + a->components->size - 1;
        for (; bk < C->nb; ++bk) {
            c = C->B[bk].ali->components;
            if (c->start + c->size - 1 >= a_beg) break;
        }
        if (bk >= C->nb) return;
        for (ib = bk; ib < C->nb; ++ib) {
            struct mafAli *b = C->B[ib].ali;
            struct mafComp *compA, *compB;
            int b_end, copyB = 0, over_beg, over_end;
            cmerge *g;
            if (align_cate == 2 && b->components->paralog == 'a') continue;
            for (c = b->components->next; c; c = c->next) copyB += c->paralog == 'c';
            if (align_cate != 0 && copyB > 1) { C->fatal = fatal_text("B: each block shall contain at most one copy paralog: %d", copyB); return; }
            if (align_cate != 0 && copyA > 0 && copyB > 0) continue;
            if (b->components->start > a_end) break;
            compA = a->components;
            compB = b->components->next;
            if (v == 0) compA = compA->next;
            if (same_species(compA, compB)) {
                /* (with s != 0 and no copy rows the stock program wants a COLOR_ROW_NAME, which no option sets) */
                if (align_cate != 0 && copyA == 0 && copyB == 0) { C->fatal = mz_xstrdup("No COLOR_ROW_NAME specified!"); return; }
                continue;
            }
            b_end = b->components->start + b->components->size - 1;
            if (a->components->start > b_end || b->components->start > a_end) continue;
            over_beg = a->components->start > b->components->start ? a->components->start : b->components->start;
            over_end = a_end < b_end ? a_end : b_end;
            if (R->nmg == R->capmg) { R->capmg = R->capmg ? 2 * R->capmg : 256; R->mg = (cmerge *)realloc(R->mg, (size_t)R->capmg * sizeof(cmerge)); if (!R->mg) mz_fatalf("out of memory"); }
            g = &R->mg[R->nmg++];
            memset(g, 0, sizeof *g);
            g->a = &C->A[ia]; g->b = &C->B[ib];
            g->beg = over_beg; g->end = over_end;
        }
    }
}

/* the colour of a merged block's top row (multic.c:78-98) */
static void colour_top_row(struct mafAli *n, struct mafAli *A, struct mafAli *B)
{
    const char pa = A->components->paralog, pb = B->components->paralog;
    struct mafComp *c;
    if (pa == pb) n->components->paralog = pa;
    else if ((pa == 'a' && pb == 'c') || (pa == 'c' && pb == 'a')) {
        for (c = (pa == 'a' ? A : B)->components->next; c; c = c->next)
            if (c->paralog == 'c') break;
        n->components->paralog = c ? 'a' : 'c';
    }
}

static void run_multic(crecord *R, int radius, int v, int minw)
{
    const int nmg = R->nmg;
    mz_job *jobs = (mz_job *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(mz_job));
    mz_out *outs = (mz_out *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(mz_out));
    int *who = (int *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(int));
    void **held = NULL;
    int i, nheld = 0;
    mz_score_profile_sync();
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (nmg > 64)
    for (i = 0; i < nmg; ++i) {
        cmerge *g = &R->mg[i];
        g->state = mz_py_begin(&g->py, g->a->ali, g->b->ali, g->beg, g->end, radius, v, NULL);
    }
    for (;;) {
        int n = 0, rc;
        for (i = 0; i < nmg; ++i)
            if (R->mg[i].state == MZ_PY_JOB) { jobs[n] = R->mg[i].py.job; who[n++] = i; }
        if (n == 0) break;
        rc = mz_yama_batch(n, jobs, outs);
        if (rc < 0) mz_fatalf("yama(gfx950): %s", mz_last_error());
        for (i = 0; i < n; ++i) if (outs[i].block) held = mz_hold(held, &nheld, outs[i].block);   /* (kept until the last wave: see mz_multiz.c) */
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (n > 64)
        for (i = 0; i < n; ++i) {
            cmerge *g = &R->mg[who[i]];
            struct mafAli *res = NULL;
            if (outs[i].status != MZ_OK) { g->state = MERGE_FAILED; g->bad_job = jobs[i]; g->bad_out = outs[i]; continue; }
            g->py.borrowed = 1;
            g->state = mz_py_step(&g->py, outs[i].cols, outs[i].OM, &res);
            if (g->state == MZ_PY_JOB || !res) continue;
            colour_top_row(res, g->a->ali, g->b->ali);
            g->have = 1;
            g->rbeg = res->components->start;
            g->rend = res->components->start + res->components->size - 1;
            if (res->textSize >= minw) {
                FILE *m = open_memstream(&g->text, &g->len);
                mafWrite(m, res);
                fclose(m);
            }
            mafAliFree(&res);
        }
    }
    for (i = 0; i < nheld; ++i) free(held[i]);
    free(held);
    free(jobs); free(outs); free(who);
}

static void mark_used(cnode *x, int beg, int end)              /* multic.c:104-122 */
{
    struct mafComp *c = x->ali->components;
    int cb, ce, i;
    if (beg < c->start || beg > c->start + c->size - 1 || end < c->start || end > c->start + c->size - 1)
        mz_fatalf("index out of boundary: %d-%d, %d-%d", beg, end, c->start, c->start + c->size - 1);
    cb = mafPos2Col(c, beg, x->ali->textSize);
    ce = mafPos2Col(c, end, x->ali->textSize);
    for (i = cb; i <= ce; ++i) x->unused[i] = 'o';
}

static void print_unused_multic(cnode *x, FILE *f)             /* multic.c:228-254 */
{
    const int size = x->ali->textSize;
    int i, j;
    for (i = 0; i < size; i = j + 1) {
        struct mafAli *part;
        while (i < size && x->unused[i] == 'o') ++i;
        if (i >= size) break;
        for (j = i; j < size && x->unused[j] == 'u'; ++j)
            ;
        --j;
        part = make_part_ali(x->ali, i, j);
        if (part) { mafWrite(f, part); mafAliFree(&part); }
    }
}

/* The multic run itself on two parsed lists (reference multic.c main loop + multih()): merged blocks to `out`,
 * unused stretches to out1 / out2 (NULL: dropped); blocks of contigs present in one list only stay in *list1 /
 * *list2.  `cate` is the s= option.  The command line below and the in-process tree driver (mz_roast.c) call it. */
int mz_multic_lists(struct mafAli **list1, struct mafAli **list2, int v, int radius, int minw, int cate,
                    FILE *out, FILE *out1, FILE *out2)
{
    FILE *fpw[2] = { out1, out2 };
    crecord R;
    int i, k, x, stop = 0;
    double tm[4];
    align_cate = cate;
    tm[1] = mz_now_s();
    memset(&R, 0, sizeof R);
    while (*list1 && *list2) {                                      /* one reference contig at a time, in file-1 order */
        struct mafAli *wk1 = NULL, *wk2 = NULL;
        char *chr = mz_xstrdup((*list1)->components->src);
        ccontig *C;
        mz_take_chr(list1, &wk1, chr);
        mz_take_chr(list2, &wk2, chr);
        free(chr);
        if (R.nct == R.capct) { R.capct = R.capct ? 2 * R.capct : 16; R.ct = (ccontig *)realloc(R.ct, (size_t)R.capct * sizeof(ccontig)); if (!R.ct) mz_fatalf("out of memory"); }
        C = &R.ct[R.nct++];
        memset(C, 0, sizeof *C);
        C->A = cnode_list(wk1, &C->na);
        C->B = cnode_list(wk2, &C->nb);
        C->m0 = R.nmg;
        if (!stop && C->A && C->B) enumerate_multic(&R, C, v);
        C->m1 = R.nmg;
        if (C->fatal) stop = 1;                             /* the stock program ends there: nothing later is aligned */
    }
    /* (the merges point into the cnode arrays, which do not move; R.mg may have: pointers into it are taken below) */
    tm[2] = mz_now_s();
    run_multic(&R, radius, v, minw);
    tm[3] = mz_now_s();

    for (k = 0; k < R.nct; ++k) {
        ccontig *C = &R.ct[k];
        for (i = C->m0; i < C->m1; ++i) {
            cmerge *g = &R.mg[i];
            if (g->state == MERGE_FAILED) {
                fflush(out); if (fpw[0]) fflush(fpw[0]); if (fpw[1]) fflush(fpw[1]);
                mz_fatal_status(&g->bad_job, &g->bad_out);
            }
            if (!g->have) continue;
            if (g->text) { if (g->len) fwrite(g->text, 1, g->len, out); free(g->text); }
            mark_used(g->a, g->rbeg, g->rend);
            mark_used(g->b, g->rbeg, g->rend);
        }
        if (C->fatal) mz_fatalf("%s", C->fatal);
        for (x = 0; x < 2; ++x) {                           /* the unused stretches: rendered one block per thread ... */
            cnode *list = x ? C->B : C->A;
            const int n = x ? C->nb : C->na;
            if (!fpw[x]) continue;
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (n > 64)
            for (i = 0; i < n; ++i)
                if (list[i].ali->textSize >= minw) {
                    FILE *m = open_memstream(&list[i].text, &list[i].len);
                    print_unused_multic(&list[i], m);
                    fclose(m);
                }
        }
        for (x = 0; x < 2; ++x) {                           /* ... and written in order */
            cnode *list = x ? C->B : C->A;
            const int n = x ? C->nb : C->na;
            for (i = 0; i < n; ++i) {
                if (list[i].text) { if (list[i].len) fwrite(list[i].text, 1, list[i].len, fpw[x]); free(list[i].text); }
                mafAliFree(&list[i].ali);
                free(list[i].unused);
            }
            free(list);
        }
    }
    if (mzi_timing())
        fprintf(stderr, "mz_multic: enumerate %.3f s (%d merges), stages + yama batches %.3f s, replay + unused parts %.3f s\n",
                tm[2] - tm[1], R.nmg, tm[3] - tm[2], mz_now_s() - tm[3]);
    free(R.mg); free(R.ct);

    return 0;
}

int mz_multic_main(int argc, char **argv)
{
    static char cmd[64];
    char *args;
    struct mafAli *l1, *l2, *a;
    FILE *fpw[2];
    int radius = 30, minw = 1, nohead = 0, v, i, x;
    double tm[2];
    size_t na = 64;
    const char *usage =
        "args: [R=?] [M=?] [C=?] file1 file2 v? [out1 out2] [nohead] [all]\n"
        "\tR(30) radius in dynamic programming.\n"
        "\tM(1) minimum output width.\n"
        "\tout1 out2(null) null: stdout; out1 out2: file names for collecting unused input.\n"
        "\tnohead(null) null: output maf header; nohead: not to output maf header.\n"
        "\tall(null) null: not to output single-row blocks; all: output all blocks.\n";

    snprintf(cmd, sizeof cmd, "multic.v%.1f", 12.1);
    argv0 = cmd;
    if (argc < 2) mz_fatalf("%s\n", usage);
    for (i = 1; i < argc; ++i) na += strlen(argv[i]) + 1;
    args = (char *)mz_xmalloc(na);
    strcpy(args, cmd); strcat(args, " ");
    for (i = 1; i < argc; ++i) { strcat(args, argv[i]); strcat(args, " "); }

    while (argc > 1 && argv[1][0] && strchr("sRMC", (x = argv[1][0])) && argv[1][1] == '=') {
        const int val = atoi(argv[1] + 2);
        if (x == 's') align_cate = val;
        else if (x == 'R') { radius = val; if (radius < 0) mz_fatalf("radius cannot be negative"); }
        else if (x == 'M') { minw = val; if (minw < 0) mz_fatalf("MIN_OUTPUT_WID cannot be negative"); }
        else if (val < 0 || val > 100) mz_fatalf("%s\n", usage);          /* C=: connection threshold, unused on this path */
        ++argv; --argc;
    }
    if (strcmp(argv[argc - 1], "all") == 0) { row2 = 0; --argc; }
    if (strcmp(argv[argc - 1], "nohead") == 0) { nohead = 1; --argc; }
    if (argc != 4 && argc != 6)
        mz_fatalf(" -- aligning two files of alignment blocks where top rows are always the reference, reference in both files can contain duplicats\n%s", usage);
    if (argc == 6) { fpw[0] = fopen(argv[4], "w"); fpw[1] = fopen(argv[5], "w"); }
    else fpw[0] = fpw[1] = stdout;
    v = atoi(argv[3]);
    if (v != 0 && v != 1) mz_fatalf("v can only be value of 0 or 1");

    if (!nohead) { fprintf(stdout, "##maf version=1 scoring=%s\n", "multih.c"); printf("# %s\n", args); }
    mz_tune_malloc();
    tm[0] = mz_now_s();
    init_scores70();
    mz_warm_start();                                     /* the GPU starts up while the inputs are read */
    l1 = mz_maf_read_all(argv[1], 1);
    l2 = mz_maf_read_all(argv[2], 1);
    tm[1] = mz_now_s();
    if (mzi_timing()) fprintf(stderr, "mz_multic: read %.3f s\n", tm[1] - tm[0]);

    mz_multic_lists(&l1, &l2, v, radius, minw, align_cate, stdout, fpw[0], fpw[1]);
    for (a = l1; a; a = a->next)                            /* contigs that only one file has */
        if (fpw[0] && (row2 == 0 || a->components->next != NULL)) mafWrite(fpw[0], a);
    for (a = l2; a; a = a->next)
        if (fpw[1] && (row2 == 0 || a->components->next != NULL)) mafWrite(fpw[1], a);
    while (l1) { a = mz_pop_first(&l1); mafAliFree(&a); }
    while (l2) { a = mz_pop_first(&l2); mafAliFree(&a); }
    /* as in multiz: without [out1 out2] both sinks ARE stdout and the stock program closes them before it writes
     * the trailer (multic.c:395-399), so "##eof maf" never appears */
    if (fpw[0] != stdout) { if (fpw[0]) fclose(fpw[0]); if (fpw[1]) fclose(fpw[1]); fprintf(stdout, "##eof maf\n"); }
    free(args);
    return 0;
}

