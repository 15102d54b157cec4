/* mz_multiz.c -- the multiz driver with ALL of its pairwise merges run as GPU batches.
 *
 * The stock driver (reference multiz.c:60-177) walks two position-sorted lists of blocks that share
 * their top (reference) row and, for every overlap, calls pre_yama() -- one small dynamic program at a
 * time.  Which overlaps exist, and which slices of which blocks they cover, depends only on the
 * reference-row coordinates of the inputs, never on an alignment result.  So this restatement walks the
 * lists once, recording in order (a) the text it would have written to out1 / out2 (unused parts of
 * blocks) and (b) one pending pre_yama() per overlap (stage 1 of mz_preyama.c: slicing, packing, band);
 * then it runs the yama() calls of all pending merges as one mz_yama_batch() (two waves when v == 0: the
 * second alignment of a pair needs the first's result), and finally replays the record, writing each
 * merged block where the stock driver would have written it (a job yama() refuses -- e.g. a band narrower
 * than 10 columns -- ends the program with the reference's message at ITS place in that order).  When out1/out2 are not given all three
 * sinks are stdout and the interleaving is preserved exactly.
 *
 * Also here, restated from their behaviour: the MAF reader (reference maf.c:10-36,89-225), keep_ali
 * (multi_util.c:468-509), retrieve_first / seperate_cp_wk (multi_util.c:805-843) and the command line of
 * multiz (multiz.c:180-294).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include "../../include/maf.h"
#include "../../include/mz_scores.h"
#include "../../include/mz_multiz.h"
#include "mz_py.h"
#include <time.h>
#include <malloc.h>

/* MZ_TIMING=1: phase times of a run on stderr */
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

__attribute__((noreturn)) void mz_fatalf(const char *fmt, ...);
__attribute__((noreturn)) void mz_fatal_status(const mz_job *j, const mz_out *o);
extern int row2;
extern char *argv0;

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)n);
    return p;
}
static char *xstrdup(const char *s) { char *p = (char *)xmalloc(strlen(s) + 1); return strcpy(p, s); }

/* ------------------------------------------------------------------------------------------------ MAF reader */

typedef struct { FILE *fp; const char *name; int line_nbr, verbose; char *line; size_t cap; } maf_in;

/* one line, newline kept; -1 at end of file */
static long in_line(maf_in *in)
{
    const ssize_t n = getline(&in->line, &in->cap, in->fp);
    if (n < 0) {
        if (!in->line) { in->line = (char *)xmalloc(16); in->cap = 16; }
        in->line[0] = 0;
        return -1;
    }
    return (long)n;
}

/* next line that is not a comment; comment lines are echoed to stdout when verbose, except the
 * end-of-file marker (reference maf.c:72-87) */
static long in_maf_line(maf_in *in)
{
    long n;
    while ((n = in_line(in)) > 1) {
        in->line_nbr++;
        if (in->line[0] != '#') break;
        if (in->verbose && strstr(in->line, "eof") == NULL) fputs(in->line, stdout);
    }
    return n;
}

/* species and contig parts of "species.contig" (reference multi_util.c:909-925) */
static void split_src(struct mafComp *c)
{
    const char *dot = strchr(c->src, '.');
    size_t n = dot ? (size_t)(dot - c->src) : strlen(c->src);
    c->name = (char *)xmalloc(n + 1);
    memcpy(c->name, c->src, n); c->name[n] = 0;
    c->contig = xstrdup((dot && dot[1]) ? dot + 1 : c->src);
}

/* "a score=... amplifier=<row> copy=<row>" (reference maf.c:89-130) */
static void parse_a_line(const char *line, struct mafAli *a)
{
    const char *p = line + 1;
    struct mafComp *c = a->components;
    int at = 0;
    a->score = (double)MIN_INT;
    while (*p) {
        const char *q;
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == '\n' || *p == 0) break;
        for (q = p; *q && *q != ' ' && *q != '\t' && *q != '\n'; ++q)
            ;
        if (!strncmp(p, "score=", 6)) a->score = atof(p + 6);
        else if (!strncmp(p, "amplifier=", 10) || !strncmp(p, "copy=", 5)) {
            const int amp = p[0] == 'a', row = atoi(p + (amp ? 10 : 5));
            for (; at < row; ++at) c = c->next;
            c->paralog = amp ? 'a' : 'c';
        }
        if (!*q) break;
        p = q + 1;
    }
}

static struct mafAli *maf_next(maf_in *in)
{
    struct mafAli *a;
    struct mafComp *c, *last = NULL;
    char *head;
    long len;
    int i, n;

    while ((len = in_maf_line(in)) != -1)
        if (in->line[0] != '#' && in->line[0] != '\n' && in->line[0] != ' ') break;
    if (len == -1) return NULL;
    if (in->line[0] != 'a')
        mz_fatalf("Expecting 'a (score=xxx)' in file %s, line %d:\n%s", in->name, in->line_nbr, in->line);
    head = xstrdup(in->line);
    a = (struct mafAli *)xmalloc(sizeof *a);
    memset(a, 0, sizeof *a);
    while ((len = in_maf_line(in)) != -1 && in->line[0] != '\n' && in->line[0] != ' ' && in->line[0] != '#') {
        char *src, *text;
        if (in->line[0] != 's') continue;                  /* i / e / q lines are ignored */
        c = (struct mafComp *)xmalloc(sizeof *c);
        memset(c, 0, sizeof *c);
        src = (char *)xmalloc((size_t)len + 1);
        text = (char *)xmalloc((size_t)len + 1);
        if (sscanf(in->line, "s %s %d %d %c %d %s", src, &c->start, &c->size, &c->strand, &c->srcSize, text) != 6)
            mz_fatalf("bad component in file %s, line %d:\n%s", in->name, in->line_nbr, src);
        c->src = xstrdup(src); free(src);
        c->text = text;
        split_src(c);
        c->paralog = 's';
        if (!a->components) { a->textSize = (int)strlen(c->text); a->components = c; }
        else {
            if (a->textSize != (int)strlen(c->text))
                mz_fatalf("line %d of %s: inconsistent row size", in->line_nbr, in->name);
            last->next = c;
        }
        last = c;
        if (c->srcSize <= 0 || c->size <= 0)
            mz_fatalf("Size <= 0 at line %d of file %s:\n%s", in->line_nbr, in->name, in->line);
        if (c->start < 0 || c->start + c->size > c->srcSize) {
            if (c != a->components)
                fprintf(stderr, "in maf entry with top row %s:%d len = %d,\n", a->components->src, a->components->start, a->components->size);
            mz_fatalf("Bad coordinates at line %d of file %s:\n%s", in->line_nbr, in->name, in->line);
        }
        for (i = n = 0; i < a->textSize; ++i) n += c->text[i] != '-';
        if (n != c->size)
            mz_fatalf("Actual size %d, claimed size %d at line %d of file %s:\n%s", n, c->size, in->line_nbr, in->name, in->line);
    }
    if (!a->components) mz_fatalf("block without rows in file %s, line %d", in->name, in->line_nbr);
    parse_a_line(head, a);
    free(head);
    in->line_nbr++;
    return a;
}

struct mafAli *mz_maf_read_all(const char *path, int verbose)
{
    maf_in in;
    struct mafAli *first = NULL, *last = NULL, *a;
    char buf[500];
    int version;

    memset(&in, 0, sizeof in);
    in.name = path; in.verbose = verbose;
    in.fp = fopen(path, "r");
    if (!in.fp) mz_fatalf("Cannot open %s.", path);
    if (!fgets(buf, sizeof buf, in.fp)) mz_fatalf("empty file %s", path);
    if (sscanf(buf, "##maf version=%d", &version) != 1) mz_fatalf("improper maf header line: %s", buf);
    while ((a = maf_next(&in)) != NULL) {
        if (last) last->next = a; else first = a;
        last = a;
    }
    fclose(in.fp);
    free(in.line);
    return first;
}

/* ------------------------------------------------------------------------------------------------ list helpers */

static struct mafAli *pop_first(struct mafAli **head)
{
    struct mafAli *a = *head;
    if (a) { *head = a->next; a->next = NULL; }
    return a;
}

/* move every block whose top row lies on `chr` from *from to the tail of *to, keeping the order */
static void take_chr(struct mafAli **from, struct mafAli **to, const char *chr)
{
    struct mafAli **pp = from, *tail = *to;
    while (tail && tail->next) tail = tail->next;
    while (*pp) {
        struct mafAli *a = *pp;
        if (strcmp(chr, a->components->src) == 0) {
            *pp = a->next;
            a->next = NULL;
            if (tail) tail->next = a; else *to = a;
            tail = a;
        } else pp = &a->next;
    }
}

/* The walk goes on cutting the blocks it holds, so every pending merge keeps private copies of its two blocks
 * until the replay.  They are never resized or freed one by one: they come from a bump arena that is released as
 * a whole (thirty malloc/free pairs per merge otherwise -- a sixth of a 20 000-block run). */
typedef struct arena_chunk { struct arena_chunk *next; size_t used, cap; } arena_chunk;
static arena_chunk *g_arena;

static void *arena_alloc(size_t n)
{
    void *p;
    n = (n + 15) & ~(size_t)15;
    if (!g_arena || g_arena->used + n > g_arena->cap) {
        const size_t cap = n > ((size_t)8 << 20) ? n : ((size_t)8 << 20);
        arena_chunk *c = (arena_chunk *)xmalloc(sizeof *c + cap);
        c->next = g_arena; c->used = 0; c->cap = cap;
        g_arena = c;
    }
    p = (char *)(g_arena + 1) + g_arena->used;
    g_arena->used += n;
    return p;
}
static char *arena_strdup(const char *s)
{
    size_t n;
    if (!s) return NULL;
    n = strlen(s) + 1;
    return (char *)memcpy(arena_alloc(n), s, n);
}
static void arena_release(void)
{
    while (g_arena) { arena_chunk *c = g_arena; g_arena = c->next; free(c); }
}

static struct mafAli *clone_ali(const struct mafAli *a)      /* arena-owned: never passed to mafAliFree() */
{
    struct mafAli *b = (struct mafAli *)arena_alloc(sizeof *b);
    struct mafComp *c, *tail = NULL;
    *b = *a;
    b->next = NULL; b->components = NULL;
    for (c = a->components; c; c = c->next) {
        struct mafComp *d = (struct mafComp *)arena_alloc(sizeof *d);
        *d = *c;
        d->next = NULL; d->mafPosMap = NULL;
        d->src = arena_strdup(c->src); d->name = arena_strdup(c->name); d->contig = arena_strdup(c->contig);
        d->text = arena_strdup(c->text);
        if (tail) tail->next = d; else b->components = d;
        tail = d;
    }
    return b;
}

/* cut the block down to what starts at reference position beg; rows left with no base go
 * (reference multi_util.c:468-509) */
static struct mafAli *keep_from(struct mafAli *a, int beg)
{
    const int len = (int)strlen(a->components->text);
    struct mafComp **pp, *c;
    int col, i, n;

    col = mafPos2Col(a->components, beg, a->textSize);
    while (col > 0 && a->components->text[col - 1] == '-') --col;
    for (pp = &a->components; (c = *pp) != NULL; ) {
        char *s;
        for (n = i = 0; i < col; ++i) n += c->text[i] != '-';
        if (c->size - n < 1) { *pp = c->next; mafCompFree(&c); continue; }
        c->start += n;
        c->size -= n;
        s = (char *)xmalloc((size_t)(len - col) + 2);
        memcpy(s, c->text + col, (size_t)(len - col));
        s[len - col] = 0;
        free(c->text);
        c->text = s;
        pp = &c->next;
    }
    a->textSize = len - col;
    a->score = mafScoreRange(a, 0, len - col);
    return a;
}

#define MZ_STAGE_THREADS 32     /* host threads of the per-merge stages (allocation-heavy: more does not help) */

/* ------------------------------------------------------------------------------------------------ the record */

enum { SINK_OUT = 0, SINK_1 = 1, SINK_2 = 2 };
typedef struct {
    int sink;                 /* where the text goes */
    char *text; size_t len;   /* recorded output (unused parts, pre_yama's side write) or NULL */
    int job;                  /* index into the merge list, or -1 */
    struct mafAli *src;       /* text still to be rendered (render_events): this block (arena copy) ... */
    int cbeg, cend;           /* ... columns cbeg..cend of it, or the whole block when cbeg < 0 */
} event;
typedef struct {
    mz_py py;
    struct mafAli *a1, *a2;   /* private copies: the walk goes on cutting the originals */
    struct mafAli *result;
    int beg, end, radius, v;  /* pre_yama()'s arguments */
    int side_ev;              /* the event that holds stage 1's side write to out2 */
    char *text; size_t len;   /* the merged block as mafWrite() prints it (rendered by the thread that built it) */
    int state;                /* MZ_PY_JOB while a yama() call is pending; MERGE_FAILED: yama() refused the job */
    mz_job bad_job; mz_out bad_out;
} merge;
#define MERGE_FAILED 99
typedef struct {
    event *ev; int nev, capev;
    merge *mg; int nmg, capmg;
    int has1, has2;           /* out1 / out2 sinks exist */
} record;

static event *new_event(record *R, int sink)
{
    if (R->nev == R->capev) { R->capev = R->capev ? 2 * R->capev : 256; R->ev = (event *)realloc(R->ev, (size_t)R->capev * sizeof(event)); if (!R->ev) mz_fatalf("out of memory"); }
    memset(&R->ev[R->nev], 0, sizeof(event));
    R->ev[R->nev].sink = sink; R->ev[R->nev].job = -1;
    return &R->ev[R->nev++];
}

/* The two writers of the walk.  What they print depends on the block as it is NOW (the walk goes on cutting
 * it), so the event keeps a copy; the text itself -- slicing, dash-column removal, scoring, formatting -- is
 * produced later by render_events(), one event per thread. */
static void rec_block(record *R, int sink, struct mafAli *a)
{
    event *e = new_event(R, sink);
    e->src = clone_ali(a); e->cbeg = -1; e->cend = -1;
}
static void rec_part(record *R, int sink, struct mafAli *a, int cbeg, int cend)
{
    event *e = new_event(R, sink);
    e->src = clone_ali(a); e->cbeg = cbeg; e->cend = cend;
}
static void render_events(record *R)
{
    const int nev = R->nev;
    int i;
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (nev > 64)
    for (i = 0; i < nev; ++i) {
        event *e = &R->ev[i];
        FILE *m;
        if (!e->src) continue;
        m = open_memstream(&e->text, &e->len);
        if (e->cbeg < 0) mafWrite(m, e->src);
        else print_part_ali_col(e->src, e->cbeg, e->cend, m);
        fclose(m);
        e->src = NULL;
    }
}

static void rec_merge(record *R, struct mafAli *a1, struct mafAli *a2, int beg, int end, int radius, int v)
{
    event *e;
    merge *g;
    if (R->nmg == R->capmg) { R->capmg = R->capmg ? 2 * R->capmg : 256; R->mg = (merge *)realloc(R->mg, (size_t)R->capmg * sizeof(merge)); if (!R->mg) mz_fatalf("out of memory"); }
    g = &R->mg[R->nmg];
    memset(g, 0, sizeof *g);
    g->a1 = clone_ali(a1); g->a2 = clone_ali(a2);
    g->beg = beg; g->end = end; g->radius = radius; g->v = v;
    /* stage 1 (run_merges) may write to out2 -- nothing of a1 left to align, mz_preyama.c:193-196 -- and that
     * text belongs at this point of the output */
    e = new_event(R, SINK_2);
    g->side_ev = R->nev - 1;
    e = new_event(R, SINK_OUT);
    e->job = R->nmg++;
}

/* the walk over two lists of one reference contig (control flow of reference multiz.c:60-177) */
static void walk(record *R, struct mafAli **wk1, struct mafAli **wk2, int v, int radius, int minw)
{
    struct mafAli *a1 = pop_first(wk1), *a2 = pop_first(wk2);
#define BEG(a) ((a)->components->start)
#define END(a) ((a)->components->start + (a)->components->size - 1)
#define WANTED(a) ((a)->components->size >= minw && (row2 == 0 || (a)->components->next != NULL))
    for (;;) {
        int beg1, end1, beg2, end2, beg, end, cb, ce;
        while (a1 && (!a2 || END(a1) < BEG(a2))) {         /* nothing of file 2 under a1 */
            if (R->has1 && WANTED(a1)) rec_block(R, SINK_1, a1);
            mafAliFree(&a1);
            a1 = pop_first(wk1);
        }
        while (a2 && (!a1 || END(a2) < BEG(a1))) {
            if (R->has2 && WANTED(a2)) rec_block(R, SINK_2, a2);
            mafAliFree(&a2);
            a2 = pop_first(wk2);
        }
        if (!a1 && !a2) break;
        if (!a1 || !a2 || END(a1) < BEG(a2) || END(a2) < BEG(a1)) continue;

        beg1 = BEG(a1); end1 = END(a1); beg2 = BEG(a2); end2 = END(a2);
        /* the part of the earlier block in front of the overlap is unused */
        if (beg1 < beg2 && beg2 - beg1 >= minw && R->has1) {
            cb = mafPos2Col(a1->components, beg1, a1->textSize);
            while (cb > 0 && a1->components->text[cb - 1] == '-') --cb;
            ce = mafPos2Col(a1->components, beg2 - 1, a1->textSize);
            while (ce < a1->textSize - 1 && a1->components->text[ce + 1] == '-') ++ce;
            rec_part(R, SINK_1, a1, cb, ce);
        } else if (beg2 < beg1 && beg1 - beg2 >= minw && R->has2) {
            cb = mafPos2Col(a2->components, beg2, a2->textSize);
            while (cb > 0 && a2->components->text[cb - 1] == '-') --cb;
            ce = mafPos2Col(a2->components, beg1 - 1, a2->textSize);
            while (ce < a2->textSize - 1 && a2->components->text[ce + 1] == '-') ++ce;
            rec_part(R, SINK_2, a2, cb, ce);
        }
        beg = beg1 > beg2 ? beg1 : beg2;
        end = end1 < end2 ? end1 : end2;
        if (beg == beg1) {                                  /* columns in front of the block's first base */
            cb = mafPos2Col(a1->components, beg1, a1->textSize);
            if (cb != 0 && R->has1) rec_part(R, SINK_1, a1, 0, cb - 1);
        }
        if (beg == beg2) {
            cb = mafPos2Col(a2->components, beg2, a2->textSize);
            if (cb != 0 && R->has2) rec_part(R, SINK_2, a2, 0, cb - 1);
        }
        rec_merge(R, a1, a2, beg, end, radius, v);

        if (end1 < end2) a2 = keep_from(a2, end1 + 1);
        if (end2 < end1) a1 = keep_from(a1, end2 + 1);
        if (end1 <= end2) {
            ce = mafPos2Col(a1->components, end1, a1->textSize);
            if (ce < a1->textSize - 1 && R->has1) rec_part(R, SINK_1, a1, ce + 1, a1->textSize - 1);
            mafAliFree(&a1);
            a1 = pop_first(wk1);
        }
        if (end2 <= end1) {
            ce = mafPos2Col(a2->components, end2, a2->textSize);
            if (ce < a2->textSize - 1 && R->has2) rec_part(R, SINK_2, a2, ce + 1, a2->textSize - 1);
            mafAliFree(&a2);
            a2 = pop_first(wk2);
        }
    }
#undef BEG
#undef END
#undef WANTED
}

/* Stage 1 of every merge, then the pending yama() calls of all of them wave after wave.  The merges are
 * independent of one another, so everything on the host side of the yama() batches -- column packing,
 * rmColDash, the band walk and smooth() before, mafBuild() and mafScoreRange() after -- runs one merge per
 * thread. */
static void run_merges(record *R, int minw)
{
    mz_job *jobs = (mz_job *)xmalloc((size_t)(R->nmg ? R->nmg : 1) * sizeof(mz_job));
    mz_out *outs = (mz_out *)xmalloc((size_t)(R->nmg ? R->nmg : 1) * sizeof(mz_out));
    int *who = (int *)xmalloc((size_t)(R->nmg ? R->nmg : 1) * sizeof(int));
    const int nmg = R->nmg, has2 = R->has2;
    const int timing = getenv("MZ_TIMING") != NULL;
    double t0 = now_s(), t1;
    int i;
    mz_score_profile_sync();
    render_events(R);
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (nmg > 64)
    for (i = 0; i < nmg; ++i) {
        merge *g = &R->mg[i];
        event *e = &R->ev[g->side_ev];
        FILE *m = open_memstream(&e->text, &e->len);
        g->state = mz_py_begin(&g->py, g->a1, g->a2, g->beg, g->end, g->radius, g->v, has2 ? m : NULL);
        fclose(m);
    }
    t1 = now_s();
    if (timing) fprintf(stderr, "mz_multiz: stage 1 of %d merges %.3f s\n", nmg, t1 - t0);
    for (;;) {
        int n = 0, rc;
        for (i = 0; i < nmg; ++i)
            if (R->mg[i].state == MZ_PY_JOB) { jobs[n] = R->mg[i].py.job; who[n++] = i; }
        if (n == 0) break;
        t0 = now_s();
        rc = mz_yama_batch(n, jobs, outs);
        if (rc < 0) mz_fatalf("yama(gfx950): %s", mz_last_error());
        t1 = now_s();
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (n > 64)
        for (i = 0; i < n; ++i) {
            merge *g = &R->mg[who[i]];
            if (outs[i].status != MZ_OK) {                  /* reported at its place in the output order, see replay() */
                g->state = MERGE_FAILED; g->bad_job = jobs[i]; g->bad_out = outs[i];
            } else
                g->state = mz_py_step(&g->py, outs[i].cols, outs[i].OM, &g->result);
            if (g->state != MZ_PY_JOB && g->state != MERGE_FAILED) {    /* finished: render and release here */
                if (g->result && g->result->components->size >= minw) {
                    FILE *m = open_memstream(&g->text, &g->len);
                    mafWrite(m, g->result);
                    fclose(m);
                }
                mafAliFree(&g->result);     /* built by this thread: released into its own arena (cheap); a1 / a2
                                             * came from the main thread and go back there, in replay() */
            }
        }
        if (timing) fprintf(stderr, "mz_multiz: yama batch of %d %.3f s (with GPU start-up in the first), next stage %.3f s\n", n, t1 - t0, now_s() - t1);
    }
    free(jobs); free(outs); free(who);
}

static void replay(record *R, FILE *out, FILE *f1, FILE *f2, int minw)
{
    int i;
    for (i = 0; i < R->nev; ++i) {
        event *e = &R->ev[i];
        FILE *f = e->sink == SINK_OUT ? out : e->sink == SINK_1 ? f1 : f2;
        if (e->job >= 0) {
            merge *g = &R->mg[e->job];
            if (g->state == MERGE_FAILED) {                 /* the stock driver got this far, then yama() ended the run */
                fflush(out); if (f1) fflush(f1); if (f2) fflush(f2);
                mz_fatal_status(&g->bad_job, &g->bad_out);
            }
            if (g->text) { if (g->len) fwrite(g->text, 1, g->len, out); free(g->text); }
            mafAliFree(&g->result);
            g->a1 = g->a2 = NULL;                           /* (arena) */
        } else if (e->text) {
            if (f && e->len) fwrite(e->text, 1, e->len, f);
            free(e->text);
        }
    }
    free(R->ev); free(R->mg);
    memset(R, 0, sizeof *R);
    arena_release();
}

int mz_multiz_lists(struct mafAli **list1, struct mafAli **list2, int v, int radius, int min_output_wid,
                    FILE *out, FILE *out1, FILE *out2)
{
    record R;
    memset(&R, 0, sizeof R);
    R.has1 = out1 != NULL; R.has2 = out2 != NULL;
    while (*list1 && *list2) {                              /* one reference contig at a time, in file-1 order */
        struct mafAli *wk1 = NULL, *wk2 = NULL;
        char *chr = xstrdup((*list1)->components->src);
        take_chr(list1, &wk1, chr);
        take_chr(list2, &wk2, chr);
        free(chr);
        walk(&R, &wk1, &wk2, v, radius, min_output_wid);
    }
    {
        const int timing = getenv("MZ_TIMING") != NULL, nmerge = R.nmg;
        const double t0 = now_s();
        double t1, t2;
        run_merges(&R, min_output_wid);
        t1 = now_s();
        replay(&R, out, out1, out2, min_output_wid);
        t2 = now_s();
        if (timing) fprintf(stderr, "mz_multiz: %d merges; yama batches + stage 2/3 %.3f s, replay %.3f s\n", nmerge, t1 - t0, t2 - t1);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ command line */

int mz_multiz_main(int argc, char **argv)
{
    static char cmd[64];
    char *args;
    struct mafAli *l1, *l2, *a;
    FILE *f1 = NULL, *f2 = NULL;
    int radius = 30, minw = 1, nohead = 0, v, i, x;
    size_t na = 64;
    const char *usage =
        "args: [R=?] [M=?] file1 file2 v? [out1 out2] [nohead] [all]\n"
        "\tR(30) radius in dynamic programming.\n"
        "\tM(1) minimum output width.\n"
        "\tout1 out2(null) null: stdout; out1 out2: file names for collecting unused input.\n"
        "\tnohead(null) null: output maf header; nohead: not to output maf header.\n"
        "\tall(null) null: not to output single-row blocks; all: output all blocks.\n";

    snprintf(cmd, sizeof cmd, "multiz.v%.1f", 11.2);
    argv0 = cmd;
    for (i = 1; i < argc; ++i) na += strlen(argv[i]) + 1;
    args = (char *)xmalloc(na);
    strcpy(args, cmd); strcat(args, " ");
    for (i = 1; i < argc; ++i) { strcat(args, argv[i]); strcat(args, " "); }

    while (argc > 1 && strchr("RMLS", (x = argv[1][0])) && argv[1][1] == '=') {
        const int val = atoi(argv[1] + 2);
        if (x == 'R') { radius = val; if (radius < 0) mz_fatalf("radius cannot be negative"); }
        else if (x == 'M') { minw = val; if (minw < 0) mz_fatalf("MIN_OUTPUT_WID cannot be negative"); }
        else if (x == 'L') { if (val < 0) mz_fatalf("LRG_BREAK_WID cannot be negative"); }
        else { if (val < 0) mz_fatalf("SML_BREAK_WID cannot be negative"); }
        ++argv; --argc;
    }
    if (argc > 1 && strcmp(argv[argc - 1], "all") == 0) { row2 = 0; --argc; }
    if (argc > 1 && strcmp(argv[argc - 1], "nohead") == 0) { nohead = 1; --argc; }
    if (argc != 4 && argc != 6)
        mz_fatalf(" -- aligning two files of alignment blocks where top rows are always the reference, reference in both files cannot have duplicats\n%s", usage);
    if (argc == 6) {
        f1 = fopen(argv[4], "w"); f2 = fopen(argv[5], "w");
        if (!f1 || !f2) mz_fatalf("Cannot open %s.", !f1 ? argv[4] : argv[5]);
    } else f1 = f2 = stdout;
    v = atoi(argv[3]);
    if (v != 0 && v != 1) mz_fatalf("v can only be value of 0, 1 ");

    if (!nohead) { fprintf(stdout, "##maf version=1 scoring=%s\n", "multiz"); printf("# %s\n", args); }
    {
        const double t0 = now_s();
        double t1, t2;
        /* This program makes millions of small allocations from up to 32 threads; letting the heaps grow (and
         * shrink) in small steps cost a fifth of the run in brk/mprotect calls and page-table locks. */
        mallopt(M_TOP_PAD, 256 << 20);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
        init_scores70();
        l1 = mz_maf_read_all(argv[1], 1);
        l2 = mz_maf_read_all(argv[2], 1);
        t1 = now_s();
        mz_multiz_lists(&l1, &l2, v, radius, minw, stdout, f1, f2);
        t2 = now_s();
        if (getenv("MZ_TIMING")) fprintf(stderr, "mz_multiz: read %.3f s, walk + merges + replay %.3f s\n", t1 - t0, t2 - t1);
    }

    for (a = l1; a; a = a->next)                            /* contigs that only one file has */
        if (f1 && (row2 == 0 || a->components->next != NULL)) mafWrite(f1, a);
    for (a = l2; a; a = a->next)
        if (f2 && (row2 == 0 || a->components->next != NULL)) mafWrite(f2, a);
    while (l1) { a = pop_first(&l1); mafAliFree(&a); }
    while (l2) { a = pop_first(&l2); mafAliFree(&a); }
    /* The stock driver closes out1 and out2 before it writes the trailer -- and without [out1 out2] both
     * ARE stdout (multiz.c:243-245,286-291), so its "##eof maf" line never reaches the output.  Same here. */
    if (f1 != stdout) { fclose(f1); fclose(f2); fprintf(stdout, "##eof maf\n"); }
    free(args);
    return 0;
}

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
    v = (cnode *)xmalloc((size_t)n * sizeof *v);
    for (i = 0; i < n; ++i) {
        v[i].ali = pop_first(&list);
        v[i].text = NULL; v[i].len = 0;
        v[i].unused = (char *)xmalloc((size_t)v[i].ali->textSize + 1);
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
    return xstrdup(buf);
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
                if (align_cate != 0 && copyA == 0 && copyB == 0) { C->fatal = xstrdup("No COLOR_ROW_NAME specified!"); return; }
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
    mz_job *jobs = (mz_job *)xmalloc((size_t)(nmg ? nmg : 1) * sizeof(mz_job));
    mz_out *outs = (mz_out *)xmalloc((size_t)(nmg ? nmg : 1) * sizeof(mz_out));
    int *who = (int *)xmalloc((size_t)(nmg ? nmg : 1) * sizeof(int));
    int i;
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
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (n > 64)
        for (i = 0; i < n; ++i) {
            cmerge *g = &R->mg[who[i]];
            struct mafAli *res = NULL;
            if (outs[i].status != MZ_OK) { g->state = MERGE_FAILED; g->bad_job = jobs[i]; g->bad_out = outs[i]; continue; }
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

int mz_multic_main(int argc, char **argv)
{
    static char cmd[64];
    char *args;
    struct mafAli *l1, *l2, *a;
    FILE *fpw[2];
    crecord R;
    int radius = 30, minw = 1, nohead = 0, v, i, k, x, stop = 0;
    double tm[4];
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
    args = (char *)xmalloc(na);
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
    mallopt(M_TOP_PAD, 256 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    tm[0] = now_s();
    init_scores70();
    l1 = mz_maf_read_all(argv[1], 1);
    l2 = mz_maf_read_all(argv[2], 1);
    tm[1] = now_s();

    memset(&R, 0, sizeof R);
    while (l1 && l2) {                                      /* one reference contig at a time, in file-1 order */
        struct mafAli *wk1 = NULL, *wk2 = NULL;
        char *chr = xstrdup(l1->components->src);
        ccontig *C;
        take_chr(&l1, &wk1, chr);
        take_chr(&l2, &wk2, chr);
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
    tm[2] = now_s();
    run_multic(&R, radius, v, minw);
    tm[3] = now_s();

    for (k = 0; k < R.nct; ++k) {
        ccontig *C = &R.ct[k];
        for (i = C->m0; i < C->m1; ++i) {
            cmerge *g = &R.mg[i];
            if (g->state == MERGE_FAILED) {
                fflush(stdout); if (fpw[0]) fflush(fpw[0]); if (fpw[1]) fflush(fpw[1]);
                mz_fatal_status(&g->bad_job, &g->bad_out);
            }
            if (!g->have) continue;
            if (g->text) { if (g->len) fwrite(g->text, 1, g->len, stdout); free(g->text); }
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
    if (getenv("MZ_TIMING"))
        fprintf(stderr, "mz_multic: read %.3f s, enumerate %.3f s (%d merges), stages + yama batches %.3f s, replay + unused parts %.3f s\n",
                tm[1] - tm[0], tm[2] - tm[1], R.nmg, tm[3] - tm[2], now_s() - tm[3]);
    free(R.mg); free(R.ct);

    for (a = l1; a; a = a->next)                            /* contigs that only one file has */
        if (fpw[0] && (row2 == 0 || a->components->next != NULL)) mafWrite(fpw[0], a);
    for (a = l2; a; a = a->next)
        if (fpw[1] && (row2 == 0 || a->components->next != NULL)) mafWrite(fpw[1], a);
    while (l1) { a = pop_first(&l1); mafAliFree(&a); }
    while (l2) { a = pop_first(&l2); mafAliFree(&a); }
    /* as in multiz: without [out1 out2] both sinks ARE stdout and the stock program closes them before it writes
     * the trailer (multic.c:395-399), so "##eof maf" never appears */
    if (fpw[0] != stdout) { if (fpw[0]) fclose(fpw[0]); if (fpw[1]) fclose(fpw[1]); fprintf(stdout, "##eof maf\n"); }
    free(args);
    return 0;
}
