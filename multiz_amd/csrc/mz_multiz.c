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
 * Also here, restated from their behaviour: keep_ali (multi_util.c:468-509) and the command line of multiz
 * (multiz.c:180-294).  The MAF reader and the list helpers are in mz_mafio.c, the multic driver in mz_multic.c.
 */
#include "mz_drivers.h"
#include <pthread.h>

/* The walk goes on cutting the blocks it holds, so every pending merge and every recorded piece of output keeps the two blocks AS
 * THEY STAND at that moment -- the rows they have, each row's start, size and text.  Nothing is copied for that but the bookkeeping:
 * cutting a block (keep_from) only moves its rows' text pointers forward, a row that runs out of bases is unlinked, not freed, and a
 * block the walk is through with is retired into the record (record.dead / dead_rows) instead of freed -- so the snapshots, which
 * come from a bump arena and share names and text with the originals, stay good until the merges are through (run_merges() hands
 * the retired blocks of all its records to the thread that frees them, free_retired(); a record that never gets there does when it
 * goes).  (A level of a
 * guide tree is half a million merges of up to thirty rows: copying every row's names and text per snapshot, allocating every cut
 * row anew and freeing both inside the walk was most of the walk.) */
typedef struct arena_chunk { struct arena_chunk *next; size_t used, cap; } arena_chunk;
/* one arena per record (a run's list walk): the tree driver walks the lists of sibling nodes on several threads */
static void *arena_alloc(arena_chunk **arena, size_t n)
{
    char *p;
    n = (n + 15) & ~(size_t)15;
    if (!*arena || (*arena)->used + n > (*arena)->cap) {
        const size_t cap = n > ((size_t)1 << 20) ? n : ((size_t)1 << 20);
        arena_chunk *c = (arena_chunk *)mz_xmalloc(sizeof *c + cap);
        c->next = *arena; c->used = 0; c->cap = cap;
        *arena = c;
    }
    p = (char *)(*arena + 1) + (*arena)->used;
    (*arena)->used += n;
    return p;
}
static void arena_release(arena_chunk **arena)
{
    while (*arena) { arena_chunk *c = *arena; *arena = c->next; free(c); }
}

static struct mafAli *clone_ali(arena_chunk **arena, const struct mafAli *a)      /* arena-owned: never passed to mafAliFree(); names and text are the original's */
{
    struct mafAli *b = (struct mafAli *)arena_alloc(arena, sizeof *b);
    struct mafComp *c, *tail = NULL;
    *b = *a;
    b->next = NULL; b->components = NULL;
    for (c = a->components; c; c = c->next) {
        struct mafComp *d = (struct mafComp *)arena_alloc(arena, sizeof *d);
        *d = *c;
        d->next = NULL; d->mafPosMap = NULL;
        if (tail) tail->next = d; else b->components = d;
        tail = d;
    }
    return b;
}

/* ------------------------------------------------------------------------------------------------ the record */

enum { SINK_OUT = 0, SINK_1 = 1, SINK_2 = 2 };
typedef struct {
    int sink;                 /* where the text goes */
    char *text; size_t len;   /* recorded output (unused parts, pre_yama's side write) or NULL */
    int job;                  /* index into the merge list, or -1 */
    struct mafAli *src;       /* text still to be rendered (render_events): this block (arena copy) ... */
    int cbeg, cend;           /* ... columns cbeg..cend of it, or the whole block when cbeg < 0 */
    struct mafAli *blk;       /* record.keep_blocks: the block itself instead of its text (heap; NULL where nothing would be printed) */
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
typedef struct {
    event *ev; int nev, capev;
    merge *mg; int nmg, capmg;
    int has1, has2;           /* out1 / out2 sinks exist */
    int keep_blocks;          /* the run ends in mz_multiz_finish_lists(): blocks are kept as blocks, nothing is rendered */
    arena_chunk *arena;       /* the snapshots of this run's events and merges */
    mz_blocks dead;           /* blocks the walk is through with (their rows' text pointers back at the start of their allocations) */
    struct mafComp *dead_rows;/* rows cut off such blocks */
} record;

/* cut the block down to what starts at reference position beg; rows left with no base go
 * (reference multi_util.c:468-509).  *off: the columns cut off this block so far (every row's text pointer stands that far into
 * its allocation) */
static struct mafAli *keep_from(record *R, struct mafAli *a, int beg, int *off)
{
    const int len = (int)strlen(a->components->text);
    struct mafComp **pp, *c;
    int col, i, n;

    col = mafPos2Col(a->components, beg, a->textSize);
    while (col > 0 && a->components->text[col - 1] == '-') --col;
    for (pp = &a->components; (c = *pp) != NULL; ) {
        for (n = i = 0; i < col; ++i) n += c->text[i] != '-';
        if (c->size - n < 1) { *pp = c->next; c->text -= *off; c->next = R->dead_rows; R->dead_rows = c; continue; }
        c->start += n;
        c->size -= n;
        c->text += col;
        pp = &c->next;
    }
    *off += col;
    a->textSize = len - col;
    a->score = mafScoreRange(a, 0, len - col);
    return a;
}
static void retire(record *R, struct mafAli *a, int off)
{
    struct mafComp *c;
    if (off) for (c = a->components; c; c = c->next) c->text -= off;
    mz_blocks_push(&R->dead, a);
}
/* The retired blocks go in the BACKGROUND: one thread frees them, a record's lot after the other, while the run goes on (the next
 * round's parsing, the replay, the destination's rendering).  Freed on all threads at once they cost more than they did inside the walk
 * -- neighbours in a list come from the same thread's heap, and threads that free into one heap at the same time queue for its lock
 * (8 s of kernel time in a 2.5 s run); freed by the caller they are a second of one thread that everybody waits for.  Nobody waits for
 * the reaper to get through: when the program ends (atexit) it is told to stop after the block in hand and the exit goes on once it
 * has -- whatever is left goes with the process.  MZ_REAPER=0: freed on the spot, by the caller. */
typedef struct reap { struct reap *next; struct mafAli **blocks; int n; struct mafComp *rows; } reap;
static pthread_mutex_t g_reap_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_reap_cv = PTHREAD_COND_INITIALIZER;
static pthread_cond_t g_reap_parked_cv = PTHREAD_COND_INITIALIZER;
static reap *g_reap_head, *g_reap_tail;
static int g_reap_started, g_reap_quit, g_reap_parked;

static void reap_now(struct mafAli **blocks, int n, struct mafComp *rows)
{
    int i;
    for (i = 0; i < n; ++i) mafAliFree(&blocks[i]);
    free(blocks);
    while (rows) { struct mafComp *c = rows; rows = c->next; mafCompFree(&c); }
}
static void reaper_park(void)                               /* (the mutex held) for good: the program is ending */
{
    g_reap_parked = 1;
    pthread_cond_broadcast(&g_reap_parked_cv);
    for (;;) pthread_cond_wait(&g_reap_cv, &g_reap_mu);
}
static void *reaper(void *arg)
{
    (void)arg;
    for (;;) {
        reap *j;
        int i;
        pthread_mutex_lock(&g_reap_mu);
        while (!g_reap_head && !g_reap_quit) pthread_cond_wait(&g_reap_cv, &g_reap_mu);
        if (g_reap_quit) reaper_park();
        j = g_reap_head; g_reap_head = j->next;
        if (!g_reap_head) g_reap_tail = NULL;
        pthread_mutex_unlock(&g_reap_mu);
        for (i = 0; i < j->n; ++i) {
            if (__atomic_load_n(&g_reap_quit, __ATOMIC_RELAXED)) { pthread_mutex_lock(&g_reap_mu); reaper_park(); }
            mafAliFree(&j->blocks[i]);
        }
        reap_now(NULL, 0, j->rows);
        free(j->blocks);
        free(j);
    }
    return NULL;
}
static void reaper_stop(void)                               /* atexit: nothing of this library runs beside the exit handlers that follow */
{
    pthread_mutex_lock(&g_reap_mu);
    __atomic_store_n(&g_reap_quit, 1, __ATOMIC_RELAXED);
    pthread_cond_broadcast(&g_reap_cv);
    while (!g_reap_parked) pthread_cond_wait(&g_reap_parked_cv, &g_reap_mu);
    pthread_mutex_unlock(&g_reap_mu);
}
static void free_retired(record *R)
{
    const char *e = getenv("MZ_REAPER");
    reap *j;
    if (!R->dead.n && !R->dead_rows) { mz_blocks_drop(&R->dead); return; }
    if (e && atoi(e) == 0) { reap_now(R->dead.p, R->dead.n, R->dead_rows); memset(&R->dead, 0, sizeof R->dead); R->dead_rows = NULL; return; }
    j = (reap *)mz_xmalloc(sizeof *j);
    j->next = NULL; j->blocks = R->dead.p; j->n = R->dead.n; j->rows = R->dead_rows;
    memset(&R->dead, 0, sizeof R->dead); R->dead_rows = NULL;
    pthread_mutex_lock(&g_reap_mu);
    if (!g_reap_started) {
        pthread_t t;
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        if (pthread_create(&t, &at, reaper, NULL) != 0) {    /* no thread to be had: on the spot */
            pthread_attr_destroy(&at);
            pthread_mutex_unlock(&g_reap_mu);
            reap_now(j->blocks, j->n, j->rows); free(j);
            return;
        }
        pthread_attr_destroy(&at);
        g_reap_started = 1;
        atexit(reaper_stop);
    }
    if (g_reap_tail) g_reap_tail->next = j; else g_reap_head = j;
    g_reap_tail = j;
    pthread_cond_signal(&g_reap_cv);
    pthread_mutex_unlock(&g_reap_mu);
}

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
    e->src = clone_ali(&R->arena, a); e->cbeg = -1; e->cend = -1;
}
static void rec_part(record *R, int sink, struct mafAli *a, int cbeg, int cend)
{
    event *e = new_event(R, sink);
    e->src = clone_ali(&R->arena, a); e->cbeg = cbeg; e->cend = cend;
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
        if (R->keep_blocks) {                               /* what the text path prints, as a block */
            if (e->cbeg < 0) e->blk = mz_ali_copy(e->src);
            else {
                struct mafAli *part = make_part_ali_col(e->src, e->cbeg, e->cend);
                if (part && (row2 == 0 || part->components->next != NULL)) e->blk = part;      /* (print_part_ali_col's condition) */
                else mafAliFree(&part);
            }
            if (e->blk) mz_ali_as_reread(e->blk);
            e->src = NULL;
            continue;
        }
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
    g->a1 = clone_ali(&R->arena, a1); g->a2 = clone_ali(&R->arena, a2);
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
    struct mafAli *a1 = mz_pop_first(wk1), *a2 = mz_pop_first(wk2);
    int off1 = 0, off2 = 0;                                 /* columns cut off a1 / a2 so far (keep_from) */
#define BEG(a) ((a)->components->start)
#define END(a) ((a)->components->start + (a)->components->size - 1)
#define WANTED(a) ((a)->components->size >= minw && (row2 == 0 || (a)->components->next != NULL))
    for (;;) {
        int beg1, end1, beg2, end2, beg, end, cb, ce;
        while (a1 && (!a2 || END(a1) < BEG(a2))) {         /* nothing of file 2 under a1 */
            if (R->has1 && WANTED(a1)) rec_block(R, SINK_1, a1);
            retire(R, a1, off1);
            a1 = mz_pop_first(wk1); off1 = 0;
        }
        while (a2 && (!a1 || END(a2) < BEG(a1))) {
            if (R->has2 && WANTED(a2)) rec_block(R, SINK_2, a2);
            retire(R, a2, off2);
            a2 = mz_pop_first(wk2); off2 = 0;
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

        if (end1 < end2) a2 = keep_from(R, a2, end1 + 1, &off2);
        if (end2 < end1) a1 = keep_from(R, a1, end2 + 1, &off1);
        if (end1 <= end2) {
            ce = mafPos2Col(a1->components, end1, a1->textSize);
            if (ce < a1->textSize - 1 && R->has1) rec_part(R, SINK_1, a1, ce + 1, a1->textSize - 1);
            retire(R, a1, off1);
            a1 = mz_pop_first(wk1); off1 = 0;
        }
        if (end2 <= end1) {
            ce = mafPos2Col(a2->components, end2, a2->textSize);
            if (ce < a2->textSize - 1 && R->has2) rec_part(R, SINK_2, a2, ce + 1, a2->textSize - 1);
            retire(R, a2, off2);
            a2 = mz_pop_first(wk2); off2 = 0;
        }
    }
#undef BEG
#undef END
#undef WANTED
}

/* The walk of one contig in pieces, side by side.  Both lists are sorted by their start in the reference and cover it once, so
 * wherever the next block of either list starts behind everything that came before in BOTH lists, the walk's state is "nothing in
 * hand": what it records for the blocks in front of such a point does not depend on the blocks behind it (every comparison the loop
 * makes with a block behind the point comes out as it does with no block at all), and the other way round.  The lists are cut at such
 * points into a few dozen pieces of about equal size, every piece is walked into a record of its own (OpenMP; inside the tree driver's
 * per-node loop this region is a nested one and runs on the calling thread), and the records are joined in order.  A list that is
 * not sorted by start is walked in one piece, as ever. */
#define WALK_PIECE_MIN 2048                                 /* blocks (both lists) below which a contig is not worth cutting */
static int walk_piece_min(void)
{
    const char *pm = getenv("MZ_WALK_PIECE_MIN");           /* (tests: pieces of a few blocks; read per contig -- the tree driver walks several nodes' lists at once) */
    return pm && atoi(pm) > 1 ? atoi(pm) : WALK_PIECE_MIN;
}

/* the blocks of the two lists in arr1[0..n1) / arr2[0..n2) (still linked), the first and last reference position of each in b / e.
 * Returns 0 with the lists as they were when the contig is not cut (lists out of order, too few blocks or no place to cut) */
static int walk_pieces(record *R, struct mafAli **arr1, const int *b1, const int *e1, int n1, struct mafAli **arr2, const int *b2, const int *e2, int n2,
                       int piece_min, int v, int radius, int minw)
{
    int i1, i2, npieces = 0, want, k, sorted = 1;
    int *cut1, *cut2, *ev0, *mg0;
    long long reach;
    record *piece;
    for (k = 1; k < n1 && sorted; ++k) if (b1[k] < b1[k - 1]) sorted = 0;
    for (k = 1; k < n2 && sorted; ++k) if (b2[k] < b2[k - 1]) sorted = 0;
    want = (n1 + n2) / (piece_min / 2 > 0 ? piece_min / 2 : 1);
    if (want > 4 * MZ_STAGE_THREADS) want = 4 * MZ_STAGE_THREADS;
    if (!sorted || want < 2) return 0;
    cut1 = (int *)mz_xmalloc(((size_t)want + 2) * sizeof(int)); cut2 = (int *)mz_xmalloc(((size_t)want + 2) * sizeof(int));
    /* the two lists in merged order; a piece may end in front of a block that starts behind `reach` */
    cut1[0] = cut2[0] = 0; npieces = 1;
    reach = -1;
    for (i1 = i2 = 0; i1 < n1 || i2 < n2; ) {
        const int take1 = i2 >= n2 || (i1 < n1 && b1[i1] <= b2[i2]);
        const int bb = take1 ? b1[i1] : b2[i2], be = take1 ? e1[i1] : e2[i2];
        if ((i1 || i2) && (long long)bb > reach && npieces < want &&
            (long long)(i1 + i2) * want >= (long long)npieces * (n1 + n2)) { cut1[npieces] = i1; cut2[npieces] = i2; ++npieces; }
        if ((long long)be > reach) reach = be;
        if (take1) ++i1; else ++i2;
    }
    cut1[npieces] = n1; cut2[npieces] = n2;
    if (npieces < 2) { free(cut1); free(cut2); return 0; }
    for (k = 1; k < npieces; ++k) {                         /* the lists, severed at the cuts */
        if (cut1[k] > 0) arr1[cut1[k] - 1]->next = NULL;
        if (cut2[k] > 0) arr2[cut2[k] - 1]->next = NULL;
    }
    piece = (record *)mz_xmalloc((size_t)npieces * sizeof *piece);
#pragma omp parallel for schedule(dynamic, 1) num_threads(MZ_STAGE_THREADS)
    for (k = 0; k < npieces; ++k) {
        struct mafAli *l1 = cut1[k] < cut1[k + 1] ? arr1[cut1[k]] : NULL, *l2 = cut2[k] < cut2[k + 1] ? arr2[cut2[k]] : NULL;
        memset(&piece[k], 0, sizeof piece[k]);
        piece[k].has1 = R->has1; piece[k].has2 = R->has2; piece[k].keep_blocks = R->keep_blocks;
        walk(&piece[k], &l1, &l2, v, radius, minw);
    }
    /* joined in order: event and merge numbers shift (the pieces' records copied side by side) */
    ev0 = (int *)mz_xmalloc(((size_t)npieces + 1) * sizeof(int)); mg0 = (int *)mz_xmalloc(((size_t)npieces + 1) * sizeof(int));
    ev0[0] = R->nev; mg0[0] = R->nmg;
    for (k = 0; k < npieces; ++k) { ev0[k + 1] = ev0[k] + piece[k].nev; mg0[k + 1] = mg0[k] + piece[k].nmg; }
    if (ev0[npieces] > R->capev) { R->capev = 2 * ev0[npieces] + 256; R->ev = (event *)realloc(R->ev, (size_t)R->capev * sizeof(event)); if (!R->ev) mz_fatalf("out of memory"); }
    if (mg0[npieces] > R->capmg) { R->capmg = 2 * mg0[npieces] + 256; R->mg = (merge *)realloc(R->mg, (size_t)R->capmg * sizeof(merge)); if (!R->mg) mz_fatalf("out of memory"); }
#pragma omp parallel for schedule(dynamic, 1) num_threads(MZ_STAGE_THREADS)
    for (k = 0; k < npieces; ++k) {
        record *P = &piece[k];
        int i;
        if (P->nev) memcpy(R->ev + ev0[k], P->ev, (size_t)P->nev * sizeof(event));
        if (P->nmg) memcpy(R->mg + mg0[k], P->mg, (size_t)P->nmg * sizeof(merge));
        for (i = 0; i < P->nev; ++i) if (R->ev[ev0[k] + i].job >= 0) R->ev[ev0[k] + i].job += mg0[k];
        for (i = 0; i < P->nmg; ++i) R->mg[mg0[k] + i].side_ev += ev0[k];
        free(P->ev); free(P->mg);
    }
    R->nev = ev0[npieces]; R->nmg = mg0[npieces];
    for (k = 0; k < npieces; ++k) {
        record *P = &piece[k];
        if (P->arena) { arena_chunk *t = P->arena; while (t->next) t = t->next; t->next = R->arena; R->arena = P->arena; }
        if (P->dead_rows) { struct mafComp *t = P->dead_rows; while (t->next) t = t->next; t->next = R->dead_rows; R->dead_rows = P->dead_rows; }
    }
    {   /* the pieces' retired blocks, behind one another */
        int total = R->dead.n, at;
        for (k = 0; k < npieces; ++k) total += piece[k].dead.n;
        if (total > R->dead.cap) { R->dead.cap = total; R->dead.p = (struct mafAli **)realloc(R->dead.p, (size_t)total * sizeof *R->dead.p); if (!R->dead.p) mz_fatalf("out of memory"); }
        for (k = 0, at = R->dead.n; k < npieces; ++k) {
            if (piece[k].dead.n) memcpy(R->dead.p + at, piece[k].dead.p, (size_t)piece[k].dead.n * sizeof *R->dead.p);
            at += piece[k].dead.n;
            mz_blocks_drop(&piece[k].dead);
        }
        R->dead.n = total;
    }
    free(piece); free(cut1); free(cut2); free(ev0); free(mg0);
    return 1;
}

static void walk_contig(record *R, struct mafAli **wk1, struct mafAli **wk2, int v, int radius, int minw)
{
    struct mafAli *a, **arr1, **arr2;
    int n1 = 0, n2 = 0, *key, cut;
    const int piece_min = walk_piece_min();
    for (a = *wk1; a; a = a->next) ++n1;
    for (a = *wk2; a; a = a->next) ++n2;
    if (n1 + n2 < piece_min) { walk(R, wk1, wk2, v, radius, minw); return; }
    arr1 = (struct mafAli **)mz_xmalloc(((size_t)n1 + 1) * sizeof *arr1);
    arr2 = (struct mafAli **)mz_xmalloc(((size_t)n2 + 1) * sizeof *arr2);
    key = (int *)mz_xmalloc(2 * ((size_t)n1 + (size_t)n2 + 1) * sizeof(int));
    for (a = *wk1, n1 = 0; a; a = a->next, ++n1) { arr1[n1] = a; key[n1] = a->components->start; }
    for (a = *wk2, n2 = 0; a; a = a->next, ++n2) { arr2[n2] = a; key[n1 + n2] = a->components->start; }
    for (cut = 0; cut < n1; ++cut) key[n1 + n2 + cut] = key[cut] + arr1[cut]->components->size - 1;
    for (cut = 0; cut < n2; ++cut) key[2 * n1 + n2 + cut] = key[n1 + cut] + arr2[cut]->components->size - 1;
    cut = walk_pieces(R, arr1, key, key + n1 + n2, n1, arr2, key + n1, key + 2 * n1 + n2, n2, piece_min, v, radius, minw);
    free(arr1); free(arr2); free(key);
    if (cut) *wk1 = *wk2 = NULL;                            /* (a walk takes every block of its lists) */
    else walk(R, wk1, wk2, v, radius, minw);
}

/* Two long lists on ONE reference contig (a level of a guide tree run chromosome by chromosome): each list walked once into an array,
 * the keys and the contig check taken from the blocks on all threads, then the walk in pieces.  Returns 0 with the lists untouched
 * when that is not the case. */
typedef struct { struct mafAli **arr; int n; } blist;
static blist list_array(struct mafAli *l)
{
    blist b = { NULL, 0 };
    int cap = 4096;
    b.arr = (struct mafAli **)mz_xmalloc((size_t)cap * sizeof *b.arr);
    for (; l; l = l->next) {
        if (b.n == cap) { cap *= 2; b.arr = (struct mafAli **)realloc(b.arr, (size_t)cap * sizeof *b.arr); if (!b.arr) mz_fatalf("out of memory"); }
        b.arr[b.n++] = l;
    }
    return b;
}
static int prepare_one_contig(record *R, struct mafAli **list1, struct mafAli **list2, mz_blocks *idx1, mz_blocks *idx2, int v, int radius, int minw)
{
    const int piece_min = walk_piece_min();
    const char *chr = (*list1)->components->src;
    blist L[2];
    int *key, n1, n2, i, other = 0, done = 0;
#pragma omp parallel for schedule(static, 1) num_threads(2)
    for (i = 0; i < 2; ++i) {
        mz_blocks *idx = i ? idx2 : idx1;
        if (idx && idx->p) { L[i].arr = idx->p; L[i].n = idx->n; idx->p = NULL; idx->n = idx->cap = 0; }      /* (the blocks are known: nothing to walk) */
        else L[i] = list_array(i ? *list2 : *list1);
    }
    n1 = L[0].n; n2 = L[1].n;
    if (n1 + n2 >= piece_min) {
        key = (int *)mz_xmalloc(2 * ((size_t)n1 + (size_t)n2 + 1) * sizeof(int));
#pragma omp parallel for schedule(static, 1024) num_threads(MZ_STAGE_THREADS) reduction(| : other)
        for (i = 0; i < n1 + n2; ++i) {
            const struct mafComp *c = (i < n1 ? L[0].arr[i] : L[1].arr[i - n1])->components;
            key[i] = c->start; key[n1 + n2 + i] = c->start + c->size - 1;
            if (c->src != chr && strcmp(c->src, chr) != 0) other |= 1;
        }
        if (!other) done = walk_pieces(R, L[0].arr, key, key + n1 + n2, n1, L[1].arr, key + n1, key + 2 * n1 + n2, n2, piece_min, v, radius, minw);
        free(key);
    }
    free(L[0].arr); free(L[1].arr);
    if (done) *list1 = *list2 = NULL;
    return done;
}

/* Stage 1 of every merge, then the pending yama() calls of all of them wave after wave.  The merges are
 * independent of one another, so everything on the host side of the yama() batches -- column packing,
 * rmColDash, the band walk and smooth() before, mafBuild() and mafScoreRange() after -- runs one merge per
 * thread.  Several RECORDS may be given: the merges of independent multiz runs (sibling nodes of a guide tree,
 * mz_roast.c) then share the same GPU batches. */
typedef struct { record *R; int i; } mref;

/* what mafBuild() does with yama()'s columns (reference mz_preyama.c:38-81), from the rows, base counts and score
 * that mz_preyama_batch() brings back: bookkeeping of the rows only -- sources in order (all of a1, then a2 below
 * its top row), starts advanced by the bases left of the slice, rows without a base dropped */
static struct mafAli *block_from_rows(const mz_preout *o, struct mafAli *a1, int cbeg1, struct mafAli *a2, int cbeg2)
{
    struct mafAli *blk = (struct mafAli *)mz_xmalloc(sizeof *blk);
    struct mafComp *src = a1->components, *tail = NULL, *nc;
    int skip = cbeg1, i, j, second = 0;
    memset(blk, 0, sizeof *blk);
    blk->textSize = o->OM;
    for (i = 0; ; ++i, src = src->next) {
        int start;
        if (src == NULL) {
            if (second) break;
            second = 1; src = a2->components->next; skip = cbeg2;
            if (src == NULL) break;
        }
        if (o->size[i] == 0) continue;
        for (start = src->start, j = 0; j < skip; ++j) start += src->text[j] != '-';
        nc = mzi_row_new(src, o->OM);
        nc->start = start;
        nc->size = o->size[i];
        memcpy(nc->text, o->rows + (size_t)i * (size_t)o->OM, (size_t)o->OM);
        if (tail) tail->next = nc; else blk->components = nc;
        tail = nc;
    }
    if (!blk->components) { free(blk); return NULL; }
    blk->score = o->score;
    return blk;
}

/* The merges with everything between block text and block text on the GPU (mz_preyama_batch()): the
 * one-stage ones (v == 1) and the two-stage ones (v == 0) whose first block has rows below its top row; the host only
 * locates the overlap columns and assembles the result's bookkeeping.  What the device path does not finish -- a merge
 * yama() refuses in its second stage, top rows that disagree ("M3 not equals N3!!") -- is left in state 0 for the host
 * stages, which reproduce the reference's messages.  Returns 0 when the score tables do not allow the device path. */
static int on_device(const merge *g)
{
    return g->v == 1 || (g->v == 0 && g->a1->components->next != NULL);
}

static int run_merges_device(mref *all, int nmg, int minw, int timing)
{
    mz_prejob *jobs;
    mz_preout *outs;
    const char **ptrs;
    int *who, *cb1, *cb2, n = 0, i, rc;
    size_t nptr = 0, *first;
    double t0 = mz_now_s(), t1;
    who = (int *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(int));
    first = (size_t *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof *first);
    /* the merges of this path, and where each one's row pointers start (the rows counted on all threads: a guide-tree level is half a
     * million merges of up to thirty rows, each row a step down a list) */
#pragma omp parallel for schedule(static, 512) num_threads(MZ_STAGE_THREADS) if (nmg > 4096)
    for (i = 0; i < nmg; ++i) {
        merge *g = &all[i].R->mg[all[i].i];
        struct mafComp *c;
        int rows = 0;
        if (on_device(g)) {
            for (c = g->a1->components; c; c = c->next) ++rows;
            for (c = g->a2->components; c; c = c->next) ++rows;
        } else rows = -1;
        who[i] = rows;
    }
    for (i = 0; i < nmg; ++i) {
        const int rows = who[i];
        if (rows < 0) continue;
        first[n] = nptr;
        who[n++] = i;                                       /* (n <= i: the entry is not needed again) */
        nptr += (size_t)rows;
    }
    if (n == 0) { free(who); free(first); return 1; }
    jobs = (mz_prejob *)mz_xmalloc((size_t)n * sizeof *jobs);
    outs = (mz_preout *)mz_xmalloc((size_t)n * sizeof *outs);
    cb1 = (int *)mz_xmalloc((size_t)n * sizeof(int)); cb2 = (int *)mz_xmalloc((size_t)n * sizeof(int));
    ptrs = (const char **)mz_xmalloc((nptr ? nptr : 1) * sizeof *ptrs);
    /* the overlap's columns in both blocks (four scans of a top row each), a merge per thread */
#pragma omp parallel for schedule(dynamic, 256) num_threads(MZ_STAGE_THREADS) if (n > 1024)
    for (i = 0; i < n; ++i) {
        merge *g = &all[who[i]].R->mg[all[who[i]].i];
        struct mafComp *c;
        mz_prejob *j = &jobs[i];
        size_t at = first[i];
        int ce1, ce2;
        j->v = g->v;
        cb1[i] = mafPos2Col(g->a1->components, g->beg, g->a1->textSize);
        ce1 = mafPos2Col(g->a1->components, g->end, g->a1->textSize);
        cb2[i] = mafPos2Col(g->a2->components, g->beg, g->a2->textSize);
        ce2 = mafPos2Col(g->a2->components, g->end, g->a2->textSize);
        j->M_all = ce1 - cb1[i] + 1; j->N_all = ce2 - cb2[i] + 1; j->radius = g->radius;
        j->rows1 = ptrs + at;
        for (c = g->a1->components, j->K = 0; c; c = c->next, ++j->K) ptrs[at++] = c->text + cb1[i];
        j->rows2 = ptrs + at;
        for (c = g->a2->components, j->L1 = 0; c; c = c->next, ++j->L1) ptrs[at++] = c->text + cb2[i];
    }
    free(first);
    rc = mz_preyama_batch(n, jobs, outs);
    if (rc == -2) { free(jobs); free(outs); free(who); free(cb1); free(cb2); free(ptrs); return 0; }
    if (rc < 0) mz_fatalf("yama(gfx950): %s", mz_last_error());
    t1 = mz_now_s();
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (n > 64)
    for (i = 0; i < n; ++i) {
        merge *g = &all[who[i]].R->mg[all[who[i]].i];
        const mz_preout *o = &outs[i];
        if (o->null_result == 1) { g->state = MZ_PY_DONE; g->result = NULL; continue; }
        if (o->null_result || (o->status != MZ_OK && g->v == 0)) continue;      /* (state 0: the host stages take it) */
        if (o->status != MZ_OK) {
            /* yama() refused the job: the message names LB / RB entries, so the host builds the job after all (rare) */
            if (mz_py_begin(&g->py, g->a1, g->a2, g->beg, g->end, g->radius, g->v, NULL) != MZ_PY_JOB) { g->state = MZ_PY_DONE; continue; }
            g->state = MERGE_FAILED; g->bad_job = g->py.job;
            memset(&g->bad_out, 0, sizeof g->bad_out);
            g->bad_out.status = o->status; g->bad_out.badrow = o->badrow; g->bad_out.OM = o->OM;
            continue;
        }
        g->result = block_from_rows(o, g->a1, cb1[i], g->a2, cb2[i]);
        g->state = MZ_PY_DONE;
        if (all[who[i]].R->keep_blocks) {                   /* the block goes on as it is, if it would have been printed */
            if (g->result && g->result->components->size < minw) mafAliFree(&g->result);
            if (g->result) mz_ali_as_reread(g->result);
            continue;
        }
        if (g->result && g->result->components->size >= minw) {
            FILE *m = open_memstream(&g->text, &g->len);
            mafWrite(m, g->result);
            fclose(m);
        }
        mafAliFree(&g->result);
    }
    mz_free_preouts(n, outs);
    if (timing) fprintf(stderr, "mz_multiz: %d merges, block text to block text on the GPU %.3f s, blocks assembled and rendered %.3f s\n",
                        n, t1 - t0, mz_now_s() - t1);
    free(jobs); free(outs); free(who); free(cb1); free(cb2); free(ptrs);
    return 1;
}

static void run_merges(record **RR, int nrec, int minw)
{
    int nmg = 0, i, r, nheld = 0;
    void **held = NULL;
    mz_job *jobs;
    mz_out *outs;
    int *who;
    mref *all;
    const int timing = mzi_timing() != 0;
    double t0 = mz_now_s(), t1;
    for (r = 0; r < nrec; ++r) nmg += RR[r]->nmg;
    jobs = (mz_job *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(mz_job));
    outs = (mz_out *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(mz_out));
    who = (int *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(int));
    all = (mref *)mz_xmalloc((size_t)(nmg ? nmg : 1) * sizeof(mref));
    for (r = 0, nmg = 0; r < nrec; ++r)
        for (i = 0; i < RR[r]->nmg; ++i) { all[nmg].R = RR[r]; all[nmg++].i = i; }
    mz_score_profile_sync();
    for (r = 0; r < nrec; ++r) render_events(RR[r]);
    /* slicing, dash columns, band, yama() -- both of them when v == 0 --, transposition, base counts and score all on the
     * GPU (MZ_HOST_PREP=1 keeps every merge on the host stages below) */
    {
        const char *e = getenv("MZ_HOST_PREP");
        if (!(e && atoi(e) != 0)) run_merges_device(all, nmg, minw, timing);
    }
    t0 = mz_now_s();
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (nmg > 64)
    for (i = 0; i < nmg; ++i) {
        record *R = all[i].R;
        merge *g = &R->mg[all[i].i];
        event *e = &R->ev[g->side_ev];
        FILE *m;
        if (g->state == MZ_PY_DONE || g->state == MERGE_FAILED) continue;      /* taken care of on the device path */
        m = open_memstream(&e->text, &e->len);
        g->state = mz_py_begin(&g->py, g->a1, g->a2, g->beg, g->end, g->radius, g->v, R->has2 ? m : NULL);
        fclose(m);
    }
    t1 = mz_now_s();
    if (timing) fprintf(stderr, "mz_multiz: stage 1 of %d merges (%d run%s) %.3f s\n", nmg, nrec, nrec == 1 ? "" : "s", t1 - t0);
    for (;;) {
        int n = 0, rc;
        for (i = 0; i < nmg; ++i) {
            merge *g = &all[i].R->mg[all[i].i];
            if (g->state == MZ_PY_JOB) { jobs[n] = g->py.job; who[n++] = i; }
        }
        if (n == 0) break;
        t0 = mz_now_s();
        rc = mz_yama_batch(n, jobs, outs);
        if (rc < 0) mz_fatalf("yama(gfx950): %s", mz_last_error());
        /* the merged columns live in the call's result blocks; a v == 0 merge aligns against them in the next wave, so
         * the blocks are kept until the last wave is through */
        for (i = 0; i < n; ++i) if (outs[i].block) held = mz_hold(held, &nheld, outs[i].block);
        t1 = mz_now_s();
#pragma omp parallel for schedule(dynamic, 16) num_threads(MZ_STAGE_THREADS) if (n > 64)
        for (i = 0; i < n; ++i) {
            merge *g = &all[who[i]].R->mg[all[who[i]].i];
            if (outs[i].status != MZ_OK) {                  /* reported at its place in the output order, see replay() */
                g->state = MERGE_FAILED; g->bad_job = jobs[i]; g->bad_out = outs[i];
            } else {
                g->py.borrowed = 1;
                g->state = mz_py_step(&g->py, outs[i].cols, outs[i].OM, &g->result);
            }
            if (g->state != MZ_PY_JOB && g->state != MERGE_FAILED && all[who[i]].R->keep_blocks) {
                if (g->result && g->result->components->size < minw) mafAliFree(&g->result);
                if (g->result) mz_ali_as_reread(g->result);
            } else if (g->state != MZ_PY_JOB && g->state != MERGE_FAILED) {    /* finished: render and release here */
                if (g->result && g->result->components->size >= minw) {
                    FILE *m = open_memstream(&g->text, &g->len);
                    mafWrite(m, g->result);
                    fclose(m);
                }
                mafAliFree(&g->result);     /* built by this thread: released into its own arena (cheap); a1 / a2
                                             * came from the main thread and go back there, in replay() */
            }
        }
        if (timing) fprintf(stderr, "mz_multiz: yama batch of %d %.3f s (with GPU start-up in the first), next stage %.3f s\n", n, t1 - t0, mz_now_s() - t1);
    }
    for (i = 0; i < nheld; ++i) free(held[i]);
    free(held);
    free(jobs); free(outs); free(who); free(all);
    /* nothing looks at the snapshots' names and text any more (what is left to replay are results and rendered text): the blocks
     * the walks retired go, those of all records side by side */
    for (r = 0; r < nrec; ++r) free_retired(RR[r]);
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
    free_retired(R);
    arena_release(&R->arena);
    memset(R, 0, sizeof *R);
}

/* the same replay into three LISTS (record.keep_blocks): every block as the next program of the stock chain would read it from the text
 * this replay does not write (mz_ali_as_reread).  The one piece of text a kept-blocks run still holds -- pre_yama()'s side write to out2
 * when nothing of the first block is left to align (mz_preyama.c:193-196: host stage 1) -- is read back here. */
typedef struct { struct mafAli *head, *tail; mz_blocks idx; } alist;
static void alist_add(alist *l, struct mafAli *a)          /* (a: one block, or a chain; already as their reader would hold them) */
{
    if (!a) return;
    if (l->tail) l->tail->next = a; else l->head = a;
    for (;;) { mz_blocks_push(&l->idx, a); if (!a->next) break; a = a->next; }
    l->tail = a;
}
static void alist_add_one(alist *l, struct mafAli *a)      /* (a: ONE block -- nothing of it is read here) */
{
    a->next = NULL;
    if (l->tail) l->tail->next = a; else l->head = a;
    mz_blocks_push(&l->idx, a);
    l->tail = a;
}
static void replay_lists(record *R, struct mafAli **out, struct mafAli **f1, struct mafAli **f2, struct mafAli **tails, mz_blocks *idx)
{
    alist L[3];
    int i;
    memset(L, 0, sizeof L);
    for (i = 0; i < R->nev; ++i) {
        event *e = &R->ev[i];
        if (e->job >= 0) {
            merge *g = &R->mg[e->job];
            if (g->state == MERGE_FAILED) mz_fatal_status(&g->bad_job, &g->bad_out);
            if (g->result) { alist_add_one(&L[SINK_OUT], g->result); g->result = NULL; }
            free(g->text);
            g->a1 = g->a2 = NULL;                           /* (arena) */
        } else if (e->blk) {
            if ((e->sink == SINK_1 && !R->has1) || (e->sink == SINK_2 && !R->has2)) mafAliFree(&e->blk);
            else alist_add_one(&L[e->sink], e->blk);
            e->blk = NULL;
        } else if (e->text) {
            if (e->len && !((e->sink == SINK_1 && !R->has1) || (e->sink == SINK_2 && !R->has2))) {
                static const char head[] = "##maf version=1\n";
                char *t = (char *)mz_xmalloc(sizeof head + e->len);
                memcpy(t, head, sizeof head - 1); memcpy(t + sizeof head - 1, e->text, e->len);
                alist_add(&L[e->sink], mz_maf_read_mem(t, sizeof head - 1 + e->len, "a side write"));
                free(t);
            }
            free(e->text);
        }
    }
    free(R->ev); free(R->mg);
    free_retired(R);
    arena_release(&R->arena);
    memset(R, 0, sizeof *R);
    *out = L[SINK_OUT].head;
    if (tails) { tails[0] = L[SINK_OUT].tail; tails[1] = f1 ? L[SINK_1].tail : NULL; tails[2] = f2 ? L[SINK_2].tail : NULL; }
    for (i = 0; i < 3; ++i) {
        if (idx && (i == 0 || (i == 1 ? f1 != NULL : f2 != NULL))) idx[i] = L[i].idx;
        else { if (idx) memset(&idx[i], 0, sizeof idx[i]); mz_blocks_drop(&L[i].idx); }
    }
    if (f1) *f1 = L[SINK_1].head; else { struct mafAli *a = L[SINK_1].head; while (a) { struct mafAli *n = a->next; a->next = NULL; mafAliFree(&a); a = n; } }
    if (f2) *f2 = L[SINK_2].head; else { struct mafAli *a = L[SINK_2].head; while (a) { struct mafAli *n = a->next; a->next = NULL; mafAliFree(&a); a = n; } }
}

/* A multiz run in three steps, so that several independent runs can share their GPU batches:
 *   mz_multiz_prepare()  walks the two lists (reference multiz.c:60-177) and records the output events and merges;
 *   mz_multiz_align()    runs every pending merge of the given runs: stage 1 on the host threads, the yama() calls
 *                        of ALL runs as one mz_yama_batch() per wave, stages 2/3;
 *   mz_multiz_finish()   replays the record of a run into its sinks, in the stock driver's order, and frees it. */
struct mz_mzrun { record R; int minw; };

struct mz_mzrun *mz_multiz_prepare(struct mafAli **list1, struct mafAli **list2, int v, int radius, int min_output_wid,
                                   int has_out1, int has_out2)
{
    return mzi_multiz_prepare_blocks(list1, list2, NULL, NULL, v, radius, min_output_wid, has_out1, has_out2);
}
/* the same with the lists' indexes, where they are known (consumed either way) */
struct mz_mzrun *mzi_multiz_prepare_blocks(struct mafAli **list1, struct mafAli **list2, mz_blocks *idx1, mz_blocks *idx2, int v, int radius,
                                           int min_output_wid, int has_out1, int has_out2)
{
    struct mz_mzrun *run = (struct mz_mzrun *)mz_xmalloc(sizeof *run);
    memset(run, 0, sizeof *run);
    run->minw = min_output_wid;
    run->R.has1 = has_out1; run->R.has2 = has_out2;
    if (*list1 && *list2 && prepare_one_contig(&run->R, list1, list2, idx1, idx2, v, radius, min_output_wid)) return run;
    mz_blocks_drop(idx1); mz_blocks_drop(idx2);
    while (*list1 && *list2) {                              /* one reference contig at a time, in file-1 order */
        struct mafAli *wk1 = NULL, *wk2 = NULL;
        char *chr = mz_xstrdup((*list1)->components->src);
        mz_take_chr(list1, &wk1, chr);
        mz_take_chr(list2, &wk2, chr);
        free(chr);
        walk_contig(&run->R, &wk1, &wk2, v, radius, min_output_wid);
    }
    return run;
}

void mz_multiz_align(struct mz_mzrun **runs, int n)
{
    record **RR = (record **)mz_xmalloc((size_t)(n ? n : 1) * sizeof *RR);
    int i;
    for (i = 0; i < n; ++i) RR[i] = &runs[i]->R;
    if (n > 0) run_merges(RR, n, runs[0]->minw);            /* (runs of one call share min_output_wid: one command line) */
    free(RR);
}

void mz_multiz_finish(struct mz_mzrun *run, FILE *out, FILE *out1, FILE *out2)
{
    replay(&run->R, out, out1, out2, run->minw);
    free(run);
}

/* the run's output as block lists instead of text: call mz_multiz_keep_blocks() between prepare and align, then this instead of
 * mz_multiz_finish().  The lists hold what a reader of the three text streams would hold (mz_ali_as_reread), in their order. */
void mz_multiz_keep_blocks(struct mz_mzrun *run) { run->R.keep_blocks = 1; }
void mz_multiz_finish_lists(struct mz_mzrun *run, struct mafAli **out, struct mafAli **out1, struct mafAli **out2)
{
    mzi_multiz_finish_tails(run, out, out1, out2, NULL, NULL);
}
/* the same, and the last block of each list in tails[0..2] (NULL for an empty one): the tree driver goes on appending to them */
/* ... and the blocks of each list in idx[0..2] */
void mzi_multiz_finish_tails(struct mz_mzrun *run, struct mafAli **out, struct mafAli **out1, struct mafAli **out2, struct mafAli **tails, mz_blocks *idx)
{
    if (!run->R.keep_blocks) mz_fatalf("mz_multiz_finish_lists: the run was aligned for text");
    replay_lists(&run->R, out, out1, out2, tails, idx);
    free(run);
}

int mz_multiz_lists(struct mafAli **list1, struct mafAli **list2, int v, int radius, int min_output_wid,
                    FILE *out, FILE *out1, FILE *out2)
{
    const int timing = mzi_timing() != 0;
    const double t0 = mz_now_s();
    double t1, t2;
    struct mz_mzrun *run = mz_multiz_prepare(list1, list2, v, radius, min_output_wid, out1 != NULL, out2 != NULL);
    const int nmerge = run->R.nmg;
    mz_multiz_align(&run, 1);
    t1 = mz_now_s();
    mz_multiz_finish(run, out, out1, out2);
    t2 = mz_now_s();
    if (timing) fprintf(stderr, "mz_multiz: %d merges; walk + yama batches + stage 2/3 %.3f s, replay %.3f s\n", nmerge, t1 - t0, t2 - t1);
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
    args = (char *)mz_xmalloc(na);
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
        const double t0 = mz_now_s();
        double t1, t2;
        mz_tune_malloc();
        init_scores70();
        mz_warm_start();                                     /* the GPU starts up while the inputs are read */
        l1 = mz_maf_read_all(argv[1], 1);
        l2 = mz_maf_read_all(argv[2], 1);
        t1 = mz_now_s();
        mz_multiz_lists(&l1, &l2, v, radius, minw, stdout, f1, f2);
        t2 = mz_now_s();
        if (mzi_timing()) fprintf(stderr, "mz_multiz: read %.3f s, walk + merges + replay %.3f s\n", t1 - t0, t2 - t1);
    }

    for (a = l1; a; a = a->next)                            /* contigs that only one file has */
        if (f1 && (row2 == 0 || a->components->next != NULL)) mafWrite(f1, a);
    for (a = l2; a; a = a->next)
        if (f2 && (row2 == 0 || a->components->next != NULL)) mafWrite(f2, a);
    while (l1) { a = mz_pop_first(&l1); mafAliFree(&a); }
    while (l2) { a = mz_pop_first(&l2); mafAliFree(&a); }
    /* The stock driver closes out1 and out2 before it writes the trailer -- and without [out1 out2] both
     * ARE stdout (multiz.c:243-245,286-291), so its "##eof maf" line never reaches the output.  Same here. */
    if (f1 != stdout) { fclose(f1); fclose(f2); fprintf(stdout, "##eof maf\n"); }
    free(args);
    return 0;
}

