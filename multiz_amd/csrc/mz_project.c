/* mz_project.c -- projecting a list of blocks onto a reference species, as the tree driver needs it between
 * merges.
 *
 * The stock roast (reference auto_mz.c:84-85,276) runs `maf_project file REF others > out` around every multiz
 * call: blocks that have a row of the reference get that row on top (reverse-complemented when it lies on the
 * minus strand), are ordered by their start in the reference contig by contig, and neighbours that continue one
 * another exactly in every row are fused (reference maf_project.c:61-84 abut, :86-173 accordion + fuse, :616-719
 * main).  That four-argument form is what is restated here; the "beautify" pass of the stand-alone tool (used only
 * when no file for the other blocks is named, maf_project.c:367-481) and its from/to and tree arguments are not.
 */
#include <pthread.h>
#include "mz_drivers.h"

/* complement of a nucleotide code, case kept; anything that is not a code becomes a blank (the table of
 * reference multi_util.c:34-38) */
static unsigned char g_compl[256];
static pthread_once_t g_compl_once = PTHREAD_ONCE_INIT;   /* begin_node() projects both sides in parallel sections */
static void compl_fill(void)
{
    static const char pairs[] = "ATCGBVDHKMRYSSWWXXNN";
    int i;
    memset(g_compl, ' ', sizeof g_compl);
    g_compl['-'] = '-';
    for (i = 0; pairs[i]; i += 2) {
        const int a = pairs[i], b = pairs[i + 1];
        g_compl[a] = (unsigned char)b; g_compl[b] = (unsigned char)a;
        g_compl[a | 0x20] = (unsigned char)(b | 0x20); g_compl[b | 0x20] = (unsigned char)(a | 0x20);
    }
}

/* every row of the block onto the other strand (reference multi_util.c:44-67) */
static void block_revcomp(struct mafAli *a)
{
    struct mafComp *c;
    pthread_once(&g_compl_once, compl_fill);
    for (c = a->components; c; c = c->next) {
        char *s = c->text, *p = c->text + a->textSize - 1;
        c->start = c->srcSize - (c->start + c->size);
        c->strand = c->strand == '-' ? '+' : '-';
        while (s <= p) {
            const char t = (char)g_compl[(unsigned char)*s];
            *s = (char)g_compl[(unsigned char)*p];
            *p = t;
            ++s; --p;
        }
    }
}

static struct mafComp *row_of(struct mafAli *a, const char *src)
{
    struct mafComp *c;
    for (c = a->components; c; c = c->next)
        if (strcmp(c->src, src) == 0) return c;
    return NULL;
}

/* The same look-up behind a 15-bit hash of the source name, kept in the row's nameID (nothing in this library reads that field; 0 = not
 * computed yet; whoever rewrites a row's src -- mz_ali_as_reread() -- resets it): the fusion pass asks for every row of a block in its
 * neighbour, twice, and a guide-tree level is half a million blocks of up to thirty rows -- 2 s of string comparisons in a 9 s run. */
static short src_hash(struct mafComp *c)                     /* (relaxed atomics: the long fusion pass asks from several threads, and two of them
                                                               * may meet in a block at the border of their ranges -- with the same value) */
{
    short v = __atomic_load_n(&c->nameID, __ATOMIC_RELAXED);
    if (v == 0) {
        unsigned h = 2166136261u;
        const unsigned char *p;
        for (p = (const unsigned char *)c->src; *p; ++p) h = (h ^ *p) * 16777619u;
        v = (short)(1 + ((h ^ (h >> 15)) % 32767u));
        __atomic_store_n(&c->nameID, v, __ATOMIC_RELAXED);
    }
    return v;
}
static struct mafComp *row_of_hashed(struct mafAli *a, struct mafComp *like)
{
    const short h = src_hash(like);
    struct mafComp *c;
    for (c = a->components; c; c = c->next)
        if (src_hash(c) == h && strcmp(c->src, like->src) == 0) return c;
    return NULL;
}

/* b continues a exactly: the same sources in both, every row of b starting where its row of a ends
 * (reference maf_project.c:61-84) */
static int continues(struct mafAli *a, struct mafAli *b)
{
    struct mafComp *c, *d;
    for (c = a->components; c; c = c->next) {
        d = row_of_hashed(b, c);
        if (!d || d->paralog != c->paralog || c->strand != d->strand || c->start + c->size != d->start) return 0;
    }
    for (c = b->components; c; c = c->next) {
        d = row_of_hashed(a, c);
        if (!d || d->paralog != c->paralog || c->strand != d->strand || d->start + d->size != c->start) return 0;
    }
    return 1;
}

/* after a fusion at column n1: if every row has dashes next to the seam, the narrowest such run is squeezed out
 * (reference maf_project.c:86-111) */
static void close_seam(struct mafAli *a, int n1)
{
    const int n = a->textSize;
    struct mafComp *c;
    int i, room = n;
    for (c = a->components; c; c = c->next) {
        int sp = 0;
        for (i = n1 - 1; i >= 0 && c->text[i] == '-'; --i) ++sp;
        for (i = n1; i < n && c->text[i] == '-'; ++i) ++sp;
        if (sp < room) room = sp;
    }
    if (room <= 0) return;
    for (c = a->components; c; c = c->next) {
        for (i = n1; i > 0 && c->text[i - 1] == '-'; --i)
            ;
        for (; i + room <= n; ++i) c->text[i] = c->text[i + room];      /* (the terminating 0 moves too) */
    }
    a->textSize -= room;
}

/* b appended to a (reference maf_project.c:114-173); b is left untouched and still owned by the caller */
static void append_block(struct mafAli *a, struct mafAli *b)
{
    const int n1 = a->textSize, n2 = b->textSize, n = n1 + n2;
    struct mafComp *c, *d, *extra = NULL, *x;
    a->textSize = n;
    for (c = a->components; c; c = c->next) {
        char *t = (char *)mz_xmalloc((size_t)n + 1);
        memcpy(t, c->text, (size_t)n1);
        t[n] = 0;
        d = row_of(b, c->src);
        if (d) {
            if (d->strand != c->strand || d->start != c->start + c->size) mz_fatalf("possible use of unprojected alignment");
            memcpy(t + n1, d->text, (size_t)n2);
            c->size += d->size;
        } else memset(t + n1, '-', (size_t)n2);
        mzi_row_free_field(c, c->text);
        c->text = t;
    }
    for (d = b->components; d; d = d->next)              /* rows only b has: dashes on the left, collected in reverse */
        if (!row_of(a, d->src)) {
            x = mafCpyComp(d);
            x->text = (char *)mz_xmalloc((size_t)n + 1);
            memset(x->text, '-', (size_t)n1);
            memcpy(x->text + n1, d->text, (size_t)n2);
            x->text[n] = 0;
            x->next = extra;
            extra = x;
        }
    for (c = a->components; c->next; c = c->next)
        ;
    c->next = extra;
    close_seam(a, n1);
    a->score = mafScoreRange(a, 0, a->textSize);
}

/* the blocks of a contig are sorted by the start of their top row with the library's qsort, as the stock tool sorts them (ties fall as
 * the library lets them fall).  The keys travel with the pointers: the comparison function sees the same values in the same order of
 * calls -- so the permutation is the library's, whatever its algorithm -- without two dependent cache misses per look at a block
 * (a guide-tree level is hundreds of thousands of blocks: 1.4 s of a 9 s run went here). */
typedef struct { int start; struct mafAli *a; } keyed;
static int by_top_start(const void *x, const void *y)
{
    return ((const keyed *)x)->start - ((const keyed *)y)->start;
}

static void fuse_neighbours(struct mafAli *list)
{
    struct mafAli *a, *b;
    for (a = list; (b = a->next) != NULL; )
        if (continues(a, b)) {
            append_block(a, b);
            a->next = b->next;
            b->next = NULL;
            mafAliFree(&b);
        } else a = b;
}

/* The same pass over a long list, with the question "does b continue a" asked for every neighbouring pair of the list AS IT STANDS on
 * all threads first (the answers only read the blocks: a guide-tree level is half a million blocks of up to thirty rows, all but a
 * few thousand of the answers are no, and each costs a chase through both blocks' rows), and the fusions made in one pass after it.
 * A fusion changes its left block, so the pair behind a fused one is asked again, there and then, as the plain pass would. */
#define FUSE_PARALLEL_MIN 20000
static void fuse_neighbours_long(struct mafAli *list)
{
    struct mafAli *a, *b, **arr;
    unsigned char *yes;
    int n = 0, i;
    const char *e = getenv("MZ_FUSE_PARALLEL_MIN");         /* (tests: the long form on short lists) */
    for (a = list; a; a = a->next) ++n;
    if (n < (e && atoi(e) > 1 ? atoi(e) : FUSE_PARALLEL_MIN)) { fuse_neighbours(list); return; }
    arr = (struct mafAli **)mz_xmalloc((size_t)n * sizeof *arr);
    yes = (unsigned char *)mz_xmalloc((size_t)n);
    for (a = list, i = 0; a; a = a->next) arr[i++] = a;
#pragma omp parallel for schedule(static, 512) num_threads(MZ_STAGE_THREADS)
    for (i = 0; i < n - 1; ++i) yes[i] = (unsigned char)continues(arr[i], arr[i + 1]);
    yes[n - 1] = 0;
    for (a = list, i = 0; (b = a->next) != NULL; ) {        /* a == arr[i] or a block that has grown by fusions; b == arr[i + 1] */
        const int fuse = (a == arr[i]) ? yes[i] : continues(a, b);
        ++i;
        if (fuse) {
            append_block(a, b);
            a->next = b->next;
            b->next = NULL;
            mafAliFree(&b);
        } else a = b;
    }
    free(arr); free(yes);
}

/* Project `all` (blocks in file order; consumed) onto `target` (a species name or a full source name).  Returns
 * the projected blocks in output order; blocks without a row of the target go to *others in file order (freed
 * when others == NULL). */
static struct mafAli *project_plain(struct mafAli *all, const char *target, struct mafAli **others)
{
    struct mafAli *A = NULL, *out = NULL, *out_tail = NULL, *oth_tail = NULL, *a, *next;
    if (others) *others = NULL;
    for (a = all; a; a = next) {
        struct mafComp *c, *b;
        next = a->next;
        a->next = NULL;
        for (c = a->components; c; c = c->next)
            if (strcmp(c->name, target) == 0 || strcmp(c->src, target) == 0) break;
        if (!c) {
            if (others) { if (oth_tail) oth_tail->next = a; else *others = a; oth_tail = a; }
            else mafAliFree(&a);
            continue;
        }
        if (c != a->components) {                        /* the target's row to the top */
            for (b = a->components; b && b->next != c; b = b->next)
                ;
            if (!b) mz_fatalf("maf_project: cannot happen");
            b->next = c->next;
            c->next = a->components;
            a->components = c;
        }
        if (c->strand == '-') block_revcomp(a);
        a->next = A;                                     /* (the stock tool collects them back to front) */
        A = a;
    }
    init_scores70();
    while (A) {                                          /* one reference contig at a time */
        const char *chr = A->components->src;
        struct mafAli *B = NULL, *prev;
        keyed *arr;
        int n = 0, i;
        for (prev = a = A; a; a = next) {
            next = a->next;
            if (strcmp(chr, a->components->src) != 0) { prev->next = next; a->next = B; B = a; }
            else prev = a;
        }
        for (a = A; a; a = a->next) ++n;
        arr = (keyed *)mz_xmalloc((size_t)n * sizeof *arr);
        for (a = A, i = 0; a; a = a->next, ++i) { arr[i].start = a->components->start; arr[i].a = a; }
        qsort(arr, (size_t)n, sizeof *arr, by_top_start);        /* the same library sort on the same order: ties fall alike */
        for (i = 1; i < n; ++i) arr[i - 1].a->next = arr[i].a;
        arr[n - 1].a->next = NULL;
        a = arr[0].a;
        free(arr);
        fuse_neighbours_long(a);
        fuse_neighbours_long(a);                         /* (the stock tool's second pass, maf_project.c:690-695) */
        if (out_tail) out_tail->next = a; else out = a;
        for (out_tail = a; out_tail->next; out_tail = out_tail->next)
            ;
        A = B;
    }
    return out;
}

/* ------------------------------------------------------------------------------------------------ long lists
 * The same projection for a list of tens or hundreds of thousands of blocks (a level of a guide tree), where the plain form's time
 * goes into walking the list again and again -- every pass a chain of cache misses, a dozen passes: here the list is walked ONCE
 * into an array, the per-block work (finding the target's row, moving it to the top, the reverse complement) runs on all threads,
 * contigs are told apart by number, and the fusion passes and the final linking work on the array.  The ORDER of everything the plain
 * form does is kept: the blocks reach the library's qsort in the order the stock tool's back-to-front collection and contig-by-contig
 * partition give them (so ties fall as they do there), and fusions are made left to right with the same questions asked. */
typedef struct { struct mafAli *a; const char *src; unsigned hash; int start, contig; } pitem;

static unsigned str_hash(const char *s)
{
    unsigned h = 2166136261u;
    for (; *s; ++s) h = (h ^ (unsigned char)*s) * 16777619u;
    return h;
}

/* one fusion pass (fuse_neighbours_long) over the blocks of arr[0..n); returns how many are left, in arr[0..) */
static int fuse_array(struct mafAli **arr, int n)
{
    unsigned char *yes;
    struct mafAli *a;
    int i, out = 0, grown = 0;
    if (n < 2) return n;
    yes = (unsigned char *)mz_xmalloc((size_t)n);
#pragma omp parallel for schedule(static, 512) num_threads(MZ_STAGE_THREADS)
    for (i = 0; i < n - 1; ++i) yes[i] = (unsigned char)continues(arr[i], arr[i + 1]);
    a = arr[0];
    for (i = 1; i < n; ++i) {                               /* a: arr[i - 1] as it was asked about, or a block that has grown by fusions */
        struct mafAli *b = arr[i];
        if (grown ? continues(a, b) : yes[i - 1]) {
            append_block(a, b);
            mafAliFree(&b);
            grown = 1;
        } else { arr[out++] = a; a = b; grown = 0; }
    }
    arr[out++] = a;
    free(yes);
    return out;
}

static struct mafAli *project_long(pitem *it, int n, const char *target, struct mafAli **others, mz_blocks *res)
{
    struct mafAli *out = NULL, *out_tail = NULL, *oth_tail = NULL;
    int *S, *S2, ns = 0, i, ncontig = 0, clash = 0;
    struct { unsigned hash; int rep; } *tab;
    int tabcap = 1024;
    if (others) *others = NULL;
    pthread_once(&g_compl_once, compl_fill);
#pragma omp parallel for schedule(static, 256) num_threads(MZ_STAGE_THREADS)
    for (i = 0; i < n; ++i) {
        struct mafAli *a = it[i].a;
        struct mafComp *c, *b;
        a->next = NULL;
        for (c = a->components; c; c = c->next)
            if (strcmp(c->name, target) == 0 || strcmp(c->src, target) == 0) break;
        it[i].src = NULL;
        if (!c) continue;
        if (c != a->components) {                        /* the target's row to the top */
            for (b = a->components; b && b->next != c; b = b->next)
                ;
            if (!b) mz_fatalf("maf_project: cannot happen");
            b->next = c->next;
            c->next = a->components;
            a->components = c;
        }
        if (c->strand == '-') block_revcomp(a);
        it[i].src = c->src; it[i].hash = str_hash(c->src); it[i].start = c->start;
    }
    /* the blocks without a row of the target, in file order; the others numbered back to front, as the stock tool collects them */
    S = (int *)mz_xmalloc(((size_t)n + 1) * sizeof(int)); S2 = (int *)mz_xmalloc(((size_t)n + 1) * sizeof(int));
    if (others) {
        for (i = 0; i < n; ++i)
            if (!it[i].src) { if (oth_tail) oth_tail->next = it[i].a; else *others = it[i].a; oth_tail = it[i].a; }
    } else {                                                /* (a quarter of a guide-tree level's blocks: the unused parts of blocks that lost their reference row) */
#pragma omp parallel for schedule(dynamic, 256) num_threads(MZ_STAGE_THREADS)
        for (i = 0; i < n; ++i) if (!it[i].src) mafAliFree(&it[i].a);
    }
    for (i = n - 1; i >= 0; --i) if (it[i].src) S[ns++] = i;
    /* contigs by number: equal hashes are taken for equal names, then every block's name is compared with its contig's first -- on all
     * threads; a clash (two names, one hash) is settled by comparing names in the table itself */
    tab = mz_xmalloc((size_t)tabcap * sizeof *tab);
    for (i = 0; i < ns; ++i) {
        pitem *p = &it[S[i]];
        int k;
        for (k = 0; k < ncontig && tab[k].hash != p->hash; ++k)
            ;
        if (k == ncontig) {
            if (ncontig == tabcap) { tabcap *= 2; tab = realloc(tab, (size_t)tabcap * sizeof *tab); if (!tab) mz_fatalf("out of memory"); }
            tab[k].hash = p->hash; tab[k].rep = S[i]; ++ncontig;
        }
        p->contig = k;
    }
#pragma omp parallel for schedule(static, 1024) num_threads(MZ_STAGE_THREADS) reduction(| : clash)
    for (i = 0; i < ns; ++i) {
        const pitem *p = &it[S[i]];
        if (p->src != it[tab[p->contig].rep].src && strcmp(p->src, it[tab[p->contig].rep].src) != 0) clash |= 1;
    }
    if (clash) {
        ncontig = 0;
        for (i = 0; i < ns; ++i) {
            pitem *p = &it[S[i]];
            int k;
            for (k = 0; k < ncontig && !(tab[k].hash == p->hash && strcmp(it[tab[k].rep].src, p->src) == 0); ++k)
                ;
            if (k == ncontig) {
                if (ncontig == tabcap) { tabcap *= 2; tab = realloc(tab, (size_t)tabcap * sizeof *tab); if (!tab) mz_fatalf("out of memory"); }
                tab[k].hash = p->hash; tab[k].rep = S[i]; ++ncontig;
            }
            p->contig = k;
        }
    }
    free(tab);
    init_scores70();
    while (ns > 0) {                                        /* one reference contig at a time: the head's; the rest goes on in reverse */
        const int chr = it[S[0]].contig;
        struct mafAli **blocks;
        keyed *arr;
        int m = 0, rest = 0;
        for (i = 0; i < ns; ++i) if (it[S[i]].contig == chr) ++m;
        arr = (keyed *)mz_xmalloc((size_t)m * sizeof *arr);
        for (i = 0, m = 0, rest = ns; i < ns; ++i) {
            if (it[S[i]].contig == chr) { arr[m].start = it[S[i]].start; arr[m].a = it[S[i]].a; ++m; }
            else S2[--rest] = S[i];
        }
        qsort(arr, (size_t)m, sizeof *arr, by_top_start);        /* the same library sort on the same order: ties fall alike */
        blocks = (struct mafAli **)mz_xmalloc((size_t)m * sizeof *blocks);
        for (i = 0; i < m; ++i) blocks[i] = arr[i].a;
        free(arr);
        m = fuse_array(blocks, m);
        m = fuse_array(blocks, m);                          /* (the stock tool's second pass, maf_project.c:690-695) */
#pragma omp parallel for schedule(static, 1024) num_threads(MZ_STAGE_THREADS) if (m > 4096)
        for (i = 0; i < m; ++i) blocks[i]->next = i + 1 < m ? blocks[i + 1] : NULL;
        if (out_tail) out_tail->next = blocks[0]; else out = blocks[0];
        out_tail = blocks[m - 1];
        if (res && !res->p) { res->p = blocks; res->n = res->cap = m; }        /* the result's index: the first contig's array, the others' behind it */
        else { if (res) for (i = 0; i < m; ++i) mz_blocks_push(res, blocks[i]); free(blocks); }
        { const int left = ns - rest; int k; for (k = 0; k < left; ++k) S[k] = S2[rest + k]; ns = left; }
    }
    free(S); free(S2);
    return out;
}

struct mafAli *mz_project_lists(struct mafAli *all, const char *target, struct mafAli **others)
{
    return mzi_project_blocks(all, NULL, target, others);
}

struct mafAli *mzi_project_blocks(struct mafAli *all, mz_blocks *idx, const char *target, struct mafAli **others)
{
    const char *e = getenv("MZ_FUSE_PARALLEL_MIN");         /* (tests: the long form on short lists) */
    const int long_min = e && atoi(e) > 1 ? atoi(e) : FUSE_PARALLEL_MIN;
    struct mafAli *a, *res;
    pitem *it;
    int n = 0, cap = long_min, i;
    if (idx && idx->p) n = idx->n;
    else for (a = all; a && n < long_min; a = a->next) ++n;
    if (n < long_min) { mz_blocks_drop(idx); return project_plain(all, target, others); }
    if (idx && idx->p) {                                    /* the blocks are known: nothing to walk */
        it = (pitem *)mz_xmalloc((size_t)n * sizeof *it);
        for (i = 0; i < n; ++i) it[i].a = idx->p[i];
        mz_blocks_drop(idx);
    } else {
        it = (pitem *)mz_xmalloc((size_t)cap * sizeof *it);
        for (a = all, n = 0; a; a = a->next) {
            if (n == cap) { cap *= 2; it = (pitem *)realloc(it, (size_t)cap * sizeof *it); if (!it) mz_fatalf("out of memory"); }
            it[n++].a = a;
        }
    }
    res = project_long(it, n, target, others, idx);
    free(it);
    return res;
}
