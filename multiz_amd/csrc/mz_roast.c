/* mz_roast.c -- the reference-guided tree driver in ONE process.
 *
 * The stock roast (reference auto_mz.c) parses the species tree (speciesTree.c:37-113) and, at every internal node,
 * glues a pipeline of child programs together with system(): cp / mv / grep on temporary MAF files, maf_project
 * around the inputs, multiz (or multic) for the merge (auto_mz.c:52-118).  Every multiz of that chain is a fresh
 * process -- with the GPU aligners on its PATH, a fresh HIP start-up of 0.1-0.2 s each -- and sibling subtrees, whose
 * merges are independent, run one after the other.
 *
 * Here the same chain runs inside one process on MAF text held in memory: the tree is parsed into explicit nodes with
 * the stock driver's stack discipline and node numbering; a node's "files" are buffers; projection is
 * mz_project_lists(), the merge is the batched driver of mz_multiz.c / mz_multic.c.  Between the steps the blocks go on as
 * LISTS (round 4; MAF text only at the ends: the leaf files and the destination): a node's multiz run ends in block lists
 * (mz_multiz_finish_lists) whose blocks are what a reader of the stock chain's intermediate files would hold (the score with
 * one decimal, a source "x.x" as "x": mz_ali_as_reread), and the next node projects those lists.  The stock chain's line
 * filters (grep -v maf / grep -v eof on the intermediate files) only ever remove header and trailer lines -- unless a
 * species name contains "maf" or "eof", or a row holds an 'f': a leaf file with such a line marks every node above it, and
 * those nodes run on MAF text as round 3 did (MZ_ROAST_TEXT=1 forces that everywhere; multic nodes always do), line filters
 * included.  No file is written and no process started.  Nodes whose children are finished are evaluated together: their multiz runs are
 * prepared (mz_multiz_prepare), then ALL their pending block-pair alignments go to the GPU as one batch per wave
 * (mz_multiz_align) -- the per-tree-level batch of BASELINE config 4 -- and each run is replayed into its node.
 *
 * Output: the destination file, block for block what the stock roast writes.  COMMENT LINES ARE NOT CARRIED: the stock
 * chain's programs echo the '#' lines of their inputs (maf_project and multiz read with mafReadAll(file, 1)), so its
 * destination holds the tools' own comments -- temp-file names with the process id -- and whatever '#' lines the
 * .sing.maf inputs had; the readers here run with verbose = 0 and the destination gets the header and the command line
 * only (tests/test_roast_inprocess.py compares the two outputs with '#' lines left out on both sides).  grep -v eof / grep -v maf of the stock chain act on text lines here
 * as well; like there, a species whose NAME contains "maf" or "eof" would lose its rows to them.
 */
#include "mz_drivers.h"
#include <ctype.h>
#include <omp.h>
#include <unistd.h>
#include <sys/resource.h>

#define ROAST_VERSION 3
#define MAX_NODES 2000

typedef struct { char *p; size_t n, cap; } buf;            /* MAF text; p == NULL: the "file" does not exist */

typedef struct rnode {
    int id;                      /* -1 for a leaf, else the stock driver's node number (creation order) */
    int left, right;             /* children (indices into the node table), -1 for a leaf */
    char **names; int nnames;    /* leaf species below this node, in the stock driver's order */
    buf mz;                      /* the node's result: the file <prefix>MZ<id> of the stock driver -- as text ... */
    struct mafAli *mzl;          /* ... or (mz_list) as the list of its blocks */
    mz_blocks mzi;               /* ... and, where known, the blocks of that list (mz_drivers.h) */
    int mz_list;
    int taint;                   /* a leaf file below has a line the stock chain's line filters would remove: text all the way up */
    int as_lists;                /* this node's multiz run ends in lists */
    struct mafAli *left_l, *right_l;       /* inputs that came as lists (left_is_l / right_is_l) */
    mz_blocks left_i, right_i, i1, i2;     /* the blocks of left_l / right_l, of l1 / l2 (where known) */
    int left_is_l, right_is_l;
    int done;
    /* a multiz step in flight */
    struct mz_mzrun *run;
    struct mafAli *l1, *l2;
    buf left_in, right_in;
    int both_leaves;
} rnode;

static struct {
    rnode nd[MAX_NODES]; int nn;
    const char *ref, *suffix;
    int use_multic, radius, minw, verbose, execute;
    int lists;                   /* blocks go from node to node as lists where the line filters allow it */
} T;

/* ------------------------------------------------------------------------------------------------ text "files" */

static double g_t[8];                                      /* MZ_TIMING: read, project, parse, walk, align, replay + render + line filters, start-up wait */
#define TIMED(slot, stmt) do { const double t_ = mz_now_s(); stmt; g_t[slot] += mz_now_s() - t_; } while (0)

static void buf_free(buf *b) { free(b->p); b->p = NULL; b->n = 0; b->cap = 0; }
static void buf_append(buf *dst, const char *s, size_t n)
{
    if (dst->n + n + 1 > dst->cap) {
        dst->cap = (dst->n + n + 1) * 2 + 4096;
        dst->p = (char *)realloc(dst->p, dst->cap);
        if (!dst->p) mz_fatalf("out of memory");
    }
    memcpy(dst->p + dst->n, s, n);
    dst->n += n;
    dst->p[dst->n] = 0;
}
static void buf_puts(buf *dst, const char *s) { buf_append(dst, s, strlen(s)); }

/* grep -v <word> src >> dst : the lines of src that do not contain word; a missing file appends nothing */
static void append_lines_without(buf *dst, const buf *src, const char *word)
{
    const char *p, *end;
    const size_t wl = strlen(word);
    if (!src->p) return;
    for (p = src->p, end = src->p + src->n; p < end; ) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const size_t len = nl ? (size_t)(nl - p) + 1 : (size_t)(end - p);
        int hit = 0;
        size_t i;
        for (i = 0; !hit && i + wl <= len; ++i) hit = memcmp(p + i, word, wl) == 0;
        if (!hit) buf_append(dst, p, len);
        p += len;
    }
}

/* grep -v <word> of a text, straight into a file (the destination of a guide-tree run is hundreds of megabytes: searched for the
 * word as a whole, written in the stretches between the lines that hold it, copied nowhere) */
static void write_lines_without(FILE *f, const char *p, size_t n, const char *word)
{
    const size_t wl = strlen(word);
    const char *end = p + n;
    while (p < end) {
        const char *hit = (const char *)memmem(p, (size_t)(end - p), word, wl), *ls, *le;
        if (!hit) { fwrite(p, 1, (size_t)(end - p), f); return; }
        ls = (const char *)memrchr(p, '\n', (size_t)(hit - p));
        ls = ls ? ls + 1 : p;
        le = (const char *)memchr(hit, '\n', (size_t)(end - hit));
        le = le ? le + 1 : end;
        if (ls > p) fwrite(p, 1, (size_t)(ls - p), f);
        p = le;
    }
}

static buf read_file(const char *path)
{
    buf b = { NULL, 0, 0 };
    FILE *f = fopen(path, "r");
    long n;
    if (!f) mz_fatalf("Cannot open %s.", path);
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    b.p = (char *)mz_xmalloc((size_t)n + 1);
    if (fread(b.p, 1, (size_t)n, f) != (size_t)n) mz_fatalf("Cannot read %s.", path);
    b.p[n] = 0; b.n = (size_t)n; b.cap = (size_t)n + 1;
    fclose(f);
    return b;
}

/* The leaf files are read ahead, side by side, while the GPU starts up (leaves_ahead(): a file that cannot be opened is left for
 * leaf_file() to fail on, at its place in the stock driver's order); each with the answer of leaf_taints(). */
static struct { const char *species; buf b; int taint; } g_ahead[MAX_NODES];
static int g_nahead;
static int leaf_taints(const buf *b, const char *species);

static buf leaf_file(const char *species, int *taint)
{
    char path[1200];
    int k;
    snprintf(path, sizeof path, "%s.%s%s", T.ref, species, T.suffix);
    if (T.verbose) printf("read %s\n", path);
    for (k = 0; k < g_nahead; ++k)
        if (g_ahead[k].b.p && strcmp(g_ahead[k].species, species) == 0) {
            buf b = g_ahead[k].b;
            memset(&g_ahead[k].b, 0, sizeof g_ahead[k].b);
            *taint = g_ahead[k].taint;
            return b;
        }
    { buf b; TIMED(0, b = read_file(path)); *taint = leaf_taints(&b, species); return b; }
}

static void free_list(struct mafAli *l) { while (l) { struct mafAli *a = mz_pop_first(&l); mafAliFree(&a); } }

/* Would `grep -v maf` / `grep -v eof` remove a block line somewhere above this leaf?  A line of the file itself that contains one of
 * the words (a species or contig name), or a row with an 'f' in it: rows are cut, squeezed and reverse-complemented on the way up,
 * and both words need an 'f' ('f' is no nucleotide code and nothing complements to it). */
static int leaf_taints(const buf *b, const char *species)
{
    const char *p = b->p, *end = b->p + b->n, *hook = getenv("MZ_ROAST_TAINT");
    if (hook && strcmp(hook, species) == 0) return 1;       /* (test hook: the nodes above this leaf run on text, fed by lists from below) */
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const size_t len = nl ? (size_t)(nl - p) : (size_t)(end - p);
        if (len && *p != '#') {
            if (memmem(p, len, "maf", 3) || memmem(p, len, "eof", 3)) return 1;
            if (*p == 's') {
                const char *t = p + len;
                while (t > p && t[-1] != ' ' && t[-1] != '\t') --t;
                if (memchr(t, 'f', (size_t)(p + len - t))) return 1;
            }
        }
        p += len + 1;
    }
    return 0;
}

/* `more` (whose last block is more_tail) behind the list whose last block is *tail (NULL: the list is empty) */
static void leaves_ahead(void)
{
    int i, k;
    for (i = 0, g_nahead = 0; i < T.nn; ++i)
        if (T.nd[i].id < 0 && strcmp(T.nd[i].names[0], T.ref) != 0) { g_ahead[g_nahead].species = T.nd[i].names[0]; memset(&g_ahead[g_nahead].b, 0, sizeof(buf)); ++g_nahead; }
#pragma omp parallel for schedule(dynamic, 1) num_threads(MZ_STAGE_THREADS) if (g_nahead > 1)
    for (k = 0; k < g_nahead; ++k) {
        char path[1200];
        FILE *f;
        long n;
        buf b = { NULL, 0, 0 };
        snprintf(path, sizeof path, "%s.%s%s", T.ref, g_ahead[k].species, T.suffix);
        f = fopen(path, "r");
        if (!f) continue;
        if (fseek(f, 0, SEEK_END) == 0 && (n = ftell(f)) >= 0 && fseek(f, 0, SEEK_SET) == 0 && (b.p = (char *)malloc((size_t)n + 1)) != NULL) {
            if (fread(b.p, 1, (size_t)n, f) == (size_t)n) {
                b.p[n] = 0; b.n = (size_t)n; b.cap = (size_t)n + 1;
                g_ahead[k].taint = leaf_taints(&b, g_ahead[k].species);
                g_ahead[k].b = b;
            } else free(b.p);
        }
        fclose(f);
    }
}

static void list_append(struct mafAli **head, struct mafAli **tail, struct mafAli *more, struct mafAli *more_tail)
{
    if (!more) return;
    if (*tail) (*tail)->next = more; else *head = more;
    *tail = more_tail;
}

/* a list as the text of the file it stands for (a node that runs on text above one that ran on lists) */
static buf list_to_text(struct mafAli *list)
{
    buf out = { NULL, 0, 0 };
    struct mafAli *a;
    FILE *m = open_memstream(&out.p, &out.n);
    fprintf(m, "##maf version=1 scoring=multiz.%d\n", ROAST_VERSION);
    for (a = list; a; a = a->next) mafWrite(m, a);
    fclose(m);
    out.cap = out.n + 1;
    free_list(list);
    return out;
}

/* maf_project <file> REF <others> > out : header, projected blocks, trailer (reference maf_project.c:592-598,777) */
/* (dst != NULL: the run's last projection -- `grep -v eof` of the text goes to dst, and nothing is returned) */
static buf project_list_text(struct mafAli *list, mz_blocks *known, const char *what, FILE *dst)
{
    const int last = dst != NULL;
    mz_blocks idx = { NULL, 0, 0 };
    buf out = { NULL, 0, 0 };
    struct mafAli *a, **blocks;
    char head[1400];
    int n = 0, i, pieces, k;
    buf *part;
    if (known) { idx = *known; memset(known, 0, sizeof *known); }
    { const double t_ = mz_now_s(); list = mzi_project_blocks(list, &idx, T.ref, NULL); g_t[1] += mz_now_s() - t_; }
    { const double t_ = mz_now_s();
    /* the blocks rendered in pieces side by side (the destination of a guide-tree run is hundreds of thousands of blocks) */
    if (idx.p) { blocks = idx.p; n = idx.n; }
    else {
        for (a = list; a; a = a->next) ++n;
        blocks = (struct mafAli **)mz_xmalloc(((size_t)n + 1) * sizeof *blocks);
        for (a = list, n = 0; a; a = a->next) blocks[n++] = a;
    }
    pieces = n < 4096 ? 1 : dst ? 8 * MZ_STAGE_THREADS : MZ_STAGE_THREADS;
    part = (buf *)mz_xmalloc((size_t)pieces * sizeof *part);
    snprintf(head, sizeof head, "##maf version=1 scoring=maf_project.v12\n# maf_project.v12 %s %s (in process)\n", what, T.ref);
    if (dst) {
        /* the destination: many pieces, each written -- in order -- as soon as it is rendered, by the first thread of the team, while the
         * others render the pieces behind it (a team of one renders everything first) */
        int next = 0, written = 0, *ready = (int *)mz_xmalloc((size_t)pieces * sizeof(int));
        memset(ready, 0, (size_t)pieces * sizeof(int));
        write_lines_without(dst, head, strlen(head), "eof");
#pragma omp parallel num_threads(MZ_STAGE_THREADS) if (pieces > 1)
        {
            if (omp_get_thread_num() == 0 && omp_get_num_threads() > 1) {
                for (; written < pieces; ++written) {
                    while (!__atomic_load_n(&ready[written], __ATOMIC_ACQUIRE)) usleep(50);
                    write_lines_without(dst, part[written].p, part[written].n, "eof");
                    free(part[written].p);
                }
            } else
                for (;;) {
                    const int kk = __atomic_fetch_add(&next, 1, __ATOMIC_RELAXED);
                    int lo, hi, j;
                    FILE *m;
                    if (kk >= pieces) break;
                    lo = (int)((long long)n * kk / pieces); hi = (int)((long long)n * (kk + 1) / pieces);
                    m = open_memstream(&part[kk].p, &part[kk].n);
                    for (j = lo; j < hi; ++j) mafWrite(m, blocks[j]);
                    fclose(m);
                    __atomic_store_n(&ready[kk], 1, __ATOMIC_RELEASE);
                }
        }
        for (; written < pieces; ++written) { write_lines_without(dst, part[written].p, part[written].n, "eof"); free(part[written].p); }
        free(ready);
    } else {
#pragma omp parallel for schedule(static, 1) num_threads(MZ_STAGE_THREADS) if (pieces > 1)
        for (k = 0; k < pieces; ++k) {
            const int lo = (int)((long long)n * k / pieces), hi = (int)((long long)n * (k + 1) / pieces);
            FILE *m = open_memstream(&part[k].p, &part[k].n);
            int j;
            for (j = lo; j < hi; ++j) mafWrite(m, blocks[j]);
            fclose(m);
        }
        buf_puts(&out, head);
        for (k = 0; k < pieces; ++k) { buf_append(&out, part[k].p, part[k].n); free(part[k].p); }
        buf_puts(&out, "##eof maf\n");
    }
    free(part); free(blocks);
    if (!last) free_list(list);                          /* (the destination's blocks: the process ends with them -- 13 M frees of a guide-tree run) */
    g_t[5] += mz_now_s() - t_; }
    (void)i;
    return out;
}
static buf project_text(const buf *in, const char *what)
{
    struct mafAli *list;
    if (!in->p) mz_fatalf("Cannot open %s.", what);
    TIMED(2, list = mz_maf_read_mem(in->p, in->n, what));
    return project_list_text(list, NULL, what, NULL);
}

/* ------------------------------------------------------------------------------------------------ the tree */

static int new_node(void)
{
    if (T.nn >= MAX_NODES) mz_fatalf("parse_tree: stack overflow");
    memset(&T.nd[T.nn], 0, sizeof(rnode));
    T.nd[T.nn].id = T.nd[T.nn].left = T.nd[T.nn].right = -1;
    return T.nn++;
}

/* the stock parser's stack machine (speciesTree.c:37-113): a name pushes a leaf, ')' closes a group, and whenever two
 * finished subtrees lie side by side on the stack they become an internal node, numbered in creation order */
static int parse_tree(const char *spec)
{
    int stack[1000], top = -1, next_id = 0;        /* stack entries: node index, or -1 for '(' */
    const char *q;
    for (q = spec; *q; ++q) {
        if (*q == '(') {
            if (++top >= 1000) mz_fatalf("parse_tree: stack overflow");
            stack[top] = -1;
        } else if (*q == ')') {
            if (top < 1 || stack[top] < 0 || stack[top - 1] != -1)
                mz_fatalf("parse error: %.*s", (int)(q - spec) + 1, spec);
            stack[top - 1] = stack[top];
            --top;
        } else if (isalpha((unsigned char)*q)) {
            const char *s = q;
            int k;
            if (++top >= 1000) mz_fatalf("parse_tree: stack overflow");
            while (isalpha((unsigned char)*q) || isdigit((unsigned char)*q) || *q == '_' || *q == '.') ++q;
            k = new_node();
            T.nd[k].names = (char **)mz_xmalloc(sizeof(char *));
            T.nd[k].names[0] = (char *)mz_xmalloc((size_t)(q - s) + 1);
            memcpy(T.nd[k].names[0], s, (size_t)(q - s)); T.nd[k].names[0][q - s] = 0;
            T.nd[k].nnames = 1;
            T.nd[k].done = 1;
            stack[top] = k;
            --q;
        } else if (*q != ' ')
            mz_fatalf("improper character in tree specification: %c", *q);
        if (top > 0 && stack[top - 1] >= 0 && stack[top] >= 0) {
            const int x = stack[top - 1], y = stack[top], k = new_node();
            rnode *nd = &T.nd[k];
            nd->left = x; nd->right = y; nd->id = next_id++;
            nd->nnames = T.nd[x].nnames + T.nd[y].nnames;
            nd->names = (char **)mz_xmalloc((size_t)nd->nnames * sizeof(char *));
            memcpy(nd->names, T.nd[x].names, (size_t)T.nd[x].nnames * sizeof(char *));
            memcpy(nd->names + T.nd[x].nnames, T.nd[y].names, (size_t)T.nd[y].nnames * sizeof(char *));
            stack[--top] = k;
        }
    }
    if (top > 0) mz_fatalf("tree specification contains too many '('");
    if (top != 0 || stack[0] < 0) mz_fatalf("tree specification is improper");
    return stack[0];
}

static int has_ref(const rnode *n)
{
    int i;
    for (i = 0; i < n->nnames; ++i) if (strcmp(n->names[i], T.ref) == 0) return 1;
    return 0;
}
static int is_single(const rnode *n, const char *name) { return n->nnames == 1 && (!name || strcmp(n->names[0], name) == 0); }

/* ------------------------------------------------------------------------------------------------ one node
 * (speciesTree.c:78-90 around mz_merge(), auto_mz.c:52-118).  begin_node() does everything up to the aligner; when
 * that is multiz the run is left prepared (nd->run) for the shared batch, and end_node() finishes the node. */

/* begin_node() in three steps, so that a round's nodes go through each step side by side:
 *   node_inputs()   (serial: buffers change hands, leaf files are read, the stock driver's messages keep their order)
 *   node_parse()    (a task per node and side: MAF text -> blocks -> projected list)
 *   node_prepare()  (a task per node: the list walk of mz_multiz_prepare -- its record has its own arena) */
static void node_inputs(rnode *nd)
{
    rnode *x = &T.nd[nd->left], *y = &T.nd[nd->right];
    buf left = { NULL, 0, 0 }, right = { NULL, 0, 0 };
    struct mafAli *left_l = NULL, *right_l = NULL;
    mz_blocks left_i = { NULL, 0, 0 }, right_i = { NULL, 0, 0 };
    int left_is_l = 0, right_is_l = 0;

    /* mv MZ<i> left.maf<id>: the child's result, text or list */
    if (x->id >= 0) { left = x->mz; left_l = x->mzl; left_i = x->mzi; left_is_l = x->mz_list; memset(&x->mz, 0, sizeof x->mz); memset(&x->mzi, 0, sizeof x->mzi); x->mzl = NULL; x->mz_list = 0; }
    if (y->id >= 0) { right = y->mz; right_l = y->mzl; right_i = y->mzi; right_is_l = y->mz_list; memset(&y->mz, 0, sizeof y->mz); memset(&y->mzi, 0, sizeof y->mzi); y->mzl = NULL; y->mz_list = 0; }
    buf_free(&nd->mz);
    nd->mzl = NULL; nd->mz_list = 0; mz_blocks_drop(&nd->mzi);
    memset(&nd->left_i, 0, sizeof nd->left_i); memset(&nd->right_i, 0, sizeof nd->right_i); memset(&nd->i1, 0, sizeof nd->i1); memset(&nd->i2, 0, sizeof nd->i2);
    nd->run = NULL; nd->l1 = nd->l2 = NULL; nd->both_leaves = 0; nd->as_lists = 0;
    nd->left_l = nd->right_l = NULL; nd->left_is_l = nd->right_is_l = 0;
    memset(&nd->left_in, 0, sizeof nd->left_in); memset(&nd->right_in, 0, sizeof nd->right_in);

    if (T.verbose) printf("node %d: %d + %d species\n", nd->id, x->nnames, y->nnames);
    if (is_single(x, T.ref) || is_single(y, T.ref)) {
        /* the reference itself on one side: the other side's blocks are already topped by it */
        const int ref_left = is_single(x, T.ref);
        rnode *other = ref_left ? y : x;
        buf *ob = ref_left ? &right : &left;
        if (other->nnames == 1) {
            buf f = leaf_file(other->names[0], &nd->taint);
            char head[64];
            snprintf(head, sizeof head, "##maf version=1 scoring=multiz.%d\n", ROAST_VERSION); buf_puts(&nd->mz, head);
            append_lines_without(&nd->mz, &f, "eof"); buf_free(&f);
        } else if (ref_left ? right_is_l : left_is_l) {
            nd->mzl = ref_left ? right_l : left_l; nd->mz_list = 1;       /* (grep -v eof finds the trailer only) */
            nd->mzi = ref_left ? right_i : left_i;
            if (ref_left) { right_l = NULL; memset(&right_i, 0, sizeof right_i); } else { left_l = NULL; memset(&left_i, 0, sizeof left_i); }
            nd->taint = other->taint;
        } else {
            char head[64];
            snprintf(head, sizeof head, "##maf version=1 scoring=multiz.%d\n", ROAST_VERSION); buf_puts(&nd->mz, head);
            append_lines_without(&nd->mz, ob, "eof");
            nd->taint = other->taint;
        }
        buf_free(&left); buf_free(&right);               /* rm -f: the closing greps of the parser find nothing */
        free_list(left_l); free_list(right_l);
        mz_blocks_drop(&left_i); mz_blocks_drop(&right_i);
        nd->done = 1;
        return;
    }
    if (x->nnames == 1) { buf_free(&left); left = leaf_file(x->names[0], &x->taint); }
    if (y->nnames == 1) { buf_free(&right); right = leaf_file(y->names[0], &y->taint); }
    if (!left.p && !left_is_l) mz_fatalf("Cannot open %s.", "left.maf");
    if (!right.p && !right_is_l) mz_fatalf("Cannot open %s.", "right.maf");
    nd->taint = x->taint || y->taint;
    nd->as_lists = T.lists && !nd->taint;
    if (!nd->as_lists) {                                 /* a node on text takes text */
        if (left_is_l) { left = list_to_text(left_l); left_l = NULL; left_is_l = 0; }
        if (right_is_l) { right = list_to_text(right_l); right_l = NULL; right_is_l = 0; }
        mz_blocks_drop(&left_i); mz_blocks_drop(&right_i);
    }
    nd->left_in = left; nd->right_in = right;
    nd->left_l = left_l; nd->right_l = right_l; nd->left_is_l = left_is_l; nd->right_is_l = right_is_l;
    nd->left_i = left_i; nd->right_i = right_i;
}

/* maf_project left REF > U1; mv U1 left (and the same on the right), then the aligner reads both: the projected
 * blocks go on as lists -- what the stock chain's write-and-read-again would re-derive (sizes, text lengths, scores
 * with one decimal: printing a score twice gives what printing it once gives) is already in them */
static void node_parse(rnode *nd, int side)
{
    buf *in = side ? &nd->right_in : &nd->left_in;
    struct mafAli *l;
    mz_blocks idx = { NULL, 0, 0 };
    if (side ? nd->right_is_l : nd->left_is_l) {
        idx = side ? nd->right_i : nd->left_i;
        memset(side ? &nd->right_i : &nd->left_i, 0, sizeof idx);
        l = mzi_project_blocks(side ? nd->right_l : nd->left_l, &idx, T.ref, NULL);
        if (side) nd->right_l = NULL; else nd->left_l = NULL;
    } else l = mzi_project_blocks(mz_maf_read_mem(in->p, in->n, side ? "right.maf" : "left.maf"), &idx, T.ref, NULL);
    if (side) { nd->l2 = l; nd->i2 = idx; } else { nd->l1 = l; nd->i1 = idx; }
    buf_free(in);
}

static void node_prepare(rnode *nd)
{
    rnode *x = &T.nd[nd->left], *y = &T.nd[nd->right];
    const int l = has_ref(x), r = has_ref(y);
    if (!l && !r) nd->both_leaves = x->nnames == 1 && y->nnames == 1;
    else if (r) { struct mafAli *t = nd->l1; const mz_blocks ti = nd->i1; nd->l1 = nd->l2; nd->l2 = t; nd->i1 = nd->i2; nd->i2 = ti; }
    if (!T.use_multic) {
        nd->run = mzi_multiz_prepare_blocks(&nd->l1, &nd->l2, &nd->i1, &nd->i2, (l || r) ? 1 : 0, T.radius, T.minw, 1, 1);
        if (nd->as_lists) mz_multiz_keep_blocks(nd->run);
    } else { mz_blocks_drop(&nd->i1); mz_blocks_drop(&nd->i2); }
}

/* the blocks of a list the stock driver would print (contigs only one side has), as their reader would hold them; the rest freed */
static struct mafAli *printed_blocks(struct mafAli *l, struct mafAli **last, mz_blocks *idx)
{
    struct mafAli *keep = NULL, **tail = &keep;
    *last = NULL;
    while (l) {
        struct mafAli *a = mz_pop_first(&l);
        if (row2 == 0 || a->components->next) { mz_ali_as_reread(a); *tail = a; tail = &a->next; *last = a; mz_blocks_push(idx, a); }
        else mafAliFree(&a);
    }
    return keep;
}

static void end_node_lists(rnode *nd)
{
    rnode *x = &T.nd[nd->left], *y = &T.nd[nd->right];
    struct mafAli *out = NULL, *u1 = NULL, *u2 = NULL, *tails[3], *p, *pt;
    mz_blocks idx[3];
    int k;
    mzi_multiz_finish_tails(nd->run, &out, &u1, &u2, tails, idx);   /* stdout, U1, U2 of `multiz M=.. left right v U1 U2` */
    nd->run = NULL;
    p = printed_blocks(nd->l1, &pt, &idx[1]); list_append(&u1, &tails[1], p, pt);      /* contigs only one side has */
    p = printed_blocks(nd->l2, &pt, &idx[2]); list_append(&u2, &tails[2], p, pt);
    nd->l1 = nd->l2 = NULL;
    if (nd->both_leaves || x->id >= 0 || y->id >= 0) {   /* >> MZ<id>: the unused parts follow (the line filters find nothing in them) */
        list_append(&out, &tails[0], u1, tails[1]);
        list_append(&out, &tails[0], u2, tails[2]);
        for (k = 0; k < idx[1].n; ++k) mz_blocks_push(&idx[0], idx[1].p[k]);
        for (k = 0; k < idx[2].n; ++k) mz_blocks_push(&idx[0], idx[2].p[k]);
    } else { free_list(u1); free_list(u2); }
    mz_blocks_drop(&idx[1]); mz_blocks_drop(&idx[2]);
    nd->mzi = idx[0];
    nd->mzl = out; nd->mz_list = 1;
    nd->done = 1;
}

static void end_node(rnode *nd)
{
    rnode *x = &T.nd[nd->left], *y = &T.nd[nd->right];
    const int v = (has_ref(x) || has_ref(y)) ? 1 : 0;
    buf out = { NULL, 0, 0 }, u1 = { NULL, 0, 0 }, u2 = { NULL, 0, 0 };
    FILE *mo, *m1, *m2;
    struct mafAli *a;

    if (nd->done) return;
    if (nd->as_lists) { end_node_lists(nd); return; }
    { char head[64]; snprintf(head, sizeof head, "##maf version=1 scoring=multiz.%d\n", ROAST_VERSION); buf_puts(&nd->mz, head); }
    mo = open_memstream(&out.p, &out.n); m1 = open_memstream(&u1.p, &u1.n); m2 = open_memstream(&u2.p, &u2.n);
    /* what `multiz M=.. left right v U1 U2` puts on stdout and into U1 / U2 (multiz.c:251-291) */
    fprintf(mo, "##maf version=1 scoring=%s\n# %s (in process, node %d)\n", T.use_multic ? "multih.c" : "multiz",
            T.use_multic ? "multic.v12.1" : "multiz.v11.2", nd->id);
    if (T.use_multic) mz_multic_lists(&nd->l1, &nd->l2, v, T.radius, T.minw, 0, mo, m1, m2);
    else mz_multiz_finish(nd->run, mo, m1, m2);
    nd->run = NULL;
    for (a = nd->l1; a; a = a->next) if (row2 == 0 || a->components->next) mafWrite(m1, a);   /* contigs only one side has */
    for (a = nd->l2; a; a = a->next) if (row2 == 0 || a->components->next) mafWrite(m2, a);
    free_list(nd->l1); free_list(nd->l2); nd->l1 = nd->l2 = NULL;
    mz_blocks_drop(&nd->i1); mz_blocks_drop(&nd->i2);
    fprintf(mo, "##eof maf\n");
    fclose(mo); fclose(m1); fclose(m2);

    buf_append(&nd->mz, out.p, out.n);                   /* >> MZ<id> */
    buf_free(&out);
    if (nd->both_leaves) {
        /* two leaves, neither the reference: the unused parts follow at once (grep -v -h eof U1 U2), and the parser's
         * closing greps do not run (both children are leaves) */
        append_lines_without(&nd->mz, &u1, "eof");
        append_lines_without(&nd->mz, &u2, "eof");
    } else if (x->id >= 0 || y->id >= 0) {
        /* mv U1 left.maf; mv U2 right.maf; then the parser: grep -v maf left.maf<id> right.maf<id> >> MZ<id> */
        append_lines_without(&nd->mz, &u1, "maf");
        append_lines_without(&nd->mz, &u2, "maf");
    }
    buf_free(&u1); buf_free(&u2);
    buf_free(&nd->left_in); buf_free(&nd->right_in);
    nd->done = 1;
}

/* ------------------------------------------------------------------------------------------------ command line */

int mz_roast_main(int argc, char **argv)
{
    static char cmd[64];
    const char *usage =
        "args: [+-] [R=?] [M=?] [P=?] [T=?] [X=?] [C=?] E=reference-species species-guid-tree maf-source destination\n"
        "\tR(30) dynamic programming radius.\n"
        "\tM(1) minimum block length of output.\n"
        "\tP(multiz) multiz: single coverage for reference row multic: no requirement on single coverage.\n"
        "\tT(/tmp) accepted for compatibility: this driver writes no temporary files\n"
        "\tX(0) utilize maf files with different suffix from differnt post processing.\n\t\t0: .sing.maf from single coverage pairwise alignment\n\t\t1: .toast.maf from full size toast\n\t\t2: .toast2.maf from reduced size toast\n";
    const char *destination;
    char *cmdline;
    size_t na = 64;
    int i, root, rounds = 0, batches = 0;
    double t0 = mz_now_s(), r_t[8];
    FILE *dst;

    snprintf(cmd, sizeof cmd, "roast.v%d", ROAST_VERSION);
    argv0 = cmd;
    if (argc == 4 && strcmp(argv[1], "--project") == 0) {
        /* test hook: `--project file.maf REF` prints what `maf_project file.maf REF others` prints (no comment echo) */
        buf in = read_file(argv[2]), out;
        memset(&T, 0, sizeof T);
        T.ref = argv[3];
        out = project_text(&in, argv[2]);
        fwrite(out.p, 1, out.n, stdout);
        return 0;
    }
    if (argc < 5) mz_fatalf("roast -- reference guided multiple alignment.\n%s", usage);
    destination = argv[argc - 1];
    for (i = 1; i < argc; ++i) na += strlen(argv[i]) + 1;
    cmdline = (char *)mz_xmalloc(na + 8);
    sprintf(cmdline, "# %s", cmd);
    for (i = 1; i < argc; ++i) { strcat(cmdline, " "); strcat(cmdline, argv[i]); }

    memset(&T, 0, sizeof T);
    T.suffix = ".sing.maf"; T.radius = 30; T.minw = 1; T.execute = 1;
    if (argc > 1 && strcmp(argv[1], "-") == 0) { T.execute = 0; T.verbose = 1; --argc; ++argv; }
    else if (argc > 1 && strcmp(argv[1], "+") == 0) { T.verbose = 1; --argc; ++argv; }
    while (argc > 1 && argv[1][0] && strchr("RMEPXCT", argv[1][0]) && argv[1][1] == '=') {
        const char *val = argv[1] + 2;
        switch (argv[1][0]) {
        case 'E': T.ref = val; break;
        case 'P':
            if (strstr("multic", val)) T.use_multic = 1;
            else if (!strstr("multiz", val)) mz_fatalf("the optional multiple aligner can be multiz or multic only.\n%s", usage);
            break;
        case 'T': break;
        case 'X': {
            const int X = atoi(val);
            if (X == 1) T.suffix = ".toast.maf";
            else if (X == 2) T.suffix = ".toast2.maf";
            else if (X != 0) mz_fatalf("Parameter X can only be 0, 1, 2, 3.\n%s", usage);
            break; }
        case 'C': { const int c = atoi(val); if (c < 0 || c > 100) mz_fatalf("%s\n", usage); break; }
        case 'R': T.radius = atoi(val); if (T.radius < 0) mz_fatalf("radius cannot be negative"); break;
        case 'M': T.minw = atoi(val); if (T.minw < 0) mz_fatalf("MIN_OUTPUT_WID cannot be negative"); break;
        }
        --argc; ++argv;
    }
    if (!T.ref) mz_fatalf("fatal -- reference is not specified.\n%s", usage);
    if (argc < 3) mz_fatalf("roast -- reference guided multiple alignment.\n%s", usage);

    { const char *e = getenv("MZ_ROAST_TEXT"); T.lists = !T.use_multic && !(e && atoi(e) != 0); }
    mz_tune_malloc();
    init_scores70();
    mz_warm_start();                                     /* the GPU starts up while the inputs are read */
    root = parse_tree(argv[1]);
    if (!T.execute) {                                     /* "-": show the plan */
        for (i = 0; i < T.nn; ++i) if (T.nd[i].id >= 0)
            printf("node %d: children %d %d, %d species\n", T.nd[i].id, T.nd[T.nd[i].left].id, T.nd[T.nd[i].right].id, T.nd[i].nnames);
        return 0;
    }

    TIMED(0, leaves_ahead());
    /* rounds: every node whose children are finished is begun; the multiz runs of the round share their GPU
     * batches; then the nodes are finished in the stock driver's order */
    for (;;) {
        struct mz_mzrun *runs[MAX_NODES];
        int ready[MAX_NODES], nready = 0, nruns = 0;
        for (i = 0; i < T.nn; ++i) {
            rnode *nd = &T.nd[i];
            if (nd->id >= 0 && !nd->done && T.nd[nd->left].done && T.nd[nd->right].done) ready[nready++] = i;
        }
        if (nready == 0) break;
        memcpy(r_t, g_t, sizeof r_t);
        {
            int todo[MAX_NODES], ntodo = 0, k;
            { const double t_ = mz_now_s();
              for (i = 0; i < nready; ++i) {
                  node_inputs(&T.nd[ready[i]]);
                  if (!T.nd[ready[i]].done) todo[ntodo++] = ready[i];
              }
              g_t[7] += mz_now_s() - t_; }
            /* the round's inputs parsed and projected, a task per node and side; then the list walks, a task per node */
            { const double t_ = mz_now_s();
              /* (a round of few nodes: one list after the other -- a long list's projection runs on all threads by itself, mz_project.c) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(MZ_STAGE_THREADS) if (ntodo > 4)
              for (k = 0; k < 2 * ntodo; ++k) node_parse(&T.nd[todo[k >> 1]], k & 1);
              g_t[2] += mz_now_s() - t_; }
            { const double t_ = mz_now_s();
              /* (a round of few nodes: one after the other, each walk cut into pieces that run side by side -- mz_multiz.c, walk_contig) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(MZ_STAGE_THREADS) if (ntodo > 4)
              for (k = 0; k < ntodo; ++k) node_prepare(&T.nd[todo[k]]);
              g_t[3] += mz_now_s() - t_; }
            for (k = 0; k < ntodo; ++k) if (T.nd[todo[k]].run) runs[nruns++] = T.nd[todo[k]].run;
        }
        if (nruns) {
            /* what is left of the GPU's start-up (mz_warm_start() above: runtime, context, code object -- ~0.2 s beside a parser that
             * contends for the same page-table lock) is waited for HERE, under its own name, not inside the first batch's time */
            if (!batches) TIMED(6, mz_warm_wait());
            TIMED(4, mz_multiz_align(runs, nruns)); ++batches;
        }
        /* replay and rendering of every node of the round, side by side (multic nodes: one after the other -- its driver
         * keeps state of its own) */
        { const double t_ = mz_now_s();
          if (T.use_multic) { for (i = 0; i < nready; ++i) end_node(&T.nd[ready[i]]); }
          else {
#pragma omp parallel for schedule(dynamic, 1) num_threads(MZ_STAGE_THREADS) if (nready > 1)
              for (i = 0; i < nready; ++i) end_node(&T.nd[ready[i]]);
          }
          g_t[5] += mz_now_s() - t_; }
        if (mzi_timing())
            fprintf(stderr, "mz_roast: round %d, %d nodes: inputs taken over %.3f s (leaves read %.3f), parsed + projected %.3f, list walks %.3f, alignment batch %.3f, replay + rendering %.3f\n",
                    rounds + 1, nready, g_t[7] - r_t[7], g_t[0] - r_t[0], g_t[2] - r_t[2], g_t[3] - r_t[3], g_t[4] - r_t[4], g_t[5] - r_t[5]);
        ++rounds;
    }
    if (T.nd[root].id < 0) mz_fatalf("tree specification is improper");

    dst = fopen(destination, "w");
    if (!dst) mz_fatalf("Cannot open %s.", destination);
    fprintf(dst, "##maf version=1 scoring=%s.%d\n%s\n", cmd, ROAST_VERSION, cmdline);
    if (T.nd[root].mz_list) project_list_text(T.nd[root].mzl, &T.nd[root].mzi, "MZ", dst);
    else {
        buf fin = project_text(&T.nd[root].mz, "MZ");
        write_lines_without(dst, fin.p, fin.n, "eof");
        buf_free(&fin);
    }
    fprintf(dst, "##eof maf\n");
    fclose(dst);
    if (mzi_timing()) {
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "mz_roast: CPU time %.2f s in the program + %.2f s in the kernel (%ld page faults), %ld MB at most\n",
                ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec, ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec, ru.ru_minflt, ru.ru_maxrss >> 10);
    }
    if (mzi_timing())
        fprintf(stderr, "mz_roast: %d internal nodes in %d rounds, %d shared alignment batches, %.3f s (reading leaves %.3f, final projection %.3f, parsing + projecting the inputs %.3f, "
                "list walks %.3f, waiting for the GPU's start-up %.3f, alignment batches with their host stages %.3f, replay + rendering + line filters %.3f)\n",
                T.nn ? T.nd[root].id + 1 : 0, rounds, batches, mz_now_s() - t0, g_t[0], g_t[1], g_t[2], g_t[3], g_t[6], g_t[4], g_t[5]);
    free(cmdline);
    return 0;
}
