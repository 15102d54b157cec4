/* mz_mafio.c -- the MAF reader of the batched drivers (reference maf.c:10-36,89-225: comment echo,
 * amplifier= / copy= tags, the reader's checks and messages) and the list helpers they share
 * (retrieve_first / seperate_cp_wk, reference multi_util.c:805-843). */
#include "mz_drivers.h"

void *mz_xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)n);
    return p;
}
char *mz_xstrdup(const char *s) { char *p = (char *)mz_xmalloc(strlen(s) + 1); return strcpy(p, s); }

/* These programs make millions of small allocations from up to 32 threads; letting the heaps grow (and shrink) in
 * small steps cost a fifth of a run in brk/mprotect calls and page-table locks. */
void mz_tune_malloc(void)
{
    mallopt(M_TOP_PAD, 256 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
}

/* ------------------------------------------------------------------------------------------------ MAF reader */

typedef struct { FILE *fp; const char *name; int line_nbr, verbose; char *line; size_t cap; FILE *echo; char *tmp; size_t tmpcap; } maf_in;

/* one line, newline kept; -1 at end of file */
static long in_line(maf_in *in)
{
    const ssize_t n = getline(&in->line, &in->cap, in->fp);
    if (n < 0) {
        if (!in->line) { in->line = (char *)mz_xmalloc(16); in->cap = 16; }
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
        if (in->verbose && strstr(in->line, "eof") == NULL) fputs(in->line, in->echo ? in->echo : stdout);
    }
    return n;
}

/* species and contig parts of "species.contig" (reference multi_util.c:909-925) */
static void split_src(struct mafComp *c)
{
    const char *dot = strchr(c->src, '.');
    size_t n = dot ? (size_t)(dot - c->src) : strlen(c->src);
    c->name = (char *)mz_xmalloc(n + 1);
    memcpy(c->name, c->src, n); c->name[n] = 0;
    c->contig = mz_xstrdup((dot && dot[1]) ? dot + 1 : c->src);
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
    head = mz_xstrdup(in->line);
    a = (struct mafAli *)mz_xmalloc(sizeof *a);
    memset(a, 0, sizeof *a);
    while ((len = in_maf_line(in)) != -1 && in->line[0] != '\n' && in->line[0] != ' ' && in->line[0] != '#') {
        char *src, *text, *name;
        const char *dot;
        struct mafComp t;
        size_t ns;
        if (in->line[0] != 's') continue;                  /* i / e / q lines are ignored */
        if (in->tmpcap < 3 * ((size_t)len + 1)) {           /* the fields of a line, parsed into the reader's own room: the row is ONE allocation (mz_maf.c) */
            in->tmpcap = 3 * ((size_t)len + 1) + 256;
            free(in->tmp);
            in->tmp = (char *)mz_xmalloc(in->tmpcap);
        }
        src = in->tmp; text = src + len + 1; name = text + len + 1;
        memset(&t, 0, sizeof t);
        if (sscanf(in->line, "s %s %d %d %c %d %s", src, &t.start, &t.size, &t.strand, &t.srcSize, text) != 6)
            mz_fatalf("bad component in file %s, line %d:\n%s", in->name, in->line_nbr, src);
        dot = strchr(src, '.');                             /* species and contig parts of "species.contig" (split_src) */
        ns = dot ? (size_t)(dot - src) : strlen(src);
        memcpy(name, src, ns); name[ns] = 0;
        t.src = src; t.name = name; t.contig = (char *)((dot && dot[1]) ? dot + 1 : src);
        t.paralog = 's';
        c = mzi_row_new(&t, (long)strlen(text));
        strcpy(c->text, text);
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

/* every block of an open MAF stream (header line first), in order; comment lines go to `echo` when verbose */
struct mafAli *mz_maf_read_stream(FILE *fp, const char *name, int verbose, FILE *echo)
{
    maf_in in;
    struct mafAli *first = NULL, *last = NULL, *a;
    char buf[500];
    int version;

    memset(&in, 0, sizeof in);
    in.name = name; in.verbose = verbose; in.fp = fp; in.echo = echo;
    if (!fgets(buf, sizeof buf, in.fp)) mz_fatalf("empty file %s", name);
    if (sscanf(buf, "##maf version=%d", &version) != 1) mz_fatalf("improper maf header line: %s", buf);
    while ((a = maf_next(&in)) != NULL) {
        if (last) last->next = a; else first = a;
        last = a;
    }
    free(in.line); free(in.tmp);
    return first;
}

struct mafAli *mz_maf_read_all(const char *path, int verbose)
{
    struct mafAli *list;
    FILE *fp = fopen(path, "r");
    if (!fp) mz_fatalf("Cannot open %s.", path);
    list = mz_maf_read_stream(fp, path, verbose, NULL);
    fclose(fp);
    return list;
}

/* the same over MAF text in memory (the in-process tree driver hands blocks from one program to the next this way) */
struct mafAli *mz_maf_read_mem(const char *text, size_t len, const char *name)
{
    struct mafAli *list;
    FILE *fp;
    if (len == 0) mz_fatalf("empty file %s", name);
    fp = fmemopen((void *)text, len, "r");
    if (!fp) mz_fatalf("Cannot open %s.", name);
    list = mz_maf_read_stream(fp, name, 0, NULL);
    fclose(fp);
    return list;
}

/* ------------------------------------------------------------------------------------------------ blocks handed on without text
 * What a block IS after mafWrite() printed it and the reader above read it again -- the tree driver hands blocks from one step to the
 * next as lists (mz_roast.c), and the next step must see what it would have parsed: the score with one decimal ("%3.1f", then atof), a
 * source "x.x" printed as "x" (mafWrite's printable_src, reference maf.c:283-288) and split again.  Everything else a row line carries
 * (start, size, strand, srcSize, text, the amplifier / copy marks of the 'a' line) survives the round trip as it is. */
void mz_ali_as_reread(struct mafAli *a)
{
    struct mafComp *c;
    /* (a whole number of moderate size prints as itself and ".0": most scores are sums of integer column scores) */
    if (a->score != (double)MIN_INT && !(a->score == (double)(long long)a->score && a->score > -1e15 && a->score < 1e15 && a->score != 0.0)) {
        char t[400]; snprintf(t, sizeof t, "%3.1f", a->score); a->score = atof(t);
    }
    a->chain_len = 0;
    for (c = a->components; c; c = c->next) {
        const char *dot = strchr(c->src, '.');
        int changed = 0;
        if (dot && dot[1] && (size_t)(dot - c->src) == strlen(dot + 1) && strncmp(c->src, dot + 1, (size_t)(dot - c->src)) == 0) {
            c->src[dot - c->src] = 0; changed = 1;                               /* "x.x" comes back as "x" */
        } else if (dot && !dot[1]) {
            c->src[dot - c->src] = 0; changed = 1;                               /* "x." is printed as "x" */
        }
        if (changed || !c->name || !c->contig) {                                 /* (otherwise name and contig are this src's already: they are
                                                                                  * copied with it from rows the reader split) */
            mzi_row_free_field(c, c->name); mzi_row_free_field(c, c->contig);
            split_src(c);
        }
        if (c->mafPosMap && c->mafPosMap != (int *)c) { free(c->mafPosMap); c->mafPosMap = NULL; }
        c->nameID = 0;
    }
}

/* a deep copy on the heap (mafAliFree() releases it), whatever owns the original */
struct mafAli *mz_ali_copy(const struct mafAli *a)
{
    struct mafAli *d = (struct mafAli *)mz_xmalloc(sizeof *d);
    struct mafComp *c, *tail = NULL;
    memset(d, 0, sizeof *d);
    d->score = a->score; d->textSize = a->textSize;
    for (c = a->components; c; c = c->next) {
        struct mafComp *nc = mzi_row_new(c, a->textSize);
        memcpy(nc->text, c->text, (size_t)a->textSize);
        if (tail) tail->next = nc; else d->components = nc;
        tail = nc;
    }
    return d;
}

/* ------------------------------------------------------------------------------------------------ list helpers */

struct mafAli *mz_pop_first(struct mafAli **head)
{
    struct mafAli *a = *head;
    if (a) { *head = a->next; a->next = NULL; }
    return a;
}

/* move every block whose top row lies on `chr` from *from to the tail of *to, keeping the order */
/* move every block whose top row lies on `chr` from *from to the tail of *to, keeping the order */
void mz_take_chr(struct mafAli **from, struct mafAli **to, const char *chr)
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

