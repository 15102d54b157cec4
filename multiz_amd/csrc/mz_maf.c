/* mz_maf.c -- the few MAF block helpers that pre_yama()/mafBuild() lean on, restated so that
 * libmzamd.so is self-contained (reference maf.c:251-377,437-451; multi_util.c:570-645,889-906).
 * An executable that also links the reference's own maf.o / multi_util.o keeps using those
 * (its definitions take precedence over a shared library's).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <malloc.h>
#include "../../include/maf.h"
#include "../../include/mz_scores.h"

__attribute__((noreturn)) void mz_fatalf(const char *fmt, ...);

int row2 = 0;      /* reference multi_util.c:24: "only print blocks with >= 2 rows" switch of the drivers */

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)n);
    return p;
}

static char *dup_or_null(const char *s)
{
    char *p;
    if (!s) return NULL;
    p = (char *)xmalloc(strlen(s) + 1);
    return strcpy(p, s);
}

/* a fresh row with the bookkeeping fields of t and no text yet (reference maf.c:437-451) */
struct mafComp *mafCpyComp(struct mafComp *t)
{
    struct mafComp *c = (struct mafComp *)xmalloc(sizeof *c);
    memset(c, 0, sizeof *c);
    c->src = dup_or_null(t->src);
    c->name = dup_or_null(t->name);
    c->contig = dup_or_null(t->contig);
    c->srcSize = t->srcSize;
    c->start = t->start;
    c->size = t->size;
    c->strand = t->strand;
    c->paralog = t->paralog;
    return c;
}

/* ROWS IN ONE ALLOCATION.  The drivers of this library (mz_multiz.c, mz_roast.c ...) make and drop rows by the ten million, and a row
 * of the reference's making is five allocations: the struct, src, name, contig and the text.  Rows the drivers make for themselves
 * (mzi_row_new) are ONE: the struct with the three names and, where its length is known, the text behind it -- marked by mafPosMap
 * pointing at the row itself (no position map can live there).  mafCompFree() knows the mark; code that replaces a field of a row
 * asks mzi_row_free_field() to free the old value, which does nothing for a value that lives inside the row's allocation.  Rows that
 * leave the library through the drop-in boundary (pre_yama()'s result: mz_preyama.c) are made the reference's way, by mafCpyComp():
 * the caller frees them with its own mafAliFree(). */
static int row_inside(const struct mafComp *c, const void *p)
{
    return c->mafPosMap == (const int *)c && (const char *)p >= (const char *)(c + 1) &&
           (const char *)p < (const char *)c + malloc_usable_size((void *)c);
}
__attribute__((visibility("hidden"))) void mzi_row_free_field(struct mafComp *c, void *p)
{
    if (p && !row_inside(c, p)) free(p);
}
/* a fresh row with the bookkeeping fields and names of t; text_len >= 0: room for that much text (and its 0) behind the names,
 * text pointing at it; < 0: no text yet (the caller sets one of its own, freed with the row) */
__attribute__((visibility("hidden"))) struct mafComp *mzi_row_new(const struct mafComp *t, long text_len)
{
    const size_t ns = t->src ? strlen(t->src) + 1 : 0, nn = t->name ? strlen(t->name) + 1 : 0, nc = t->contig ? strlen(t->contig) + 1 : 0;
    struct mafComp *c = (struct mafComp *)xmalloc(sizeof *c + ns + nn + nc + (text_len >= 0 ? (size_t)text_len + 1 : 0));
    char *p = (char *)(c + 1);
    memset(c, 0, sizeof *c);
    if (t->src) { c->src = (char *)memcpy(p, t->src, ns); p += ns; }
    if (t->name) { c->name = (char *)memcpy(p, t->name, nn); p += nn; }
    if (t->contig) { c->contig = (char *)memcpy(p, t->contig, nc); p += nc; }
    if (text_len >= 0) { c->text = p; p[text_len] = 0; }
    c->srcSize = t->srcSize;
    c->start = t->start;
    c->size = t->size;
    c->strand = t->strand;
    c->paralog = t->paralog;
    c->mafPosMap = (int *)c;
    return c;
}

void mafCompFree(struct mafComp **pc)
{
    struct mafComp *c = *pc;
    if (!c) return;
    if (c->mafPosMap == (int *)c) {
        mzi_row_free_field(c, c->src); mzi_row_free_field(c, c->text); mzi_row_free_field(c, c->contig); mzi_row_free_field(c, c->name);
    } else { free(c->src); free(c->text); free(c->contig); free(c->name); free(c->mafPosMap); }
    free(c);
    *pc = NULL;
}

void mafAliFree(struct mafAli **pa)
{
    struct mafComp *c, *nx;
    if (!pa || !*pa) return;
    for (c = (*pa)->components; c; c = nx) { nx = c->next; mafCompFree(&c); }
    free(*pa);
    *pa = NULL;
}

static int digits10(int x)
{
    int n = 1;
    if (x < 0) mz_fatalf("digitsBaseTen: negative argument %d", x);
    while (x >= 10) { x /= 10; ++n; }
    return n;
}

/* the source name as the reference re-assembles it for printing (multi_util.c:889-906,
 * maf.c:283-288): <part before the first '.'> '.' <rest>; the rest is dropped when it is empty
 * or equal to the first part (so "x.x" prints as "x") */
static void printable_src(const char *src, char *out, size_t cap)
{
    const char *dot = strchr(src, '.');
    size_t n = dot ? (size_t)(dot - src) : strlen(src);
    const char *rest = (dot && dot[1] != '\0') ? dot + 1 : NULL;
    if (n >= cap) n = cap - 1;
    memcpy(out, src, n);
    out[n] = '\0';
    if (rest && strcmp(out, rest) != 0)
        snprintf(out + n, cap - n, ".%s", rest);
}

/* one block in MAF text, column-aligned the way the reference writer does it (maf.c:251-294): the block is put together in one
 * buffer -- the fields laid down by hand, what "s %-*s %*d %*d %c %*d %s\n" prints -- and written with one call (a guide-tree run
 * prints thirteen million rows at its destination alone; a formatted print per row was most of that time) */
static char *put_right(char *p, int width, int v)          /* %*d of a non-negative number */
{
    char tmp[12];
    int n = 0, k;
    unsigned u = (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    for (k = n; k < width; ++k) *p++ = ' ';
    while (n) *p++ = tmp[--n];
    return p;
}

void mafWrite(FILE *f, struct mafAli *a)
{
    struct mafComp *c;
    int wsrc = 0, wstart = 0, wsize = 0, wsrcsize = 0, row = 0, rows = 0;
    size_t need = 512, text = 0;
    char name[512], stack[16384], *buf, *p;

    for (c = a->components; c; c = c->next) {
        int n = (int)strlen(c->src);
        if (n > wsrc) wsrc = n;
        text += strlen(c->text);
        ++rows;
    }
    /* (the widths of the numbers are taken below, behind the header line: digits10() ends the program on a negative one, and the
     * stock writer has printed the header by then) */
    need += (size_t)rows * ((size_t)wsrc + 64) + text;
    buf = need <= sizeof stack ? stack : (char *)xmalloc(need);
    p = buf;
    *p++ = 'a';
    if (a->score != MIN_INT) p += sprintf(p, " score=%3.1f", a->score);
    for (c = a->components; c; c = c->next, ++row) {
        if (c->paralog == 'a') p += sprintf(p, " amplifier=%d", row);
        else if (c->paralog == 'c') p += sprintf(p, " copy=%d", row);
        else if (c->paralog != 's') { fwrite(buf, 1, (size_t)(p - buf), f); mz_fatalf("Wrong character: '%c'", c->paralog); }
    }
    *p++ = '\n';
    for (c = a->components; c; c = c->next) {
        int n;
        if (c->start < 0 || c->size < 0 || c->srcSize < 0) fwrite(buf, 1, (size_t)(p - buf), f);      /* (digits10() is about to end the program) */
        if ((n = digits10(c->start)) > wstart) wstart = n;
        if ((n = digits10(c->size)) > wsize) wsize = n;
        if ((n = digits10(c->srcSize)) > wsrcsize) wsrcsize = n;
    }
    for (c = a->components; c; c = c->next) {
        size_t n;
        printable_src(c->src, name, sizeof name);
        n = strlen(name);
        *p++ = 's'; *p++ = ' ';
        memcpy(p, name, n); p += n;
        for (; n < (size_t)wsrc; ++n) *p++ = ' ';
        *p++ = ' ';
        p = put_right(p, wstart, c->start); *p++ = ' ';
        p = put_right(p, wsize, c->size); *p++ = ' ';
        *p++ = c->strand; *p++ = ' ';
        p = put_right(p, wsrcsize, c->srcSize); *p++ = ' ';
        n = strlen(c->text);
        memcpy(p, c->text, n); p += n;
        *p++ = '\n';
    }
    *p++ = '\n';
    fwrite(buf, 1, (size_t)(p - buf), f);
    if (buf != stack) free(buf);
}

/* column (0-based) of sequence position pos in row c (reference multi_util.c:633-645) */
int mafPos2Col(struct mafComp *c, int pos, int textSize)
{
    int col, p = c->start - 1;
    if (pos < c->start || pos >= c->start + c->size)
        mz_fatalf("mafPos2Col: %d not in %d-%d", pos, c->start, c->start + c->size - 1);
    for (col = 0; col < textSize; ++col)
        if (c->text[col] != '-' && ++p == pos) break;
    return col;
}

/* squeeze out columns that are '-' in every row, in place (reference maf.c:357-382) */
struct mafAli *mafColDashRm(struct mafAli *a)
{
    struct mafComp *c;
    int keep = 0, col;
    if (!a) return NULL;
    for (col = 0; col < a->textSize; ++col) {
        for (c = a->components; c; c = c->next)
            if (c->text[col] != '-') break;
        if (!c) continue;
        if (keep < col)
            for (c = a->components; c; c = c->next) c->text[keep] = c->text[col];
        ++keep;
    }
    if (keep < a->textSize) {
        a->textSize = keep;
        for (c = a->components; c; c = c->next) c->text[keep] = '\0';
    }
    return a;
}

/* columns cbeg..cend of a block as a block of its own: rows without a base are dropped, starts are
 * advanced past the bases to the left, all-dash columns squeezed out, score recomputed
 * (reference multi_util.c:570-618) */
struct mafAli *make_part_ali_col(struct mafAli *ali, int cbeg, int cend)
{
    const int width = cend - cbeg + 1;
    struct mafAli *out;
    struct mafComp *c, *nc, *tail = NULL;

    if (width == 0) return NULL;
    out = (struct mafAli *)xmalloc(sizeof *out);
    memset(out, 0, sizeof *out);
    out->textSize = width;
    out->score = mafScoreRange(ali, cbeg, width);        /* also the reference's range check */
    for (c = ali->components; c; c = c->next) {
        int before = 0, bases = 0, i;
        for (i = 0; i < cbeg; ++i) before += c->text[i] != '-';
        for (i = cbeg; i <= cend; ++i) bases += c->text[i] != '-';
        if (bases == 0) continue;
        nc = mafCpyComp(c);
        nc->start = c->start + before;
        nc->size = bases;
        nc->text = (char *)xmalloc((size_t)width + 1);
        memcpy(nc->text, c->text + cbeg, (size_t)width);
        nc->text[width] = '\0';
        if (tail) tail->next = nc; else out->components = nc;
        tail = nc;
    }
    if (!out->components) { mafAliFree(&out); return NULL; }
    out = mafColDashRm(out);
    out->score = mafScoreRange(out, 0, out->textSize);
    return out;
}

int print_part_ali_col(struct mafAli *ali, int cbeg, int cend, FILE *fp)
{
    struct mafAli *part = make_part_ali_col(ali, cbeg, cend);
    if (part && (row2 == 0 || part->components->next != NULL)) mafWrite(fp, part);
    mafAliFree(&part);
    return 0;
}

/* drop rows that are '-' throughout; NULL (block freed) when none is left (reference maf.c:384-417) */
struct mafAli *mafRowDashRm(struct mafAli *a)
{
    struct mafComp **pp, *c;
    if (!a) return NULL;
    for (pp = &a->components; (c = *pp) != NULL; ) {
        const char *s = c->text;
        while (*s == '-') ++s;
        if (*s == '\0') { *pp = c->next; c->next = NULL; mafCompFree(&c); }
        else pp = &c->next;
    }
    if (!a->components) { mafAliFree(&a); return NULL; }
    return a;
}

/* columns cbeg..cend of every row, all-dash ROWS dropped, columns kept as they are (reference maf.c:488-523;
 * multic prints the unused stretches of its inputs with it) */
struct mafAli *make_part_ali(struct mafAli *ali, int cbeg, int cend)
{
    const int width = cend - cbeg + 1;
    struct mafAli *out = (struct mafAli *)xmalloc(sizeof *out);
    struct mafComp *c, *nc, *tail = NULL;
    memset(out, 0, sizeof *out);
    for (c = ali->components; c; c = c->next) {
        int before = 0, bases = 0, i;
        for (i = 0; i < cbeg; ++i) before += c->text[i] != '-';
        for (i = cbeg; i <= cend; ++i) bases += c->text[i] != '-';
        nc = mafCpyComp(c);
        nc->start = c->start + before;
        nc->size = bases;
        nc->text = (char *)xmalloc((size_t)width + 1);
        memcpy(nc->text, c->text + cbeg, (size_t)width);
        nc->text[width] = '\0';
        if (tail) tail->next = nc; else out->components = nc;
        tail = nc;
    }
    out->textSize = width;
    out = mafRowDashRm(out);
    if (out) out->score = mafScoreRange(out, 0, width);
    return out;
}
