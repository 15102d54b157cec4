/* mz_drivers.h -- internal: what the batched command-line drivers (mz_multiz.c, mz_multic.c) share:
 * the MAF reader and list helpers of mz_mafio.c, the staged pre_yama() of mz_py.h, small utilities. */
#ifndef MZAMD_MZ_DRIVERS_H
#define MZAMD_MZ_DRIVERS_H

#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <time.h>
#include <malloc.h>
#include "../../include/maf.h"
#include "../../include/mz_scores.h"
#include "../../include/mz_multiz.h"
#include "mz_py.h"

__attribute__((noreturn)) void mz_fatalf(const char *fmt, ...);
__attribute__((noreturn)) void mz_fatal_status(const mz_job *j, const mz_out *o);
extern int row2;
extern char *argv0;

#ifndef MZ_STAGE_THREADS
#define MZ_STAGE_THREADS 32     /* host threads of the per-merge stages (allocation-heavy: more does not help) */
#endif
#define MERGE_FAILED 99         /* merge state: yama() refused the job (beside the MZ_PY_* states) */

/* MZ_TIMING=1: phase times of a run on stderr (mz_host.c: parsed once; "0" is quiet) */
__attribute__((visibility("hidden"))) int mzi_timing(void);
static inline double mz_now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* mz_maf.c: rows in one allocation */
__attribute__((visibility("hidden"))) struct mafComp *mzi_row_new(const struct mafComp *t, long text_len);
__attribute__((visibility("hidden"))) void mzi_row_free_field(struct mafComp *c, void *p);
/* mz_mafio.c */
struct mafAli *mz_maf_read_stream(FILE *fp, const char *name, int verbose, FILE *echo);
struct mafAli *mz_maf_read_mem(const char *text, size_t len, const char *name);
void mz_ali_as_reread(struct mafAli *a);                 /* the block as the reader would see it after mafWrite() */
struct mafAli *mz_ali_copy(const struct mafAli *a);      /* deep heap copy */
/* The blocks of a long list, in the list's order, BESIDE the list (p == NULL: not known).  Between the steps of the tree driver a level's
 * blocks -- hundreds of thousands -- go from replay to projection to the next walk: whoever makes such a list knows its blocks and
 * hands them on, so that nobody has to walk the list (a cache miss per block) to find them again.  An index belongs to one list as it
 * stands: whoever changes the list drops the index (mz_blocks_drop). */
typedef struct { struct mafAli **p; int n, cap; } mz_blocks;
static inline void mz_blocks_drop(mz_blocks *b) { if (b) { free(b->p); b->p = NULL; b->n = b->cap = 0; } }
static inline void mz_blocks_push(mz_blocks *b, struct mafAli *a)
{
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 1024;
        b->p = (struct mafAli **)realloc(b->p, (size_t)b->cap * sizeof *b->p);
        if (!b->p) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)((size_t)b->cap * sizeof *b->p));
    }
    b->p[b->n++] = a;
}
/* mz_project.c */
struct mafAli *mz_project_lists(struct mafAli *all, const char *target, struct mafAli **others);
/* the same with the index of `all` (consumed; NULL or empty: not known) going in and the result's coming out (left empty for a short list) */
__attribute__((visibility("hidden"))) struct mafAli *mzi_project_blocks(struct mafAli *all, mz_blocks *idx, const char *target, struct mafAli **others);
/* mz_multiz.c */
__attribute__((visibility("hidden"))) void mzi_multiz_finish_tails(struct mz_mzrun *run, struct mafAli **out, struct mafAli **out1, struct mafAli **out2, struct mafAli **tails, mz_blocks *idx);
__attribute__((visibility("hidden"))) struct mz_mzrun *mzi_multiz_prepare_blocks(struct mafAli **list1, struct mafAli **list2, mz_blocks *idx1, mz_blocks *idx2, int v, int radius,
                                                                               int min_output_wid, int has_out1, int has_out2);
/* mz_multic.c */
int mz_multic_lists(struct mafAli **list1, struct mafAli **list2, int v, int radius, int minw, int cate,
                    FILE *out, FILE *out1, FILE *out2);
void *mz_xmalloc(size_t n);
/* append a result block of mz_yama_batch() (mz_out.block) to a list that is freed after the last wave of yama() calls */
static inline void **mz_hold(void **list, int *n, void *block)
{
    if ((*n & (*n + 1)) == 0) {                          /* 0, 1, 3, 7, ...: room for twice as many and one */
        list = (void **)realloc(list, (size_t)(2 * *n + 2) * sizeof *list);
        if (!list) mz_fatalf("Ran out of memory trying to allocate %lu.", (unsigned long)((2 * *n + 2) * sizeof *list));
    }
    list[(*n)++] = block;
    return list;
}
char *mz_xstrdup(const char *s);
struct mafAli *mz_pop_first(struct mafAli **head);
void mz_take_chr(struct mafAli **from, struct mafAli **to, const char *chr);   /* blocks on contig chr, order kept */
void mz_tune_malloc(void);                                                     /* heap growth in big steps */

#endif
