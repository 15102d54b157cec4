/* include/maf.h -- the MAF block/row structures at the drop-in boundary.
 *
 * Layout-compatible restatement of the reference's data model (reference maf.h:13-57); the
 * struct layouts ARE the ABI of pre_yama() (x86-64 offsets
 * mafAli{next@0, score@8, components@16, textSize@24, chain_len@28} = 32 bytes,
 * mafComp{next@0, name@8, src@16, text@24, contig@32, mafPosMap@40, srcSize@48, start@52,
 * size@56, nameID@60, strand@62, paralog@63} = 64 bytes).  tests/test_abi.py checks them.
 * A maintainer building the stock drivers keeps using the reference's own maf.h; this one
 * exists so the shim and its tests compile without the reference tree.
 */
#ifndef MZAMD_MAF_H
#define MZAMD_MAF_H

#include <stdio.h>

/* "no score" sentinel, reference maf.h:10-11 */
#define MAX_INT ((int)(~(1u << (sizeof(int) * 8 - 1))))
#define MIN_INT ((int)(1u << (sizeof(int) * 8 - 1)))

struct mafAli;
struct mafComp;

struct mafFile {                 /* reference maf.h:13-24 */
    struct mafFile *next;
    int version;
    char *scoring;
    struct mafAli *alignments;
    char *fileName;
    int line_nbr;
    int verbose;
    FILE *fp;
};

struct mafAli {                  /* reference maf.h:29-37: one alignment block */
    struct mafAli *next;
    double score;
    struct mafComp *components;  /* rows, top row = reference sequence */
    int textSize;                /* columns */
    int chain_len;
};

struct mafComp {                 /* reference maf.h:42-57: one row of a block */
    struct mafComp *next;
    char *name;                  /* species part of src */
    char *src;                   /* species.chrom */
    char *text;                  /* textSize characters, '-' for gaps */
    char *contig;
    int *mafPosMap;
    int srcSize;
    int start;                   /* 0-based; relative to the reverse strand if strand == '-' */
    int size;                    /* non-dash characters in text */
    short nameID;
    char strand;
    char paralog;
};

/* the helpers of the reference's maf.c / multi_util.c that pre_yama() depends on; libmzamd.so
 * carries its own restatements so that it is self-contained (an executable that also links the
 * reference's maf.o / multi_util.o simply keeps using those). */
struct mafComp *mafCpyComp(struct mafComp *c);                      /* reference maf.c:437-451 */
void mafCompFree(struct mafComp **pc);                              /* reference maf.c:305-319 */
void mafAliFree(struct mafAli **pa);                                /* reference maf.c:321-337 */
void mafWrite(FILE *f, struct mafAli *a);                           /* reference maf.c:251-294 */
int  mafPos2Col(struct mafComp *c, int pos, int textSize);          /* reference multi_util.c:633-645 */
struct mafAli *mafColDashRm(struct mafAli *a);                      /* reference maf.c:339-377 */
struct mafAli *make_part_ali_col(struct mafAli *a, int cbeg, int cend);   /* reference multi_util.c:570-618 */
int  print_part_ali_col(struct mafAli *a, int cbeg, int cend, FILE *fp);  /* reference multi_util.c:620-629 */
struct mafAli *mafRowDashRm(struct mafAli *a);                      /* reference maf.c:384-417 */
struct mafAli *make_part_ali(struct mafAli *a, int cbeg, int cend); /* reference maf.c:488-523 */

#endif
