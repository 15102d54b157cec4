/* mz_py.h -- pre_yama() in stages (internal to libmzamd; see mz_preyama.c, mz_multiz.c) */
#ifndef MZAMD_MZ_PY_H
#define MZAMD_MZ_PY_H

#include <stdio.h>
#include "../../include/maf.h"
#include "../../include/mz_amd.h"

enum { MZ_PY_NULL = 0, MZ_PY_JOB = 1, MZ_PY_DONE = 2 };

typedef struct mz_py {
    struct mafAli *a1, *a2;          /* borrowed: must outlive the last stage */
    int radius, v, stage;
    int borrowed;                    /* 1: the merged columns handed to mz_py_step() stay the caller's (inside a result block of
                                        mz_yama_batch(), released with mz_free_outs() after the last stage); 0: they are a
                                        malloc()ed block of their own and this structure frees it */
    int K, L, M, N, M_all, N_all, cbeg1, cbeg2, cend2;
    unsigned char **A, **B, **merged, **ref1, **ref2, **merged2;
    int *map1, *map2, *LB, *RB;
    mz_job job;                      /* the yama() call to make next (valid after MZ_PY_JOB) */
} mz_py;

int  mz_py_begin(mz_py *p, struct mafAli *a1, struct mafAli *a2, int beg, int end, int radius, int v, FILE *fpw2);
int  mz_py_step(mz_py *p, unsigned char *flat, int om, struct mafAli **result);
void mz_py_free(mz_py *p);

#endif
